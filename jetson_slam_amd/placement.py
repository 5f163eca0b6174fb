"""Rank placement for the multi-GPU batch mode: every rank runs next to ITS GPU.

The north-star regime streams images from pinned host memory (2 x H x W bytes per pair; ~52 GB/s per GPU over PCIe gen 5 at the EuRoC shape): at
8 ranks that is > 400 GB/s of host DRAM reads, and a rank whose pinned buffers live on the other socket pulls them over the inter-socket
link.  So, before it allocates anything, a rank (1) finds the NUMA node of its GPU from sysfs (`numa_node` / `local_cpulist` of the GPU's PCI
device), (2) restricts itself to that node's cores with `sched_setaffinity` - threads and, by first touch, the pinned pages that are allocated
afterwards then live on that node - and (3) reports what it did.  Nothing here is specific to ROCm: the PCI address comes from the caller (torch's
device properties in bench.py), the sysfs root can be replaced (tests/test_placement.py runs this against a faked tree).

The reference has no counterpart (one GPU, device 0 hard-wired: orb_gpu.cpp:32)."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]  (the kernel's cpulist format; empty / malformed parts are skipped)"""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        try:
            if "-" in part:
                a, b = part.split("-", 1)
                cpus.extend(range(int(a), int(b) + 1))
            else:
                cpus.append(int(part))
        except ValueError:
            continue
    return sorted(set(cpus))


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'"""
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def pci_address(domain, bus, device, function=0):
    return "%04x:%02x:%02x.%x" % (domain, bus, device, function)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa(pci_addr, sysfs_root="/sys"):
    """(numa_node, cpus) of the PCI device: numa_node is -1 when the platform reports none; cpus is the device's local_cpulist, else the
    node's cpulist, else [] (unknown)."""
    dev = os.path.join(sysfs_root, "bus", "pci", "devices", pci_addr)
    node_txt = _read(os.path.join(dev, "numa_node"))
    try:
        node = int(node_txt) if node_txt is not None else -1
    except ValueError:
        node = -1
    cpus = parse_cpulist(_read(os.path.join(dev, "local_cpulist")) or "")
    if not cpus and node >= 0:
        cpus = parse_cpulist(_read(os.path.join(sysfs_root, "devices", "system", "node", "node%d" % node, "cpulist")) or "")
    return node, cpus


def plan(pci_addr, allowed, local_rank=0, ranks_on_node=1, sysfs_root="/sys"):
    """The cores a rank should bind to: the GPU's node-local cores that the process may use (`allowed` = its current affinity mask).  When several
    ranks share one NUMA node (GPUs of one socket) the node's cores are divided among them by local rank, so that their host-side threads (the
    oracle's OpenMP team, the extractor threads) do not sit on top of each other.  Returns a dict that says what was found and what to do; `cpus`
    is empty when there is nothing to do (no NUMA information, or none of the node's cores is allowed)."""
    node, local = gpu_numa(pci_addr, sysfs_root)
    allowed = sorted(set(allowed))
    usable = [c for c in local if c in set(allowed)]
    info = {"pci": pci_addr, "numa_node": node, "node_cpus": format_cpulist(local), "allowed_cpus": len(allowed), "cpus": [], "bound": False}
    if not usable:
        info["note"] = "no NUMA information for the GPU" if not local else "none of the GPU's node-local cores is in the affinity mask"
        return info
    if ranks_on_node > 1:
        share = max(1, len(usable) // ranks_on_node)
        k = local_rank % ranks_on_node
        mine = usable[k * share:(k + 1) * share] if (k + 1) * share <= len(usable) else usable[-share:]
        usable = mine or usable
    info["cpus"] = usable
    return info


def ranks_sharing_node(pci_addrs, sysfs_root="/sys"):
    """for every GPU of the job (index = local rank): (how many of the job's GPUs share its NUMA node, its index among them)"""
    nodes = [gpu_numa(a, sysfs_root)[0] for a in pci_addrs]
    out = []
    for i, n in enumerate(nodes):
        same = [j for j, m in enumerate(nodes) if m == n and n >= 0]
        out.append((len(same), same.index(i)) if same else (1, 0))
    return out


def bind(info):
    """apply a plan: restrict the calling process (all future threads, and by first touch the pages it allocates from now on) to info['cpus']"""
    info.setdefault("bound", False)
    if info.get("cpus") and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, set(info["cpus"]))
            info["bound"] = True
        except OSError as e:
            info["note"] = "sched_setaffinity failed: %s" % e
    info["cpus"] = format_cpulist(info.get("cpus") or [])
    return info
