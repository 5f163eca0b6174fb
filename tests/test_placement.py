"""jetson_slam_amd/placement.py against a faked sysfs tree: the NUMA node and node-local cores of a GPU, the cores a rank binds to (affinity mask
respected, ranks that share a node split its cores), cpulist parsing / formatting.  No GPU, no real sysfs."""
import os

from jetson_slam_amd import placement as pl


def _fake_sysfs(root, gpus, nodes):
    """gpus: {pci_addr: (numa_node, local_cpulist or None)}; nodes: {node: cpulist}"""
    for addr, (node, cl) in gpus.items():
        d = os.path.join(root, "bus", "pci", "devices", addr)
        os.makedirs(d)
        open(os.path.join(d, "numa_node"), "w").write("%d\n" % node)
        if cl is not None:
            open(os.path.join(d, "local_cpulist"), "w").write(cl + "\n")
    for node, cl in nodes.items():
        d = os.path.join(root, "devices", "system", "node", "node%d" % node)
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cl + "\n")


def test_cpulist_round_trip():
    assert pl.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert pl.parse_cpulist("") == [] and pl.parse_cpulist("x,2-,5") == [5]
    assert pl.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert pl.parse_cpulist(pl.format_cpulist(range(96, 192))) == list(range(96, 192))
    assert pl.pci_address(0, 0x1b, 0) == "0000:1b:00.0"


def test_eight_gpus_two_sockets(tmp_path):
    root = str(tmp_path)
    addrs = [pl.pci_address(0, 0x10 + 8 * i, 0) for i in range(8)]
    gpus = {a: (0 if i < 4 else 1, "0-95,192-287" if i < 4 else "96-191,288-383") for i, a in enumerate(addrs)}
    _fake_sysfs(root, gpus, {0: "0-95,192-287", 1: "96-191,288-383"})
    allowed = list(range(384))
    share = pl.ranks_sharing_node(addrs, root)
    assert share == [(4, 0), (4, 1), (4, 2), (4, 3), (4, 0), (4, 1), (4, 2), (4, 3)]
    seen = []
    for lr, a in enumerate(addrs):
        n, k = share[lr]
        info = pl.plan(a, allowed, k, n, root)
        assert info["numa_node"] == (0 if lr < 4 else 1) and len(info["cpus"]) == 48
        node_cpus = set(pl.parse_cpulist(gpus[a][1]))
        assert set(info["cpus"]) <= node_cpus                         # node-local cores only
        seen.append(set(info["cpus"]))
    for i in range(8):
        for j in range(i + 1, 8):
            assert not (seen[i] & seen[j])                            # ranks do not sit on top of each other


def test_affinity_mask_and_missing_information(tmp_path):
    root = str(tmp_path)
    a, b, c = pl.pci_address(0, 3, 0), pl.pci_address(0, 4, 0), pl.pci_address(0, 5, 0)
    _fake_sysfs(root, {a: (1, None), b: (-1, None), c: (0, "0-7")}, {0: "0-7", 1: "8-15"})
    # local_cpulist missing: the node's cpulist is used; the container's affinity mask (16 cores of one socket, say) is respected
    info = pl.plan(a, [10, 11, 12, 13, 40], 0, 1, root)
    assert info["numa_node"] == 1 and info["cpus"] == [10, 11, 12, 13]
    # no NUMA information at all: nothing to do, and it says so
    info = pl.plan(b, list(range(16)), 0, 1, root)
    assert info["cpus"] == [] and "note" in info
    # the mask holds none of the node's cores (a cgroup pinned to the other socket): leave the process alone
    info = pl.plan(c, [8, 9, 10], 0, 1, root)
    assert info["cpus"] == [] and "affinity mask" in info["note"]
    # a PCI address that does not exist
    assert pl.gpu_numa(pl.pci_address(0, 9, 0), root) == (-1, [])


def test_bind_applies_the_plan():
    allowed = sorted(os.sched_getaffinity(0))
    try:
        info = pl.bind({"cpus": allowed[:1]})
        assert info["bound"] and sorted(os.sched_getaffinity(0)) == allowed[:1] and info["cpus"] == str(allowed[0])
    finally:
        os.sched_setaffinity(0, set(allowed))
    assert pl.bind({"cpus": []})["bound"] is False
