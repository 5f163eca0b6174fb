"""Calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on a known byte count: a 1 GiB device-to-device copy (wide coalesced loads)
and a 1 GiB fill (writes only).  Run under: rocprofv3 --pmc FETCH_SIZE ...  /  --pmc WRITE_SIZE ..."""
import torch
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
a.fill_(7)
torch.cuda.synchronize()
b = torch.empty_like(a)
b.copy_(a)
torch.cuda.synchronize()
c = a.view(torch.float32).sum()
torch.cuda.synchronize()
print(float(c) != 0)
