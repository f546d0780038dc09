"""FETCH_SIZE / WRITE_SIZE calibration on a known byte count (MI355X_MICROARCH.md, HBM section): a device copy
of a 64 MiB float4 buffer, repeated; run under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE)."""
import torch
a = torch.rand(16 * 1024 * 1024, device="cuda")     # 64 MiB
b = torch.empty_like(a)
for _ in range(20):
    b.copy_(a)
torch.cuda.synchronize()
print("copied", a.numel() * 4, "bytes x 20")
