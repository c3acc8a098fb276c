"""One cfg 4 (ii) rank sort (2^28 f32 uniform in [-1, 1) -> u32 ranks) for a kernel trace: which pass costs what."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import radix_sorting_amd as rsa
n = 1 << 28
t = torch.empty(n, dtype=torch.int64, device="cuda")
rsa.fill_splitmix(t, seed=6)
f = (((t >> 40) & 0xFFFFFF) - (1 << 23)).to(torch.float32) * (2.0 ** -23)
del t
ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
for _ in range(2):
    ranks, info = rsa.radix_sort_rank(f, ib, dtype=rsa.F32)
torch.cuda.synchronize()
print(info.ncols, info.kept_columns())
