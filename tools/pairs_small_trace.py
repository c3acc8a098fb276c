"""Kernel trace target: rank and key + payload sorts of 2^24 and 2^25 pairs on the histogram-less two-level route (RSX_TWO_LEVEL_MIN_LOG2=24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSX_TWO_LEVEL_MIN_LOG2"] = "24"
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
src = torch.empty(n, dtype=torch.int32, device="cuda")
ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
for r in range(5):
    rsa.fill_splitmix(src, 100 + r)
    _, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
torch.cuda.synchronize()
print("route", info.hybrid)
