"""One line: 2^28 u64 keys & MASK sorted 10 times (after 2), per-class kernel ms from the library's profile.  python tools/cfg3_classes.py [mask]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402

rsa.require_gpu()
mask = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0xFFFFFFFFFF
n = 1 << 28
src = torch.empty(n, dtype=torch.int64, device="cuda")
aux = torch.empty_like(src)
for r in range(12):
    rsa.fill_splitmix(src, seed=70 + r, mask=mask)
    torch.cuda.synchronize()
    if r == 2:
        rsa.profile_begin()
    _, info = rsa.radix_sort(src, aux, dtype=rsa.U64)
torch.cuda.synchronize()
p = rsa.profile_end()
K = 10
print("mask %x route %d: level 1 %.3f ms  level 2 %.3f  leaves %.3f  sum %.3f" % (mask, info.hybrid, p.scatter_ms / K, p.narrow_ms / K, p.leaf_ms / K,
                                                                                 (p.scatter_ms + p.narrow_ms + p.leaf_ms + p.hist_ms) / K))
