"""A few sorts of 10^6 and 10^5 u32 keys for a kernel trace (where the 60-75 us of a medium-sized sort go)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import radix_sorting_amd as rsa
for n in (1000000, 100000):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    aux = torch.empty_like(src)
    for rep in range(5):
        rsa.fill_splitmix(src, seed=rep + 1)
        torch.cuda.synchronize()
        rsa.radix_sort(src, aux, dtype=rsa.U32)
        torch.cuda.synchronize()
