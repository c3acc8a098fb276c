"""Rank sorts (f32 keys -> u32 ranks) and key + payload sorts of 256 Ki .. 12 Mi elements at the library's defaults: where one MSB
pass + the pairs' leaves (route 1) hands over to one pass per column (route 0) and to the route without a histogram (5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for n in (1 << 18, 1 << 19, 1 << 20, 5 << 18, 3 << 19, 1 << 21, 3 << 20, 1 << 22, 6 << 20, 1 << 23, 12 << 20):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    k1 = torch.empty_like(src); v0 = torch.empty_like(src); v1 = torch.empty_like(src)
    rsa.reload_env()
    br = bp = 1e9
    for r in range(12):
        rsa.fill_splitmix(src, 100 + r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
        torch.cuda.synchronize()
        br = min(br, time.perf_counter() - t0)
    for r in range(12):
        rsa.fill_splitmix(src, 200 + r); rsa.fill_splitmix(v0, 300 + r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, pinfo = rsa.radix_sort_pairs(src, k1, v0, v1, dtype=rsa.F32)
        torch.cuda.synchronize()
        bp = min(bp, time.perf_counter() - t0)
    print("n = %9d  rank %.1f us (route %d)   pairs %.1f us (route %d)" % (n, br * 1e6, info.hybrid, bp * 1e6, pinfo.hybrid), flush=True)
