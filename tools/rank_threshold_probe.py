"""Rank sorts (f32 keys -> u32 ranks) and key + payload sorts of 16 Mi .. 128 Mi elements: the library's default against the
histogram-less two-level route forced from 2^24 (RSX_TWO_LEVEL_MIN_LOG2=24).  Best of 6 fresh sorts each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for n in (1 << 23, 5 << 21, 3 << 22, 1 << 24, 3 << 23, 1 << 25):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    k1 = torch.empty_like(src); v0 = torch.empty_like(src); v1 = torch.empty_like(src)
    for name, envs in (("default", {}), ("two levels from 2^23", {"RSX_TWO_LEVEL_MIN_LOG2": "23"})):
        os.environ.pop("RSX_TWO_LEVEL_MIN_LOG2", None)
        os.environ.update(envs)
        rsa.reload_env()
        br = bp = 1e9
        for r in range(6):
            rsa.fill_splitmix(src, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
            torch.cuda.synchronize()
            br = min(br, time.perf_counter() - t0)
        for r in range(6):
            rsa.fill_splitmix(src, 200 + r); rsa.fill_splitmix(v0, 300 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, pinfo = rsa.radix_sort_pairs(src, k1, v0, v1, dtype=rsa.F32)
            torch.cuda.synchronize()
            bp = min(bp, time.perf_counter() - t0)
        print("n = %10d  %-22s rank %.3f ms (route %d)   pairs %.3f ms (route %d)" % (n, name, br * 1e3, info.hybrid, bp * 1e3, pinfo.hybrid), flush=True)
    del src, ib, k1, v0, v1
