"""(f32, u32) pairs and f32 -> u32 rank sorts beyond 2^28 elements: the route without a histogram (default, round 6: up to 2^29) against one pass
per column (RSX_NO_BLIND=1: what round 5 ran there).  python tools/pairs_big_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402

rsa.require_gpu()
for n in ((3 << 27), (1 << 29)):
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    work = torch.empty_like(keys)
    rsa.fill_splitmix(keys, seed=9)
    for what in ("ranks", "pairs"):
        for name, envs in (("without a histogram", {}), ("RSX_NO_BLIND=1", {"RSX_NO_BLIND": "1"})):
            os.environ.pop("RSX_NO_BLIND", None)
            os.environ.update(envs)
            rsa.reload_env()
            if what == "ranks":
                ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
            else:
                vals = torch.arange(n, dtype=torch.int32, device="cuda")
                vin, ka, va = torch.empty_like(vals), torch.empty_like(keys), torch.empty_like(vals)
            best = 1e9
            for r in range(5):
                work.copy_(keys)
                if what == "pairs":
                    vin.copy_(vals)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                if what == "ranks":
                    _, info = rsa.radix_sort_rank(work, ib, dtype=rsa.F32)
                else:
                    _, _, info = rsa.radix_sort_pairs(work, ka, vin, va, dtype=rsa.F32)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    best = min(best, e0.elapsed_time(e1))
            print("%-5s n = %10d  %-20s %.3f ms = %.1f Gkeys/s (route %d)" % (what, n, name, best, n / best / 1e6, info.hybrid), flush=True)
            if what == "ranks":
                del ib
            else:
                del vals, vin, ka, va
            torch.cuda.empty_cache()
os.environ.pop("RSX_NO_BLIND", None)
