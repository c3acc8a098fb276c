"""Where the histogram-less two-level route starts to pay, at the library's defaults: u32 and u64 keys of 48 Mi .. 128 Mi
elements, default (the route from 2^26 keys on) against RSX_NO_BLIND=1.  Best of 8 fresh sorts each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for dt, tdt in ((rsa.U32, torch.int32), (rsa.U64, torch.int64)):
    for n in (3 << 24, 7 << 23, 1 << 26, 5 << 24, 3 << 25, 1 << 27):
        bufs = [torch.empty(n, dtype=tdt, device="cuda") for _ in range(2)]
        aux = torch.empty(n, dtype=tdt, device="cuda")
        for name, envs in (("default", {}), ("from 2^25", {"RSX_BLIND_MIN_LOG2": "25"}), ("RSX_NO_BLIND=1", {"RSX_NO_BLIND": "1"})):
            for k in ("RSX_BLIND_MIN_LOG2", "RSX_NO_BLIND"):
                os.environ.pop(k, None)
            os.environ.update(envs)
            rsa.reload_env()
            best = 1e9
            for r in range(8):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 100 + r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, info = rsa.radix_sort(b, aux, dt)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print("%s n = %10d  %-16s %.3f ms  %.1f Gkeys/s (route %d)" % ("u32" if dt == rsa.U32 else "u64", n, name, best * 1e3, n / best / 1e9, info.hybrid), flush=True)
        del bufs, aux
