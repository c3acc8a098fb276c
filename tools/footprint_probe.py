"""Device memory the library holds after one sort (hipMemGetInfo before the first call and after it), and the sort's time with
the level-1 slots in the caller's second buffer (default) and all in scratch memory (RSX_NO_AUX_SLOTS=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for dt, tdt, n in ((rsa.U32, torch.int32, 1 << 28), (rsa.U32, torch.int32, 40000000), (rsa.U64, torch.int64, 1 << 27), (rsa.U64, torch.int64, 1 << 28)):
    bufs = [torch.empty(n, dtype=tdt, device="cuda") for _ in range(2)]
    aux = torch.empty(n, dtype=tdt, device="cuda")
    for name, envs in (("slots in aux", {}), ("RSX_NO_AUX_SLOTS=1", {"RSX_NO_AUX_SLOTS": "1"})):
        os.environ.pop("RSX_NO_AUX_SLOTS", None)
        os.environ.update(envs)
        rsa.release_stream()
        rsa.reload_env()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        best = 1e9
        for r in range(8):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort(b, aux, dt)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        held = free0 - torch.cuda.mem_get_info()[0]
        print("%s n = %10d  %-20s %.3f ms (route %d)   library holds %.3f GiB = %.2f x the keys" % ("u32" if dt == rsa.U32 else "u64", n, name, best * 1e3, info.hybrid, held / 2**30, held / (n * (4 if dt == rsa.U32 else 8))), flush=True)
    del bufs, aux
