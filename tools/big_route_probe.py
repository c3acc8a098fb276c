import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for lg, extra in ((29, 0), (29, 1 << 28), (28, 1 << 27), (28, 0), (27, 0)):
    n = (1 << lg) + extra
    bufs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    aux = torch.empty(n, dtype=torch.int32, device="cuda")
    for mode in ("default", "RSX_NO_BLIND", "RSX_NO_HYBRID"):
        if mode != "default":
            os.environ[mode] = "1"
        rsa.reload_env()
        best = 1e9
        for r in range(5):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort(b, aux, rsa.U32)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("n = %d  %-14s %.3f ms  %.1f Gkeys/s (route %d)" % (n, mode, best * 1e3, n / best / 1e9, info.hybrid))
        os.environ.pop("RSX_NO_HYBRID", None)
        os.environ.pop("RSX_NO_BLIND", None)
    del bufs, aux
