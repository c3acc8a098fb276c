"""Where the histogram-less two-level route starts to pay: 2^25 .. 2^27 u32 keys, with the route forced on from 2^25
(RSX_TWO_LEVEL_MIN_LOG2=25) against one pass per kept column (RSX_NO_HYBRID=1).  Best of 8 fresh sorts each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for n in (1 << 25, 3 << 24, 1 << 26, 3 << 25, 1 << 27):
    bufs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    aux = torch.empty(n, dtype=torch.int32, device="cuda")
    for name, envs in (("blind from 2^25", {"RSX_TWO_LEVEL_MIN_LOG2": "25"}), ("histogram first, from 2^25", {"RSX_TWO_LEVEL_MIN_LOG2": "25", "RSX_NO_BLIND": "1"}),
                       ("one pass per column", {"RSX_NO_HYBRID": "1"})):
        for k in ("RSX_TWO_LEVEL_MIN_LOG2", "RSX_NO_BLIND", "RSX_NO_HYBRID"):
            os.environ.pop(k, None)
        os.environ.update(envs)
        rsa.reload_env()
        best = 1e9
        for r in range(8):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort(b, aux, rsa.U32)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("n = %10d  %-28s %.3f ms  %.1f Gkeys/s (route %d)" % (n, name, best * 1e3, n / best / 1e9, info.hybrid), flush=True)
    del bufs, aux
