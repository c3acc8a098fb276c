"""8 Mi .. 64 Mi u32 keys (the reference's own headline, 4*10^7 keys, among them): the library's default route against the
sort without a histogram forced from 2^22 keys on (RSX_BLIND_MIN_LOG2=22) and against one pass per column (RSX_NO_HYBRID=1).
Best of 10 fresh sorts each, wall clock around one blocking sort.  `--trace n`: sorts of n keys only, for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
trace = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "--trace" else 0
sizes = [trace] if trace else [1 << 23, 10000000, 3 << 22, 1 << 24, 3 << 23, 1 << 25, 40000000, 3 << 24, 7 << 23, 1 << 26]
variants = (("default", {}), ("no histogram from 2^22", {"RSX_BLIND_MIN_LOG2": "22"}), ("RSX_NO_HYBRID=1", {"RSX_NO_HYBRID": "1"}))
if trace:
    variants = variants[1:2]
for n in sizes:
    bufs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    aux = torch.empty(n, dtype=torch.int32, device="cuda")
    row = []
    for name, envs in variants:
        for k in ("RSX_BLIND_MIN_LOG2", "RSX_NO_HYBRID"):
            os.environ.pop(k, None)
        os.environ.update(envs)
        rsa.reload_env()
        best = 1e9
        for r in range(10):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort(b, aux, rsa.U32)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append("%s %.1f us (route %d)" % (name, best * 1e6, info.hybrid))
    print("n = %9d  " % n + "   ".join(row), flush=True)
    del bufs, aux
