"""8-byte keys, 3 Mi .. 8 Mi: the library's default against the histogram-less route forced from 2^22 keys (RSX_BLIND_MIN_LOG2=22)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for mask in (None, 0xFFFFFFFFFF):
    for n in (3 << 20, 1 << 22, 5 << 20, 6 << 20, 7 << 20, 1 << 23):
        bufs = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(2)]
        aux = torch.empty(n, dtype=torch.int64, device="cuda")
        out = []
        for name, envs in (("default", {}), ("from 2^22", {"RSX_BLIND_MIN_LOG2": "22"})):
            os.environ.pop("RSX_BLIND_MIN_LOG2", None)
            os.environ.update(envs)
            rsa.reload_env()
            best = 1e9
            for r in range(12):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 100 + r)
                if mask is not None:
                    b &= mask
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, info = rsa.radix_sort(b, aux, rsa.U64)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out.append("%s %.1f us (route %d)" % (name, best * 1e6, info.hybrid))
        print("u64%s n = %d Mi: %s" % (" & 0xFFFFFFFFFF" if mask else "", n >> 20, "   ".join(out)), flush=True)
