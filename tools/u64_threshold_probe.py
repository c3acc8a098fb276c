"""8-byte keys, 8 Mi .. 192 Mi: the library's default against the histogram-less two-level route forced from 2^23 keys
(RSX_BLIND_MIN_LOG2=23) and against RSX_NO_BLIND=1; uniform keys and keys & 0xFFFFFFFFFF (five kept columns).  Best of 6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for mask in (None, 0xFFFFFFFFFF):
    for n in (1 << 23, 1 << 24, 3 << 23, 1 << 25, 3 << 24, 1 << 26, 3 << 25, 1 << 27, 3 << 26):
        bufs = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(2)]
        aux = torch.empty(n, dtype=torch.int64, device="cuda")
        out = []
        for name, envs in (("default", {}), ("from 2^23", {"RSX_BLIND_MIN_LOG2": "23"}), ("RSX_NO_BLIND=1", {"RSX_NO_BLIND": "1"})):
            for k in ("RSX_BLIND_MIN_LOG2", "RSX_NO_BLIND"):
                os.environ.pop(k, None)
            os.environ.update(envs)
            rsa.reload_env()
            best = 1e9
            for r in range(6):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 100 + r)
                if mask is not None:
                    b &= mask
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, info = rsa.radix_sort(b, aux, rsa.U64)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out.append("%s %.3f ms (route %d)" % (name, best * 1e3, info.hybrid))
        print("u64%s n = %4d Mi: %s" % (" & 0xFFFFFFFFFF" if mask else "", n >> 20, "   ".join(out)), flush=True)
        del bufs, aux
