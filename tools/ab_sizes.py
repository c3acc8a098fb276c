"""A/B of an environment switch over array sizes, inside one process (boxes differ by several percent): keys-only sorts of
uniform u32 or u64 keys, best of a few device-event timings per setting, settings alternating.
python tools/ab_sizes.py VAR v1 v2 u32|u64 n [n ...]      (n in Mi keys; AB_MASK=0x.. : keys & mask)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


MASK = int(os.environ.get("AB_MASK", "0"), 0)


def main():
    rsa.require_gpu()
    var, v1, v2, kind = sys.argv[1:5]
    dt, tdt = (rsa.U64, torch.int64) if kind == "u64" else (rsa.U32, torch.int32)
    for arg in sys.argv[5:]:
        n = int(float(arg) * (1 << 20))
        src = torch.empty(n, dtype=tdt, device="cuda")
        aux = torch.empty_like(src)
        best = {v1: 1e9, v2: 1e9}
        route = {}
        for rnd in range(3):
            for v in (v1, v2):
                os.environ[var] = v
                rsa.reload_env()
                for rep in range(4):
                    if MASK:
                        rsa.fill_splitmix(src, seed=11 + rep, mask=MASK)
                    else:
                        rsa.fill_splitmix(src, seed=11 + rep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    _, info = rsa.radix_sort(src, aux, dtype=dt)
                    e1.record()
                    torch.cuda.synchronize()
                    if rep:
                        best[v] = min(best[v], e0.elapsed_time(e1))
                    route[v] = info.hybrid
        print("%s %6.1f Mi keys: %s=%s %.3f ms (route %d)   %s=%s %.3f ms (route %d)   ratio %.3f" %
              (kind, n / (1 << 20), var, v1, best[v1], route[v1], var, v2, best[v2], route[v2], best[v2] / best[v1]), flush=True)
        del src, aux
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
