"""Per-class kernel times (rsx_profile, HIP events) of one keys-only u32 sort of n keys: python tools/big_profile.py n [n ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def main():
    rsa.require_gpu()
    for arg in sys.argv[1:]:
        n = int(arg)
        src = torch.empty(n, dtype=torch.int32, device="cuda")
        aux = torch.empty_like(src)
        best = None
        for rep in range(4):
            rsa.fill_splitmix(src, seed=5 + rep)
            torch.cuda.synchronize()
            rsa.profile_begin()
            res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
            torch.cuda.synchronize()
            p = rsa.profile_end()
            tot = p.scatter_ms + p.narrow_ms + p.leaf_ms + p.hist_ms
            if rep and (best is None or tot < best[0]):
                best = (tot, p.scatter_ms, p.narrow_ms, p.leaf_ms, p.hist_ms, info.hybrid)
        print("n = %d (route %d): level-1 / whole-key passes %.3f ms, narrowing pass %.3f, leaves %.3f, histogram %.3f; sum %.3f ms = %.1f Gkeys/s" %
              (n, best[5], best[1], best[2], best[3], best[4], best[0], n / best[0] / 1e6), flush=True)
        del src, aux
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
