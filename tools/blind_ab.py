"""A/B inside one process (boxes differ by several percent): the sort of 2^28 uniform u32 keys under different settings of
an environment switch, kernel classes timed by the library's own events (rsx_profile).  python tools/blind_ab.py VAR v1 v2 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def main():
    rsa.require_gpu()
    var, vals = sys.argv[1], sys.argv[2:]
    n = 1 << 28
    src0 = torch.empty(n, dtype=torch.int32, device="cuda")
    rsa.fill_splitmix(src0, seed=3)
    aux = torch.empty_like(src0)
    src = torch.empty_like(src0)
    for rnd in range(3):
        for v in vals:
            os.environ[var] = v
            rsa.reload_env()
            tot = []
            for i in range(6):
                src.copy_(src0)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if i == 2:
                    rsa.profile_begin()
                a.record()
                res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
                b.record()
                torch.cuda.synchronize()
                if i >= 2:
                    tot.append(a.elapsed_time(b))
            p = rsa.profile_end()
            k = len(tot)
            print("%s=%s round %d: route %d, %.3f ms per sort; level-1 / whole-key passes %.3f, narrowing pass %.3f, leaves %.3f, histogram %.3f" %
                  (var, v, rnd, info.hybrid, sum(tot) / k, p.scatter_ms / k, p.narrow_ms / k, p.leaf_ms / k, p.hist_ms / k), flush=True)


if __name__ == "__main__":
    main()
