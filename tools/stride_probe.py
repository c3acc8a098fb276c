"""The level-1 pass at 1.5 x 2^30 u32 keys (20 % slower than its neighbours, profiles/r05/l1_pass_at_1p5x2p30_probe.txt) with the 256
level-1 slots spaced k KiB further apart (RSX_CAP1_PAD_KIB): is it where the slots lie?  python tools/stride_probe.py n pad pad ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def main():
    rsa.require_gpu()
    n = int(sys.argv[1])
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    aux = torch.empty_like(src)
    for pad in sys.argv[2:]:
        os.environ["RSX_CAP1_PAD_KIB"] = pad
        rsa.reload_env()
        best = None
        for rep in range(4):
            rsa.fill_splitmix(src, seed=5 + rep)
            torch.cuda.synchronize()
            rsa.profile_begin()
            res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
            torch.cuda.synchronize()
            p = rsa.profile_end()
            if rep and (best is None or p.scatter_ms < best[0]):
                best = (p.scatter_ms, p.narrow_ms, p.leaf_ms, info.hybrid)
        print("n = %d, %6s KiB more per level-1 slot (route %d): level-1 pass %.3f ms = %.2f TB/s, level-2 %.3f, leaves %.3f" %
              (n, pad, best[3], best[0], n * 8 / best[0] / 1e9, best[1], best[2]), flush=True)


if __name__ == "__main__":
    main()
