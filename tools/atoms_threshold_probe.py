#!/usr/bin/env python3
"""From which size do the passes in whole 64-byte atoms (rsx_pass32a_kernel, rsx_pass16a_kernel) pay?  u32 keys, fresh unsorted
input per call, the blocking sort, best of a few; default against RSX_NO_PASS32A=1 and RSX_NO_PASS16A=1 (which implies the former)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa


def main():
    rsa.require_gpu()
    dev = torch.device("cuda:0")
    sizes = [int(x * (1 << 20)) for x in (40, 48, 52, 56, 64, 72, 80, 96, 112, 128, 160, 192, 256)]
    variants = [("default", {}), ("RSX_NO_PASS32A=1", {"RSX_NO_PASS32A": "1"}), ("RSX_NO_PASS16A=1", {"RSX_NO_PASS16A": "1"})]
    print("%12s " % "n" + " ".join("%18s" % v[0] for v in variants) + "   (us per sort)")
    for n in sizes:
        bufs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
        aux = torch.empty(n, dtype=torch.int32, device=dev)
        row = []
        for name, envs in variants:
            for k in ("RSX_NO_PASS32A", "RSX_NO_PASS16A"):
                os.environ.pop(k, None)
            os.environ.update(envs)
            rsa.reload_env()
            best = 1e9
            for r in range(8):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 1000 + r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, info = rsa.radix_sort(b, aux, rsa.U32)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row.append("%12.1f (r%d)" % (best * 1e6, info.hybrid))
        print("%12d " % n + " ".join("%18s" % x for x in row), flush=True)


if __name__ == "__main__":
    main()
