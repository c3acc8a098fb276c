#!/usr/bin/env python3
"""A loop of device-pointer sorts of one size (u32, fresh input per sort): run under `rocprofv3 --kernel-trace --stats` to
see what the kernels of a mid-size sort cost, or alone for the wall time.  Usage: mid_trace.py [log2n | n] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa

def main():
    rsa.require_gpu()
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = (1 << a) if a < 64 else a
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    bufs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
    aux = torch.empty(n, dtype=torch.int32, device=dev)
    best, tot = 1e9, 0.0
    how = None
    for r in range(reps):
        b = bufs[r & 1]
        rsa.fill_splitmix(b, 1000 + r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, info = rsa.radix_sort(b, aux, rsa.U32)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = min(best, dt)
        tot += dt
        how = info.hybrid
    print("n = %d: best %.1f us, mean %.1f us per sort (hybrid %d)" % (n, best * 1e6, tot / reps * 1e6, how))

if __name__ == "__main__":
    main()
