#!/usr/bin/env python3
"""Device-pointer radix_sort time against n (u32 keys, fresh unsorted input per call): where the paths hand over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa

def main():
    rsa.require_gpu()
    dev = torch.device("cuda:0")
    print(f"{'n':>11} {'us per sort':>12} {'Mkeys/s':>10} {'inplace_async':>14}")
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 31   # log2 of the largest size
    for lg in range(8, top + 1):
        for n in ((1 << lg), (1 << lg) + (1 << lg) // 2):
            if n > (1 << top):
                continue
            reps = max(3, min(200, (1 << 26) // n))
            bufs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
            aux = torch.empty(n, dtype=torch.int32, device=dev)
            best = 1e9
            for r in range(reps):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 1000 + r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rsa.radix_sort(b, aux, rsa.U32)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            best2 = 1e9
            for r in range(reps):
                b = bufs[r & 1]
                rsa.fill_splitmix(b, 1000 + r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rsa.radix_sort_inplace_async(b, aux, rsa.U32)
                torch.cuda.synchronize()
                best2 = min(best2, time.perf_counter() - t0)
            print(f"{n:11d} {best * 1e6:12.1f} {n / best / 1e6:10.1f} {best2 * 1e6:14.1f}")

if __name__ == "__main__":
    main()
