#!/usr/bin/env python3
"""Why does the histogram kernel take 0.236 ms inside bench.py's loop and 0.19-0.21 ms alone?  Three loops of
2^28-key sorts, the histogram kernel's time from rsx_profile (HIP events on the launch stream):
  rotate   every step sorts a DIFFERENT 1 GiB batch (bench.py: 23 batches, each touched once)
  same     every step sorts the SAME buffer, refilled in place before the step (same pages every time)
  hist     the histogram alone (rsx_histogram_device) over rotating batches / over one buffer"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa

def main():
    rsa.require_gpu()
    n = 1 << 28
    dev = torch.device("cuda:0")
    aux = torch.empty(n, dtype=torch.int32, device=dev)
    batches = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(12)]
    for i, b in enumerate(batches):
        rsa.fill_splitmix(b, seed=100 + i)
    torch.cuda.synchronize()
    def loop(name, pick, refill):
        for i in range(2):
            if refill: rsa.fill_splitmix(pick(i), seed=500 + i)
            rsa.radix_sort(pick(i), aux, rsa.U32)
        torch.cuda.synchronize()
        K = 10
        if refill:
            ms = 0.0
            for i in range(K):
                rsa.fill_splitmix(pick(i), seed=700 + i)
                torch.cuda.synchronize()
                rsa.profile_begin()
                rsa.radix_sort(pick(i), aux, rsa.U32)
                torch.cuda.synchronize()
                p = rsa.profile_end()
                ms += p.hist_ms / max(p.hist_launches, 1)
            print("%-46s histogram kernel %.4f ms per launch" % (name, ms / K))
        else:
            rsa.profile_begin()
            for i in range(K):
                rsa.radix_sort(pick(2 + i), aux, rsa.U32)
            torch.cuda.synchronize()
            p = rsa.profile_end()
            print("%-46s histogram kernel %.4f ms per launch" % (name, p.hist_ms / max(p.hist_launches, 1)))
    loop("sorts of different batches (bench.py's loop)", lambda i: batches[i % len(batches)], False)
    loop("sorts of one buffer, refilled in place", lambda i: batches[0], True)
    # the histogram alone
    import ctypes as C
    h = torch.empty(4 * 256 + 8, dtype=torch.int64, device=dev)
    def hist(b):
        rsa.check(rsa.lib().rsx_histogram_device(b.data_ptr(), n, rsa.U32, 0, h.data_ptr(), h[1024:].data_ptr(), None))
    for name, pick in (("histogram alone, rotating batches", lambda i: batches[i % len(batches)]), ("histogram alone, one buffer", lambda i: batches[0])):
        for i in range(3):
            hist(pick(i))
        torch.cuda.synchronize()
        rsa.profile_begin()
        for i in range(10):
            hist(pick(3 + i))
        torch.cuda.synchronize()
        p = rsa.profile_end()
        print("%-46s histogram kernel %.4f ms per launch" % (name, p.hist_ms / max(p.hist_launches, 1)))

if __name__ == "__main__":
    main()
