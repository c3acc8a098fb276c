"""rsx_sort_inplace_async (what the distributed sort's sub-range sorts call) against the blocking sort at the same sizes, u32 keys:
ms per sort (HIP events, median of 5) and the route each took.  python tools/async_sizes.py [sizes in Mi ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def med(f, reps=7):
    ts = []
    for i in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = f()
        b.record()
        torch.cuda.synchronize()
        if i >= 2:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], r


def main():
    rsa.require_gpu()
    sizes = [int(x) for x in sys.argv[1:]] or [64, 128, 256, 512]
    for mi in sizes:
        n = mi << 20
        src0 = torch.empty(n, dtype=torch.int32, device="cuda")
        rsa.fill_splitmix(src0, seed=5)
        src, aux = torch.empty_like(src0), torch.empty_like(src0)

        def blocking():
            src.copy_(src0)
            return rsa.radix_sort(src, aux, dtype=rsa.U32)[1].hybrid

        def asyn():
            src.copy_(src0)
            rsa.radix_sort_inplace_async(src, aux, dtype=rsa.U32)
            return None
        rsa.reload_env()
        tb, rb = med(blocking)
        ta, _ = med(asyn)
        tc, _ = med(lambda: src.copy_(src0))
        route = rsa.async_route() if hasattr(rsa, "async_route") else -1
        ok = bool((src[1:].view(torch.int32) ^ -2**31 >= src[:-1] ^ -2**31).all().item())
        print("%4d Mi u32 keys: blocking %.3f ms (route %d) | inplace_async %.3f ms (route %s, sorted %s) | (both include a copy of %.3f ms)" %
              (mi, tb, rb, ta, route, ok, tc), flush=True)
        del src0, src, aux
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
