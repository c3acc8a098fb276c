"""One-off check of array sizes beyond every BASELINE.json configuration: sorts n u32 keys (default 2^32 + 4097, 16 GiB per
buffer) on one GPU and verifies sortedness, the key sum and the key xor in chunks.  Not part of the test suite (40 GiB of HBM
and half a minute); run as  python tools/big_sort_check.py [log2n] [extra] [reps: timed repetitions afterwards]."""
import sys
import time

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402

CHUNK = 1 << 28
SIGN = -(1 << 31)


def checksums(t):
    s, x = 0, 0
    for o in range(0, t.numel(), CHUNK):
        c = t[o:o + CHUNK].to(torch.int64) & 0xFFFFFFFF
        s += int(c.sum().item())
        r = c
        while r.numel() > 1:   # xor-fold
            h = r.numel() // 2
            odd = r[2 * h:]
            r = r[:h] ^ r[h:2 * h]
            if odd.numel():
                r[0] ^= odd[0]
        x ^= int(r[0].item())
    return s, x


def main():
    rsa.require_gpu()
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = (1 << log2n) + (int(sys.argv[2]) if len(sys.argv) > 2 else 4097)
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    aux = torch.empty_like(src)
    rsa.fill_splitmix(src, seed=5)
    before = checksums(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = True
    for o in range(0, n - 1, CHUNK):
        c = res[o:o + CHUNK + 1] ^ SIGN   # unsigned order as signed
        ok &= bool((c[1:] >= c[:-1]).all().item())
    after = checksums(res)
    print("n = %d: %.2f ms, %.1f Gkeys/s, ncols %d, route %d, sorted %s, sum/xor preserved %s" %
          (n, dt * 1e3, n / dt / 1e9, info.ncols, info.hybrid, ok, before == after))
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if reps:
        # the same sort again (scratch memory allocated, clocks up): the best of `reps`, by device events around the call
        best = None
        for _ in range(reps):
            rsa.fill_splitmix(src, seed=5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            rsa.radix_sort(src, aux, dtype=rsa.U32)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        print("steady: %.3f ms, %.1f Gkeys/s" % (best, n / best / 1e6))
    sys.exit(0 if ok and before == after else 1)


if __name__ == "__main__":
    main()
