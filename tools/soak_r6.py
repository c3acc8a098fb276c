"""Soak of route 6 (rsx_logroute.hpp): random sizes, bit lengths, constant top bits, signed / complemented keys, clustered mantissas --
every result compared with torch.sort element for element (and, RSX_VERIFY=2, checked by the library's own checksum kernels).
python tools/soak_r6.py [seconds = 120] [seed = 1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSX_LOG_MIN_LOG2"] = "20"
os.environ["RSX_VERIFY"] = "2"
import radix_sorting_amd as rsa  # noqa: E402

SIGN = -(1 << 63)


def main():
    rsa.require_gpu()
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    g = torch.Generator(device="cpu")
    g.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rnd = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())  # noqa: E731
    t_end = time.time() + seconds
    sorts, keys, routes, kinds = 0, 0, {}, {}
    while time.time() < t_end:
        log2n = rnd(20, 26)
        n = (1 << log2n) + rnd(-4097, 4097) * rnd(0, 1)
        bmax = rnd(20, 46)
        r = torch.empty(n, dtype=torch.int64, device="cuda")
        rsa.fill_splitmix(r, seed=rnd(1, 1 << 30))
        b = 1 + (((r >> 58) & 63) % bmax)
        one = torch.ones_like(r)
        z = (one << (b - 1)) + (r & ((one << (b - 1)) - 1))
        kind = rnd(0, 5)
        dt, order = rsa.U64, rsa.ASCENDING
        if kind == 1:                                   # constant bits above the varying ones
            z = z | (rnd(1, 255) << 56)
        elif kind == 2:                                 # signed, positive
            dt = rsa.I64
        elif kind == 3:                                 # complemented keys, descending
            z, order = ~z, rsa.DESCENDING
        elif kind == 4:                                 # mantissas clustered in some buckets: a level-2 slot overflows
            z = torch.where((b > 14) & ((b & 3) == 0), z & ~(torch.full_like(z, 0xFF) << (b - 12).clamp(min=0)), z)
        elif kind == 5:                                 # sorted or nearly sorted input
            z = torch.sort(z).values
            if rnd(0, 1):
                z[rnd(0, n - 1)] = 0
        del r, b, one
        if dt == rsa.I64:
            want = torch.sort(z).values
        else:
            want = torch.sort(z ^ SIGN).values ^ SIGN
        if order == rsa.DESCENDING:
            want = want.flip(0)
        src, aux = z.clone(), torch.empty_like(z)
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
        torch.cuda.synchronize()
        if not bool((res == want).all().item()):
            print("MISMATCH: n %d bmax %d kind %d route %d" % (n, bmax, kind, info.hybrid), flush=True)
            sys.exit(1)
        sorts += 1
        keys += n
        routes[info.hybrid] = routes.get(info.hybrid, 0) + 1
        kinds[(kind, info.hybrid)] = kinds.get((kind, info.hybrid), 0) + 1
        del z, want, src, aux, res
    print("soak_r6: %d sorts, %.2f G keys, all equal to torch.sort; routes %s" % (sorts, keys / 1e9, dict(sorted(routes.items()))))
    print("  (input kind, route) -> sorts: %s" % dict(sorted(kinds.items())))
    print("  kinds: 0 plain Zipf-like, 1 constant top byte, 2 signed, 3 complemented + descending, 4 clustered mantissas, 5 (nearly) sorted; "
          "early exits report route 0")


if __name__ == "__main__":
    main()
