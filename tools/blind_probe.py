"""The histogram-less two-level route (rsx_hybrid.hpp, rsx_blind_*) against the histogram-first one: correctness against
torch.sort and time per sort, per key type and input.  Run on the GPU box:  python tools/blind_probe.py [log2n]."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa  # noqa: E402


def timed(src0, aux, dtype, reps=6):
    ts, route = [], None
    for _ in range(reps):
        src = src0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, info = rsa.radix_sort(src, aux, dtype=dtype)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        route = info.hybrid
    return res, route, sorted(ts)[len(ts) // 2] * 1e3


def check_u32(res, src0):
    want = torch.sort((src0.to(torch.int64) & 0xFFFFFFFF))[0]
    return bool(((res.to(torch.int64) & 0xFFFFFFFF) == want).all().item())


def main():
    rsa.require_gpu()
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    n = (1 << log2n) + 12345
    g = torch.Generator(device="cuda").manual_seed(7)
    cases = {}
    u = torch.randint(-2 ** 31, 2 ** 31, (n,), dtype=torch.int32, device="cuda", generator=g)
    cases["uniform"] = u
    cases["sorted"] = torch.sort(u)[0] ^ -(1 << 31)   # sorted as unsigned
    cases["low byte constant"] = u & ~0xFF
    cases["top byte skewed (half the keys in one digit)"] = torch.where(u & 1 == 1, u & 0x00FFFFFF, u)
    cases["(top, next) clustered"] = u & ~0x00F00000
    cases["low byte hot (7/8 zero)"] = torch.where(u & 0x700 != 0, u & ~0xFF, u)
    aux = torch.empty_like(u)
    for name, src0 in cases.items():
        for blind in (1, 0):
            os.environ["RSX_NO_BLIND"] = "0" if blind else "1"
            rsa.reload_env()
            res, route, ms = timed(src0, aux, rsa.U32)
            ok = check_u32(res, src0)
            print("%-50s blind %d: route %d, %.3f ms, %s" % (name, blind, route, ms, "ok" if ok else "WRONG"), flush=True)
    del cases, u, aux
    # u64 uniform
    n64 = n // 2
    hi = torch.randint(-2 ** 31, 2 ** 31, (n64,), dtype=torch.int64, device="cuda", generator=g)
    lo = torch.randint(0, 2 ** 32, (n64,), dtype=torch.int64, device="cuda", generator=g)
    k = (hi << 32) | lo
    aux = torch.empty_like(k)
    want = None
    for blind in (1, 0):
        os.environ["RSX_NO_BLIND"] = "0" if blind else "1"
        rsa.reload_env()
        res, route, ms = timed(k, aux, rsa.I64)
        if want is None:
            want = torch.sort(k)[0]
        ok = bool((res == want).all().item())
        print("%-50s blind %d: route %d, %.3f ms, %s" % ("i64 uniform, n/2 keys", blind, route, ms, "ok" if ok else "WRONG"), flush=True)


if __name__ == "__main__":
    main()
