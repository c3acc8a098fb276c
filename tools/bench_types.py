#!/usr/bin/env python3
"""2^28 keys of other key types than the headline's (keys only, in HBM, fresh input per call): ms per sort."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa

CASES = [("u32 1 byte", rsa.U32, torch.int32, rsa.ASCENDING), ("u64 1 byte", rsa.U64, torch.int64, rsa.ASCENDING),
         ("u8", rsa.U8, torch.uint8, rsa.ASCENDING), ("i8 desc", rsa.I8, torch.int8, rsa.DESCENDING),
         ("i16", rsa.I16, torch.int16, rsa.ASCENDING), ("u16 desc", rsa.U16, torch.int16, rsa.DESCENDING), ("f64", rsa.F64, torch.float64, rsa.ASCENDING),
         ("i32 desc", rsa.I32, torch.int32, rsa.DESCENDING)]

def main():
    rsa.require_gpu()
    n = 1 << 28
    only = os.environ.get("RSX_BENCH_TYPES")   # e.g. "i16,u8": a subset
    for name, code, tdt, order in CASES:
        if only and name not in only.split(","):
            continue
        bufs = [torch.empty(n, dtype=tdt, device="cuda") for _ in range(2)]
        aux = torch.empty(n, dtype=tdt, device="cuda")
        best, cols = 1e9, 0
        for r in range(6):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 50 + r, mask=0x00FF0000 if "1 byte" in name else 0xFFFFFFFFFFFFFFFF)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res, info = rsa.radix_sort(b, aux, code, order)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            cols = info.ncols
        print(f"{name:9s} 2^28 keys: {best * 1e3:.2f} ms  {n / best / 1e9:.1f} Gkeys/s  ({cols} columns)")
        del bufs, aux
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
