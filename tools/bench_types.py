import sys, time
sys.path.insert(0, "/root/repo")
import torch, radix_sorting_amd as rsa
rsa.require_gpu()
n = 1 << 28
for name, tdt, dt in (("u8", torch.uint8, rsa.U8), ("i16", torch.int16, rsa.I16), ("f64", torch.int64, rsa.F64), ("i32 desc", torch.int32, rsa.I32)):
    src = torch.empty(n, dtype=tdt, device="cuda"); aux = torch.empty_like(src); keep = torch.empty_like(src)
    rsa.fill_splitmix(keep, seed=5)
    ts = []
    for rep in range(4):
        src.copy_(keep); torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=1 if "desc" in name else 0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print("%-9s 2^28 keys: %.2f ms  %.1f Gkeys/s  (%d columns)" % (name, t * 1e3, n / t / 1e9, info.ncols))
