"""Device memory the library holds after one sort (hipMemGetInfo before the first call and after it), and the sort's time with
the level-1 slots in the caller's second buffer (default) and all in scratch memory (RSX_NO_AUX_SLOTS=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for dt, tdt, n in ((rsa.U32, torch.int32, 1 << 28), (rsa.U32, torch.int32, 40000000), (rsa.U64, torch.int64, 1 << 27), (rsa.U64, torch.int64, 1 << 28)):
    bufs = [torch.empty(n, dtype=tdt, device="cuda") for _ in range(2)]
    aux = torch.empty(n, dtype=tdt, device="cuda")
    for name, envs in (("slots in aux", {}), ("RSX_NO_AUX_SLOTS=1", {"RSX_NO_AUX_SLOTS": "1"})):
        os.environ.pop("RSX_NO_AUX_SLOTS", None)
        os.environ.update(envs)
        rsa.release_stream()
        rsa.reload_env()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        best = 1e9
        for r in range(8):
            b = bufs[r & 1]
            rsa.fill_splitmix(b, 100 + r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, info = rsa.radix_sort(b, aux, dt)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        held = free0 - torch.cuda.mem_get_info()[0]
        print("%s n = %10d  %-20s %.3f ms (route %d)   library holds %.3f GiB = %.2f x the keys" % ("u32" if dt == rsa.U32 else "u64", n, name, best * 1e3, info.hybrid, held / 2**30, held / (n * (4 if dt == rsa.U32 else 8))), flush=True)
    del bufs, aux
# rank and key + payload sorts of f32 keys: the level-1 slots of keys and payloads in the caller's spare buffers (the second key /
# payload buffers; a rank sort's index buffer for the indices) or all four slot arrays in scratch memory (RSX_NO_AUX_SLOTS=1)
for n in (1 << 28, 1 << 26):
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    kin = torch.empty_like(keys)
    for what in ("f32 -> u32 ranks", "f32 + u32 payload"):
        for name, envs in (("slots in spare buffers", {}), ("RSX_NO_AUX_SLOTS=1", {"RSX_NO_AUX_SLOTS": "1"})):
            os.environ.pop("RSX_NO_AUX_SLOTS", None)
            os.environ.update(envs)
            rsa.release_stream()
            rsa.reload_env()
            rsa.fill_splitmix(keys, 6)
            if what.endswith("ranks"):
                ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
            else:
                vals = torch.arange(n, dtype=torch.int32, device="cuda")
                vin, ka, va = torch.empty_like(vals), torch.empty_like(keys), torch.empty_like(vals)
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            best = 1e9
            for r in range(6):
                kin.copy_(keys)
                if not what.endswith("ranks"):
                    vin.copy_(vals)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if what.endswith("ranks"):
                    _, info = rsa.radix_sort_rank(kin, ib, dtype=rsa.F32)
                else:
                    _, _, info = rsa.radix_sort_pairs(kin, ka, vin, va, dtype=rsa.F32)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            held = free0 - torch.cuda.mem_get_info()[0]
            print("%-18s n = %10d  %-24s %.3f ms (route %d)   library holds %.3f GiB = %.2f x the keys" % (what, n, name, best * 1e3, info.hybrid, held / 2**30, held / (4.0 * n)), flush=True)
            if what.endswith("ranks"):
                del ib
            else:
                del vals, vin, ka, va
    del keys, kin
os.environ.pop("RSX_NO_AUX_SLOTS", None)
