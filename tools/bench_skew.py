"""Whole-sort time of 2^28 u32 keys for input orders / distributions that stress the per-(wave, digit) LDS atomics."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radix_sorting_amd as rsa

rsa.require_gpu()
n = 1 << 28
dev = torch.device("cuda", 0)


def run(name, make):
    src = make()
    aux = torch.empty_like(src)
    keep = src.clone()
    times = []
    for rep in range(4):
        src.copy_(keep)
        torch.cuda.synchronize()
        rsa.profile_begin()
        t0 = time.perf_counter()
        res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        prof = rsa.profile_end()
    t = sorted(times)[1]
    print(json.dumps({"input": name, "ms_per_sort": t * 1e3, "Gkeys_per_s": n / t / 1e9, "kept_columns": info.ncols,
                      "scatter_ms_per_launch": prof.scatter_ms / max(prof.scatter_launches, 1), "hist_ms": prof.hist_ms}))


def uniform():
    t = torch.empty(n, dtype=torch.int32, device=dev)
    rsa.fill_splitmix(t, seed=1)
    return t


def nearly_sorted():
    t = uniform()
    u = (t.to(torch.int64) & 0xFFFFFFFF).sort().values
    u = torch.where(u >= (1 << 31), u - (1 << 32), u).to(torch.int32)
    u[0], u[1] = u[1].clone(), u[0].clone()
    u[n // 2], u[n // 2 + 1] = u[n // 2 + 1].clone(), u[n // 2].clone()
    return u


def reverse_sorted():
    return nearly_sorted().flip(0).contiguous()


def few_values():
    t = uniform()
    return (t & 0x03030303).contiguous()      # 4 values per byte column: every (wave, digit) cell is hot


def sawtooth():
    return (torch.arange(n, dtype=torch.int64, device=dev) % 1000003 * 4099 % (1 << 31)).to(torch.int32)


for name, make in (("uniform", uniform), ("sorted but for two swaps", nearly_sorted), ("reverse sorted", reverse_sorted),
                   ("4 values per byte", few_values), ("sawtooth", sawtooth)):
    run(name, make)
