"""Round-5 soak (tools/soak_r4.py with the sizes and inputs round 5 added: 4-byte keys up to 2^31 -- the larger leaf shapes and the
counting leaves --, 8-byte keys from 48 Mi -- the level-1 pass in atoms --, rank sorts of keys whose varying bits are packed and of
floats on a grid).  Round-4 soak: keys-only sorts (and, for 4-byte keys, rank and key + payload sorts: shapes 1x / 2x in the report) of random sizes (4.5 Mi .. 300 Mi keys, the range the sorts without a histogram cover with a wave /
a workgroup per leaf) and random shapes of input -- uniform, constant top bits (digits below them), low bits clustered
everywhere or in some buckets only, constant columns, a few strays -- under RSX_VERIFY=2: the library itself checks every
result on the device (sorted, the input's key sum and key mix) whatever route the sort took.  Prints sorts per route.

    RSX_VERIFY=2 python tools/soak_r5.py [seconds]
"""
import os
import sys
import time

os.environ.setdefault("RSX_VERIFY", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import radix_sorting_amd as rsa

rsa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261003)
t_end = time.time() + budget
routes, shapes = {}, {}
nsorts = 0
while time.time() < t_end:
    n = int(rng.choice([rng.integers(9 << 20, 60 << 20), rng.integers(60 << 20, 160 << 20), rng.integers(160 << 20, 300 << 20),
                        rng.integers(260 << 20, 560 << 20), rng.integers(500 << 20, 1100 << 20), rng.integers(1000 << 20, (2048 << 20) + 60000)]))
    dt, tdt = [(rsa.U32, torch.int32), (rsa.I32, torch.int32), (rsa.F32, torch.int32), (rsa.U64, torch.int64)][int(rng.integers(0, 4))]
    if dt == rsa.U64 and n > (400 << 20):
        n //= 4
    order = int(rng.integers(0, 2))
    src = torch.empty(n, dtype=tdt, device="cuda")
    aux = torch.empty_like(src)
    shape = int(rng.integers(0, 8))
    bits = 8 * src.element_size()
    full = (1 << bits) - 1
    mask = full
    if shape == 1:                       # constant top bits
        mask = full >> int(rng.integers(1, 8))
    elif shape == 2:                     # low bits from few values, everywhere
        mask = full & ~int(rng.choice([0x0FF0, 0x03F0, 0xF0F0, 0x00FF]))
    elif shape == 3:                     # a constant column somewhere
        mask = full & ~(0xFF << (8 * int(rng.integers(0, bits // 8))))
    elif shape == 7 and dt == rsa.U64:   # 8-byte keys below 2^40 / 2^32 / 2^44: four-byte level-2 slots where the leaves fit the low word
        mask = int(rng.choice([0xFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFF]))
    if shape == 6 and dt == rsa.F32 and n <= (256 << 20):   # keys no byte scheme spreads (rank sorts pack them / take them for fixed-point numbers)
        mask = 0xFFF000FF
    rsa.fill_splitmix(src, seed=int(rng.integers(1, 1 << 40)), mask=mask)
    if shape == 6 and dt == rsa.F32 and n <= (256 << 20) and rng.random() < 0.5:   # floats on a grid: (int24 - 2^23) * 2^-23
        src.copy_((((src.to(torch.int64) & 0xFFFFFF) - (1 << 23)).to(torch.float32) * (2.0 ** -23)).view(torch.int32))
    if shape == 1 and rng.random() < 0.5:
        src |= int(rng.integers(0, 1 << 7)) << (bits - 7) if dt != rsa.U64 else 0
    if shape == 4:                       # low bits clustered in some (digit, digit) buckets only
        top = (src >> (bits - 16)) & 0xFFFF
        sel = (top % 97) == 5
        src.copy_(torch.where(sel, src & ~0x0FF0, src))   # (masked assignment overflows torch's index arithmetic above 2^31 elements)
        del top, sel
    elif shape == 5:                     # a few strays above constant top bits
        src &= full >> 3 if dt != rsa.U64 else full
        idx = torch.from_numpy(rng.integers(0, n, size=3)).cuda()
        src[idx] = src[idx] | (1 << (bits - 2))
    rsa.reload_env()                      # (no back-off: every sort may try every route)
    kind = int(rng.integers(0, 4)) if (dt != rsa.U64 and n <= (256 << 20)) else 0
    if shape == 6 and dt == rsa.F32 and n <= (256 << 20):
        kind = 1
    if kind == 1:                        # stable ranks (RSX_VERIFY=2: a permutation through which the keys do not descend, ties in index order)
        ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
        res, info = rsa.radix_sort_rank(src, ib, dtype=dt, order=order)
        del ib
    elif kind == 2:                      # key + payload (no descent, the input's key sum and pair mix)
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        vaux = torch.empty_like(vals)
        res, _, info = rsa.radix_sort_pairs(src, aux, vals, vaux, dtype=dt, order=order)
        del vals, vaux
    else:
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)     # RSX_VERIFY=2 raises on a wrong result
    torch.cuda.synchronize()
    shape = shape + 10 * kind
    routes[int(info.hybrid)] = routes.get(int(info.hybrid), 0) + 1
    shapes[(shape, int(info.hybrid))] = shapes.get((shape, int(info.hybrid)), 0) + 1
    nsorts += 1
    del src, aux, res
print("soak ok: %d sorts under RSX_VERIFY=%s; by route %s; by (input shape, route) %s" % (
    nsorts, os.environ.get("RSX_VERIFY"), dict(sorted(routes.items())), dict(sorted(shapes.items()))))
