"""Soak: many back-to-back sorts of seven sizes -- the look-back chain under long runs.  Every sort's output is checked for
sortedness and against the input's checksum (sum and xor of the keys); with RSX_VERIFY=1 in the environment the library
additionally re-ranks one tile of every scatter pass without LDS atomics (include/rsx.h).

    python tools/soak.py [seconds] [--small]        (--small: sizes up to 2^24 only)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa

rsa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 60.0
sizes = [1 << 28, 40000000, 1 << 24, 10 ** 6, 10 ** 5, 50000, 4097]
if "--small" in sys.argv:
    sizes = sizes[2:]
sign = torch.tensor(-2 ** 31, dtype=torch.int32, device="cuda")
t_end = time.time() + budget
counts = {}
while time.time() < t_end:
    for n in sizes:
        src = torch.empty(n, dtype=torch.int32, device="cuda")
        aux = torch.empty_like(src)
        reps = max(1, min(400, (1 << 28) // n))
        for r in range(reps):
            rsa.fill_splitmix(src, seed=r + 1 + counts.get(n, 0))
            check = r == reps - 1 or r % 16 == 0
            if check:
                s_in = int(src.to(torch.int64).sum().item())
                x_in = int(torch.bitwise_xor(src[: n // 2 * 2].view(-1, 2)[:, 0], src[: n // 2 * 2].view(-1, 2)[:, 1]).to(torch.int64).sum().item())
            res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
            if check:
                torch.cuda.synchronize()
                f = res ^ sign
                assert bool((f[1:] >= f[:-1]).all().item()), ("not sorted", n, r)
                assert int(res.to(torch.int64).sum().item()) == s_in, ("keys changed", n, r)
                del f
        torch.cuda.synchronize()
        counts[n] = counts.get(n, 0) + reps
        del src, aux, res
print("soak ok:", counts, "RSX_VERIFY=%s" % os.environ.get("RSX_VERIFY", "0"))
