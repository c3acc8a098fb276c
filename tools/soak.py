"""Soak: many back-to-back sorts (no result check beyond sortedness of the last one of each size): the look-back chain under
long runs.  python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sizes = [1 << 28, 40000000, 1 << 24, 10 ** 6, 10 ** 5, 50000, 4097]
t_end = time.time() + budget
counts = {}
while time.time() < t_end:
    for n in sizes:
        src = torch.empty(n, dtype=torch.int32, device="cuda")
        aux = torch.empty_like(src)
        reps = max(1, min(400, (1 << 28) // n))
        for r in range(reps):
            rsa.fill_splitmix(src, seed=r + 1 + counts.get(n, 0))
            res, info = rsa.radix_sort(src, aux, dtype=rsa.U32)
        torch.cuda.synchronize()
        f = res ^ torch.tensor(-2 ** 31, dtype=torch.int32, device="cuda")
        assert bool((f[1:] >= f[:-1]).all().item()), n
        counts[n] = counts.get(n, 0) + reps
        del src, aux, res, f
print("soak ok:", counts)
