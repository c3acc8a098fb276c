"""How often does an evenly spread array lose its attempt without a histogram to a slot overflow?  Fresh uniform u32 keys (and f32
keys -> ranks), sizes whose (digit, digit) buckets have little room in their slots; rsx_reload_env() before every sort (no back-off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import radix_sorting_amd as rsa
rsa.require_gpu()
for n in (11800000, 12582912, 13107200, 13369344, 25600000, 26738688, 40000000):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    aux = torch.empty_like(src)
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    routes, rroutes = {}, {}
    for r in range(150):
        rsa.fill_splitmix(src, 5000 + r)
        rsa.reload_env()
        _, info = rsa.radix_sort(src, aux, rsa.U32)
        routes[int(info.hybrid)] = routes.get(int(info.hybrid), 0) + 1
        rsa.fill_splitmix(src, 9000 + r)
        rsa.reload_env()
        _, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
        rroutes[int(info.hybrid)] = rroutes.get(int(info.hybrid), 0) + 1
    print("n = %9d (mean bucket %.1f): keys by route %s   ranks by route %s" % (n, n / 65536.0, routes, rroutes), flush=True)
    del src, aux, ib
