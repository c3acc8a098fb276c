import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import radix_sorting_amd as rsa
rsa.require_gpu()
n = 1 << 28
src = torch.empty(n, dtype=torch.int32, device='cuda')
ib = torch.empty(2 * n, dtype=torch.int32, device='cuda')
k1 = torch.empty_like(src); v0 = torch.empty_like(src); v1 = torch.empty_like(src)
for rnd in range(3):
    for grid in ('8192', '65536', '16384'):
        os.environ['RSX_LEAF_GRID'] = grid
        rsa.reload_env()
        ts = []
        for i in range(6):
            rsa.fill_splitmix(src, seed=10 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        tp = []
        for i in range(6):
            rsa.fill_splitmix(src, seed=20 + i)
            v0.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rsa.radix_sort_pairs(src, k1, v0, v1, dtype=rsa.F32)
            torch.cuda.synchronize()
            tp.append(time.perf_counter() - t0)
        print('grid', grid, 'rank %.3f ms (route %d)  pairs %.3f ms' % (sorted(ts)[2] * 1e3, info.hybrid, sorted(tp)[2] * 1e3), flush=True)
