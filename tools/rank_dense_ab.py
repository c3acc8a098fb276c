"""A/B inside one process: rank sort and key + payload sort of 2^28 f32 keys with the level-2 pass writing 2-byte key slots
(default) or whole keys (RSX_NO_DENSE_SLOTS=1).  Median of six fresh sorts, three rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa
rsa.require_gpu()
n = 1 << 28
src = torch.empty(n, dtype=torch.int32, device='cuda')
ib = torch.empty(2 * n, dtype=torch.int32, device='cuda')
k1 = torch.empty_like(src); v0 = torch.empty_like(src); v1 = torch.empty_like(src)
for rnd in range(3):
    for nd in ('1', '0'):
        os.environ['RSX_NO_DENSE_SLOTS'] = nd
        rsa.reload_env()
        ts, tp = [], []
        for i in range(6):
            rsa.fill_splitmix(src, seed=10 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r, info = rsa.radix_sort_rank(src, ib, dtype=rsa.F32)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        for i in range(6):
            rsa.fill_splitmix(src, seed=20 + i)
            rsa.fill_splitmix(v0, seed=30 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, pinfo = rsa.radix_sort_pairs(src, k1, v0, v1, dtype=rsa.F32)
            torch.cuda.synchronize()
            tp.append(time.perf_counter() - t0)
        print('RSX_NO_DENSE_SLOTS=%s: rank %.3f ms (route %d)  pairs %.3f ms (route %d)' %
              (nd, sorted(ts)[2] * 1e3, info.hybrid, sorted(tp)[2] * 1e3, pinfo.hybrid), flush=True)
