"""A/B of the level-2 leaf launches' grid (RSX_LEAF_GRID) inside one process: keys-only sorts of 2^28 and 2^29 u32 keys, a rank
sort and a key + payload sort of 2^28 f32 keys.  Median of six fresh sorts each, three rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import radix_sorting_amd as rsa
rsa.require_gpu()
n = 1 << 29
src = torch.empty(n, dtype=torch.int32, device='cuda')
aux = torch.empty_like(src)
for rnd in range(3):
    for grid in ('4096', '8192', '65536'):
        os.environ['RSX_LEAF_GRID'] = grid
        rsa.reload_env()
        out = []
        for m in (n, n // 2):
            ts = []
            for i in range(6):
                rsa.fill_splitmix(src[:m], seed=10 + i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r, info = rsa.radix_sort(src[:m], aux[:m], dtype=rsa.U32)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out.append('2^%d keys %.3f ms (route %d)' % (m.bit_length() - 1, sorted(ts)[2] * 1e3, info.hybrid))
        print('grid', grid, '; '.join(out), flush=True)
