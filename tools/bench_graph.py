"""rsx_sort_inplace_async captured into a HIP graph and replayed, against the synchronous rsx_sort_device, u32 keys in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import radix_sorting_amd as rsa

rsa.require_gpu()
s = torch.cuda.Stream()
for n in (1000, 10000, 100000, 1000000, 10000000, 40000000, 1 << 28):
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    keep = torch.empty_like(src)
    aux = torch.empty_like(src)
    rsa.fill_splitmix(keep, seed=1)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        src.copy_(keep)
        rsa.radix_sort_inplace_async(src, aux, dtype=rsa.U32, stream=s)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_inplace_async(src, aux, dtype=rsa.U32, stream=torch.cuda.current_stream())
    reps = 200 if n <= 10000000 else 20

    def timed(fn):
        ts = []
        for _ in range(reps):
            src.copy_(keep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2]

    t_graph = timed(lambda: g.replay())
    t_async = timed(lambda: rsa.radix_sort_inplace_async(src, aux, dtype=rsa.U32))
    t_sync = timed(lambda: rsa.radix_sort(src, aux, dtype=rsa.U32))
    print("n = %9d: graph replay %8.1f us   async enqueue %8.1f us   synchronous rsx_sort_device %8.1f us" %
          (n, t_graph * 1e6, t_async * 1e6, t_sync * 1e6))
