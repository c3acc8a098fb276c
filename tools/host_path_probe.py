import os, sys, time
"""PCIe-inclusive time of the host-pointer entry point rsx_sort() (what the C++ template wrapper calls): reused buffers,
freshly allocated buffers, and already sorted input (transfer in + histogram only)."""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import radix_sorting_amd as rsa
n = 1 << 28
a = np.random.default_rng(1).integers(0, 1 << 32, size=n, dtype=np.uint32)
aux = np.zeros_like(a)
rsa.radix_sort_host(a[:1 << 20].copy(), aux[:1 << 20], rsa.U32)
src = a.copy()
for rep in range(4):
    np.copyto(src, a)
    t0 = time.perf_counter()
    res, info = rsa.radix_sort_host(src, aux, rsa.U32)
    dt = time.perf_counter() - t0
    print("reused buffers: %.1f ms" % (dt * 1e3))
for rep in range(3):
    src2 = a.copy()
    t0 = time.perf_counter()
    res, info = rsa.radix_sort_host(src2, aux, rsa.U32)
    dt = time.perf_counter() - t0
    print("fresh src: %.1f ms" % (dt * 1e3))
    del src2
# already-sorted input: early exit, only H2D
t0 = time.perf_counter(); res, info = rsa.radix_sort_host(res, aux, rsa.U32); print("sorted input (H2D + hist only): %.1f ms, exit %d" % ((time.perf_counter() - t0) * 1e3, info.early_exit))
# the same sort through rsx_sort_multi with 1, 2, 4 ranks on device 0 (one host thread per rank while a phase runs)
for ranks in (1, 2, 4, 8):
    for rep in range(3):
        np.copyto(src, a)
        t0 = time.perf_counter()
        res, info = rsa.radix_sort_multi_host(src, aux, rsa.U32, devices=[0] * ranks)
        dt = time.perf_counter() - t0
    print("rsx_sort_multi, %d rank(s) on device 0, reused buffers: %.1f ms (sorted: %s)" % (ranks, dt * 1e3, bool(np.all(res[1:] >= res[:-1]))))
