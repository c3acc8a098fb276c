"""PCIe-inclusive rate of the host-pointer entry point rsx_sort() (what the C++ template wrapper calls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, radix_sorting_amd as rsa
n = 1 << 28
a = ol.splitmix_fill(n, ol.U32, 1)
aux = np.zeros_like(a)
rsa.radix_sort_host(a[:1 << 20].copy(), aux[:1 << 20], rsa.U32)   # warm up: context, workspace
for rep in range(3):
    src = a.copy()
    t0 = time.perf_counter()
    res, info = rsa.radix_sort_host(src, aux, rsa.U32)
    dt = time.perf_counter() - t0
    print("rsx_sort host pointers, 2^28 u32 (pageable memory): %.1f ms -> %.2f Gkeys/s (result_in_aux %d)" % (dt * 1e3, n / dt / 1e9, info.result_in_aux))
assert np.all(res[:-1] <= res[1:])
