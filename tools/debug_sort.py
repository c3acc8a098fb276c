import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib as ol, radix_sorting_amd as rsa
for n in [8192, 8193, 16384, 32768, 32769, 40000, 65535, 65536, 100000, 300001, 1<<20]:
    a = ol.splitmix_fill(n, ol.U32, 1)
    src = torch.from_numpy(a.view(np.int32).copy()).cuda(); aux = torch.zeros_like(src)
    res, info = rsa.radix_sort(src, aux, dtype=rsa.U32); torch.cuda.synchronize()
    got = res.cpu().numpy().view(np.uint32); want = np.sort(a, kind="stable")
    bad = np.nonzero(got != want)[0]
    print(n, "cols", info.kept_columns(), "in_aux", info.result_in_aux, "mismatches", bad.size, bad[:8], flush=True)
    if bad.size:
        # which pass? check the multiset
        print("   multiset equal:", np.array_equal(np.sort(got), want))
