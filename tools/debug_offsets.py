import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib as ol, radix_sorting_amd as rsa
lib = rsa.lib()
for n in [40000, 65536, 100000, 1<<20]:
    a = ol.splitmix_fill(n, ol.U32, 1)
    src = torch.from_numpy(a.view(np.int32).copy()).cuda()
    nseg, tps, se = C.c_uint32(), C.c_uint32(), C.c_uint64()
    offs = np.zeros(128*4*256, dtype=np.uint64)
    rc = lib.rsx_debug_offsets(C.c_void_p(src.data_ptr()), C.c_size_t(n), 2, 0, C.byref(nseg), C.byref(tps), C.byref(se), C.c_void_p(offs.ctypes.data), C.c_size_t(offs.size), None)
    print(n, "rc", rc, "nseg", nseg.value, "tps", tps.value, "seg_elems", se.value)
    S = nseg.value
    got = offs[:S*4*256].reshape(S, 4, 256)
    want = np.zeros_like(got)
    cnt = np.zeros((S, 4, 256), dtype=np.uint64)
    for s in range(S):
        seg = a[s*se.value:(s+1)*se.value]
        for c in range(4):
            cnt[s, c] = np.bincount((seg >> (8*c)) & 0xFF, minlength=256)
    for c in range(4):
        tot = cnt[:, c].sum(axis=0)
        dbase = np.concatenate([[0], np.cumsum(tot)[:-1]]).astype(np.uint64)
        run = dbase.copy()
        for s in range(S):
            want[s, c] = run
            run = run + cnt[s, c]
    bad = np.argwhere(got != want)
    print("   mismatching offsets:", len(bad), bad[:5].tolist())
    if len(bad):
        s_, c_, d_ = bad[0]
        print("   got", got[s_, c_, d_:d_+4], "want", want[s_, c_, d_:d_+4], "cnt", cnt[s_, c_, d_:d_+4])
