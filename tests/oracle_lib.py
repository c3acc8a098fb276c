"""ctypes doorway to the CPU checker (oracle/liboracle.so, oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, tools/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
(radix_sorting_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# dtype codes (oracle/rs_oracle.h RSO_*, identical to include/rsx.h rsx_dtype)
U8, U16, U32, U64, I8, I16, I32, I64, F32, F64 = range(10)
DTYPE_NAMES = ["uint8_t", "uint16_t", "uint32_t", "uint64_t", "int8_t", "int16_t",
               "int32_t", "int64_t", "float", "double"]
NP_DTYPES = [np.uint8, np.uint16, np.uint32, np.uint64, np.int8, np.int16, np.int32,
             np.int64, np.float32, np.float64]
# unsigned view types (bit patterns; lets float NaN payloads compare exactly)
NP_BITS = [np.uint8, np.uint16, np.uint32, np.uint64, np.uint8, np.uint16, np.uint32,
           np.uint64, np.uint32, np.uint64]
DTYPE_SIZE = [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
ASC, DESC = 0, 1


class Info(C.Structure):
    _fields_ = [("n_unsorted", C.c_uint64), ("key_bytes", C.c_uint32), ("ncols", C.c_uint32),
                ("cols", C.c_uint32 * 8), ("early_exit", C.c_uint32), ("result_in_aux", C.c_uint32)]


def build_oracle():
    """(Re)build oracle/liboracle.so and, where /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        lib.rso_kdf.restype = C.c_uint64
        lib.rso_kdf.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.rso_histogram.restype = None
        lib.rso_histogram.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                      C.c_void_p, C.POINTER(C.c_uint64)]
        lib.rso_sort.restype = C.c_int
        lib.rso_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(Info)]
        lib.rso_sort_records.restype = C.c_int
        lib.rso_sort_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_int, C.c_int, C.POINTER(Info)]
        for f in (lib.rso_sort_rank, lib.rso_sort_rank_asheader):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_int,
                          C.c_size_t, C.POINTER(Info)]
        lib.rso_sort_main_hist.restype = C.c_int
        lib.rso_sort_main_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.POINTER(Info)]
        lib.rso_fnv1a64.restype = C.c_uint64
        lib.rso_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.rso_fill_splitmix.restype = None
        lib.rso_fill_splitmix.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64]
        _oracle = lib
    return _oracle


def ref():
    """The real reference compiled in place (oracle/_ref/libref.so) or None when absent."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref.so")
        if not os.path.exists(path):
            if os.path.exists("/root/reference/radix_sort.hpp"):
                build_oracle()
            if not os.path.exists(path):
                return None
        lib = C.CDLL(path)
        lib.ref_sort.restype = C.c_int
        lib.ref_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        lib.ref_sort_rank.restype = C.c_int
        lib.ref_sort_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        lib.ref_sort_kv.restype = C.c_int
        lib.ref_sort_kv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        lib.ref_sort_sortrec.restype = C.c_int
        lib.ref_sort_sortrec.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        lib.ref_rank_sortrec_u8idx.restype = C.c_int
        lib.ref_rank_sortrec_u8idx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.ref_sizeof_sortrec.restype = C.c_size_t
        lib.ref_sort_main_hist.restype = C.c_int
        lib.ref_sort_main_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        _ref = lib
    return _ref


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fnv1a64(a):
    a = np.ascontiguousarray(a)
    return int(oracle().rso_fnv1a64(ptr(a), a.nbytes))


def splitmix_fill(n, dtype, seed, mask=0xFFFFFFFFFFFFFFFF):
    """SURVEY.md section 4 generator: element i = low sizeof(T) bytes of splitmix64() & mask.
    Returned as the unsigned bit-pattern array of the element width."""
    a = np.empty(n, dtype=NP_BITS[dtype])
    oracle().rso_fill_splitmix(ptr(a), n, DTYPE_SIZE[dtype], seed, mask)
    return a


def oracle_sort(bits, dtype, order=ASC):
    """Run the C restatement on a copy.  Returns (result_bits, result_in_aux, info)."""
    src = np.array(bits, dtype=NP_BITS[dtype], copy=True)
    aux = np.full_like(src, 0xA5)
    info = Info()
    r = oracle().rso_sort(ptr(src), ptr(aux), src.size, dtype, order, C.byref(info))
    assert r in (0, 1)
    return (aux if r else src), r, info


def ref_sort(bits, dtype, order=ASC):
    """Run the real reference on a copy.  Returns (result_bits, result_in_aux)."""
    src = np.array(bits, dtype=NP_BITS[dtype], copy=True)
    aux = np.full_like(src, 0xA5)
    r = ref().ref_sort(ptr(src), ptr(aux), src.size, dtype, order)
    assert r in (0, 1)
    return (aux if r else src), r


def hvt_bytes_for(n):
    """Counter width radix_sort picks by n (radix_sort.hpp:102-114)."""
    return 1 if n < 256 else 2 if n < (1 << 16) else 4 if n < (1 << 32) else 8


def oracle_sort_main_hist(bits, dtype, hvt_bytes, order=ASC):
    """rs_sort_main with a caller's Hist through the C restatement: (result, in_aux, hist[256*kb] uint64)."""
    src = np.array(bits, dtype=NP_BITS[dtype], copy=True)
    aux = np.full_like(src, 0xA5)
    hist = np.zeros(256 * DTYPE_SIZE[dtype], dtype=np.uint64)
    info = Info()
    r = oracle().rso_sort_main_hist(ptr(src), ptr(aux), src.size, dtype, order, hvt_bytes, ptr(hist), C.byref(info))
    return (aux if r else src), r, hist


def ref_sort_main_hist(bits, dtype, hvt_bytes):
    """The same through the real reference's rs_sort_main with a zeroed std::vector<HVT>."""
    src = np.array(bits, dtype=NP_BITS[dtype], copy=True)
    aux = np.full_like(src, 0xA5)
    hist = np.zeros(256 * DTYPE_SIZE[dtype], dtype=np.uint64)
    r = ref().ref_sort_main_hist(ptr(src), ptr(aux), src.size, dtype, hvt_bytes, ptr(hist))
    assert r in (0, 1)
    return (aux if r else src), r, hist


def oracle_rank(bits, dtype, idx_bytes=4, order=ASC, asheader=False):
    src = np.ascontiguousarray(bits, dtype=NP_BITS[dtype])
    idt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[idx_bytes]
    ib = np.full(2 * src.size, 0xA5, dtype=idt) if src.size else np.zeros(0, dtype=idt)
    info = Info()
    f = oracle().rso_sort_rank_asheader if asheader else oracle().rso_sort_rank
    r = f(ptr(src), DTYPE_SIZE[dtype], 0, dtype, order, ptr(ib), idx_bytes, src.size, C.byref(info))
    n = src.size
    return (ib[n:] if r else ib[:n]), r, info, ib


def ref_rank(bits, dtype, idx_bytes=4, order=ASC):
    src = np.ascontiguousarray(bits, dtype=NP_BITS[dtype])
    idt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[idx_bytes]
    ib = np.full(2 * src.size, 0xA5, dtype=idt)
    r = ref().ref_sort_rank(ptr(src), ptr(ib), src.size, dtype, idx_bytes, order)
    n = src.size
    return (ib[n:] if r else ib[:n]), r, ib


def kdf_keys(bits, dtype, order=ASC):
    """numpy restatement of basic_kdfs::kdf (radix_sort_basic_kdf.hpp:19-46) on bit patterns."""
    u = np.asarray(bits, dtype=NP_BITS[dtype])
    nb = DTYPE_SIZE[dtype] * 8
    top = NP_BITS[dtype](1 << (nb - 1))
    if dtype in (I8, I16, I32, I64):
        k = u ^ top
    elif dtype in (F32, F64):
        neg = (u >> NP_BITS[dtype](nb - 1)).astype(bool)
        k = np.where(neg, ~u, u ^ top)
    else:
        k = u.copy()
    if order == DESC:
        k = ~k
    return k.astype(NP_BITS[dtype])


def stable_argsort_by_kdf(bits, dtype, order=ASC):
    return np.argsort(kdf_keys(bits, dtype, order), kind="stable")


def oracle_rank_by_records(bits, dtype, order=ASC):
    """The stable ranks of rs_sort_rank (radix_sort_rank.hpp:22-92, Listing-6 semantics) for 4-byte keys, computed the way
    SURVEY.md 8d pins cfg 4: the C restatement of radix_sort on {key, u32 index} RECORDS (rso_sort_records: sequential moves,
    no gather through the index -- ten times faster than rso_sort_rank at 10^8 keys, which is what lets the GPU suite compare
    whole rank arrays at production sizes).  Same kept columns, so the same half of the index buffer (radix_sort_rank.hpp:91
    = radix_sort.hpp:92); tests/test_oracle.py pins it against rso_sort_rank and the real reference.
    Returns (ranks uint32[n], result_in_second_half, info)."""
    assert DTYPE_SIZE[dtype] == 4
    src = np.ascontiguousarray(bits, dtype=NP_BITS[dtype])
    n = src.size
    recs = np.empty((n, 2), dtype=np.uint32)
    recs[:, 0] = src.view(np.uint32)
    recs[:, 1] = np.arange(n, dtype=np.uint32)
    aux = np.empty_like(recs)
    info = Info()
    r = oracle().rso_sort_records(ptr(recs), ptr(aux), n, 8, 0, dtype, order, C.byref(info))
    assert r in (0, 1)
    out = aux if r else recs
    return np.ascontiguousarray(out[:, 1]), r, info


def ranks_by_compound_sort(bits, dtype, order=ASC):
    """Stable ranks of 4-byte keys as np.sort of the 64-bit compounds (KDF key << 32 | index): the order rs_sort_rank's stable
    passes produce (radix_sort_rank.hpp:82-90) written down directly -- every compound is
    different, so any correct sort gives it.  NOT the restatement: it stands in for rso_sort_rank where that needs most of a
    minute per case (10^8 keys and more; the C loop gathers src[idx[j]] in every pass), and tests/test_oracle.py pins the two
    against each other (and both against the reference's record sort) at the sizes the C loop finishes in seconds.
    Returns (ranks uint32[n], result_in_second_half) -- the half by the parity of the kept columns (radix_sort_rank.hpp:91)."""
    assert DTYPE_SIZE[dtype] == 4
    u = np.ascontiguousarray(bits).view(np.uint32)
    n = u.size
    if dtype == F32:
        m = (u >> np.uint32(31)) * np.uint32(0x7FFFFFFF)
        m |= np.uint32(0x80000000)
        k = u ^ m
    elif dtype == I32:
        k = u ^ np.uint32(0x80000000)
    else:
        k = u.copy()
    if order == DESC:
        np.invert(k, out=k)
    kept = 0
    if n:
        diff = np.bitwise_or.reduce(k ^ k[0])
        kept = sum(1 for j in range(4) if (int(diff) >> (8 * j)) & 0xFF)
    presorted = n < 2 or bool(np.all(k[1:] >= k[:-1]))
    c = k.astype(np.uint64)
    c <<= np.uint64(32)
    c |= np.arange(n, dtype=np.uint64)
    c.sort()
    c &= np.uint64(0xFFFFFFFF)
    return c.astype(np.uint32), (0 if presorted else kept & 1)


def want_ranks(bits, dtype, order=ASC, big=1 << 24):
    """(ranks, result_in_second_half) of a rank sort of 4-byte keys for the GPU suite: the C restatement (rso_sort_rank) up to
    `big` keys, ranks_by_compound_sort above (pinned against it in tests/test_oracle.py) -- the C loop's gathers cost a minute
    per 10^8 keys, and the suite has ten minutes."""
    n = np.asarray(bits).size
    if n <= big:
        r, in_aux, _, _ = oracle_rank(bits, dtype, 4, order)
        return r.copy(), in_aux
    return ranks_by_compound_sort(bits, dtype, order)
