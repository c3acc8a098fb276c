"""Loop 1 of rs_sort_main (radix_sort.hpp:48-58) bin for bin, and the caller-supplied Hist (radix_sort.hpp:28-33).

rsx_capture_histogram brings the device's per-column digit counts back to the host; they must equal the oracle's
rso_histogram on the same input (every dtype, both orders, sizes around every tile / wave / vector boundary, skipped and
skewed columns), and -- combined with rsx_info exactly as include/radix_sort.hpp does (hist_post_state) -- reproduce what
the REAL rs_sort_main leaves in a caller's histogram (tests/golden/kat_table.json, hist_post).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "kat_table.json")) as f:
    KAT = json.load(f)
_CARRIER = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


def to_dev(bits):
    a = np.ascontiguousarray(bits)
    return torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()


def oracle_counts(a, dt, order):
    hist = np.zeros(256 * 8, dtype=np.uint64)
    nu = C.c_uint64()
    ol.oracle().rso_histogram(ol.ptr(a), a.size, ol.DTYPE_SIZE[dt], 0, dt, order, ol.ptr(hist), C.byref(nu))
    return hist[:256 * ol.DTYPE_SIZE[dt]], nu.value


def sort_with_capture(a, dt, order=ol.ASC):
    kb = ol.DTYPE_SIZE[dt]
    counts = np.full(256 * kb, 0xDEADBEEF, dtype=np.uint64)
    src = to_dev(a)
    aux = torch.zeros_like(src)
    rsa.capture_histogram(counts)
    try:
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    finally:
        rsa.capture_histogram(None)
    torch.cuda.synchronize()
    return res.cpu().numpy().view(ol.NP_BITS[dt]), info, counts


@pytest.mark.parametrize("dt", range(10), ids=ol.DTYPE_NAMES)
def test_histogram_bin_for_bin(dt):
    rng = np.random.default_rng(99 + dt)
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    sizes = [2, 3, 63, 64, 65, 255, 257, 1023, 1025, 4097, 8191, 16385, 65537, 100003, 262144 + 5, 1 << 20, (1 << 21) + 12345]
    for trial, n in enumerate(sizes):
        mask = full
        if trial % 4 == 1:
            for b in range(ol.DTYPE_SIZE[dt]):
                if rng.random() < 0.5:
                    mask &= ~(0xFF << (8 * b))
        if trial % 4 == 2:
            mask &= int(rng.integers(0, full, dtype=np.uint64, endpoint=True))
        if trial % 4 == 3:
            mask &= 0x0303030303030303 | 0xFF   # four values per upper byte: crowded counters
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        if trial % 5 == 4:
            a = a[ol.stable_argsort_by_kdf(a, dt)]   # pre-sorted: the counts are still delivered
        for order in (ol.ASC, ol.DESC):
            got, info, counts = sort_with_capture(a, dt, order)
            want, n_unsorted = oracle_counts(a, dt, order)
            assert np.array_equal(counts, want), (ol.DTYPE_NAMES[dt], n, hex(mask), order)
            assert (info.early_exit == 2) == (n_unsorted < 2)
            wres, _, _ = ol.oracle_sort(a, dt, order)
            assert np.array_equal(got, wres)


def hist_post_state(counts, info, kb, hvt_bytes):
    """include/radix_sort.hpp rsx_detail::hist_post_state in numpy (pre-zeroed Hist)."""
    out = np.zeros(256 * kb, dtype=np.uint64)
    kept = set() if info.early_exit else set(info.kept_columns())
    for j in range(kb):
        c = counts[256 * j:256 * j + 256]
        out[256 * j:256 * j + 256] = np.cumsum(c) if j in kept else c
    if hvt_bytes < 8:
        out &= np.uint64((1 << (8 * hvt_bytes)) - 1)
    return out


@pytest.mark.parametrize("row", KAT["hist_post"], ids=lambda r: "%s-n%d-m%s-p%d" % (r["dtype"], r["n"], r["mask"].lstrip("0"), r["presorted"]))
def test_caller_histogram_matches_the_reference(row):
    dt = row["dtype_code"]
    a = ol.splitmix_fill(row["n"], dt, row["seed"], int(row["mask"], 16))
    if row["presorted"]:
        a = a[ol.stable_argsort_by_kdf(a, dt)]
    got, info, counts = sort_with_capture(a, dt)
    assert "%016x" % ol.fnv1a64(got) == row["fnv_out"] and info.result_in_aux == row["result_in_aux"]
    post = hist_post_state(counts, info, ol.DTYPE_SIZE[dt], row["hvt_bytes"])
    assert "%016x" % ol.fnv1a64(post) == row["fnv_hist_u64"]


def test_capture_is_one_shot_and_checks_its_room():
    a = ol.splitmix_fill(50000, ol.U32, 3)
    src = to_dev(a)
    aux = torch.zeros_like(src)
    small = np.zeros(256, dtype=np.uint64)          # a u32 sort needs 1024 entries
    rsa.capture_histogram(small)
    with pytest.raises(rsa.RsxError, match="room for 256 entries"):
        rsa.radix_sort(src, aux, dtype=ol.U32)
    counts = np.zeros(1024, dtype=np.uint64)
    rsa.capture_histogram(counts)
    src = to_dev(a)
    rsa.radix_sort(src, aux, dtype=ol.U32)
    first = counts.copy()
    assert first.sum() == 4 * a.size
    counts[:] = 7
    src = to_dev(a)
    rsa.radix_sort(src, aux, dtype=ol.U32)            # not armed any more
    assert (counts == 7).all()
    # rank and pair sorts deliver the counts too
    ib = torch.zeros(2 * a.size, dtype=torch.int32, device="cuda")
    rsa.capture_histogram(counts)
    rsa.radix_sort_rank(to_dev(a), ib, dtype=ol.U32)
    assert np.array_equal(counts, first)
    counts[:] = 0
    vals = torch.arange(a.size, dtype=torch.int32, device="cuda")
    rsa.capture_histogram(counts)
    rsa.radix_sort_pairs(to_dev(a), torch.zeros_like(src), vals, torch.zeros_like(vals), dtype=ol.U32)
    assert np.array_equal(counts, first)


def test_python_binding_rejects_mismatched_buffers():
    """every scratch / aux tensor is checked for element size, length and device before its pointer is used."""
    src = torch.zeros(1000, dtype=torch.int32, device="cuda")
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_inplace_async(src, torch.zeros(1000, dtype=torch.int8, device="cuda"), dtype=ol.U32)
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort(src, torch.zeros(999, dtype=torch.int32, device="cuda"), dtype=ol.U32)
    vals = torch.zeros(1000, dtype=torch.int32, device="cuda")
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_pairs(src, torch.zeros(500, dtype=torch.int32, device="cuda"), vals, torch.zeros_like(vals), dtype=ol.U32)
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_pairs(src, torch.zeros_like(src), vals, torch.zeros(1000, dtype=torch.int64, device="cuda"), dtype=ol.U32)
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_pairs_inplace_async(src, torch.zeros_like(src), vals, torch.zeros(10, dtype=torch.int32, device="cuda"),
                                           dtype=ol.U32)
    recs = torch.zeros(1600, dtype=torch.uint8, device="cuda")
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_records_tagged(recs, torch.zeros(1599, dtype=torch.uint8, device="cuda"), 16, 8, ol.F32)
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_records_tagged(recs, torch.zeros_like(recs), 12, 8, ol.F32)        # 1600 % 12 != 0
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_records_tagged(recs, torch.zeros_like(recs), 16, 14, ol.F32)       # key sticks out of the record
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort(src.cpu(), torch.zeros_like(src), dtype=ol.U32)
    with pytest.raises(rsa.RsxError):
        rsa.radix_sort_rank(torch.zeros(10, dtype=torch.int16, device="cuda"), torch.zeros(20, dtype=torch.int32, device="cuda"),
                            dtype=ol.U32)
