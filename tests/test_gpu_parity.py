"""Parity of the HIP path against the oracle, through the C ABI, on a real MI355X.

Bit-exact bar: output bytes, returned buffer, kept-column list and early exits
must equal the CPU restatement (itself pinned to the reference in
tests/test_oracle.py) on the same seeded inputs; the committed golden hashes
(from the real reference headers) are checked directly as well.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "kat_table.json")) as f:
    KAT = json.load(f)

torch = pytest.importorskip("torch")

# bit patterns travel in same-width signed torch tensors; the rsx dtype code says what they mean
_CARRIER = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def to_dev(bits):
    a = np.ascontiguousarray(bits)
    return torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()


def to_bits(t, dt):
    return t.cpu().numpy().view(ol.NP_BITS[dt])


def gpu_sort(bits, dt, order=ol.ASC, fill=0x5A):
    src = to_dev(bits)
    aux = torch.full_like(src, int.from_bytes(bytes([fill]) * src.element_size(), "little"))   # every byte = fill
    res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    torch.cuda.synchronize()
    return to_bits(res, dt), info, to_bits(src, dt), to_bits(aux, dt)


def _id(row):
    return "%s-n%d-s%d-m%s-o%d" % (row["dtype"], row["n"], row["seed"], row["mask"].lstrip("0") or "0", row.get("order", 0))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.mark.parametrize("row", KAT["scalar"] + KAT["scalar_extra"], ids=_id)
def test_golden_table(row):
    """SURVEY.md section 4 known answers (hashes of the real reference's outputs)."""
    dt = row["dtype_code"]
    a = ol.splitmix_fill(row["n"], dt, row["seed"], int(row["mask"], 16))
    if row["n"] == 0:
        return
    got, info, _, _ = gpu_sort(a, dt, row["order"])
    assert "%016x" % ol.fnv1a64(got) == row["fnv_out"]
    assert info.result_in_aux == row["result_in_aux"]


@pytest.mark.parametrize("dt", range(10), ids=ol.DTYPE_NAMES)
def test_sweep_vs_oracle(dt):
    rng = np.random.default_rng(4321 + dt)
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    sizes = [2, 3, 63, 64, 65, 255, 256, 257, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 65535, 65536, 65537,
             100000, 300001]
    for trial, n in enumerate(sizes):
        mask = full
        if trial % 3 == 1:
            for b in range(ol.DTYPE_SIZE[dt]):
                if rng.random() < 0.5:
                    mask &= ~(0xFF << (8 * b))
        if trial % 3 == 2:
            mask &= int(rng.integers(0, full, dtype=np.uint64, endpoint=True))
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        for order in (ol.ASC, ol.DESC):
            want, want_aux, winfo = ol.oracle_sort(a, dt, order)
            got, info, _, _ = gpu_sort(a, dt, order)
            assert info.result_in_aux == want_aux, (n, hex(mask), order)
            assert info.kept_columns() == list(winfo.cols[:winfo.ncols])
            assert info.early_exit == winfo.early_exit
            assert np.array_equal(got, want), (n, hex(mask), order)


@pytest.mark.parametrize("dt", range(10), ids=ol.DTYPE_NAMES)
def test_small_sort_boundaries_and_general_path_at_small_n(dt, monkeypatch):
    """n * sizeof(key) <= 64 KiB takes the one-workgroup kernel (csrc/rsx_small.hpp): its size limits and wave-slice
    boundaries; and the same inputs through the general kernels (RSX_NO_SMALL_SORT=1), which they would otherwise never see."""
    rng = np.random.default_rng(99 + dt)
    size = ol.DTYPE_SIZE[dt]
    cap = 65536 // size
    full = (1 << (8 * size)) - 1
    sizes = sorted({2, 15, 16, 17, 1023, 1024, 1025, 1087, 1088, 1089, cap // 2 - 1, cap // 2, cap // 2 + 1, cap - 64, cap - 1, cap,
                    cap + 1})
    for trial, n in enumerate(sizes):
        mask = full if trial % 2 == 0 else full & ~(0xFF << (8 * int(rng.integers(0, size))))
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        if trial % 5 == 4:
            a = np.sort(a.view(ol.NP_BITS[dt]))[::-1].copy()      # descending bit patterns: heavy digit skew per wave slice
        for order in (ol.ASC, ol.DESC):
            want, want_aux, winfo = ol.oracle_sort(a, dt, order)
            for general in (False, True):
                if general:
                    monkeypatch.setenv("RSX_NO_SMALL_SORT", "1")
                else:
                    monkeypatch.delenv("RSX_NO_SMALL_SORT", raising=False)
                got, info, _, _ = gpu_sort(a, dt, order)
                assert info.result_in_aux == want_aux, (n, hex(mask), order, general)
                assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), (n, hex(mask), order, general)
                assert info.early_exit == winfo.early_exit
                assert np.array_equal(got, want), (n, hex(mask), order, general)
    monkeypatch.delenv("RSX_NO_SMALL_SORT", raising=False)


@pytest.mark.parametrize("dt", [ol.U32, ol.I64, ol.F32, ol.F64], ids=lambda d: ol.DTYPE_NAMES[d])
def test_default_tiles_at_medium_sizes(dt, monkeypatch):
    """Arrays below 96 default tiles take quarter tiles; RSX_NO_SMALL_TILES=1 sends the same inputs through the 32 Ki-key
    tiles (one partial tile, a few tiles, tile boundary +-1)."""
    monkeypatch.setenv("RSX_NO_SMALL_TILES", "1")
    rng = np.random.default_rng(7 + dt)
    tile = 32768 if ol.DTYPE_SIZE[dt] == 4 else 16384
    for n in (tile - 1, tile, tile + 1, 3 * tile + 17, 300001):
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), (1 << (8 * ol.DTYPE_SIZE[dt])) - 1)
        for order in (ol.ASC, ol.DESC):
            want, want_aux, winfo = ol.oracle_sort(a, dt, order)
            got, info, _, _ = gpu_sort(a, dt, order)
            assert info.result_in_aux == want_aux and np.array_equal(got, want), (n, order)


def test_contract_early_exits_leave_aux_untouched():
    # pre-sorted (with duplicates) -> src returned, aux byte-for-byte untouched (radix_sort.hpp:60-62)
    a = np.sort(ol.splitmix_fill(100000, ol.U32, 3, 0xFFFFF))
    got, info, src_after, aux_after = gpu_sort(a, ol.U32, fill=0x5A)
    assert info.early_exit == 2 and info.result_in_aux == 0 and info.ncols == 0
    assert np.array_equal(src_after, a) and np.all(aux_after == 0x5A5A5A5A)
    # all-equal input is pre-sorted too (appendix A item 3)
    got, info, _, aux_after = gpu_sort(np.full(70000, 0xDEADBEEF, dtype=np.uint32), ol.U32)
    assert info.early_exit == 2 and np.all(aux_after == 0x5A5A5A5A)
    # reverse-sorted input is NOT an early exit
    got, info, _, _ = gpu_sort(a[::-1].copy(), ol.U32)
    assert info.early_exit == 0 and np.array_equal(got, a)
    # a single descent anywhere (also across wave / block / vector boundaries) defeats the early exit
    for pos in (0, 1, 3, 4, 63, 255, 256, 1023, 1024, 4095, 4096, 50000, 99998):
        b = a.copy()
        b[pos], b[pos + 1] = a[-1], a[0]
        if b[pos] <= b[pos + 1]:
            continue
        want, want_aux, winfo = ol.oracle_sort(b, ol.U32)
        got, info, _, _ = gpu_sort(b, ol.U32)
        assert info.early_exit == 0 and np.array_equal(got, want) and info.result_in_aux == want_aux, pos


@pytest.mark.parametrize("mask,cols,in_aux", [(0x00FFFFFF, [0, 1, 2], 1), (0x0000FFFF, [0, 1], 0),
                                                 (0x000000FF, [0], 1), (0xFF00FF00, [1, 3], 0)])
def test_contract_column_skip_and_parity(mask, cols, in_aux):
    a = ol.splitmix_fill(200000, ol.U32, 77, mask)
    got, info, src_after, _ = gpu_sort(a, ol.U32)
    assert info.kept_columns() == cols and info.result_in_aux == in_aux
    assert np.array_equal(got, np.sort(a, kind="stable"))
    if len(cols) == 1:
        assert np.array_equal(src_after, a)   # one kept column: the input buffer is only read


def test_unaligned_device_pointers():
    """Sub-tensors that start off a 16-byte boundary (head/tail path of the histogram kernel)."""
    for dt in (ol.U8, ol.U16, ol.U32, ol.U64):
        a = ol.splitmix_fill(50021, dt, 5)
        for off in (1, 3):
            src_full = to_dev(np.concatenate([np.zeros(off, a.dtype), a]))
            aux_full = torch.zeros_like(src_full)
            res, info = rsa.radix_sort(src_full[off:], aux_full[off:], dtype=dt)
            torch.cuda.synchronize()
            want, want_aux, _ = ol.oracle_sort(a, dt)
            assert info.result_in_aux == want_aux and np.array_equal(to_bits(res, dt), want)


def test_float_order_and_test_int_fixture():
    vals = np.array([128.0, 646464.0, 0.0, -0.0, -0.5, 0.5, -128.0, -np.inf, np.nan, np.inf], dtype=np.float32)
    got, info, _, _ = gpu_sort(vals.view(np.uint32), ol.F32)
    assert [int(x) for x in got] == [0xff800000, 0xc3000000, 0xbf000000, 0x80000000, 0x00000000, 0x3f000000,
                                    0x43000000, 0x491dd400, 0x7f800000, 0x7fc00000]      # README.md:612-623
    ti = KAT["test_int"]
    a = np.fromfile(os.path.join(GOLDEN, ti["file"]), dtype=np.uint32)
    asc, info, _, _ = gpu_sort(a, ol.I32)
    assert "%016x" % ol.fnv1a64(asc) == ti["fnv_ascending"] and info.result_in_aux == ti["ascending_in_aux"]
    desc, info, _, _ = gpu_sort(asc, ol.I32, ol.DESC)
    assert "%016x" % ol.fnv1a64(desc) == ti["fnv_descending"] and info.result_in_aux == ti["descending_in_aux"]


def test_skewed_digits():
    """Degenerate digit distributions: the contention cases of the wave ranking and the look-back."""
    n = 500000
    rng = np.random.default_rng(9)
    cases = [np.full(n, 7, dtype=np.uint32),                                     # (caught by the early exit)
             np.where(rng.random(n) < 0.99, 0x01010101, 0x02020202).astype(np.uint32),   # two values
             (rng.integers(0, 4, n) * 0x40404040).astype(np.uint32),             # four values
             rng.zipf(1.3, n).astype(np.uint32),                                 # Zipf
             np.arange(n, dtype=np.uint32)[::-1].copy(),                         # strictly descending
             (np.arange(n, dtype=np.uint32) * 2654435761).astype(np.uint32)]     # multiplicative hash
    for a in cases:
        want, want_aux, winfo = ol.oracle_sort(a, ol.U32)
        got, info, _, _ = gpu_sort(a, ol.U32)
        assert info.result_in_aux == want_aux and np.array_equal(got, want)


@pytest.mark.parametrize("nhot", [1, 2, 3, 4, 6])
def test_hot_digit_columns(nhot):
    """Byte columns with `nhot` frequent digits (each a sixteenth of the keys or more) next to a uniform rest: the HOT
    kernels rank up to four of them with ballots, the others with the LDS counters.  Keys (default tiles: n above 96
    tiles), key + payload, and ranks of float keys (narrowed keys: the column comes through the flags)."""
    n = 96 * 32768 + 12345
    rng = np.random.default_rng(400 + nhot)
    share = 0.9 / nhot
    def column():
        hot = rng.choice(256, size=nhot, replace=False)
        pick = rng.random(n)
        col = rng.integers(0, 256, n)
        for k in range(nhot):
            col = np.where((pick >= k * share) & (pick < (k + 1) * share), hot[k], col)
        return col.astype(np.uint32)
    a = column() | (column() << 8) | (column() << 16) | (column() << 24)
    want, want_aux, winfo = ol.oracle_sort(a, ol.U32)
    got, info, _, _ = gpu_sort(a, ol.U32)
    assert info.result_in_aux == want_aux and np.array_equal(got, want)
    perm = ol.stable_argsort_by_kdf(a, ol.I32, ol.DESC)
    keys, keys_aux = to_dev(a), torch.zeros(n, dtype=torch.int32, device="cuda")
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    kr, vr, _ = rsa.radix_sort_pairs(keys, keys_aux, vals, torch.zeros_like(vals), dtype=ol.I32, order=ol.DESC)
    torch.cuda.synchronize()
    assert np.array_equal(to_bits(kr, ol.U32), a[perm]) and np.array_equal(vr.cpu().numpy().astype(np.int64), perm)
    wr, whalf, _, _ = ol.oracle_rank(a, ol.F32, 4)
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, rinfo = rsa.radix_sort_rank(to_dev(a), ib, dtype=ol.F32)
    torch.cuda.synchronize()
    assert rinfo.result_in_aux == whalf and np.array_equal(to_bits(ranks, ol.U32), wr)


# ---- key + payload and rank ----------------------------------------------------------------

def test_pairs_kv_pin():
    """SURVEY.md section 4 key+payload pin: {f32 key, u32 payload = i}, bits & 0xFFF000FF."""
    kv = KAT["kv"][0]
    n = kv["n"]
    k = ol.splitmix_fill(n, ol.U32, kv["seed"], int(kv["mask"], 16))
    keys, keys_aux = to_dev(k), torch.zeros(n, dtype=torch.int32, device="cuda")
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    vals_aux = torch.zeros_like(vals)
    kr, vr, info = rsa.radix_sort_pairs(keys, keys_aux, vals, vals_aux, dtype=ol.F32)
    torch.cuda.synchronize()
    assert info.result_in_aux == kv["result_in_aux"]
    assert "%016x" % ol.fnv1a64(to_bits(vr, ol.U32)) == kv["fnv_out_payloads"]
    assert "%016x" % ol.fnv1a64(to_bits(kr, ol.U32)) == kv["fnv_out_keys"]


@pytest.mark.parametrize("dt", [ol.U8, ol.U16, ol.U32, ol.U64, ol.I32, ol.F32, ol.F64], ids=lambda d: ol.DTYPE_NAMES[d])
@pytest.mark.parametrize("vbytes", [4, 8])
def test_pairs_vs_oracle_records(dt, vbytes):
    n = 150001
    k = ol.splitmix_fill(n, dt, 31 + dt, (1 << (8 * ol.DTYPE_SIZE[dt])) - 1 if dt not in (ol.U32, ol.F32) else 0xFFF000FF)
    vt = torch.int32 if vbytes == 4 else torch.int64
    keys, keys_aux = to_dev(k), to_dev(np.zeros_like(k))
    vals = (torch.arange(n, dtype=vt, device="cuda") * 3 + 1)
    vals_aux = torch.zeros_like(vals)
    kr, vr, info = rsa.radix_sort_pairs(keys, keys_aux, vals, vals_aux, dtype=dt)
    torch.cuda.synchronize()
    perm = ol.stable_argsort_by_kdf(k, dt)
    _, want_aux, winfo = ol.oracle_sort(k, dt)
    assert info.result_in_aux == want_aux
    assert np.array_equal(to_bits(kr, dt), k[perm])
    assert np.array_equal(vr.cpu().numpy(), perm.astype(np.int64) * 3 + 1)


@pytest.mark.parametrize("dt", [ol.U8, ol.I16, ol.U32, ol.F32, ol.U64, ol.F64], ids=lambda d: ol.DTYPE_NAMES[d])
@pytest.mark.parametrize("vbytes", [4, 8])
def test_small_pairs_and_rank_boundaries(dt, vbytes, monkeypatch):
    """Pairs and ranks that fit LDS twice (2 * n * (key + payload bytes) <= 128 KiB) take the one-workgroup kernel
    (csrc/rsx_small.hpp): its limits, both orders, and the same inputs through the general kernels."""
    kb = ol.DTYPE_SIZE[dt]
    cap = 131072 // (2 * (kb + vbytes))
    full = (1 << (8 * kb)) - 1
    vt = torch.int32 if vbytes == 4 else torch.int64
    rng = np.random.default_rng(5 + dt + vbytes)
    for trial, n in enumerate(sorted({2, 3, 64, 65, 1025, cap // 2 + 1, cap - 1, cap, cap + 1})):
        mask = full if trial % 2 else full & ~(0xFF << (8 * int(rng.integers(0, kb))))
        k = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        for order in (ol.ASC, ol.DESC):
            perm = ol.stable_argsort_by_kdf(k, dt, order)
            _, want_aux, winfo = ol.oracle_sort(k, dt, order)
            rwant, rhalf, rinfo, _ = ol.oracle_rank(k, dt, vbytes, order)
            for general in (False, True):
                if general:
                    monkeypatch.setenv("RSX_NO_SMALL_SORT", "1")
                else:
                    monkeypatch.delenv("RSX_NO_SMALL_SORT", raising=False)
                keys, keys_aux = to_dev(k), to_dev(np.zeros_like(k))
                vals = torch.arange(n, dtype=vt, device="cuda") * 5 + 2
                vals_aux = torch.zeros_like(vals)
                kr, vr, info = rsa.radix_sort_pairs(keys, keys_aux, vals, vals_aux, dtype=dt, order=order)
                torch.cuda.synchronize()
                assert info.result_in_aux == want_aux and info.early_exit == winfo.early_exit, (n, order, general)
                if not winfo.early_exit:
                    assert np.array_equal(to_bits(kr, dt), k[perm]), (n, order, general)
                    assert np.array_equal(vr.cpu().numpy(), perm.astype(np.int64) * 5 + 2), (n, order, general)
                ib = torch.full((2 * n,), -1, dtype=vt, device="cuda")
                ranks, rinf = rsa.radix_sort_rank(to_dev(k), ib, dtype=dt, order=order)
                torch.cuda.synchronize()
                assert rinf.result_in_aux == rhalf and rinf.early_exit == rinfo.early_exit, (n, order, general)
                assert np.array_equal(ranks.cpu().numpy().astype(np.uint64), rwant.astype(np.uint64)), (n, order, general)
    monkeypatch.delenv("RSX_NO_SMALL_SORT", raising=False)


@pytest.mark.parametrize("row", KAT["rank"], ids=_id)
def test_rank_golden(row):
    dt = row["dtype_code"]
    a = ol.splitmix_fill(row["n"], dt, row["seed"], int(row["mask"], 16))
    src = to_dev(a)
    ib = torch.full((2 * a.size,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(src, ib, dtype=dt)
    torch.cuda.synchronize()
    assert "%016x" % ol.fnv1a64(to_bits(ranks, ol.U32)) == row["fnv_stable_argsort"]
    if row["reference_is_correct"]:
        assert info.result_in_aux == row["reference_result_half"]
    assert np.array_equal(to_bits(src, dt), a)      # src is const (radix_sort_rank.hpp:97)


@pytest.mark.parametrize("dt", range(10), ids=ol.DTYPE_NAMES)
def test_rank_vs_oracle(dt):
    rng = np.random.default_rng(77 + dt)
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    for n, mask, ibytes in ((1, full, 4), (2, full, 4), (777, full, 8), (8192, full & 0xFFFF00FFFFFF00FF, 4),
                            (100003, full, 4), (100003, full & 0x00FF00FF00FF00FF, 8)):
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        for order in (ol.ASC, ol.DESC):
            want, whalf, winfo, _ = ol.oracle_rank(a, dt, ibytes, order)
            it = torch.int32 if ibytes == 4 else torch.int64
            ib = torch.full((2 * n,), -1, dtype=it, device="cuda")
            ranks, info = rsa.radix_sort_rank(to_dev(a), ib, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert info.result_in_aux == whalf and info.early_exit == winfo.early_exit
            assert np.array_equal(ranks.cpu().numpy().astype(np.uint64), want.astype(np.uint64)), (n, hex(mask), order)


@pytest.mark.parametrize("dt", [ol.U64, ol.I64, ol.F64, ol.U32, ol.F32, ol.I32, ol.U16])
def test_rank_narrowed_keys(dt, monkeypatch):
    """A rank sort hands on only the key bytes later passes need, in the narrowest type that holds them (u64 -> u32 -> u16
    -> u8).  Column patterns that narrow early, late, twice in one step and never; the same with the switch off."""
    kb = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * kb)) - 1
    masks = [full, full & 0xFF000000000000FF, full & 0x0000FFFF0000FFFF, full & 0x00FF00FF00FF00FF, full & 0xFFFFFFFF00000000,
             full & 0x000000FFFFFFFFFF, full & 0x00000000FFFF00FF, full & 0xFF00FF0000000000]
    for j, mask in enumerate(masks):
        if mask == 0:
            continue
        n = (300001, 40001, 1 << 20)[j % 3]
        a = ol.splitmix_fill(n, dt, 1000 + 7 * j + dt, mask)
        for order in (ol.ASC, ol.DESC):
            want, whalf, winfo, _ = ol.oracle_rank(a, dt, 4, order)
            for off in (False, True):
                if off:
                    monkeypatch.setenv("RSX_NO_NARROW_KEYS", "1")
                else:
                    monkeypatch.delenv("RSX_NO_NARROW_KEYS", raising=False)
                src = to_dev(a)
                ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
                ranks, info = rsa.radix_sort_rank(src, ib, dtype=dt, order=order)
                torch.cuda.synchronize()
                assert info.result_in_aux == whalf and info.ncols == winfo.ncols, (hex(mask), order, off)
                assert np.array_equal(to_bits(ranks, ol.U32), want), (hex(mask), order, off)
                assert np.array_equal(to_bits(src, dt), a)
    monkeypatch.delenv("RSX_NO_NARROW_KEYS", raising=False)


def test_rank_contract_presorted():
    a = np.sort(ol.splitmix_fill(5000, ol.U32, 1))
    ib = torch.full((10000,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(to_dev(a), ib, dtype=ol.U32)
    torch.cuda.synchronize()
    out = ib.cpu().numpy()
    assert info.early_exit == 2 and info.result_in_aux == 0
    assert np.array_equal(out[:5000], np.arange(5000)) and np.all(out[5000:] == -1)   # radix_sort_rank.hpp:52,:55-57


# ---- host-pointer entry points (what the C++ template wrapper calls) ----------------------------

def test_host_pointer_sort_and_rank():
    for dt in (ol.U32, ol.F32, ol.U64, ol.I16, ol.U8):
        a = ol.splitmix_fill(123457, dt, 50 + dt)
        want, want_aux, _ = ol.oracle_sort(a, dt)
        src, aux = a.copy(), np.full_like(a, 0x5A)
        res, info = rsa.radix_sort_host(src, aux, dt)
        assert info.result_in_aux == want_aux and np.array_equal(res, want)
        for ibytes, it in ((4, np.uint32), (8, np.uint64)):
            ib = np.zeros(2 * a.size, dtype=it)
            ranks, info = rsa.radix_sort_rank_host(a, ib, dt)
            wr, whalf, _, _ = ol.oracle_rank(a, dt, ibytes)
            assert info.result_in_aux == whalf and np.array_equal(ranks, wr)
    # pre-sorted host input: aux untouched
    a = np.sort(ol.splitmix_fill(5000, ol.U32, 2))
    src, aux = a.copy(), np.full_like(a, 0x5A5A5A5A)
    res, info = rsa.radix_sort_host(src, aux, ol.U32)
    assert res is src and info.early_exit == 2 and np.all(aux == 0x5A5A5A5A)
    # narrow IdxType (radix_tests.cpp:75 uses uint8_t)
    a = ol.splitmix_fill(200, ol.U8, 3)
    ib = np.full(400, 0xEE, dtype=np.uint8)
    ranks, info = rsa.radix_sort_rank_host(a, ib, ol.U8)
    wr, whalf, _, _ = ol.oracle_rank(a, ol.U8, 1)
    assert info.result_in_aux == whalf and np.array_equal(ranks, wr)


@pytest.mark.parametrize("dt", [ol.U8, ol.I8])
@pytest.mark.parametrize("order", [rsa.ASCENDING, rsa.DESCENDING])
def test_one_byte_keys_are_written_from_the_histogram(dt, order, monkeypatch):
    """Keys-only sorts of 1-byte keys never scatter: the sorted array is the histogram written out (rsx_fill_runs_kernel).
    Bit-exact against the oracle and against the scatter path (RSX_NO_FILL_RUNS=1), returned buffer and untouched source
    included; value sets with gaps, one dominant value, two values, sorted and constant inputs; an unaligned second buffer
    takes the scatter path."""
    import torch
    tdt = torch.uint8 if dt == ol.U8 else torch.int8
    rng = np.random.default_rng(7)
    cases = []
    for n in (17, 70001, (1 << 20) + 3, (1 << 24) + 5):
        cases.append(("uniform", ol.splitmix_fill(n, dt, 3 + n % 11)))
        cases.append(("gaps", ol.splitmix_fill(n, dt, 5, mask=0xA5)))
        a = ol.splitmix_fill(n, dt, 9)
        a[rng.random(n) < 0.97] = a[0]
        cases.append(("dominant", a))
        cases.append(("two values", (ol.splitmix_fill(n, dt, 11) & 1).astype(a.dtype) * 200))
        cases.append(("constant", np.full(n, 77, dtype=a.dtype)))
        cases.append(("sorted", np.sort(ol.splitmix_fill(n, dt, 13).view(np.int8 if dt == ol.I8 else np.uint8)).view(a.dtype)
                      if order == rsa.ASCENDING else ol.splitmix_fill(n, dt, 13)))
    for name, a in cases:
        want, want_aux, _ = ol.oracle_sort(a, dt, order)
        outs = []
        for no_fill in (False, True):
            if no_fill:
                monkeypatch.setenv("RSX_NO_FILL_RUNS", "1")
            src = torch.from_numpy(a.view(np.int8 if dt == ol.I8 else np.uint8).copy()).to("cuda").view(tdt)
            aux = torch.full_like(src, 0x5A)
            res, info = rsa.radix_sort(src, aux, dt, order)
            if no_fill:
                monkeypatch.delenv("RSX_NO_FILL_RUNS")
            got = res.cpu().numpy().view(a.dtype)
            assert info.result_in_aux == want_aux and np.array_equal(got, want), (name, a.size, no_fill)
            assert np.array_equal(src.cpu().numpy().view(a.dtype), a), (name, a.size, "source untouched")
            if not info.result_in_aux:
                assert bool((aux == 0x5A).all()), (name, a.size, "aux untouched")
            outs.append(got)
        assert np.array_equal(outs[0], outs[1])
    # a second buffer that is not 16-byte aligned: the scatter path, same result
    a = ol.splitmix_fill(100003, dt, 21)
    want, want_aux, _ = ol.oracle_sort(a, dt, order)
    src = torch.from_numpy(a.view(np.uint8).copy()).to("cuda").view(tdt)
    big = torch.zeros(a.size + 16, dtype=tdt, device="cuda")
    aux = big[3:3 + a.size]
    res, info = rsa.radix_sort(src, aux, dt, order)
    assert info.result_in_aux == want_aux and np.array_equal(res.cpu().numpy().view(a.dtype), want)


@pytest.mark.parametrize("dt,mask,const", [(ol.U32, 0x00FF0000, 0x12000034), (ol.I32, 0xFF000000, 0x00ABCDEF),
                                            (ol.F32, 0x000000FF, 0xC2F00000), (ol.U64, 0x0000FF0000000000, 0x0123000000456789),
                                            (ol.I16, 0x00FF, 0x8100), (ol.F64, 0x00FF000000000000, 0x4000000000000001)])
def test_keys_that_differ_in_one_byte_are_written_from_the_histogram(dt, mask, const, monkeypatch):
    """One kept column (radix_sort.hpp:64-70): the sorted array is written from that column's histogram, whatever the key
    width.  Large arrays decide on the device (speculative first pass and fill kernel both enqueued), small ones and
    RSX_NO_SPECULATION=1 on the host; RSX_NO_FILL_RUNS=1 scatters.  All bit-exact against the oracle, both orders."""
    import torch
    bits = ol.NP_BITS[dt]
    tdt = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[ol.DTYPE_SIZE[dt]]
    for n in (100003, (1 << 24) + 5):
        a = (ol.splitmix_fill(n, dt, 31 + n % 7, mask=mask) | np.array(const, dtype=bits)).astype(bits)
        for order in (rsa.ASCENDING, rsa.DESCENDING):
            want, want_aux, winfo = ol.oracle_sort(a, dt, order)
            for env in ({}, {"RSX_NO_SPECULATION": "1"}, {"RSX_NO_FILL_RUNS": "1"}):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                src = torch.from_numpy(a.view({1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[ol.DTYPE_SIZE[dt]]).copy()).to("cuda")
                aux = torch.full_like(src, 0x5A)
                res, info = rsa.radix_sort(src, aux, dt, order)
                for k in env:
                    monkeypatch.delenv(k)
                got = res.cpu().numpy().view(bits)
                assert info.ncols == 1 and info.result_in_aux == want_aux == 1, (n, order, env)
                assert np.array_equal(got, want), (n, order, env)
                assert np.array_equal(src.cpu().numpy().view(bits), a), (n, order, env, "source untouched")


@pytest.mark.parametrize("dt", [ol.U16, ol.I16])
@pytest.mark.parametrize("order", [rsa.ASCENDING, rsa.DESCENDING])
def test_two_byte_keys_as_one_sixteen_bit_digit(dt, order, monkeypatch):
    """Keys-only sorts of 2-byte keys from 2^20 keys on: the joint histogram of both bytes written out (rsx_joint16_kernel,
    rsx_fill16_kernel) -- result in `src` as after two passes; one kept column: from that byte's histogram into `aux`; sorted:
    nothing.  Bit-exact against the oracle and against the scatter passes (RSX_NO_FILL_RUNS=1), returned buffer included."""
    import torch
    rng = np.random.default_rng(11)
    cases = []
    for n in (1 << 20, (1 << 20) + 3, (1 << 24) + 5):
        cases.append(("uniform", ol.splitmix_fill(n, dt, 3 + n % 11)))
        cases.append(("high byte constant", ol.splitmix_fill(n, dt, 5, mask=0x00FF) | np.uint16(0x4200)))
        cases.append(("low byte constant", ol.splitmix_fill(n, dt, 6, mask=0xFF00) | np.uint16(0x0042)))
        cases.append(("few values", (ol.splitmix_fill(n, dt, 7) % 5).astype(np.uint16) * np.uint16(13001)))
        a = ol.splitmix_fill(n, dt, 9)
        a[rng.random(n) < 0.98] = a[0]
        cases.append(("dominant", a))
        cases.append(("constant", np.full(n, 0x8001, dtype=np.uint16)))
    for name, a in cases:
        want, want_aux, _ = ol.oracle_sort(a, dt, order)
        for no_fill in (False, True):
            if no_fill:
                monkeypatch.setenv("RSX_NO_FILL_RUNS", "1")
            src = torch.from_numpy(a.view(np.int16).copy()).to("cuda")
            aux = torch.full_like(src, 0x5A5A)
            res, info = rsa.radix_sort(src, aux, dt, order)
            if no_fill:
                monkeypatch.delenv("RSX_NO_FILL_RUNS")
            got = res.cpu().numpy().view(np.uint16)
            assert info.result_in_aux == want_aux and np.array_equal(got, want), (name, a.size, no_fill)
            if info.early_exit:
                assert bool((aux == 0x5A5A).all()) and np.array_equal(src.cpu().numpy().view(np.uint16), a), (name, "early exit")
    # unaligned buffers: the scatter passes
    a = ol.splitmix_fill((1 << 20) + 9, dt, 21)
    want, want_aux, _ = ol.oracle_sort(a, dt, order)
    big = torch.zeros(a.size + 16, dtype=torch.int16, device="cuda")
    src = big[1:1 + a.size]
    src.copy_(torch.from_numpy(a.view(np.int16).copy()))
    aux = torch.zeros(a.size, dtype=torch.int16, device="cuda")
    res, info = rsa.radix_sort(src, aux, dt, order)
    assert info.result_in_aux == want_aux and np.array_equal(res.cpu().numpy().view(np.uint16), want)


@pytest.mark.parametrize("dt", [ol.U8, ol.I16, ol.U32, ol.I32, ol.F32, ol.U64, ol.F64])
def test_small_host_arrays_through_pinned_staging(dt, monkeypatch):
    """rsx_sort / rsx_sort_rank on host arrays the one-launch kernels take: keys read from and results written to pinned memory
    by the kernel itself (rsx.hip, host_small_path).  Same results, returned buffer and untouched other buffer as the staged
    path (RSX_NO_HOST_SMALL=1) and as the oracle, at sizes on either side of the limits (64 KiB of keys; 128 KiB of keys and
    two index halves), both orders, every index width that holds n."""
    kb = ol.DTYPE_SIZE[dt]
    lim = 65536 // kb
    sizes = [2, 3, 257, 4097, lim - 1, lim, lim + 1]
    for n in sizes:
        for order in (rsa.ASCENDING, rsa.DESCENDING):
            a = ol.splitmix_fill(n, dt, 900 + n % 97 + dt, mask=0xFFFFFFFFFFFFFFFF if n % 2 else 0x0000FF00FF0000FF)
            want, want_aux, _ = ol.oracle_sort(a, dt, order)
            src, aux = a.copy(), np.full_like(a, 0x5A)
            res, info = rsa.radix_sort_host(src, aux, dt, order)
            assert info.result_in_aux == want_aux and np.array_equal(res, want), (n, order)
            other = src if info.result_in_aux else aux
            assert np.array_equal(other, a if info.result_in_aux else np.full_like(a, 0x5A)), (n, order, "other buffer")
            monkeypatch.setenv("RSX_NO_HOST_SMALL", "1")
            src2, aux2 = a.copy(), np.full_like(a, 0x5A)
            res2, info2 = rsa.radix_sort_host(src2, aux2, dt, order)
            monkeypatch.delenv("RSX_NO_HOST_SMALL")
            assert info2.result_in_aux == info.result_in_aux and np.array_equal(res2, res), (n, order, "staged path")
    for n in (2, 200, 256, 5000, 65536 // (kb + 4) - 1, 65536 // (kb + 4), 65536 // (kb + 4) + 1, 65536 // (kb + 8) + 1):
        a = ol.splitmix_fill(n, dt, 77 + n % 31 + dt, mask=0xFFFFFFFFFFFFFFFF if n % 3 else 0x000000FFFF000000)
        for ibytes, it in ((1, np.uint8), (2, np.uint16), (4, np.uint32), (8, np.uint64)):
            if ibytes < 8 and n > (1 << (8 * ibytes)):
                continue
            ib = np.full(2 * n, 0xEE, dtype=it)
            ranks, info = rsa.radix_sort_rank_host(a, ib, dt)
            wr, whalf, _, _ = ol.oracle_rank(a, dt, ibytes)
            assert info.result_in_aux == whalf and np.array_equal(ranks, wr), (n, ibytes)
    # pre-sorted small host input: aux untouched, src returned
    a = np.sort(ol.splitmix_fill(3000, ol.U32, 2))
    src, aux = a.copy(), np.full_like(a, 0x5A5A5A5A)
    res, info = rsa.radix_sort_host(src, aux, ol.U32)
    assert res is src and info.early_exit == 2 and np.all(aux == 0x5A5A5A5A) and np.array_equal(src, a)


def test_records_with_host_keys():
    """radix_tests.cpp:45-69 / :121-146 shapes: 16-byte records, 1-byte key from an opaque KeyFunc."""
    recs = np.zeros(8, dtype=np.dtype([("key", np.uint8), ("pad", np.uint8, 7), ("name", np.uint64)]))
    recs["key"] = [255, 45, 3, 45, 2, 45, 1, 255]
    recs["name"] = np.arange(8)
    for keys, expect in ((recs["key"].copy(), [6, 4, 2, 1, 3, 5, 0, 7]),
                         ((~recs["key"]).astype(np.uint8), [0, 7, 1, 3, 5, 2, 4, 6])):
        src, aux = recs.copy(), np.zeros_like(recs)
        res, info = rsa.radix_sort_records_host(src, aux, keys)
        assert info.result_in_aux == 1 and res is aux
        assert [int(x) for x in res["name"]] == expect
    # larger: 24-byte records, u32 keys with duplicates; compare with the oracle's record sort
    n = 70001
    k = ol.splitmix_fill(n, ol.U32, 8, 0xFF00FF)
    rec = np.zeros(n, dtype=np.dtype([("k", np.uint32), ("a", np.uint32), ("b", np.uint64), ("c", np.uint64)]))
    rec["k"], rec["a"], rec["b"], rec["c"] = k, np.arange(n), np.arange(n) * 7, ~np.arange(n, dtype=np.uint64)
    src, aux = rec.copy(), np.zeros_like(rec)
    res, info = rsa.radix_sort_records_host(src, aux, k.copy())
    s2, a2 = rec.copy(), np.zeros_like(rec)
    oinfo = ol.Info()
    r = ol.oracle().rso_sort_records(ol.ptr(s2), ol.ptr(a2), n, 24, 0, ol.U32, 0, C.byref(oinfo))
    assert info.result_in_aux == r and np.array_equal(res, a2 if r else s2)


@pytest.mark.parametrize("layout", ["f32@8of16", "i16@2of12", "f64@8of24", "u8@0of16", "u64@3of19"])
@pytest.mark.parametrize("order", [ol.ASC, ol.DESC])
def test_records_with_declared_key(layout, order):
    """rsx_sort_records_tagged[_device]: the key is a scalar field at a byte offset of the record; extraction,
    rank sort and gather on the device.  Bit-exact against the oracle's record sort, host and device pointers."""
    spec = {"f32@8of16": (ol.F32, 8, 16, 0xFFF000FF), "i16@2of12": (ol.I16, 2, 12, 0xFFFF), "f64@8of24": (ol.F64, 8, 24, ~0xFF),
            "u8@0of16": (ol.U8, 0, 16, 0xFF), "u64@3of19": (ol.U64, 3, 19, 0x0000FFFFFFFF00FF)}[layout]
    dt, off, rb, mask = spec
    kb = ol.DTYPE_SIZE[dt]
    for n in (2, 1000, 70001, 300007):
        rng = np.random.default_rng(n + off)
        rec = rng.integers(0, 256, size=(n, rb), dtype=np.uint8)
        keys = ol.splitmix_fill(n, dt, n + 5, mask & ((1 << (8 * kb)) - 1))
        rec[:, off:off + kb] = keys.view(np.uint8).reshape(n, kb)
        s2, a2 = rec.copy(), np.full_like(rec, 0x5A)
        oinfo = ol.Info()
        r = ol.oracle().rso_sort_records(ol.ptr(s2), ol.ptr(a2), n, rb, off, dt, order, C.byref(oinfo))
        want = a2 if r else s2
        # host pointers
        src, aux = rec.copy(), np.full_like(rec, 0x5A)
        res, info = rsa.radix_sort_records_tagged_host(src, aux, off, dt, order)
        assert info.result_in_aux == r and np.array_equal(res, want), (layout, n, "host")
        assert info.kept_columns() == list(oinfo.cols[:oinfo.ncols])
        # device pointers
        dsrc = torch.from_numpy(rec.copy().reshape(-1)).cuda()
        daux = torch.full_like(dsrc, 0x5A)
        dres, dinfo = rsa.radix_sort_records_tagged(dsrc, daux, rb, off, dt, order)
        torch.cuda.synchronize()
        assert dinfo.result_in_aux == r and (dres is daux) == bool(r)
        assert np.array_equal(dres.cpu().numpy().reshape(n, rb), want), (layout, n, "device")
    # pre-sorted by the key: source returned, auxiliary buffer untouched
    srt = want.copy()
    aux = np.full_like(srt, 0x5A)
    res, info = rsa.radix_sort_records_tagged_host(srt, aux, off, dt, order)
    assert res is srt and info.early_exit == 2 and (aux == 0x5A).all()


@pytest.mark.parametrize("dt", [ol.U32, ol.I16, ol.F32, ol.U64, ol.F64, ol.U8], ids=lambda d: ol.DTYPE_NAMES[d])
def test_inplace_async_sort(dt):
    """rsx_sort_inplace_async: every pass scheduled on the device, no host synchronisation, result always in the first buffer
    (odd numbers of kept columns are copied back); sorted input leaves the scratch buffer untouched."""
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    rng = np.random.default_rng(31 + dt)
    for trial, n in enumerate((2, 1000, 16384, 16385, 70001, 300001, 1 << 22)):
        mask = full if trial % 2 == 0 else full & ~(0xFF << (8 * int(rng.integers(0, size))))    # odd / even column counts
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        for order in (ol.ASC, ol.DESC):
            want, _, winfo = ol.oracle_sort(a, dt, order)
            buf = to_dev(a)
            scratch = torch.full_like(buf, int.from_bytes(bytes([0x5A]) * buf.element_size(), "little"))
            rsa.radix_sort_inplace_async(buf, scratch, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert np.array_equal(to_bits(buf, dt), want), (n, hex(mask), order)
    srt = to_dev(want)
    scratch = torch.full_like(srt, int.from_bytes(bytes([0x5A]) * srt.element_size(), "little"))
    keep = scratch.clone()
    rsa.radix_sort_inplace_async(srt, scratch, dtype=dt, order=order)
    torch.cuda.synchronize()
    assert np.array_equal(to_bits(srt, dt), want) and torch.equal(scratch, keep)


@pytest.mark.parametrize("dt,vbytes", [(ol.U32, 4), (ol.F32, 8), (ol.U64, 4), (ol.I64, 8), (ol.U16, 4), (ol.I8, 4)])
def test_inplace_async_pairs(dt, vbytes):
    """rsx_sort_pairs_inplace_async: keys and payloads end in the first buffers whatever the number of kept columns; no
    host synchronisation; a sorted input leaves the scratch buffers untouched."""
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    vt = torch.int32 if vbytes == 4 else torch.int64
    rng = np.random.default_rng(131 + dt + vbytes)
    cap = 131072 // (2 * (size + vbytes))
    for trial, n in enumerate((2, 777, cap, cap + 1, 70001, 300001, 1 << 21)):
        mask = full if trial % 2 == 0 else full & ~(0xFF << (8 * int(rng.integers(0, size))))    # odd / even column counts
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        for order in (ol.ASC, ol.DESC):
            perm = ol.stable_argsort_by_kdf(a, dt, order)
            keys, ks = to_dev(a), to_dev(np.zeros_like(a))
            vals = torch.arange(n, dtype=vt, device="cuda") * 5 + 1
            vs = torch.zeros_like(vals)
            rsa.radix_sort_pairs_inplace_async(keys, ks, vals, vs, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert np.array_equal(to_bits(keys, dt), a[perm]), (n, hex(mask), order)
            assert np.array_equal(vals.cpu().numpy(), perm.astype(np.int64) * 5 + 1), (n, hex(mask), order)
    ks.fill_(7)
    vs.fill_(9)
    before_k, before_v = keys.clone(), vals.clone()
    rsa.radix_sort_pairs_inplace_async(keys, ks, vals, vs, dtype=dt, order=order)     # sorted now: nothing moves
    torch.cuda.synchronize()
    assert torch.equal(keys, before_k) and torch.equal(vals, before_v) and bool((ks == 7).all()) and bool((vs == 9).all())


@pytest.mark.parametrize("n", [5000, 300001, 1 << 23])
def test_inplace_async_sort_in_a_hip_graph(n):
    """The whole sort captured into a graph once and replayed on new keys (three different column counts)."""
    s = torch.cuda.Stream()
    buf = torch.empty(n, dtype=torch.int32, device="cuda")
    scratch = torch.empty_like(buf)
    with torch.cuda.stream(s):
        rsa.fill_splitmix(buf, seed=1, stream=s)
        rsa.radix_sort_inplace_async(buf, scratch, dtype=ol.U32, stream=s)     # sizes the workspace outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_inplace_async(buf, scratch, dtype=ol.U32, stream=torch.cuda.current_stream())
    for seed, mask in ((11, 0xFFFFFFFF), (12, 0x00FFFFFF), (13, 0x0000FF00)):
        a = ol.splitmix_fill(n, ol.U32, seed, mask)
        buf.copy_(to_dev(a))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want, _, _ = ol.oracle_sort(a, ol.U32)
        assert np.array_equal(to_bits(buf, ol.U32), want), (n, seed)


@pytest.mark.parametrize("dt,itype", [(ol.U32, "int32"), (ol.F32, "int32"), (ol.I64, "int64"), (ol.U16, "int32")])
def test_rank_inplace_async_vs_oracle(dt, itype):
    """rsx_sort_rank_inplace_async (radix_sort_rank.hpp:97-112 without a host synchronisation): the oracle's ranks, always in
    the first half of the index buffer whatever the number of kept columns; sorted keys give 0 .. n-1; the keys untouched."""
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    idx_bytes = 4 if itype == "int32" else 8
    for n, mask in ((2, full), (1000, full), (5000, full & ~0xFF), (70001, full), (300001, full & ~(0xFF << 8)), (1200003, full)):
        a = ol.splitmix_fill(n, dt, 41 + n % 13, mask)
        for order in (ol.ASC, ol.DESC):
            src = to_dev(a)
            ib = torch.full((2 * n,), -7, dtype=getattr(torch, itype), device="cuda")
            ranks = rsa.radix_sort_rank_inplace_async(src, ib, dtype=dt, order=order)
            torch.cuda.synchronize()
            want = ol.oracle_rank(a, dt, idx_bytes, order)[0]
            assert np.array_equal(ranks.cpu().numpy().view(want.dtype), want), (n, hex(mask), order)
            assert np.array_equal(to_bits(src, dt), a)
    s = np.sort(ol.splitmix_fill(50000, dt, 3, full).view(ol.NP_BITS[dt]))
    if dt in (ol.U32, ol.U16):       # (sorted bit patterns are sorted keys for the unsigned types)
        ib = torch.full((100000,), -7, dtype=getattr(torch, itype), device="cuda")
        ranks = rsa.radix_sort_rank_inplace_async(to_dev(s), ib, dtype=dt)
        torch.cuda.synchronize()
        assert np.array_equal(ranks.cpu().numpy(), np.arange(50000))


@pytest.mark.parametrize("n", [5000, 300001])
def test_rank_inplace_async_in_a_hip_graph(n):
    """The device-scheduled rank sort captured once and replayed on new keys with 4, 3 and 1 kept columns: the ranks land in
    the first half every time."""
    s = torch.cuda.Stream()
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(s):
        rsa.fill_splitmix(keys, seed=1, stream=s)
        rsa.radix_sort_rank_inplace_async(keys, ib, dtype=ol.F32, stream=s)     # sizes the workspace outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_rank_inplace_async(keys, ib, dtype=ol.F32, stream=torch.cuda.current_stream())
    for seed, mask in ((11, 0xFFFFFFFF), (12, 0x00FFFFFF), (13, 0x0000FF00)):
        a = ol.splitmix_fill(n, ol.F32, seed, mask)
        keys.copy_(to_dev(a))
        ib.fill_(-3)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(ib[:n].cpu().numpy().view(np.uint32), ol.oracle_rank(a, ol.F32, 4)[0]), (n, seed)


@pytest.mark.parametrize("n", [5000, 300001, 1 << 22])
def test_graph_with_a_caller_owned_workspace_survives_larger_sorts(n):
    """a graph captured from rsx_sort_inplace_async refers to the library's cached workspace, which a later
    larger sort on the same (device, stream) frees and reallocates.  The *_ws forms keep every piece of device state in a
    workspace the caller owns: the graph is replayed after much larger sorts, after rsx_release_stream and rsx_release, on
    the capture stream and on another one, keys and key + payload."""
    s = torch.cuda.Stream()
    buf = torch.empty(n, dtype=torch.int32, device="cuda")
    scratch = torch.empty_like(buf)
    vals = torch.empty(n, dtype=torch.int64, device="cuda")
    vscratch = torch.empty_like(vals)
    ws = torch.empty(rsa.workspace_bytes(n, ol.U32, 0), dtype=torch.uint8, device="cuda")
    ws2 = torch.empty(rsa.workspace_bytes(n, ol.U32, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(rsa.RsxError, match="workspace"):
        rsa.radix_sort_inplace_async_ws(buf, scratch, ws[:1024], dtype=ol.U32)
    g, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_inplace_async_ws(buf, scratch, ws, dtype=ol.U32, stream=torch.cuda.current_stream())
    with torch.cuda.graph(g2, stream=s):
        rsa.radix_sort_pairs_inplace_async_ws(buf, scratch, vals, vscratch, ws2, dtype=ol.U32, stream=torch.cuda.current_stream())

    def check(seed, mask, pairs):
        a = ol.splitmix_fill(n, ol.U32, seed, mask)
        buf.copy_(to_dev(a))
        vals.copy_(torch.arange(n, dtype=torch.int64, device="cuda"))
        torch.cuda.synchronize()
        (g2 if pairs else g).replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_bits(buf, ol.U32), ol.oracle_sort(a, ol.U32)[0]), (n, seed, pairs)
        if pairs:
            assert np.array_equal(vals.cpu().numpy().astype(np.uint64), ol.stable_argsort_by_kdf(a, ol.U32).astype(np.uint64))

    check(11, 0xFFFFFFFF, False)
    check(12, 0x00FFFFFF, True)
    # much larger sorts on the capture stream and on the default stream: the library's own workspaces grow
    big = torch.empty(1 << 25, dtype=torch.int32, device="cuda")
    baux = torch.empty_like(big)
    with torch.cuda.stream(s):
        rsa.fill_splitmix(big, seed=5, stream=s)
        rsa.radix_sort(big, baux, dtype=ol.U32, stream=s)
        rsa.radix_sort_inplace_async(big, baux, dtype=ol.U32, stream=s)
    s.synchronize()                      # (the default stream is about to overwrite `big`)
    rsa.fill_splitmix(big, seed=6)
    rsa.radix_sort(big, baux, dtype=ol.U32)
    torch.cuda.synchronize()
    check(13, 0x0000FF00, False)
    check(14, 0xFFFFFFFF, True)
    rsa.release_stream(s)
    check(15, 0xFF00FFFF, False)
    rsa.lib().rsx_release()
    check(16, 0xFFFFFFFF, True)
    check(17, 0xFFFFFFFF, False)


# ---- BASELINE.json sizes: size-independent properties + full comparison where the oracle is quick enough ----

def test_full_size_2p28_u32_properties():
    """cfg 2: 2^28 u32 (seed 1).  Sortedness, multiset (histogram of histograms) and checksum invariants,
    plus bit-exact equality with the oracle on the whole array."""
    n = 1 << 28
    a = ol.splitmix_fill(n, ol.U32, 1)
    src = to_dev(a)
    aux = torch.empty_like(src)
    rsa.reload_env()          # (no back-off left over from an earlier test's called-off attempt: the route is asserted)
    res, info = rsa.radix_sort(src, aux, dtype=ol.U32)
    torch.cuda.synchronize()
    assert info.kept_columns() == [0, 1, 2, 3] and info.result_in_aux == 0
    # BASELINE.json's headline in its production geometry: no histogram, two MSB passes into slots (the second of two-byte
    # values), rsx_leaf16_kernel on slots of ~4096 values: everything below checks THAT route
    assert info.hybrid == 5, info.hybrid
    got = to_bits(res, ol.U32)
    assert np.all(got[:-1] <= got[1:])
    assert int(got.astype(np.uint64).sum()) == int(a.astype(np.uint64).sum())
    assert int(np.bitwise_xor.reduce(got)) == int(np.bitwise_xor.reduce(a))
    for j in range(4):
        assert np.array_equal(np.bincount((got >> (8 * j)) & 0xFF, minlength=256),
                              np.bincount((a >> (8 * j)) & 0xFF, minlength=256))
    want, want_aux, _ = ol.oracle_sort(a, ol.U32)
    assert np.array_equal(got, want)
    # idempotence: sorting the result again is the pre-sorted early exit
    res2, info2 = rsa.radix_sort(res, aux if res is src else src, dtype=ol.U32)
    torch.cuda.synchronize()
    assert info2.early_exit == 2 and res2 is res
