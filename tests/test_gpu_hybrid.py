"""One MSB pass (or two) and leaves (csrc/rsx_hybrid.hpp; README.md:647-650) against the oracle, through the C ABI.

The hybrid changes HOW the kept columns are gone through, never the result: output bytes, returned buffer, kept-column
list and early exits must equal the CPU restatement of rs_sort_main (radix_sort.hpp:31-93) exactly as on the
one-pass-per-column path.  `info.hybrid` says which way a sort went (0 one pass per kept column, 1 one MSB pass + leaves,
2 two MSB passes + leaves, 3 two levels whose (digit, digit) buckets were too large for leaves: LSB-first passes inside
the level-1 buckets, 4 two MSB passes + leaves with the second pass written into slack slots without a second count), so
that every branch is known to have run.
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

_CARRIER = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def gpu_sort(bits, dt, order=ol.ASC):
    a = np.ascontiguousarray(bits)
    src = torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()
    aux = torch.full_like(src, 0x5A5A5A5A5A5A5A5A >> (64 - 8 * a.itemsize))
    res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    torch.cuda.synchronize()
    return res.cpu().numpy().view(ol.NP_BITS[dt]), info


def check(a, dt, order, want_hybrid=None, what=""):
    want, want_aux, winfo = ol.oracle_sort(a, dt, order)
    got, info = gpu_sort(a, dt, order)
    assert info.result_in_aux == want_aux, (what, info.hybrid)
    assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), (what, info.hybrid)
    assert info.early_exit == winfo.early_exit, (what, info.hybrid)
    assert np.array_equal(got, want), (what, info.hybrid)
    if want_hybrid is not None:
        assert info.hybrid == want_hybrid, (what, info.hybrid, want_hybrid)
    return info


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.fixture(autouse=True)
def _two_levels_from_4mi(monkeypatch):
    """Two levels pay from 2^27 keys on and that is where the library starts using them; the tests lower the threshold to
    2^22 (RSX_TWO_LEVEL_MIN_LOG2) so that every branch runs at sizes the oracle sorts in a second."""
    monkeypatch.setenv("RSX_TWO_LEVEL_MIN_LOG2", "22")
    # (the routes these tests name start with the histogram; the sorts without one have their own tests at the end)
    monkeypatch.setenv("RSX_NO_BLIND", "1")
    yield


WIDE = [ol.U32, ol.I32, ol.F32, ol.U64, ol.I64, ol.F64]


@pytest.mark.parametrize("dt", WIDE, ids=[ol.DTYPE_NAMES[d] for d in WIDE])
def test_one_level_vs_oracle(dt):
    """Uniform keys of mid-size arrays: every bucket of the top byte fits a leaf."""
    rng = np.random.default_rng(700 + dt)
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    for n in (16385 * 4 // size + 1, 20000, 100000, 300001, 1000000, (1 << 21) + 3):
        for trial in range(3):
            mask = full
            if trial == 1:   # one constant byte somewhere below the top: the leaves skip it (radix_sort.hpp:64-70)
                mask &= ~(0xFF << (8 * int(rng.integers(0, size - 1))))
            if trial == 2:   # the top byte constant: the MSB pass goes by the highest KEPT column
                mask &= ~(0xFF << (8 * (size - 1)))
            a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
            for order in (ol.ASC, ol.DESC):
                info = check(a, dt, order, None, (n, hex(mask), order))
                if info.ncols >= 3 and dt not in (ol.F32, ol.F64):
                    assert info.hybrid == 1, (n, hex(mask), order, info.hybrid)


def test_one_level_big_leaves():
    """Buckets beyond the small leaf's 8 Ki keys take the leaf that fills the LDS (32 Ki four-byte keys)."""
    for n, dt in ((3000000, ol.U32), (5000000, ol.U32), (7000000, ol.I32), (3000000, ol.U64)):   # (the first: the 16 Ki-key shape)
        a = ol.splitmix_fill(n, dt, 77, (1 << (8 * ol.DTYPE_SIZE[dt])) - 1)
        for order in (ol.ASC, ol.DESC):
            check(a, dt, order, 1, (n, dt, order))


@pytest.mark.parametrize("dt", [ol.U32, ol.F32, ol.U64], ids=["u32", "f32", "u64"])
def test_two_levels_vs_oracle(dt):
    """Arrays whose top-byte buckets exceed a leaf: a second pass inside the buckets, then 65536 leaves."""
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    for n in ((1 << 23) + 12345, (1 << 24) + 1):
        masks = [full]
        if size == 8:
            masks.append(0x0000FFFFFFFFFFFF)       # six kept columns
            masks.append(0x00FF00FFFF00FFFF)       # the two top KEPT columns are not adjacent
        else:
            masks.append(0xFFFF00FF if False else full)
        for mask in masks:
            if dt == ol.F32:
                # floats with flat top bytes: random bit patterns (NaNs, infinities and denormals among them)
                a = ol.splitmix_fill(n, dt, 6, mask)
            else:
                a = ol.splitmix_fill(n, dt, 5, mask)
            for order in (ol.ASC, ol.DESC):
                check(a, dt, order, 2, (n, hex(mask), order))


def test_two_levels_with_clustered_top_bytes_fall_back_inside_the_buckets():
    """Flat top-byte and second-byte histograms, but the two bytes are equal in every key: the (digit, digit) buckets are
    256 times the estimate and do not fit a leaf -- the segmented passes go on LSB first inside the level-1 buckets."""
    n = (1 << 23) + 777
    a = ol.splitmix_fill(n, ol.U32, 11, 0xFFFFFFFF).view(np.uint32).copy()
    a = (a & np.uint32(0xFF00FFFF)) | ((a >> np.uint32(8)) & np.uint32(0x00FF0000))
    for order in (ol.ASC, ol.DESC):
        check(a, ol.U32, order, 3, order)
    b = ol.splitmix_fill(n, ol.U64, 12, 0xFFFFFFFFFFFFFFFF).view(np.uint64).copy()
    b = (b & np.uint64(0xFF00FFFFFFFFFFFF)) | ((b >> np.uint64(8)) & np.uint64(0x00FF000000000000))
    check(b, ol.U64, ol.ASC, 3, "u64")


def test_skewed_top_bytes_take_one_pass_per_column():
    """98 % of the keys on one top byte: no bucket fits anything, the plan must say so before the first pass."""
    rng = np.random.default_rng(5)
    for n in (100000, 1000000, (1 << 23) + 5):
        a = ol.splitmix_fill(n, ol.U32, 9, 0xFFFFFFFF).view(np.uint32).copy()
        heavy = rng.random(n) < 0.98
        a[heavy] = (a[heavy] & np.uint32(0x00FFFFFF)) | np.uint32(0x42000000)
        info = check(a, ol.U32, ol.ASC, None, n)
        if n >= 1000000:
            assert info.hybrid == 0, (n, info.hybrid)
    # floats of one magnitude (cfg 4 (ii)-like): top byte skewed
    f = (rng.random(1 << 22, dtype=np.float32) * 2 - 1).view(np.uint32)
    check(f, ol.F32, ol.ASC, None, "floats in [-1, 1)")
    check(f, ol.F32, ol.DESC, None, "floats in [-1, 1) descending")


def test_hybrid_early_exits_and_few_columns():
    """Sorted input, two kept columns, one kept column: the hybrid must not change the reference's exits (radix_sort.hpp:60-70)."""
    n = 1000000
    a = np.sort(ol.splitmix_fill(n, ol.U32, 3, 0xFFFFFFFF).view(np.uint32))
    info = check(a, ol.U32, ol.ASC, 0, "sorted")
    assert info.early_exit == 2
    check(a[::-1].copy(), ol.U32, ol.ASC, None, "reversed")
    check(ol.splitmix_fill(n, ol.U32, 4, 0x0000FFFF), ol.U32, ol.ASC, 0, "two columns")
    check(ol.splitmix_fill(n, ol.U32, 4, 0x00FF0000), ol.U32, ol.ASC, 0, "one column")
    check(ol.splitmix_fill(n, ol.U32, 4, 0xFF00FF0F), ol.U32, ol.DESC, 1, "three columns, top byte kept")


def test_hybrid_off_is_the_same_sort(monkeypatch):
    """RSX_NO_HYBRID=1 (one pass per kept column) and the default give the same bytes in the same buffer."""
    a = ol.splitmix_fill(3000001, ol.U32, 21, 0xFFFFFFFF)
    got1, info1 = gpu_sort(a, ol.U32)
    monkeypatch.setenv("RSX_NO_HYBRID", "1")
    rsa.reload_env()
    try:
        got0, info0 = gpu_sort(a, ol.U32)
    finally:
        monkeypatch.delenv("RSX_NO_HYBRID")
        rsa.reload_env()
    assert info1.hybrid == 1 and info0.hybrid == 0
    assert info1.result_in_aux == info0.result_in_aux
    assert np.array_equal(got0, got1)


@pytest.mark.parametrize("dt", [ol.U32, ol.I32, ol.U64], ids=["u32", "i32", "u64"])
def test_slack_two_levels_vs_oracle(dt):
    """From 2^26 keys on the second pass is tried WITHOUT counting first: every (digit, digit) bucket has a slot of 1.25
    times its expected size in a scratch array, the bucket sizes come off the look-back chain, the leaves gather from the
    slots (info.hybrid == 4)."""
    n = (1 << 26) + 4321
    a = ol.splitmix_fill(n, dt, 17, (1 << (8 * ol.DTYPE_SIZE[dt])) - 1)
    for order in (ol.ASC, ol.DESC):
        check(a, dt, order, 4, (dt, order))


def test_slack_overflow_falls_back_to_the_counted_pass():
    """Flat byte histograms (the plan chooses two levels) but one (digit, digit) bucket holds four times its share: its slot
    overflows, the attempt is discarded and the counted second pass runs from the untouched pass-1 output (hybrid 2); and
    with the top two bytes equal in every key the counted path finds its buckets too large for leaves as well (hybrid 3)."""
    n = (1 << 26) + 99
    a = ol.splitmix_fill(n, ol.U32, 23, 0xFFFFFFFF).view(np.uint32).copy()
    a[1000:1000 + 3000 * 7:7] = (a[1000:1000 + 3000 * 7:7] & np.uint32(0x0000FFFF)) | np.uint32(0x12340000)
    check(a, ol.U32, ol.ASC, 2, "one heavy (digit, digit) bucket")
    # the LAST slot far too small (a hundred thousand keys with the top sixteen bits all ones): its runs must not be written
    # behind the end of the scratch array
    c = ol.splitmix_fill(n, ol.U32, 25, 0xFFFFFFFF).view(np.uint32).copy()
    c[5:5 + 100000 * 3:3] |= np.uint32(0xFFFF0000)
    check(c, ol.U32, ol.ASC, 3, "heavy last (digit, digit) bucket")
    b = ol.splitmix_fill(n, ol.U32, 24, 0xFFFFFFFF).view(np.uint32).copy()
    b = (b & np.uint32(0xFF00FFFF)) | ((b >> np.uint32(8)) & np.uint32(0x00FF0000))
    check(b, ol.U32, ol.DESC, 3, "top two bytes equal")


def test_slack_off_is_the_counted_pass(monkeypatch):
    a = ol.splitmix_fill((1 << 26) + 5, ol.U32, 31, 0xFFFFFFFF)
    monkeypatch.setenv("RSX_NO_SLACK", "1")
    check(a, ol.U32, ol.ASC, 2, "RSX_NO_SLACK=1")


def test_a_dominant_digit_in_a_low_column_keeps_the_pass_kernels():
    """Evenly spread top bytes, but 99 % of the keys share their LOW byte: a leaf would put 64 lanes on one LDS counter in
    that column, the pass kernels rank such digits with ballots (HOT) -- the plan must keep the sort on them (hybrid 0)."""
    rng = np.random.default_rng(8)
    for n in (1000000, (1 << 23) + 9):
        a = ol.splitmix_fill(n, ol.U32, 33, 0xFFFFFFFF).view(np.uint32).copy()
        sel = rng.random(n) < 0.99
        a[sel] &= np.uint32(0xFFFFFF00)
        check(a, ol.U32, ol.ASC, 0, n)
        b = a.copy()
        b[sel] = (b[sel] & np.uint32(0xFFFF00FF)) | np.uint32(0x00004200)      # ... or their second byte
        check(b, ol.U32, ol.DESC, 0, n)


def _dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()


@pytest.mark.parametrize("dt", [ol.F32, ol.U32, ol.I32], ids=["f32", "u32", "i32"])
def test_rank_sort_two_levels_vs_oracle(dt):
    """Rank sorts of 4-byte keys that spread over their top two bytes (cfg 4 (i)): two MSB passes of (key, index), the second
    into slack slots, and leaves that carry the pair as one 8-byte value (rsx_leaf_pairs_kernel); ranks, returned half and
    kept columns are the oracle's (radix_sort_rank.hpp:22-92, Listing 6 semantics)."""
    n = (1 << 23) + 4567
    for mask in (0xFFFFFFFF,):
        a = ol.splitmix_fill(n, dt, 6, mask)
        for order in (ol.ASC, ol.DESC):
            ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
            ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=dt, order=order)
            torch.cuda.synchronize()
            want, whalf, winfo, _ = ol.oracle_rank(a, dt, 4, order)
            assert info.hybrid == 4, (dt, order, info.hybrid)
            assert info.result_in_aux == whalf
            assert info.kept_columns() == list(winfo.cols[:winfo.ncols])
            assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), (dt, order)


def test_pairs_sort_two_levels_vs_oracle():
    """Key + payload sorts the same way: keys and payloads where the parity rule says, stable."""
    n = (1 << 23) + 321
    a = ol.splitmix_fill(n, ol.U32, 9, 0xFFFFFFFF).view(np.uint32).copy()
    a &= np.uint32(0xFFFFFF0F)                      # duplicates: stability is observable through the payloads
    for order in (ol.ASC, ol.DESC):
        keys = _dev(a)
        vals = torch.arange(n, dtype=torch.int32, device="cuda") * 3 + 1
        k, v, info = rsa.radix_sort_pairs(keys, torch.zeros_like(keys), vals, torch.zeros_like(vals), dtype=ol.U32, order=order)
        torch.cuda.synchronize()
        perm = ol.stable_argsort_by_kdf(a, ol.U32, order)
        assert info.hybrid == 4, info.hybrid
        assert np.array_equal(k.cpu().numpy().view(np.uint32), a[perm])
        assert np.array_equal(v.cpu().numpy().astype(np.int64), perm.astype(np.int64) * 3 + 1)


def test_rank_sort_slot_overflow_falls_back():
    """One (digit, digit) bucket four times its share: the slot overflows, nothing has been written, the ordinary passes run."""
    n = (1 << 23) + 99
    a = ol.splitmix_fill(n, ol.U32, 23, 0xFFFFFFFF).view(np.uint32).copy()
    a[1000:1000 + 1200 * 7:7] = (a[1000:1000 + 1200 * 7:7] & np.uint32(0x0000FFFF)) | np.uint32(0x12340000)
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=ol.U32)
    torch.cuda.synchronize()
    want, whalf, _, _ = ol.oracle_rank(a, ol.U32, 4)
    assert info.hybrid == 0 and info.result_in_aux == whalf
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize("dt", [ol.U32, ol.F32, ol.I32], ids=["u32", "f32", "i32"])
def test_rank_and_pairs_one_level_vs_oracle(dt):
    """Mid-size rank sorts and key + payload sorts of 4-byte keys: one MSB pass of (key, payload) and the pairs' leaves."""
    rng = np.random.default_rng(40 + dt)
    for n in (20001, 100000, 500001, 1000000):
        for mask in (0xFFFFFFFF, 0xFFFF00FF):
            a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
            for order in (ol.ASC, ol.DESC):
                ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
                ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=dt, order=order)
                torch.cuda.synchronize()
                want, whalf, winfo, _ = ol.oracle_rank(a, dt, 4, order)
                assert info.result_in_aux == whalf and info.kept_columns() == list(winfo.cols[:winfo.ncols]), (n, hex(mask), order)
                assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), (n, hex(mask), order, info.hybrid)
                if dt != ol.F32:
                    assert info.hybrid == 1, (n, hex(mask), order, info.hybrid)
        keys = _dev(a)
        vals = torch.arange(n, dtype=torch.int32, device="cuda") * 7 + 3
        k, v, info = rsa.radix_sort_pairs(keys, torch.zeros_like(keys), vals, torch.zeros_like(vals), dtype=dt)
        torch.cuda.synchronize()
        perm = ol.stable_argsort_by_kdf(a, dt)
        assert np.array_equal(k.cpu().numpy().view(np.uint32), a.view(np.uint32)[perm]), n
        assert np.array_equal(v.cpu().numpy().astype(np.int64), perm.astype(np.int64) * 7 + 3), n


# ---- sorts without a histogram (rsx_blind_precheck_kernel ... ; info.hybrid == 5) -------------------------------------------
@pytest.fixture
def blind_on(monkeypatch):
    monkeypatch.setenv("RSX_NO_BLIND", "0")   # (also makes the context forget earlier attempts: rsx_reload_env)
    yield monkeypatch


@pytest.mark.parametrize("dt", WIDE, ids=lambda d: ol.DTYPE_NAMES[d] if hasattr(ol, "DTYPE_NAMES") else str(d))
def test_blind_two_levels_vs_oracle(dt, blind_on):
    """Keys that spread over all their columns: no histogram kernel runs, the sample proves what the plan needs, both MSB
    passes write into slots; result, returned buffer, kept columns as the oracle's."""
    for n in ((1 << 22) + 777, 6000001):
        for order in (ol.ASC, ol.DESC):
            a = ol.splitmix_fill(n, dt, 1000 + n % 97 + order, (1 << (8 * ol.DTYPE_SIZE[dt])) - 1)
            check(a, dt, order, 5, (n, dt, order))


def test_blind_called_off_by_the_sample(blind_on):
    """What the sample cannot prove sends the sort down the ordinary path: sorted input (early exit, `aux` untouched), a
    constant column (a different returned buffer), a dominant top digit, a hot low column."""
    n = (1 << 22) + 4321
    base = ol.splitmix_fill(n, ol.U32, 77, 0xFFFFFFFF).view(np.uint32)
    srt = np.sort(base)
    info = check(srt, ol.U32, ol.ASC, 0, "sorted")
    assert info.early_exit == 2
    for name, a in (("low byte constant", base & np.uint32(0xFFFFFF00)),
                    ("column 2 constant", base & np.uint32(0xFF00FFFF)),
                    ("half the keys in top digit 0", np.where(base & 1 == 1, base & np.uint32(0x00FFFFFF), base)),
                    ("low byte 7/8 zero", np.where(base & 0x700 != 0, base & np.uint32(0xFFFFFF00), base))):
        for _ in range(3):   # (the attempt, the skipped sort after it, the next attempt)
            info = check(np.ascontiguousarray(a.astype(np.uint32)), ol.U32, ol.ASC, None, name)
            assert info.hybrid != 5, (name, info.hybrid)


def test_blind_called_off_by_an_overflowing_slot(blind_on):
    """Clustering the sample does not see: a top digit with 1.6 times its share overflows its level-1 slot, a (digit, digit)
    pair with five times its share its level-2 slot.  The attempt has only read the caller's array: the ordinary sort follows
    and the next sort of this context does not try."""
    n = (1 << 22) + 99
    base = ol.splitmix_fill(n, ol.U32, 78, 0xFFFFFFFF).view(np.uint32).copy()
    a = base.copy()
    extra = np.flatnonzero((a >> 24) == 0x11)[: int(0.6 * n / 256)]
    a[extra] = (a[extra] & np.uint32(0x00FFFFFF)) | np.uint32(0x77000000)
    b = base.copy()
    b[5000:5000 + 300 * 11:11] = (b[5000:5000 + 300 * 11:11] & np.uint32(0x0000FFFF)) | np.uint32(0x43210000)
    for name, x in (("level 1", a), ("level 2", b)):
        blind_on.setenv("RSX_NO_BLIND", "0")
        info = check(x, ol.U32, ol.ASC, None, name)
        assert info.hybrid != 5, (name, info.hybrid)
        u = ol.splitmix_fill(n, ol.U32, 79, 0xFFFFFFFF)
        assert check(u, ol.U32, ol.ASC, None, "after").hybrid != 5     # (skipped: one sort after the first failure)
        assert check(u, ol.U32, ol.ASC, None, "after").hybrid == 5     # (tried again, and kept)


def test_blind_whole_result_verification(blind_on):
    """RSX_VERIFY=2 brackets the sort with checksums whatever its route."""
    blind_on.setenv("RSX_VERIFY", "2")
    a = ol.splitmix_fill((1 << 22) + 5, ol.U32, 80, 0xFFFFFFFF)
    check(a, ol.U32, ol.ASC, 5, "verify=2")


@pytest.mark.parametrize("dt", [ol.U32, ol.F32, ol.I32], ids=["u32", "f32", "i32"])
def test_blind_rank_and_pairs_vs_oracle(dt, blind_on):
    """Rank sorts and key + payload sorts of 4-byte keys the same way: the first MSB pass makes the indices (or reads the
    caller's payloads) and writes into slots like the second; ranks / pairs where the parity rule says."""
    n = (1 << 23) + 4097
    a = ol.splitmix_fill(n, dt, 91 + dt, 0xFFFFFFFF)
    for order in (ol.ASC, ol.DESC):
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=dt, order=order)
        torch.cuda.synchronize()
        want, whalf, winfo, _ = ol.oracle_rank(a, dt, 4, order)
        assert info.hybrid == 5, (dt, order, info.hybrid)
        assert info.result_in_aux == whalf and info.kept_columns() == list(winfo.cols[:winfo.ncols])
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), (dt, order)
        keys = _dev(a)
        vals = torch.arange(n, dtype=torch.int32, device="cuda") * 5 + 2
        k, v, info = rsa.radix_sort_pairs(keys, torch.zeros_like(keys), vals, torch.zeros_like(vals), dtype=dt, order=order)
        torch.cuda.synchronize()
        perm = ol.stable_argsort_by_kdf(a, dt, order)
        assert info.hybrid == 5 and not info.result_in_aux, (dt, order, info.hybrid)
        assert np.array_equal(k.cpu().numpy().view(np.uint32), a.view(np.uint32)[perm])
        assert np.array_equal(v.cpu().numpy().astype(np.int64), perm.astype(np.int64) * 5 + 2)


def test_blind_rank_called_off(blind_on):
    """A slot overflows under a rank sort: the index buffer has not been written, the ordinary passes follow."""
    n = (1 << 23) + 99
    a = ol.splitmix_fill(n, ol.U32, 23, 0xFFFFFFFF).view(np.uint32).copy()
    a[1000:1000 + 1200 * 7:7] = (a[1000:1000 + 1200 * 7:7] & np.uint32(0x0000FFFF)) | np.uint32(0x12340000)
    want, whalf, _, _ = ol.oracle_rank(a, ol.U32, 4)
    for _ in range(2):
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=ol.U32)
        torch.cuda.synchronize()
        assert info.hybrid != 5 and info.result_in_aux == whalf
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize("mask,ncols", [(0xFFFFFFFFFF, 5), (0xFFFFFFFF, 4), (0x00FFFFFFFFFF00FF, 6)])
def test_blind_with_constant_columns(mask, ncols, blind_on):
    """8-byte keys with constant byte columns (BASELINE.json's cfg 3): the sample proves the kept ones, the level-1 pass checks
    the others on every key; kept columns and returned buffer as the oracle's (an odd number of kept columns ends in aux)."""
    n = (1 << 22) + 555
    for dt, order in ((ol.U64, ol.ASC), (ol.U64, ol.DESC), (ol.I64, ol.ASC)):
        a = ol.splitmix_fill(n, dt, 300 + ncols, mask)
        info = check(a, dt, order, 5, (hex(mask), dt, order))
        assert info.ncols == ncols


def test_blind_constant_column_disproved_by_one_key(blind_on):
    """One key, nowhere near the sampled places, differs in a column the sample took for constant: the level-1 pass finds
    it, the attempt is called off, the ordinary sort keeps that column."""
    n = (1 << 22) + 555
    a = ol.splitmix_fill(n, ol.U64, 310, 0xFFFFFFFFFF).copy()
    for where in (1000, n // 2 + 777, n - 3):
        b = a.copy()
        b[where] |= np.uint64(0x0100000000000000)
        info = check(b, ol.U64, ol.ASC, None, where)
        assert info.hybrid != 5 and info.ncols == 6, (where, info.hybrid, info.ncols)
        blind_on.setenv("RSX_NO_BLIND", "0")   # (forget the back-off)


@pytest.mark.parametrize("blind", ["0", "1"], ids=["histogram-first", "without-histogram"])
def test_leaves_of_8_byte_keys_sorted_by_their_top_columns(blind, monkeypatch):
    """Leaves of 8-byte keys with five or more columns left sort by the top three of them and finish with odd-even
    transposition on whole keys (rsx_hybrid.hpp); keys that cluster in exactly those columns (sixteen values per byte:
    thousands of equal 24-bit prefixes per leaf) must fall back to all columns.  Both against the oracle, with the
    shortcut on and off."""
    monkeypatch.setenv("RSX_NO_BLIND", "0" if blind == "1" else "1")
    for n in (1000003, (1 << 22) + 999):
        for mask in (0xFFFFFFFFFFFFFFFF, 0xFFFF0F0F0FFFFFFF, 0xFF0F0F0FFFFFFFFF, 0xFFFF000103FFFFFF):
            for dt, order in ((ol.U64, ol.ASC), (ol.F64, ol.DESC)):
                a = ol.splitmix_fill(n, dt, 400 + (mask & 0xFF00) // 256, mask)
                info = check(a, dt, order, None, (n, hex(mask), dt, order))
                monkeypatch.setenv("RSX_NO_LEAF_PREFIX", "1")
                info2 = check(a, dt, order, None, (n, hex(mask), dt, order, "no prefix"))
                monkeypatch.delenv("RSX_NO_LEAF_PREFIX")
                assert info.hybrid == info2.hybrid


def test_blind_fuzz(blind_on):
    """Seeded random inputs at the sizes where sorts may skip the histogram: constant byte columns (several, anywhere), random
    bit masks, duplicates, nearly sorted arrays, a few stray keys in otherwise constant columns.  Every result against the
    oracle; the route must have been taken by a good share of the cases (the others were called off, which is the point)."""
    rng = np.random.default_rng(2026)
    taken = 0
    cases = 48
    for k in range(cases):
        dt = [ol.U32, ol.I32, ol.F32, ol.U64, ol.I64, ol.F64, ol.U64, ol.U64][k % 8]
        size = ol.DTYPE_SIZE[dt]
        full = (1 << (8 * size)) - 1
        n = int(rng.integers(1 << 22, 6000000))
        mask = full
        style = int(rng.integers(0, 6))
        if style in (1, 5):                       # constant byte columns
            for b in range(size):
                if rng.random() < 0.35:
                    mask &= ~(0xFF << (8 * b))
        elif style == 2:                          # random bit mask
            mask &= int(rng.integers(0, full, dtype=np.uint64, endpoint=True))
        a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
        if style == 3:                            # nearly sorted
            a = np.sort(a)
            for _ in range(int(rng.integers(0, 3))):
                i, j = rng.integers(0, n, size=2)
                a[i], a[j] = a[j], a[i]
        elif style == 4:                          # a tenth of the keys from a small pool
            pool = a[:64].copy()
            sel = rng.random(n) < 0.1
            a[sel] = pool[rng.integers(0, 64, size=int(sel.sum()))]
        elif style == 5:                          # strays in the constant columns
            bits = a.view(ol.NP_BITS[dt])
            for _ in range(int(rng.integers(1, 4))):
                bits[int(rng.integers(0, n))] ^= ol.NP_BITS[dt](~mask & full) & ol.NP_BITS[dt](int(rng.integers(0, full, dtype=np.uint64, endpoint=True)))
        order = int(rng.integers(0, 2))
        blind_on.setenv("RSX_NO_BLIND", "0")      # (no back-off between cases)
        info = check(a, dt, order, None, (k, n, dt, order, style, hex(mask)))
        taken += info.hybrid == 5
    assert taken >= cases // 6, taken


def test_blind_dense_slots_off_is_the_same_sort(blind_on):
    """4-byte keys without a histogram: the level-2 pass writes only the low half of the derived keys and the leaves put the
    rest back from the slot's digits; RSX_NO_DENSE_SLOTS=1 writes whole keys.  Both against the oracle, all key types of
    that width, both orders (the upper half of a float's or a signed key's derived form is not its raw upper half)."""
    for n in ((1 << 22) + 1234, 6000001):         # (slots of 256 keys: the library writes whole keys there unless told otherwise)
        for dt in (ol.U32, ol.I32, ol.F32):
            for order in (ol.ASC, ol.DESC):
                a = ol.splitmix_fill(n, dt, 500 + dt + order, 0xFFFFFFFF)
                blind_on.setenv("RSX_DENSE_SLOTS", "1")
                check(a, dt, order, 5, (dt, order, "two-byte slots"))
                blind_on.delenv("RSX_DENSE_SLOTS")
                blind_on.setenv("RSX_NO_DENSE_SLOTS", "1")
                check(a, dt, order, 5, (dt, order, "whole keys"))
                blind_on.delenv("RSX_NO_DENSE_SLOTS")


# ---- MSB digits below constant top bits (SegCtl::shift1 / shift2; 4-byte keys, two-byte slots) ---------------------------
@pytest.mark.parametrize("dt", [ol.U32, ol.I32, ol.F32], ids=["u32", "i32", "f32"])
def test_blind_digits_below_constant_top_bits(dt, blind_on):
    """4-byte keys whose top bits are the same in every key -- values below 2^30 / 2^27 / 2^25, or one rank's share of a
    distributed sort (top byte in [64, 128)): all four byte columns are kept (the reference's view, radix_sort.hpp:64-70), but
    the top byte takes few values and two passes by whole bytes would fill a few slots only.  The sample sees the constant
    bits, the passes go by the sixteen bits below the highest varying one, the level-1 pass checks the constant bits on every
    key; result, kept columns and returned buffer as the oracle's."""
    for n in ((1 << 22) + 321, 5555555):
        for mask, base in ((0x3FFFFFFF, 0), (0x07FFFFFF, 0), (0x01FFFFFF, 0), (0x3FFFFFFF, 0x40000000), (0x00FFFFFF, 0xA5000000)):
            for order in (ol.ASC, ol.DESC):
                a = (ol.splitmix_fill(n, dt, 900 + n % 31 + order, mask).view(np.uint32) | np.uint32(base)).view(ol.NP_BITS[dt])
                want_route = 5 if mask != 0x00FFFFFF else None      # (a constant top BYTE is a skipped column: three kept, another route)
                info = check(np.ascontiguousarray(a), dt, order, want_route, (n, hex(mask), hex(base), dt, order))
                if mask == 0x00FFFFFF:
                    assert info.hybrid != 5 and info.ncols == 3


def test_blind_constant_top_bits_disproved_by_one_key(blind_on):
    """One key with a bit set above what the sample saw: the level-1 pass finds it, the attempt is called off, the ordinary sort
    follows (and gives the oracle's result)."""
    n = (1 << 22) + 77
    for where in (1, n // 2, n - 1):
        a = ol.splitmix_fill(n, ol.U32, 950, 0x07FFFFFF).view(np.uint32).copy()
        a[where] |= np.uint32(0x20000000)
        blind_on.setenv("RSX_NO_BLIND", "0")
        info = check(a, ol.U32, ol.ASC, None, where)
        assert info.hybrid != 5, (where, info.hybrid)


def test_blind_unaligned_range_of_top_bytes_is_called_off(blind_on):
    """Top bytes 63 .. 127: the bits that vary reach bit 30, the digits below them fill 130 of 256 slots twice over -- the sample
    says no."""
    n = (1 << 22) + 5
    a = ol.splitmix_fill(n, ol.U32, 960, 0xFFFFFFFF).view(np.uint32)
    top = np.uint32(63) + (a >> 24) % np.uint32(65)
    a = (a & np.uint32(0x00FFFFFF)) | (top << 24)
    info = check(np.ascontiguousarray(a), ol.U32, ol.ASC, None, "63..127")
    assert info.hybrid != 5


def test_blind_no_shift_switch(blind_on):
    a = ol.splitmix_fill((1 << 22) + 9, ol.U32, 970, 0x07FFFFFF)
    assert check(a, ol.U32, ol.ASC, None, "shifted").hybrid == 5
    blind_on.setenv("RSX_NO_SHIFT", "1")
    assert check(a, ol.U32, ol.ASC, None, "RSX_NO_SHIFT=1").hybrid != 5
