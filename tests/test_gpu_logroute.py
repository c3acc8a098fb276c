"""8-byte keys by (bit length, mantissa) digits (radix_sorting_amd/csrc/rsx_logroute.hpp; rsx_info.hybrid == 6) against the oracle.

BASELINE.json's cfg 3 (iv) -- Zipf-like u64 keys, `key = 2^(b-1) + (r & (2^(b-1) - 1))`, `b = 1 + (r >> 58) % 40`, r = splitmix64
 -- is generated HERE on the host with the oracle's splitmix64 and numpy, sorted by the CPU restatement of
rs_sort_main (radix_sort.hpp:31-93) and compared with the device's result bit for bit, with the route asserted: whole arrays at
2^24 and 2^26 keys (2^28: tests/test_gpu_fullsize.py), smaller ones with the route's floor lowered (RSX_LOG_MIN_LOG2=20).  Around
it the cases the route must get right or must leave alone: odd lengths, constant bits above the varying ones, signed keys,
a pre-sorted array (the early exit, aux untouched), keys that cluster in their mantissa bits (a level-2 slot overflows: the
attempt is lost and the ordinary path sorts the untouched input), keys up to 2^44 (taken) and 2^45 (refused), descending order.
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

LOG_ROUTE = 6


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.fixture(autouse=True)
def _fresh_routes():
    rsa.reload_env()
    yield
    torch.cuda.empty_cache()


def zipf_like(n, seed, bmax=40):
    """SURVEY.md 8d cfg 3 (iv), integer-only: identical bits on host and device."""
    r = ol.splitmix_fill(n, ol.U64, seed)
    b = np.uint64(1) + ((r >> np.uint64(58)) % np.uint64(bmax))
    one = np.uint64(1)
    return (one << (b - one)) + (r & ((one << (b - one)) - one))


def _sort(a, dt, order=rsa.ASCENDING):
    src = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
    aux = torch.full_like(src, 0x5A5A5A5A5A5A5A5A)
    res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    torch.cuda.synchronize()
    return res, info, src, aux


def _check(a, dt, want_route, order=rsa.ASCENDING, what=""):
    want, want_aux, winfo = ol.oracle_sort(a, dt, ol.DESC if order == rsa.DESCENDING else ol.ASC)
    res, info, src, aux = _sort(a, dt, order)
    if want_route is not None:
        assert info.hybrid == want_route, (what, info.hybrid)
    assert info.result_in_aux == want_aux, what
    assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), what
    got = res.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want.view(np.uint64)), what
    return info


@pytest.mark.parametrize("log2n", [24, 25, 26])
def test_cfg3_zipf_like_against_the_oracle(log2n, monkeypatch):
    if log2n == 24:      # (16 Mi keys lie below the route's floor of 24 Mi keys, where one pass per column is as fast: lowered)
        monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    _check(zipf_like(1 << log2n, 33), rsa.U64, LOG_ROUTE, what="2^%d" % log2n)


def test_leaves_of_up_to_10240_values(monkeypatch):
    """Arrays beyond 2^28 + 2^24 keys take the leaves' larger shape (512 threads, 8192 bins); forced here at a size the oracle sorts in
    seconds, and at 2^29 keys -- where it is the shape the size selects -- checked by properties (tests/test_gpu_big.py style)."""
    monkeypatch.setenv("RSX_LOG_LEAF_BIG", "1")
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    _check(zipf_like(1 << 24, 35), rsa.U64, LOG_ROUTE, what="2^24, leaves of 10240")
    _check(zipf_like((1 << 22) + 3, 36, 44), rsa.U64, LOG_ROUTE, what="2^22 + 3, bmax 44, leaves of 10240")


def test_two_to_the_29_zipf_like_keys():
    n = 1 << 29
    r = torch.empty(n, dtype=torch.int64, device="cuda")
    rsa.fill_splitmix(r, seed=37)
    b = 1 + (((r >> 58) & 63) % 40)
    one = torch.ones_like(r)
    keys = (one << (b - 1)) + (r & ((one << (b - 1)) - 1))
    del r, b, one
    before = (int(keys.sum().item()), int((keys * (keys >> 7)).sum().item()))
    aux = torch.empty_like(keys)
    res, info = rsa.radix_sort(keys, aux, dtype=rsa.U64)
    torch.cuda.synchronize()
    assert info.hybrid == LOG_ROUTE and info.ncols == 5 and info.result_in_aux == 1
    assert bool((res[1:] >= res[:-1]).all().item())                      # (keys below 2^40: signed order is unsigned order)
    assert (int(res.sum().item()), int((res * (res >> 7)).sum().item())) == before


def test_the_floor():
    info = _check(zipf_like(1 << 24, 34), rsa.U64, None, what="2^24, default floor")
    assert info.hybrid != LOG_ROUTE
    _check(zipf_like(3 << 23, 34), rsa.U64, LOG_ROUTE, what="24 Mi, default floor")


@pytest.mark.parametrize("n", [(1 << 20), (1 << 20) + 1, (1 << 21) + 12345, (3 << 20) - 7])
def test_small_arrays_with_the_floor_lowered(n, monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    _check(zipf_like(n, 5 + n % 7), rsa.U64, LOG_ROUTE, what="n=%d" % n)


def test_constant_bits_above_the_varying_ones(monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    a = zipf_like(1 << 21, 8) | np.uint64(0xAB00000000000000)
    info = _check(a, rsa.U64, LOG_ROUTE, what="top byte 0xAB")
    assert info.ncols == 5
    # signed keys: positive values, the KDF's sign flip is a constant bit above them
    _check(zipf_like(1 << 21, 9), rsa.I64, LOG_ROUTE, what="i64")


def test_presorted_input_takes_the_early_exit(monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    monkeypatch.setenv("RSX_NO_BLIND", "1")      # (straight to this route's own histogram)
    a = np.sort(zipf_like(1 << 21, 10))
    res, info, src, aux = _sort(a, rsa.U64)
    assert info.early_exit == 2 and res is src and info.ncols == 0
    assert bool((aux == 0x5A5A5A5A5A5A5A5A).all().item())       # radix_sort.hpp:60-62: aux untouched
    assert np.array_equal(res.cpu().numpy().view(np.uint64), a)


def test_clustered_mantissas_lose_the_attempt_and_sort_anyway(monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    # the eight bits below the level-1 digit's are the same in every key of a bucket: one level-2 slot would take a whole bucket
    a = zipf_like(1 << 22, 11)
    b = np.uint64(64) - np.uint64(1) - np.floor(np.log2(a.astype(np.float64))).astype(np.uint64)   # leading zeros (exact below 2^53)
    blen = np.uint64(64) - b
    sh = np.where(blen > 12, blen - np.uint64(12), np.uint64(0)).astype(np.uint64)
    a = np.where(blen > 12, a & ~(np.uint64(0xFF) << sh), a).astype(np.uint64)
    info = _check(a, rsa.U64, None, what="clustered")
    assert info.hybrid != LOG_ROUTE


@pytest.mark.parametrize("bmax,taken", [(44, True), (45, False), (25, True), (24, False)])
def test_the_window_of_bit_lengths(bmax, taken, monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    info = _check(zipf_like(1 << 22, 12, bmax), rsa.U64, None, what="bmax=%d" % bmax)
    assert (info.hybrid == LOG_ROUTE) == taken, (bmax, info.hybrid)


def test_derived_keys_that_are_not_the_element_images(monkeypatch):
    """What spreads is the DERIVED key: complemented Zipf-like keys sorted descending, and negative doubles whose bit patterns,
    complemented by the KDF (radix_sort_basic_kdf.hpp:32-46), are Zipf-like -- every kernel of the route must derive before it cuts."""
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    z = zipf_like(1 << 21, 15)
    _check(~z, rsa.U64, LOG_ROUTE, order=rsa.DESCENDING, what="~zipf descending")
    neg = ~z                                     # sign bit set, exponent field all ones above the varying bits ... as raw bits of doubles:
    neg = (neg & np.uint64(0x800FFFFFFFFFFFFF)) | np.uint64(0x3FF0000000000000)   # finite negative doubles -(1.x): the KDF complements them
    _check(neg, rsa.F64, None, what="negative doubles")


def test_descending_and_switched_off(monkeypatch):
    monkeypatch.setenv("RSX_LOG_MIN_LOG2", "20")
    a = zipf_like(1 << 21, 13)
    info = _check(a, rsa.U64, None, order=rsa.DESCENDING, what="descending")
    assert info.hybrid != LOG_ROUTE               # (complemented keys: the magnitudes no longer spread)
    monkeypatch.setenv("RSX_NO_LOG", "1")
    info = _check(a, rsa.U64, None, what="RSX_NO_LOG")
    assert info.hybrid != LOG_ROUTE


def test_whole_result_verified_on_the_device(monkeypatch):
    """RSX_VERIFY=2 brackets any route with the checksum kernel: sortedness and the multiset of keys, on the device."""
    monkeypatch.setenv("RSX_VERIFY", "2")
    _check(zipf_like(1 << 25, 14), rsa.U64, LOG_ROUTE, what="RSX_VERIFY=2")
