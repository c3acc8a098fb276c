"""BASELINE.json's full-size configurations (2^28 elements) on one MI355X.

cfg 2 (2^28 u32) is in tests/test_gpu_parity.py with a full oracle comparison.  Here: cfg 3 (2^28 u64 with skipped
columns, uniform and Zipf-like) and cfg 4 (2^28 f32 keys + u32 payload = stable ranks).  At this size the checks are
the size-independent properties the domain offers -- sortedness by KDF, stability, permutation checksums, the
returned-buffer rule, idempotence -- computed on the GPU with torch; one u64 case is also compared with the oracle.
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N = 1 << 28
# the routes (rsx_info.hybrid) BASELINE.json's configurations take at this size: 5 = no histogram, two MSB passes into slots and
# leaves; 6 = 8-byte keys by (bit length, mantissa) digits -- the Zipf-like keys (rsx_logroute.hpp); DESIGN.md 4b / 4c / 4g
ZIPF_ROUTE = 6
RANK_ROUTE = {"random_bits": 5, "uniform_pm1": 5, "duplicate_heavy": 5}   # (duplicate_heavy: by its packed varying bits, SegCtl::compact; uniform_pm1: floats on a grid as fixed-point integers, SegCtl::ckind)
SIGN64 = -(1 << 63)
SIGN32 = -(1 << 31)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.fixture(autouse=True)
def _fresh_routes():
    """The library skips its next sorts without a histogram after an attempt that was called off (per context); a test that
    asserts the route (rsx_info.hybrid) must not depend on what ran before it: rsx_reload_env() forgets."""
    rsa.reload_env()
    yield


def _sorted_unsigned64(t):
    f = t ^ SIGN64                      # unsigned order == signed order with the top bit flipped
    return bool((f[1:] >= f[:-1]).all().item())


def _checksums(t):
    return int(t.sum().item()), int((t * (t >> 7)).sum().item())   # order-independent wrap-around sums


@pytest.mark.parametrize("mask,cols,in_aux", [(0xFFFFFFFFFFFFFFFF, 8, 0), (0x000000FFFFFFFFFF, 5, 1), (0x00000000FFFFFFFF, 4, 0)])
def test_cfg3_u64_column_skipping(mask, cols, in_aux):
    src = torch.empty(N, dtype=torch.int64, device="cuda")
    aux = torch.empty_like(src)
    rsa.fill_splitmix(src, seed=3, mask=mask)
    before = _checksums(src)
    res, info = rsa.radix_sort(src, aux, dtype=rsa.U64)
    torch.cuda.synchronize()
    assert info.ncols == cols and info.result_in_aux == in_aux          # SURVEY.md 8d cfg 3, appendix A item 5
    assert info.hybrid == 5, info.hybrid                                # no histogram, two MSB passes into slots, leaves
    assert _sorted_unsigned64(res)
    assert _checksums(res) == before
    if cols in (5, 8):
        # bit for bit against the oracle: P = 5 (four-byte level-2 slots, rsx_leafk_kernel's SLOT32 form) and P = 8 (whole-key slots,
        # the 5120-key leaves carried as u64: the shape only arrays above 2^27 keys select) -- seconds of host time on the GPU box
        a = ol.splitmix_fill(N, ol.U64, 3, mask)
        want, want_aux, _ = ol.oracle_sort(a, ol.U64)
        assert want_aux == in_aux
        assert np.array_equal(res.cpu().numpy().view(np.uint64), want)
    other = src if res is aux else aux
    res2, info2 = rsa.radix_sort(res, other, dtype=rsa.U64)              # idempotence: the pre-sorted early exit
    torch.cuda.synchronize()
    assert info2.early_exit == 2 and res2 is res


def test_cfg3_u64_zipf_like():
    """SURVEY.md 8d cfg 3 (iv): key = 2^(b-1) + low bits, b = 1 + (r >> 58) % 40 -- heavy duplicates, skewed high digits, P = 5.
    The keys are made on the HOST (the oracle's splitmix64 + numpy, integer-only) and the whole result is compared with the oracle's
    (rs_sort_main restated, radix_sort.hpp:31-93): bit for bit, the returned buffer and the kept columns included."""
    r = ol.splitmix_fill(N, ol.U64, 33)
    one = np.uint64(1)
    b = one + ((r >> np.uint64(58)) % np.uint64(40))
    a = (one << (b - one)) + (r & ((one << (b - one)) - one))
    del r, b
    keys = torch.from_numpy(a.view(np.int64)).cuda()
    before = _checksums(keys)
    aux = torch.empty_like(keys)
    res, info = rsa.radix_sort(keys, aux, dtype=rsa.U64)
    torch.cuda.synchronize()
    assert info.ncols == 5 and info.result_in_aux == 1
    assert info.hybrid == ZIPF_ROUTE, info.hybrid                       # (bit length, mantissa) digits: rsx_logroute.hpp
    assert _sorted_unsigned64(res) and _checksums(res) == before
    got = res.cpu().numpy().view(np.uint64)
    del keys, aux, res
    want, want_aux, winfo = ol.oracle_sort(a, ol.U64)
    assert want_aux == 1 and list(winfo.cols[:winfo.ncols]) == info.kept_columns()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", ["random_bits", "uniform_pm1", "duplicate_heavy"])
def test_cfg4_f32_keys_u32_ranks(variant):
    """radix_sort_rank on 2^28 float keys: output = stable ranks in the half the column parity dictates."""
    bits = torch.empty(N, dtype=torch.int32, device="cuda")
    if variant == "random_bits":          # NaNs, infinities, denormals
        rsa.fill_splitmix(bits, seed=6)
    elif variant == "duplicate_heavy":    # SURVEY.md 8d cfg 4 (iii): stability observable
        rsa.fill_splitmix(bits, seed=12, mask=0xFFF000FF)
    else:                                 # (ii): (int24 - 2^23) * 2^-23, exact on every platform
        r = torch.empty(N, dtype=torch.int64, device="cuda")
        rsa.fill_splitmix(r, seed=7)
        f = (((r >> 40) & 0xFFFFFF) - (1 << 23)).to(torch.float32) * (2.0 ** -23)
        bits = f.view(torch.int32).contiguous()
        del r, f
    ib = torch.full((2 * N,), -1, dtype=torch.int32, device="cuda")
    keep = bits.clone()
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert torch.equal(bits, keep)                                       # src is const (radix_sort_rank.hpp:97)
    assert info.hybrid == RANK_ROUTE[variant], (variant, info.hybrid)
    assert info.result_in_aux == (info.ncols & 1)                        # radix_sort_rank.hpp:88,:91
    if variant == "random_bits":
        # the whole rank array against the host's (oracle_lib.ranks_by_compound_sort, pinned against the C restatement of
        # rs_sort_rank in tests/test_oracle.py): the 5120-pair compound leaves at the only size that selects them
        want, want_aux = ol.ranks_by_compound_sort(ol.splitmix_fill(N, ol.F32, 6, 0xFFFFFFFF), ol.F32)
        assert want_aux == info.result_in_aux
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want)
        del want
    r64 = ranks.to(torch.int64)
    # a permutation: every index once
    seen = torch.zeros(N, dtype=torch.int8, device="cuda")
    seen[r64] = 1
    assert int(seen.sum().item()) == N
    del seen
    # keys in rank order are non-decreasing by KDF, and equal KDF keys keep input order (stability)
    g = bits[r64]
    kdf = torch.where(g < 0, ~g, g ^ SIGN32) ^ SIGN32                    # float KDF, then to signed-comparable
    ok_order = kdf[1:] >= kdf[:-1]
    assert bool(ok_order.all().item())
    same = kdf[1:] == kdf[:-1]
    assert bool((r64[1:][same] > r64[:-1][same]).all().item())
    if variant == "duplicate_heavy":
        assert int(same.sum().item()) > N // 2


def test_cfg4_pairs_random_bits_against_the_host():
    """Key + payload sort of 2^28 random bit patterns (route 5: rsx_leafp_kernel's 5120-pair shape writes keys AND payloads):
    payloads = indices must come out as the host's stable ranks, the keys as the input gathered through them."""
    a = ol.splitmix_fill(N, ol.F32, 6, 0xFFFFFFFF)
    want, _ = ol.ranks_by_compound_sort(a, ol.F32)
    keys = torch.from_numpy(a.view(np.int32)).cuda()
    vals = torch.arange(N, dtype=torch.int32, device="cuda")
    ka, va = torch.empty_like(keys), torch.empty_like(vals)
    kr, vr, info = rsa.radix_sort_pairs(keys, ka, vals, va, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == 5, info.hybrid
    assert np.array_equal(vr.cpu().numpy().view(np.uint32), want)
    assert np.array_equal(kr.cpu().numpy().view(np.uint32), a.view(np.uint32)[want])


def test_cfg4_pairs_f32_u32_payload():
    """The same configuration through the key+payload entry point (keys and payloads both move)."""
    keys = torch.empty(N, dtype=torch.int32, device="cuda")
    rsa.fill_splitmix(keys, seed=12, mask=0xFFF000FF)
    want_route = 0                                                       # (a column with two values: one pass per column)
    vals = torch.arange(N, dtype=torch.int32, device="cuda")
    ka, va = torch.empty_like(keys), torch.empty_like(vals)
    orig = keys.clone()
    kr, vr, info = rsa.radix_sort_pairs(keys, ka, vals, va, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == want_route, info.hybrid
    assert torch.equal(orig[vr.to(torch.int64)], kr)                     # payload still belongs to its key
    kdf = torch.where(kr < 0, ~kr, kr ^ SIGN32) ^ SIGN32
    assert bool((kdf[1:] >= kdf[:-1]).all().item())
    same = kdf[1:] == kdf[:-1]
    assert bool((vr[1:][same] > vr[:-1][same]).all().item())            # stable


@pytest.mark.parametrize("dt,tdt", [(rsa.U8, "uint8"), (rsa.I16, "int16")], ids=["u8", "i16"])
def test_counter_width_above_2p30(dt, tdt):
    """n >= 2^30 switches the chain's status words (and the digit offsets) to 64 bits, as radix_sort.hpp:102-114 widens its
    counters.  2^30 + 12345 narrow keys (1 and 2 kept columns): sortedness and the multiset (256 / 65536-bin counts)."""
    n = (1 << 30) + 12345
    tt = getattr(torch, tdt)
    src = torch.empty(n, dtype=tt, device="cuda")
    aux = torch.empty_like(src)
    rsa.fill_splitmix(src, seed=77)
    bins = 256 if dt == rsa.U8 else 65536
    as_index = (src.to(torch.int32) + (0 if dt == rsa.U8 else 32768))
    before = torch.bincount(as_index, minlength=bins)
    del as_index
    res, info = rsa.radix_sort(src, aux, dtype=dt)
    torch.cuda.synchronize()
    assert info.ncols == (1 if dt == rsa.U8 else 2) and info.result_in_aux == (1 if dt == rsa.U8 else 0)
    assert bool((res[1:] >= res[:-1]).all().item())
    as_index = (res.to(torch.int32) + (0 if dt == rsa.U8 else 32768))
    assert torch.equal(torch.bincount(as_index, minlength=bins), before)


def test_rank_above_2p30_narrowed_keys():
    """A rank sort beyond 2^30 elements (64-bit status words) whose keys narrow on the way (u16 -> u8): 2^30 + 12345 u16 keys
    -> u32 ranks.  Checked through the definition: keys[rank] is non-decreasing, equal keys keep index order, and the ranks'
    sum is that of 0 .. n-1."""
    n = (1 << 30) + 12345
    keys = torch.empty(n, dtype=torch.int16, device="cuda")
    rsa.fill_splitmix(keys, seed=99)
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(keys, ib, dtype=rsa.U16)
    torch.cuda.synchronize()
    assert info.ncols == 2 and info.result_in_aux == 0
    total = 0
    step = 1 << 28
    prev_key, prev_rank = None, None
    for o in range(0, n, step):
        r = ranks[o:o + step].to(torch.int64) & 0xFFFFFFFF
        total += int(r.sum().item())
        k = keys[r].to(torch.int32) & 0xFFFF
        assert bool((k[1:] >= k[:-1]).all().item())
        same = k[1:] == k[:-1]
        assert bool((r[1:][same] > r[:-1][same]).all().item())
        if prev_key is not None:
            assert int(k[0]) > prev_key or (int(k[0]) == prev_key and int(r[0]) > prev_rank)
        prev_key, prev_rank = int(k[-1]), int(r[-1])
        del r, k, same
    assert total == n * (n - 1) // 2
