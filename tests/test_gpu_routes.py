"""Every production route and leaf shape against the oracle AT THE SIZE THAT SELECTS IT, with the route asserted.

The differential tests of tests/test_gpu_hybrid.py lower the library's thresholds (RSX_TWO_LEVEL_MIN_LOG2=22) so that every
branch runs on arrays the oracle sorts in a second -- but a slot of 64 keys is not the geometry production runs in.  Here
nothing is lowered: the arrays are as large as the sizes at which the library itself picks each leaf shape (slots of 1280,
2048, 2560, 3840 and 5120 values; u64 leaves carried as u64 and as u32; pairs carried as 8-byte values), the result is
compared bit for bit with the CPU restatement of rs_sort_main / rs_sort_rank (radix_sort.hpp:31-93,
radix_sort_rank.hpp:22-92), and rsx_info.hybrid must name the route the size is meant to take (5: no histogram, two MSB
passes into slots, leaves -- DESIGN.md 4c) -- a sort that was silently diverted would pass a result-only test.

The oracle needs one to three seconds per case on one host core (rso_sort: 0.15 Gkeys/s).
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MI = 1 << 20
_CARRIER = {4: np.int32, 8: np.int64}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.fixture(autouse=True)
def _fresh_routes():
    rsa.reload_env()     # (no back-off from an earlier attempt that was called off: the route is asserted)
    yield
    torch.cuda.empty_cache()


def _sort_and_compare(a, dt, order, want_route, what):
    want, want_aux, winfo = ol.oracle_sort(a, dt, order)
    src = torch.from_numpy(np.ascontiguousarray(a).view(_CARRIER[a.itemsize]).copy()).cuda()
    aux = torch.full_like(src, 0x5A5A5A5A)
    res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    torch.cuda.synchronize()
    assert info.hybrid == want_route, (what, info.hybrid)
    assert info.result_in_aux == want_aux, what
    assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), what
    got = res.cpu().numpy().view(ol.NP_BITS[dt])
    assert np.array_equal(got, want), what
    return info


# n, what selects: slots of cap = 1.25 n / 65536 (small buckets: mean + 7 sigma) rounded up to 256 values; rsx_leaf16_kernel's 2560-value shape (eleven bin
# bits) up to 128 Mi keys, its 5120-value shape (twelve) above
@pytest.mark.parametrize("n_mi", [64, 96, 128, 192])
def test_u32_without_histogram_at_the_sizes_that_pick_each_leaf_shape(n_mi):
    n = n_mi * MI + 12345 * (n_mi % 3)
    a = ol.splitmix_fill(n, ol.U32, 4000 + n_mi, 0xFFFFFFFF)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, ("u32", n_mi))


@pytest.mark.parametrize("n,seed", [(10000000, 10), (40000000, 40), (48 * MI + 1, 48)])
def test_u32_mid_size_arrays_a_wave_per_leaf(n, seed):
    """BASELINE.json configs[0]'s sizes (radix_bench.cpp:135-138: 10^7 and 4 * 10^7 keys; the key file's stand-in is splitmix64
    seed 40) and the largest array whose slots one wave takes (1024 values): rsx_leaf16w_kernel."""
    a = ol.splitmix_fill(n, ol.U32, seed, 0xFFFFFFFF)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, ("u32", n))
    _sort_and_compare(a, ol.F32, ol.DESC, 5, ("f32 desc", n))


@pytest.mark.parametrize("dt,n_mi", [(ol.U32, 64), (ol.U64, 20)], ids=["u32", "u64"])
def test_level1_slots_all_in_scratch_memory(dt, n_mi, monkeypatch):
    """By default the level-1 slots that fit (n / cap1 of the 256) lie in the caller's second buffer -- every other test of this
    file runs that way --; RSX_NO_AUX_SLOTS=1 keeps them all in the library's scratch array."""
    monkeypatch.setenv("RSX_NO_AUX_SLOTS", "1")
    n = n_mi * MI + 17
    a = ol.splitmix_fill(n, dt, 4150 + n_mi, (1 << (8 * ol.DTYPE_SIZE[dt])) - 1)
    _sort_and_compare(a, dt, ol.ASC, 5, ("all slots in scratch", n_mi))


@pytest.mark.parametrize("digit", [0x05, 0xF3])
def test_level1_slot_overflows_after_the_second_buffer_was_written(digit):
    """One top digit with 1.4 times its share: too little for the sample (8192 keys) to notice, too much for the digit's level-1
    slot (1.25 times the mean).  The attempt is called off AFTER its level-1 pass has written -- with the slots in the caller's
    second buffer (digit 0x05) the lost run goes over the slot's own beginning, in the scratch array (0xF3) behind the last slot --
    and the histogram-first sort then uses that buffer as the reference does (radix_sort.hpp:82-92)."""
    n = 16 * MI + 3
    a = ol.splitmix_fill(n, ol.U32, 4160 + digit, 0xFFFFFFFF).view(np.uint32).copy()
    idx = np.arange(1000, 1000 + 26000 * 7, 7)
    a[idx] = (a[idx] & np.uint32(0x00FFFFFF)) | np.uint32(digit << 24)
    _sort_and_compare(a, ol.U32, ol.ASC, 0, ("level-1 overflow", hex(digit)))


@pytest.mark.parametrize("case", ["clustered low bits", "a wave per leaf", "ragged leaves"])
def test_u32_1e7_a_row_of_sixteen_lanes_per_leaf(case, monkeypatch):
    """10^7 keys (radix_bench.cpp:135-138's smaller size): slots of up to 256 values, four leaves per wave (rsx_leaf16q_kernel).
    Low sixteen bits from 64 x 16 values everywhere and 16 values in some buckets (more rounds, as many as the wave's neediest
    row wants); RSX_NO_LEAF16Q=1 (rsx_leaf16w_kernel as before); leaves of different sizes next to each other in a wave."""
    n = 10000000
    a = ol.splitmix_fill(n, ol.U32, 43, 0xFFFFFFFF).view(np.uint32).copy()
    if case == "clustered low bits":
        a &= np.uint32(0xFFFFFC0F)
        a[((a >> np.uint32(16)) % np.uint32(97)) == 5] &= np.uint32(0xFFFF000F)
    elif case == "a wave per leaf":
        monkeypatch.setenv("RSX_NO_LEAF16Q", "1")
    else:
        # a fifth of every fourth (digit, digit) bucket moved into its neighbour (a slot holds 1.25 x the mean): leaves of 0.8 and
        # 1.2 x the mean side by side in a wave
        d2 = (a >> np.uint32(16)) & np.uint32(3)
        a[(d2 == 1) & ((a & np.uint32(0xF)) < np.uint32(3))] ^= np.uint32(0x00010000)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, ("1e7", case))
    _sort_and_compare(a, ol.I32, ol.DESC, 5, ("1e7 i32 desc", case))


@pytest.mark.parametrize("n", [13000000, 13369344, 26738688, 45600000])
def test_slots_hold_an_even_array_whatever_its_size(n):
    """A slot's capacity is 1.25 times the mean bucket AND at least seven standard deviations above it (rsx.hip, slot_cap_for).
    With 1.25 x alone these sizes -- mean buckets of 198, 204 and 408 keys in slots of 256 / 512 -- lost nearly every attempt
    (13 Mi keys: 147 of 150) to one overflowing slot of the 65536 and fell back to one pass per column.  45.6 M keys: the largest
    array whose 1024-value slots are filled by rsx_pass16a_kernel (128 of their places kept for the slot's back, round 5)."""
    src = torch.empty(n, dtype=torch.int32, device="cuda")
    aux = torch.empty_like(src)
    for seed in range(12):
        rsa.fill_splitmix(src, 8800 + seed)
        rsa.reload_env()
        res, info = rsa.radix_sort(src, aux, rsa.U32)
        assert info.hybrid == 5, (n, seed, info.hybrid)
    got = res.cpu().numpy().view(np.uint32)
    assert np.all(got[1:] >= got[:-1])


def test_u32_mid_size_low_bits_clustered():
    """... with keys whose low sixteen bits take 64 x 16 values everywhere, and only 16 values in some buckets: the wave kernel
    has no list to hand a leaf to -- it goes on (more rounds of register passes) until the leaf is in order."""
    n = 40000000
    a = ol.splitmix_fill(n, ol.U32, 41, 0xFFFFFC0F).view(np.uint32).copy()
    _sort_and_compare(a, ol.U32, ol.ASC, 5, "low bits & 0xFC0F")
    b = ol.splitmix_fill(n, ol.U32, 42, 0xFFFFFFFF).view(np.uint32).copy()
    few = ((b >> 16) % 389) == 7
    b[few] &= np.uint32(0xFFFF000F)
    _sort_and_compare(b, ol.U32, ol.ASC, 5, "16 values in some buckets")


@pytest.mark.parametrize("case", ["clustered low bits", "16 values in some buckets", "a workgroup per leaf"])
def test_u32_a_wave_per_leaf_up_to_2048_values(case, monkeypatch):
    """Slots of 1025 .. 2048 values (52 .. 100 Mi keys) are one wave's too (rsx_leaf16w_kernel, two chunks of sixteen values per
    lane; the slots are rsx_pass16a_kernel's: read from both ends).  Low sixteen bits from 1024 values everywhere, from 16 values
    in some buckets (the wave goes on until its leaf is in order); RSX_NO_LEAF16W2K=1: the 128-thread workgroup shape as before."""
    n = 72 * MI + 333
    if case == "clustered low bits":
        a = ol.splitmix_fill(n, ol.U32, 5300, 0xFFFFFC0F)
    elif case == "16 values in some buckets":
        a = ol.splitmix_fill(n, ol.U32, 5301, 0xFFFFFFFF).view(np.uint32).copy()
        a[((a >> 16) % 389) == 7] &= np.uint32(0xFFFF000F)
    else:
        monkeypatch.setenv("RSX_NO_LEAF16W2K", "1")
        a = ol.splitmix_fill(n, ol.U32, 5302, 0xFFFFFFFF)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, case)
    _sort_and_compare(a, ol.F32, ol.DESC, 5, (case, "f32 desc"))


@pytest.mark.parametrize("n_mi,mask,base", [(128, 0x07FFFFFF, 0x18000000), (40, 0x3FFFFFFF, 0), (200, 0x1FFFFFFF, 0xE0000000)])
def test_u32_constant_top_bits_at_production_sizes(n_mi, mask, base):
    """One rank's sub-range of a distributed sort (2^27 keys whose top byte lies in [24, 32)), values below 2^30, keys with
    their top three bits set: the MSB digits lie below the constant bits (SegCtl::shift1 / shift2), route 5, bit for bit."""
    n = n_mi * MI + 31
    a = ol.splitmix_fill(n, ol.U32, 4700 + n_mi, mask).view(np.uint32) | np.uint32(base)
    _sort_and_compare(np.ascontiguousarray(a), ol.U32, ol.ASC, 5, ("u32", n_mi, hex(mask), hex(base)))
    _sort_and_compare(np.ascontiguousarray(a), ol.U32, ol.DESC, 5, ("u32 desc", n_mi, hex(mask), hex(base)))


@pytest.mark.parametrize("switch", ["RSX_NO_LEAF16", "RSX_NO_DENSE_SLOTS"])
@pytest.mark.parametrize("n_mi", [96, 192])     # (fallback kernels since round 4: two of their four cut shapes are enough here)
def test_u32_round3_leaves_at_the_same_sizes(n_mi, switch, monkeypatch):
    """The leaves of round 3 stay in the library (rsx_leaf_sort_kernel works off what rsx_leaf16_kernel leaves alone, and takes
    everything when the sample finds the low sixteen bits clustered): its cut shapes (2048, 3072, 5120, 8 Ki keys) on two-byte
    slots (RSX_NO_LEAF16=1: 3073 .. 5120-value slots only, as in round 3) and on whole keys (RSX_NO_DENSE_SLOTS=1)."""
    monkeypatch.setenv(switch, "1")
    n = n_mi * MI + 777
    a = ol.splitmix_fill(n, ol.I32, 4100 + n_mi, 0xFFFFFFFF)
    _sort_and_compare(a, ol.I32, ol.DESC, 5, ("i32 desc", n_mi, switch))


def test_u32_every_leaf_through_the_list(monkeypatch):
    """RSX_LEAF16_MAXBIN=0: rsx_leaf16_kernel puts every leaf on its list and the list launch sorts them all."""
    monkeypatch.setenv("RSX_LEAF16_MAXBIN", "0")
    n = 160 * MI + 5
    a = ol.splitmix_fill(n, ol.F32, 4200, 0xFFFFFFFF)
    _sort_and_compare(a, ol.F32, ol.ASC, 5, "f32, every leaf listed")


def test_u32_low_bits_clustered_in_some_buckets_only():
    """Keys whose low sixteen bits take few values, but only in some (digit, digit) buckets -- the sample (8192 keys of the
    whole array) sees nothing of it: those leaves' bins hold 10 .. 25 keys (two more register passes) or more (the list)."""
    n = 160 * MI
    a = ol.splitmix_fill(n, ol.U32, 4300, 0xFFFFFFFF).view(np.uint32).copy()
    top = a >> 16
    some = (top % 97) == 5                     # one bucket in 97: low bits from 256 values -> bins of ~10 keys of 2560
    a[some] &= np.uint32(0xFFFF0FF0)
    few = (top % 389) == 7                     # one in 389: 16 values -> bins of 160 keys
    a[few] &= np.uint32(0xFFFF000F)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, "locally clustered low bits")


def test_u32_low_bits_clustered_everywhere_keeps_the_route_and_changes_the_leaves():
    """The same clustering in every bucket: every column still has 16 or more values and no hot digit, so the sort still goes
    without a histogram -- and the sample's count over rsx_leaf16_kernel's bins (SegCtl::leaf16) hands every leaf to
    rsx_leaf_sort_kernel."""
    n = 160 * MI + 3
    a = ol.splitmix_fill(n, ol.U32, 4400, 0xFFFFFC0F)
    _sort_and_compare(a, ol.U32, ol.ASC, 5, "low bits & 0xFC0F")


@pytest.mark.parametrize("force", ["1", "2", "3", "4", "5", "6"], ids=["counting", "10240-value shape + counting", "20480-value shape + counting",
                                                                     "6144", "7680", "15360"])
@pytest.mark.parametrize("case", ["uniform", "constant top bits, descending", "low bits clustered in some buckets", "low bits clustered everywhere",
                                  "few values"])
def test_u32_leaves_of_the_large_slots(case, force, monkeypatch):
    """Arrays beyond 2^28 keys leave two-byte slots of more than 5120 values: rsx_leaf16_kernel in its 10240- / 20480-value shapes
    with the counting leaves (rsx_leafc_kernel) behind them, the counting leaves alone for the slots of 2^31 keys
    (csrc/rsx_leafc.hpp; the reference's last two passes, radix_sort.hpp:82-90).  Those sizes are beyond the oracle
    (tests/test_gpu_big.py checks them by properties), so RSX_FORCE_LEAFC sends the slots of 160 Mi keys (3328 values) through
    the same launches: bit for bit, route asserted.  Clustered low bits: the larger shapes leave such leaves to the
    counting kernel's list launch -- or all of them, when the sample sees the clustering."""
    if force in ("4", "5", "6") and case not in ("uniform", "low bits clustered in some buckets"):
        pytest.skip("the same kernel in another shape: two cases each")
    monkeypatch.setenv("RSX_FORCE_LEAFC", force)
    n = 160 * MI + 77
    dt, order = ol.U32, ol.ASC
    if case == "uniform":
        a = ol.splitmix_fill(n, ol.F32, 5100, 0xFFFFFFFF)
        dt = ol.F32
    elif case == "constant top bits, descending":
        a = np.ascontiguousarray(ol.splitmix_fill(n, ol.U32, 5101, 0x1FFFFFFF).view(np.uint32) | np.uint32(0xE0000000))
        order = ol.DESC
    elif case == "low bits clustered in some buckets":
        a = ol.splitmix_fill(n, ol.U32, 5102, 0xFFFFFFFF).view(np.uint32).copy()
        top = a >> 16
        a[(top % 97) == 5] &= np.uint32(0xFFFF0FF0)
        a[(top % 389) == 7] &= np.uint32(0xFFFF000F)
        a[(top % 1009) == 11] &= np.uint32(0xFFFF0000)      # every value of the leaf the same
    elif case == "low bits clustered everywhere":
        a = ol.splitmix_fill(n, ol.U32, 5103, 0xFFFFFC0F)
    else:
        a = ol.splitmix_fill(n, ol.I32, 5104, 0xFFFFF00F)   # 256 values of the low sixteen bits: thirteen of each in every leaf
        dt = ol.I32
    _sort_and_compare(a, dt, order, 5, (case, force))


@pytest.mark.parametrize("n_mi,mask", [(5, 0xFFFFFFFFFFFFFFFF), (6, 0xFFFFFFFFFF), (9, 0xFFFFFFFFFFFFFFFF), (20, 0xFFFFFFFFFFFFFFFF), (48, 0xFFFFFFFFFFFFFFFF),
                                       (96, 0xFFFFFFFFFFFFFFFF), (160, 0xFFFFFFFFFFFFFFFF), (12, 0xFFFFFFFFFF), (48, 0xFFFFFFFFFF), (96, 0xFFFFFFFFFF),
                                       (24, 0xFFFFFFFF), (96, 0xFFFFFFFF)])
def test_u64_without_histogram(n_mi, mask):
    """8-byte keys from 4.5 Mi keys on (leaves in four shapes by the slots' capacity: a wave per leaf for up to 256 keys -- arrays
    up to 13 Mi --, up to 1280 keys with 1024 bins -- up to 64 Mi --, 2560 with 2048, 5120 with 4096): uniform (six columns per leaf: the top three + odd-even transposition, carried as u64)
    and with constant top bytes (leaves carried as u32; the constant columns checked on every key by the level-1 pass)."""
    n = n_mi * MI + 4242
    a = ol.splitmix_fill(n, ol.U64, 4500 + n_mi, mask)
    _sort_and_compare(a, ol.U64, ol.ASC, 5, ("u64", n_mi, hex(mask)))


@pytest.mark.parametrize("switch", [None, "RSX_NO_PASS32A"], ids=["atoms of eight keys", "the chained level-1 pass"])
def test_u64_level1_pass_in_whole_atoms(switch, monkeypatch):
    """8-byte keys from 48 Mi keys on: the level-1 pass writes whole 64-byte atoms (rsx_pass32a_kernel<u64>: a bucket lies at both
    ends of its slot, the chained level-2 pass reads a tile more per bucket); RSX_NO_PASS32A=1: round 4's chained pass.  Signed
    keys descending (the generic digit) and masked keys whose result ends in the second buffer (five kept columns)."""
    if switch:
        monkeypatch.setenv(switch, "1")
    n = 56 * MI + 4099
    a = ol.splitmix_fill(n, ol.I64, 5200, 0xFFFFFFFFFFFFFFFF)
    _sort_and_compare(a, ol.I64, ol.DESC, 5, ("i64 desc", switch))
    b = ol.splitmix_fill(n, ol.U64, 5201, 0xFFFFFFFFFF)
    info = _sort_and_compare(b, ol.U64, ol.ASC, 5, ("u64 & 0xFFFFFFFFFF", switch))
    assert info.result_in_aux == 1
    if switch is None:
        # every level-1 slot in scratch memory (none in the caller's second buffer); doubles of both signs, ascending
        monkeypatch.setenv("RSX_NO_AUX_SLOTS", "1")
        rsa.reload_env()
        c = ol.splitmix_fill(48 * MI + 1, ol.F64, 5202, 0xFFFFFFFFFFFFFFFF)     # (random bit patterns: NaNs and both zeros among them)
        _sort_and_compare(c, ol.F64, ol.ASC, 5, "f64, slots in scratch")


def test_u64_device_scheduled_with_the_level1_pass_in_atoms():
    """rsx_sort_inplace_async on 8-byte keys at a size whose level-1 pass writes atoms: the tile table and the chained level-2
    pass's status words have a row more per bucket (two-ended slots), also inside a workspace sized by rsx_workspace_bytes_fast."""
    n = 64 * MI + 17
    a = ol.splitmix_fill(n, ol.U64, 5203, 0xFFFFFFFFFFFFFFFF)
    want, _, _ = ol.oracle_sort(a, ol.U64, ol.ASC)
    buf = torch.from_numpy(a.view(np.int64).copy()).cuda()
    scratch = torch.empty_like(buf)
    rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U64)
    assert rsa.async_route() == 5
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), want)
    ws = torch.empty(rsa.workspace_bytes_fast(n, rsa.U64), dtype=torch.uint8, device="cuda")
    buf.copy_(torch.from_numpy(a.view(np.int64)))
    rsa.radix_sort_inplace_async_ws(buf, scratch, ws, dtype=rsa.U64)
    assert rsa.async_route_ws(ws, n, rsa.U64) == 5
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("n_mi,maxbin", [(7, "0"), (20, "0"), (80, "0"), (7, None), (20, None)])
def test_u64_small_leaves_lists_and_fat_bins(n_mi, maxbin, monkeypatch):
    """The smaller shapes of rsx_leafk_kernel (1280 keys / 1024 bins, 2560 / 2048): every leaf through the list launch
    (RSX_LEAF16_MAXBIN=0), and bins of 10 .. 25 keys (two more register passes) or more (the list) in some buckets only."""
    n = n_mi * MI + 31
    a = ol.splitmix_fill(n, ol.U64, 4700 + n_mi, 0xFFFFFFFFFFFFFFFF).view(np.uint64).copy()
    if maxbin is not None:
        monkeypatch.setenv("RSX_LEAF16_MAXBIN", maxbin)
    else:
        top = a >> np.uint64(48)
        a[(top % np.uint64(53)) == 3] &= np.uint64(0xFFFF0FFFFFFFFFFF)     # 16 values in the leaf's top byte: ~20 keys in the fullest bins
        a[(top % np.uint64(211)) == 9] &= np.uint64(0xFFFF000FFFFFFFFF)    # the bin bits from 16 values: bins of ~20 of 320 keys
    _sort_and_compare(a, ol.U64, ol.ASC, 5, ("u64 small leaves", n_mi, maxbin))


@pytest.mark.parametrize("case", ["u64 asc", "i64 desc", "f64 asc", "f64 negative desc", "whole-key slots", "every leaf by the network",
                                  "fat bins in some buckets", "chained level-2 pass"])
def test_u64_four_byte_slots(case, monkeypatch):
    """8-byte keys whose leaves sort columns of the low word only (here: 40 varying bits below constant ones): the level-2 pass
    writes the low word of every derived key (SegCtl::narrow) and rsx_leafk_kernel's SLOT32 form puts the upper word back from
    the first key and the slot's digits -- for every KDF (the derived upper word is constant whatever the type's flips do).
    RSX_NO_NARROW_SLOTS=1: whole keys as before.  That form has no list: leaves with bins too full for the register passes go
    through Batcher's network over the whole leaf (RSX_LEAF16_MAXBIN=0: every leaf; clustering in some buckets: those)."""
    n = 20 * MI + 123
    r = ol.splitmix_fill(n, ol.U64, 4900, 0xFFFFFFFFFF).view(np.uint64).copy()
    dt, order = ol.U64, ol.ASC
    if case == "i64 desc":
        dt, order = ol.I64, ol.DESC
    elif case == "f64 asc":
        dt = ol.F64
        r |= np.uint64(0x3FF0000000000000)          # doubles in [1, 1 + 2^-12)
    elif case == "f64 negative desc":
        dt, order = ol.F64, ol.DESC
        r |= np.uint64(0xBFF0000000000000)
    elif case == "whole-key slots":
        monkeypatch.setenv("RSX_NO_NARROW_SLOTS", "1")
    elif case == "every leaf by the network":
        monkeypatch.setenv("RSX_LEAF16_MAXBIN", "0")
    elif case == "chained level-2 pass":          # (round 5's pass instead of rsx_pass64a_kernel: dense slots, no back)
        monkeypatch.setenv("RSX_NO_PASS64A", "1")
    elif case == "fat bins in some buckets":
        top = (r >> np.uint64(24)) & np.uint64(0xFFFF)
        r[(top % np.uint64(53)) == 3] &= np.uint64(0xFFFFFFFFFF0F0FFF)     # the leaf's keys in 16 bins of ~20: two more register passes
        r[(top % np.uint64(211)) == 9] &= np.uint64(0xFFFFFFFFFF000FFF)    # ... in ONE bin: the network
    _sort_and_compare(r, dt, order, 5, ("u64 four-byte slots", case))


@pytest.mark.parametrize("n_mi,mask", [(13, 0xFFFFFFFFFF), (48, 0xFFFFFFFF), (64, 0xFFFFFFFF), (80, 0xFFFFFFFFFF), (200, 0xFFFFFFFFFF)])
def test_u64_level2_pass_in_whole_atoms(n_mi, mask):
    """rsx_pass64a_kernel (rsx_pass64.hpp) at sizes that select the SLOT32 leaves' shapes (slots of 512 .. 5120 four-byte values read
    from both ends), at 64 Mi keys -- where the 128 places of a slot's back would not leave its front mean + 6 sigma and the chained
    pass stays -- and with an odd count: whole arrays against the oracle, route asserted."""
    a = ol.splitmix_fill(n_mi * MI + 77, ol.U64, 5300 + n_mi, mask)
    _sort_and_compare(a, ol.U64, ol.ASC, 5, ("u64 level-2 atoms", n_mi, hex(mask)))


@pytest.mark.parametrize("case", ["u64 & 0xFFFFFFFFFF", "u64 & 0xFFFFFFFF", "i64 descending", "f64 with a constant exponent", "constant top bytes",
                                  "48 varying bits", "switched off", "one key varies above the level-1 digit"])
def test_u64_level1_slots_of_low_words(case, monkeypatch):
    """8-byte keys in which nothing below the level-1 digit varies above bit 32 (SegCtl::narrow == 2: BASELINE.json's cfg 3 (ii) and
    (iii)): rsx_pass32a_kernel<u64, ..., OT = u32> writes the low word of every derived key into four-byte level-1 slots in the
    caller's second buffer, rsx_pass64a_kernel<u32, u32> reads them: 12 instead of 16 and 8 instead of 12 bytes per key, which the
    library's own profile shows.  Keys that vary in 48 bits keep whole keys at level 1 (narrow == 1); a key that differs from the
    first one in a column the sample took for constant calls the attempt off after the narrow form has written."""
    n = 40 * MI + 33
    dt, order, route = ol.U64, ol.ASC, 5
    l1_bytes, l2_bytes = 12, 8
    if case == "u64 & 0xFFFFFFFFFF":
        a = ol.splitmix_fill(n, ol.U64, 5401, 0xFFFFFFFFFF)
    elif case == "u64 & 0xFFFFFFFF":
        a = ol.splitmix_fill(n, ol.U64, 5402, 0xFFFFFFFF)
    elif case == "i64 descending":
        a = ol.splitmix_fill(n, ol.I64, 5403, 0xFFFFFFFFFF)
        dt, order = ol.I64, ol.DESC
    elif case == "f64 with a constant exponent":
        a = ol.splitmix_fill(n, ol.U64, 5404, 0xFFFFFFFFFF) | np.uint64(0x3FF0000000000000)     # (bit patterns of doubles in [1, 1 + 2^-12))
        dt = ol.F64
    elif case == "constant top bytes":
        a = ol.splitmix_fill(n, ol.U64, 5405, 0xFFFFFFFFFF) | np.uint64(0xAB00CD0000000000)
    elif case == "48 varying bits":
        a = ol.splitmix_fill(n, ol.U64, 5406, 0xFFFFFFFFFFFF)
        l1_bytes, l2_bytes = 16, 12
    elif case == "switched off":
        monkeypatch.setenv("RSX_NO_NARROW_LEVEL1", "1")
        rsa.reload_env()
        a = ol.splitmix_fill(n, ol.U64, 5407, 0xFFFFFFFFFF)
        l1_bytes, l2_bytes = 16, 12
    else:
        a = ol.splitmix_fill(n, ol.U64, 5408, 0xFFFFFFFFFF).view(np.uint64).copy()
        a[n // 3] |= np.uint64(1 << 52)
        route = None
    want, want_aux, winfo = ol.oracle_sort(a, dt, order)
    src = torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
    aux = torch.full_like(src, 0x5A5A5A5A)
    rsa.profile_begin()
    res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
    torch.cuda.synchronize()
    prof = rsa.profile_end()
    if route is None:
        assert info.hybrid != 5, (case, info.hybrid)
    else:
        assert info.hybrid == 5, (case, info.hybrid)
        assert prof.scatter_bytes == n * l1_bytes and prof.narrow_bytes == n * l2_bytes and prof.leaf_bytes == n * 12, \
            (case, prof.scatter_bytes / n, prof.narrow_bytes / n, prof.leaf_bytes / n)
    assert info.result_in_aux == want_aux and info.kept_columns() == list(winfo.cols[:winfo.ncols]), case
    assert np.array_equal(res.cpu().numpy().view(np.uint64), np.ascontiguousarray(want).view(np.uint64)), case


def _ranks_and_pairs(a, want_route, what):
    """A rank sort and a key + payload sort of the f32 keys `a` against the oracle's ranks; want_route None: any but 5."""
    n = len(a)
    # up to 16 Mi keys the C restatement of rs_sort_rank with Listing 6's loop; above, the same ranks as sorted (key, index)
    # compounds (oracle_lib.want_ranks: pinned against each other in tests/test_oracle.py)
    want, want_aux = ol.want_ranks(a, ol.F32)
    bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert (info.hybrid == 5) if want_route == 5 else (info.hybrid != 5), (what, info.hybrid)
    assert info.result_in_aux == want_aux == (info.ncols & 1), what
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), what
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), a.view(np.uint32)), what      # (a rank sort leaves its keys alone)
    del ib, ranks
    rsa.reload_env()
    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ka, va = torch.empty_like(bits), torch.empty_like(vals)
    kr, vr, info = rsa.radix_sort_pairs(bits, ka, vals, va, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert (info.hybrid == 5) if want_route == 5 else (info.hybrid != 5), (what, info.hybrid)
    assert np.array_equal(vr.cpu().numpy().view(np.uint32), want), what
    assert np.array_equal(kr.cpu().numpy().view(np.uint32), a.view(np.uint32)[want]), what


@pytest.mark.parametrize("n_mi", [5, 9, 16, 64, 160])
def test_f32_ranks_and_pairs_without_histogram(n_mi):
    """Rank sorts and key + payload sorts of 4-byte keys take the route from 4 Mi pairs on; the leaves (rsx_leafp_kernel) come
    in five shapes chosen by the slots' capacity: a wave per leaf for up to 256 / 512 pairs (.. 27 Mi pairs), 1280 pairs and 1024 bins
    (.. 64 Mi), 2560 and 2048 (.. 2^27), 5120 and 4096 (160 Mi here; 2^28: tests/test_gpu_fullsize.py).  From 6.4 Mi pairs on (a
    level-1 slot then holds a tile) the level-1 slots that fit lie in the caller's spare buffers: the second key / payload buffers
    of a key + payload sort; of a rank sort the index buffer's first half for the indices and its second half for the keys."""
    n = n_mi * MI + 99
    _ranks_and_pairs(ol.splitmix_fill(n, ol.F32, 4600 + n_mi, 0xFFFFFFFF), 5, ("ranks and pairs", n_mi))


def test_f32_ranks_and_pairs_level1_slots_all_in_scratch_memory(monkeypatch):
    monkeypatch.setenv("RSX_NO_AUX_SLOTS", "1")
    n = 24 * MI + 5
    _ranks_and_pairs(ol.splitmix_fill(n, ol.F32, 4651, 0xFFFFFFFF), 5, "all four slot arrays in scratch")


@pytest.mark.parametrize("case", ["random bits", "every key twice", "low byte from 16 values", "packed varying bits"])
def test_f32_ranks_and_pairs_in_the_leaves_for_10240_pairs(case, monkeypatch):
    """Beyond 2^28 pairs a slot holds up to 10240: rsx_leafp_kernel's LeafKCfg<1024, 10240, 4, 13> (fourteen position bits in the
    compound) and, for leaves with fat bins, rsx_leaf_pairs_kernel's 12288-pair shape.  Forced here at sizes the oracle sorts in
    seconds (RSX_PAIRS_LEAF_BIG=1); at 2^29 pairs -- where the size selects them -- checked by properties below."""
    monkeypatch.setenv("RSX_PAIRS_LEAF_BIG", "1")
    rsa.reload_env()
    n = 12 * MI + 1234
    a = ol.splitmix_fill(n, ol.F32, 4680, 0xFFFFFFFF).view(np.uint32).copy()
    if case == "every key twice":
        a[1::2] = a[:-1:2][: len(a[1::2])]
    elif case == "low byte from 16 values":
        a &= np.uint32(0xFFFFFF0F)
    elif case == "packed varying bits":
        # (rank sorts only: key + payload sorts hand the keys back and do not pack them)
        a &= np.uint32(0xFFF000FF)
        want, want_aux = ol.want_ranks(a, ol.F32)
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks, info = rsa.radix_sort_rank(torch.from_numpy(a.view(np.int32).copy()).cuda(), ib, dtype=rsa.F32)
        torch.cuda.synchronize()
        assert info.hybrid == 5 and info.result_in_aux == want_aux, info.hybrid
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want)
        return
    _ranks_and_pairs(a, 5, ("leaves of 10240 pairs", case))


@pytest.mark.parametrize("n", [(1 << 29), (3 << 27) + 4099])
def test_f32_ranks_and_pairs_beyond_two_to_the_28(n):
    """2^29 and 1.5 x 2^28 (f32, u32) pairs and rank sorts on the route without a histogram (round 5: one pass per column beyond 2^28):
    order by the derived key, stability (payloads = indices ascend among equal keys), every payload still with its key, nothing lost."""
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * (1 << 30):
        pytest.skip("needs 24 GiB of free HBM")
    SIGN = -(1 << 31)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    rsa.fill_splitmix(keys, seed=4690)
    keys &= ~0x0101                                                      # (2^30 distinct keys: a third of the pairs share theirs with another)
    orig = keys.clone()

    def check(kr, vr):
        kdf = torch.where(kr < 0, ~kr, kr ^ SIGN)                        # (radix_sort_basic_kdf.hpp:32-46 as signed order: ^ SIGN twice)
        kdf = kdf ^ SIGN
        assert bool((kdf[1:] >= kdf[:-1]).all().item())
        same = kdf[1:] == kdf[:-1]
        assert bool((vr[1:][same] > vr[:-1][same]).all().item())         # stable
        del kdf, same
        assert torch.equal(orig[vr.to(torch.int64)], kr)                 # every payload with its key
        assert int(vr.to(torch.int64).sum().item()) == n * (n - 1) // 2  # ... and every index once (with the order above: a permutation)

    vals = torch.arange(n, dtype=torch.int32, device="cuda")
    ka, va = torch.empty_like(keys), torch.empty_like(vals)
    kr, vr, info = rsa.radix_sort_pairs(keys, ka, vals, va, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == 5, info.hybrid
    check(kr, vr)
    del vals, ka, va, kr, vr
    torch.cuda.empty_cache()
    keys.copy_(orig)
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(keys, ib, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == 5 and info.result_in_aux == 0, info.hybrid
    assert torch.equal(keys, orig)                                       # (a rank sort leaves its keys alone)
    check(orig[ranks.to(torch.int64)], ranks)


@pytest.mark.parametrize("digit", [0x05, 0xF3])
def test_f32_pairs_level1_slot_overflows_after_the_spare_buffers_were_written(digit):
    """test_level1_slot_overflows_after_the_second_buffer_was_written for (key, payload) and (key, index) compounds: one top digit
    of the DERIVED key with 1.4 times its share -- in a slot that lies in the spare buffers (0x05) or in scratch (0xF3).  The
    attempt is called off after its level-1 pass has written the second key / payload buffers (the index buffer); the sort behind it
    starts from the untouched first ones."""
    n = 16 * MI + 3
    a = ol.splitmix_fill(n, ol.F32, 4660 + digit, 0xFFFFFFFF).view(np.uint32).copy()
    idx = np.arange(1000, 1000 + 26000 * 7, 7)
    # (f32 ascending: positive floats get their sign bit set, negative ones are complemented, radix_sort_basic_kdf.hpp:32-46)
    top = (digit ^ 0x80) if digit >= 0x80 else (~digit & 0xFF)
    a[idx] = (a[idx] & np.uint32(0x00FFFFFF)) | np.uint32(top << 24)
    _ranks_and_pairs(a.view(np.float32), None, ("level-1 overflow", hex(digit)))


def test_f32_pairs_hold_less_scratch_with_slots_in_the_spare_buffers(monkeypatch):
    """What the library holds after one key + payload sort of 24 Mi pairs (hipMemGetInfo): level-1 slots of 2 x 120 MiB, of which
    204 / 256 lie in the caller's second buffers by default."""
    n = 24 * MI
    held = {}
    for name in ("spare", "scratch"):
        if name == "scratch":
            monkeypatch.setenv("RSX_NO_AUX_SLOTS", "1")
        bits = torch.empty(n, dtype=torch.int32, device="cuda")
        rsa.fill_splitmix(bits, seed=4670)
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        ka, va = torch.empty_like(bits), torch.empty_like(vals)
        rsa.release_stream()
        rsa.reload_env()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        kr, vr, info = rsa.radix_sort_pairs(bits, ka, vals, va, dtype=rsa.F32)
        torch.cuda.synchronize()
        assert info.hybrid == 5
        held[name] = free0 - torch.cuda.mem_get_info()[0]
        del bits, vals, ka, va, kr, vr
    rsa.release_stream()
    assert held["spare"] + 150 * MI < held["scratch"], held


@pytest.mark.parametrize("shape", ["every key twice", "every key twice, small leaves", "every key twice, a wave per leaf",
                                   "low byte from 16 values", "low byte from 16 values, small leaves",
                                   "low byte from 16 values, a wave per leaf", "every leaf through the list"])
def test_f32_ranks_stable_through_the_compound_leaves(shape, monkeypatch):
    """rsx_leafp_kernel sorts (key half, position in the slot) compounds: equal keys must keep their order
    (radix_sort_rank.hpp:82-90).  Every key twice (ties everywhere, the bins even); a low byte with 16 values (ties and fat bins:
    the sample hands every leaf to rsx_leaf_pairs_kernel); RSX_LEAF16_MAXBIN=0 (every leaf through the list launch)."""
    n = (40 if shape.endswith("small leaves") else 10 if shape.endswith("a wave per leaf") else 72) * MI + 6
    if shape.startswith("every key twice"):
        half = ol.splitmix_fill(n // 2, ol.F32, 4800, 0xFFFFFFFF)
        a = np.concatenate([half, half])
    elif shape.startswith("low byte from 16 values"):
        a = ol.splitmix_fill(n, ol.F32, 4801, 0xFFFFFF0F)
    else:
        monkeypatch.setenv("RSX_LEAF16_MAXBIN", "0")
        a = ol.splitmix_fill(n, ol.F32, 4802, 0xFFFFFFFF)
    want, want_aux = ol.want_ranks(a, ol.F32, big=1 << 23)
    bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == 5, (shape, info.hybrid)
    assert info.result_in_aux == want_aux
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), shape


@pytest.mark.parametrize("case", ["f32 & 0xFFF000FF, 16 Mi", "f32 & 0xFFF000FF, 96 Mi", "f32 & 0xFFF000FF desc, 24 Mi", "i32 three runs, 40 Mi",
                                  "u32 four runs, 20 Mi"])
def test_rank_sort_by_packed_varying_bits(case):
    """Rank sorts of keys whose byte columns do not spread but whose VARYING bits, packed together, do (README.md:716-758;
    BASELINE.json's cfg 4 (iii): f32 & 0xFFF000FF -- 2, 32 and 256 values per column, 20 varying bits): the sample finds the varying
    bits, the level-1 pass packs them (and checks every key against the others), and the sort goes without a histogram on the
    packed keys (SegCtl::compact).  Ranks as the oracle's, ties in index order (every key 16 .. 4096 times), the route asserted."""
    name, size = case.rsplit(", ", 1)
    n = int(size.split()[0]) * MI + 77
    dt, order, mask = ol.F32, ol.ASC, 0xFFF000FF
    if name.endswith("desc"):
        order = ol.DESC
    elif name.startswith("i32"):
        dt, mask = ol.I32, 0xFF0FF0FC                     # runs 31-24, 19-12, 7-2: 22 varying bits, every byte column kept
    elif name.startswith("u32"):
        dt, mask = ol.U32, 0xF0F0FF0F                     # runs 31-28, 23-20, 15-8, 3-0: 20 varying bits
    a = ol.splitmix_fill(n, dt, 5100 + len(case), mask).view(np.uint32).copy()
    want, want_aux = ol.want_ranks(a, dt, order, big=1 << 22)
    bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=dt, order=order)
    torch.cuda.synchronize()
    assert info.hybrid == 5, (case, info.hybrid)
    assert info.result_in_aux == want_aux == 0
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), case
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), a), case          # the keys are only read


@pytest.mark.parametrize("case", ["five runs", "one key differs outside the varying bits", "fifteen varying bits", "key + payload"])
def test_packed_keys_are_not_for_everybody(case):
    """What the packing does not take falls back to the histogram-first sort: more than four runs of varying bits; fewer than 17
    varying bits; a key that differs from the first one in a bit the sample saw constant (found by the level-1 pass on EVERY key:
    the attempt is called off after that pass); key + payload sorts (the caller wants the keys back)."""
    n = 16 * MI + 5
    mask = {"five runs": 0xF8F8F8F9, "fifteen varying bits": 0xFE0000FF}.get(case, 0xFFF000FF)
    a = ol.splitmix_fill(n, ol.F32, 5200 + len(case), mask).view(np.uint32).copy()
    if case.startswith("one key"):
        a[n - 12345] ^= np.uint32(0x00004000)
    want, want_aux = ol.want_ranks(a, ol.F32, ol.ASC, big=1 << 22)
    bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
    if case == "key + payload":
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        ka, va = torch.empty_like(bits), torch.empty_like(vals)
        kr, vr, info = rsa.radix_sort_pairs(bits, ka, vals, va, dtype=rsa.F32)
        torch.cuda.synchronize()
        assert info.hybrid == 0, info.hybrid
        assert np.array_equal(vr.cpu().numpy().view(np.uint32), want)
        assert np.array_equal(kr.cpu().numpy().view(np.uint32), a[want])
        return
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=rsa.F32)
    torch.cuda.synchronize()
    assert info.hybrid == 0, (case, info.hybrid)
    assert info.result_in_aux == want_aux
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), case


def _grid_floats(n, seed):
    """SURVEY.md 8d cfg 4 (ii): (int24 - 2^23) * 2^-23, uniform in [-1, 1) on a grid of 2^-23 (exact on every platform)"""
    r = ol.splitmix_fill(n, ol.U64, seed, (1 << 64) - 1).view(np.uint64)
    k = ((r >> np.uint64(40)) & np.uint64(0xFFFFFF)).astype(np.int64) - (1 << 23)
    return (k.astype(np.float32) * np.float32(2.0 ** -23)).view(np.uint32).copy()


@pytest.mark.parametrize("case", ["[-1, 1) on a grid of 2^-23", "the same, descending", "prices: cents up to 40000.00", "one -0.0", "one NaN",
                                  "one key off the grid"])
def test_rank_sort_of_floats_on_a_grid(case):
    """f32 keys that are whole multiples of one power of two (BASELINE.json's cfg 4 (ii); measurements, prices) are fixed-point
    numbers: the sample finds the grid and the range, the level-1 pass converts every key (and checks it), and the rank sort goes
    without a histogram on integers that spread as the values do (SegCtl::ckind).  A key that is not on the grid, not finite, or
    -0.0 (which the reference orders before +0.0) calls the attempt off: the histogram-first sort then gives the same ranks."""
    n = 24 * MI + 11
    order, want_route = ol.ASC, 5
    a = _grid_floats(n, 5300 + len(case))
    if case.endswith("descending"):
        order = ol.DESC
    elif case.startswith("prices"):
        r = ol.splitmix_fill(n, ol.U32, 5301, 0xFFFFFFFF).view(np.uint32)
        a = ((r % np.uint32(4000001)).astype(np.float32) * np.float32(1.0 / 128.0)).view(np.uint32).copy()   # multiples of 2^-7 up to 31250
    elif case == "one -0.0":
        a[n // 3] = np.uint32(0x80000000)
        want_route = 0
    elif case == "one NaN":
        a[n // 5] = np.uint32(0x7FC00001)
        want_route = 0
    elif case == "one key off the grid":
        a[n - 777] = np.float32(0.3).view(np.uint32)       # 0.3 is no multiple of 2^-23 (a few of its low bits lie below the grid)
        want_route = 0
    want, want_aux = ol.want_ranks(a, ol.F32, order, big=1 << 22)
    bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
    ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
    ranks, info = rsa.radix_sort_rank(bits, ib, dtype=rsa.F32, order=order)
    torch.cuda.synchronize()
    assert info.hybrid == want_route, (case, info.hybrid)
    assert info.result_in_aux == want_aux
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), case
