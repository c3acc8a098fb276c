"""rsx_sort_inplace_async chooses its route on the device (round 4): one MSB pass + leaves for mid-size arrays, the sort
without a histogram for large ones, histogram + one pass per kept column otherwise -- with nobody reading a verdict back.

What is checked: the sorted keys against the CPU restatement of rs_sort_main (radix_sort.hpp:31-93), always in `buf`, and the
route the device took (rsx_async_route, as rsx_info.hybrid) -- for plain calls and for a HIP graph captured ONCE and replayed on
inputs that take different routes: the histogram-first kernels enqueued behind an attempt must do nothing when it went through
and everything when it was called off.
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

_CARRIER = {4: np.int32, 8: np.int64}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.fixture(autouse=True)
def _fresh(monkeypatch):
    rsa.reload_env()
    yield


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(_CARRIER[a.itemsize]).copy()).cuda()


def run(a, dt, order=ol.ASC):
    buf = to_dev(a)
    scratch = torch.full_like(buf, 0x5B5B5B5B)
    rsa.radix_sort_inplace_async(buf, scratch, dtype=dt, order=order)
    route = rsa.async_route()
    got = buf.cpu().numpy().view(ol.NP_BITS[dt])
    want, _, winfo = ol.oracle_sort(a, dt, order)
    assert np.array_equal(got, want), (len(a), dt, order, route)
    return route, winfo


@pytest.mark.parametrize("dt", [ol.U32, ol.F32, ol.I32, ol.U64], ids=["u32", "f32", "i32", "u64"])
def test_one_level_on_the_device(dt):
    """Mid-size arrays: the device-side plan picks one MSB pass + leaves (route 1) for keys that spread over their top column
    and one pass per column (route 0) for keys that do not."""
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    for n in (70001, (1 << 20) + 3, 3000000):
        for order in (ol.ASC, ol.DESC):
            route, _ = run(ol.splitmix_fill(n, dt, 7000 + n % 89 + order, full), dt, order)
            assert route == 1, (n, dt, order, route)
    n = (1 << 20) + 17
    # a constant top byte below which the keys spread: still one level (by the highest KEPT column), result in `buf` (odd count)
    route, winfo = run(ol.splitmix_fill(n, dt, 7100, full >> 8), dt)
    assert route == 1 and winfo.ncols == ol.DTYPE_SIZE[dt] - 1
    # half the keys in one top digit: its bucket is larger than a leaf -> one pass per column
    a = ol.splitmix_fill(n, dt, 7101, full)
    half = a.copy()
    half[::2] &= ol.NP_BITS[dt](full >> 8)
    route, _ = run(half, dt)
    assert route == 0, route
    # few values in the top byte: buckets several times the mean, the leaves' larger shapes (chosen on the device)
    for bits in ((7, 6) if ol.DTYPE_SIZE[dt] == 4 else (7,)):     # (the largest leaf holds 32 Ki 4-byte or 16 Ki 8-byte keys)
        top_mask = full ^ (((0xFF << bits) & 0xFF) << (8 * ol.DTYPE_SIZE[dt] - 8))
        route, _ = run(ol.splitmix_fill(n, dt, 7102 + bits, top_mask), dt)
        assert route == 1, (bits, route)
    # sorted input: nothing moves
    srt = np.sort(ol.kdf_keys(a, dt))
    if dt in (ol.U32, ol.U64):
        route, winfo = run(srt, dt)
        assert route == 0 and winfo.early_exit == 2


def test_one_level_with_a_caller_owned_workspace():
    n = (1 << 21) + 5
    a = ol.splitmix_fill(n, ol.U32, 7200, 0xFFFFFFFF)
    buf, scratch = to_dev(a), torch.empty(n, dtype=torch.int32, device="cuda")
    ws = torch.empty(rsa.workspace_bytes(n, rsa.U32), dtype=torch.uint8, device="cuda")
    rsa.radix_sort_inplace_async_ws(buf, scratch, ws, dtype=rsa.U32)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), ol.oracle_sort(a, ol.U32)[0])


@pytest.mark.parametrize("dt,n", [(ol.U32, 10000000), (ol.F32, (1 << 24) + 99), (ol.U32, (1 << 27) + 12345), (ol.U64, (3 << 24) + 7)],
                         ids=["u32-1e7", "f32-16Mi", "u32-128Mi", "u64-48Mi"])
def test_without_histogram_on_the_device(dt, n):
    """Large arrays: the attempt without a histogram is enqueued first; it goes through for keys that spread over all their
    columns (route 5) and is called off -- the histogram-first kernels behind it then do the work -- for others."""
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    a = ol.splitmix_fill(n, dt, 7300, full)
    for order in (ol.ASC, ol.DESC):
        route, _ = run(a, dt, order)
        assert route == 5, (dt, n, order, route)
    # a constant low byte (4-byte keys need all four columns; 8-byte keys: an odd number of kept columns -> the copy home)
    route, winfo = run(a & ol.NP_BITS[dt](full ^ 0xFF), dt)
    assert route == (5 if ol.DTYPE_SIZE[dt] == 8 else 0), route
    if dt != ol.F32:     # (a float's derived low byte is 0x00 or 0xFF by its sign: two values, a kept -- and hot -- column)
        assert winfo.ncols == ol.DTYPE_SIZE[dt] - 1
    # a dominant top digit: the sample calls the attempt off
    half = a.copy()
    half[::2] &= ol.NP_BITS[dt](full >> 8)
    route, _ = run(half, dt)
    assert route == 0, route
    # a slot overflows although the sample saw nothing: one (digit, digit) pair with many times its share
    b = a.copy()
    top = ol.NP_BITS[dt](0x4321) << ol.NP_BITS[dt](8 * ol.DTYPE_SIZE[dt] - 16)
    low = ol.NP_BITS[dt](full >> 16)
    idx = np.arange(5000, 5000 + 9000 * 11, 11)
    b[idx] = (b[idx] & low) | top
    route, _ = run(b, dt)
    assert route == 0, route


def test_u64_below_two_to_the_40_on_the_device():
    """8-byte keys in which nothing below the level-1 digit varies above bit 32 (SegCtl::narrow == 2): the device-scheduled sort
    enqueues both forms of both MSB passes, the sample picks the one that keeps low words in the level-1 slots -- which lie in the
    caller's scratch buffer; odd numbers of kept columns end there (the copy home); a key that varies above bit 40 after all calls
    the attempt off and the gated passes sort; and one captured graph replays both."""
    n = (96 << 20) + 3
    a = ol.splitmix_fill(n, ol.U64, 7400, 0xFFFFFFFFFF)
    for order in (ol.ASC, ol.DESC):
        route, winfo = run(a, ol.U64, order)
        assert route == 5 and winfo.ncols == 5, (order, route)
    route, winfo = run(a & np.uint64(0xFFFFFFFF), ol.I64)
    assert route == 5 and winfo.ncols == 4
    b = a.copy()
    b[n // 5] |= np.uint64(1 << 50)
    route, _ = run(b, ol.U64)
    assert route != 5, route
    # one capture, replayed on keys the narrow form takes, on keys it must leave alone, and again
    s = torch.cuda.Stream()
    buf = to_dev(a)
    scratch = torch.empty_like(buf)
    with torch.cuda.stream(s):
        rsa.radix_sort_inplace_async(buf, scratch, dtype=ol.U64, stream=s)     # sizes the workspace outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_inplace_async(buf, scratch, dtype=ol.U64, stream=torch.cuda.current_stream())
    for name, keys, want_route in (("below 2^40", a, 5), ("all 64 bits", ol.splitmix_fill(n, ol.U64, 7401, 0xFFFFFFFFFFFFFFFF), 5),
                                  ("one key above", b, None),
                                  # (a lost attempt: the device-side back-off lets the next sort of the context go by the gated passes)
                                  ("below 2^40, backing off", a[::-1].copy(), None), ("below 2^40 again", a[::-1].copy(), 5)):
        buf.copy_(to_dev(keys))
        scratch.fill_(0x6C6C6C6C)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        route = rsa.async_route(s)
        assert (route == want_route) if want_route is not None else (route != 5), (name, route)
        assert np.array_equal(buf.cpu().numpy().view(np.uint64), ol.oracle_sort(keys, ol.U64)[0]), name
    rsa.release_stream(s)


@pytest.mark.parametrize("n", [1 << 23, 1 << 27])
def test_routes_inside_one_captured_graph(n, monkeypatch):
    """One capture, replayed on inputs that take different routes.  2^23 keys lie between the reach of one level and the
    default floor of the sorts without a histogram: RSX_BLIND_MIN_LOG2=23 puts the floor there for this test."""
    if n == 1 << 23:
        monkeypatch.setenv("RSX_BLIND_MIN_LOG2", "23")
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
    base = ol.splitmix_fill(n, ol.U32, 7400, 0xFFFFFFFF).view(np.uint32)
    cases = [("uniform", base, 5), ("column 1 constant", base & np.uint32(0xFFFF00FF), 0), ("uniform again", base[::-1].copy(), 5),
             ("top digit dominant", np.where(np.arange(n) % 2 == 0, base & np.uint32(0x00FFFFFF), base).astype(np.uint32), 0),
             ("sorted", np.sort(base[: n // 4]).repeat(4)[:n], 0), ("uniform, third time", base ^ np.uint32(0x5A5A5A5A), 5)]
    for name, a, want_route in cases:
        a = np.ascontiguousarray(a)
        buf.copy_(to_dev(a))
        scratch.fill_(0x6C6C6C6C)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()     # (the replay runs on the current stream; rsx_async_route waits for the capture stream's)
        route = rsa.async_route(s)
        assert route == want_route, (name, route)
        want, _, _ = ol.oracle_sort(a, ol.U32)
        assert np.array_equal(buf.cpu().numpy().view(np.uint32), want), name
    rsa.release_stream(s)


@pytest.mark.parametrize("n", [(1 << 24) + 5, (1 << 25) + 77])
def test_ranks_and_pairs_without_histogram_on_the_device(n):
    """rsx_sort_rank_inplace_async and rsx_sort_pairs_inplace_async (4-byte keys, 4-byte indices / payloads, 16 Mi .. 2^28): the
    attempt without a histogram is enqueued first, its leaves write the ranks to the first half of the index buffer / the pairs
    to (keys, vals); equal keys keep their order (radix_sort_rank.hpp:82-90).  Called off (a dominant top digit; a constant
    column): the gated histogram-first kernels do the work."""
    base = ol.splitmix_fill(n, ol.F32, 7500, 0xFFFFFFFF)
    twice = np.concatenate([base[: n // 2], base[: n - n // 2]])          # every key (at least) twice: ties everywhere
    half = twice.copy()
    half[::2] &= np.uint32(0x00FFFFFF)                                      # a dominant top digit
    const_col = twice & np.uint32(0xFFFF00FF)                               # column 1 constant: three kept columns
    # (routes of the rank sort / of the key + payload sort.  "constant column": the f32 KDF makes the constant byte 0x00 or 0xFF by
    # the sign, so all four columns are kept; no byte scheme takes a column with two values, but the RANK sort packs the 24
    # varying bits -- SegCtl::compact, round 5 -- and stays on route 5; the pair sort wants its keys back and does not)
    cases = (("uniform, ties", twice, 5, 5), ("dominant top digit", half, 0, 0), ("constant column", const_col, 5, 0))
    for name, a, want_route, want_pair_route in (cases if n < (1 << 25) else cases[:2]):
        a = np.ascontiguousarray(a)
        want, _ = ol.want_ranks(a, ol.F32, big=1 << 22)
        bits = torch.from_numpy(a.view(np.int32).copy()).cuda()
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks = rsa.radix_sort_rank_inplace_async(bits, ib, dtype=rsa.F32)
        route = rsa.async_route()
        assert route == want_route, (name, "ranks", route)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32), a.view(np.uint32)), name      # the keys are only read
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), (name, "ranks")
        del ib, ranks
        vals = torch.arange(n, dtype=torch.int32, device="cuda")
        ks, vs = torch.empty_like(bits), torch.empty_like(vals)
        rsa.radix_sort_pairs_inplace_async(bits, ks, vals, vs, dtype=rsa.F32)
        route = rsa.async_route()
        assert route == want_pair_route, (name, "pairs", route)
        assert np.array_equal(vals.cpu().numpy().view(np.uint32), want), (name, "pairs")
        assert np.array_equal(bits.cpu().numpy().view(np.uint32), a.view(np.uint32)[want]), (name, "pairs' keys")
        del vals, ks, vs, bits


def test_rank_routes_inside_one_captured_graph():
    """One capture of rsx_sort_rank_inplace_async, replayed on inputs that take different routes."""
    n = 1 << 25
    s = torch.cuda.Stream()
    bits = torch.empty(n, dtype=torch.int32, device="cuda")
    ib = torch.empty(2 * n, dtype=torch.int32, device="cuda")
    with torch.cuda.stream(s):
        rsa.fill_splitmix(bits, seed=1, stream=s)
        rsa.radix_sort_rank_inplace_async(bits, ib, dtype=ol.U32, stream=s)     # sizes the workspace outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rsa.radix_sort_rank_inplace_async(bits, ib, dtype=ol.U32, stream=torch.cuda.current_stream())
    base = ol.splitmix_fill(n, ol.U32, 7600, 0xFFFFFFFF).view(np.uint32)
    cases = [("uniform", base, 5), ("top digit dominant", np.where(np.arange(n) % 2 == 0, base & np.uint32(0x00FFFFFF), base).astype(np.uint32), 0),
             ("sorted", np.sort(base), 0), ("uniform, ties", np.concatenate([base[: n // 2]] * 2), 5)]
    for name, a, want_route in cases:
        a = np.ascontiguousarray(a)
        bits.copy_(to_dev(a))
        ib.fill_(-1)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        route = rsa.async_route(s)
        assert route == want_route, (name, route)
        want, _ = ol.want_ranks(a, ol.U32, big=1 << 22)
        assert np.array_equal(ib[:n].cpu().numpy().view(np.uint32), want), name
    rsa.release_stream(s)


@pytest.mark.parametrize("dt,n", [(ol.U32, (1 << 25) + 33), (ol.U64, (1 << 24) + 9)], ids=["u32-32Mi", "u64-16Mi"])
def test_a_kept_graph_takes_the_fast_route_inside_its_own_workspace(dt, n):
    """rsx_sort_inplace_async_ws with a workspace of rsx_workspace_bytes_fast: the slots of the sort without a histogram lie in
    the CALLER's workspace, so a graph that is kept -- nothing it touches can move -- runs on route 5 like every other sort
    (round 4: "never makes the attempt").  One capture, replayed on inputs that take routes 5, 0 (a dominant top digit), 0 (sorted), 5;
    a second graph with its own workspace replays in between; results against the oracle, routes read back from the workspace."""
    carrier = torch.int32 if ol.DTYPE_SIZE[dt] == 4 else torch.int64
    full = (1 << (8 * ol.DTYPE_SIZE[dt])) - 1
    s = torch.cuda.Stream()
    graphs = []
    for k in range(2):
        buf = torch.empty(n, dtype=carrier, device="cuda")
        scratch = torch.empty_like(buf)
        ws = torch.empty(rsa.workspace_bytes_fast(n, dt), dtype=torch.uint8, device="cuda")
        assert ws.numel() > rsa.workspace_bytes(n, dt) + n * ol.DTYPE_SIZE[dt] // 4
        with torch.cuda.stream(s):
            rsa.fill_splitmix(buf, seed=1 + k, stream=s)
            rsa.radix_sort_inplace_async_ws(buf, scratch, ws, dtype=dt, stream=s)     # (the library's first use outside a capture)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rsa.radix_sort_inplace_async_ws(buf, scratch, ws, dtype=dt, stream=torch.cuda.current_stream())
        graphs.append((g, buf, scratch, ws))
    base = ol.splitmix_fill(n, dt, 7700, full)
    half = base.copy()
    half[::2] &= ol.NP_BITS[dt](full >> 8)
    cases = [("uniform", base, 5), ("top digit dominant", half, 0), ("sorted", np.sort(base), 0), ("uniform again", base[::-1].copy(), 5)]
    for i, (name, a, want_route) in enumerate(cases):
        g, buf, scratch, ws = graphs[i & 1]
        a = np.ascontiguousarray(a)
        buf.copy_(to_dev(a))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert rsa.async_route_ws(ws, n, dt, s) == want_route, name
        want, _, _ = ol.oracle_sort(a, dt)
        assert np.array_equal(buf.cpu().numpy().view(ol.NP_BITS[dt]), want), name
    # a workspace of the old size still works: histogram first
    g, buf, scratch, _ = graphs[0]
    small = torch.empty(rsa.workspace_bytes(n, dt), dtype=torch.uint8, device="cuda")
    buf.copy_(to_dev(base))
    rsa.radix_sort_inplace_async_ws(buf, scratch, small, dtype=dt)
    torch.cuda.synchronize()
    assert rsa.async_route_ws(small, n, dt) == 0
    assert np.array_equal(buf.cpu().numpy().view(ol.NP_BITS[dt]), ol.oracle_sort(base, dt)[0])


def test_the_profile_books_what_the_device_chose():
    """rsx_profile around device-scheduled sorts (the multi-GPU path's local sorts, bench.py's roofline object): the launches of
    the route the device did NOT take -- the gated histogram-first kernels behind an attempt that went through, the attempt's own
    kernels when it was called off -- add no bytes and no launch to the classes, only to called_off_*; the verdict travels to a
    pinned word behind the attempt, nobody waits for it."""
    n = (64 << 20) + 11
    buf = torch.empty(n, dtype=torch.int32, device="cuda")
    scratch = torch.empty_like(buf)
    # uniform keys: the attempt goes through -- one level-1 pass (whole keys), one level-2 pass (two bytes per key), leaves
    rsa.fill_splitmix(buf, 9100)
    rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)   # (scratch memory allocated outside the window)
    rsa.fill_splitmix(buf, 9101)
    torch.cuda.synchronize()
    rsa.profile_begin()
    rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)
    assert rsa.async_route() == 5
    p = rsa.profile_end()
    assert p.hist_launches == 0 and p.hist_bytes == 0
    assert p.scatter_launches == 1 and p.scatter_bytes == 8 * n
    assert p.narrow_launches == 1 and p.narrow_bytes == 6 * n
    assert p.leaf_launches >= 1 and p.leaf_bytes == 6 * n
    assert p.called_off_launches >= 5     # histogram, four gated passes (+ the leaf shapes of the one-level route)
    # keys with one hot top digit: the sample calls the attempt off -- the histogram and four passes are what ran
    a = ol.splitmix_fill(n, ol.U32, 9102, 0xFFFFFFFF).view(np.uint32).copy()
    a[::2] &= np.uint32(0x00FFFFFF)
    buf.copy_(torch.from_numpy(a.view(np.int32)))
    torch.cuda.synchronize()
    rsa.profile_begin()
    rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)
    route = rsa.async_route()
    p = rsa.profile_end()
    assert route == 0
    assert p.hist_launches == 1 and p.hist_bytes == 4 * n
    assert p.scatter_launches == 4 and p.scatter_bytes == 4 * 8 * n
    assert p.narrow_launches == 0 and p.leaf_launches == 0
    assert p.called_off_launches >= 1
    got = buf.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, np.sort(a))


def test_device_scheduled_sorts_back_off_on_the_device():
    """An input that passes the sample and then overflows a level-1 slot (a top digit with 1.6 times its share) costs a
    device-scheduled sort its attempt -- a full pass -- on every call, and nobody reads a verdict back: the back-off lives in the
    control block (SegCtl::boff_skip).  After a lost attempt the next sort of the context does not try, the one after does, then
    two do not: seen through rsx_profile, which books an attempt's pass under called_off_ms.  An attempt that goes through
    resets it.  Every sort is checked against the oracle's order."""
    n = (48 << 20) + 99
    base = ol.splitmix_fill(n, ol.U32, 9200, 0xFFFFFFFF).view(np.uint32).copy()
    a = base.copy()
    extra = np.flatnonzero((a >> 24) == 0x11)[: int(0.6 * n / 256)]
    a[extra] = (a[extra] & np.uint32(0x00FFFFFF)) | np.uint32(0x77000000)
    want = np.sort(a)
    src = torch.from_numpy(a.view(np.int32).copy()).cuda()
    buf = torch.empty_like(src)
    scratch = torch.empty_like(src)
    tried = []
    for i in range(6):
        buf.copy_(src)
        torch.cuda.synchronize()
        rsa.profile_begin()
        rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)
        route = rsa.async_route()
        p = rsa.profile_end()
        assert route == 0, (i, route)
        assert np.array_equal(buf.cpu().numpy().view(np.uint32), want), i
        tried.append(p.called_off_ms)     # (a lost attempt at 48 Mi keys: ~0.10 ms; the launches that return at once: ~0.02)
    mid = (min(tried) + max(tried)) / 2     # (what the launches that return at once cost depends on what the context sorted before)
    assert max(tried) > 1.5 * min(tried) and [t > mid for t in tried] == [True, False, True, False, False, True], tried
    # uniform keys: the attempt is made at once only when no skip is pending -- and going through clears the doubling
    rsa.fill_splitmix(buf, 9201)
    rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)     # (skip 1 of 4 pending after the sixth sort above)
    for _ in range(4):
        rsa.fill_splitmix(buf, 9202)
        rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32)
    assert rsa.async_route() == 5
    got = buf.cpu().numpy().view(np.uint32)
    assert np.all(got[1:] >= got[:-1])


def _presplit(n, G, seed):
    """What rank 0 of G receives in a distributed sort: keys below 2^32 / G in G pieces, each in stable order of its top byte."""
    a = ol.splitmix_fill(n, ol.U32, seed) & np.uint32((1 << 32) // G - 1)
    pieces = []
    for p in range(G):
        piece = a[p * (n // G):(p + 1) * (n // G)]
        pieces.append(piece[np.argsort(piece >> np.uint32(24), kind="stable")])
    return np.concatenate(pieces)


@pytest.mark.parametrize("G", [1, 2, 4])
def test_keys_an_msd_split_has_ordered_take_the_fast_route_with_the_callers_word(G):
    """rsx_sort_inplace_async_hint (RSX_HINT_EVEN_TOP_DIGITS): keys in pieces that are each in order of their top byte are even over
    the array and clustered at every place the sample reads -- without the hint the attempt without a histogram is called off
    (G = 1, 2), with it the sort takes route 5; the result is the oracle's either way.  A hint that is WRONG (keys that do cluster)
    costs the attempt, not the result."""
    n = 1 << 24
    a = _presplit(n, G, 40 + G)
    want, _, _ = ol.oracle_sort(a, ol.U32)
    routes = []
    for hints in (0, rsa.HINT_EVEN_TOP_DIGITS):
        rsa.reload_env()
        buf = to_dev(a)
        scratch = torch.empty_like(buf)
        rsa.radix_sort_inplace_async(buf, scratch, dtype=rsa.U32, hints=hints)
        routes.append(rsa.async_route())
        assert np.array_equal(buf.cpu().numpy().view(np.uint32), want), (G, hints)
    assert routes[1] == 5, routes
    if G <= 2:
        assert routes[0] != 5, routes
    # a wrong hint: half the keys share their top two bytes
    b = a.copy()
    b[::2] = (b[::2] & np.uint32(0xFFFF)) | np.uint32(0x12340000 & ((1 << 32) // G - 1))
    rsa.reload_env()
    buf = to_dev(b)
    rsa.radix_sort_inplace_async(buf, torch.empty_like(buf), dtype=rsa.U32, hints=rsa.HINT_EVEN_TOP_DIGITS)
    assert rsa.async_route() != 5
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), ol.oracle_sort(b, ol.U32)[0])
