"""Host logic of the multi-GPU sort (radix_sorting_amd/multi.py) on CPU: world_size-2 and -3 gloo runs.

The device steps are injected: an oracle-backed engine (numpy + the C restatement) stands in for
HipEngine, so what is exercised here is the splitter choice, the count exchange, the all-to-all-v
plumbing and the ordering argument (concatenated rank outputs == stable sort of concatenated input).
The HIP engine itself is covered by tests/test_gpu_multi.py on the GPU box.
"""
import os
import socket
import sys

import numpy as np
import pytest

import oracle_lib as ol
from radix_sorting_amd import multi

torch = pytest.importorskip("torch")
import torch.distributed as dist          # noqa: E402
import torch.multiprocessing as mp        # noqa: E402


class OracleEngine:
    """CPU stand-in for multi.HipEngine (tests only): same interface, numpy / oracle arithmetic."""

    def __init__(self, dtype, order=0):
        self.dtype, self.order = dtype, order
        self.kb = ol.DTYPE_SIZE[dtype]
        self.split_passes = 0

    def _bits(self, t):
        return t.numpy().view(ol.NP_BITS[self.dtype])

    def msd_split(self, shard, out, column=-1):
        bits = self._bits(shard)
        k = ol.kdf_keys(bits, self.dtype, self.order)
        column = self.kb - 1 if column < 0 else column
        top = ((k >> ol.NP_BITS[self.dtype](8 * column)) & ol.NP_BITS[self.dtype](0xFF)).astype(np.int64)
        out.numpy().view(bits.dtype)[:bits.size] = bits[np.argsort(top, kind="stable")]
        self.split_passes += 1
        return np.bincount(top, minlength=256).astype(np.uint64)

    def histogram(self, shard):
        k = ol.kdf_keys(self._bits(shard), self.dtype, self.order)
        h = np.zeros(self.kb * 256, dtype=np.int64)
        for c in range(self.kb):
            d = ((k >> ol.NP_BITS[self.dtype](8 * c)) & ol.NP_BITS[self.dtype](0xFF)).astype(np.int64)
            h[256 * c:256 * c + 256] = np.bincount(d, minlength=256)
        return torch.from_numpy(h)

    def msd_split_known(self, shard, out, column, hist_all, hot=False):
        counts = self.msd_split(shard, out, column)
        assert bool(hot) == (counts.sum() > 0 and int(counts.max()) >= int(counts.sum()) // 8 + 1)   # (the engine is told what the counts say)
        assert np.array_equal(counts.astype(np.int64), hist_all.numpy()[256 * column:256 * column + 256])

    def local_sort(self, keys, aux):
        res, in_aux, info = ol.oracle_sort(self._bits(keys), self.dtype, self.order)
        target = aux if in_aux else keys
        target.numpy().view(res.dtype)[:res.size] = res
        return target[:res.size], info

    def sort_inplace_async(self, buf, scratch):
        res, _, _ = ol.oracle_sort(self._bits(buf), self.dtype, self.order)
        buf.numpy().view(res.dtype)[:res.size] = res

    def empty(self, n, like):
        return torch.empty(n, dtype=like.dtype)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _skew(whole, dtype, skew):
    """`skew` percent of the keys get the top KDF byte 0x42 (ascending unsigned view): one dominant top digit."""
    if not skew:
        return whole
    bits = 8 * ol.DTYPE_SIZE[dtype]
    out = whole.copy()
    sel = (np.arange(out.size) % 100) < skew
    top = ol.NP_BITS[dtype](0x42) << ol.NP_BITS[dtype](bits - 8)
    low = ol.NP_BITS[dtype]((1 << (bits - 8)) - 1)
    out[sel] = (out[sel] & low) | top
    return out


def _skew2(whole, dtype, skew):
    """`skew` percent of the keys get the top TWO KDF bytes 0x42, 0x17: a dominant (digit, next byte) pair, which one
    level of refinement cannot divide."""
    if not skew:
        return whole
    bits = 8 * ol.DTYPE_SIZE[dtype]
    out = whole.copy()
    sel = (np.arange(out.size) % 100) < skew
    top = ol.NP_BITS[dtype](0x4217) << ol.NP_BITS[dtype](bits - 16)
    low = ol.NP_BITS[dtype]((1 << (bits - 16)) - 1)
    out[sel] = (out[sel] & low) | top
    return out


def _worker(rank, world, port, dtype, order, n_per_rank, mask, seed, outdir, chunks=None, skew=0, skew2=0, slices=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        carrier = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[ol.DTYPE_SIZE[dtype]]
        n = n_per_rank[rank]
        first = sum(n_per_rank[:rank])
        whole = _skew2(_skew(ol.splitmix_fill(sum(n_per_rank), dtype, seed, mask), dtype, skew), dtype, skew2)
        shard = torch.from_numpy(whole[first:first + n].view(carrier).copy())
        scratch = None
        if skew:      # preallocated buffers that are too small for what this rank receives must be replaced, not trusted
            scratch = {"part": torch.empty(n, dtype=shard.dtype), "recv": torch.empty(n // 2, dtype=shard.dtype),
                       "aux": torch.empty(n // 2, dtype=shard.dtype)}
        engine = OracleEngine(dtype, order)
        res, stats = multi.distributed_sort(shard, engine, chunks=chunks, scratch=scratch, split_slices=slices)
        np.save(os.path.join(outdir, "out%d.npy" % rank), res.numpy().view(ol.NP_BITS[dtype]).copy())
        np.save(os.path.join(outdir, "passes%d.npy" % rank), np.asarray([engine.split_passes, stats.get("refine_levels", 0),
                                                                         stats.get("split_column", -1)]))
        np.save(os.path.join(outdir, "slices%d.npy" % rank), np.asarray([stats.get("split_slices", 1)]))
        np.save(os.path.join(outdir, "recv%d.npy" % rank), np.asarray(stats.get("recv_counts", [n])))
        np.save(os.path.join(outdir, "heavy%d.npy" % rank), np.asarray(stats.get("heavy_digits", []), dtype=np.int64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype,order,mask", [
    (2, ol.U32, 0, 0xFFFFFFFF),
    (2, ol.F32, 1, 0xFFFFFFFF),
    (3, ol.U32, 0, 0x00FFFFFF),       # constant top byte: split by byte 2
    (2, ol.I64, 0, 0xFFFFFFFFFFFFFFFF),
    (3, ol.U16, 0, 0xFFFF),
])
def test_distributed_sort_matches_single_sort(tmp_path, world, dtype, order, mask):
    n_per_rank = [40000 + 137 * r for r in range(world)]      # ragged shards
    seed = 91
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, order, n_per_rank, mask, seed, str(tmp_path)), nprocs=world, join=True)
    whole = ol.splitmix_fill(sum(n_per_rank), dtype, seed, mask)
    want, _, _ = ol.oracle_sort(whole, dtype, order)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert got.size == want.size
    assert np.array_equal(got, want)          # bit-identical to sorting the whole array on one rank
    sizes = [np.load(os.path.join(str(tmp_path), "out%d.npy" % r)).size for r in range(world)]
    if dtype == ol.U32:
        assert max(sizes) < 1.2 * sum(sizes) / world      # uniform keys -> balanced splitters, also below a constant top byte


@pytest.mark.parametrize("world,dtype,order,skew,chunks", [(2, ol.U32, 0, 90, None), (3, ol.U32, 0, 90, 1), (2, ol.F32, 1, 95, None),
                                                            (3, ol.U64, 0, 60, 4), (2, ol.U8, 0, 90, None)])
def test_distributed_sort_with_a_dominant_top_digit(tmp_path, world, dtype, order, skew, chunks):
    """SURVEY.md section 7 "Skew in the exchange" (README.md:647-650 is the one-line spec): 60-95 % of the keys share one top
    byte.  An 8-bit MSD digit cannot split them; the heavy digit is split by the next byte as well, the receive buffers
    follow the exchanged counts, and the result is still bit-identical to one rank sorting everything."""
    n_per_rank = [30000 + 211 * r for r in range(world)]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, order, n_per_rank, (1 << 64) - 1, 77, str(tmp_path), chunks, skew), nprocs=world,
             join=True)
    whole = _skew(ol.splitmix_fill(sum(n_per_rank), dtype, 77), dtype, skew)
    want, _, _ = ol.oracle_sort(whole, dtype, order)
    outs = [np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(outs), want)
    heavy = np.load(os.path.join(str(tmp_path), "heavy0.npy"))
    if ol.DTYPE_SIZE[dtype] > 1:
        assert heavy.size == 1                                  # the dominant digit was refined by the next byte ...
        sizes = [o.size for o in outs]
        assert max(sizes) < 1.35 * sum(sizes) / world           # ... and the ranks are balanced again
    else:
        assert heavy.size == 0                                  # one-byte keys: nothing below to refine by


@pytest.mark.parametrize("world,dtype,mask,column", [(2, ol.U32, 0x00FFFFFF, 2), (3, ol.U64, 0x000000FFFFFFFFFF, 4),
                                                     (2, ol.U64, 0x00000000000000FF, 0)])
def test_constant_top_bytes_cost_one_split_pass(tmp_path, world, dtype, mask, column):
    """The byte to split by comes from ONE read of the shard (every column's counts, one all-gather): keys whose top bytes
    are constant over all ranks are split exactly once, by the highest byte that varies -- no trial pass per constant byte."""
    n_per_rank = [20000 + 97 * r for r in range(world)]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, 0, n_per_rank, mask, 5, str(tmp_path)), nprocs=world, join=True)
    whole = ol.splitmix_fill(sum(n_per_rank), dtype, 5, mask)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ol.oracle_sort(whole, dtype)[0])
    for r in range(world):
        passes, levels, col = np.load(os.path.join(str(tmp_path), "passes%d.npy" % r))
        assert passes == 1 and levels == 0 and col == column, (r, passes, levels, col)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_dominant_digit_and_byte_pair_is_refined_by_the_third_byte(tmp_path, world):
    """80 % of the keys share their top TWO bytes: refining the heavy top digit by the next byte leaves one (digit, byte) bin
    as heavy as before; the refinement goes on by the third byte, and the ranks end up balanced."""
    dtype = ol.U32
    n_per_rank = [12000 + 53 * r for r in range(world)]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, 0, n_per_rank, 0xFFFFFFFF, 29, str(tmp_path), None, 0, 80), nprocs=world, join=True)
    whole = _skew2(ol.splitmix_fill(sum(n_per_rank), dtype, 29, 0xFFFFFFFF), dtype, 80)
    outs = [np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(outs), ol.oracle_sort(whole, dtype)[0])
    _, levels, _ = np.load(os.path.join(str(tmp_path), "passes0.npy"))
    assert levels == 2
    sizes = [o.size for o in outs]
    assert max(sizes) < 1.35 * sum(sizes) / world


@pytest.mark.parametrize("chunks", [1, 3, 16])
def test_distributed_sort_chunk_counts(tmp_path, chunks):
    """One all-to-all and one sort (chunks = 1) against the pipelined form with few and with many sub-ranges."""
    world, dtype = 2, ol.U32
    n_per_rank = [30000, 41111]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, 0, n_per_rank, 0xFFFFFFFF, 13, str(tmp_path), chunks), nprocs=world, join=True)
    whole = ol.splitmix_fill(sum(n_per_rank), dtype, 13, 0xFFFFFFFF)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ol.oracle_sort(whole, dtype)[0])


@pytest.mark.parametrize("world,n_per_rank,slices,chunks,dtype", [
    (2, [30000, 41111], 2, 4, ol.U32),
    (3, [50000, 20001, 9000], 3, 3, ol.I32),
    (2, [70000, 3000], 4, 2, ol.U64),        # the second rank's shard is shorter than one part: its later parts are empty
])
def test_split_pass_in_slices(tmp_path, world, n_per_rank, slices, chunks, dtype):
    """The split pass in consecutive parts of the shard (the first sub-range's pieces of part 0 leave before the rest is
    split): a piece is one run per part, the receive layout keeps (sub-range, source rank, part) order, the result is the
    single sort's; every non-empty part costs one split pass."""
    port = _free_port()
    mask = (1 << (8 * ol.DTYPE_SIZE[dtype])) - 1
    mp.spawn(_worker, args=(world, port, dtype, 0, n_per_rank, mask, 17, str(tmp_path), chunks, 0, 0, slices), nprocs=world, join=True)
    whole = ol.splitmix_fill(sum(n_per_rank), dtype, 17, mask)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ol.oracle_sort(whole, dtype)[0])
    for r in range(world):
        assert int(np.load(os.path.join(str(tmp_path), "slices%d.npy" % r))[0]) == slices
        b = multi.slice_bounds(n_per_rank[r], slices)
        nonempty = sum(1 for i in range(slices) if b[i + 1] > b[i])
        assert int(np.load(os.path.join(str(tmp_path), "passes%d.npy" % r))[0]) == nonempty


def test_sliced_split_steps_aside_for_heavy_bins(tmp_path):
    """A dominant top digit needs its run contiguous for the refinement: the split is then made in one piece."""
    world, dtype, n_per_rank = 2, ol.U32, [40000, 40000]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, dtype, 0, n_per_rank, 0xFFFFFFFF, 19, str(tmp_path), 4, 70, 0, 3), nprocs=world, join=True)
    whole = _skew(ol.splitmix_fill(sum(n_per_rank), dtype, 19, 0xFFFFFFFF), dtype, 70)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, ol.oracle_sort(whole, dtype)[0])
    assert int(np.load(os.path.join(str(tmp_path), "slices0.npy"))[0]) == 1


def test_slice_bounds():
    assert multi.slice_bounds(100000, 1) == [0, 100000]
    b = multi.slice_bounds(100000, 3)
    assert b[0] == 0 and b[-1] == 100000 and all(x % 4096 == 0 for x in b[:-1]) and b == sorted(b)
    assert multi.slice_bounds(3000, 4) == [0, 3000, 3000, 3000, 3000]
    assert multi.slice_bounds(0, 2) == [0, 0, 0]


def test_heavy_digits():
    h = np.full(256, 100, dtype=np.uint64)
    assert multi.heavy_digits(h, 8, 3) == []
    h[7] = 100000
    assert multi.heavy_digits(h, 8, 3) == [7]
    assert multi.heavy_digits(h, 8, 0) == []          # no lower byte
    assert multi.heavy_digits(h, 1, 3) == []
    h[9] = 90000
    assert multi.heavy_digits(h, 2, 3) == []          # 47 % and 53 %: each below 1.25 fair shares of two
    assert multi.heavy_digits(h, 4, 3) == [7, 9]


def test_choose_chunks():
    rng = np.random.default_rng(5)
    h = rng.integers(0, 1000, 256).astype(np.uint64)
    h[40:60] = 0
    for world in (1, 2, 3, 8):
        lut = multi.choose_splitters(h, world)
        for chunks in (1, 4, 8):
            c = multi.choose_chunks(h, lut, world, chunks)
            assert c.min() >= 0 and c.max() < chunks
            for d in range(world):
                dig = np.nonzero(lut == d)[0]
                if dig.size:
                    assert np.all(np.diff(c[dig]) >= 0)          # sub-ranges are contiguous and in digit order


def test_count_matrix():
    rng = np.random.default_rng(11)
    hists = rng.integers(0, 1000, (3, 256)).astype(np.uint64)
    lut = multi.choose_splitters(hists.sum(axis=0), 3)
    m = multi.count_matrix(hists, lut, 3)
    assert m.shape == (3, 3) and np.array_equal(m.sum(axis=1), hists.sum(axis=1))
    for s in range(3):
        for d in range(3):
            assert m[s, d] == hists[s][lut == d].sum()


def test_choose_splitters_properties():
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for hist in (rng.integers(0, 1000, 256), np.r_[np.zeros(255, np.int64), [5000]],
                     np.r_[[10 ** 9], rng.integers(0, 10, 255)], np.zeros(256, np.int64)):
            lut = multi.choose_splitters(hist.astype(np.uint64), world)
            assert lut.dtype == np.uint8 and lut.shape == (256,)
            assert np.all(np.diff(lut.astype(np.int64)) >= 0)         # contiguous, monotone digit ranges
            assert lut.max() < world
    # uniform histogram -> equal shares
    lut = multi.choose_splitters(np.full(256, 100, dtype=np.uint64), 8)
    assert np.array_equal(np.bincount(lut, minlength=8), np.full(8, 32))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher: the GPU-free parent starts N ranks through
    torch.distributed.run, relays their output and prints rank 0's JSON line LAST; a failing rank fails the parent."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fake_bench.py"
    # every rank prints 100 lines of noise before and after rank 0's result, unsynchronised, on the pipe they share; the
    # result itself is longer than PIPE_BUF (as the real line is) so a torn or glued line would show
    script.write_text(
        "import os, sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "r = int(os.environ['RANK']); w = int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "for i in range(100):\n"
        "    print('noise from rank %%d line %%d' %% (r, i), flush=True)\n"
        "bench.emit_result({'metric': 'm', 'value': 1.5, 'n_gpus': w, 'argv': sys.argv[1:], 'pad': 'x' * 6000}, r)\n"
        "for i in range(100):\n"
        "    sys.stdout.write('late noise from rank %%d line %%d' %% (r, i) + ('\\n' if i %% 3 else ''))\n"
        "    sys.stdout.flush()\n"
        "sys.exit(3 if '--fail' in sys.argv and r == 1 else 0)\n" % root)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = ("import sys; sys.path.insert(0, %r); import bench; bench.launch_ranks(sys.argv[1:], 2, script=%r)" % (root, str(script)))
    out = subprocess.run([sys.executable, "-c", driver, "--gpus", "2", "--steps", "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    import json
    last = json.loads(lines[-1])
    assert out.stdout.endswith(lines[-1] + "\n")
    assert last["n_gpus"] == 2 and last["argv"] == ["--gpus", "2", "--steps", "3"] and last["pad"] == "x" * 6000
    assert any("noise from rank 1" in l for l in lines[:-1])
    assert sum('"metric"' in l for l in lines) == 1
    bad = subprocess.run([sys.executable, "-c", driver, "--fail"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


def _safe_worker(rank, world, port, n_per_rank, seed, outdir, latch):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if not latch:
        os.environ["RSX_MULTI_SAFE"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = n_per_rank[rank]
        first = sum(n_per_rank[:rank])
        whole = ol.splitmix_fill(sum(n_per_rank), ol.U32, seed, 0xFFFFFFFF)
        shard = torch.from_numpy(whole[first:first + n].view(np.int32).copy())
        if latch:      # what a failure of the chunk pipeline leaves behind: every later sort of the process is the safe one
            multi._SAFE["latched"], multi._SAFE["why"] = True, "test"
        res, stats = multi.distributed_sort(shard, OracleEngine(ol.U32, 0), chunks=4, split_slices=2)
        assert stats["safe_mode"] and stats["chunks"] == 1, stats
        np.save(os.path.join(outdir, "out%d.npy" % rank), res.numpy().view(np.uint32).copy())
    finally:
        multi._SAFE["latched"], multi._SAFE["why"] = False, None
        os.environ.pop("RSX_MULTI_SAFE", None)
        dist.destroy_process_group()


@pytest.mark.parametrize("latch", [False, True], ids=["RSX_MULTI_SAFE=1", "latched by a failure"])
def test_safe_mode_is_one_exchange_and_one_sort(tmp_path, latch):
    """RSX_MULTI_SAFE=1 (or the latch a failed first contact sets): whatever chunks / slices were asked for, the exchange is one
    all_to_all_single and the local sort one sort -- and the result is the same (README.md:647-650)."""
    world, n_per_rank, seed = 3, [30000, 30111, 29950], 55
    port = _free_port()
    mp.spawn(_safe_worker, args=(world, port, n_per_rank, seed, str(tmp_path), latch), nprocs=world, join=True)
    want, _, _ = ol.oracle_sort(ol.splitmix_fill(sum(n_per_rank), ol.U32, seed, 0xFFFFFFFF), ol.U32, 0)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "out%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got, want)
