"""Device steps of the multi-GPU sort (HipEngine) on one MI355X.

The driver runs the 8-GPU job itself; here the G ranks of a job are played one after the other on a
single GPU, with the all-to-all-v replaced by slicing, so that the histogram / MSD-partition / local
sort kernels and the splitter logic are checked bit for bit against the oracle.
"""
import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa
from radix_sorting_amd import multi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

_CARRIER = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


def to_dev(bits):
    a = np.ascontiguousarray(bits)
    return torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()


@pytest.mark.parametrize("dt,order", [(ol.U32, 0), (ol.F32, 1), (ol.U64, 0), (ol.I16, 0), (ol.U8, 0)])
def test_top_histogram_and_partition(dt, order):
    n = 1500001
    a = ol.splitmix_fill(n, dt, 17 + dt)
    eng = multi.HipEngine(dt, order)
    shard = to_dev(a)
    hist = eng.top_histogram(shard).cpu().numpy()
    k = ol.kdf_keys(a, dt, order)
    top = (k >> ol.NP_BITS[dt](8 * (ol.DTYPE_SIZE[dt] - 1))).astype(np.int64)
    assert np.array_equal(hist, np.bincount(top, minlength=256))
    for world in (2, 8, 256):
        lut = multi.choose_splitters(hist.astype(np.uint64), world) if world < 256 else np.arange(256, dtype=np.uint8)
        out = torch.zeros_like(shard)
        counts = eng.partition(shard, out, lut, world, hist.astype(np.uint64))
        torch.cuda.synchronize()
        dest = lut[top]
        assert np.array_equal(counts, np.bincount(dest, minlength=world).astype(np.uint64))
        want = a[np.argsort(dest, kind="stable")]                     # stable partition by destination
        assert np.array_equal(out.cpu().numpy().view(ol.NP_BITS[dt]), want)


def test_partition_rejects_a_wrong_histogram():
    a = ol.splitmix_fill(100000, ol.U32, 5)
    eng = multi.HipEngine(ol.U32)
    shard = to_dev(a)
    hist = eng.top_histogram(shard).cpu().numpy().astype(np.uint64)
    hist[3] += 1
    hist[200] -= 1
    with pytest.raises(rsa.RsxError, match="disagrees"):
        eng.partition(shard, torch.zeros_like(shard), np.arange(256, dtype=np.uint8) // 64, 4, hist)


@pytest.mark.parametrize("world,dt,mask", [(4, ol.U32, 0xFFFFFFFF), (8, ol.F32, 0xFFFFFFFF), (3, ol.U32, 0x00FFFFFF),
                                             (8, ol.U64, 0xFFFFFFFFFFFFFFFF)])
def test_simulated_ranks_match_single_sort(world, dt, mask):
    """Play `world` ranks on one GPU: histogram -> all-reduce (sum) -> splitters -> partition -> exchange -> local sort."""
    n_per_rank = [200000 + 1000 * r for r in range(world)]
    whole = ol.splitmix_fill(sum(n_per_rank), dt, 23, mask)
    eng = multi.HipEngine(dt)
    shards, first = [], 0
    for r in range(world):
        shards.append(to_dev(whole[first:first + n_per_rank[r]]))
        first += n_per_rank[r]
    hists = [eng.top_histogram(s).cpu().numpy().astype(np.uint64) for s in shards]
    lut = multi.choose_splitters(sum(hists), world)
    parts, counts = [], []
    for r in range(world):
        out = torch.zeros_like(shards[r])
        counts.append(eng.partition(shards[r], out, lut, world, hists[r]))
        parts.append(out)
    torch.cuda.synchronize()
    results = []
    for dst in range(world):                       # what the all-to-all-v delivers to rank dst, in source order
        pieces = []
        for src in range(world):
            off = int(counts[src][:dst].sum())
            pieces.append(parts[src][off:off + int(counts[src][dst])])
        recv = torch.cat(pieces) if pieces else shards[0][:0]
        aux = torch.zeros_like(recv)
        res, info = eng.local_sort(recv, aux)
        torch.cuda.synchronize()
        results.append(res.cpu().numpy().view(ol.NP_BITS[dt]))
    want, _, _ = ol.oracle_sort(whole, dt)
    assert np.array_equal(np.concatenate(results), want)


def test_distributed_sort_single_process():
    """world == 1 path of multi.distributed_sort (what `bench.py --gpus 1` does not use, kept honest anyway)."""
    a = ol.splitmix_fill(300000, ol.U32, 4)
    res, stats = multi.distributed_sort(to_dev(a), multi.HipEngine(ol.U32))
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy().view(np.uint32), np.sort(a))
