"""Multi-GPU sort: one MSD-digit split, an all-to-all-v bucket exchange, local LSD sorts.

The reference is single-threaded; its README suggests exactly this hybrid for
parallelism ("one MSB pass then LSB sort the sub-results", README.md:647-650).
SURVEY.md section 8e is the contract implemented here, one process per GPU:

  1. every rank runs ONE stable scatter pass by the top KDF byte of its shard
     (rsx_msd_split_device) and gets the 256 counts of that byte;
  2. the counts of all ranks are all-gathered (G x 256 numbers) and every rank
     derives the same splitters -- contiguous top-digit ranges holding ~n/G keys
     each -- and the whole G x G count matrix (no second count exchange);
  3. a destination's keys are a contiguous range of the split shard, so the
     buckets go out as they lie with ``all_to_all_single`` (RCCL all-to-all-v
     over xGMI; each directed pair of GPUs has its own link);
  4. each rank LSD-sorts what it received (rsx_sort_device).

Splitting by the byte itself (256 digits) and not by destination (G buckets)
keeps the pass on the plain-digit kernel: with G = 2..8 buckets every LDS counter
would be hit by 8..32 lanes of each instruction, which serialises them.

Receive order is by source rank and the split is stable, so with shards held in
global index order the keys of one received bucket range are ordered by (source
rank, top byte, index) -- equal keys by (source rank, index) -- and the
concatenation of the ranks' stable local sorts is the stable sort of the
concatenated input: bit-identical to the reference run on the whole array.

The device work sits behind a small engine object so that the host logic
(splitters, count matrix, collectives) can be exercised on CPU with the gloo
backend in tests (tests/ inject an oracle-backed engine); the product engine is
HipEngine and it fails loudly without a GPU.
"""
import ctypes as C

import numpy as np

from . import (ASCENDING, DTYPE_SIZE, Info, RsxError, _stream_ptr, check, lib, radix_sort)


def choose_splitters(top_hist, world):
    """Map each of the 256 top digits to a destination rank.

    Contiguous, monotone ranges; digit d goes to the rank whose ideal share
    [r*total/world, (r+1)*total/world) contains the midpoint of d's run.  Every
    rank computes this from the same all-reduced histogram, so the result is identical everywhere.
    """
    h = np.asarray(top_hist, dtype=np.uint64).astype(np.float64)
    total = float(h.sum())
    lut = np.zeros(256, dtype=np.uint8)
    if total == 0 or world == 1:
        return lut
    before = np.concatenate([[0.0], np.cumsum(h)[:-1]])
    mid = before + h / 2.0
    r = np.floor(mid * world / total).astype(np.int64)
    r = np.clip(r, 0, world - 1)
    r = np.maximum.accumulate(r)          # monotone even with empty digits
    return r.astype(np.uint8)


def count_matrix(hists, lut, world):
    """matrix[s, d] = number of keys of rank s whose top digit belongs to rank d (hists: world x 256 counts)."""
    m = np.zeros((world, world), dtype=np.uint64)
    lut = np.asarray(lut, dtype=np.int64)
    for d in range(world):
        m[:, d] = np.asarray(hists, dtype=np.uint64)[:, lut == d].sum(axis=1)
    return m


class HipEngine:
    """Device steps of the distributed sort through librsx.so (the product path)."""

    def __init__(self, dtype, order=ASCENDING):
        import torch
        self.torch = torch
        self.dtype = dtype
        self.order = order
        self.kb = DTYPE_SIZE[dtype]
        self._hist = None
        self._flag = None

    def top_histogram(self, shard):
        torch = self.torch
        if self._hist is None:
            self._hist = torch.zeros(256 * self.kb, dtype=torch.int64, device=shard.device)
            self._flag = torch.zeros(1, dtype=torch.int32, device=shard.device)
        check(lib().rsx_histogram_device(shard.data_ptr(), shard.numel(), self.dtype, self.order,
                                         self._hist.data_ptr(), self._flag.data_ptr(), _stream_ptr()))
        return self._hist[256 * (self.kb - 1):].clone()      # counts of the top KDF byte, on device

    def partition(self, shard, out, lut, world, top_hist_host):
        counts = np.zeros(world, dtype=np.uint64)
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        th = np.ascontiguousarray(top_hist_host, dtype=np.uint64)
        check(lib().rsx_partition_device(shard.data_ptr(), out.data_ptr(), shard.numel(), self.dtype, self.order,
                                         lut.ctypes.data, world, th.ctypes.data, counts.ctypes.data, _stream_ptr()))
        return counts

    def msd_split(self, shard, out, column=-1):
        """out = shard in stable order of KDF byte `column` (-1: the top one), enqueued; returns the byte's 256 counts (host)."""
        hist = np.zeros(256, dtype=np.uint64)
        check(lib().rsx_msd_split_device(shard.data_ptr(), out.data_ptr(), shard.numel(), self.dtype, self.order, column,
                                         hist.ctypes.data, _stream_ptr()))
        return hist

    def local_sort(self, keys, aux):
        res, info = radix_sort(keys, aux, dtype=self.dtype, order=self.order)
        return res, info

    def empty(self, n, like):
        return self.torch.empty(n, dtype=like.dtype, device=like.device)


def distributed_sort(shard, engine, group=None, recv_capacity=None, scratch=None, force_exchange=False):
    """Sort the concatenation of every rank's ``shard`` (rank order = global index order).

    Returns (sorted_local, stats): rank r ends up with the r-th contiguous slice of
    the globally sorted sequence (slice sizes follow the splitters, not n/G).
    ``scratch`` may carry preallocated tensors {"part", "recv", "aux"} (>= capacity) to keep
    allocation out of a timed region.  ``force_exchange`` runs every exchange step even in a one-rank group (the
    collectives then talk to the rank itself): that is how the RCCL path is exercised on a one-GPU box.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = shard.numel()
    if world == 1 and not (force_exchange and dist.is_initialized()):
        aux = scratch["aux"][:n] if scratch else engine.empty(n, shard)
        res, info = engine.local_sort(shard, aux)
        return res, {"sent": 0, "received": n, "local_info": info}

    # 1-2: one stable pass by the top KDF byte and its counts; everybody's counts -> identical splitters and the whole
    # count matrix on every rank.  A byte that is constant over ALL ranks would send every key to one rank: split by the
    # next byte down instead (every rank sees the same gathered counts and takes the same decision).
    part = scratch["part"][:n] if scratch else engine.empty(n, shard)
    column = engine.kb - 1
    while True:
        local_hist = engine.msd_split(shard, part, column)
        mine = torch.from_numpy(local_hist.astype(np.int64)).to(shard.device)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        hists = torch.stack(gathered).cpu().numpy().astype(np.uint64)
        if column == 0 or np.count_nonzero(hists.sum(axis=0)) > 1:
            break
        column -= 1
    lut = choose_splitters(hists.sum(axis=0), world)
    matrix = count_matrix(hists, lut, world)          # matrix[s, d]: keys rank s sends to rank d
    send_counts, recv_counts = matrix[rank], matrix[:, rank]
    n_recv = int(recv_counts.sum())
    if scratch:
        if n_recv > scratch["recv"].numel():
            raise RsxError("rank %d receives %d keys, more than the %d-key receive buffer" %
                           (rank, n_recv, scratch["recv"].numel()))
        recv = scratch["recv"][:n_recv]
        aux = scratch["aux"][:n_recv]
    else:
        recv = engine.empty(n_recv, shard)
        aux = engine.empty(n_recv, shard)

    # 3: the buckets (all-to-all-v).  The exchange moves opaque bytes: every key width then works on every backend
    # (gloo has no int16)
    es = shard.element_size()
    dist.all_to_all_single(recv.view(torch.uint8), part.view(torch.uint8),
                           output_split_sizes=[int(x) * es for x in recv_counts],
                           input_split_sizes=[int(x) * es for x in send_counts], group=group)

    # 4: local LSD sort of the received bucket range
    res, info = engine.local_sort(recv, aux)
    sent = int(send_counts.sum() - send_counts[rank])
    return res, {"sent": sent, "received": n_recv, "local_info": info, "lut": lut, "split_column": column,
                 "send_counts": send_counts, "recv_counts": recv_counts}
