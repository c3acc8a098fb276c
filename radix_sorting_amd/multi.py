"""Multi-GPU sort: one MSD-digit split, an all-to-all-v bucket exchange, local LSD sorts.

The reference is single-threaded; its README suggests exactly this hybrid for
parallelism ("one MSB pass then LSB sort the sub-results", README.md:647-650).
SURVEY.md section 8e is the contract implemented here, one process per GPU:

  1. every rank counts ALL byte columns of its shard in one read (rsx_histogram_device; the counts stay on the
     device) and the counts of all ranks are all-gathered once (G x key bytes x 256 numbers): every rank then knows
     which byte to split by -- the highest one that varies over all ranks -- without a trial pass;
  2. every rank runs ONE stable scatter pass by that byte (rsx_msd_split_async: nothing is counted again, nothing
     waits for the host) and derives, from the same gathered counts, the same splitters -- contiguous digit ranges
     holding ~n/G keys each -- and the whole G x G count matrix (no second count exchange);
  3. a destination's keys are a contiguous range of the split shard, so the
     buckets go out as they lie (RCCL all-to-all-v over xGMI; each directed pair of GPUs has its own link) -- as ONE
     all_to_all_single by default, or (RSX_MULTI_CHUNKS > 1, grouped send/recv) in CHUNKS: a destination's digit range
     is cut into sub-ranges, which are independent sorting problems, and sub-range j is sorted (step 4) while
     sub-range j+1 is still on the links;
  4. each rank sorts what it received (every sub-range of it), in place and without a host synchronisation
     (rsx_sort_inplace_async_hint: the gathered counts say whether the received digits are even, which a sample of keys that
     arrive in digit order, piece by piece, cannot see).  Sub-ranges are in digit order, so the receive buffer ends up
     sorted as a whole.
  (split_slices > 1: step 2's pass runs in consecutive parts of the shard and the first sub-range's pieces of part 0 are
  on the links while the other parts are split; see split_plan.)

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
import os

import numpy as np

from . import (ASCENDING, DTYPE_SIZE, Info, RsxError, _stream_ptr, check, lib, radix_sort, radix_sort_inplace_async,
               spin)


def choose_splitters(top_hist, world):
    """Map each bin (a top digit, or a (top digit, next digit) pair where a digit was refined: see split_plan) to a
    destination rank.

    Contiguous, monotone ranges; bin b goes to the rank whose ideal share
    [r*total/world, (r+1)*total/world) contains the midpoint of b's run.  Every
    rank computes this from the same gathered histogram, so the result is identical everywhere.
    """
    h = np.asarray(top_hist, dtype=np.uint64).astype(np.float64)
    total = float(h.sum())
    lut = np.zeros(h.size, dtype=np.uint8 if world <= 256 else np.int64)
    if total == 0 or world == 1:
        return lut
    before = np.concatenate([[0.0], np.cumsum(h)[:-1]])
    mid = before + h / 2.0
    r = np.floor(mid * world / total).astype(np.int64)
    r = np.clip(r, 0, world - 1)
    r = np.maximum.accumulate(r)          # monotone even with empty bins
    return r.astype(lut.dtype)


_OVERLAP_STREAMS = {}      # (device, group, to_self) -> stream or None (HipEngine.overlap_stream)
# RSX_MULTI_SAFE=1 -- and, once set by a failure, for the rest of the process --: the exchange is ONE all_to_all_single and ONE
# local sort, no sub-ranges, no grouped send/recv, no second stream, no calibration.  Nothing of the chunk pipeline below has
# run on more than one physical GPU yet; this is what a first contact with eight ranks falls back to.
_SAFE = {"latched": False, "why": None}


def safe_mode():
    return os.environ.get("RSX_MULTI_SAFE", "") == "1" or _SAFE["latched"]


def _agree_failed(failed, group, device):
    """Every rank learns whether ANY rank failed (one small all-reduce): the ranks must leave the pipeline together."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def count_matrix(hists, lut, world):
    """matrix[s, d] = number of keys of rank s whose bin belongs to rank d (hists: world x bins counts)."""
    m = np.zeros((world, world), dtype=np.uint64)
    lut = np.asarray(lut, dtype=np.int64)
    for d in range(world):
        m[:, d] = np.asarray(hists, dtype=np.uint64)[:, lut == d].sum(axis=1)
    return m


HEAVY_FACTOR = 1.25      # a digit holding more than this many fair shares (total / world) is split by the next byte too


def heavy_digits(global_hist, world, column):
    """Digits of the split byte that cannot be balanced as a whole (SURVEY.md section 7 "Skew in the exchange"; the
    one-line spec is README.md:647-650): more than HEAVY_FACTOR * total / world keys, and a lower byte to refine by."""
    h = np.asarray(global_hist, dtype=np.uint64)
    total = int(h.sum())
    if column == 0 or world < 2 or total == 0:
        return []
    return [int(d) for d in np.nonzero(h.astype(np.float64) > HEAVY_FACTOR * total / world)[0]]


def heavy_bins(share, world, bytes_left):
    """Indices of the bins (global counts `share`) that hold more than HEAVY_FACTOR fair shares, while there is a lower byte to
    refine them by.  split_plan asks this once per refinement level."""
    share = np.asarray(share, dtype=np.float64)
    total = float(share.sum())
    if bytes_left <= 0 or world < 2 or total == 0:
        return []
    return [int(i) for i in np.nonzero(share > HEAVY_FACTOR * total / world)[0]]


class HipEngine:
    """Device steps of the distributed sort through librsx.so (the product path)."""

    def __init__(self, dtype, order=ASCENDING):
        import torch
        self.torch = torch
        self.dtype = dtype
        self.order = order
        self.kb = DTYPE_SIZE[dtype]

        self.split_passes = 0      # stable passes over a whole shard or a run of it (tests count them)

    takes_even_hint = True     # sort_inplace_async(buf, scratch, even=...)

    def msd_split(self, shard, out, column=-1):
        """out = shard in stable order of KDF byte `column` (-1: the top one), enqueued; returns the byte's 256 counts (host)."""
        hist = np.zeros(256, dtype=np.uint64)
        check(lib().rsx_msd_split_device(shard.data_ptr(), out.data_ptr(), shard.numel(), self.dtype, self.order, column,
                                         hist.ctypes.data, _stream_ptr()))
        self.split_passes += 1
        return hist

    def histogram(self, shard):
        """Counts of every KDF byte column of the shard: an int64 tensor [key bytes * 256] on the shard's device, only
        enqueued (rsx_histogram_device; column c's 256 counts are [256 c, 256 c + 256))."""
        torch = self.torch
        h = torch.empty(self.kb * 256 + 1, dtype=torch.int64, device=shard.device)     # (+ the pre-sorted flag's word)
        check(lib().rsx_histogram_device(shard.data_ptr(), shard.numel(), self.dtype, self.order, h.data_ptr(),
                                         h[self.kb * 256:].data_ptr(), _stream_ptr()))
        return h[:self.kb * 256]

    def msd_split_known(self, shard, out, column, hist_all, hot=False):
        """msd_split for a caller that holds the shard's counts (histogram()): no second count, no host synchronisation.
        hot: one digit of the column holds an eighth of the shard or more (RSX_SPLIT_HOT: the ballot-ranked pass)."""
        check(lib().rsx_msd_split_async(shard.data_ptr(), out.data_ptr(), shard.numel(), self.dtype, self.order,
                                        column | (0x100 if hot else 0), hist_all.data_ptr(), _stream_ptr()))
        self.split_passes += 1

    def local_sort(self, keys, aux):
        res, info = radix_sort(keys, aux, dtype=self.dtype, order=self.order)
        return res, info

    def sort_inplace_async(self, buf, scratch, even=False):
        """Stable sort of buf in place (scratch: as many elements), only enqueued on the current stream.
        even: the caller has counted the keys of buf by their top varying byte and found no digit with twice its share
        (rsx_sort_inplace_async_hint: what arrives here was put in order of that byte by the senders' split passes, piece by
        piece, and would look clustered to the sample of a sort without a histogram)."""
        radix_sort_inplace_async(buf, scratch, dtype=self.dtype, order=self.order, hints=1 if even else 0)

    def overlap_stream(self, group=None, to_self=False):
        """A stream whose kernels run concurrently with the process group's send/recv kernels, or None.

        The HIP runtime maps streams onto a few hardware queues and kernels of two streams that share a queue run one after
        the other -- measured on MI355X: with the sorts on the default stream the RCCL kernels of the next sub-range ran
        strictly before them.  So: four candidate streams, each tried against one dummy exchange (every rank does all four
        rounds: the rounds are collective); the first whose marker kernel finishes well before the exchange wins.  Cached.
        """
        torch = self.torch
        import torch.distributed as dist
        main = torch.cuda.current_stream()
        dev = main.device
        key = (dev.index, id(group), bool(to_self))
        if key in _OVERLAP_STREAMS:
            return _OVERLAP_STREAMS[key]
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        nb = 1 << 20
        sendb = torch.zeros(nb * world, dtype=torch.uint8, device=dev)
        recvb = torch.zeros(nb * world, dtype=torch.uint8, device=dev)
        tiny = torch.zeros(256, dtype=torch.float32, device=dev)
        peers = [p for p in range(world) if p != rank] or ([rank] if to_self else [])
        chosen = None
        hold_us = 20000
        for cand in [torch.cuda.Stream(device=dev) for _ in range(4)]:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            # the exchange waits for the main stream, which spins for 20 ms: whatever shares RCCL's hardware queue is held
            # up that long, whatever does not finishes at once
            spin(hold_us)
            ops = []
            for p in peers:
                ops.append(dist.P2POp(dist.isend, sendb[p * nb:(p + 1) * nb], p, group))
                ops.append(dist.P2POp(dist.irecv, recvb[p * nb:(p + 1) * nb], p, group))
            works = dist.batch_isend_irecv(ops) if ops else []
            with torch.cuda.stream(cand):
                tiny.add_(1.0)
                e1.record(cand)
            for w in works:
                w.wait()
            torch.cuda.synchronize()
            if chosen is None and works and e0.elapsed_time(e1) < hold_us / 2000.0:
                chosen = cand
        _OVERLAP_STREAMS[key] = chosen
        return chosen

    def empty(self, n, like):
        return self.torch.empty(n, dtype=like.dtype, device=like.device)


def choose_chunks(global_hist, lut, world, chunks):
    """Cut every destination's bin range into at most `chunks` contiguous sub-ranges of about equal global counts.

    Returns chunk_of[bins] (monotone inside a destination's range).  Keys of different sub-ranges never compare equal in
    the split byte(s), so the sub-ranges of a destination are independent sorting problems in key order.
    """
    h = np.asarray(global_hist, dtype=np.uint64)
    chunk_of = np.zeros(h.size, dtype=np.int64)
    lut = np.asarray(lut, dtype=np.int64)
    for d in range(world):
        bins = np.nonzero(lut == d)[0]
        if bins.size == 0:
            continue
        sub = np.zeros(h.size, dtype=np.uint64)
        sub[bins] = h[bins]
        chunk_of[bins] = choose_splitters(sub, chunks)[bins]
    return chunk_of


def slice_bounds(n, slices):
    """[a_0 = 0, a_1, ..., a_slices = n]: the shard in `slices` consecutive parts that start on multiples of 4096 elements
    (16-byte loads stay aligned); a short shard leaves the later parts empty -- every rank has the same NUMBER of parts."""
    b = [min(n, ((n * s // slices + 4095) // 4096) * 4096) for s in range(slices)] + [n]
    b[0] = 0
    return b


def _is_hot(counts):
    """Plan::hot's rule (rsx_plan_kernel) on a column's 256 counts: one digit with an eighth of the keys or more."""
    total = int(counts.sum())
    return total > 0 and int(counts.max()) >= total // 8 + 1


def split_plan(shard, part, engine, group, world, tmp=None, slices=1):
    """Steps 1-2 of the distributed sort: split the shard into BINS in key order and gather every rank's bin counts.

    A bin is a digit of the split byte -- the highest byte that varies over all ranks -- or, for a bin that holds more
    than HEAVY_FACTOR fair shares of all the keys (a dominant digit cannot be divided among ranks as a whole), the 256
    bins of its keys' next lower byte, and so on down the bytes while a bin stays too heavy: such a bin's run of `part`
    gets another stable pass by the next byte.
    Returns (counts[world, bins] uint64, column, heavy, levels): `part` holds the shard ordered by bin, every rank has the
    same bins in the same order; `heavy` = the digits of the split byte that were refined, `levels` = how many bytes
    deep the refinement went.  `tmp`: a scratch tensor for the refinement passes (allocated when absent).

    slices > 1 (and no bin too heavy): the shard is split in `slices` consecutive parts, each into its own part of
    `part` -- part 0 here, the others by the closures returned as sliced["pending"], which the caller enqueues AFTER it has
    put part 0's first pieces on the links: the split of the rest then runs under that exchange.  A fifth value is
    returned: None, or {"bounds", "hists": [world, slices, bins] counts, "pending"}.
    """
    import torch
    import torch.distributed as dist

    def gather(local):
        """all-gather one int64 tensor per rank (it stays where it is until the single copy to the host)"""
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local, group=group)
        return torch.stack(gathered).cpu().numpy().astype(np.uint64)

    # every byte column's counts in ONE read of the shard, one all-gather: the byte to split by is the highest one that
    # varies over ALL ranks (a byte that is constant everywhere would send every key to one rank) -- every rank sees the
    # same gathered counts and decides the same, without a trial pass per constant byte
    import time
    marks = getattr(engine, "host_marks", None)      # (bench.py: where the host's share of the step goes)
    t0 = time.perf_counter()
    slices = max(1, int(slices))
    bounds = slice_bounds(shard.numel(), slices)
    halls = [engine.histogram(shard[bounds[i]:bounds[i + 1]]) if bounds[i + 1] > bounds[i]
             else torch.zeros(engine.kb * 256, dtype=torch.int64, device=shard.device) for i in range(slices)]
    t1 = time.perf_counter()
    every_s = gather(torch.cat(halls) if slices > 1 else halls[0]).reshape(world, slices, engine.kb, 256)
    t2 = time.perf_counter()
    if marks is not None:
        marks["histogram_enqueue_ms"] = (t1 - t0) * 1e3
        marks["gather_counts_ms"] = (t2 - t1) * 1e3      # (waits for the histogram kernel: one read of the shard)
    every = every_s.sum(axis=1)
    column = 0
    for c in range(engine.kb - 1, -1, -1):
        if np.count_nonzero(every[:, c, :].sum(axis=0)) > 1:
            column = c
            break
    hists = every[:, column, :]                     # [world, 256]
    rank = dist.get_rank(group)
    heavy = heavy_bins(hists.sum(axis=0), world, column)
    if slices > 1 and not heavy:
        # part by part; nothing is refined (a heavy bin's run must be contiguous in `part`), so this is the whole plan
        def split_one(i):
            if bounds[i + 1] > bounds[i]:
                engine.msd_split_known(shard[bounds[i]:bounds[i + 1]], part[bounds[i]:bounds[i + 1]], column, halls[i],
                                       hot=_is_hot(every_s[rank, i, column, :]))
        split_one(0)
        pending = [lambda i=i: split_one(i) for i in range(1, slices)]
        return hists, column, heavy, 0, {"bounds": bounds, "hists": every_s[:, :, column, :], "pending": pending}
    hall = halls[0]
    for h in halls[1:]:
        hall = hall + h
    engine.msd_split_known(shard, part, column, hall, hot=_is_hot(hists[rank]))
    levels = 0
    # refinement, level by level: bins that are still too heavy are split by the next lower byte (their runs of `part`
    # are contiguous); one all-gather of the sub-counts per level
    while column - levels > 0:
        too = heavy_bins(hists.sum(axis=0), world, column - levels)
        if not too:
            break
        levels += 1
        local = hists[rank].astype(np.int64)
        first = np.concatenate([[0], np.cumsum(local)])
        sub_local = np.zeros((len(too), 256), dtype=np.int64)
        for i, b in enumerate(too):
            a, e = int(first[b]), int(first[b + 1])
            if e > a:
                run = part[a:e]
                t = tmp[:e - a] if tmp is not None and tmp.numel() >= e - a else engine.empty(e - a, shard)
                sub_local[i] = engine.msd_split(run, t, column - levels).astype(np.int64)
                run.copy_(t)
        sub_all = gather(torch.from_numpy(sub_local.reshape(-1)).to(shard.device)).reshape(world, len(too), 256)
        cols, k = [], 0
        for b in range(hists.shape[1]):
            if k < len(too) and too[k] == b:
                cols.append(sub_all[:, k, :])
                k += 1
            else:
                cols.append(hists[:, b:b + 1])
        hists = np.concatenate(cols, axis=1)
    return hists, column, heavy, levels, None


def distributed_sort(shard, engine, group=None, recv_capacity=None, scratch=None, force_exchange=False, chunks=None,
                     split_slices=None):
    """Sort the concatenation of every rank's ``shard`` (rank order = global index order).

    Returns (sorted_local, stats): rank r ends up with the r-th contiguous slice of
    the globally sorted sequence (slice sizes follow the splitters, not n/G).
    ``scratch`` may carry preallocated tensors {"part", "recv", "aux"} (>= capacity) to keep
    allocation out of a timed region.  ``force_exchange`` runs every exchange step even in a one-rank group (the
    rank then sends to itself): that is how the RCCL path is exercised on a one-GPU box.  ``chunks``: sub-ranges a
    destination's digit range is cut into; the exchange of sub-range j+1 overlaps the local sort of sub-range j
    (default: RSX_MULTI_CHUNKS or 1 = one ``all_to_all_single`` and one local sort, nothing overlapped.  Round 6, one rank sending
    itself 2^29 keys: 6.75 ms per step with 1 sub-range, 8.3 with 2, 8.45 with 4 -- a sub-range's sort and grouped send / receive
    cost more than the whole range's, and what the overlap wins on real links is not measured: the pipeline is opt-in).
    ``split_slices`` (default: RSX_MULTI_SPLIT_SLICES or 1): the split pass itself in that many consecutive parts of the
    shard, so that the first sub-range's pieces of part 0 are on the links while the rest of the shard is still being split
    (only with chunks > 1 and no bin heavy enough to be refined; a piece is then one run per part).
    """
    import time
    import torch
    import torch.distributed as dist

    t_enter = time.perf_counter()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = shard.numel()
    if world == 1 and not (force_exchange and dist.is_initialized()):
        aux = scratch["aux"][:n] if scratch else engine.empty(n, shard)
        res, info = engine.local_sort(shard, aux)
        return res, {"sent": 0, "received": n, "local_info": info}

    # 1-2: the shard ordered by bin (split_plan) and everybody's bin counts -> identical splitters and the whole count
    # matrix on every rank, no second count exchange.  The receive buffer is sized from the counts, never from n / G:
    # preallocated scratch is used when it is large enough and replaced when it is not.
    part = scratch["part"][:n] if scratch else engine.empty(n, shard)
    if chunks is None:
        chunks = int(os.environ.get("RSX_MULTI_CHUNKS", "1"))
    nchunks = max(1, min(int(chunks), 64))
    if split_slices is None:
        split_slices = int(os.environ.get("RSX_MULTI_SPLIT_SLICES", "1"))
    if safe_mode():
        nchunks = 1
    nslices = max(1, min(int(split_slices), 16)) if nchunks > 1 else 1
    on_gpu = shard.is_cuda
    ev = (lambda: torch.cuda.Event(enable_timing=True)) if on_gpu else (lambda: None)
    e_begin, e_split, e_xchg, e_end = ev(), ev(), ev(), ev()
    if on_gpu:
        e_begin.record()
    hists, column, heavy, levels, sliced = split_plan(shard, part, engine, group, world,
                                                      tmp=scratch.get("aux") if scratch else None, slices=nslices)
    if sliced is None:
        nslices = 1
    t_planned = time.perf_counter()     # (includes the split pass's one synchronisation and the gather of the counts)
    if on_gpu:
        e_split.record()
    local_hist = hists[rank]
    lut = choose_splitters(hists.sum(axis=0), world)
    matrix = count_matrix(hists, lut, world)          # matrix[s, d]: keys rank s sends to rank d
    send_counts, recv_counts = matrix[rank], matrix[:, rank]
    n_recv = int(recv_counts.sum())
    if scratch and n_recv <= scratch["recv"].numel() and n_recv <= scratch["aux"].numel():
        recv = scratch["recv"][:n_recv]
        aux = scratch["aux"][:n_recv]
    else:
        recv = engine.empty(n_recv, shard)
        aux = engine.empty(n_recv, shard)
        if scratch is not None:                        # keep the larger buffers for the next call
            scratch["recv"], scratch["aux"] = recv, aux

    # 3-4: sub-range by sub-range: the pieces of sub-range j go out (grouped send/recv = all-to-all-v; opaque bytes, so
    # every key width works on every backend), and as soon as they are in, they are sorted in place while the next
    # sub-range is on the links.  The split shard is in digit order: piece (destination d, sub-range j) of this rank is
    # part[first[a] : first[b]] for the digits [a, b) of that sub-range.  Receive layout: sub-range major, source minor.
    recorded = {"xchg": False}
    total_bins = hists.sum(axis=0).astype(np.float64)

    def local_sort(lo, hi, sel):
        """the received keys recv[lo:hi] -- the bins `sel` of the split, in pieces that are each in bin order -- sorted in place.
        The engine is told when the gathered counts say those bins are even (no bin with 1.5 times the mean, nothing refined):
        what a sample of the pieces cannot see (rsx_sort_inplace_async_hint)."""
        if hi - lo <= 1:
            return
        if getattr(engine, "takes_even_hint", False):
            c = total_bins[sel]
            even = levels == 0 and c.size >= 2 and float(c.max()) <= 1.5 * float(c.mean())
            engine.sort_inplace_async(recv[lo:hi], aux[lo:hi], even=even)
        else:
            engine.sort_inplace_async(recv[lo:hi], aux[lo:hi])

    def phases():
        """GPU time of the step's phases on this rank (ms; needs a synchronisation: called by bench.py after its own)"""
        if not on_gpu:
            return None
        torch.cuda.synchronize()
        out = {"split_ms": e_begin.elapsed_time(e_split), "total_ms": e_begin.elapsed_time(e_end)}
        if recorded["xchg"]:
            out["exchange_ms"] = e_split.elapsed_time(e_xchg)
            out["sort_ms"] = e_xchg.elapsed_time(e_end)
        else:
            out["exchange_and_sort_ms"] = e_split.elapsed_time(e_end)   # (the chunk pipeline overlaps the two)
        return out

    # the exchange moves opaque units of the widest integer type that divides the element size (every key width on every backend)
    es = shard.element_size()
    xdt = {8: torch.int64, 4: torch.int32}.get(es, torch.uint8)      # (NCCL has no 16-bit integer type)
    xu = es // {torch.int64: 8, torch.int32: 4, torch.uint8: 1}[xdt]      # units per element
    PIECE_LIMIT = 1 << 30      # bytes one send / receive may carry (see one_exchange)

    def one_exchange():
        # A piece of 2^31 bytes or more does not arrive whole: the one-rank forced exchange of bench.py's 2^29 four-byte keys (ONE
        # piece of 2 GiB, the rank to itself) delivered a part of it, whatever the unit of the counts, and the sort behind it put
        # out a sorted array of the wrong keys (found in round 6 by comparing with torch.sort; bench.py checks checksums since).
        # Between G >= 2 ranks a piece is at most half a shard: bench.py's sizes stay below the limit.  Above it the pieces go in
        # parts of at most 1 GiB: slices of the one piece in a one-rank group, grouped send / receive otherwise.
        big = max(int(recv_counts.max()), int(send_counts.max())) * es >= (1 << 31)
        if not big:
            dist.all_to_all_single(recv.view(xdt), part.view(xdt),
                                   output_split_sizes=[int(x) * xu for x in recv_counts],
                                   input_split_sizes=[int(x) * xu for x in send_counts], group=group)
        elif world == 1:
            step_elems = PIECE_LIMIT // es
            rv, pv = recv.view(xdt), part.view(xdt)
            for a in range(0, n_recv, step_elems):
                b = min(a + step_elems, n_recv)
                dist.all_to_all_single(rv[a * xu:b * xu], pv[a * xu:b * xu], output_split_sizes=[(b - a) * xu],
                                       input_split_sizes=[(b - a) * xu], group=group)
        else:
            rv, pv = recv.view(xdt), part.view(xdt)
            soffs = np.concatenate([[0], np.cumsum(send_counts.astype(np.int64))])
            roffs1 = np.concatenate([[0], np.cumsum(recv_counts.astype(np.int64))])
            step_elems = PIECE_LIMIT // es
            ops = []
            for p in range(world):
                for a in range(0, int(send_counts[p]), step_elems):
                    b = min(a + step_elems, int(send_counts[p]))
                    ops.append(dist.P2POp(dist.isend, pv[(int(soffs[p]) + a) * xu:(int(soffs[p]) + b) * xu], p, group))
                for a in range(0, int(recv_counts[p]), step_elems):
                    b = min(a + step_elems, int(recv_counts[p]))
                    ops.append(dist.P2POp(dist.irecv, rv[(int(roffs1[p]) + a) * xu:(int(roffs1[p]) + b) * xu], p, group))
            for w in (dist.batch_isend_irecv(ops) if ops else []):
                w.wait()
        if on_gpu:
            e_xchg.record()
            recorded["xchg"] = True
        local_sort(0, n_recv, np.nonzero(lut.astype(np.int64) == rank)[0])
        if on_gpu:
            e_end.record()

    if nchunks == 1:
        one_exchange()
        return recv, {"sent": int(send_counts.sum() - send_counts[rank]), "received": n_recv, "local_info": None, "lut": lut,
                      "safe_mode": safe_mode(), "safe_why": _SAFE["why"], "phases": phases,
                      "split_column": column, "heavy_digits": heavy, "refine_levels": levels, "chunks": 1,
                      "send_counts": send_counts, "recv_counts": recv_counts,
                      "imbalance": float(matrix.sum(axis=0).max()) * world / max(float(matrix.sum()), 1.0)}
    chunk_of = choose_chunks(hists.sum(axis=0), lut, world, nchunks)
    # per part of the split (one part unless the split pass was sliced): every rank's bin counts and, for this rank, where
    # the part and its bins' runs start in `part`.  A piece (destination, sub-range) is one run per part.
    sl_hists = sliced["hists"] if sliced else hists[:, None, :]                  # [world, parts, bins]
    sl_begin = sliced["bounds"] if sliced else [0, n]
    first = [np.concatenate([[0], np.cumsum(sl_hists[rank][i].astype(np.int64))]) + sl_begin[i] for i in range(nslices)]
    part_b, recv_b = part.view(xdt), recv.view(xdt)
    lut64 = lut.astype(np.int64)
    mine_digits = lut64 == rank
    # first contact: the calibration (a 20 ms spin and four dummy exchanges) and the grouped send/recv below are what has never
    # run between physical GPUs.  A rank on which either raises tells the others (one all-reduce) and ALL of them finish this
    # sort -- and every later one -- the safe way: `part` is intact, the single all-to-all rewrites every byte of `recv`.
    side, failed, why = None, False, None
    if hasattr(engine, "overlap_stream"):
        try:
            side = engine.overlap_stream(group, force_exchange)
        except Exception as exc:      # noqa: BLE001
            failed, why = True, "overlap_stream: %r" % (exc,)
        if on_gpu and dist.is_initialized() and _agree_failed(failed, group, shard.device):
            _SAFE["latched"], _SAFE["why"] = True, why or "another rank's calibration failed"
            one_exchange()
            return recv, {"sent": int(send_counts.sum() - send_counts[rank]), "received": n_recv, "local_info": None, "lut": lut,
                          "safe_mode": True, "safe_why": _SAFE["why"], "phases": phases, "split_column": column,
                          "heavy_digits": heavy, "refine_levels": levels, "chunks": 1, "send_counts": send_counts,
                          "recv_counts": recv_counts,
                          "imbalance": float(matrix.sum(axis=0).max()) * world / max(float(matrix.sum()), 1.0)}
    main = torch.cuda.current_stream() if side is not None else None
    # receive layout: sub-range major, then source rank, then part (= global index order inside a source rank)
    roffs = np.zeros((nchunks, world, nslices + 1), dtype=np.int64)
    acc = 0
    for j in range(nchunks):
        sel_r = np.nonzero(mine_digits & (chunk_of == j))[0]
        for p in range(world):
            for i in range(nslices):
                roffs[j, p, i] = acc
                acc += int(sl_hists[p][i][sel_r].sum()) if sel_r.size else 0
            roffs[j, p, nslices] = acc
    chunk_end = [int(roffs[j, world - 1, nslices]) for j in range(nchunks)]

    def exchange(j, parts):
        """enqueue sub-range j's pieces of the given parts of the split; returns the works"""
        ops = []
        for p in range(world):
            sel_s = np.nonzero((lut64 == p) & (chunk_of == j))[0]                 # the bins I send rank p for sub-range j
            for i in parts:
                roff = int(roffs[j, p, i])
                rcnt = int(roffs[j, p, i + 1]) - roff                               # what rank p's part i sends me
                scnt = int(sl_hists[rank][i][sel_s].sum()) if sel_s.size else 0
                soff = int(first[i][sel_s[0]]) if sel_s.size else 0
                dst = recv_b[roff * xu:(roff + rcnt) * xu]
                src = part_b[soff * xu:(soff + scnt) * xu]
                if p == rank and not force_exchange:
                    if rcnt:
                        dst.copy_(src)
                else:
                    lim = PIECE_LIMIT // es * xu      # (units one send / receive may carry: see one_exchange)
                    for a in range(0, scnt * xu, lim):
                        ops.append(dist.P2POp(dist.isend, src[a:a + lim], p, group))
                    for a in range(0, rcnt * xu, lim):
                        ops.append(dist.P2POp(dist.irecv, dst[a:a + lim], p, group))
        return dist.batch_isend_irecv(ops) if ops else []

    # sub-range j+1 is submitted before sub-range j is sorted: with the sorts on a stream of their own (another hardware
    # queue than RCCL's) the two overlap; on a shared queue the order of submission is simply the order of execution.
    # A sliced split: part 0's pieces of sub-range 0 go out first, THEN the rest of the shard is split (under them).
    try:
        works0 = exchange(0, [0])
    except Exception as exc:      # noqa: BLE001  (the first grouped send/recv: nothing has been waited for yet)
        _SAFE["latched"], _SAFE["why"] = True, "batch_isend_irecv: %r" % (exc,)
        raise RuntimeError("the grouped send/recv of the chunk pipeline failed (%r); RSX_MULTI_SAFE is now latched for this process: "
                           "the next distributed_sort uses one all_to_all_single" % (exc,)) from exc
    if sliced:
        for fn in sliced["pending"]:
            fn()
        works0 = list(works0) + list(exchange(0, list(range(1, nslices))))
    nxt = (works0, 0, chunk_end[0])
    for j in range(nchunks):
        works, begin, end = nxt
        if j + 1 < nchunks:
            nxt = (exchange(j + 1, list(range(nslices))), end, chunk_end[j + 1])
        if side is not None:
            side.wait_stream(main)                     # this rank's own piece was copied on the main stream
            with torch.cuda.stream(side):
                for w in works:
                    w.wait()
                local_sort(begin, end, np.nonzero(mine_digits & (chunk_of == j))[0])
        else:
            for w in works:
                w.wait()                               # (NCCL: the current stream waits; gloo: the host does)
            local_sort(begin, end, np.nonzero(mine_digits & (chunk_of == j))[0])
    if side is not None:
        main.wait_stream(side)
    if on_gpu:
        e_end.record()
    sent = int(send_counts.sum() - send_counts[rank])
    t_submitted = time.perf_counter()
    return recv, {"host_ms_split_and_counts": (t_planned - t_enter) * 1e3, "host_ms_submit_exchange_and_sorts": (t_submitted - t_planned) * 1e3,
                  "sent": sent, "received": n_recv, "local_info": None, "lut": lut, "split_column": column,
                  "heavy_digits": heavy, "refine_levels": levels, "chunks": nchunks, "split_slices": nslices,
                  "overlap_stream": side is not None, "safe_mode": False, "safe_why": None, "phases": phases,
                  "send_counts": send_counts, "recv_counts": recv_counts,
                  # the largest rank's share of the keys over the fair share (1.0 = balanced): what is left of the skew
                  "imbalance": float(matrix.sum(axis=0).max()) * world / max(float(matrix.sum()), 1.0)}
