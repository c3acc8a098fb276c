"""Seeded differential fuzz of the three entry points against the oracle: random sizes (clustered around the small-sort
limits, tile sizes and their multiples), key types, column masks, skew, orders."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

_CARRIER = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def _dev(bits):
    a = np.ascontiguousarray(bits)
    return torch.from_numpy(a.view(_CARRIER[a.itemsize]).copy()).cuda()


def _size(rng):
    anchors = [1, 64, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 98304, 262144, 786432, 1048576, 1572864, 3145728, 4194304,
               5000000]   # (2^20: 2-byte keys become one 16-bit digit)
    kind = rng.integers(0, 4)
    if kind == 0:
        return int(rng.integers(2, 3000))
    if kind == 1:
        return max(2, int(anchors[rng.integers(0, len(anchors))] + rng.integers(-3, 4)))
    if kind == 2:
        return int(rng.integers(3000, 300000))
    return int(rng.integers(300000, int(os.environ.get("RSX_FUZZ_MAXN", "1200000"))))


def _keys(rng, n, dt):
    size = ol.DTYPE_SIZE[dt]
    full = (1 << (8 * size)) - 1
    mask = full
    style = rng.integers(0, 5)
    if style == 1:      # some constant byte columns
        for b in range(size):
            if rng.random() < 0.4:
                mask &= ~(0xFF << (8 * b))
    elif style == 2:    # random bit mask
        mask &= int(rng.integers(0, full, dtype=np.uint64, endpoint=True))
    a = ol.splitmix_fill(n, dt, int(rng.integers(1, 1 << 30)), mask)
    if style == 3:      # heavy duplicates: few distinct values
        pool = a[: max(1, int(rng.integers(1, 40)))]
        a = pool[rng.integers(0, len(pool), size=n)].copy()
    elif style == 4:    # nearly sorted
        a = np.sort(a)
        k = int(rng.integers(0, 4))
        for _ in range(k):
            i, j = rng.integers(0, n, size=2)
            a[i], a[j] = a[j], a[i]
    return a


# RSX_FUZZ_CHUNKS / RSX_FUZZ_SEED: a longer or a different run of the same fuzz (40 cases per chunk)
_CHUNKS = int(os.environ.get("RSX_FUZZ_CHUNKS", "6"))
_SEED = int(os.environ.get("RSX_FUZZ_SEED", "20240"))


@pytest.mark.parametrize("chunk", range(_CHUNKS))
def test_fuzz_keys_pairs_ranks(chunk):
    rsa.require_gpu()
    rng = np.random.default_rng(_SEED + chunk)
    for case in range(40):
        dt = int(rng.integers(0, 10))
        n = _size(rng)
        order = int(rng.integers(0, 2))
        a = _keys(rng, n, dt)
        what = int(rng.integers(0, 3))
        tag = (chunk, case, ol.DTYPE_NAMES[dt], n, order, what)
        if what == 0:
            want, want_aux, winfo = ol.oracle_sort(a, dt, order)
            src = _dev(a)
            aux = torch.full_like(src, 0x5A)
            res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert info.result_in_aux == want_aux and info.early_exit == winfo.early_exit, tag
            assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), tag
            assert np.array_equal(res.cpu().numpy().view(ol.NP_BITS[dt]), want), tag
            if winfo.early_exit:
                assert bool((aux == 0x5A).all().item()), tag
        elif what == 1:
            vb = int(rng.choice([4, 8]))
            vt = torch.int32 if vb == 4 else torch.int64
            perm = ol.stable_argsort_by_kdf(a, dt, order)
            _, want_aux, winfo = ol.oracle_sort(a, dt, order)
            keys, keys_aux = _dev(a), _dev(np.zeros_like(a))
            vals = torch.arange(n, dtype=vt, device="cuda") * 7 + 3
            vals_aux = torch.zeros_like(vals)
            kr, vr, info = rsa.radix_sort_pairs(keys, keys_aux, vals, vals_aux, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert info.result_in_aux == want_aux, tag
            if not winfo.early_exit:
                assert np.array_equal(kr.cpu().numpy().view(ol.NP_BITS[dt]), a[perm]), tag
                assert np.array_equal(vr.cpu().numpy(), perm.astype(np.int64) * 7 + 3), tag
        else:
            ib_bytes = int(rng.choice([4, 8]))
            it = torch.int32 if ib_bytes == 4 else torch.int64
            want, whalf, winfo, _ = ol.oracle_rank(a, dt, ib_bytes, order)
            ib = torch.full((2 * n,), -1, dtype=it, device="cuda")
            ranks, info = rsa.radix_sort_rank(_dev(a), ib, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert info.result_in_aux == whalf and info.early_exit == winfo.early_exit, tag
            assert np.array_equal(ranks.cpu().numpy().astype(np.uint64), want.astype(np.uint64)), tag


@pytest.mark.parametrize("chunk", range(max(2, _CHUNKS // 3)))
def test_fuzz_host_multi_and_split(chunk):
    """rsx_sort_multi (1-6 ranks on device 0), the inplace sort and the MSD split against the oracle on fuzzed inputs."""
    from radix_sorting_amd import multi
    rsa.require_gpu()
    rng = np.random.default_rng(_SEED + 7000 + chunk)
    for case in range(20):
        dt = int(rng.integers(0, 10))
        n = _size(rng)
        order = int(rng.integers(0, 2))
        a = _keys(rng, n, dt)
        what = int(rng.integers(0, 3))
        tag = (chunk, case, ol.DTYPE_NAMES[dt], n, order, what)
        want, want_aux, winfo = ol.oracle_sort(a, dt, order)
        if what == 0:
            ranks = int(rng.integers(1, 7))
            src, aux = a.copy(), np.full_like(a, 0x5A)
            res, info = rsa.radix_sort_multi_host(src, aux, dt, order, [0] * ranks)
            assert bool(info.result_in_aux) == bool(want_aux) and info.early_exit == winfo.early_exit, tag
            assert np.array_equal(res, want), tag
            if winfo.early_exit:
                assert np.all(aux == 0x5A), tag
        elif what == 1:
            buf = _dev(a)
            scratch = torch.zeros_like(buf)
            rsa.radix_sort_inplace_async(buf, scratch, dtype=dt, order=order)
            torch.cuda.synchronize()
            assert np.array_equal(buf.cpu().numpy().view(ol.NP_BITS[dt]), want), tag
        else:
            kb = ol.DTYPE_SIZE[dt]
            col = int(rng.integers(0, kb))
            eng = multi.HipEngine(dt, order)
            shard = _dev(a)
            out = torch.zeros_like(shard)
            hist = eng.msd_split(shard, out, col)
            torch.cuda.synchronize()
            k = ol.kdf_keys(a, dt, order)
            dig = ((k >> ol.NP_BITS[dt](8 * col)) & ol.NP_BITS[dt](0xFF)).astype(np.int64)
            assert np.array_equal(hist, np.bincount(dig, minlength=256).astype(np.uint64)), tag
            assert np.array_equal(out.cpu().numpy().view(ol.NP_BITS[dt]), a[np.argsort(dig, kind="stable")]), tag
