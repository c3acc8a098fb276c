"""The stability guard: ranking by returning LDS atomics rests on gfx950 handing the lanes of one
instruction that hit the same cell their values in lane order.  Guarded by (1) a device self-check in two shapes when a
context is created (rsx.hip, lds_order_selfcheck: 8 waves of bare atomics and the production shape of 16 waves with
staging traffic), (2) RSX_VERIFY=1, which re-ranks one tile of every host-scheduled scatter pass without LDS atomics and
fails the call on any difference, and (3) this soak: 20 s of back-to-back sorts whose outputs are checked."""
import os
import subprocess
import sys

import pytest

import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


def _run(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_soak_20_seconds():
    out = _run([os.path.join("tools", "soak.py"), "20"], {})
    assert out.returncode == 0 and "soak ok" in out.stdout, out.stdout + out.stderr


VERIFY_SCRIPT = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import oracle_lib as ol, radix_sorting_amd as rsa
from radix_sorting_amd import multi
carrier = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}
def dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(carrier[a.itemsize]).copy()).cuda()
n_checked = 0
for dt, n, mask in [(ol.U32, 3000001, 0xFFFFFFFF), (ol.U64, 1500001, 0xFFFFFFFFFF), (ol.F32, 2000003, 0xFFFFFFFF),
                    (ol.I16, 1000001, 0xFFFF), (ol.U8, 900001, 0xFF), (ol.F64, 700001, (1 << 64) - 1), (ol.U32, 200001, 0x03030303)]:
    a = ol.splitmix_fill(n, dt, 5 + dt, mask)
    for order in (0, 1):
        src = dev(a); aux = torch.zeros_like(src)
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy().view(ol.NP_BITS[dt]), ol.oracle_sort(a, dt, order)[0])
    ib = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
    ranks, _ = rsa.radix_sort_rank(dev(a), ib, dtype=dt)
    torch.cuda.synchronize()
    assert np.array_equal(ranks.cpu().numpy().view(np.uint32), ol.oracle_rank(a, dt, 4)[0])
    vals = torch.arange(n, dtype=torch.int64, device="cuda")
    k, v, _ = rsa.radix_sort_pairs(dev(a), torch.zeros_like(dev(a)), vals, torch.zeros_like(vals), dtype=dt)
    torch.cuda.synchronize()
    assert np.array_equal(v.cpu().numpy().astype(np.uint64), ol.stable_argsort_by_kdf(a, dt).astype(np.uint64))
    n_checked += 4
a = ol.splitmix_fill(2500001, ol.U32, 99)
eng = multi.HipEngine(ol.U32)
out = torch.zeros(a.size, dtype=torch.int32, device="cuda")
eng.msd_split(dev(a), out)
torch.cuda.synchronize()
print("verify ok", n_checked)
""" % (ROOT, ROOT)


def test_every_pass_verified_by_ballot_reranking():
    """RSX_VERIFY=1: keys, pairs, ranks (narrowed keys, generated indices, key-less last pass) and the MSD split of seven key
    types all pass the per-pass tile verification (and equal the oracle)."""
    out = _run(["-c", VERIFY_SCRIPT], {"RSX_VERIFY": "1"})
    assert out.returncode == 0 and "verify ok 28" in out.stdout, out.stdout + out.stderr


def test_verification_failure_is_reported():
    """The mismatch path end to end (RSX_VERIFY_INJECT makes the verifier expect every key XOR 1): the sort fails with
    RSX_EVERIFY and says what disagreed."""
    out = _run(["-c", VERIFY_SCRIPT], {"RSX_VERIFY": "1", "RSX_VERIFY_INJECT": "1"})
    assert out.returncode != 0
    assert "rsx error -5" in out.stderr and "ballot-ranked" in out.stderr, out.stderr


def test_soak_with_verification():
    out = _run([os.path.join("tools", "soak.py"), "6", "--small"], {"RSX_VERIFY": "1"})
    assert out.returncode == 0 and "soak ok" in out.stdout and "RSX_VERIFY=1" in out.stdout, out.stdout + out.stderr


ASYNC_VERIFY_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import oracle_lib as ol
import radix_sorting_amd as rsa
carrier = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}
def dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(carrier[a.itemsize]).copy()).cuda()
n_checked = 0
for dt, n, mask in [(ol.U32, 3000001, 0xFFFFFFFF), (ol.U64, 1500001, 0xFFFFFFFFFF), (ol.F32, 2000003, 0xFFFFFFFF), (ol.I32, 400001, 0x00FF00FF)]:
    a = ol.splitmix_fill(n, dt, 15 + dt, mask)
    for order in (0, 1):
        buf = dev(a); scratch = torch.zeros_like(buf)
        rsa.radix_sort_inplace_async(buf, scratch, dtype=dt, order=order)
        ib = torch.zeros(2 * n, dtype=torch.int32, device="cuda")
        ranks = rsa.radix_sort_rank_inplace_async(dev(a), ib, dtype=dt, order=order)
        keys = dev(a); vals = torch.arange(n, dtype=torch.int64, device="cuda")
        rsa.radix_sort_pairs_inplace_async(keys, torch.zeros_like(keys), vals, torch.zeros_like(vals), dtype=dt, order=order)
        assert rsa.verify_poll() == 0
        assert np.array_equal(buf.cpu().numpy().view(ol.NP_BITS[dt]), ol.oracle_sort(a, dt, order)[0])
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), ol.oracle_rank(a, dt, 4, order)[0])
        assert np.array_equal(vals.cpu().numpy().astype(np.uint32), ol.oracle_rank(a, dt, 4, order)[0])
        n_checked += 3
print("async verify ok", n_checked)
""" % (ROOT, ROOT)


def test_device_scheduled_sorts_are_verified_too():
    """RSX_VERIFY=1 covers the *_inplace_async sorts (keys, key + payload, ranks): every device-scheduled pass has one tile
    re-ranked with ballots on the device, the mismatches are collected by rsx_verify_poll."""
    out = _run(["-c", ASYNC_VERIFY_SCRIPT], {"RSX_VERIFY": "1"})
    assert out.returncode == 0 and "async verify ok 24" in out.stdout, out.stdout + out.stderr


def test_async_verification_failure_is_reported_at_the_poll():
    out = _run(["-c", ASYNC_VERIFY_SCRIPT], {"RSX_VERIFY": "1", "RSX_VERIFY_INJECT": "1"})
    assert out.returncode != 0
    assert "rsx error -5" in out.stderr and "device-scheduled" in out.stderr, out.stderr


WHOLE_VERIFY_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import oracle_lib as ol
import radix_sorting_amd as rsa
carrier = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}
def dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(carrier[a.itemsize]).copy()).cuda()
routes = set()
for dt, n, mask in [(ol.U32, 100003, 0xFFFFFFFF), (ol.U32, 3000001, 0xFFFFFFFF), (ol.I32, 5000000, 0xFFFFFFFF), (ol.U32, (1 << 23) + 7, 0xFFFFFFFF),
                    (ol.U64, (1 << 23) + 7, 0xFFFFFFFFFF), (ol.F32, 2000003, 0xFFFFFFFF), (ol.U32, (1 << 26) + 11, 0xFFFFFFFF),
                    (ol.U16, 3000001, 0xFFFF), (ol.U32, 70000, 0x00FF00FF)]:
    a = ol.splitmix_fill(n, dt, 25 + dt, mask)
    for order in (0, 1):
        src = dev(a); aux = torch.zeros_like(src)
        res, info = rsa.radix_sort(src, aux, dtype=dt, order=order)
        torch.cuda.synchronize()
        routes.add(int(info.hybrid))
        if n <= 5000000:
            assert np.array_equal(res.cpu().numpy().view(ol.NP_BITS[dt]), ol.oracle_sort(a, dt, order)[0])
import os
os.environ["RSX_NO_BLIND"] = "1"        # (the same sizes with the histogram first: the slack route that starts from its counts)
rsa.reload_env()
for dt, n, mask in [(ol.U32, (1 << 26) + 11, 0xFFFFFFFF), (ol.U64, (1 << 23) + 7, 0xFFFFFFFFFF)]:   # (slots from counts; counted second pass)
    a = ol.splitmix_fill(n, dt, 27, mask)
    src = dev(a); aux = torch.zeros_like(src)
    res, info = rsa.radix_sort(src, aux, dtype=dt)
    torch.cuda.synchronize()
    routes.add(int(info.hybrid))
print("whole verify ok, routes", sorted(routes))
""" % (ROOT, ROOT)


def test_whole_result_verification_over_every_route():
    """RSX_VERIFY=2: the sort as it always runs -- leaves, slack slots, speculation -- with its result checked on the device
    (sorted, and the input's key sum and key mix): every route of csrc/rsx_hybrid.hpp under it."""
    out = _run(["-c", WHOLE_VERIFY_SCRIPT], {"RSX_VERIFY": "2", "RSX_TWO_LEVEL_MIN_LOG2": "22"})
    assert out.returncode == 0 and "whole verify ok, routes [0, 1, 2, 4, 5]" in out.stdout, out.stdout + out.stderr


def test_whole_result_verification_failure_is_reported():
    out = _run(["-c", WHOLE_VERIFY_SCRIPT], {"RSX_VERIFY": "2", "RSX_VERIFY_INJECT": "1"})
    assert out.returncode != 0
    assert "rsx error -5" in out.stderr and "RSX_VERIFY=2" in out.stderr, out.stderr


PAIRS_VERIFY_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np, torch
import oracle_lib as ol
import radix_sorting_amd as rsa
routes = set()
for dt, n, mask in [(ol.F32, 70001, 0xFFFFFFFF), (ol.U32, 600000, 0xFFFFFFFF), (ol.I32, (1 << 22) + 99, 0xFFFFFFFF), (ol.U32, 3000001, 0x00FFFF0F),
                    (ol.F32, 6000001, 0xFFFFFFFF)]:
    a = ol.splitmix_fill(n, dt, 35 + dt, mask)
    for order in (0, 1):
        keys = torch.from_numpy(a.view(np.int32).copy()).cuda()
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks, info = rsa.radix_sort_rank(keys, ib, dtype=dt, order=order)          # RSX_VERIFY=2 checks the ranks on the device
        torch.cuda.synchronize()
        routes.add(("rank", int(info.hybrid)))
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), ol.oracle_rank(a, dt, 4, order)[0])
        vals = torch.arange(n, dtype=torch.int32, device="cuda") * 3 + 1
        ka, va = torch.empty_like(keys), torch.empty_like(vals)
        kr, vr, info = rsa.radix_sort_pairs(keys, ka, vals, va, dtype=dt, order=order)   # ... and the pairs
        torch.cuda.synchronize()
        routes.add(("pairs", int(info.hybrid)))
print("pairs verify ok, routes", sorted(routes))
""" % (ROOT, ROOT)


def test_whole_result_verification_of_rank_and_pair_sorts():
    """RSX_VERIFY=2 for rank sorts (a permutation of 0 .. n-1 through which the keys do not descend, equal keys in index order)
    and key + payload sorts (no descent, the input's key sum and pair mix), on the routes they take -- one pass per column, one
    MSB pass + leaves, two MSB passes into slots + the compound leaves."""
    out = _run(["-c", PAIRS_VERIFY_SCRIPT], {"RSX_VERIFY": "2", "RSX_TWO_LEVEL_MIN_LOG2": "22"})
    assert out.returncode == 0 and "pairs verify ok" in out.stdout, out.stdout + out.stderr
    assert "('rank', 5)" in out.stdout and "('pairs', 5)" in out.stdout and "('rank', 1)" in out.stdout, out.stdout


def test_whole_result_verification_of_rank_sorts_reports_failure():
    out = _run(["-c", PAIRS_VERIFY_SCRIPT], {"RSX_VERIFY": "2", "RSX_VERIFY_INJECT": "1"})
    assert out.returncode != 0
    assert "rsx error -5" in out.stderr and "RSX_VERIFY=2" in out.stderr and "ranks" in out.stderr, out.stderr
