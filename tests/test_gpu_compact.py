"""SURVEY.md 8 f4 / README.md:716-758, "key compaction": with RSX_COMPACT_BITS=1 a rank sort whose keys vary in few bits
spread over several bytes packs those bits together first and sorts the packed values in fewer passes.  The ranks, the
returned half of the index buffer and the reported kept columns must stay exactly the reference's."""
import os
import subprocess
import sys

import pytest

import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import oracle_lib as ol, radix_sorting_amd as rsa
carrier = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}
def dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(carrier[a.itemsize]).copy()).cuda()
cases = [  # dtype, n, mask: varying bits / varying bytes
    (ol.U32, 1500001, 0x0F0F0F0F),          # 16 bits in 4 bytes: 4 passes -> 2
    (ol.U32, 700001, 0x01010101),           # 4 bits in 4 bytes: 4 -> 1 (the ranks must end in the FIRST half: 4 is even)
    (ol.U32, 900001, 0x00030303),           # 6 bits in 3 bytes: 3 -> 1 (second half)
    (ol.U64, 1200001, 0x0103000701030007),  # 14 bits in 6 bytes: 6 -> 2
    (ol.I32, 800001, 0x80000F01),           # sign bit + 5 bits: the KDF flips the sign bit, still one varying bit there
    (ol.F32, 1000003, 0x3F000707),          # float keys
    (ol.U16, 600001, 0x0303),               # 4 bits in 2 bytes: 2 -> 1
    (ol.U32, 1000001, 0xFFFFFFFF),          # nothing to gain: the usual passes
    (ol.U32, 1000001, 0xFFF000FF),          # cfg 4 (iii): 20 bits in 3 bytes: 3 passes either way
    (ol.U64, 500001, 0xAAAAAAAAAAAAAAAA),   # 32 runs of one bit: more than the kernel's eight runs -> the usual passes
]
for dt, n, mask in cases:
    a = ol.splitmix_fill(n, dt, 11 + dt, mask)
    for order in (0, 1):
        ib = torch.full((2 * n,), -1, dtype=torch.int32, device="cuda")
        ranks, info = rsa.radix_sort_rank(dev(a), ib, dtype=dt, order=order)
        torch.cuda.synchronize()
        want, half, winfo, _ = ol.oracle_rank(a, dt, 4, order)
        assert info.result_in_aux == half, (dt, hex(mask), info.result_in_aux, half)
        assert info.kept_columns() == list(winfo.cols[:winfo.ncols]), (dt, hex(mask))
        assert np.array_equal(ranks.cpu().numpy().view(np.uint32), want), (dt, hex(mask), order)
print("compact ok", len(cases))
""" % (ROOT, ROOT)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


@pytest.mark.parametrize("compact", ["1", "0"])
def test_rank_sorts_with_and_without_bit_compaction(compact):
    env = dict(os.environ, RSX_COMPACT_BITS=compact)
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0 and "compact ok 10" in out.stdout, out.stdout + out.stderr


def test_bit_compaction_under_pass_verification():
    env = dict(os.environ, RSX_COMPACT_BITS="1", RSX_VERIFY="1")
    out = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0 and "compact ok 10" in out.stdout, out.stdout + out.stderr
