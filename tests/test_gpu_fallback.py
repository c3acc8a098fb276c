"""The fallback scatter kernel (rsx_scatter_kernel: ranking through per-wave LDS match tables, no reliance on the lane
order of returning LDS atomics).  The library chooses it when the device self-check fails; RSX_FORCE_TABLE_RANK=1 forces it.
The choice is made once per context, so the parity tests are re-run in a child process with the variable set."""
import os
import subprocess
import sys

import pytest

import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_on_the_fallback_kernels():
    rsa.require_gpu()
    env = dict(os.environ, RSX_FORCE_TABLE_RANK="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu",
                          "-k", "(golden or sweep or contract or pairs or rank or skewed or unaligned or records) and not inplace_async"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
