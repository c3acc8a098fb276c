"""The C++ template surface (include/radix_sort.hpp, radix_sort_rank.hpp) on the GPU.

tests/cpp/dropin_check   our own program: exact stable order / returned pointer / rank checks through the templates.
oracle/_ref/radix_tests_dropin   the reference's UNMODIFIED radix_tests.cpp compiled against include/ and linked
                         with librsx.so (built where /root/reference exists; it travels to the GPU box as a binary).
"""
import os
import subprocess

import pytest

import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    rsa.require_gpu()


def test_dropin_check_program():
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "cpp"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "dropin_check OK" in out.stdout, out.stdout + out.stderr


def test_reference_tests_run_unmodified_on_our_headers():
    exe = os.path.join(ROOT, "oracle", "_ref", "radix_tests_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/radix_tests_dropin not built (needs /root/reference at build time)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "All tests OK." in out.stdout, out.stdout + out.stderr
    for line in ("Sorting struct sortrec... OK", "Sorting struct sortrec** (reverse)... OK", "Sorting float[]... OK",
                 "Rank sorting struct sortrec... OK"):
        assert line in out.stdout
