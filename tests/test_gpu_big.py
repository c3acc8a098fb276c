"""Beyond 2^32 keys: the counter width follows n (radix_sort.hpp:102-114 picks uint64_t counters from 2^32 elements on; here
the status words of the look-back chain are 64-bit from 2^30 keys on and every offset is 64-bit).  2^32 + 4097 u32 keys,
16 GiB per buffer, generated and checked on the device: sortedness, the key sum and the key xor (size-independent
properties: nothing of this size goes through the oracle).  37 ms of GPU time for the sort itself."""
import os
import subprocess
import sys

import pytest

import radix_sorting_amd as rsa

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sort_of_more_than_2p32_keys():
    rsa.require_gpu()
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 44 * (1 << 30):
        pytest.skip("needs 44 GiB of free HBM (two 16 GiB buffers and the checks' temporaries)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_sort_check.py"), "32", "4097"], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "sorted True" in out.stdout and "preserved True" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("no_blind,route", [("1", 4), ("0", 5)], ids=["histogram-first", "without-histogram"])
def test_two_levels_with_medium_leaves_at_2p29_keys(no_blind, route):
    """2^29 + 12345 u32 keys: 65536 (digit, digit) buckets of 8 Ki keys -- the slack route with the 16 Ki-key leaf shape, with
    the histogram first (rsx_info.hybrid == 4) and without one (5); sortedness and checksums on the device."""
    rsa.require_gpu()
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * (1 << 30):
        pytest.skip("needs 20 GiB of free HBM")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_sort_check.py"), "29", "12345"], capture_output=True,
                         text=True, timeout=900, env=dict(os.environ, RSX_NO_BLIND=no_blind))
    assert out.returncode == 0 and ("route %d" % route) in out.stdout and "sorted True" in out.stdout and \
        "preserved True" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("log2n,extra,free_gib", [(28, 50000000, 8), (28, 130000000, 10), (29, 200000000, 16), (30, 7, 24), (30, 500000000, 34),
                                                  (31, 4097, 44)],
                         ids=["304 Mi (6144-value slots)", "380 Mi (7680)", "703 Mi (15360)", "2^30 (20480)", "1.47 x 2^30 (counting, 32768)",
                              "2^31 (counting, 40960)"])
def test_two_levels_without_histogram_up_to_2p31_keys(log2n, extra, free_gib):
    """2^30 and 2^31 u32 keys keep the route of BASELINE.json's headline (two MSB passes without a histogram, leaves): the
    level-2 slots hold two-byte values -- 16 Ki of them at 2^30 keys (rsx_leaf16_kernel's 20480-value shape), 32 Ki at 2^31 (the
    counting leaves, csrc/rsx_leafc.hpp); every offset still fits 32 bits.  Sortedness and checksums on the device."""
    rsa.require_gpu()
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < free_gib * (1 << 30):
        pytest.skip("needs %d GiB of free HBM" % free_gib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_sort_check.py"), str(log2n), str(extra)], capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "route 5" in out.stdout and "sorted True" in out.stdout and "preserved True" in out.stdout, \
        out.stdout + out.stderr
