"""CPU-side checks of the drop-in boundary: librsx.so loads, exports every symbol
include/rsx.h declares, and refuses to sort without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import radix_sorting_amd as rsa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "rsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _header_symbols()
    assert len(names) >= 14
    handle = C.CDLL(rsa.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), "librsx.so lacks %s declared in include/rsx.h" % name
    # and the ctypes binding table covers the header exactly
    assert sorted(n for n, _, _ in rsa.ABI) == names


def test_dtype_sizes_and_version():
    lib = rsa.lib()
    assert [lib.rsx_dtype_size(d) for d in range(10)] == rsa.DTYPE_SIZE
    assert lib.rsx_dtype_size(42) == 0
    assert b"gfx950" in lib.rsx_version()
    assert lib.rsx_workspace_bytes(1 << 28, rsa.U32, 0) > (1 << 28) // 8192 * 1024


def test_trivial_sizes_need_no_device():
    """n < 2 returns src untouched before any device work (radix_sort.hpp:100-101)."""
    src = np.array([5], dtype=np.uint32)
    aux = np.array([0xA5], dtype=np.uint32)
    res, info = rsa.radix_sort_host(src, aux, rsa.U32)
    assert res is src and info.early_exit == 1 and aux[0] == 0xA5


@pytest.mark.skipif(rsa.device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    src = np.array([3, 1, 2], dtype=np.uint32)
    aux = np.zeros(3, dtype=np.uint32)
    with pytest.raises(rsa.RsxError, match="no gfx950"):
        rsa.radix_sort_host(src, aux, rsa.U32)
    assert list(src) == [3, 1, 2] and not aux.any()
    ib = np.zeros(6, dtype=np.uint32)
    with pytest.raises(rsa.RsxError, match="no gfx950"):
        rsa.radix_sort_rank_host(src, ib, rsa.U32)


def test_bad_arguments_are_rejected():
    lib = rsa.lib()
    res = C.c_void_p()
    a = np.zeros(4, dtype=np.uint32)
    assert lib.rsx_sort(a.ctypes.data, a.ctypes.data, 4, 99, 0, C.byref(res), None) == -1
    assert b"bad argument" in lib.rsx_last_error()
    assert lib.rsx_sort_rank(a.ctypes.data, a.ctypes.data, 300, rsa.U32, 1, 0, C.byref(res), None) == -1
    assert b"does not fit" in lib.rsx_last_error()
