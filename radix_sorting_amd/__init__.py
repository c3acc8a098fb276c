"""radix_sorting_amd -- MI355X-native LSD radix sort behind eloj/radix-sorting's surface.

The product is ``librsx.so`` (hand-written HIP for gfx950, C ABI in
``include/rsx.h``) plus the C++ template headers in ``include/``.  This Python
package is only the thin host-side doorway used by the tests, ``bench.py`` and
the multi-GPU driver: it binds the C ABI with ctypes and passes torch tensors'
``data_ptr()`` / current stream straight through.  PyTorch is plumbing here
(device memory, streams, ``torch.distributed``); no sorting happens in Python
and there is no CPU fallback -- a missing library or GPU raises.

Function names and argument meaning follow the reference
(``radix_sort(src, aux, n, kdf)`` radix_sort.hpp:98-99,
``radix_sort_rank(src, index_buffer, n, kdf)`` radix_sort_rank.hpp:97-98).
"""
import ctypes as C
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RSX_LIB: another build of the library, for A/B measurements of compile-time variants inside one gpurun call)
LIB_PATH = os.environ.get("RSX_LIB") or os.path.join(_HERE, "librsx.so")

# rsx_dtype (include/rsx.h)
U8, U16, U32, U64, I8, I16, I32, I64, F32, F64 = range(10)
DTYPE_SIZE = [1, 2, 4, 8, 1, 2, 4, 8, 4, 8]
ASCENDING, DESCENDING = 0, 1


class RsxError(RuntimeError):
    pass


class Profile(C.Structure):
    """rsx_profile: HIP-event kernel timings collected between profile_begin() and profile_end()."""
    _fields_ = [("hist_ms", C.c_double), ("scatter_ms", C.c_double), ("hist_launches", C.c_uint64),
                ("scatter_launches", C.c_uint64), ("hist_bytes", C.c_uint64), ("scatter_bytes", C.c_uint64),
                ("leaf_ms", C.c_double), ("leaf_launches", C.c_uint64), ("leaf_bytes", C.c_uint64),
                ("narrow_ms", C.c_double), ("narrow_launches", C.c_uint64), ("narrow_bytes", C.c_uint64),
                ("called_off_ms", C.c_double), ("called_off_launches", C.c_uint64)]


class Info(C.Structure):
    """rsx_info: what the front half of rs_sort_main decided (radix_sort.hpp:48-80)."""
    _fields_ = [("key_bytes", C.c_uint32), ("ncols", C.c_uint32), ("cols", C.c_uint32 * 8),
                ("early_exit", C.c_uint32), ("result_in_aux", C.c_uint32), ("hybrid", C.c_uint32)]

    def kept_columns(self):
        return [int(self.cols[i]) for i in range(self.ncols)]


# every symbol include/rsx.h declares: (name, restype, argtypes)
_VP, _SZ, _I, _U32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32
_PVP, _PINFO = C.POINTER(C.c_void_p), C.POINTER(Info)
ABI = [
    ("rsx_device_count", _I, []),
    ("rsx_last_error", C.c_char_p, []),
    ("rsx_version", C.c_char_p, []),
    ("rsx_dtype_size", _SZ, [_I]),
    ("rsx_workspace_bytes", _SZ, [_SZ, _I, _SZ]),
    ("rsx_release", None, []),
    ("rsx_reload_env", None, []),
    ("rsx_sort", _I, [_VP, _VP, _SZ, _I, _I, _PVP, _PINFO]),
    ("rsx_release_stream", None, [_VP]),
    ("rsx_sort_inplace_async", _I, [_VP, _VP, _SZ, _I, _I, _VP]),
    ("rsx_sort_inplace_async_hint", _I, [_VP, _VP, _SZ, _I, _I, _VP, C.c_uint32]),
    ("rsx_sort_inplace_async_ws", _I, [_VP, _VP, _SZ, _I, _I, _VP, _SZ, _VP]),
    ("rsx_workspace_bytes_fast", _SZ, [_SZ, _I]),
    ("rsx_async_route_ws", _I, [_VP, _SZ, _SZ, _I, _VP, C.POINTER(C.c_uint32)]),
    ("rsx_sort_pairs_inplace_async_ws", _I, [_VP, _VP, _VP, _VP, _SZ, _I, _SZ, _I, _VP, _SZ, _VP]),
    ("rsx_sort_pairs_inplace_async", _I, [_VP, _VP, _VP, _VP, _SZ, _I, _SZ, _I, _VP]),
    ("rsx_capture_histogram", _I, [_VP, _SZ]),
    ("rsx_sort_device", _I, [_VP, _VP, _SZ, _I, _I, _VP, _PVP, _PINFO]),
    ("rsx_sort_pairs_device", _I, [_VP, _VP, _VP, _VP, _SZ, _I, _SZ, _I, _VP, _PINFO]),
    ("rsx_sort_rank", _I, [_VP, _VP, _SZ, _I, _SZ, _I, _PVP, _PINFO]),
    ("rsx_sort_rank_device", _I, [_VP, _VP, _SZ, _I, _SZ, _I, _VP, _PVP, _PINFO]),
    ("rsx_sort_records", _I, [_VP, _VP, _SZ, _SZ, _VP, _SZ, _PVP, _PINFO]),
    ("rsx_sort_records_tagged", _I, [_VP, _VP, _SZ, _SZ, _SZ, _I, _I, _PVP, _PINFO]),
    ("rsx_sort_records_tagged_device", _I, [_VP, _VP, _SZ, _SZ, _SZ, _I, _I, _VP, _PVP, _PINFO]),
    ("rsx_sort_rank_keys", _I, [_VP, _SZ, _VP, _SZ, _SZ, _PVP, _PINFO]),
    ("rsx_histogram_device", _I, [_VP, _SZ, _I, _I, _VP, _VP, _VP]),
    ("rsx_msd_split_device", _I, [_VP, _VP, _SZ, _I, _I, _I, _VP, _VP]),
    ("rsx_msd_split_async", _I, [_VP, _VP, _SZ, _I, _I, _I, _VP, _VP]),
    ("rsx_sort_rank_inplace_async", _I, [_VP, _VP, _SZ, _I, _SZ, _I, _VP]),
    ("rsx_verify_poll", _I, [_VP, C.POINTER(C.c_uint64)]),
    ("rsx_async_route", _I, [_VP, C.POINTER(C.c_uint32)]),
    ("rsx_sort_multi", _I, [_VP, _VP, _SZ, _I, _I, _VP, _I, _PVP, _PINFO]),
    ("rsx_profile_begin", _I, []),
    ("rsx_profile_end", _I, [C.POINTER(Profile)]),
    ("rsx_fill_splitmix_device", _I, [_VP, _SZ, _SZ, C.c_uint64, C.c_uint64, C.c_uint64, _VP]),
    ("rsx_spin_device", _I, [C.c_uint64, _VP]),
]

_lib = None


def lib():
    """The loaded librsx.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RsxError("%s is missing: build it with `make lib` (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (SONAME
        # libamdhip64.so.7, the same as /opt/rocm's); whichever copy is mapped first serves every
        # later NEEDED entry of that SONAME, and a second copy cannot see the GPU.  Where torch is
        # installed it therefore has to be imported before librsx.so pulls in /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            f = getattr(handle, name)      # AttributeError here means the ABI and the header diverged
            f.restype = res
            f.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RsxError("rsx error %d: %s" % (rc, lib().rsx_last_error().decode()))


def reload_env():
    """Have the library read its RSX_* environment switches again (it reads them once, at its first call)."""
    lib().rsx_reload_env()


def device_count():
    return int(lib().rsx_device_count())


def require_gpu():
    if device_count() <= 0:
        raise RsxError("no gfx950 (MI355X) device visible to HIP; radix_sorting_amd has no CPU path")


# ---- torch-facing helpers -------------------------------------------------------------------

def _torch_dtype_code(t):
    import torch
    table = {torch.uint8: U8, torch.int8: I8, torch.int16: I16, torch.int32: I32, torch.int64: I64,
             torch.float32: F32, torch.float64: F64}
    for name, code in (("uint16", U16), ("uint32", U32), ("uint64", U64)):
        if hasattr(torch, name):
            table[getattr(torch, name)] = code
    if t.dtype not in table:
        raise RsxError("unsupported tensor dtype %s" % t.dtype)
    return table[t.dtype]


def _stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def _check_dev(*tensors):
    """Every buffer handed to the kernels as a raw pointer: a contiguous tensor on the CURRENT device."""
    import torch
    cur = torch.cuda.current_device()
    for t in tensors:
        if not t.is_cuda or not t.is_contiguous():
            raise RsxError("expected contiguous device tensors")
        if t.device.index != cur:
            raise RsxError("tensor on cuda:%d but the current device is cuda:%d (the library works on the current device)"
                           % (t.device.index, cur))


def _same_shape(a, b, what):
    """`b` is the ping-pong partner of `a`: same element size, at least as many elements."""
    if b.element_size() != a.element_size() or b.numel() < a.numel():
        raise RsxError("%s does not match its buffer (element size %d vs %d, %d vs %d elements)"
                       % (what, b.element_size(), a.element_size(), b.numel(), a.numel()))


_armed_hist = threading.local()   # keeps the armed array alive: the library only holds its address


def capture_histogram(hist):
    """rsx_capture_histogram: arm the calling thread's next blocking sort to write the counts of loop 1
    (radix_sort.hpp:48-58) into `hist` (numpy uint64, >= 256 * key bytes entries); None disarms.

    The library keeps the raw address until a sort reaches its histogram: prefer ``with capturing_histogram(hist):``,
    which disarms on the way out whatever happened in between (an exception before the sort, an *_async call that
    never captures), so that no later sort of this thread writes into an array that is gone."""
    if hist is None:
        check(lib().rsx_capture_histogram(None, 0))
        _armed_hist.ref = None
    else:
        import numpy as np
        if hist.dtype != np.uint64 or not hist.flags["C_CONTIGUOUS"]:
            raise RsxError("capture_histogram wants a contiguous numpy uint64 array")
        check(lib().rsx_capture_histogram(hist.ctypes.data, hist.size))
        _armed_hist.ref = hist


class capturing_histogram:
    """Context manager around capture_histogram: armed on entry, disarmed on exit (rsx_capture_histogram(NULL, 0))."""

    def __init__(self, hist):
        self.hist = hist

    def __enter__(self):
        capture_histogram(self.hist)
        return self.hist

    def __exit__(self, *exc):
        capture_histogram(None)
        return False


def radix_sort(src, aux, dtype=None, order=ASCENDING, stream=None):
    """radix_sort(src, aux, n) on device tensors (radix_sort.hpp:98-115).

    Returns (result, info): ``result`` is ``src`` or ``aux`` by the reference's
    returned-pointer rule.  ``dtype`` overrides the rsx_dtype code, so bit
    patterns held in an int32/int64 tensor can be sorted as uint32/uint64/float.
    The scatter passes are only enqueued on ``stream``.
    """
    _check_dev(src, aux)
    code = _torch_dtype_code(src) if dtype is None else dtype
    if src.element_size() != DTYPE_SIZE[code]:
        raise RsxError("src does not match the key type")
    _same_shape(src, aux, "aux")
    res, info = C.c_void_p(), Info()
    check(lib().rsx_sort_device(src.data_ptr(), aux.data_ptr(), src.numel(), code, order, _stream_ptr(stream),
                                C.byref(res), C.byref(info)))
    return (aux if info.result_in_aux else src), info


HINT_EVEN_TOP_DIGITS = 1


def radix_sort_inplace_async(buf, scratch, dtype=None, order=ASCENDING, stream=None, hints=0):
    """rsx_sort_inplace_async: no host synchronisation, the sorted keys always end in ``buf`` (graph-capturable).
    ``hints`` (rsx_sort_inplace_async_hint): HINT_EVEN_TOP_DIGITS -- the caller has counted the keys by their top varying byte."""
    _check_dev(buf, scratch)
    code = _torch_dtype_code(buf) if dtype is None else dtype
    if buf.element_size() != DTYPE_SIZE[code]:
        raise RsxError("buf does not match the key type")
    _same_shape(buf, scratch, "scratch")
    if hints:
        check(lib().rsx_sort_inplace_async_hint(buf.data_ptr(), scratch.data_ptr(), buf.numel(), code, order, _stream_ptr(stream),
                                                int(hints)))
    else:
        check(lib().rsx_sort_inplace_async(buf.data_ptr(), scratch.data_ptr(), buf.numel(), code, order, _stream_ptr(stream)))
    return buf


def workspace_bytes_fast(n, dtype):
    """rsx_workspace_bytes_fast: a workspace that also holds the slots of a sort without a histogram (the *_ws sort then takes
    that route inside it)."""
    return int(lib().rsx_workspace_bytes_fast(n, dtype))


def async_route_ws(workspace, n, dtype, stream=None):
    """rsx_async_route_ws: the route the last rsx_sort_inplace_async_ws in `workspace` took (waits for the stream)."""
    r = C.c_uint32(0)
    check(lib().rsx_async_route_ws(workspace.data_ptr(), workspace.numel() * workspace.element_size(), n, dtype, _stream_ptr(stream),
                                   C.byref(r)))
    return int(r.value)


def workspace_bytes(n, dtype, payload_bytes=0):
    """rsx_workspace_bytes: what a workspace for the *_ws entry points must hold at least."""
    return int(lib().rsx_workspace_bytes(n, dtype, payload_bytes))


def radix_sort_inplace_async_ws(buf, scratch, workspace, dtype=None, order=ASCENDING, stream=None):
    """rsx_sort_inplace_async_ws: as radix_sort_inplace_async with every piece of device state in `workspace` (a uint8 device
    tensor the caller keeps for as long as a graph captured from this call lives)."""
    _check_dev(buf, scratch, workspace)
    code = _torch_dtype_code(buf) if dtype is None else dtype
    if buf.element_size() != DTYPE_SIZE[code]:
        raise RsxError("buf does not match the key type")
    _same_shape(buf, scratch, "scratch")
    check(lib().rsx_sort_inplace_async_ws(buf.data_ptr(), scratch.data_ptr(), buf.numel(), code, order, workspace.data_ptr(),
                                          workspace.numel() * workspace.element_size(), _stream_ptr(stream)))
    return buf


def radix_sort_pairs_inplace_async_ws(keys, keys_scratch, vals, vals_scratch, workspace, dtype=None, order=ASCENDING, stream=None):
    _check_dev(keys, keys_scratch, vals, vals_scratch, workspace)
    code = _torch_dtype_code(keys) if dtype is None else dtype
    if keys.element_size() != DTYPE_SIZE[code] or vals.element_size() not in (4, 8) or vals.numel() != keys.numel():
        raise RsxError("keys/vals do not match")
    _same_shape(keys, keys_scratch, "keys_scratch")
    _same_shape(vals, vals_scratch, "vals_scratch")
    check(lib().rsx_sort_pairs_inplace_async_ws(keys.data_ptr(), keys_scratch.data_ptr(), vals.data_ptr(), vals_scratch.data_ptr(),
                                                keys.numel(), code, vals.element_size(), order, workspace.data_ptr(),
                                                workspace.numel() * workspace.element_size(), _stream_ptr(stream)))
    return keys, vals


def release_stream(stream=None):
    """rsx_release_stream: free the library's workspace of (current device, stream)."""
    lib().rsx_release_stream(_stream_ptr(stream))


def radix_sort_pairs_inplace_async(keys, keys_scratch, vals, vals_scratch, dtype=None, order=ASCENDING, stream=None):
    """rsx_sort_pairs_inplace_async: keys and payloads sorted in place by the keys, no host synchronisation."""
    _check_dev(keys, keys_scratch, vals, vals_scratch)
    code = _torch_dtype_code(keys) if dtype is None else dtype
    if keys.element_size() != DTYPE_SIZE[code] or vals.element_size() not in (4, 8) or vals.numel() != keys.numel():
        raise RsxError("keys/vals do not match")
    _same_shape(keys, keys_scratch, "keys_scratch")
    _same_shape(vals, vals_scratch, "vals_scratch")
    check(lib().rsx_sort_pairs_inplace_async(keys.data_ptr(), keys_scratch.data_ptr(), vals.data_ptr(), vals_scratch.data_ptr(),
                                             keys.numel(), code, vals.element_size(), order, _stream_ptr(stream)))
    return keys, vals


def radix_sort_rank_inplace_async(src, index_buffer, dtype=None, order=ASCENDING, stream=None):
    """rsx_sort_rank_inplace_async: the stable argsort of ``src`` without a host synchronisation; the ranks always end in
    ``index_buffer[:n]`` (``index_buffer``: 2n int32 / int64 entries).  Returns that view."""
    _check_dev(src, index_buffer)
    code = _torch_dtype_code(src) if dtype is None else dtype
    n = src.numel()
    if src.element_size() != DTYPE_SIZE[code] or index_buffer.element_size() not in (4, 8) or index_buffer.numel() < 2 * n:
        raise RsxError("index_buffer must hold 2n 4- or 8-byte entries")
    check(lib().rsx_sort_rank_inplace_async(src.data_ptr(), index_buffer.data_ptr(), n, code, index_buffer.element_size(), order,
                                            _stream_ptr(stream)))
    return index_buffer[:n]


def verify_poll(stream=None):
    """rsx_verify_poll: wait for the stream and return the mismatches RSX_VERIFY=1 found in device-scheduled sorts since
    the last poll (raises RsxError if there are any)."""
    bad = C.c_uint64(0)
    check(lib().rsx_verify_poll(_stream_ptr(stream), C.byref(bad)))
    return int(bad.value)


def async_route(stream=None):
    """rsx_async_route: wait for the stream and return the route (as rsx_info.hybrid) the last radix_sort_inplace_async on it took."""
    r = C.c_uint32(0)
    check(lib().rsx_async_route(_stream_ptr(stream), C.byref(r)))
    return int(r.value)


def radix_sort_pairs(keys, keys_aux, vals, vals_aux, dtype=None, order=ASCENDING, stream=None):
    """Stable key+payload sort (struct-of-arrays); returns (keys_result, vals_result, info)."""
    _check_dev(keys, keys_aux, vals, vals_aux)
    code = _torch_dtype_code(keys) if dtype is None else dtype
    if keys.element_size() != DTYPE_SIZE[code] or vals.numel() != keys.numel() or vals.element_size() not in (4, 8):
        raise RsxError("keys/vals do not match")
    _same_shape(keys, keys_aux, "keys_aux")
    _same_shape(vals, vals_aux, "vals_aux")
    info = Info()
    check(lib().rsx_sort_pairs_device(keys.data_ptr(), keys_aux.data_ptr(), vals.data_ptr(), vals_aux.data_ptr(),
                                      keys.numel(), code, vals.element_size(), order, _stream_ptr(stream),
                                      C.byref(info)))
    if info.result_in_aux:
        return keys_aux, vals_aux, info
    return keys, vals, info


def radix_sort_rank(src, index_buffer, dtype=None, order=ASCENDING, stream=None):
    """radix_sort_rank(src, index_buffer, n) on device tensors (radix_sort_rank.hpp:97-112).

    ``index_buffer`` holds 2n int32/int64 entries; returns (ranks_view, info) where
    ranks_view is the half of index_buffer the reference would return.
    """
    _check_dev(src, index_buffer)
    code = _torch_dtype_code(src) if dtype is None else dtype
    n = src.numel()
    if src.element_size() != DTYPE_SIZE[code]:
        raise RsxError("src does not match the key type")
    if index_buffer.numel() < 2 * n or index_buffer.element_size() not in (4, 8):
        raise RsxError("index_buffer must hold 2n 4- or 8-byte entries")
    res, info = C.c_void_p(), Info()
    check(lib().rsx_sort_rank_device(src.data_ptr(), index_buffer.data_ptr(), n, code, index_buffer.element_size(),
                                     order, _stream_ptr(stream), C.byref(res), C.byref(info)))
    half = index_buffer[n:2 * n] if info.result_in_aux else index_buffer[:n]
    return half, info


def fill_splitmix(t, seed, mask=0xFFFFFFFFFFFFFFFF, first_index=0, stream=None):
    """Fill a device tensor with the SURVEY.md section 4 / 8d splitmix64 sequence (counter-based on the GPU)."""
    _check_dev(t)
    check(lib().rsx_fill_splitmix_device(t.data_ptr(), t.numel(), t.element_size(), seed, mask, first_index,
                                         _stream_ptr(stream)))
    return t


def spin(microseconds, stream=None):
    """Keep the (current) stream busy for about `microseconds`."""
    require_gpu()
    check(lib().rsx_spin_device(int(microseconds), _stream_ptr(stream)))


def profile_begin():
    check(lib().rsx_profile_begin())


def profile_end():
    p = Profile()
    check(lib().rsx_profile_end(C.byref(p)))
    return p


# ---- host (numpy) entry points: the C ABI exactly as the C++ template wrapper calls it -----

def radix_sort_host(src, aux, dtype, order=ASCENDING):
    """rsx_sort on host numpy buffers; returns (result_array, info)."""
    res, info = C.c_void_p(), Info()
    check(lib().rsx_sort(src.ctypes.data, aux.ctypes.data, src.size, dtype, order, C.byref(res), C.byref(info)))
    return (aux if info.result_in_aux else src), info


def radix_sort_multi_host(src, aux, dtype, order=ASCENDING, devices=None):
    """rsx_sort_multi on host numpy buffers: one process, the work spread over `devices` (HIP device indices, one per rank;
    default: every visible device once).  Returns (result_array, info) with the reference's returned-pointer rule."""
    if devices is None:
        devices = list(range(max(device_count(), 1)))
    dev = (C.c_int * len(devices))(*devices)
    res, info = C.c_void_p(), Info()
    check(lib().rsx_sort_multi(src.ctypes.data, aux.ctypes.data, src.size, dtype, order, dev, len(devices), C.byref(res),
                               C.byref(info)))
    return (aux if info.result_in_aux else src), info


def radix_sort_rank_host(src, index_buffer, dtype, order=ASCENDING):
    res, info = C.c_void_p(), Info()
    n = src.size
    check(lib().rsx_sort_rank(src.ctypes.data, index_buffer.ctypes.data, n, dtype, index_buffer.itemsize, order,
                              C.byref(res), C.byref(info)))
    return (index_buffer[n:2 * n] if info.result_in_aux else index_buffer[:n]), info


def radix_sort_records_tagged_host(src, aux, key_offset, key_dtype, order=ASCENDING):
    """rsx_sort_records_tagged: numpy records (structured array or 2-D rows) ordered by the scalar field at key_offset."""
    res, info = C.c_void_p(), Info()
    n = len(src)
    rec_bytes = src.nbytes // max(n, 1)
    check(lib().rsx_sort_records_tagged(src.ctypes.data, aux.ctypes.data, n, rec_bytes, key_offset, key_dtype, order,
                                        C.byref(res), C.byref(info)))
    return (aux if info.result_in_aux else src), info


def radix_sort_records_tagged(src, aux, rec_bytes, key_offset, key_dtype, order=ASCENDING, stream=None):
    """rsx_sort_records_tagged_device on torch uint8 tensors holding n records of rec_bytes each."""
    require_gpu()
    _check_dev(src, aux)
    nbytes = src.numel() * src.element_size()
    if rec_bytes <= 0 or nbytes % rec_bytes:
        raise RsxError("src holds %d bytes: not a whole number of %d-byte records" % (nbytes, rec_bytes))
    if aux.numel() * aux.element_size() < nbytes:
        raise RsxError("aux is smaller than src")
    if key_offset < 0 or key_offset + DTYPE_SIZE[key_dtype] > rec_bytes:
        raise RsxError("the key does not lie inside the record")
    res, info = C.c_void_p(), Info()
    n = nbytes // rec_bytes
    check(lib().rsx_sort_records_tagged_device(src.data_ptr(), aux.data_ptr(), n, rec_bytes, key_offset, key_dtype, order,
                                               _stream_ptr(stream), C.byref(res), C.byref(info)))
    return (aux if res.value == aux.data_ptr() else src), info


def radix_sort_records_host(src, aux, keys):
    """rsx_sort_records: records (numpy structured / 2-D array rows) ordered by precomputed unsigned keys."""
    res, info = C.c_void_p(), Info()
    n = keys.size
    rec_bytes = src.nbytes // max(n, 1)
    check(lib().rsx_sort_records(src.ctypes.data, aux.ctypes.data, n, rec_bytes, keys.ctypes.data, keys.itemsize,
                                 C.byref(res), C.byref(info)))
    return (aux if info.result_in_aux else src), info
