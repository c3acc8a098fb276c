#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/libref.so).

Run in the build container only (needs /root/reference to build oracle/_ref).
What is committed is data: generator parameters, FNV-1a-64 hashes of inputs and
of the reference's outputs, which buffer the reference returned, and the
captured test_int input.  No reference source is copied.

The rows reproduce SURVEY.md section 4's known-answer table; the hashes SURVEY
printed are kept next to the regenerated ones and must agree.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

FULL = 0xFFFFFFFFFFFFFFFF

# (dtype, n, seed, mask, survey_in, survey_out, survey_in_aux)
SCALAR_ROWS = [
    (ol.U32, 0, 1, FULL, "cbf29ce484222325", "cbf29ce484222325", 0),
    (ol.U32, 1, 1, FULL, "a17a51b6c98cc32d", "a17a51b6c98cc32d", 0),
    (ol.U32, 2, 1, FULL, "82f1a798c20f8b37", "718ff5fc393e7dab", 0),
    (ol.U32, 255, 1, FULL, "7f6d3348b643441d", "990323ec6cce5419", 0),
    (ol.U32, 256, 1, FULL, "dccef8978b21416c", "8d1e60ca60a5a348", 0),
    (ol.U32, 65535, 1, FULL, "53db29ee26b2868c", "2c05ee623b0a2314", 0),
    (ol.U32, 65536, 1, FULL, "41b6d98a643f890c", "f2f8c05c651c6d58", 0),
    (ol.U32, 1048576, 1, FULL, "3cbb776a58ec99f5", "9684760268fe1ff5", 0),
    (ol.U32, 1048576, 2, 0x00FFFFFF, "c53c0d6b23792272", "320e712a9aa0f46a", 1),
    (ol.U32, 1048576, 2, 0xFF0000FF, "07c490d88e23e77b", "976b0e451502b5f3", 0),
    (ol.U32, 1048576, 2, 0x0000FF00, "6be54720ed570fea", "6be1e01eb2f8efaa", 1),
    (ol.U64, 255, 3, FULL, "6cfaf96f3272aa2e", "5f4c3645b367953e", 0),
    (ol.U64, 65536, 3, FULL, "d401532f54bb5468", "fbaf00c9bc11df14", 0),
    (ol.U64, 1048576, 3, FULL, "98b9821c0c0e222d", "2ce45a0c6f698f69", 0),
    (ol.U64, 1048576, 3, 0x000000FFFFFFFFFF, "d4f239655e1730fd", "3c5255d5ed3413a1", 1),
    (ol.U64, 1048576, 3, 0x00000000FFFFFFFF, "5c8f71ffa9344a24", "5d34ede8f60e8de8", 0),
    (ol.I32, 255, 4, FULL, "3297e2b46701bddd", "16778852b91a4879", 0),
    (ol.I32, 65536, 4, FULL, "904cbd4e8ff58049", "270a3a3f5cce0dd9", 0),
    (ol.I32, 1048576, 4, FULL, "296236c57b5a996a", "27dcf43da95df4da", 0),
    (ol.I64, 65536, 5, FULL, "292e9cf47b7b1b5b", "0eee24b97f4e494b", 0),
    (ol.I64, 1048576, 5, FULL, "d44a61cea9af5b7f", "94d08db6456a89f7", 0),
    (ol.F32, 255, 6, FULL, "ea15b76dc180ef51", "a68d5ed114569dd9", 0),
    (ol.F32, 65536, 6, FULL, "4110941df62e42d5", "495b070186f83b45", 0),
    (ol.F32, 1048576, 6, FULL, "2454326489a88896", "3adc2d62a141856e", 0),
    (ol.F64, 65536, 7, FULL, "ecca398c4d3d1aa2", "05fb9da72a654a7e", 0),
    (ol.F64, 1048576, 7, FULL, "3023587469a7c901", "e875fb3443bafbc9", 0),
    (ol.U16, 1048576, 8, FULL, "ba42322d698283b5", "9a008dd44063262d", 0),
    (ol.U16, 1048576, 8, 0x00FF, "30c671f5f7e458c3", "e013a33d57488f23", 1),
    (ol.U8, 1048576, 9, FULL, "8e72d16bfa4aec76", "9527267aa8fce5b2", 1),
    (ol.I16, 1048576, 10, FULL, "9747993b38f1f308", "9f4c8f2e02f08c64", 0),
    (ol.I8, 1048576, 11, FULL, "8a65595023d149e2", "2f0ca9d958f4d6a8", 1),
]
# extra rows (not in SURVEY) so every dtype also has a descending and a small-n pin
EXTRA_ROWS = [(dt, n, 20 + dt, FULL, order)
              for dt in range(10) for n in (3, 1000, 70000) for order in (ol.ASC, ol.DESC)]

# rank rows: (dtype, n, seed, mask, survey_ranks_fnv or None, reference_is_correct)
RANK_ROWS = [
    (ol.U8, 1048576, 9, FULL, "52066ff778cb5161", True),
    (ol.U32, 1048576, 2, 0xFF, "0dbd04b65ae39a7d", True),
    (ol.U32, 1048576, 2, 0xFF00, "20901103796c4d55", True),
    (ol.U32, 1048576, 1, FULL, "b8c2901067ab1435", False),
    (ol.F32, 1048576, 6, FULL, "50321446c98fce35", False),
]


# caller-supplied histogram rows: (dtype, n, seed, mask, presorted)
HIST_ROWS = [(dt, n, 40 + dt, mask, pre)
             for dt in range(10)
             for (n, mask, pre) in ((200, FULL, 0), (3000, FULL, 0), (100000, FULL, 0), (100000, 0x00FFFF00FF00FFFF, 0),
                                    (5000, FULL, 1), (300, 0xFF, 0))]


def hx(v):
    return "%016x" % v


def main():
    ol.build_oracle()
    ref = ol.ref()
    assert ref is not None, "needs /root/reference (build container only)"
    out = {"generator": "tools/gen_golden.py", "source": "oracle/_ref/libref.so (reference headers compiled in place)",
           "prng": "splitmix64, one call per element, & mask, low sizeof(T) bytes", "hash": "FNV-1a-64 over result bytes",
           "scalar": [], "scalar_extra": [], "rank": [], "kv": [], "test_int": {}, "hist_post": []}

    for dt, n, seed, mask, s_in, s_out, s_aux in SCALAR_ROWS:
        a = ol.splitmix_fill(n, dt, seed, mask)
        res, in_aux = ol.ref_sort(a, dt)
        row = {"dtype": ol.DTYPE_NAMES[dt], "dtype_code": dt, "n": n, "seed": seed, "mask": hx(mask), "order": 0,
               "fnv_in": hx(ol.fnv1a64(a)), "fnv_out": hx(ol.fnv1a64(res)), "result_in_aux": in_aux,
               "survey_fnv_in": s_in, "survey_fnv_out": s_out, "survey_result_in_aux": s_aux}
        assert row["fnv_in"] == s_in and row["fnv_out"] == s_out and in_aux == s_aux, row
        out["scalar"].append(row)

    for dt, n, seed, mask, order in EXTRA_ROWS:
        a = ol.splitmix_fill(n, dt, seed, mask)
        res, in_aux = ol.ref_sort(a, dt, order)
        out["scalar_extra"].append({"dtype": ol.DTYPE_NAMES[dt], "dtype_code": dt, "n": n, "seed": seed,
                                    "mask": hx(mask), "order": order, "fnv_in": hx(ol.fnv1a64(a)),
                                    "fnv_out": hx(ol.fnv1a64(res)), "result_in_aux": in_aux})

    for dt, n, seed, mask, s_fnv, ref_ok in RANK_ROWS:
        a = ol.splitmix_fill(n, dt, seed, mask)
        rres, rhalf, _ = ol.ref_rank(a, dt, 4)
        target = ol.stable_argsort_by_kdf(a, dt).astype(np.uint32)
        row = {"dtype": ol.DTYPE_NAMES[dt], "dtype_code": dt, "n": n, "seed": seed, "mask": hx(mask),
               "idx_bytes": 4, "fnv_stable_argsort": hx(ol.fnv1a64(target)),
               "fnv_reference_output": hx(ol.fnv1a64(rres)), "reference_result_half": rhalf,
               "reference_is_correct": bool(np.array_equal(rres, target)), "survey_fnv": s_fnv}
        assert row["fnv_stable_argsort"] == s_fnv, row
        assert row["reference_is_correct"] == ref_ok, row
        out["rank"].append(row)

    # key + payload pin: struct {float k; uint32_t v;}, n = 2^20, state 12, k bits & 0xFFF000FF, v = i
    n = 1048576
    k = ol.splitmix_fill(n, ol.U32, 12, 0xFFF000FF)
    rec = np.empty((n, 2), dtype=np.uint32)
    rec[:, 0] = k
    rec[:, 1] = np.arange(n, dtype=np.uint32)
    src = rec.copy()
    aux = np.zeros_like(src)
    r = ref.ref_sort_kv(ol.ptr(src), ol.ptr(aux), n, ol.F32, 4, 0)
    res = aux if r else src
    kv = {"key_dtype": "float", "payload": "uint32_t index", "n": n, "seed": 12, "mask": hx(0xFFF000FF),
          "fnv_in": hx(ol.fnv1a64(rec)), "fnv_out_aos": hx(ol.fnv1a64(res)),
          "fnv_out_payloads": hx(ol.fnv1a64(np.ascontiguousarray(res[:, 1]))),
          "fnv_out_keys": hx(ol.fnv1a64(np.ascontiguousarray(res[:, 0]))), "result_in_aux": r,
          "survey": {"fnv_in": "71626d7dc023da7c", "fnv_out_aos": "e7ac645507e9b9d8",
                     "fnv_out_payloads": "63592a75c6faac31"}}
    assert kv["fnv_in"] == kv["survey"]["fnv_in"] and kv["fnv_out_aos"] == kv["survey"]["fnv_out_aos"]
    assert kv["fnv_out_payloads"] == kv["survey"]["fnv_out_payloads"]
    assert np.array_equal(res[:, 1], ol.stable_argsort_by_kdf(k, ol.F32).astype(np.uint32))
    out["kv"].append(kv)

    # test_int fixture (radix_tests.cpp:179-207)
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "gen_test_int")
        subprocess.run(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tools", "gen_test_int.cpp")], check=True)
        raw = subprocess.run([exe], check=True, capture_output=True).stdout
    ti = np.frombuffer(raw, dtype=np.uint32).copy()
    assert ti.size == 50000
    ti.tofile(os.path.join(ROOT, "tests", "golden", "test_int_input.bin"))
    asc, asc_aux = ol.ref_sort(ti, ol.I32, 0)
    desc, desc_aux = ol.ref_sort(asc, ol.I32, 1)
    out["test_int"] = {"file": "test_int_input.bin", "n": 50000, "fnv_in": hx(ol.fnv1a64(ti)),
                       "fnv_ascending": hx(ol.fnv1a64(asc)), "fnv_descending": hx(ol.fnv1a64(desc)),
                       "ascending_in_aux": asc_aux, "descending_in_aux": desc_aux,
                       "survey": {"fnv_in": "0de97f5a5d33246c", "fnv_ascending": "9e66d085f1dd25c0",
                                  "fnv_descending": "7a6a54dc00ad17bc"}}
    assert out["test_int"]["fnv_in"] == "0de97f5a5d33246c"
    assert out["test_int"]["fnv_ascending"] == "9e66d085f1dd25c0"
    assert out["test_int"]["fnv_descending"] == "7a6a54dc00ad17bc"

    # rs_sort_main with a caller-supplied Hist (radix_sort.hpp:28-33): what the reference leaves in the histogram storage
    for dt, n, seed, mask, presorted in HIST_ROWS:
        a = ol.splitmix_fill(n, dt, seed, mask)
        if presorted:
            a = a[ol.stable_argsort_by_kdf(a, dt)]
        hv = ol.hvt_bytes_for(n)
        res, in_aux, hist = ol.ref_sort_main_hist(a, dt, hv)
        out["hist_post"].append({"dtype": ol.DTYPE_NAMES[dt], "dtype_code": dt, "n": n, "seed": seed, "mask": hx(mask),
                                 "presorted": presorted, "hvt_bytes": hv, "fnv_in": hx(ol.fnv1a64(a)),
                                 "fnv_out": hx(ol.fnv1a64(res)), "result_in_aux": in_aux,
                                 "fnv_hist_u64": hx(ol.fnv1a64(hist))})

    path = os.path.join(ROOT, "tests", "golden", "kat_table.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, "rows:", len(out["scalar"]), len(out["scalar_extra"]), len(out["rank"]), len(out["hist_post"]))


if __name__ == "__main__":
    main()
