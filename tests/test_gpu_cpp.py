"""The C++ template surface (include/radix_sort.hpp, radix_sort_rank.hpp) on the GPU.

tests/cpp/dropin_check   our own program: exact stable order / returned pointer / rank checks through the templates.
oracle/_ref/radix_tests_dropin   the reference's UNMODIFIED radix_tests.cpp compiled against include/ and linked
                         with librsx.so (built where /root/reference exists; it travels to the GPU box as a binary).
"""
import json
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
    """Own C++ checks through the templates: exact stable order, returned pointers, free-function KeyFuncs (a9), and
    rs_sort_main / rs_sort_rank with caller-supplied histogram storage (a11) -- whose final contents must hash to what
    the REAL rs_sort_main left in a std::vector<HVT> (tests/golden/kat_table.json, hist_post)."""
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_check")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "cpp"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "dropin_check OK" in out.stdout, out.stdout + out.stderr
    with open(os.path.join(ROOT, "tests", "golden", "kat_table.json")) as f:
        golden = json.load(f)["hist_post"]
    want = {(r["dtype_code"], r["n"], r["seed"], r["mask"], r["presorted"], r["hvt_bytes"]):
            (r["fnv_out"], r["result_in_aux"], r["fnv_hist_u64"]) for r in golden}
    seen = 0
    for line in out.stdout.splitlines():
        if not line.startswith("HIST "):
            continue
        _, dt, n, seed, mask, pre, hvt, fnv_out, in_aux, fnv_hist = line.split()
        key = (int(dt), int(n), int(seed), mask, int(pre), int(hvt))
        assert key in want, line
        assert want[key] == (fnv_out, int(in_aux), fnv_hist), (line, want[key])
        seen += 1
    assert seen == len(golden) == 60


def test_reference_tests_run_unmodified_on_our_headers():
    exe = os.path.join(ROOT, "oracle", "_ref", "radix_tests_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/radix_tests_dropin not built (needs /root/reference at build time)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "All tests OK." in out.stdout, out.stdout + out.stderr
    for line in ("Sorting struct sortrec... OK", "Sorting struct sortrec** (reverse)... OK", "Sorting float[]... OK",
                 "Rank sorting struct sortrec... OK"):
        assert line in out.stdout


def _cli(name):
    exe = os.path.join(ROOT, "tools", name)
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "cli"], check=True)
    return exe


@pytest.mark.parametrize("ktype,mask,where", [("uint32_t", None, "source"), ("uint32_t", "00FFFFFF", "auxilary"),
                                               ("uint64_t", "FFFFFFFFFF", "auxilary"), ("float", None, "source"),
                                               ("int32_t", None, "source"), ("double", None, "source"), ("uint8_t", None, "auxilary")])
def test_radix_cli_counterpart(tmp_path, ktype, mask, where):
    """tools/radix prints the reference's lines (radix_experiment.cpp:241-285) and sorts correctly."""
    args = [_cli("radix"), "1000000", "0", "0", ktype] + ([mask] if mask else [])
    out = subprocess.run(args, capture_output=True, text=True, timeout=600, cwd=tmp_path)   # no key file there: generated
    assert out.returncode == 0, out.stdout + out.stderr
    n = {"uint8_t": 1000000, "uint64_t": 1000000, "double": 1000000}.get(ktype, 1000000)
    for needle in ("src='40M_32bit_keys.dat', entries=1000000, use_mmap=0, use_huge=0, type='%s'" % ktype,
                   "Sorting %d entries..." % n, "Verifying sort... Forward sorted OK.", "[...]",
                   "Sorted %d entries in " % n, "Result in the %s buffer" % where):
        assert needle in out.stdout, (needle, out.stdout)
    if mask:
        assert "Applying value mask to input." in out.stdout
    dump = [l for l in out.stdout.splitlines() if len(l) > 10 and l[8] == ":" and l[:8].isdigit()]
    assert len(dump) == 20 and dump[0].startswith("00000000: ") and dump[-1].startswith("%08d: " % (n - 1))


def test_radix_bench_counterpart(tmp_path):
    """tools/radix_bench prints the reference's row names and columns for the sizes 1 .. 4*10^7, and with --verify every
    radix row's output -- the 40 M-key rows of BASELINE.json configs[0] included -- is compared with std::sort / the stable
    argsort outside the timed region (what radix_experiment.cpp:137-174 does for `radix`)."""
    out = subprocess.run([_cli("radix_bench"), "--min-time", "0.05", "--filter", "radix_sort", "--device", "0", "--verify"],
                         capture_output=True, text=True, timeout=1200, cwd=tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [l.split() for l in out.stdout.splitlines() if l.startswith("FSu32/")]
    names = [r[0] for r in rows]
    for kind, ref in (("radix_sort", "std::sort of the same keys"), ("radix_sort_rank", "the stable argsort"),
                      ("radix_sort_device", "std::sort of the same keys")):
        for n in (1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 40000000):
            assert "FSu32/%s/%d" % (kind, n) in names
            assert "verified: FSu32/%s/%d == %s" % (kind, n, ref) in out.stdout
    assert "DIFFERS" not in out.stdout
    assert "KeyRate" in out.stdout and "bytes_per_second" in out.stdout
    bad = subprocess.run([_cli("radix_bench"), "--device", "99"], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert bad.returncode == 3 and "no such HIP device" in bad.stderr


def test_radix_cli_device_flag(tmp_path):
    ok = subprocess.run([_cli("radix"), "100000", "0", "0", "uint32_t", "--device", "0"], capture_output=True, text=True, timeout=300,
                        cwd=tmp_path)
    assert ok.returncode == 0 and "Verifying sort... Forward sorted OK." in ok.stdout, ok.stdout + ok.stderr
    bad = subprocess.run([_cli("radix"), "--device", "99", "100000"], capture_output=True, text=True, timeout=120, cwd=tmp_path)
    assert bad.returncode == 3 and "no such HIP device" in bad.stdout


def test_report_script(tmp_path):
    """tools/report.sh, the counterpart of the reference's bench.sh:6-18: uname, git revision, lscpu, the GPU, four `radix`
    runs on the whole key file and `radix_bench --device --verify`, written to bench-<date>.txt."""
    out = subprocess.run(["sh", os.path.join(ROOT, "tools", "report.sh"), "0", "--min-time", "0.05"], capture_output=True, text=True, timeout=1800, cwd=tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    name = out.stdout.strip().splitlines()[-1]
    text = open(os.path.join(str(tmp_path), name)).read()
    assert text.count("Sorted 40000000 entries in ") == 4 and text.count("Forward sorted OK.") == 4
    assert "Linux" in text and "gfx950" in text and "CPU(s):" in text
    assert "verified: FSu32/radix_sort/40000000 == std::sort of the same keys" in text
    assert "verified: FSu32/radix_sort_device/40000000 == std::sort of the same keys" in text
