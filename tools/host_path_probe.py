"""PCIe-inclusive time of the host-pointer entry point rsx_sort() -- what the C++ template wrapper calls for
radix_sort(src, aux, n) on host arrays (radix_experiment.cpp:203-206 times exactly that call) -- at 10^4, 4*10^7 and 2^28
u32 keys, with the caller's buffers reused across calls and freshly allocated, with RSX_HOST_REGISTER off and on (a
process each: the switch is read once).  SURVEY.md 8 f3.

    python tools/host_path_probe.py        -> the table; gpurun_out/host_path.txt
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import radix_sorting_amd as rsa
    rows = []
    rng = np.random.default_rng(1)
    big = rng.integers(0, 1 << 32, size=1 << 28, dtype=np.uint32)
    warm = big[:1 << 20].copy()
    rsa.radix_sort_host(warm, np.zeros_like(warm), rsa.U32)
    for n in (10 ** 4, 4 * 10 ** 7, 1 << 28):
        a = big[:n]
        src, aux = a.copy(), np.zeros(n, dtype=np.uint32)
        reused = []
        for rep in range(5):
            np.copyto(src, a)
            t0 = time.perf_counter()
            res, info = rsa.radix_sort_host(src, aux, rsa.U32)
            reused.append(time.perf_counter() - t0)
        assert bool(np.all(res[1:] >= res[:-1]))
        fresh = []
        # (with RSX_HOST_REGISTER=1 the library keeps the buffers it has seen page-locked: a caller must not free them --
        # the fresh-buffer case is exactly what that mode is not for, and it is not run in it)
        for rep in range(0 if os.environ.get("RSX_HOST_REGISTER") == "1" else 3):
            s2, a2 = a.copy(), np.zeros(n, dtype=np.uint32)
            t0 = time.perf_counter()
            res, info = rsa.radix_sort_host(s2, a2, rsa.U32)
            fresh.append(time.perf_counter() - t0)
            del s2, a2
        t0 = time.perf_counter()
        res2, info2 = rsa.radix_sort_host(res, aux if res is src else src, rsa.U32)     # already sorted: H2D + histogram only
        t_sorted = time.perf_counter() - t0
        rows.append({"n": n, "reused_first_ms": reused[0] * 1e3, "reused_ms": sorted(reused[1:])[len(reused[1:]) // 2] * 1e3,
                     "fresh_ms": sorted(fresh)[len(fresh) // 2] * 1e3 if fresh else float("nan"), "sorted_input_ms": t_sorted * 1e3})
    print("ROWS " + json.dumps(rows), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    lines = []
    for mode in ("0", "1"):
        env = dict(os.environ, RSX_HOST_REGISTER=mode)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=1800)
        row = [l for l in p.stdout.splitlines() if l.startswith("ROWS ")]
        if p.returncode != 0 or not row:
            raise SystemExit("child failed: " + p.stdout + p.stderr)
        for r in json.loads(row[0][5:]):
            lines.append("RSX_HOST_REGISTER=%s  n = %10d u32 keys: reused buffers %9.3f ms (first call %9.3f ms), fresh buffers %9.3f ms, "
                         "already sorted %9.3f ms  -> %.2f Gkeys/s PCIe-inclusive" % (mode, r["n"], r["reused_ms"], r["reused_first_ms"],
                                                                                   r["fresh_ms"], r["sorted_input_ms"],
                                                                                   r["n"] / r["reused_ms"] / 1e6))
    text = "\n".join(lines)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "host_path.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
