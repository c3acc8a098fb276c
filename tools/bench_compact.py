"""SURVEY.md 8 f4 (README.md:716-758, key compaction) measured: 2^28-key rank sorts (f32 / u32 keys -> u32 ranks) with the
library's RSX_COMPACT_BITS switch off and on (a process each: the switch is read once).  Inputs: keys whose varying bits
are sparse over the bytes (where compaction saves passes) and BASELINE.json's cfg 4 inputs (where it cannot).

    python tools/bench_compact.py            -> gpurun_out/bench_compact.json (both modes, rows side by side)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, K, W = 1 << 28, 5, 2
CASES = [("u32 & 0x0F0F0F0F (16 varying bits in 4 bytes)", "U32", 0x0F0F0F0F),
         ("u32 & 0x00FF0F0F (16 bits in 3 bytes)", "U32", 0x00FF0F0F),
         ("u32 & 0x01010101 (4 bits in 4 bytes)", "U32", 0x01010101),
         ("cfg4 (i) f32 random bits", "F32", 0xFFFFFFFF),
         ("cfg4 (iii) f32 & 0xFFF000FF", "F32", 0xFFF000FF)]


def child():
    import torch
    import radix_sorting_amd as rsa
    rsa.require_gpu()
    ib = torch.empty(2 * N, dtype=torch.int32, device="cuda")
    rows = []
    for name, dt, mask in CASES:
        code = getattr(rsa, dt)
        batches = []
        for i in range(K + W):
            t = torch.empty(N, dtype=torch.int32, device="cuda")
            rsa.fill_splitmix(t, seed=60 + i, mask=mask)
            batches.append(t)
        for i in range(W):
            rsa.radix_sort_rank(batches[i], ib, dtype=code)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            _, info = rsa.radix_sort_rank(batches[i], ib, dtype=code)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / K
        rows.append({"input": name, "kept_columns": info.ncols, "ms_per_sort": dt_s * 1e3, "Gkeys_per_s": N / dt_s / 1e9})
        del batches
        torch.cuda.empty_cache()
    print("ROWS " + json.dumps(rows), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child()
    out = {}
    for mode in ("0", "1"):
        env = dict(os.environ, RSX_COMPACT_BITS=mode)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=1200)
        line = [l for l in p.stdout.splitlines() if l.startswith("ROWS ")]
        if p.returncode != 0 or not line:
            raise SystemExit("child failed: " + p.stdout + p.stderr)
        out["RSX_COMPACT_BITS=" + mode] = json.loads(line[0][5:])
    table = []
    for a, b in zip(out["RSX_COMPACT_BITS=0"], out["RSX_COMPACT_BITS=1"]):
        table.append({"input": a["input"], "kept_columns": a["kept_columns"], "ms_plain": a["ms_per_sort"], "ms_compacted": b["ms_per_sort"],
                      "speedup": a["ms_per_sort"] / b["ms_per_sort"]})
        print("%-50s P=%d  plain %.3f ms  compacted %.3f ms  x%.2f" % (a["input"], a["kept_columns"], a["ms_per_sort"], b["ms_per_sort"],
                                                                     a["ms_per_sort"] / b["ms_per_sort"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(table, open(os.path.join(ROOT, "gpurun_out", "bench_compact.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
