#!/usr/bin/env python3
"""profiles/pmc_kernels.json from a roofline table of tools/summarize_prof.py: the counter traffic per launch of every kernel
of a bench.py step (what bench.py reports as roofline.traffic for the step's dominant kernel), with the commit the counters
were measured at.
Usage: tools/make_pmc_json.py profiles/r04/bench/roofline_table.json <round> <commit>"""
import json
import sys

table, rnd, commit = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = json.load(open(table))
out = {
    "round": rnd,
    "measured_at_commit": commit,
    "workload": "bench.py --steps 20 --warmup 3, 2^28 u32 keys per launch",
    "source": "%s (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, per-dispatch "
              "averages over the working dispatches; fabric bytes = 2 x FETCH_SIZE + WRITE_SIZE in KiB units, the gfx950 "
              "correction of MI355X_MICROARCH.md's HBM section -- FETCH_SIZE reports half of the bytes of wide coalesced reads, "
              "checked in the same run on rsx_hist_kernel, which reads exactly 2^30 bytes per launch)" % table,
    "kernels": {r["kernel"]: {"hbm_bytes_per_launch": r["fabric_bytes_per_launch"], "algorithmic_bytes_per_launch": r["algorithmic_bytes"],
                              "avg_us_under_rocprofv3": r["avg_us"], "calls": r["calls"],
                              "lds_bank_conflict_share": r["lds_bank_conflict_share"]}
                for r in rows if r["fabric_bytes_per_launch"]},
}
json.dump(out, open("profiles/pmc_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
