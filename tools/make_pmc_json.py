#!/usr/bin/env python3
"""profiles/pmc_scatter.json from a roofline table of tools/summarize_prof.py: the counter traffic of the scatter kernel per
launch (what bench.py reports as roofline.traffic), with the commit the counters were measured at.
Usage: tools/make_pmc_json.py profiles/r03/bench/roofline_table.json <round> <commit>"""
import json
import sys

table, rnd, commit = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = json.load(open(table))
sc = [r for r in rows if "rsx_scatter2_kernel<u32, NoVal" in r["kernel"] and r["fabric_bytes_per_launch"]]
# the timed steps of bench.py make both their passes with the SEG instantiation (last template argument true: passes into
# slots); the plain one in the same trace belongs to the RSX_NO_HYBRID comparison bench.py runs afterwards
seg = [r for r in sc if r["kernel"].rstrip().endswith("u32, true>")]   # (the full-width one: bench.py's `roofline` object)
main = max(seg or sc, key=lambda r: r["calls"])
out = {
    "round": rnd,
    "measured_at_commit": commit,
    "kernel": "rsx_scatter2_kernel<u32,NoVal,u32> (Sc2Cfg 16 waves, 32 Ki-key tile)",
    "instantiation": main["kernel"],
    "workload": "bench.py --steps 5 --warmup 1, 2^28 u32 keys per launch",
    "source": "%s (tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, per-dispatch "
              "averages over %d dispatches; fabric bytes = 2 x FETCH_SIZE + WRITE_SIZE in KiB units, the gfx950 correction of "
              "MI355X_MICROARCH.md's HBM section -- FETCH_SIZE reports half of the bytes of wide coalesced reads, checked in the "
              "same run on rsx_hist_kernel, which reads exactly 2^30 bytes per launch)" % (table, main["calls"]),
    "hbm_bytes_per_launch": main["fabric_bytes_per_launch"],
    "algorithmic_bytes_per_launch": main["algorithmic_bytes"],
    "other_kernels_of_the_step": {r["kernel"]: {"fabric_bytes_per_launch": r["fabric_bytes_per_launch"],
                                                "algorithmic_bytes": r["algorithmic_bytes"], "avg_us_under_rocprofv3": r["avg_us"]}
                                  for r in rows if r is not main and r["fabric_bytes_per_launch"]},
    "note": "one read and one write of the keys per pass; the 5 % above the algorithmic bytes are the status words of the "
            "look-back chain and the partial 64-byte atoms at run boundaries",
}
json.dump(out, open("profiles/pmc_scatter.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:600])
