#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories of tools/profile_bench.sh.

Per kernel instantiation: calls and average duration (kernel trace), the PMC counters per dispatch, and -- for the
histogram and scatter kernels, whose algorithmic bytes per launch follow from the template arguments and n = 2^28 --
achieved algorithmic TB/s, its fraction of the 8 TB/s HBM peak, and the counter traffic beside it:
    fabric bytes = 2 * FETCH_SIZE + WRITE_SIZE   [KiB units -> bytes]
(MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

N = 1 << 28
SIZES = {"unsigned int": 4, "unsigned long long": 8, "unsigned long": 8, "unsigned short": 2, "unsigned char": 1, "rsx::NoVal": 0,
         "float": 4, "double": 8, "int": 4}


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("rsx::", "").replace("unsigned long long", "u64").replace("unsigned long", "u64").replace("unsigned int", "u32")
    name = name.replace("unsigned short", "u16").replace("unsigned char", "u8")
    return name.replace("void ", "")[:120]


def template_types(name):
    m = re.search(r"<(.*)>", name)
    if not m:
        return []
    parts, depth, cur = [], 0, ""
    for ch in m.group(1):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


def algorithmic_bytes(name):
    """n * (key read + key written + 2 * payload) for a scatter launch, n * key for a histogram launch."""
    t = template_types(re.sub(r"\(.*", "", name))
    if "rsx_scatter2_kernel" in name and len(t) >= 3:
        kin = SIZES.get(t[0], 0)
        val = SIZES.get(t[1], 0)
        tail = [x for x in t[-2:] if x in SIZES]      # <..., KTO> or <..., KTO, SEG>: the type the keys are written in
        kout = SIZES[tail[-1]] if tail else kin
        return N * (kin + kout + 2 * val)
    if ("rsx_hist_kernel" in name or "rsx_seg_hist1_kernel" in name or "rsx_seg_hist_kernel" in name) and t:
        return N * SIZES.get(t[0], 0)
    if "rsx_leaf_sort_kernel" in name and t:      # a leaf pass reads and writes every key once (rsx_hybrid.hpp)
        if t[-1] == "true" and len(t) >= 3 and t[-2] in SIZES:      # <KT, shape, CT, DENSE>: the slots hold CT-wide values
            return N * (SIZES[t[-2]] + SIZES.get(t[0], 0))
        return N * 2 * SIZES.get(t[0], 0)
    if "rsx_pass32a_kernel" in name and t:        # round 5: the level-1 pass in whole atoms: keys in, keys out (round 6: <..., OT>: what a slot holds)
        return N * (SIZES.get(t[0], 0) + SIZES.get(t[-1] if len(t) >= 6 and t[-1] in SIZES else t[0], 0))
    if "rsx_pass64a_kernel" in name and len(t) >= 2:      # round 6: the level-2 pass of 8-byte keys in whole atoms: keys in, four- or eight-byte values out
        return N * (SIZES.get(t[0], 0) + SIZES.get(t[1], 0))
    if ("rsx_pass16a_kernel" in name or "rsx_pass16_kernel" in name) and t:   # the level-2 pass: keys in, two bytes per key out
        return N * (SIZES.get(t[0], 0) + 2)
    if "rsx_leafk8_kernel" in name and t:         # 8-byte keys carried as 8-byte values: whole keys in and out
        return N * 2 * SIZES.get(t[0], 0)
    if "rsx_leafk_kernel" in name and t:          # whole 8-byte keys in and out (rsx_leaf16.hpp); SLOT32: four-byte slots in
        if t[-1] == "true":
            return N * (4 + SIZES.get(t[0], 0))
        return N * 2 * SIZES.get(t[0], 0)
    if ("rsx_leaf16_kernel" in name or "rsx_leaf16w_kernel" in name) and t:         # two-byte slots in, whole keys out (rsx_leaf16.hpp)
        return N * (2 + SIZES.get(t[0], 0))
    # route 6 (rsx_logroute.hpp) on BASELINE's cfg 3 (iv), the only configuration that takes it: 14 of the 32 equally likely bit
    # lengths are "small" (counted, written out by the fill kernel), the other 20 / 32 of the keys go through both passes and the
    # leaves as four-byte values (the library's own profile books the exact counts: bench_configs.json)
    SMALL = 14.0 / 32.0
    if "rsx_log_hist_kernel" in name:
        return N * 8
    if "rsx_log_pass1_kernel" in name:
        return int(N * 8 + (1 - SMALL) * N * 8)
    if "rsx_log_pass2_kernel" in name or "rsx_log_leaf_kernel" in name:
        return int((1 - SMALL) * N * 12)
    if "rsx_log_fill_kernel" in name:
        return int(SMALL * N * 8)
    if ("rsx_leaf_pairs_kernel" in name or "rsx_leafp_kernel" in name) and len(t) >= 2:
        # key + payload slots in; payloads out, and the keys too for pair sorts (a rank sort writes ranks only): the lower figure
        # (<KT, VT, shape, K16 = true>, round 6: the key slots hold two bytes per key)
        return N * (2 * SIZES.get(t[1], 0) + (2 if t[-1] == "true" else SIZES.get(t[0], 0)))
    return None


def main(out):
    durations = {}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (%s)" % os.path.relpath(f, out))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                print("%-100s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (
                    short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))
                durations[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
    # Launches that DO something: a route that is enqueued before the device has decided (speculative leaves, the attempts
    # of DESIGN.md 4c that the sample calls off) leaves launches of a few microseconds in the trace; averaged in, they make
    # an instantiation look faster than it is.  A working launch lasts at least a quarter of the kernel's longest one.
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        per = defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                per[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in per.items():
            work = [x for x in v if x >= 0.25 * max(v)]
            if k in durations and len(work) != len(v):
                print("   (%s: %d of %d launches do nothing; the %d working ones average %.1f us)" % (
                    short(k)[:80], len(v) - len(work), len(v), len(work), sum(work) / len(work) / 1e3))
            if k in durations:
                durations[k] = (len(work), sum(work) / len(work))
    counters = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(lambda: defaultdict(int))
            vals = defaultdict(lambda: defaultdict(list))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k in vals:
                for cn, v in vals[k].items():   # (the same rule for the counters: launches that do nothing count nothing)
                    work = [x for x in v if x >= 0.25 * max(v)] if max(v) > 0 else v
                    acc[k][cn] = sum(work)
                    calls[k][cn] = len(work)
            print("== counters (%s): per-dispatch averages" % os.path.relpath(f, out))
            for k in sorted(acc):
                if "rsx_" not in k:
                    continue
                print("  " + short(k))
                for cn in sorted(acc[k]):
                    avg = acc[k][cn] / calls[k][cn]
                    counters[k][cn] = avg
                    print("      %-24s %16.1f   (over %d dispatches)" % (cn, avg, calls[k][cn]))
    print("== roofline per instantiation (algorithmic bytes at n = 2^28 / average duration; peak 8000 GB/s)")
    table = []
    for name, (calls, avg_ns) in sorted(durations.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        ab = algorithmic_bytes(name)
        if ab is None or avg_ns <= 0:
            continue
        if avg_ns < 30000 or ab / avg_ns > 8000.0:   # (a launch that finds nothing to do -- the list launch behind
            continue            #  rsx_leaf16_kernel, the instantiation for the other carried type or slot width: 65536 workgroups
                                #  that return take ~30 us -- moves no bytes: no roofline row, and nothing above the peak)
        key = next((k for k in counters if re.sub(r"\(.*", "", k) == re.sub(r"\(.*", "", name)), None)
        fetch = counters.get(key, {}).get("FETCH_SIZE")
        write = counters.get(key, {}).get("WRITE_SIZE")
        traffic = (2 * fetch + write) * 1024 if fetch is not None and write is not None else None
        lds_c, lds_a = counters.get(key, {}).get("SQ_LDS_BANK_CONFLICT"), counters.get(key, {}).get("SQ_LDS_IDX_ACTIVE")
        gbps = ab / avg_ns
        row = {"kernel": short(name), "calls": calls, "avg_us": avg_ns / 1e3, "algorithmic_bytes": ab, "achieved_GBps": gbps,
               "frac_of_8TBps": gbps / 8000.0, "fabric_bytes_per_launch": traffic,
               "traffic_over_algorithmic": traffic / ab if traffic else None,
               "lds_bank_conflict_share": lds_c / lds_a if lds_c is not None and lds_a else None}
        table.append(row)
        print("%-100s %5d x %9.1f us  %6.0f GB/s  frac %.3f  traffic %s  LDS conflicts %s" % (
            row["kernel"], calls, row["avg_us"], gbps, row["frac_of_8TBps"],
            "%.2fx" % row["traffic_over_algorithmic"] if traffic else "n/a",
            "%.0f%%" % (100 * row["lds_bank_conflict_share"]) if row["lds_bank_conflict_share"] is not None else "n/a"))
    with open(os.path.join(out, "roofline_table.json"), "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
