#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs produced by tools/profile_bench.sh: per-kernel stats and PMC sums/averages."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("rsx::", "").replace("unsigned long long", "u64").replace("unsigned int", "u32")
    return name.replace("void ", "")[:110]


def main(out):
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (%s)" % os.path.relpath(f, out))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                print("%-72s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (
                    short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["Percentage"]))
    for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            calls = defaultdict(lambda: defaultdict(int))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    calls[k][row["Counter_Name"]] += 1
            print("== counters (%s): per-dispatch averages" % os.path.relpath(f, out))
            for k in sorted(acc):
                if "rsx_" not in k:
                    continue
                print("  " + k)
                for cn in sorted(acc[k]):
                    print("      %-24s %16.1f   (over %d dispatches)" % (cn, acc[k][cn] / calls[k][cn], calls[k][cn]))


if __name__ == "__main__":
    main(sys.argv[1])
