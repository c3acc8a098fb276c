#!/bin/sh
# Counterpart of the reference's bench.sh (bench.sh:6-18): a simple performance report, mainly useful to make sure no major
# regressions sneak in.  Writes bench-<date>.txt in the current directory: uname, git revision, lscpu, the GPU
# (rocminfo | grep gfx), four runs of `radix` on the whole key file (the reference varies use_mmap / use_huge; they are
# accepted and echoed here -- the sort runs in HBM), and `radix_bench --device --verify`.
# Usage: tools/report.sh [device index] [further radix_bench arguments, e.g. --min-time 0.05]
HERE=$(cd "$(dirname "$0")" && pwd)
DEV=${1:-0}
[ $# -gt 0 ] && shift
REPFILE=bench-$(date +"%Y-%m-%d.%s").txt
uname -a >>$REPFILE
(cd "$HERE/.." && git rev-parse HEAD 2>/dev/null || echo "no git revision") >>$REPFILE
lscpu >>$REPFILE
(rocminfo 2>/dev/null | grep -i -E "gfx|Marketing Name|Compute Unit" | sort | uniq -c) >>$REPFILE
# (built only if missing: a stale-looking time stamp on a fresh copy of the tree would rebuild the library, minutes of hipcc)
if [ ! -x "$HERE/radix" ] || [ ! -x "$HERE/radix_bench" ] || [ ! -f "$HERE/../radix_sorting_amd/librsx.so" ]; then
	make -s -C "$HERE/.." lib cli >/dev/null
fi
echo "Running benchmarks. Writing result to $REPFILE"
# warm-up
"$HERE/radix" 0 0 0 --device $DEV >/dev/null
"$HERE/radix" 0 0 0 --device $DEV >>$REPFILE 2>&1
"$HERE/radix" 0 1 0 --device $DEV >>$REPFILE 2>&1
"$HERE/radix" 0 0 1 --device $DEV >>$REPFILE 2>&1
"$HERE/radix" 0 1 1 --device $DEV >>$REPFILE 2>&1
"$HERE/radix_bench" --device $DEV --verify "$@" >>$REPFILE 2>&1
echo "$REPFILE"
