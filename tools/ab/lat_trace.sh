#!/bin/bash
# per-kernel times of forward + backward of one net at the given sample counts, one size per process: bash tools/ab/lat_trace.sh 4096,16384
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in ${1//,/ }; do
  rm -rf /tmp/lt_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt_$n -- python $R/tools/ab/lat_timing.py bwd $n > /tmp/lt_$n.log 2>&1
  tail -1 /tmp/lt_$n.log
  f=$(find /tmp/lt_$n -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("   %-86s calls %5s avg %9.1f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
