#!/bin/bash
# per-kernel times of the one-call training step at small ray counts (rocprofv3 kernel trace): bash tools/ab/lat_train_trace.sh "64,256,800" [env...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rays in ${1//,/ }; do
  rm -rf /tmp/tr_$rays
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$rays -- python $R/tools/ab/train_points.py --rays $rays --steps 20 > /tmp/tr_$rays.log 2>&1
  tail -1 /tmp/tr_$rays.log
  f=$(find /tmp/tr_$rays -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("   %-86s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
