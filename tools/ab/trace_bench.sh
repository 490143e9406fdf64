#!/bin/bash
# rocprofv3 kernel trace of a bench.py training section: bash tools/ab/trace_bench.sh <workload> <train-rays> [top-n]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tb -- python $R/bench.py --workload $1 --steps 1 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays $2 --train-steps 20 > /tmp/tb.json 2>/tmp/tb.err
python - <<'PY'
import json
d = json.loads(open('/tmp/tb.json').read().strip().splitlines()[-1])
t = d.get('train', {})
print({k: t.get(k) for k in ('ms_per_step', 'mlp_roofline_frac', 'host_enqueue_ms_per_step', 'c_abi_calls_per_step')})
PY
f=$(find /tmp/tb -name "*kernel_stats.csv" | head -1)
python - "$f" ${3:-24} <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print("   %-100s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
