#!/bin/bash
# Timeline of ONE one-call training step (kernel, queue, start offset, duration) from rocprofv3's kernel trace:
#   bash tools/ab/step_timeline.sh <rays> [workload] [steps]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rays=${1:-64}; wl=${2:-nerf}; steps=${3:-30}
rm -rf /tmp/tl_$rays
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$rays -- python $R/tools/ab/train_loop.py $rays $wl $steps > /tmp/tl_$rays.log 2>&1
tail -2 /tmp/tl_$rays.log
f=$(find /tmp/tl_$rays -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete step: from the last-but-one adam_kernel's end to the last adam_kernel's end
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-2] + 1, adam[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = {}
print("   step of %d kernels, %.1f us from the first kernel's start to Adam's end" % (b - a, (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].replace("snerf::", "").replace("void ", "")
    name = name[:name.find("(")] if "(" in name else name
    print("   q%-3s +%8.1f us  %7.1f us  grid %-8s wg %-5s %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name[:70]))
PY
