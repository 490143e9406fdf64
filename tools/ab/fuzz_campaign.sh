#!/bin/bash
# The randomised sweeps as one campaign: bash tools/ab/fuzz_campaign.sh <seed> [train cases] [render cases] [ops cases] [input-grad cases]
# -> gpurun_out/fz5/{train,render,ops,ig}_<seed>.log (one line per case, flagged cases with their fp64 adjudication)
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-5001}
mkdir -p $R/gpurun_out/fz5
cd $R
python tools/ab/fuzz_train.py ${2:-500} $S > gpurun_out/fz5/train_$S.log 2>&1 &
python tools/ab/fuzz_render.py ${3:-500} $S > gpurun_out/fz5/render_$S.log 2>&1 &
python tools/ab/fuzz_ops.py ${4:-400} $S > gpurun_out/fz5/ops_$S.log 2>&1 &
python tools/ab/fuzz_input_grads.py ${5:-150} $S > gpurun_out/fz5/ig_$S.log 2>&1 &
wait
for k in train render ops ig; do echo "== $k seed $S: $(tail -1 gpurun_out/fz5/${k}_$S.log)"; grep -A1 "^BAD\|^EXC" gpurun_out/fz5/${k}_$S.log | head -40; done
