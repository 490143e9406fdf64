#!/bin/bash
R=$GRAFT_REPO_ROOT
cp $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so /tmp/orig.so
for lib in $LIBS; do
  cp $R/$lib $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so
  python $R/bench.py --steps 20 --warmup 3 --no-pmc --no-alt --cpu-rays 0 --train-rays 0 --points= 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', l['ms_per_step'], l['roofline']['avg_launch_ms'], l['roofline']['frac'])"
done
cp /tmp/orig.so $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so
