#!/bin/bash
# A/B of library variants (tools/ubench/build_variant.sh) on the fp32 training step: per-kernel averages from rocprofv3.
# LIBS: space-separated .so paths relative to the repo root; each is copied over libsmplnerf_hip.so of the box's scratch copy.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so /tmp/orig.so
for lib in $LIBS; do
  cp $R/$lib $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so
  rm -rf /tmp/abprof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof -- python $R/bench.py --steps 1 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays ${RAYS:-4096} --train-steps 10 > /tmp/ab.json 2>/dev/null
  python - <<PY
import json,glob,csv
l=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); t=l['train']
print('$lib ms/step %.3f mlp_ms %.3f frac %.4f'%(t['ms_per_step'], t['mlp_kernels_ms_per_step'], t['mlp_roofline_frac']))
for f in glob.glob('/tmp/abprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in (${KERNELS:-'wgrad_kernel','wgrad_direct','mlp_fwd_kernel','mlp_bwd_kernel'})):
            print('   %-60s calls %4s avg %.3f ms'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])*1e-6))
PY
done
cp /tmp/orig.so $R/smpl_nerf_amd/csrc/libsmplnerf_hip.so
