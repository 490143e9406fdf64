#!/bin/bash
# A/B of the wgrad variants on the fp32 training step: per-kernel averages from rocprofv3 --kernel-trace --stats
# CONFIGS: space-separated "fold:prefetch" pairs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in ${CONFIGS:-1:4 0:4 2:4}; do
  f=${cfg%%:*}; pf=${cfg##*:}
  rm -rf /tmp/abprof
  SNERF_WGRAD_FOLD=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof -- python $R/bench.py --steps 1 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays ${RAYS:-4096} --train-steps 10 > /tmp/ab.json 2>/dev/null
  python - <<PY
import json,glob,csv
l=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); t=l['train']
print('fold=$f prefetch=$pf ms/step %.3f mlp_ms %.3f frac %.4f'%(t['ms_per_step'], t['mlp_kernels_ms_per_step'], t['mlp_roofline_frac']))
for f in glob.glob('/tmp/abprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('wgrad_kernel','wgrad_direct')):
            print('   %-50s calls %4s avg %.3f ms'%(r['Name'][:50], r['Calls'], float(r['AverageNs'])*1e-6))
PY
done
