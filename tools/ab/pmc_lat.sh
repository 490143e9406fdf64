#!/bin/bash
# SQ counters of the latency-class kernels at a few sample counts (three passes), reduced per kernel by tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lat
mkdir -p $OUT
# (r06: the size model decides which calls take the latency kernels - 4096 and 16384 samples do)
CMD="python $R/tools/ab/lat_timing.py ${1:-infer} ${2:-4096,16384,262144}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/p1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p2 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM --output-format csv -d /tmp/p3 -- $CMD > /dev/null 2>&1
for k in "mlp_fwd_lat_kernel<1" "mlp_fwd_lat_kernel<4" "mlp_bwd_lat_kernel<1" "mlp_bwd_lat_kernel<4"; do
  python $R/tools/pmc_summary.py "$k" /tmp/p1 /tmp/p2 /tmp/p3
done > $OUT/summary.json 2>/dev/null
python - <<PY
import json,re
txt=open('$OUT/summary.json').read()
for blob in re.split(r'\n(?=\{)', txt):
    if not blob.strip(): continue
    d=json.loads(blob)
    for k,v in d.items():
        c=v['counters']; wc=c['SQ_WAVE_CYCLES']
        print(k[:70], 'ms %.4f'%(v['avg_seconds']*1e3))
        print('   mfma_busy %.3f clock %.2f'%(v.get('mfma_busy_frac',0), v.get('effective_clock_ghz',0)),
              ' per wave-cycle: wait_any %.3f wait_inst_any %.3f wait_lds %.3f act_valu %.3f act_lds %.3f act_vmem %.3f act_misc %.3f act_sca %.3f'%tuple(c.get(n,0)/wc for n in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_ACTIVE_INST_MISC','SQ_ACTIVE_INST_SCA')))
        print('   insts: valu/mfma %.2f lds/mfma %.2f vmem/mfma %.3f salu/mfma %.2f smem/mfma %.3f  waves %.0f'%(c.get('SQ_INSTS_VALU',0)/max(c.get('SQ_INSTS_MFMA',1),1), c.get('SQ_INSTS_LDS',0)/max(c.get('SQ_INSTS_MFMA',1),1), c.get('SQ_INSTS_VMEM',0)/max(c.get('SQ_INSTS_MFMA',1),1), c.get('SQ_INSTS_SALU',0)/max(c.get('SQ_INSTS_MFMA',1),1), c.get('SQ_INSTS_SMEM',0)/max(c.get('SQ_INSTS_MFMA',1),1), c.get('SQ_WAVES',0)))
PY
