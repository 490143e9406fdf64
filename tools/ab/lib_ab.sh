#!/bin/bash
# lib_ab.sh <variant.so> [precisions...]: the render bench line with the shipped library and with a variant library, three
# interleaved runs each (run on the GPU box; the variant comes from tools/ubench/build_variant.sh or a hand-linked .so)
V=$1; shift
PRECS=${@:-fp32}
B="python bench.py --steps 20 --warmup 3 --no-pmc --no-alt --cpu-rays 0 --train-rays 0 --points="
L=smpl_nerf_amd/csrc/libsmplnerf_hip.so
cp $L /tmp/shipped.so
for r in 1 2 3; do
  for v in shipped variant; do
    if [ $v = shipped ]; then cp /tmp/shipped.so $L; else cp $V $L; fi
    for p in $PRECS; do
      $B --precision $p 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v $p', round(d['ms_per_step'],3))"
    done
  done
done
cp /tmp/shipped.so $L
