#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# -> gpurun_out/<tag>/ : kernel-trace stats of the default bench run, three PMC passes without the training section (clean
#    render rows) and three with it; tools/make_pmc_profile.py reduces them to gpurun_out/<tag>/pmc_summary.json.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
# (a) render only: every dispatch of a render kernel is one of the bench's frame launches (12 steps x {coarse, fine})
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 10 --warmup 2 --no-pmc --train-rays 0 --points= --netwidth-points= > $OUT/bench_under_rocprof.json.log 2> $OUT/trace.err
# (b) the training section (all precisions) behind a one-step render
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python $ROOT/bench.py --steps 1 --warmup 1 --no-pmc --cpu-rays 0 --points= --netwidth-points= > $OUT/train_under_rocprof.json.log 2> $OUT/trace_train.err
# (c) smpl_nerf: render + training rows of the warp kernels (warp_fwd_resident_kernel, warp_bwd_light_kernel) and the
#     INPUT_GRAD dgrad variants; fp32 only, no CPU leg
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_smpl -- python $ROOT/bench.py --workload smpl_nerf --steps 10 --warmup 2 --no-pmc --no-alt --cpu-rays 0 --train-steps 10 --points= > $OUT/smpl_nerf_under_rocprof.json.log 2> $OUT/trace_smpl.err
timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmcs_sq -- python $ROOT/bench.py --workload smpl_nerf --steps 3 --warmup 1 --cpu-rays 0 --train-steps 3 --no-pmc --no-alt --points= > $OUT/pmcs_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcs_fetch -- python $ROOT/bench.py --workload smpl_nerf --steps 3 --warmup 1 --cpu-rays 0 --train-steps 3 --no-pmc --no-alt --points= > $OUT/pmcs_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcs_write -- python $ROOT/bench.py --workload smpl_nerf --steps 3 --warmup 1 --cpu-rays 0 --train-steps 3 --no-pmc --no-alt --points= > $OUT/pmcs_write.log 2>&1
RENDER="--steps 3 --warmup 1 --cpu-rays 0 --train-rays 0 --no-pmc --points= --netwidth-points="
TRAIN="--steps 1 --warmup 1 --cpu-rays 0 --train-steps 3 --no-pmc --points= --netwidth-points="
timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq -- python $ROOT/bench.py $RENDER > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $ROOT/bench.py $RENDER > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $ROOT/bench.py $RENDER > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmct_sq -- python $ROOT/bench.py $TRAIN > $OUT/pmct_sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmct_fetch -- python $ROOT/bench.py $TRAIN > $OUT/pmct_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmct_write -- python $ROOT/bench.py $TRAIN > $OUT/pmct_write.log 2>&1
# (d) the multi-rank path at world 8 on this 1-GPU box: launcher, rendezvous, replica broadcast and the flat gradient all-reduce
#     (gloo, ranks share the device - a dry run of the code path, not a scaling measurement)
timeout 900 python $ROOT/bench.py --gpus 8 --steps 3 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays 512 --train-steps 3 > $OUT/bench_8ranks_1gpu_gloo.json.log 2> $OUT/bench_8ranks.err
# (e) other workloads / operating points of the reference (one bench line each)
timeout 600 python $ROOT/bench.py --coarse-only --no-pmc --no-alt --steps 20 > $OUT/bench_coarse_only.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py --workload smpl_nerf --no-pmc --steps 10 --cpu-rays 1024 --cpu-train-rays 256 > $OUT/bench_smpl_nerf.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py --workload smpl_nerf --res 256 --no-pmc --no-alt --steps 5 --cpu-rays 0 --points= > $OUT/bench_smpl_nerf_256.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py --workload append_vertices --res 256 --no-pmc --no-alt --steps 5 --cpu-rays 64 --points= > $OUT/bench_append_vertices_256.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py --workload append_smpl_params --no-pmc --no-alt --steps 10 --cpu-rays 1024 --points= > $OUT/bench_append_smpl_params.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py > $OUT/bench_default.json.log 2>/dev/null
# (f) round 4: the one-call training step at the README's 64-ray batch (kernel trace), the operating-point table with the ray-chunk
#     sizes and the autograd form beside it, the input-gradient contraction against the r03 form
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train64 -- python $ROOT/bench.py --steps 1 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --netwidth-points= --train-rays 64 --train-steps 50 > $OUT/train64_under_rocprof.json.log 2> $OUT/trace_train64.err
find $OUT/trace_train64 -name "*kernel_stats.csv" -exec cp {} $OUT/train64_kernel_stats.csv \;
rm -rf $OUT/trace_train64
timeout 600 python $ROOT/tools/ab/train_points.py --autograd --chunks 0,1024,2048 --steps 20 2>/dev/null | grep rays > $OUT/train_points.txt
( timeout 300 python $ROOT/tools/ab/input_grad_ab.py; timeout 300 python $ROOT/tools/ab/input_grad_ab.py --spr 64 --cols 60 ) 2>/dev/null | grep -v amdgpu.ids > $OUT/input_grad_ab.txt
# (g) training fed by on-device ray generation over configs[3]'s data set shape (1200 frames of 256 x 256, sharded by image), shuffled epochs
timeout 900 python $ROOT/bench.py --res 256 --train-from-raygen --raygen-frames 1200 --train-rays 2048 --train-steps 20 --steps 2 --warmup 1 --no-alt --no-pmc --cpu-rays 0 --points= > $OUT/bench_train_from_raygen.json.log 2>/dev/null
# (h) the pose- / vertex-conditioned workloads with the gradient flowing into the per-ray inputs (estimator trained / d goal_pose)
timeout 600 python $ROOT/bench.py --workload append_vertices --train-input-grads --no-alt --no-pmc --steps 3 --cpu-rays 0 --points= > $OUT/bench_append_vertices_input_grads.json.log 2>/dev/null
timeout 600 python $ROOT/bench.py --workload append_smpl_params --train-input-grads --no-alt --no-pmc --steps 3 --cpu-rays 0 --points= > $OUT/bench_append_smpl_params_input_grads.json.log 2>/dev/null
# (i) multi-rank dry runs on this 1-GPU box (gloo, ranks share the device): strong scaling (one frame split by rows), the coarse-only
#     line north_star quotes at 1 / 2 / 4 / 8 GPUs, and the RCCL branch at world size one
timeout 900 python $ROOT/bench.py --gpus 2 --scaling strong --steps 3 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays 0 > $OUT/bench_2ranks_strong_1gpu_gloo.json.log 2>/dev/null
timeout 900 python $ROOT/bench.py --gpus 8 --coarse-only --steps 3 --warmup 1 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays 0 > $OUT/bench_8ranks_coarse_only_1gpu_gloo.json.log 2>/dev/null
SNERF_BENCH_FORCE_GROUP=1 SNERF_DIST_BACKEND=nccl timeout 600 python $ROOT/bench.py --steps 5 --warmup 2 --no-pmc --no-alt --cpu-rays 0 --points= --train-rays 2048 --train-steps 10 > $OUT/bench_world1_rccl.json.log 2>/dev/null
cd $ROOT
python tools/make_pmc_profile.py $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmct_sq $OUT/pmct_fetch $OUT/pmct_write > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
python tools/make_pmc_profile.py --smpl-nerf $OUT/pmcs_sq $OUT/pmcs_fetch $OUT/pmcs_write > $OUT/pmc_summary_smpl_nerf.json 2> $OUT/pmc_summary_smpl.err
find $OUT/trace_smpl -name "*kernel_stats.csv" -exec cp {} $OUT/smpl_nerf_kernel_stats.csv \;
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
find $OUT/trace_train -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
# keep what travels back small: the raw per-dispatch counter CSVs are reduced above
rm -rf $OUT/trace $OUT/trace_train $OUT/trace_smpl $OUT/pmcs_sq $OUT/pmcs_fetch $OUT/pmcs_write $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write $OUT/pmct_sq $OUT/pmct_fetch $OUT/pmct_write
ls -la $OUT
# (j) round 5: the latency-class kernels of small calls - isolated per-launch times of forward / training forward / dgrad against the
#     throughput kernels (SNERF_LAT=0), and their SQ counters
( echo "# latency-class kernels (default selection)"; timeout 300 bash $ROOT/tools/ab/lat_trace.sh 4096,12288,16384,51200
  echo "# throughput kernels (SNERF_LAT=0)"; SNERF_LAT=0 timeout 300 bash $ROOT/tools/ab/lat_trace.sh 4096,12288,16384,51200
  echo "# inference, HIP events"; timeout 200 python $ROOT/tools/ab/lat_timing.py infer 4096,12288,16384,20480,32768,51200,65536,153600
  SNERF_LAT=0 timeout 200 python $ROOT/tools/ab/lat_timing.py infer 4096,12288,16384,20480,32768,51200,65536,153600 ) 2>/dev/null | grep -v "amdgpu.ids\|simple_timer" > $OUT/lat_kernels.txt
timeout 400 bash $ROOT/tools/ab/pmc_lat.sh infer 4096,16384 2>/dev/null | grep -v amdgpu.ids > $OUT/pmc_lat.txt
# (k) round 5: --netwidth above 256 - per-launch times / fractions of the 320 .. 512 kernels, their SQ counters beside the 256 kernel,
#     and what one wave per SIMD can issue (microbenchmark, built by tools/ab/micro/build.sh into csrc/build/)
timeout 400 python $ROOT/tools/ab/width_timing.py 256,320,384,448,512 2>/dev/null | grep width > $OUT/width_timing.txt
timeout 600 bash $ROOT/tools/ab/pmc_width.sh 256,320,512 524288 2>/dev/null | grep -v amdgpu.ids > $OUT/pmc_width.txt
[ -x $ROOT/smpl_nerf_amd/csrc/build/mfma_issue ] && timeout 120 $ROOT/smpl_nerf_amd/csrc/build/mfma_issue > $OUT/mfma_issue.txt 2>/dev/null
# (l) round 6: timelines of one small training step (kernel, queue, start, duration) from the kernel trace; host cost of a small
#     render (single-call dispatch under no_grad); the layer-by-layer path of widths above the fused kernels
( timeout 200 bash $ROOT/tools/ab/step_timeline.sh 64 nerf 30 ) 2>/dev/null | grep -v "amdgpu.ids\|simple_timer" > $OUT/step_timeline_64.txt
( timeout 200 bash $ROOT/tools/ab/step_timeline.sh 64 smpl_nerf 30 ) 2>/dev/null | grep -v "amdgpu.ids\|simple_timer" > $OUT/step_timeline_64_smpl_nerf.txt
( timeout 200 bash $ROOT/tools/ab/step_timeline.sh 800 nerf 10 ) 2>/dev/null | grep -v "amdgpu.ids\|simple_timer" > $OUT/step_timeline_800.txt
( for n in 64 128 800; do timeout 120 python $ROOT/tools/ab/host_profile_render.py $n 200; done ) 2>/dev/null | grep "rays\|bit for" > $OUT/host_profile_render.txt
( timeout 300 python $ROOT/tools/ab/layered_timing.py 768; timeout 300 python $ROOT/tools/ab/layered_timing.py 1024 524288 131072 ) 2>/dev/null | grep width > $OUT/layered_timing.txt
( for r in 64 128 256 800 2048 4096; do timeout 200 python $ROOT/tools/ab/train_loop.py $r nerf 30; done; for r in 64 4096; do timeout 200 python $ROOT/tools/ab/train_loop.py $r smpl_nerf 30; done ) 2>/dev/null | grep rays > $OUT/train_loop.txt
