python -m pytest tests -m gpu -x -q -k "grad or train or bwd or backward or fit or warp" 2>&1 | tail -3
for i in 1 2; do for v in old new; do cp ab/$v.so smpl_nerf_amd/csrc/libsmplnerf_hip.so; echo -n "$v: "; python bench.py --steps 5 --warmup 2 --cpu-rays 0 --train-steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['train']; print(t['ms_per_step'], t['loss_last'], json.dumps(t['kernels_ms_per_step']))"; done; done
cp ab/new.so smpl_nerf_amd/csrc/libsmplnerf_hip.so
