L=smpl_nerf_amd/csrc/libsmplnerf_hip.so
cp $L /tmp/shipped.so
for r in 1 2; do
for v in shipped variant; do
  if [ $v = shipped ]; then cp /tmp/shipped.so $L; else cp tools/ubench/v_conc.so $L; fi
  echo "== $v"; python tools/ab/train_points.py --rays 128,256,512,800,1024,2048 --chunks 2048 --steps 40 2>&1 | grep rays | cut -c1-70
done; done
cp /tmp/shipped.so $L
