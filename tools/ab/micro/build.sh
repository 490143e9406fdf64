#!/bin/bash
# builds the microbenchmarks of this directory into smpl_nerf_amd/csrc/build/ (git-ignored; travels to the GPU box with the snapshot)
R=$(cd $(dirname $0)/../../.. && pwd)
mkdir -p $R/smpl_nerf_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/ab/micro/mfma_issue.hip -o $R/smpl_nerf_amd/csrc/build/mfma_issue
