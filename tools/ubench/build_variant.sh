#!/bin/bash
# build_variant.sh <name> <file.hip> [extra hipcc flags]: recompiles one source of csrc/ with extra flags and links it with the
# objects of the last full build into tools/ubench/v_<name>.so (A/B material for tools/ubench/helpers_bench)
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
C=smpl_nerf_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $C/$src -o /tmp/v_${name}.o
objs=$(ls $C/build/*.o | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ubench/v_${name}.so $objs /tmp/v_${name}.o
echo tools/ubench/v_${name}.so
