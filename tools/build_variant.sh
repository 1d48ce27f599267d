#!/bin/bash
# Build an experimental variant of libobtg_hip.so with extra -D flags for gjk_kernels.hip / bern_kernels.hip:
#   tools/build_variant.sh NAME -DSOME_SWITCH=1 ...   ->  optimalbeziertrajectorygeneration_amd/exp_NAME.so
# (for A/B experiments: put the switch in the source under #ifdef, build both, run the probes with OBTG_LIB set)
# Select it at run time with OBTG_LIB=optimalbeziertrajectorygeneration_amd/exp_NAME.so (see _capi.py).
set -e
NAME=$1; shift
PKG=$(dirname "$0")/../optimalbeziertrajectorygeneration_amd
C=$PKG/csrc; O=$C/build/exp_$NAME; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -fvisibility=hidden -fno-gpu-rdc"
/opt/rocm/bin/hipcc $F "$@" -x hip -c $C/bern_kernels.hip -o $O/bern.o &
/opt/rocm/bin/hipcc $F -ffp-contract=off "$@" -x hip -c $C/gjk_kernels.hip -o $O/gjk.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $PKG/exp_$NAME.so $O/bern.o $O/gjk.o $C/build/capi.o $C/build/tables.o $C/build/comm.o $C/build/libm_check.o $C/build/source_hash.o -ldl
echo $PKG/exp_$NAME.so
