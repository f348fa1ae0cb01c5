#!/bin/bash
# Builds libclarabel_hipkkt.so for gfx950 in-tree (clarabel.jl_amd/).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libclarabel_hipkkt.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result"
mkdir -p ../../build/obj
OBJ=../../build/obj
pids=()
for f in hipkkt_abi.cpp hipkkt_setup.cpp hipkkt_factor.cpp hipkkt_solve.cpp symbolic.cpp ordering.cpp assemble.cpp; do
  $HIPCC $FLAGS -x c++ -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $f -o $OBJ/${f%.cpp}.o & pids+=($!)
done
$HIPCC $FLAGS -c kernels.hip -o $OBJ/kernels.o & pids+=($!)
$HIPCC $FLAGS -c assemble_dev.hip -o $OBJ/assemble_dev.o & pids+=($!)
$HIPCC $FLAGS -c front_block.hip -o $OBJ/front_block.o & pids+=($!)
$HIPCC $FLAGS -c front_sweep.hip -o $OBJ/front_sweep.o & pids+=($!)
$HIPCC $FLAGS -c front_block2.hip -o $OBJ/front_block2.o & pids+=($!)
$HIPCC $FLAGS -c probe.hip -o $OBJ/probe.o & pids+=($!)
# scaling.hip mirrors the reference's cone formulas operation by operation: no FMA contraction
$HIPCC $FLAGS -ffp-contract=off -c scaling.hip -o $OBJ/scaling.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/hipkkt_abi.o $OBJ/hipkkt_setup.o $OBJ/hipkkt_factor.o $OBJ/hipkkt_solve.o $OBJ/symbolic.o $OBJ/ordering.o $OBJ/assemble.o $OBJ/assemble_dev.o $OBJ/scaling.o $OBJ/front_block.o $OBJ/front_block2.o $OBJ/front_sweep.o $OBJ/probe.o $OBJ/kernels.o
echo "built $(readlink -f $OUT)"
