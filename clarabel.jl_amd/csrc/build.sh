#!/bin/bash
# Builds the HIP library for gfx950 in-tree (clarabel.jl_amd/), twice.  hipcc cross-compiles without a GPU.
#   libclarabel_hipkkt.so          the PRODUCT: what the Julia glue, bench.py and smoke() load.  Reads no HIPKKT_* switch (hipkkt_debug_set
#                                  refuses), contains neither the first form of the front-batch kernel nor the debug flags.
#   libclarabel_hipkkt_testing.so  the same sources with -DHIPKKT_TESTING + front_block.hip: what tests/ load (hipkkt_debug_set works).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $HIPKKT_EXTRA_FLAGS"   # (development: e.g. -DHIPKKT_SWEEP_TRACE)
mkdir -p ../../build/obj ../../build/obj_testing
HOSTSRC="hipkkt_abi.cpp hipkkt_setup.cpp hipkkt_factor.cpp hipkkt_solve.cpp symbolic.cpp ordering.cpp assemble.cpp"
# kernels are identical in both builds (nothing in a .hip file depends on HIPKKT_TESTING): compiled once
KSRC="kernels.hip assemble_dev.hip front_sweep.hip front_block2.hip probe.hip"
OBJ=../../build/obj
OBJT=../../build/obj_testing
pids=()
for f in $HOSTSRC; do
  $HIPCC $FLAGS -x c++ -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $f -o $OBJ/${f%.cpp}.o & pids+=($!)
done
# only the files that look at HIPKKT_TESTING are compiled a second time
for f in hipkkt_abi.cpp hipkkt_setup.cpp hipkkt_factor.cpp; do
  $HIPCC $FLAGS -DHIPKKT_TESTING -x c++ -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $f -o $OBJT/${f%.cpp}.o & pids+=($!)
done
for f in $KSRC; do
  $HIPCC $FLAGS -c $f -o $OBJ/${f%.hip}.o & pids+=($!)
done
$HIPCC $FLAGS -c front_block.hip -o $OBJT/front_block.o & pids+=($!)
# scaling.hip mirrors the reference's cone formulas operation by operation: no FMA contraction
$HIPCC $FLAGS -ffp-contract=off -c scaling.hip -o $OBJ/scaling.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
COMMON="$OBJ/hipkkt_solve.o $OBJ/symbolic.o $OBJ/ordering.o $OBJ/assemble.o $OBJ/assemble_dev.o $OBJ/scaling.o $OBJ/front_block2.o $OBJ/front_sweep.o $OBJ/probe.o $OBJ/kernels.o"
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libclarabel_hipkkt.so $OBJ/hipkkt_abi.o $OBJ/hipkkt_setup.o $OBJ/hipkkt_factor.o $COMMON
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libclarabel_hipkkt_testing.so $OBJT/hipkkt_abi.o $OBJT/hipkkt_setup.o $OBJT/hipkkt_factor.o $OBJT/front_block.o $COMMON
echo "built $(readlink -f ../libclarabel_hipkkt.so) and $(readlink -f ../libclarabel_hipkkt_testing.so)"
