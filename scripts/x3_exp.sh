#!/bin/bash
# Experiment builds of the planes form of the tile GEMM (prefill_gemm.hip, L2Z_X3_EXP bits: 1 no loads in the loop, 2 no MFMAs,
# 4 no X operand reads, 8 no W reads / splits, 16 no barrier): libraries under llama2.zig_amd/exp/ (git-ignored; they travel
# with gpurun), selected by L2Z_LIB.  Results are wrong by construction: timing only.
set -e
cd "$(dirname "$0")/../llama2.zig_amd/csrc"
make -s
mkdir -p ../exp
for e in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DL2Z_X3_EXP=$e -c prefill_gemm.hip -o ../exp/prefill_gemm_$e.o &
done
wait
for e in "$@"; do
  objs=$(ls *.o | grep -v '^prefill_gemm.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../exp/libl2z_x3e$e.so $objs ../exp/prefill_gemm_$e.o -ldl -Wl,-rpath,/opt/rocm/lib
done
ls -la ../exp/*.so
