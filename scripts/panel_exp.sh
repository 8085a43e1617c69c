#!/bin/bash
# Experiment builds of the panel kernel (prefill_panel.hip, L2Z_PN_EXP bits: 1 no W loads, 2 no MFMA, 4 no operand reads,
# 8 the first swizzle, row & 7): libraries under llama2.zig_amd/exp/ (git-ignored; they travel with gpurun), selected by L2Z_LIB.
set -e
cd "$(dirname "$0")/../llama2.zig_amd/csrc"
make -s
mkdir -p ../exp
for e in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DL2Z_PN_EXP=$e -c prefill_panel.hip -o ../exp/prefill_panel_$e.o
  objs=$(ls *.o | grep -v '^prefill_panel.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../exp/libl2z_pn$e.so $objs ../exp/prefill_panel_$e.o -ldl -Wl,-rpath,/opt/rocm/lib
done
ls -la ../exp/*.so
