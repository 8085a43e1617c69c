#!/bin/bash
# Measurement build of the library with in-kernel wall-clock stamps in the stream form of the prefill GEMM
# (-DL2Z_X3_TIMELINE): llama2.zig_amd/exp/libl2z_x3tl.so, loaded by scripts/x3_timeline.py through L2Z_LIB.  Never the product library.
set -e
cd "$(dirname "$0")/../llama2.zig_amd/csrc"
make -s
mkdir -p ../exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DL2Z_X3_TIMELINE -c prefill_gemm.hip -o ../exp/prefill_gemm_tl.o
objs=$(ls *.o | grep -v '^prefill_gemm.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../exp/libl2z_x3tl.so $objs ../exp/prefill_gemm_tl.o -ldl -Wl,-rpath,/opt/rocm/lib
rm -f ../exp/prefill_gemm_tl.o
ls -la ../exp/libl2z_x3tl.so
