#!/bin/bash
# Measurement build of the library with in-kernel wall-clock stamps in the split attention kernel (-DL2Z_TIMELINE):
# llama2.zig_amd/libllama2_hip_tl.so, loaded by scripts/attn_timeline.py through L2Z_LIB.  Never the product library.
set -e
cd "$(dirname "$0")/../llama2.zig_amd/csrc"
make -s all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DL2Z_TIMELINE -c attention.hip -o /tmp/attention_tl.o
OBJS="tunables.o matvec.o /tmp/attention_tl.o fused_small.o misc_kernels.o prefill_gemm.o prefill_skinny.o prefill_attention.o p2p.o weights.o runstate.o forward.o prefill_host.o hooks.o comm.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o ../libllama2_hip_tl.so $OBJS -ldl -Wl,-rpath,/opt/rocm/lib
ls -la ../libllama2_hip_tl.so
