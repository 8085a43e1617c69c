#!/bin/bash
repo=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $repo/gpurun_out
{
for n in 64 128; do
for e in 1 2 4 8; do
  rm -rf /tmp/prof_sk; L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_sk$e.so rocprofv3 --kernel-trace -d /tmp/prof_sk -o p -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/sk.log 2>&1 || tail -3 /tmp/sk.log
  python $repo/scripts/stream_sk_scan.py $(find /tmp/prof_sk -name "*.db" | head -1) "$n tokens, every product in $e K ranges"
done
done
} > $repo/gpurun_out/r6_37_stream_sk_scan.txt 2>&1
cat $repo/gpurun_out/r6_37_stream_sk_scan.txt
