cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 100 128; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 5 "" 2>&1 | grep prefill
  for v in 1 0; do
    L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_tile$v.so python $repo/scripts/prefill_ab.py llama2-7b $n 5 "" 2>&1 | grep prefill | sed "s/^/   [tile form $v for q|k|v and W1|W3] /"
  done
done
} > $repo/gpurun_out/r05zzz_tile_65_128.txt 2>&1
cat $repo/gpurun_out/r05zzz_tile_65_128.txt
