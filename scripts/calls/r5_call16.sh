cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
for n in 20 32 48 64; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn16.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed 's/^/   [two ring buffers] /'
done > $repo/gpurun_out/r05r_panel_depth2.txt 2>&1
cat $repo/gpurun_out/r05r_panel_depth2.txt
