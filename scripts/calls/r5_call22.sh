cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT
{
for n in 40 48 64; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn64.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed "s/^/   [ranges of 512, 4 waves x 2 buffers] /"
done
} > $repo/gpurun_out/r05zz_panel_kr512.txt 2>&1
cat $repo/gpurun_out/r05zz_panel_kr512.txt
L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn64.so timeout 600 python -m pytest $repo/tests/test_gpu_parity.py -m gpu -x -q -k "panel_kernel" 2>&1 | tail -1
