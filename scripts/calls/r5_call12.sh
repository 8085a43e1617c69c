#!/bin/bash
# round 5, call 12: the panel kernel taken apart -- experiment builds (scripts/panel_exp.sh) under the kernel trace, per
# product (q|k|v, Wo, W1|W3, W2 = the launch's place in its layer), then the whole prefill with the new swizzle.
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT; O=$repo/gpurun_out
{
for n in 32 64; do
 for v in def 8 1 2 4 5; do
  lib=$repo/llama2.zig_amd/libllama2_hip.so; [ $v != def ] && lib=$repo/llama2.zig_amd/exp/libl2z_pn$v.so
  rm -rf /tmp/pe; L2Z_LIB=$lib timeout 300 rocprofv3 --kernel-trace -d /tmp/pe -o p -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/pe.log 2>&1 || tail -3 /tmp/pe.log
  python - $n $v $(find /tmp/pe -name "*.db" | head -1) <<'PY'
import sqlite3, sys, collections
n, v, db = sys.argv[1:4]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
slot = collections.defaultdict(list); i = 0; red = collections.defaultdict(list)
for name, s, e in rows:
    if "prefill_panel" in name: slot[i % 4].append(e - s); i += 1
    elif "panel_reduce" in name: red[name.split("panel_reduce")[1][:3]].append(e - s)
names = ["q|k|v", "Wo", "W1|W3", "W2"]
med = lambda x: sorted(x)[len(x) // 2] / 1e3
print(f"{n} tokens, build {v}: " + "  ".join(f"{names[k]} {med(slot[k]):7.1f}" for k in range(4)) + f"  | sum/layer {sum(med(slot[k]) for k in range(4)):7.1f} us | reduces " + " ".join(f"{k} {med(x):5.1f}" for k, x in sorted(red.items())))
PY
 done
done
} > $O/r05l_panel_parts.txt 2>&1
cat $O/r05l_panel_parts.txt
{
for n in 20 32 40 48 64; do
  python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill
  L2Z_LIB=$repo/llama2.zig_amd/exp/libl2z_pn8.so python $repo/scripts/prefill_ab.py llama2-7b $n 8 "" 2>&1 | grep prefill | sed 's/^/   [row \& 7 build] /'
done
} > $O/r05l_prefill_swizzle_ab.txt 2>&1
cat $O/r05l_prefill_swizzle_ab.txt
cd $repo && timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "panel" 2>&1 | tail -3
