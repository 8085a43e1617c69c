cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 16 32; do python scripts/prefill_ab.py llama2-7b $n 4 "" "L2Z_PF_SLAB=0" "L2Z_PF_SLAB_NST=4" "L2Z_PF_SLAB_NST=8"; done
for v in "L2Z_PF_SLAB_NST=4" "L2Z_PF_SLAB_NST=16"; do
  echo "== $v"
  env $v bash scripts/pf_prof.sh llama2-7b 16 2>&1 | grep prefill_slab | cut -c1-110
done
echo "== 32 tokens NST=4"
L2Z_PF_SLAB_NST=4 bash scripts/pf_prof.sh llama2-7b 32 2>&1 | grep prefill_slab | cut -c1-110
L2Z_PF_SLAB_NST=4 python scripts/slab_stamp.py 32 0
