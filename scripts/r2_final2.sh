#!/bin/bash
# round 2, final measurements after the prefill work: what profiles/ and DESIGN.md quote (tag r02)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 > $O/r02_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/r02_pytest_gpu.log
tail -3 $O/r02_pytest_gpu.log
python bench.py > $O/r02_bench.json 2> $O/r02_bench.err; tail -c 300 $O/r02_bench.err
python bench.py --steps 20 --warmup 5 > $O/r02_bench_driver_args.json 2>> $O/r02_bench.err
bash scripts/r2_prof.sh r02 > $O/r02_prof.log 2>&1
bash scripts/pmc_traffic.sh r02 > $O/r02_pmc.log 2>&1; tail -9 $O/r02_pmc.log
bash scripts/pf_prof.sh llama2-7b 512 > $O/r02_prefill512_llama2-7b.md 2>/dev/null; head -9 $O/r02_prefill512_llama2-7b.md | cut -c1-150
bash scripts/pf_prof.sh llama2-7b 128 > $O/r02_prefill128_llama2-7b.md 2>/dev/null
bash scripts/pf_prof.sh llama2-7b 16 > $O/r02_prefill16_llama2-7b.md 2>/dev/null
( python scripts/prefill_ab.py llama2-7b 512 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_DMA=0" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 256 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_DMA=0" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 128 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_TILE=2" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 64 4 ""
  python scripts/prefill_ab.py llama2-7b 16 4 "" "L2Z_PF_SKINNY_FORM=2" "L2Z_PF_SKINNY_FORM=0"
  python scripts/prefill_ab.py stories110M 256 6 "" "L2Z_PF_FUSE=0" "L2Z_PF_TILE=2"
  python scripts/prefill_ab.py stories15M 250 6 "" ) 2>&1 | grep prefill | tee $O/r02_prefill_ab.txt
for w in 2 4 8; do bash scripts/r2_sharded_prefill_prof.sh $w > $O/sp$w.log 2>&1; done
grep -h "ranks\|1 rank" $O/sp*.log | grep -v "^#"
