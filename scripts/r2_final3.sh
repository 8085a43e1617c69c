#!/bin/bash
# round 2: refresh of what the prefill attention change touches (bench line, prefill tables, A/B)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/r02_bench.json 2> $O/r02_bench.err; tail -c 300 $O/r02_bench.err
python bench.py --steps 20 --warmup 5 > $O/r02_bench_driver_args.json 2>> $O/r02_bench.err
bash scripts/pf_prof.sh llama2-7b 512 > $O/r02_prefill512_llama2-7b.md 2>/dev/null; head -12 $O/r02_prefill512_llama2-7b.md | cut -c1-150
bash scripts/pf_prof.sh llama2-7b 128 > $O/r02_prefill128_llama2-7b.md 2>/dev/null
bash scripts/pf_prof.sh llama2-7b 16 > $O/r02_prefill16_llama2-7b.md 2>/dev/null
( python scripts/prefill_ab.py llama2-7b 2000 3 "" "L2Z_PF_ATTN=2"
  python scripts/prefill_ab.py llama2-7b 512 4 "" "L2Z_PF_ATTN=2" "L2Z_PF_FUSE=0" "L2Z_PF_DMA=0" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 256 4 "" "L2Z_PF_ATTN=2" "L2Z_PF_FUSE=0" "L2Z_PF_DMA=0" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 128 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_TILE=2" "L2Z_PF_ORDER=0"
  python scripts/prefill_ab.py llama2-7b 64 4 "" "L2Z_PF_SKINNY_TMS=4"
  python scripts/prefill_ab.py llama2-7b 16 4 "" "L2Z_PF_FUSE=0" "L2Z_PF_SKINNY_FORM=2" "L2Z_PF_SKINNY_FORM=0"
  python scripts/prefill_ab.py stories110M 256 6 "" "L2Z_PF_ATTN=0" "L2Z_PF_FUSE=0" "L2Z_PF_TILE=2"
  python scripts/prefill_ab.py stories15M 250 6 "" ) 2>&1 | grep prefill | tee $O/r02_prefill_ab.txt
for w in 2 4 8; do bash scripts/r2_sharded_prefill_prof.sh $w > $O/sp$w.log 2>&1; done
python - <<PY
import json
for f in ("r02_bench.json","r02_bench_driver_args.json"):
    d=json.loads(open("gpurun_out/"+f).read().strip().split("\n")[-1])
    r=d["roofline"]; ex=d.get("extra",{})
    print(f, round(d["value"],2), round(d["ms_per_step"],4), "frac", round(r["frac"],4), "prefill", ex.get("prefill",{}).get("ms"), ex.get("prefill",{}).get("roofline",{}).get("frac"), ex.get("prefill",{}).get("ms_by_prompt_tokens"), "long", ex.get("long_context",{}).get("tokens_per_s"))
PY
