echo "== full"; python scripts/prefill_bench.py 2>&1 | grep -E "llama2-7b: prompt  256"
for v in 1 3 7 15 8 2 4; do echo "== dbg $v"; L2Z_LIB=$PWD/scripts/dbg_lib$v.so python scripts/prefill_bench.py 2>&1 | grep -E "llama2-7b: prompt  256"; done
