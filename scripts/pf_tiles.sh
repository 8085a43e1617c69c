for t in 0 1 2 3 4 5 6 7 8; do echo "== tile $t"; L2Z_PF_TILE=$t python scripts/prefill_bench.py 2>&1 | grep -E "llama2-7b|110M: prompt  256"; done
