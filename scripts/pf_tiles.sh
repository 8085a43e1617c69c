for t in ${TILES:-0 1 2 4 6 7}; do echo "== tile $t chunk ${L2Z_PF_CHUNK:-256}"; L2Z_PF_TILE=$t python scripts/prefill_bench.py 2>&1 | grep -E "llama2-7b: prompt  (256|512)|110M: prompt  256"; done
