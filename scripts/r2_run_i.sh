#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -k "prefill" > $O/r2i_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2i_pytest.log
tail -3 $O/r2i_pytest.log
python scripts/prefill_ab.py llama2-7b 512 4 "" "L2Z_PF_XCD=0" 2>&1 | tee $O/r2i_prefill_ab.txt
python scripts/prefill_ab.py llama2-7b 256 4 "" 2>&1 | tee -a $O/r2i_prefill_ab.txt
