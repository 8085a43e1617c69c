#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -k "prefill" > $O/r2d_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2d_pytest.log
tail -8 $O/r2d_pytest.log
python scripts/prefill_ab.py llama2-7b 512 4 "L2Z_PF_DMA=0" "L2Z_PF_DMA=1" "L2Z_PF_DMA=2" "L2Z_PF_DMA=3" "L2Z_PF_DMA=3,L2Z_PF_TILE=2" "L2Z_PF_DMA=3,L2Z_PF_TILE=6" 2>&1 | tee $O/r2d_prefill_ab.txt
python scripts/prefill_ab.py llama2-7b 256 4 "L2Z_PF_DMA=0" "L2Z_PF_DMA=1" "L2Z_PF_DMA=2" "L2Z_PF_DMA=3" 2>&1 | tee -a $O/r2d_prefill_ab.txt
python scripts/prefill_ab.py stories110M 512 4 "L2Z_PF_DMA=0" "L2Z_PF_DMA=3" 2>&1 | tee -a $O/r2d_prefill_ab.txt
NTOK=512 bash scripts/pf_pmc.sh 2>&1 | tee $O/r2d_pf_pmc.txt
