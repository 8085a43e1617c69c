#!/bin/bash
# round 2, GPU call C: prefill GEMM (direct-to-LDS), loader at full size, CLI -g, first profiles
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -k "prefill or cli or c_abi or fused or p2p" > $O/r2c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2c_pytest.log
tail -15 $O/r2c_pytest.log
for d in 1 0; do echo "== L2Z_PF_DMA=$d"; L2Z_PF_DMA=$d PF_NO_STEPPED=1 PF_SIZES=64,128,256,512 python scripts/prefill_bench.py 2>&1 | grep -v "^$"; done | tee $O/r2c_prefill.txt
for t in 8 2 6; do echo "== L2Z_PF_TILE=$t dma"; L2Z_PF_TILE=$t PF_NO_STEPPED=1 PF_SIZES=256,512 python scripts/prefill_bench.py 2>&1 | grep "7b"; done | tee -a $O/r2c_prefill.txt
python scripts/upload_rate.py 2>&1 | tee $O/r2c_upload.txt
