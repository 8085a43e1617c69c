#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 ./scripts/lds_fill_probe > gpurun_out/r6_53_lds_fill_probe.txt 2>&1; grep "row walk\|own 8 MB.*grid  256" gpurun_out/r6_53_lds_fill_probe.txt
