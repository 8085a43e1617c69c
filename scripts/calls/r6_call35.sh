#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 ./scripts/lds_fill_probe > gpurun_out/r6_39_lds_fill_probe.txt 2>&1; cat gpurun_out/r6_39_lds_fill_probe.txt
