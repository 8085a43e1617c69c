#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O
for n in 64 128; do bash scripts/stream_pmc_traffic.sh $n > $O/r06_final_stream_pmc_traffic_$n.md 2>&1; tail -5 $O/r06_final_stream_pmc_traffic_$n.md; done
