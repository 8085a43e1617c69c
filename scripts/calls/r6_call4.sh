#!/bin/bash
# round 6, GPU call 4: perf floors of everything the gate holds + PMC traffic of the panel kernel at 48 / 64 / 96 tokens + the new diag test
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out; O=gpurun_out
python scripts/perf_floor.py > $O/r6_4_perf_floor_a.json 2> $O/r6_4_perf_floor_a.err
python scripts/perf_floor.py > $O/r6_4_perf_floor_b.json 2> $O/r6_4_perf_floor_b.err
for n in 48 64 96; do bash scripts/pf_pmc_traffic.sh $n > $O/r6_4_panel_pmc_traffic_$n.md 2>&1; done
cat $O/r6_4_perf_floor_a.json; tail -3 $O/r6_4_perf_floor_a.err; cat $O/r6_4_panel_pmc_traffic_64.md
