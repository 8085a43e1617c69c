#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x -s 2>&1 | grep -E "stream form, tiles|passed|failed|Error" | tail -8
rm -rf /tmp/prof_t; ( cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o p -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_x3.py -q -x -k "tiles_of_every_width and 64" > /tmp/t.log 2>&1; python $GRAFT_REPO_ROOT/scripts/rocprof_summary.py $(find /tmp/prof_t -name "*.db" | head -1) "the tile-width test" | grep x3_stream )
} > gpurun_out/r6_54_tile_width_test.txt 2>&1
cat gpurun_out/r6_54_tile_width_test.txt
