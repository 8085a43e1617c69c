#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
python scripts/prefill_ab.py llama2-7b 128 5 "L2Z_PF_X3=0" "" "L2Z_PF_X3_SK=2424,L2Z_PF_X3_TOK=128" "L2Z_PF_X3_SK=1414,L2Z_PF_X3_TOK=128" "L2Z_PF_X3_SK=2422,L2Z_PF_X3_TOK=128" "L2Z_PF_X3_SK=4444,L2Z_PF_X3_TOK=128" "L2Z_PF_X3_SK=2828,L2Z_PF_X3_TOK=128"
python scripts/prefill_ab.py llama2-7b 100 5 "L2Z_PF_X3=0" "" "L2Z_PF_X3_SK=2424,L2Z_PF_X3_TOK=128"
python scripts/prefill_ab.py llama2-7b 96 5 "L2Z_PF_X3=0" "" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=2424,L2Z_PF_X3_TOK=128"
python scripts/prefill_ab.py llama2-7b 64 5 "L2Z_PF_X3=0" "" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=2424,L2Z_PF_X3_TOK=64" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=4848,L2Z_PF_X3_TOK=64" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=4444,L2Z_PF_X3_TOK=64"
python scripts/prefill_ab.py llama2-7b 32 5 "L2Z_PF_X3=0" "" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=4848,L2Z_PF_X3_TOK=32" "L2Z_PF_PANEL=0,L2Z_PF_X3_SK=4444,L2Z_PF_X3_TOK=32"
python scripts/prefill_ab.py llama2-7b 256 5 "L2Z_PF_X3=0" "" "L2Z_PF_X3_SK=1212,L2Z_PF_X3_TOK=128" "L2Z_PF_X3_SK=2222,L2Z_PF_X3_TOK=128"
} > gpurun_out/r6_16_x3_sk.txt 2>&1
cat gpurun_out/r6_16_x3_sk.txt
