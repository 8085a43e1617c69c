#!/bin/bash
# HBM read traffic of a short-prompt (skinny-kernel) prefill: FETCH_SIZE per kernel vs the weight bytes
repo=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sk
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_sk -o p --output-format csv -- python $repo/scripts/prefill_prof.py ${SHAPE:-llama2-7b} ${NTOK:-16} > /tmp/pmc_sk.log 2>&1 || tail -5 /tmp/pmc_sk.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pmc_sk/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].replace("l2z::(anonymous namespace)::", "")[:60]; n[k] += 1; acc[k] += float(r["Counter_Value"])
for k in acc: print(f"{k:62s} launches {n[k]:5d}  FETCH_SIZE x 2 = {acc[k]*2048/1e9:8.3f} GB total, {acc[k]*2048/n[k]/1e6:8.2f} MB per launch")
PY
