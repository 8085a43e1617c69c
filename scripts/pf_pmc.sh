# PMC passes over one 256-token 7B prefill: where do the GEMM waves wait?
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/prefill_prof.py ${SHAPE:-llama2-7b} ${NTOK:-256} > /tmp/pmc_log$i.txt 2>&1 || tail -5 /tmp/pmc_log$i.txt
  python - "$i" <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
f = glob.glob(f"/tmp/pmc{i}/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv for pass", i); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("l2z::(anonymous namespace)::", "")[:40]; n[k] += 1
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in acc.items():
    if "gemm" in k or "skinny" in k or "attention" in k:
        print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
done
