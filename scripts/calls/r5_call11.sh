#!/bin/bash
# round 5, call 11: where do the panel kernel's waves spend their cycles?  One --pmc pass (kernel-trace only) per chunk length.
cd /tmp && export TMPDIR=/tmp
repo=$GRAFT_REPO_ROOT; O=$repo/gpurun_out
for n in 32 64; do
  rm -rf /tmp/pmc_pn$n
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_pn$n -o p --output-format csv -- \
    python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/pmc_pn$n.log 2>&1; grep -v 'rocprofv3\|output_stream' /tmp/pmc_pn$n.log | head -12 | cut -c1-300
  python - $n <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
f = glob.glob(f"/tmp/pmc_pn{n}/**/*counter_collection.csv", recursive=True)
if not f: print("no csv"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void l2z::", "").split("(")[0]
    if not ("prefill_panel" in k or "panel_reduce" in k or "prefill_gemm_dma" in k): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
print(f"## {n} tokens: per launch averages (quad-cycles for SQ_WAVE_CYCLES / WAIT / ACTIVE; cycles for MFMA_BUSY, GRBM)")
for k, d in acc.items():
    c = max(cnt[k], 1); wc = d["SQ_WAVE_CYCLES"] or 1
    print(f"{k}: launches {cnt[k]}  wave_cycles {d['SQ_WAVE_CYCLES']/c:.3g}  parked(WAIT_ANY) {d['SQ_WAIT_ANY']/wc:.2f}  issue-stall(WAIT_INST_ANY) {d['SQ_WAIT_INST_ANY']/wc:.2f}  active {d['SQ_ACTIVE_INST_ANY']/wc:.2f}  "
          f"mfma_busy/gui_active {d['SQ_VALU_MFMA_BUSY_CYCLES']/max(d['GRBM_GUI_ACTIVE'],1):.3f}  gui_active {d['GRBM_GUI_ACTIVE']/c:.3g}")
PY
done > $O/r05k_panel_pmc.md 2>&1
cat $O/r05k_panel_pmc.md
