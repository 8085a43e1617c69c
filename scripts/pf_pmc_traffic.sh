#!/bin/bash
# HBM traffic per launch of the panel prefill kernel and its reduce launches (7B shape) from the PMC counters, collected as
# scripts/pmc_traffic.sh does for the decode kernels (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only):
#   pf_pmc_traffic.sh [n_tokens] > gpurun_out/<name>.md
n=${1:-64}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pfpmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pfpmc_$c -o p --output-format csv -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/pfpmc_$c.log 2>&1 || tail -5 /tmp/pfpmc_$c.log
done
python - "$n" <<'PY'
import csv, glob, sys, collections
n = int(sys.argv[1])
dim, hid = 4096, 11008
names = ["q|k|v", "Wo", "W1|W3", "W2"]
wbytes = [4 * 3 * dim * dim, 4 * dim * dim, 4 * 2 * hid * dim, 4 * dim * hid]
tms = (n + 15) // 16; kr = 512 if tms <= 3 else 256      # round 6: three tiles against ranges of 512
ranges = [-(-dim // kr), -(-dim // kr), -(-dim // kr), -(-hid // kr)]
rows_n = [3 * dim, dim, 2 * hid, dim]
part = [4 * r * 16 * tms * N for r, N in zip(ranges, rows_n)]     # partial products written by the panel launch, read by the reduce
xbytes = [4 * 16 * tms * k for k in (dim, dim, dim, hid)]         # the activation matrix [16 tms, K] the product multiplies
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}; cnt = collections.Counter()
for c in acc:
    f = glob.glob(f"/tmp/pfpmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", c); sys.exit(1)
    recs = sorted((r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))
    i = j = 0
    for r in recs:
        k = r["Kernel_Name"]
        if "prefill_panel" in k: key = ("panel", i % 4); i += 1
        elif "panel_reduce" in k: key = ("reduce", j % 4); j += 1
        else: continue
        acc[c][key] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": cnt[key] += 1
print(f"# PMC HBM traffic per launch, panel prefill of {n} tokens, llama2-7b shape\n")
print("FETCH_SIZE (KB) x 1024 x 2 (gfx950: half of a wide coalesced stream is reported, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB) x 1024; "
      "separate --pmc passes, kernel-trace only (scripts/pf_pmc_traffic.sh).  W = the product's weight bytes; partials = ranges x 16 TMS x N x 4 bytes.\n")
print("X = the activation matrix [16 tms, K]: every block keeps ITS range of it in LDS, so each of the 8 XCDs' L2s fetches it once -- "
      "(read - W) / X is how many times X came in (<= 8: nothing else is read twice).\n")
print("| launch | product | launches | read bytes | written bytes | W bytes | X bytes | partial bytes | read / W | (read - W) / X | written / partials |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for kind in ("panel", "reduce"):
    for p in range(4):
        key = (kind, p); c = max(cnt[key], 1)
        rd = acc["FETCH_SIZE"][key] / c * 1024 * 2; wr = acc["WRITE_SIZE"][key] / c * 1024
        if kind == "panel": print(f"| prefill_panel | {names[p]} | {cnt[key]} | {rd:.0f} | {wr:.0f} | {wbytes[p]} | {xbytes[p]} | {part[p]} | {rd / wbytes[p]:.3f} | {(rd - wbytes[p]) / xbytes[p]:.2f} | {wr / part[p]:.3f} |")
        else: print(f"| panel_reduce | {names[p]} | {cnt[key]} | {rd:.0f} | {wr:.0f} | - | - | {part[p]} | (read / partials {rd / part[p]:.3f}) | - | - |")
PY
