#!/bin/bash
# HBM traffic per launch of the STREAM form of the bf16-core prefill GEMM (7B shape) from the PMC counters, collected as
# scripts/pmc_traffic.sh does for the decode kernels (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only):
#   stream_pmc_traffic.sh [n_tokens] > gpurun_out/<name>.md
n=${1:-64}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/stpmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/stpmc_$c -o p --output-format csv -- python $repo/scripts/prefill_prof.py llama2-7b $n > /tmp/stpmc_$c.log 2>&1 || tail -5 /tmp/stpmc_$c.log
done
python - "$n" <<'PY'
import csv, glob, sys, collections
n = int(sys.argv[1])
dim, hid = 4096, 11008
names = {"qkv": "q|k|v", "wo": "Wo", "w13": "W1|W3", "w2": "W2"}
wbytes = {"qkv": 4 * 3 * dim * dim, "wo": 4 * dim * dim, "w13": 4 * 2 * hid * dim, "w2": 4 * dim * hid}
K = {"qkv": dim, "wo": dim, "w13": dim, "w2": hid}
N = {"qkv": 3 * dim, "wo": dim, "w13": 2 * hid, "w2": dim}
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}; cnt = collections.Counter(); grid = {}
for c in acc:
    f = glob.glob(f"/tmp/stpmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", c); sys.exit(1)
    recs = sorted((r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))
    n1 = 0
    for r in recs:
        k = r["Kernel_Name"]
        if "prefill_x3_stream<" not in k: continue
        epi = int(k.split("prefill_x3_stream<")[1].split(",")[0])
        if epi == 6: key = "qkv"
        elif epi == 7: key = "w13"
        elif epi == 1: key = "wo" if n1 % 2 == 0 else "w2"; n1 += 1
        else: continue
        acc[c][key] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": cnt[key] += 1; grid[key] = (int(r.get("Grid_Size", 0)) // max(int(r.get("Workgroup_Size", 1)), 1), int(r.get("Workgroup_Size", 0)))
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
B = ge.load_package().binding
sk = {k: max(B.prefill_cores(N[k], n, K[k]) - 1, 1) for k in N}      # K ranges per tile (host logic: l2z_prefill_cores)
print(f"# PMC HBM traffic per launch, stream form of the bf16-core prefill GEMM, {n} tokens, llama2-7b shape\n")
print("FETCH_SIZE (KB) x 1024 x 2 (gfx950: half of a wide coalesced stream is reported, MI355X_MICROARCH.md HBM section) + WRITE_SIZE (KB) x 1024; "
      "separate --pmc passes, kernel-trace only (scripts/stream_pmc_traffic.sh).  W = the product's weight bytes: every element is requested by ONE "
      "block, once.  partials = ranges x tokens x features x 4 bytes: the K ranges' sums, written through and read back once by the blocks "
      "that finish the tile (nothing with one range; Wo and W2 leave theirs to the next rmsnorm launch, which reads them: written here, not read).  X = the activation planes [tokens][3][K] bf16: every block of a K range reads them "
      "through its XCD's L2, so the memory side sees them up to 8 x.  out = the f32 output (+ its planes where the SwiGLU epilogue writes them).\n")
print("| product | launches | blocks x threads | K ranges | read bytes | W bytes | partial bytes | X bytes | read / W | (read - partials read back) / W | (read - W - partials read back) / X | written bytes | written / (partials + out) |\n|---|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for key in ("qkv", "wo", "w13", "w2"):
    c = max(cnt[key], 1)
    rd = acc["FETCH_SIZE"][key] / c * 1024 * 2; wr = acc["WRITE_SIZE"][key] / c * 1024
    x = n * K[key] * 6
    tok = (n + 31) // 32 * 32
    part = sk[key] * tok * N[key] * 4 if sk[key] > 1 else 0
    part_read = 0 if key in ("wo", "w2") else part     # (deferred: the next rmsnorm launch reads them)
    out = n * N[key] * 4 if key != "w13" else n * hid * 4 + n * hid * 6
    print(f"| {names[key]} | {cnt[key]} | {grid.get(key)} | {sk[key]} | {rd:.0f} | {wbytes[key]} | {part} | {x} | {rd / wbytes[key]:.3f} | {(rd - part_read) / wbytes[key]:.3f} | {(rd - wbytes[key] - part_read) / x:.2f} | {wr:.0f} | {wr / (part + out):.2f} |")
PY
