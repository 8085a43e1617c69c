#!/bin/bash
# round 3, GPU call J: bench.py --gpus 8 with all eight ranks on the ONE GPU of the box -- the control path of the
# real thing (24 child processes, three legs, eight gloo ranks), timed end to end
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29588 \
    bench.py --gpus 8 --steps 20 --warmup 5 > $O/r03_mp8.json 2> $O/r03_mp8.err
echo "rc=$? wall=$(echo "$(date +%s.%N) - $t0" | bc) s" | tee $O/r03_mp8.txt
python - <<'PY' | tee -a gpurun_out/r03_mp8.txt
import json
lines=[l for l in open("gpurun_out/r03_mp8.json").read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1])
print("value", d.get("value"), "transport", d.get("comm",{}).get("transport"), "rccl", d.get("comm",{}).get("rccl"))
for l in d["comm"]["legs"]:
    print(" leg", l["transport"], l["ok"], l.get("tokens_per_s"), l.get("why"), "wall", round(l.get("wall_s",0),1), "prefill", (l.get("prefill_sharded") or {}).get("ms"))
PY
tail -n 5 $O/r03_mp8.err
