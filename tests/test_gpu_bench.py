"""bench.py's contract at N = 1 on the GPU box: ONE JSON line with the fields the driver reads, the
`roofline` and `cpu_baseline` objects, and round 3's hygiene fields -- on a small workload so that it runs in
seconds (the headline 7B run is the driver's own)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stories110M", "--steps", "64",
                        "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 64 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "tokens/s" and out["dtype"] == "f32" and out["data"] == "synthetic" and out["vs_baseline"] is None
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert "workload" in out["config"] and "stories110M" in out["config"]["workload"]
    r = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "whole_token_frac", "kernel",
              "algorithmic_bytes_per_launch", "avg_launch_ms", "stream_read_probe", "d2d_copy_probe"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or "NOT read in this run" in r["traffic_source"]
    assert "NOT a ceiling" in r["stream_read_probe"]["note"]
    assert r["d2d_copy_probe"]["copied_avg"] and r["d2d_copy_probe"]["copied_avg"] > 100.0  # GB/s copied; traffic is twice that
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "oracle" in c["sample"]
    # round 6: the line names the host CPU and the build of the port that was timed (BASELINE.md's plan)
    assert c["cpu_model"] and c["cpu_model"] in c["sample"] and "-march=" in c["build"] and c["build"] in c["sample"]
    by_shape = out["extra"]["cpu_baseline_by_shape"]
    assert set(by_shape) == {"stories15M", "stories110M"} and all(v["value"] > 0 and v["cores"] == 1 for v in by_shape.values())
    assert out["extra"]["stories15M_tokens_per_s"] > 0 and out["extra"]["prefill"]["roofline"]["bound"] == "mfma"


@pytest.mark.parametrize("leg_name", ["rccl", "rccl-allreduce"])
def test_rccl_leg_runs_with_one_rank(gpu, leg_name):
    """The RCCL leg of `bench.py --gpus N` as the driver would start it, with ONE rank (RCCL refuses two ranks on one
    device): torch imported first (gloo control plane), then this library and its dlopen of RCCL -- the order the legs
    use -- ncclCommInitRank, captured ncclAllGather per vector (scheme B: ncclAllReduce of the partial vectors), the leg's record.  L2Z_BENCH_FORCE_DIST=1 makes the
    single-GPU invocation take the multi-rank path."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, L2Z_BENCH_FORCE_DIST="1", L2Z_COMM=leg_name, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stories110M", "--steps", "16",
                        "--warmup", "1", "--no-extra"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout.decode()[-2000:] + p.stderr.decode()[-2000:]
    out = json.loads(lines[0])
    leg = out["comm"]["legs"][0]
    assert leg["transport"] == leg_name and leg["ok"] and leg["rccl_ranks"] == [1], leg
    assert leg["scheme"] == ("B" if leg_name == "rccl-allreduce" else "A") and bool(leg["runstate_form"] & 8) == (leg["scheme"] == "B")
    assert leg["rccl_library"]["version"] > 20000 and "rccl" in leg["rccl_library"]["path"], leg["rccl_library"]
    assert out["value"] > 0 and (leg_name != "rccl" or out["comm"]["rccl"]["initialised"])
    print("RCCL library of the leg:", leg["rccl_library"])
