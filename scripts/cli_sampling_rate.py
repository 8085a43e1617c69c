"""Tokens/s of the CLI on a synthetic checkpoint file at -t 0 (device loop) and with the reference's default
sampling (-t 1.0 -p 0.9: logits to the host every token, softmax + top-p there).  cli_sampling_rate.py [shape] [steps]"""
import os, subprocess, sys, tempfile, time, re
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); ck = pkg.checkpoint
shape = sys.argv[1] if len(sys.argv) > 1 else "stories110M"
steps = sys.argv[2] if len(sys.argv) > 2 else "256"
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
cli = os.path.join(ROOT, "llama2.zig_amd", "host", "llama2")
tok = os.path.join(ROOT, "tests", "golden", "tokenizer.bin")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    path = os.path.join(d, "m.bin")
    ck.write_checkpoint(path, cfg, ck.synth_blob(cfg, shared, 1), shared)
    for name, extra in (("-t 0", ["-t", "0"]), ("-t 1.0 -p 0.9 (defaults)", []), ("-t 1.0 -p 1.0", ["-p", "1.0"]), ("-t 0.8 -p 0.9", ["-t", "0.8"])):
        best = None
        for rep in range(3):
            out = subprocess.run([cli, path, "-z", tok, "-n", steps, "-s", "7", "-v", "-i", "Once upon a time"] + extra,
                                 capture_output=True, text=True)
            m = re.search(r"(\d+) tokens per second", out.stdout + out.stderr)
            if m: best = max(best or 0.0, float(m.group(1)))
            elif rep == 0: print("rc", out.returncode, "stderr:", repr(out.stderr[-300:]))
        print(f"{shape} {name}: {best} tok/s (best of 3, as the CLI reports)")
