"""The C++ CLI on a full-size (27 GB) llama2-7b-shape synthetic checkpoint file in /dev/shm: tokens/s as the CLI
reports them (the reference's rule, main.zig:1043-1050) at -t 0 and with the reference's default sampling, a short
prompt and a ~500-token one (tokenizer + batched prefill + decode)."""
import os, re, subprocess, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); ck = pkg.checkpoint
orc = ge.load_oracle()
cfg, shared = ck.LLAMA2_7B, False
path = "/dev/shm/l2z_cli7b.bin"
t0 = time.perf_counter()
blob = orc.synth_fill(cfg.as_i32(), shared, 7, os.cpu_count() or 1)
ck.write_checkpoint(path, cfg, blob, shared)
del blob
print(f"wrote {os.path.getsize(path) / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s", flush=True)
root = os.environ.get("GRAFT_REPO_ROOT", ".")
exe, tok = os.path.join(root, "llama2.zig_amd", "host", "llama2"), os.path.join(root, "tests", "golden", "tokenizer.bin")
long_prompt = "Once upon a time, there was a little girl named Lily who loved to play outside in the sunshine. " * 21
try:
    for name, args in (("-t 0, short prompt", ["-t", "0", "-i", "Once upon a time"]),
                       ("defaults (-t 1.0 -p 0.9), short prompt", ["-i", "Once upon a time"]),
                       ("-t 0, ~500-token prompt, 700 steps", ["-t", "0", "-n", "700", "-i", long_prompt]),
                       ("-t 0, ~500-token prompt, L2Z_PREFILL=0 (stepped)", ["-t", "0", "-n", "700", "-i", long_prompt, "ENV:L2Z_PREFILL=0"])):
        env = dict(os.environ)
        argv = [a for a in args if not a.startswith("ENV:")]
        for a in args:
            if a.startswith("ENV:"):
                k, v = a[4:].split("=")
                env[k] = v
        if "-n" not in argv: argv += ["-n", "256"]
        t0 = time.perf_counter()
        r = subprocess.run([exe, path, "-z", tok, "-s", "7", "-v"] + argv, capture_output=True, text=True, env=env, timeout=900)
        m = re.search(r"(\d+) tokens per second", r.stderr)
        print(f"7B shape, {name}: {m.group(1) if m else 'rc %d' % r.returncode} tok/s (CLI), {time.perf_counter() - t0:.1f} s wall incl. 27 GB upload", flush=True)
finally:
    os.remove(path)
