"""Host -> device rate of l2z_weights_init at FULL size: a 27 GB llama2-7b-shape checkpoint file
(seeded synthetic weights, written by checkpoint.py's writer into /dev/shm) is mmapped and uploaded
the way the CLI does it (host/llama2_main.cpp); then once more through the CLI itself (-v prints the time),
single GPU and `-g 2` (each rank uploads its own rows).  Spot-checks the device copy against the file."""
import os, subprocess, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
orc = ge.load_oracle()
cfg, shared = ck.LLAMA2_7B, False
if os.environ.get("UP_LAYERS"):  # smaller run for hosts without ~70 GB of RAM
    cfg = ck.Config(cfg.dim, cfg.hidden_dim, int(os.environ["UP_LAYERS"]), cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size, cfg.seq_len)
path = "/dev/shm/l2z_upload_test.bin"
t0 = time.perf_counter()
blob = orc.synth_fill(cfg.as_i32(), shared, 7, os.cpu_count() or 1)
ck.write_checkpoint(path, cfg, blob, shared)
n = blob.size
probe = [(o, blob[o:o + 1024].copy()) for o in (0, n // 3, n - 1024)]
del blob
print(f"wrote {os.path.getsize(path) / 1e9:.2f} GB to {path} in {time.perf_counter() - t0:.1f} s", flush=True)
c2, sh, mm = ck.read_checkpoint(path)
for rep in range(3):
    t0 = time.perf_counter(); w = B.Weights(c2, np.asarray(mm), sh); dt = time.perf_counter() - t0
    ok = all(np.array_equal(w.read(o, 1024), v) for o, v in probe)
    print(f"l2z_weights_init from the mmapped file: "
          f"{n * 4 / 1e9:.2f} GB in {dt:.2f} s = {n * 4 / dt / 1e9:.1f} GB/s  (device copy == file: {ok})", flush=True)
    w.close()
del mm
exe = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "llama2.zig_amd", "host", "llama2")
tok = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests", "golden", "tokenizer.bin")
for g in (1, 2):
    env = dict(os.environ, L2Z_GRID_CAP="256", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.perf_counter()
    r = subprocess.run([exe, path, "-t", "0", "-n", "8", "-v", "-z", tok, "-g", str(g), "--tokens"], capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in r.stderr.splitlines() if l.startswith(("weights:", "tokens:")) or "tokens per second" in l or "error" in l]
    print(f"CLI -g {g} (rc {r.returncode}, {time.perf_counter() - t0:.1f} s wall): " + " | ".join(lines), flush=True)
os.remove(path)
