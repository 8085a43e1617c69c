import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.Config(4096, 11008, 6, 32, 32, 32000, 2048)   # 7B dims, 6 layers: 5.9 GB
n = ck.weights_count(cfg, False)
blob = np.zeros(n, np.float32); blob[::4096] = 1.0
for rep in range(2):
    t0 = time.perf_counter(); w = B.Weights(cfg, blob, False); dt = time.perf_counter() - t0
    print(f"l2z_weights_init from a pageable host blob: {n*4/1e9:.2f} GB in {dt:.2f} s = {n*4/dt/1e9:.1f} GB/s")
    w.close()
# via an mmapped file (what the CLI does)
path = "/tmp/up.bin"; ck.write_checkpoint(path, cfg, blob, False)
c2, sh, mm = ck.read_checkpoint(path)
t0 = time.perf_counter(); w = B.Weights(c2, np.asarray(mm), sh); dt = time.perf_counter() - t0
print(f"from an mmapped checkpoint file (page cache warm): {n*4/dt/1e9:.1f} GB/s")
os.remove(path)
