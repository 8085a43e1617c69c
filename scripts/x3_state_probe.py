"""Does the 1024-token prefill time depend on what the process did before?  (perf_floor.py read 82 ms where the isolated A/B reads 75.)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
import numpy as np, __graft_entry__ as ge, perf_floor
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg = ck.LLAMA2_7B
w, s = B.Weights(cfg, None, False, seed=2024), B.RunState(cfg)
print("1024 first                 : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s, cfg, 1024))
print("512                        : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s, cfg, 512))
for n in (16, 32, 48, 64, 96, 128, 256):
    perf_floor.prefill_ms(B, ck, w, s, cfg, n)
print("1024 after the short ones  : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s, cfg, 1024))
perf_floor.decode_kinds(B, ck, "llama2-7b", 8, w, s)
print("1024 after decode timing   : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s, cfg, 1024))
s2 = B.RunState(cfg)
print("1024, a fresh RunState     : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s2, cfg, 1024))
time.sleep(5)
print("1024 after 5 s idle        : %.2f ms" % perf_floor.prefill_ms(B, ck, w, s2, cfg, 1024))
xs = []
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, 1023).tolist()
for i in range(30):
    t0 = time.perf_counter(); s2.prefill(toks, 0, w); xs.append((time.perf_counter() - t0) * 1e3)
print("30 in a row: " + " ".join("%.1f" % x for x in xs))
