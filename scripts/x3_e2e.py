"""End-to-end error against float64: the last position's logits of a 300-token prompt on the stories110M dims, from
(a) the stepped HIP loop, (b) the batched prefill on the f32 matrix cores, (c) the batched prefill on the bf16 ones
(three-term split), each against tests/ref_numpy.py's float64 model.  x3_e2e.py [n_tokens]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np, __graft_entry__ as ge
from ref_numpy import NumpyModel
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
c0 = ck.STORIES110M
cfg = ck.Config(c0.dim, c0.hidden_dim, c0.n_layers, c0.n_heads, c0.n_kv_heads, 4096, n + 4)
blob = ck.synth_blob(cfg, True, 77)
toks = [1] + np.random.default_rng(5).integers(2, cfg.vocab_size, n - 1).tolist()
m = NumpyModel(ck, cfg, blob, True)
for pos, t in enumerate(toks):
    ref = m.transformer(t, pos)
w = B.Weights(cfg, blob, True)
def report(tag, got):
    err = got.astype(np.float64) - ref
    print(f"  {tag:46s}: max |err| {np.abs(err).max():.3e}  rms err {np.sqrt((err**2).mean()):.3e}  mean err {err.mean():+.3e}  (rms logit {np.sqrt((ref**2).mean()):.3f})")
print(f"110M dims, vocab 4096, {n} tokens, last position's logits against float64:")
s = B.RunState(cfg)
for pos, t in enumerate(toks):
    s.transformer(t, pos, w)
step = s.logits(); report("stepped loop (f32 mat-vec kernels)", step); s.close()
res = {}
for x3 in (0, 1):
    B.option_set("L2Z_PF_X3", x3)
    s = B.RunState(cfg); s.prefill(toks, 0, w); res[x3] = s.logits(); s.close()
    report(f"batched prefill, L2Z_PF_X3={x3}", res[x3])
print(f"  prefill f32 cores vs stepped: {np.abs(res[0] - step).max():.3e}   prefill bf16 cores vs stepped: {np.abs(res[1] - step).max():.3e}   f32 cores vs bf16 cores: {np.abs(res[0] - res[1]).max():.3e}")
