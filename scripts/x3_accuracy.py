"""Error of ONE prefill GEMM against float64, f32 matrix cores vs bf16 three-term split (L2Z_PF_X3):
x3_accuracy.py <shape> <n_tokens>.  The value-cache rows of layer 0 after a batched prefill are
Wv . rmsnorm(embedding row) -- one [P, dim] x [kv_dim, dim]^T product of the tile / panel kernels -- and the
host regenerates the synthetic tensors, so the float64 truth costs one numpy matmul at any shape."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
shape, n = sys.argv[1], int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); B.option_set(k, int(v)); print("  ", kv)
seed = 7
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
w = B.Weights(cfg, None, shared, seed=seed)
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
t = {t.name: t for t in ck.tensor_table(cfg, shared)}
def tensor(name, row0, rows, width):
    return ck.synth_values(t[name].offset + row0 * width, rows * width, seed, t[name].scale, t[name].bias).reshape(rows, width)
rms = tensor("rms_att_weight", 0, 1, cfg.dim)[0].astype(np.float64)
wv = tensor("wv", 0, cfg.kv_dim, cfg.dim).astype(np.float64)
emb = np.stack([tensor("token_embedding_table", tk, 1, cfg.dim)[0] for tk in toks]).astype(np.float64)
xn64 = emb * (1.0 / np.sqrt((emb * emb).mean(axis=1, keepdims=True) + 1e-5)) * rms
xn32 = xn64.astype(np.float32).astype(np.float64)       # the GEMM's input as the device holds it (up to the rmsnorm's own rounding)
truth = xn32 @ wv.T
scale = np.abs(xn32) @ np.abs(wv.T)                      # sum |a_i b_i| per output
f32chain = (xn32.astype(np.float32) @ wv.T.astype(np.float32)).astype(np.float64)
print(f"{shape}: V rows of layer 0, {n} tokens x {cfg.kv_dim} features, K = {cfg.dim}")
print(f"  numpy f32 matmul          : max |err| / sum|ab| = {np.abs(f32chain - truth).max() / 1:.3e} abs, {(np.abs(f32chain - truth) / scale).max():.3e} rel, rms rel {np.sqrt(((f32chain - truth) ** 2).mean()) / np.sqrt((truth ** 2).mean()):.3e}")
S, kvd = cfg.seq_len, cfg.kv_dim
for x3 in (0, 1):
    B.option_set("L2Z_PF_X3", x3)
    s = B.RunState(cfg)
    s.prefill(toks, 0, w)
    got = s.read("value_cache", 0, S * kvd).astype(np.float64)
    # device layout is permuted back to (pos, kv_dim) by runstate_read
    got = got.reshape(S, kvd)[:n]
    err = got - truth
    print(f"  L2Z_PF_X3={x3} ({'bf16 x 3 split, 6 products' if x3 else 'f32 MFMA chain'}): max |err| {np.abs(err).max():.3e}, max |err| / sum|ab| {(np.abs(err) / scale).max():.3e}, rms err / rms value {np.sqrt((err ** 2).mean()) / np.sqrt((truth ** 2).mean()):.3e}, mean err * sign(value) / mean |value| {(err * np.sign(truth)).mean() / np.abs(truth).mean():+.3e}")
    s.close()
