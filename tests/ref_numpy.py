"""Independent float64 numpy restatement of the forward pass (src/main.zig:285-430).

A second opinion on the C oracle: written from the reference's control flow,
sharing no code with oracle/llama2_oracle.c, evaluated in float64 with plain
numpy reductions.  It cannot be bit-identical to any f32 reading of the
reference; tests compare at ~1e-4 relative, which catches indexing, layout,
GQA, RoPE-convention and eps-placement mistakes (the things an author would
get wrong the same way twice in C and HIP).
"""
import numpy as np


class NumpyModel:
    def __init__(self, ck, cfg, blob, shared):
        self.c = cfg
        self.w = {k: v.astype(np.float64) for k, v in ck.carve(cfg, blob, shared).items()}
        L, S, kvd = cfg.n_layers, cfg.seq_len, cfg.kv_dim
        self.kc = np.zeros((L, S, kvd))
        self.vc = np.zeros((L, S, kvd))

    @staticmethod
    def rmsnorm(x, w):
        # main.zig:452-454: sum/n, THEN +1e-5, then 1/sqrt
        return x * (1.0 / np.sqrt(np.mean(x * x) + 1e-5)) * w

    def transformer(self, token, pos):
        c, w = self.c, self.w
        hs, kvd, kv_mul = c.head_size, c.kv_dim, c.kv_mul
        x = w["token_embedding_table"][token].copy()
        for l in range(c.n_layers):
            xb = self.rmsnorm(x, w["rms_att_weight"][l])
            q = w["wq"][l] @ xb
            k = w["wk"][l] @ xb
            v = w["wv"][l] @ xb
            # RoPE main.zig:336-351: interleaved pairs (i, i+1), freq from (i % head_size)
            for i in range(0, c.dim, 2):
                freq = 1.0 / (10000.0 ** ((i % hs) / hs))
                fcr, fci = np.cos(pos * freq), np.sin(pos * freq)
                q[i], q[i + 1] = q[i] * fcr - q[i + 1] * fci, q[i] * fci + q[i + 1] * fcr
                if i < kvd:
                    k[i], k[i + 1] = k[i] * fcr - k[i + 1] * fci, k[i] * fci + k[i + 1] * fcr
            self.kc[l, pos], self.vc[l, pos] = k, v
            out = np.zeros(c.dim)
            for h in range(c.n_heads):
                kh = (h // kv_mul) * hs
                K = self.kc[l, : pos + 1, kh : kh + hs]
                V = self.vc[l, : pos + 1, kh : kh + hs]
                att = (K @ q[h * hs : (h + 1) * hs]) / np.sqrt(hs)
                att = np.exp(att - att.max())
                att /= att.sum()
                out[h * hs : (h + 1) * hs] = att @ V
            x = x + w["wo"][l] @ out
            xb = self.rmsnorm(x, w["rms_ffn_weight"][l])
            h1 = w["w1"][l] @ xb
            h3 = w["w3"][l] @ xb
            x = x + w["w2"][l] @ (h1 * (1.0 / (1.0 + np.exp(-h1))) * h3)
        x = self.rmsnorm(x, w["rms_final_weight"])
        return w["wcls"] @ x
