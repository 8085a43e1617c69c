"""A second, independent opinion on the oracle's WHOLE forward pass (the reference's own known-answer
tests cover matmul, the weighted row sum and softmax; rmsnorm, RoPE, GQA attention, SwiGLU and the pass as a whole rested
on one restatement of main.zig:285-430 plus a float64 reading of it).

The reference implements the Llama-2 architecture in llama2.c's checkpoint layout.  Hugging Face `transformers` carries
its own implementation of that architecture (LlamaForCausalLM: written by other people, organised differently -- batched
tensors, a causal mask instead of a loop over timesteps, rotate-half RoPE on permuted q / k rows instead of the
reference's adjacent pairs, main.zig:346-349).  Here the same seeded synthetic checkpoint is loaded into both: the oracle
steps token by token (its KV cache), the HF model sees the whole sequence at once; logits at EVERY position must agree
to fp32 rounding.  CPU only (torch here is the ROCm build without a device); no part of the product is involved.

The q / k permutation is the one llama2.c's exporter undoes when it converts HF weights (export.py permute_reverse) and
HF's own conversion script applies to Meta's: rows (2i, 2i+1) of a head <-> rows (i, i + head_size / 2)."""
import numpy as np
import pytest

CONFIGS = [
    ("gqa-unshared", dict(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=32), False),
    ("mha-shared", dict(dim=48, hidden_dim=128, n_layers=3, n_heads=4, n_kv_heads=4, vocab_size=300, seq_len=24), True),
    ("mqa", dict(dim=96, hidden_dim=256, n_layers=2, n_heads=6, n_kv_heads=1, vocab_size=1000, seq_len=40), True),
    ("head-size-6", dict(dim=36, hidden_dim=100, n_layers=2, n_heads=6, n_kv_heads=3, vocab_size=97, seq_len=16), False),
    ("head-size-128", dict(dim=256, hidden_dim=704, n_layers=2, n_heads=2, n_kv_heads=1, vocab_size=320, seq_len=48), False),
]


def _to_hf(w: np.ndarray, n_heads: int) -> np.ndarray:
    """rows of one head in the reference's order (pairs (2i, 2i+1) rotate together) -> HF's (i with i + hs/2)"""
    rows, cols = w.shape
    hs = rows // n_heads
    return w.reshape(n_heads, hs // 2, 2, cols).transpose(0, 2, 1, 3).reshape(rows, cols)


@pytest.mark.parametrize("name,kw,shared", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_oracle_forward_pass_agrees_with_hf_llama(ck, orc, name, kw, shared):
    torch = pytest.importorskip("torch")
    tf = pytest.importorskip("transformers")
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=4242)
    W = ck.carve(cfg, blob, shared)
    hf_cfg = tf.LlamaConfig(hidden_size=cfg.dim, intermediate_size=cfg.hidden_dim, num_hidden_layers=cfg.n_layers,
                            num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab_size,
                            max_position_embeddings=cfg.seq_len, rms_norm_eps=1e-5, rope_theta=10000.0, hidden_act="silu",
                            tie_word_embeddings=bool(shared), attention_bias=False, mlp_bias=False,
                            head_dim=cfg.dim // cfg.n_heads, attn_implementation="eager")
    torch.manual_seed(0)
    m = tf.LlamaForCausalLM(hf_cfg).eval().float()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sd = {"model.embed_tokens.weight": t(W["token_embedding_table"]), "model.norm.weight": t(W["rms_final_weight"]),
          "lm_head.weight": t(W["token_embedding_table"] if shared else W["wcls"])}
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}."
        sd[p + "input_layernorm.weight"] = t(W["rms_att_weight"][l])
        sd[p + "self_attn.q_proj.weight"] = t(_to_hf(W["wq"][l], cfg.n_heads))
        sd[p + "self_attn.k_proj.weight"] = t(_to_hf(W["wk"][l], cfg.n_kv_heads))
        sd[p + "self_attn.v_proj.weight"] = t(W["wv"][l])
        sd[p + "self_attn.o_proj.weight"] = t(W["wo"][l])
        sd[p + "post_attention_layernorm.weight"] = t(W["rms_ffn_weight"][l])
        sd[p + "mlp.gate_proj.weight"] = t(W["w1"][l])
        sd[p + "mlp.down_proj.weight"] = t(W["w2"][l])
        sd[p + "mlp.up_proj.weight"] = t(W["w3"][l])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    rng = np.random.default_rng(3)
    toks = [1] + rng.integers(0, cfg.vocab_size, cfg.seq_len - 1).tolist()
    with torch.no_grad():
        hf = m(torch.tensor([toks])).logits[0].numpy()
    om = orc.Model(cfg.as_i32(), blob, shared)
    worst = 0.0
    for pos, tok in enumerate(toks):
        ref = om.transformer(tok, pos)
        # two fp32 implementations with different summation orders (BLAS GEMMs vs the reference's lane sums): the
        # bound the GPU is held to against the oracle, tests/test_gpu_parity.py
        np.testing.assert_allclose(ref, hf[pos], rtol=5e-5, atol=5e-5, err_msg=f"{name} pos {pos}")
        assert int(np.argmax(ref)) == int(np.argmax(hf[pos])) or np.sort(ref)[-1] - np.sort(ref)[-2] < 1e-4
        worst = max(worst, float(np.abs(ref - hf[pos]).max()))
    om.close()
    print(f"oracle vs HF LlamaForCausalLM {name}: max |logit diff| over {len(toks)} positions {worst:.2e}")
