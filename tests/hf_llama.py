"""Hugging Face's LlamaForCausalLM loaded with a llama2.c-layout checkpoint (tests/test_oracle_vs_hf.py says why): an
implementation of the reference's architecture written by other people, used as an independent opinion on whole-pass
logits.  CPU, fp32.  Test infrastructure only."""
import numpy as np


def to_hf_rows(w: np.ndarray, n_heads: int) -> np.ndarray:
    """rows of one head in the reference's order (pairs (2i, 2i+1) rotate together, main.zig:346-349) -> HF's
    rotate-half order (i with i + head_size / 2): the permutation llama2.c's exporter undoes (export.py permute_reverse)"""
    rows, cols = w.shape
    hs = rows // n_heads
    return w.reshape(n_heads, hs // 2, 2, cols).transpose(0, 2, 1, 3).reshape(rows, cols)


def build(ck, cfg, blob, shared):
    """-> LlamaForCausalLM (eval, fp32, eager attention) holding the checkpoint's tensors"""
    import torch
    import transformers as tf
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
        sd[p + "self_attn.q_proj.weight"] = t(to_hf_rows(W["wq"][l], cfg.n_heads))
        sd[p + "self_attn.k_proj.weight"] = t(to_hf_rows(W["wk"][l], cfg.n_kv_heads))
        sd[p + "self_attn.v_proj.weight"] = t(W["wv"][l])
        sd[p + "self_attn.o_proj.weight"] = t(W["wo"][l])
        sd[p + "post_attention_layernorm.weight"] = t(W["rms_ffn_weight"][l])
        sd[p + "mlp.gate_proj.weight"] = t(W["w1"][l])
        sd[p + "mlp.down_proj.weight"] = t(W["w2"][l])
        sd[p + "mlp.up_proj.weight"] = t(W["w3"][l])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return m


def logits(m, toks) -> np.ndarray:
    """[len(toks), vocab]: the whole sequence in one causal pass"""
    import torch
    with torch.no_grad():
        return m(torch.tensor([list(toks)])).logits[0].numpy()


def logits_in_subprocess(kw: dict, shared: bool, seed: int, toks) -> np.ndarray:
    """The same, computed by a CHILD process: importing torch (its own HIP runtime and RCCL copies) into a process that
    also drives libllama2_hip.so breaks that library's RCCL initialisation later on -- GPU tests keep torch out."""
    import json, os, subprocess, sys, tempfile
    with tempfile.TemporaryDirectory() as d:
        spec = os.path.join(d, "spec.json")
        json.dump(dict(kw=kw, shared=bool(shared), seed=int(seed), toks=[int(t) for t in toks], out=os.path.join(d, "hf.npy")), open(spec, "w"))
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        subprocess.run([sys.executable, os.path.abspath(__file__), spec], check=True, env=env, timeout=600)
        return np.load(os.path.join(d, "hf.npy"))


if __name__ == "__main__":
    import json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    spec = json.load(open(sys.argv[1]))
    ck = ge.load_package().checkpoint
    cfg = ck.Config(**spec["kw"])
    m = build(ck, cfg, ck.synth_blob(cfg, spec["shared"], spec["seed"]), spec["shared"])
    np.save(spec["out"], logits(m, spec["toks"]))

