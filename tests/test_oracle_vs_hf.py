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
    # BASELINE configs 1-2 at full shape (6 layers, 32000-word vocabulary, shared classifier), 96 positions
    ("stories15M-shape", dict(dim=288, hidden_dim=768, n_layers=6, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=256), True),
]


@pytest.mark.parametrize("name,kw,shared", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_oracle_forward_pass_agrees_with_hf_llama(ck, orc, name, kw, shared):
    pytest.importorskip("torch")
    pytest.importorskip("transformers")
    import hf_llama
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=4242)
    m = hf_llama.build(ck, cfg, blob, shared)
    rng = np.random.default_rng(3)
    toks = [1] + rng.integers(0, cfg.vocab_size, min(cfg.seq_len, 96) - 1).tolist()
    hf = hf_llama.logits(m, toks)
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
