"""The opt-in forms of the decode pass built in round 4 against the default launch chain, through the C
ABI on the GPU: the duo mat-vecs (L2Z_DUO), the two-chain overlapped pass (L2Z_DUO + L2Z_OVERLAP) and the persistent
launches (L2Z_ENGINE).  Each of them keeps the default chain's units, thread -> column map, summation order and
epilogues, so logits and greedy tokens must be IDENTICAL -- at positions where the attention form is the same (the duo
forms take 256-thread attention blocks everywhere: different bits from pos 128 on at head size 128, compared there at the
logit tolerance).  A wide-row shape: dim 4096 (the narrowest the duo / engine kernels take), two layers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WIDE = dict(dim=4096, hidden_dim=11008, n_layers=2, n_heads=32, n_kv_heads=32, vocab_size=32000, seq_len=384)
WIDE_GQA = dict(dim=4096, hidden_dim=8192, n_layers=3, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=320)  # tests/test_gpu_p2p.py's wide shape
RESET = {"L2Z_DUO": 0, "L2Z_OVERLAP": 1, "L2Z_ENGINE": 0, "L2Z_OVERLAP_EDGES": 15}


def _want_form(opts):
    """l2z_runstate_form's bits for a set of options (a form refused by the shape would pass the comparison trivially)."""
    if opts.get("L2Z_ENGINE"):
        return 4
    if opts.get("L2Z_DUO"):
        return 3 if opts.get("L2Z_OVERLAP", 1) else 1
    return 0


def _run(B, cfg, w, opts, n_tok, probe_pos):
    for k, v in opts.items():
        B.option_set(k, v)
    try:
        s = B.RunState(cfg)
    finally:
        for k in opts:
            B.option_set(k, RESET[k])
    assert s.form() == _want_form(opts), f"{opts}: the runstate runs form {s.form()}"
    B.option_set("L2Z_PREFILL", 0)
    s.greedy_begin([])
    toks = np.array(s.greedy_run(w, n_tok))
    logits = []
    for pos in probe_pos:
        s.transformer(int(toks[pos - 1]) if 0 < pos <= len(toks) else 1, pos, w)
        logits.append(s.logits().copy())
    s.close()
    return toks, logits


@pytest.mark.parametrize("opts", [{"L2Z_DUO": 1, "L2Z_OVERLAP": 0}, {"L2Z_DUO": 1, "L2Z_OVERLAP": 1},
                                  {"L2Z_DUO": 1, "L2Z_OVERLAP": 1, "L2Z_OVERLAP_EDGES": 9}, {"L2Z_ENGINE": 1}],
                         ids=["duo", "duo+overlap", "duo+overlap-edges-9", "engine"])
@pytest.mark.parametrize("shape", [WIDE, WIDE_GQA], ids=["mha-11008", "gqa-8192"])
def test_opt_in_forms_keep_the_chains_bits(gpu, ck, opts, shape):
    B = gpu
    cfg = ck.Config(**shape)
    w = B.Weights(cfg, None, False, seed=77)
    probe = [0, 5, 60, 127]          # the attention form is the default chain's below pos 128 in every mode
    ref_t, ref_l = _run(B, cfg, w, {}, 140, probe)
    got_t, got_l = _run(B, cfg, w, opts, 140, probe)
    engine = "L2Z_ENGINE" in opts    # the persistent launches keep the default attention forms at every position
    n_same = 140 if engine else 128
    assert np.array_equal(got_t[:n_same], ref_t[:n_same]), f"{opts}: greedy tokens differ"
    for pos, a, b in zip(probe, got_l, ref_l):
        assert np.array_equal(a, b), f"{opts}: logits at pos {pos} differ by {np.abs(a - b).max():.3g}"
    if not engine:                   # beyond: other attention blocks, same mathematics
        _, l_ref = _run(B, cfg, w, {}, 0, [200])
        _, l_got = _run(B, cfg, w, opts, 0, [200])
        np.testing.assert_allclose(l_got[0], l_ref[0], rtol=5e-5, atol=5e-5)
    w.close()
