#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/.

  reference_kats.json   the reference's own known-answer vectors, as data
                        (inputs + expected outputs of src/main.zig:1078-1103)
  toy_*.bin             small seeded llama2.c-v0 checkpoints (<= 300 KB each)
  toy_*.npz             what the C oracle (default reading: VW=8, no FMA,
                        sequential @reduce) produces for them: greedy token ids
                        and the logits of the first positions

The reference itself cannot be run in the build image (no Zig 0.16 compiler,
no checkpoint), so the toy expectations come from the oracle: they pin the
oracle against regressions and give the GPU tests a fixed target, but they are
NOT outputs of the Zig binary -- "parity unpinned" for transformer() end to end
(see oracle/llama2_oracle.h).

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ck = ge.load_package().checkpoint
orc = ge.load_oracle()
orc.set_mode(8, False, False)

kats = {"source": "cgbur/llama2.zig src/main.zig:1078-1103 (test blocks), restated as data",
        "matmul": [
            {"name": "matrix_multiplies", "d": 3, "n": 3, "w": list(range(1, 10)), "x": [1, 2, 3],
             "expect": [14.0, 32.0, 50.0]},
            {"name": "vector_length_less_than_width_case", "d": 2, "n": 12,
             "w": list(range(1, 25)), "x": list(range(1, 13)),
             "expect": [float(sum((i * 12 + j + 1) * (j + 1) for j in range(12))) for i in range(2)]},
        ]}
json.dump(kats, open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)

MODELS = [
    ("toy_gqa_unshared", dict(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2,
                              vocab_size=512, seq_len=32), False, 1001, []),
    ("toy_mha_shared", dict(dim=48, hidden_dim=128, n_layers=3, n_heads=4, n_kv_heads=4,
                            vocab_size=300, seq_len=24), True, 1002, [7, 11, 13]),
]
meta = {"oracle_mode": [8, 0, 0], "models": []}
for name, kw, shared, seed, prompt in MODELS:
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed)
    ck.write_checkpoint(os.path.join(HERE, name + ".bin"), cfg, blob, shared)
    m = orc.Model(cfg.as_i32(), blob, shared)
    toks, margins = m.generate_greedy(prompt, cfg.seq_len)
    fed = [1] + toks[:7].tolist()  # the tokens transformer() sees at pos 0..7
    m2 = orc.Model(cfg.as_i32(), blob, shared)
    logits = np.stack([m2.transformer(t, p) for p, t in enumerate(fed)])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), tokens=toks, margins=margins,
                        fed_tokens=np.array(fed, np.int32), logits=logits)
    meta["models"].append({"checkpoint": name + ".bin", "expected": name + ".npz", "shared": shared,
                           "seed": seed, "prompt": prompt, "config": kw,
                           "min_margin": float(margins.min())})
    print(name, "tokens", toks[:10], "min margin", margins.min())
json.dump(meta, open(os.path.join(HERE, "toy_models.json"), "w"), indent=1)
