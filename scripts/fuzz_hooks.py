"""The kernel-level entry points (l2z_matmul, _fused, rmsnorm, softmax, dot, weighted row sum, argmax)
on random sizes -- unaligned, tiny, around every kernel-selection threshold -- against the oracle.
usage: fuzz_hooks.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B = pkg.binding; orc = ge.load_oracle()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0); bad = 0
NS = [1, 2, 3, 4, 5, 7, 8, 12, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 288, 300, 511, 512, 516, 767, 768, 772,
      1020, 1024, 1028, 1152, 2044, 2048, 4092, 4096, 4100, 4352, 8192, 11008, 11012]
def relerr(got, ref, scale):
    return float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64)) / scale))
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    n = int(rng.choice(NS)); d = int(rng.choice([1, 2, 3, 5, 16, 33, 64, 257, 1000, 2049]))
    x = rng.standard_normal(n, dtype=np.float32)
    w = (rng.standard_normal((d, n), dtype=np.float32) / np.float32(np.sqrt(n)))
    absdot = np.abs(w.astype(np.float64)) @ np.abs(x.astype(np.float64)) + 1e-30
    ok = relerr(B.matmul(x, w), orc.matmul(x, w), absdot) <= 4e-6
    N = int(rng.choice([2, 3])); ws = [w] + [(rng.standard_normal((d, n), dtype=np.float32) / np.float32(np.sqrt(n))) for _ in range(N - 1)]
    gf, rf = B.matmul_fused(x, ws), orc.matmul_fused(x, ws)
    for g, r, ww in zip(gf, rf, ws):
        ok = ok and relerr(g, r, np.abs(ww.astype(np.float64)) @ np.abs(x.astype(np.float64)) + 1e-30) <= 4e-6
    rw = rng.standard_normal(n, dtype=np.float32)
    ok = ok and np.allclose(B.rmsnorm(x, rw), orc.rmsnorm(x, rw), rtol=2e-6, atol=2e-6)
    sm = rng.standard_normal(n, dtype=np.float32) * 3
    ok = ok and np.allclose(B.softmax(sm), orc.softmax(sm), rtol=1e-5, atol=1e-7 * np.sqrt(n))
    y = rng.standard_normal(n, dtype=np.float32)
    ok = ok and abs(float(B.vector_dot_product(x, y)) - float(orc.vector_dot_product(x, y))) <= 4e-6 * float(np.abs(x.astype(np.float64)) @ np.abs(y.astype(np.float64)) + 1e-30)
    hs = int(rng.choice([1, 3, 11, 48, 64, 128, 130])); stride = hs + int(rng.integers(0, 40)); T = int(rng.choice([1, 2, 9, 77, 300]))
    rows = rng.standard_normal(T * stride, dtype=np.float32); wt = rng.random(T, dtype=np.float32)
    ok = ok and np.allclose(B.vector_weighted_sum_rows(hs, rows, stride, wt), orc.vector_weighted_sum_rows(hs, rows, stride, wt), rtol=1e-5, atol=1e-5)
    v = rng.integers(-3, 4, n).astype(np.float32)  # many ties
    ok = ok and B.argmax(v) == int(orc.argmax(v))
    print(("ok " if ok else "BAD"), "n", n, "d", d, "N", N, "hs", hs, "stride", stride, "T", T); bad += not ok
print("bad:", bad)
