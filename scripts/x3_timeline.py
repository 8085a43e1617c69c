"""Where a stream-form GEMM launch of the batched prefill spends its time, from wall-clock stamps (100 MHz) taken by every block
(measurement build: scripts/x3_timeline.sh, -DL2Z_X3_TIMELINE): the last launch of each epilogue kind of one prefill.
usage: L2Z_LIB=llama2.zig_amd/exp/libl2z_x3tl.so x3_timeline.py <shape> <n_tokens>"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
shape, n = sys.argv[1], int(sys.argv[2])
cfg, shared = {k: (c, sh) for k, c, sh in ck.iter_configs()}[shape]
w = B.Weights(cfg, None, shared, seed=1); s = B.RunState(cfg)
toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
for _ in range(3): s.prefill(toks, 0, w)
s.synchronize()
L = B.lib()
buf = np.zeros(8 * 1024 * 8, np.int64)
L.l2z_x3_timeline_dump.argtypes = [C.c_void_p]; L.l2z_x3_timeline_dump.restype = C.c_int
assert L.l2z_x3_timeline_dump(buf.ctypes.data) == 0
t = buf.reshape(8, 1024, 8)
names = {6: "q|k|v", 7: "W1|W3", 1: "W2 (the last launch with the residual epilogue)"}
print(f"# stream-form launches by their blocks' own clocks: {shape}, {n} tokens; us, median over the launch's blocks [min .. max]")
print("# entry: after the launch's first block; the phases: duration")
for epi, nm in names.items():
    a = t[epi]; a = a[a[:, 0] > 0]
    if len(a) == 0: continue
    t0 = a[:, 0].min()
    def st(x): return f"{np.median(x) / 100:6.2f} [{x.min() / 100:6.2f} .. {x.max() / 100:6.2f}]"
    print(f"{nm}: {len(a)} blocks, launch span (first entry -> last end) {(a[:, 7].max() - t0) / 100:.2f} us")
    print(f"  entry after the first block      {st(a[:, 0] - t0)}")
    print(f"  setup + first stage landed       {st(a[:, 1] - a[:, 0])}")
    print(f"  the stage loop                   {st(a[:, 2] - a[:, 1])}")
    print(f"  k-groups summed (LDS)            {st(a[:, 3] - a[:, 2])}")
    if (a[:, 4] > 0).all():
        print(f"  partial sums written + drained   {st(a[:, 4] - a[:, 3])}")
        print(f"  arrival counted, siblings waited {st(a[:, 5] - a[:, 4])}")
        print(f"  the ranges' sums read and added  {st(a[:, 6] - a[:, 5])}")
        print(f"  epilogue                         {st(a[:, 7] - a[:, 6])}")
    else:
        print(f"  epilogue (one K range)           {st(a[:, 7] - a[:, 3])}")
    print(f"  end after the first block's entry {st(a[:, 7] - t0)}")
