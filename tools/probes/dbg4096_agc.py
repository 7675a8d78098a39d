import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import composable_sdr_amd as cs, oracle_lib as O
from composable_sdr_amd import _lib
from synth import synth_cf32
M, kf = 4096, 0.3
nfs = [40, 9, 250]
x = synth_cf32(M * sum(nfs), M, seed=4096)
x = (x + np.complex64(0.02 - 0.01j)).astype(np.complex64)
deno = cs.Chain(channels=M, max_frames=max(nfs))
rows = np.concatenate([deno.process(x[sum(nfs[:i]) * M:sum(nfs[:i + 1]) * M]) for i in range(len(nfs))], axis=1)
print("deno plane finite:", np.isfinite(rows.view(np.float32)).all(), "max", np.abs(rows).max())
r = 567
z = np.ascontiguousarray(rows[r])
pipe = cs.automaticGainControl(23.0, max_samples=z.size)
st = pipe._start(); yg = pipe._process(st, z); pipe._done(st)
yo = O.Agc(23.0).execute_block(z)
bad = np.nonzero(~np.isfinite(yg.view(np.float32).reshape(-1, 2)).all(axis=1))[0]
print("row", r, "gpu nonfinite at", bad[:10], "oracle finite", np.isfinite(yo.view(np.float32)).all())
lo = max(0, (bad[0] if bad.size else 100) - 6)
for i in range(lo, lo + 14):
    print(i, "x", z[i], "|x|", abs(z[i]), "gpu", yg[i], "orc", yo[i])
# gain trajectory of the oracle around there
a = O.Agc(23.0)
for i in range(lo + 10):
    a.execute_block(z[i:i + 1])
    if i >= lo - 2: print(i, a.state)
