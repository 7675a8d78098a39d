import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import composable_sdr_amd as cs
from synth import synth_cf32
M = 1024
frames = [int(v) for v in sys.argv[1:]] or [37, 200]
x = synth_cf32(M * sum(frames), M, seed=31)
kw = dict(channels=M, demod="fm", kf=0.3, max_frames=max(frames))
a = cs.Chain(**kw)
os.environ["CSDR_RUN1024_V1"] = "1"
b = cs.Chain(**kw)
pos = 0
for f in frames:
    xa = x[pos * M:(pos + f) * M]; pos += f
    ga = a.process(xa); gb = b.process(xa)
    d = np.abs(ga.astype(np.float64) - gb)
    d = np.minimum(d, np.abs(d - 1 / 0.3))
    bad = np.argwhere(d > 1e-3)
    cols = sorted(set(bad[:, 1].tolist()))
    print(f, a.kernel_time()[0], "bad samples", len(bad), "columns", cols[:20], "rows of first col", bad[bad[:, 1] == cols[0]][:8, 0].tolist() if cols else [])
    if cols:
        rows = sorted(set(bad[bad[:, 1] == cols[0]][:, 0].tolist()))
        print("  all rows of col", cols[0], ":", rows[:80], "n", len(rows))
        r0 = rows[0]
        print("  v2:", ga[r0, :6], " v1:", gb[r0, :6])
