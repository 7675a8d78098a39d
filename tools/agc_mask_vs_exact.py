"""new FM+AGC route (energy words + mask pass) against the round-1 route (CF32 plane + k_agc_spec), same inputs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import composable_sdr_amd as cs
from synth import synth_cf32
M = 256
frames = [int(v) for v in sys.argv[1:]] or [4096, 33, 2048]
x = synth_cf32(M * sum(frames), M, seed=11)
kw = dict(channels=M, demod="fm", kf=0.3, agc=10.0, max_frames=max(frames))
os.environ["CSDR_AGC_FM_MASK"] = "1"
a = cs.Chain(**kw)
del os.environ["CSDR_AGC_FM_MASK"]
b = cs.Chain(**kw)
print(a.path, "|", b.path)
pos = 0
for f in frames:
    xa = x[pos * M:(pos + f) * M]; pos += f
    ga = a.process(xa); gb = b.process(xa)
    za, zb = ga == 0, gb == 0
    d = np.abs(ga.astype(np.float64) - gb); d = np.minimum(d, np.abs(d - 1 / 0.3))
    print(f, "zeros", int(za.sum()), int(zb.sum()), "mask mismatches", int((za != zb).sum()), "max |diff|", float(d.max()), "median", float(np.median(d[~zb])) if (~zb).any() else None, "agc stats", a.agc_stats(), b.agc_stats())
