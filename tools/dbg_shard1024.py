import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import composable_sdr_amd as cs
from synth import synth_cf32
M, G, kf = 1024, 8, 0.3
for frames in ([40], [96], [96, 40], [40, 96], [96, 96], [7, 96]):
    x = synth_cf32(M * sum(frames), M, seed=3)
    for demod in ("fm", "none"):
        full = cs.Chain(channels=M, demod=demod, kf=kf, max_frames=max(frames))
        wf, pos = [], 0
        for f in frames:
            wf.append(full.process(x[pos * M:(pos + f) * M])); pos += f
        wf = np.concatenate(wf, axis=1); full.close()
        for g in (0, 3):
            ch = cs.Chain(channels=M, demod=demod, kf=kf, chan_first=g, chan_stride=G, max_frames=max(frames))
            got, pos, names = [], 0, []
            for f in frames:
                got.append(ch.process(x[pos * M:(pos + f) * M])); pos += f; names.append(ch.kernel_time()[0])
            got = np.concatenate(got, axis=1); ch.close()
            want = wf[g::G]
            if demod == "fm":
                d = np.abs(got.astype(np.float64) - want); d = np.minimum(d, np.abs(d - 1 / kf))
            else:
                d = np.abs(got - want) / np.abs(wf).max()
            per_call, pos = [], 0
            for f in frames:
                per_call.append(float(np.median(d[:, pos:pos + f]))); pos += f
            print(frames, demod, "g", g, names, "median err per call", ["%.1e" % v for v in per_call], "first cols", ["%.1e" % v for v in np.median(d, axis=0)[:3]])
