import os as _os; _os.environ.setdefault("CSDR_DIAG", "1")
import os, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
path = "/tmp/s1_trace.bin"; os.environ["CSDR_SHARD1024_TRACE"] = path
import numpy as np, torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf, G = 1024, 65536, 8
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=1)
out = torch.empty(M * nf * 2 // G, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod="fm", max_frames=nf, flags=_lib.FLAG_QUIET, chan_first=0, chan_stride=G)
for i in range(30): ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
torch.cuda.synchronize()
raw = np.fromfile(path, dtype=np.uint64).astype(np.int64)
A = raw[:384].reshape(96, 4); B = raw[384:].reshape(96, 4)
rows = []
for s in range(12, 60):
    a, b = A[s], B[s]
    rows.append((a[1]-a[0], a[2]-a[1], a[3]-a[2], A[s+1,0]-a[3], b[1]-b[0], b[2]-b[1], b[3]-b[2], B[s+1,0]-b[3]))
r = np.median(np.array(rows), axis=0).astype(int)
print("role 0: scan %d  tail %d  [barQ+Qwork] %d  toNextP %d | role 1: dma2+partials+radix2 %d  stages32,16 %d  stages8..1+Ywrite %d  [barQ+Qwork+wait] %d  step %d" % (*r, int(np.median([A[s+1,0]-A[s,0] for s in range(12,60)]))))
