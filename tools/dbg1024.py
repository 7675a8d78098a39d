import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
demod = sys.argv[1]; nf = int(sys.argv[2]); M = 1024
dev = torch.device("cuda", 0)
x = synth_cf32_torch(M * nf, M, dev, seed=5)
out = torch.zeros(M * nf * 2, dtype=torch.float32, device=dev)
ch = cs.Chain(channels=M, demod=demod, max_frames=nf, flags=_lib.FLAG_QUIET)
for i in range(3):
    ch.process_device(x.data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    print("call", i, "ok", ch.kernel_time()[0], flush=True)
os.environ["CSDR_RUN1024_V1"] = "1"
ch2 = cs.Chain(channels=M, demod=demod, max_frames=nf, flags=_lib.FLAG_QUIET)
out2 = torch.zeros_like(out)
for i in range(3):
    ch2.process_device(x.data_ptr(), M * nf, out2.data_ptr(), 0)
torch.cuda.synchronize()
w = 1 if demod == "fm" else 2
a = out[:M * nf * w].view(M, nf, w); b = out2[:M * nf * w].view(M, nf, w)
d = (a - b).abs()
print("max diff vs v1", float(d.max()), "mean", float(d.mean()), "bad rows", int((d.amax(dim=(1, 2)) > 1e-2).sum()))
