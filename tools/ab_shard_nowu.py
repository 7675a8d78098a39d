import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
M, nf = 1024, 65536
dev = torch.device("cuda", 0)
xs = [synth_cf32_torch(M * nf, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
out = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
for g in (0, 3):
    ch = cs.Chain(channels=M, demod="fm", kf=0.3, max_frames=nf, flags=_lib.FLAG_QUIET, chan_first=g, chan_stride=8)
    for rep in range(2):
        for i in range(50): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400
        print(f"NOWU={os.environ.get('CSDR_NOWU','default')} g={g}: {dt*1e6:7.1f} us  {ch.kernel_time()[0]}", flush=True)
    ch.close()
