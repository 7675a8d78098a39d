import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from synth import synth_cf32_torch
dev = torch.device("cuda", 0)
for M, nf, agc in ((64, 1048576, 10.0), (4096, 16384, 23.0), (4096, 4096, 23.0), (1024, 65536, 10.0), (256, 262144, 10.0)):
    xs = [synth_cf32_torch(M * nf, min(M, 256) if M > 1024 else M, dev, seed=20260101 + 7919 * i) for i in range(2)]
    out = torch.empty(M * nf, dtype=torch.float32, device=dev)
    ch = cs.Chain(channels=M, demod="fm", kf=0.3, agc=agc, max_frames=nf, flags=_lib.FLAG_QUIET)
    for i in range(8): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    n = max(20, int(0.4 / 0.0006))
    t0 = time.perf_counter()
    for i in range(n): ch.process_device(xs[i & 1].data_ptr(), M * nf, out.data_ptr(), 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"M={M} nf={nf} FM + AGC: {dt*1e6:8.1f} us per call  tile-major calls {ch.agc_tile_major_calls()}  [{ch.path}]", flush=True)
    ch.close()
