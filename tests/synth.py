"""Synthetic CF32 input of SURVEY.md section 8(d): DC offset + FM carriers on every
4th channel centre + AWGN.  numpy version (tests, fixtures, CPU baseline) and a
torch version that produces the same kind of signal on the GPU for bench.py."""
import numpy as np


def channel_centre(k, M):
    """Centre of output channel k in the input spectrum, rad/sample
    (Liquid.chs:816-818 pre-shift)."""
    return 2.0 * np.pi * k / M - np.pi * (M - 1) / M


def active_channels(M):
    act = [k for k in range(M) if k % 4 == 1]
    return act if act else [0]


def synth_cf32(n, M, seed=20260101, n0=0, dc=(0.01 + 0.01j), sigma=0.05, amp=0.5):
    """Samples n0 .. n0+n-1 of the section-8(d) signal (noise is seeded per call)."""
    rng = np.random.Generator(np.random.PCG64(seed + (n0 % 1000003)))
    t = np.arange(n0, n0 + n, dtype=np.float64)
    act = active_channels(M)
    a = amp / np.sqrt(len(act))
    fm = (1.0 / M) / 64.0                    # cycles/sample
    dev = 0.2 * (1.0 / M) / 2.0              # peak deviation, cycles/sample
    x = np.full(n, dc, dtype=np.complex128)

    def carriers(lo, hi):                    # every sample sums its tones in the same order whatever the blocking
        tb, xb = t[lo:hi], x[lo:hi]
        for i, k in enumerate(act):
            wk = channel_centre(k, M)
            ph = wk * tb + (dev / fm) * np.sin(2 * np.pi * fm * tb + 0.37 * i)
            xb += a * np.exp(1j * ph)
    if n * len(act) <= (1 << 22):
        carriers(0, n)
    else:
        # n x (M / 4) complex exponentials (40 s for 2 M samples of the 1024-channel signal on one core: most of the GPU suite's wall
        # time): blocks of time on a thread pool (numpy's ufuncs release the GIL)
        import os
        from concurrent.futures import ThreadPoolExecutor
        blk = 1 << 15
        with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda lo: carriers(lo, min(n, lo + blk)), range(0, n, blk)))
    x += (sigma / np.sqrt(2)) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def synth_cf32_torch(n, M, device, seed=20260101, dc=(0.01, 0.01), sigma=0.05, amp=0.5, block=1 << 24):
    """Same signal family generated on `device` (noise from torch's generator)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, 2), dtype=torch.float32, device=device)
    act = active_channels(M)
    a = amp / np.sqrt(len(act))
    fm = (1.0 / M) / 64.0
    dev = 0.2 * (1.0 / M) / 2.0
    wk = torch.tensor([channel_centre(k, M) for k in act], dtype=torch.float64, device=device)
    ph0 = torch.tensor([0.37 * i for i in range(len(act))], dtype=torch.float64, device=device)
    # keep the per-block carrier matrix under ~256 MiB
    block = max(4096, min(block, (1 << 25) // max(1, len(act))))
    for s in range(0, n, block):
        e = min(n, s + block)
        t = torch.arange(s, e, dtype=torch.float64, device=device)
        ph = wk[:, None] * t[None, :] + (dev / fm) * torch.sin(2 * np.pi * fm * t[None, :] + ph0[:, None])
        ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
        re = a * torch.cos(ph).sum(0) + dc[0]
        im = a * torch.sin(ph).sum(0) + dc[1]
        nz = torch.randn((e - s, 2), generator=g, device=device, dtype=torch.float32) * (sigma / np.sqrt(2))
        out[s:e, 0] = re + nz[:, 0]
        out[s:e, 1] = im + nz[:, 1]
    return out
