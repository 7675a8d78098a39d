"""N > 1 path on CPU: world_size-2 gloo processes drive composable_sdr_amd.sharded.ShardedChain
(partitioning, warm-up prefix, gather, mix all-reduce).  No GPU here, so the per-rank chain is a
stand-in built on the CPU oracle (test infrastructure) with the product Chain's interface; on
the GPU box test_gpu_parity.py::test_sharded_* runs the same logic on the HIP chain."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class OracleChain:
    """composable_sdr_amd.Chain's process() surface on top of oracle_lib (CPU stand-in)."""

    def __init__(self, cfg):
        import oracle_lib as O
        self.cfg = cfg
        self.full = O.Chain(cfg.channels, dc_block=cfg.dc_block, agc_db=cfg.agc, demod=cfg.demod, kf=cfg.kf, mix=False)
        self.c0 = cfg.chan_first
        self.cn = cfg.chan_count or cfg.channels - cfg.chan_first
        self.G = cfg.chan_stride if cfg.chan_stride > 1 else 1
        self.O = O

    def process(self, x):
        y = self.full.process(x)
        y = y[self.c0::self.G] if self.G > 1 else y[self.c0:self.c0 + self.cn]
        if self.cfg.mix and self.cfg.channels > 1:
            return self.O.mix_f32(y) if y.dtype == np.float32 else y.sum(axis=0).astype(y.dtype)
        return y


class OracleTail:
    """the CSDR_FLAG_TAIL_ONLY handle's process() surface on top of oracle_lib: one Agc [+ FreqDem] per row, streaming-stateful"""

    def __init__(self, cfg):
        import oracle_lib as O
        assert cfg.tail_only and cfg.agc != 0.0
        self.C, self.fm = cfg.channels, cfg.demod == "fm"
        self.agc = [O.Agc(cfg.agc) for _ in range(self.C)]
        self.dem = [O.FreqDem(cfg.kf) for _ in range(self.C)] if self.fm else None

    def process(self, plane):
        z = np.ascontiguousarray(plane, dtype=np.complex64).reshape(self.C, -1)
        y = [self.agc[k].execute_block(z[k]) for k in range(self.C)]
        if self.fm:
            y = [self.dem[k].demodulate_block(y[k]) for k in range(self.C)]
        return np.stack(y) if z.shape[1] else np.empty((self.C, 0), dtype=np.float32 if self.fm else np.complex64)


def _worker_hybrid(rank, world, port, q, demod, nf):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        from composable_sdr_amd.pipes import ChainConfig
        from composable_sdr_amd.sharded import ShardedChain
        from synth import synth_cf32
        M = 16
        x = synth_cf32(M * nf, M, seed=43)
        cfg = ChainConfig(channels=M, demod=demod, kf=0.3, agc=3.0, max_frames=1024)
        sc = ShardedChain(cfg, mode="hybrid", chain_factory=OracleChain, tail_factory=OracleTail)
        full = sc.gather(sc.process_stream(x))
        if rank == 0:
            q.put((full, O.Chain(M, demod=demod, kf=0.3, agc_db=3.0).process(x)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("demod,nf", [("fm", 6000), ("none", 6016)])
def test_hybrid_time_front_alltoall_channel_tail_world2(demod, nf):
    """linear front end on time stripes, one all-to-all (unequal stripes: send / receive pairs; equal: all_to_all_single),
    AGC + squelch [+ freqdem] tail on channel blocks over the whole span -- against the single-stream oracle: same mute decisions
    (the threshold sits >= 3 dB from every channel level), open samples to the stripe tolerance of the time partition."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hybrid, args=(r, 2, port, q, demod, nf)) for r in range(2)]
    for p in procs:
        p.start()
    full, want = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full.shape == want.shape
    mism = int(np.sum((full == 0) != (want == 0)))
    op = want != 0
    assert mism == 0 and 0.05 < op.mean() < 0.95, (mism, op.mean())
    if demod == "fm":
        d = np.abs((full.astype(np.float64) - want + 0.5 / 0.3) % (1 / 0.3) - 0.5 / 0.3)
        assert np.median(d[op]) < 1e-5 and np.quantile(d, 0.999) < 5e-4
    else:
        assert np.max(np.abs(full - want)) < 1e-4 * np.abs(want).max()
    # rank 0's stripe is the head of the stream and the tail is exact given its input: bit-identical there
    assert np.array_equal(full[:, :3008], want[:, :3008])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, mix, q, demod="fm", interleave=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as O
        from composable_sdr_amd.pipes import ChainConfig
        from composable_sdr_amd.sharded import ShardedChain
        from synth import synth_cf32
        M, nf = 16, 6000
        x = synth_cf32(M * nf, M, seed=42)
        cfg = ChainConfig(channels=M, demod=demod, kf=0.3, mix=mix, max_frames=1024)
        sc = ShardedChain(cfg, mode=mode, chain_factory=OracleChain, interleave=interleave)
        local = sc.process_stream(x)
        if mode == "channel" and mix:
            local = sc.mix_allreduce(local)
            full = local if rank == 0 else None
        else:
            full = sc.gather(local)
        if rank == 0:
            want = O.Chain(M, demod=demod, kf=0.3, mix=mix).process(x)
            q.put((full, want))
    finally:
        dist.destroy_process_group()


def _run(mode, mix, demod="fm", interleave=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, mix, q, demod, interleave)) for r in range(2)]
    for p in procs:
        p.start()
    full, want = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return full, want


def test_time_stripes_world2_match_single_stream():
    full, want = _run("time", False, "none")
    assert full.shape == want.shape
    # rank 1 starts 2062 frames (32768 samples + FIR window) early from a zero state: the DC blocker / FIR transient is gone
    err = np.abs(full - want)
    assert err.max() < 2e-5 * np.abs(want).max()
    # rank 0's stripe is the head of the stream: bit-identical
    assert np.array_equal(full[:, :3008], want[:, :3008])
    full, want = _run("time", False, "fm")
    d = np.abs((full.astype(np.float64) - want + 0.5 / 0.3) % (1 / 0.3) - 0.5 / 0.3)
    tone = np.arange(16) % 4 == 1
    assert np.median(d) < 1e-6 and d[tone].max() < 1e-4
    assert np.array_equal(full[:, :3008], want[:, :3008])


def test_channel_shards_world2_are_exact_slices():
    full, want = _run("channel", False)
    assert np.array_equal(full, want)


def test_channel_shards_mix_allreduce_world2():
    full, want = _run("channel", True)
    assert full.shape == want.shape
    # per-rank left folds + one SUM: same terms, different association
    assert np.max(np.abs(full - want)) < 1e-4


def test_interleaved_channel_shards_world2():
    """rank g owns channels g, g + 2, ...: gather puts the rows back in channel order; the mixed variant reduces the two
    partial left folds with one all-reduce"""
    full, want = _run("channel", False, "fm", interleave=True)
    assert np.array_equal(full, want)
    full, want = _run("channel", True, "none", interleave=True)
    assert full.shape == want.shape
    assert np.max(np.abs(full - want)) < 1e-4 * max(1.0, np.abs(want).max())


def test_bounds_helpers():
    from composable_sdr_amd.sharded import channel_bounds, stripe_bounds
    assert [stripe_bounds(600, 2, r) for r in range(2)] == [(0, 304), (304, 600)]
    from composable_sdr_amd.sharded import warmup_frames
    assert warmup_frames(256) == 142 and warmup_frames(16) == 2062 and warmup_frames(4096) == 22
    assert [stripe_bounds(10, 4, r) for r in range(4)] == [(0, 10), (10, 10), (10, 10), (10, 10)]
    assert [channel_bounds(256, 8, r) for r in range(8)] == [(32 * r, 32) for r in range(8)]
    assert [channel_bounds(20, 8, r) for r in range(8)] == [(0, 3), (3, 3), (6, 3), (9, 3), (12, 3), (15, 3), (18, 2), (20, 0)]
