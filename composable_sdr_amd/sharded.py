"""Multi-GPU sharding of the chain: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Two partitions of the reference's `-c M` channelizer:

  mode="time"     every rank runs ALL channels on its own contiguous time stripe of the stream.
                  Everything up to the FFT is linear and time-invariant and freqdem needs one
                  sample of memory, so a stripe only needs a warm-up prefix (13 frames of FIR
                  window + 32768 samples for the DC blocker's (1-alpha)^n tail): no data-path
                  collective, input read once across the node, `--mix` stays local.  Not valid
                  with the AGC on (its state depends on unbounded history) nor for the AM / WBFM
                  tails (peak detector, de-emphasis and decimator memory are not warmed).
  mode="channel"  every rank sees the whole stream and produces channels
                  [chan_first, chan_first+chan_count) -- what the per-channel AGC/squelch/demod
                  tails and the per-channel sinks need.  `--mix` = local left-fold over the owned
                  channels + one all-reduce(SUM) of nf elements per chunk (latency-bound, 16-32 KiB).
                  interleave=True: rank g owns the channels g, g+G, g+2G, ...; the
                  C ABI prunes the M-point DFT of a frame to one length-G fold + one M/G-point DFT, so
                  a rank does 1/G of the DFT, transpose and tail work (DC blocker, pre-mix and FIR
                  still run on every branch: they are the part channel sharding cannot divide).
                  With GPU-resident buffers the all-reduce runs on the output tensor itself
                  (process_device_mix: RCCL, no host hop).

Outputs stay on the rank that made them (per-channel files are written per rank); gather()
collects them on rank 0 for tests.
"""
from dataclasses import replace

import numpy as np

from .pipes import Chain, ChainConfig

WARMUP_SAMPLES = 32768     # DC blocker: (1 - 0.0005)^32768 = 7.6e-8 of the state is left


def warmup_frames(M):
    """Frames a stripe starts early: FIR window (13) + freqdem (1) + the DC blocker's tail, even."""
    w = 14 + -(-WARMUP_SAMPLES // M)
    return w + (w & 1)


def stripe_bounds(nf_total, world, rank):
    """Frames [t0, t1) of rank `rank`: contiguous, balanced, multiples of 16 except the last."""
    per = -(-nf_total // world)
    per = -(-per // 16) * 16
    t0 = min(nf_total, rank * per)
    return t0, min(nf_total, t0 + per)


def channel_bounds(M, world, rank):
    per = -(-M // world)
    c0 = min(M, rank * per)
    return c0, min(M, c0 + per) - c0


class ShardedChain:
    def __init__(self, cfg: ChainConfig, mode="time", rank=None, world=None, group=None, chain_factory=Chain, interleave=False):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = rank if rank is not None else (self.dist.get_rank(group) if self.dist else 0)
        self.world = world if world is not None else (self.dist.get_world_size(group) if self.dist else 1)
        self.group, self.mode, self.cfg = group, mode, cfg
        if mode == "time":
            if cfg.agc != 0.0 and self.world > 1:
                raise ValueError("time stripes cannot carry the AGC state; use mode='channel' with the AGC on")
            if cfg.demod in ("am", "wbfm") and self.world > 1:
                # the warm-up prefix covers the DC blocker, the FIR window and freqdem's one sample; the AM peak
                # detector (0.99^n), the de-emphasis IIR and the decimator phase are not warmed by it
                raise ValueError("time stripes do not warm the %s tail's memory; use mode='channel'" % cfg.demod)
            self.chain = chain_factory(cfg)
        elif mode == "channel":
            self.interleave = bool(interleave) and self.world > 1
            if self.interleave:
                if cfg.channels % self.world:
                    raise ValueError("interleaved channel ownership needs world | channels")
                self.c0, self.cn = self.rank, cfg.channels // self.world
                self.chain = chain_factory(replace(cfg, chan_first=self.rank, chan_count=0, chan_stride=self.world))
            else:
                c0, cn = channel_bounds(cfg.channels, self.world, self.rank)
                self.c0, self.cn = c0, cn
                # the rank's partial mix is reduced across ranks afterwards
                self.chain = chain_factory(replace(cfg, chan_first=c0, chan_count=cn)) if cn else None
        else:
            raise ValueError(mode)

    # ------------------------------------------------------------------ time stripes
    def process_stream(self, x):
        """x: the WHOLE stream (host array of nf*M CF32).  Returns this rank's output:
        time mode -> [C][t1-t0] (or [t1-t0] mixed) for its stripe; channel mode -> [cn][nf]."""
        M = self.cfg.channels
        x = np.ascontiguousarray(x, dtype=np.complex64)
        nf = x.size // M
        if self.mode == "channel":
            return self._run(self.chain, x, 0, nf) if self.chain is not None else None
        t0, t1 = stripe_bounds(nf, self.world, self.rank)
        w0 = max(0, t0 - warmup_frames(M))
        w0 -= w0 & 1                                  # even start: same pre-mix sign as the full stream
        if hasattr(self.chain, "seek_frames"):
            self.chain.seek_frames(w0)
        if t1 <= t0:
            return self._run(self.chain, x, t0, t0)
        out = self._run(self.chain, x, w0, t1)
        return out[..., t0 - w0:]

    def _run(self, chain, x, f0, f1):
        M, step = self.cfg.channels, self.cfg.max_frames
        parts = [chain.process(x[a * M:min(f1, a + step) * M]) for a in range(f0, f1, step)]
        if not parts:
            return chain.process(x[:0])
        return np.concatenate(parts, axis=-1)

    # ------------------------------------------------------------------ collectives
    def mix_allreduce(self, partial):
        """channel mode + mix: sum the per-rank partial mixes (RCCL all-reduce on GPUs)."""
        if self.dist is None or self.world == 1:
            return partial
        import torch
        t = torch.from_numpy(np.ascontiguousarray(partial).view(np.float32).copy())
        dev = torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend(self.group) == "nccl" else None
        if dev is not None:
            t = t.to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().view(partial.dtype).reshape(partial.shape)

    def process_device_mix(self, x_dev, out_dev, stream=0):
        """channel mode + mix with GPU-resident tensors (torch, on this rank's device): the chain writes the rank's partial
        mix straight into `out_dev`, then ONE all-reduce(SUM) on that tensor over RCCL/xGMI -- no host copy.  `x_dev`:
        the chunk (interleaved CF32 as float32), the same on every rank; returns the number of output elements."""
        n = self.chain.process_device(x_dev.data_ptr(), x_dev.numel() // 2, out_dev.data_ptr(), stream)
        if self.dist is not None and self.world > 1:
            if out_dev.is_cuda:
                import torch
                cur = torch.cuda.current_stream(out_dev.device)
                if int(stream or 0) != int(cur.cuda_stream):
                    # the collective runs on torch's current stream: order it behind the chain's launches on `stream`
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.ExternalStream(int(stream or 0), device=out_dev.device))
                    cur.wait_event(ev)
            self.dist.all_reduce(out_dev, op=self.dist.ReduceOp.SUM, group=self.group)
        return n

    def gather(self, local):
        """Collect every rank's output on rank 0 (time mode: concatenated in time; channel mode:
        stacked by channel).  Returns None on other ranks."""
        if self.dist is None or self.world == 1:
            return local
        objs = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(local, objs, dst=0, group=self.group)
        if self.rank != 0:
            return None
        if self.mode == "channel" and getattr(self, "interleave", False):
            full = np.empty((self.cfg.channels,) + objs[0].shape[1:], dtype=objs[0].dtype)
            for g, o in enumerate(objs):
                full[g::self.world] = o                  # row m of rank g is channel g + G*m
            return full
        objs = [o for o in objs if o is not None and o.size]
        return np.concatenate(objs, axis=-1 if self.mode == "time" else 0)
