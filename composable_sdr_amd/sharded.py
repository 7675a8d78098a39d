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

  mode="hybrid"   SURVEY 8e(B), for the configurations the two above cannot scale: per-channel AGC / squelch tails (state
                  depends on unbounded history: no time stripes) behind a front end that channel sharding cannot divide (DC
                  blocker, pre-mix and FIR see every branch).  Every rank runs the LINEAR front end (DC blocker -> pre-mix ->
                  firpfbch, DeNo) on its own time stripe behind the warm-up prefix of mode="time", ONE all-to-all turns the
                  [M][stripe] planes into [M / G][whole span] planes (rank g receives its channel block [g M / G, (g + 1) M / G) of
                  every stripe: 7/8 of 8 bytes per sample cross xGMI at G = 8), and the rank's tail handle (CSDR_FLAG_TAIL_ONLY:
                  automaticGainControl [-> fmDemodulator] on M / G rows) walks the whole time span.  Every stage divides by G.
                  Front end to the stripe tolerance of mode="time"; the tail is exact given its input.

Outputs stay on the rank that made them (per-channel files are written per rank); gather()
collects them on rank 0 for tests.
"""
import ctypes as C
from dataclasses import replace

import numpy as np

from . import _lib
from ._lib import check, lib
from .pipes import Chain, ChainConfig, _Handle


class _stdout_to_stderr:
    """RCCL prints a version banner on the process's stdout when a communicator is created; a caller whose stdout is a protocol
    (bench.py: ONE JSON line) gets it on stderr instead.  File-descriptor level: the print comes from native code."""

    def __enter__(self):
        import os
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        import os
        try:
            C.CDLL(None).fflush(None)          # the banner sits in C stdio's buffer (stdout is a pipe or a file): out with it while fd 1 is stderr
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class Comm:
    """`csdr_comm` (include/csdr.h): the collectives UNDER the C ABI -- RCCL over xGMI, one process per GPU.  What a non-Python host
    (the Haskell program, host/soapy_sdr_file.cpp) calls; ShardedChain uses it instead of torch.distributed whenever the ranks are
    on GPUs.  Bootstrapping: rank 0 makes `unique_id()`, every rank creates the communicator with those bytes (collective)."""

    def __init__(self, rank, world, unique_id, device=-1):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique_id: %d bytes" % _lib.COMM_ID_BYTES)
        h = C.c_void_p()
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        with _stdout_to_stderr():
            check(lib().csdr_comm_create(rank, world, buf, device, C.byref(h)))
        self._h = _Handle(h, lib().csdr_comm_destroy)
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        check(lib().csdr_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_process_group(cls, dist, group=None, device=-1):
        """One csdr_comm per rank of an initialised torch.distributed group: rank 0's id travels over the group's own store."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0], device)

    @property
    def h(self):
        return self._h.h

    def allreduce_f32(self, d_ptr, count, stream=0):
        check(lib().csdr_comm_allreduce_f32(self.h, C.c_void_p(d_ptr), count, C.c_void_p(stream)))

    def broadcast(self, d_ptr, nbytes, root=0, stream=0):
        check(lib().csdr_comm_broadcast(self.h, C.c_void_p(d_ptr), nbytes, root, C.c_void_p(stream)))

    def hybrid_exchange(self, d_plane, d_recv, chan_per_rank, stripe_frames, elem_bytes=8, stream=0):
        fr = (C.c_uint32 * len(stripe_frames))(*stripe_frames)
        check(lib().csdr_hybrid_exchange(self.h, C.c_void_p(d_plane), C.c_void_p(d_recv), chan_per_rank, fr, elem_bytes, C.c_void_p(stream)))

    def close(self):
        self._h.close()

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
    def __init__(self, cfg: ChainConfig, mode="time", rank=None, world=None, group=None, chain_factory=Chain, interleave=False,
                 tail_factory=None, comm=None):
        """comm: a `Comm` (csdr_comm, the C ABI's collectives) to use for the data-path exchanges.  None: when the process group's
        backend is RCCL ("nccl") one is made from the group (`Comm.from_process_group`), so that on GPUs every collective of the
        data path goes through the C entry points a non-Python host would call; gloo groups (the CPU tests, several ranks on one
        GPU) keep torch.distributed."""
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = rank if rank is not None else (self.dist.get_rank(group) if self.dist else 0)
        self.world = world if world is not None else (self.dist.get_world_size(group) if self.dist else 1)
        self.group, self.mode, self.cfg = group, mode, cfg
        self.comm = comm
        if comm is None and self.dist is not None and self.world > 1 and self.dist.get_backend(group) == "nccl" and chain_factory is Chain:
            self.comm = Comm.from_process_group(self.dist, group, cfg.device)
        self._s_comm = None                          # second stream of the overlapped hybrid step
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
        elif mode == "hybrid":
            if cfg.channels % self.world:
                raise ValueError("hybrid sharding needs world | channels")
            if cfg.agc == 0.0:
                raise ValueError("hybrid sharding is for the AGC configurations; without the AGC use mode='time'")
            if cfg.demod not in ("none", "fm"):
                raise ValueError("hybrid sharding: demod none / fm")
            if cfg.mix:
                # a rank's tail would return the partial mix of ITS channel block only and nothing here reduces those:
                # `--mix` shards by channel (mode="channel": local left fold + one all-reduce)
                raise ValueError("hybrid sharding does not reduce --mix across ranks; use mode='channel' with mix")
            self.interleave = False
            self.cn = cfg.channels // self.world
            self.c0 = self.rank * self.cn                     # contiguous channel blocks: the front end's plane is already grouped by destination rank
            # linear front end on the time stripe (all channels, DeNo, no AGC); per-channel tail on the owned rows
            self.chain = chain_factory(replace(cfg, agc=0.0, demod="none", mix=False, chan_first=0, chan_count=0, chan_stride=0))
            self.tail = (tail_factory or chain_factory)(replace(cfg, channels=self.cn, chan_first=0, chan_count=0, chan_stride=0, tail_only=True,
                                                                dc_block=False))
        else:
            raise ValueError(mode)

    # ------------------------------------------------------------------ hybrid: time-sharded front end, all-to-all, channel-sharded tail
    def hybrid_front(self, x):
        """x: the WHOLE stream (host array).  This rank's front-end output on its stripe: ([M][t1 - t0] complex64, stripe bounds of every rank)."""
        M = self.cfg.channels
        x = np.ascontiguousarray(x, dtype=np.complex64)
        nf = x.size // M
        bounds = [stripe_bounds(nf, self.world, r) for r in range(self.world)]
        t0, t1 = bounds[self.rank]
        w0 = max(0, t0 - warmup_frames(M))
        w0 -= w0 & 1                                  # even start: same pre-mix sign as the full stream
        if hasattr(self.chain, "seek_frames"):
            self.chain.seek_frames(w0)
        front = self._run(self.chain, x, w0, t1)[:, t0 - w0:] if t1 > t0 else np.empty((M, 0), dtype=np.complex64)
        return np.ascontiguousarray(front), bounds

    def hybrid_exchange(self, plane, bounds):
        """plane: this rank's front-end output [M][n][2] (torch, CF32 as float32 pairs).  Rank p owns the channel block
        [p M / G, (p + 1) M / G), so the plane IS [G][M / G][n][2] grouped by destination: one all-to-all (equal stripes:
        all_to_all_single; else send / receive pairs) returns the list of this rank's rows of every stripe, in time order."""
        import torch
        G, g = self.world, self.rank
        mine = [plane[p * self.cn:(p + 1) * self.cn] for p in range(G)]
        if self.dist is None or G == 1:
            return [mine[g]]
        if all(b1 - b0 == bounds[0][1] - bounds[0][0] for (b0, b1) in bounds):
            out = torch.empty_like(plane)
            self.dist.all_to_all_single(out, plane.contiguous(), group=self.group)
            return list(out.view(G, self.cn, -1, 2))
        recv = [torch.empty((self.cn, b1 - b0, 2), dtype=plane.dtype, device=plane.device) for (b0, b1) in bounds]
        recv[g] = mine[g]
        ops = []
        for p in range(G):
            if p == g:
                continue
            if mine[p].numel():
                ops.append(self.dist.P2POp(self.dist.isend, mine[p].contiguous(), p, self.group))
            if recv[p].numel():
                ops.append(self.dist.P2POp(self.dist.irecv, recv[p], p, self.group))
        for w in (self.dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return recv

    def hybrid_tail(self, pieces):
        """pieces: this rank's rows [M / G][n_k] (complex64) of consecutive stretches of the stream, in time order.  The tail handle
        is streaming-stateful, so the stretches go through it one after the other; returns [M / G][sum n_k] (or [sum n_k] mixed)."""
        step = self.cfg.max_frames
        parts = []
        for rows in pieces:
            n = rows.shape[1]
            for a in range(0, n, step):
                parts.append(self.tail.process(np.ascontiguousarray(rows[:, a:min(n, a + step)]).reshape(-1)))
        if not parts:
            return self.tail.process(np.empty(0, dtype=np.complex64))
        return np.concatenate(parts, axis=-1)

    def process_stream_hybrid(self, x):
        """x: the WHOLE stream (host array).  Returns this rank's output [M / G][nf]: the channel block [rank M / G, (rank + 1) M / G)."""
        import torch
        M = self.cfg.channels
        front, bounds = self.hybrid_front(x)
        plane = torch.from_numpy(front.view(np.float32).reshape(M, front.shape[1], 2))
        if self.dist is not None and self.world > 1 and self.dist.get_backend(self.group) == "nccl":
            plane = plane.to(torch.device("cuda", torch.cuda.current_device()))
        pieces = [np.ascontiguousarray(t.cpu().numpy()).view(np.complex64).reshape(self.cn, -1) for t in self.hybrid_exchange(plane, bounds)]
        return self.hybrid_tail(pieces)

    def _exchange_device(self, plane_dev, recv_dev, nfs, stream):
        """All-to-all of one sub-stripe's plane ([M][nfs][2] float32 -> [G][M / G][nfs][2]) on `stream` (an int handle).  RCCL through
        the C ABI (csdr_hybrid_exchange) when the ranks have a csdr_comm; otherwise torch.distributed (gloo: through host memory)."""
        import torch
        G = self.world
        if G == 1 and self.comm is None:
            recv_dev.copy_(plane_dev)
            return
        if self.comm is not None:
            self.comm.hybrid_exchange(plane_dev.data_ptr(), recv_dev.data_ptr(), self.cn, [nfs] * G, 8, stream)
            return
        ext = torch.cuda.ExternalStream(int(stream), device=plane_dev.device) if (plane_dev.is_cuda and stream) else None
        with (torch.cuda.stream(ext) if ext is not None else _nullctx()):
            if plane_dev.is_cuda and self.dist.get_backend(self.group) != "nccl":
                # (test harness: several ranks on one GPU over gloo -- the exchange goes through host memory)
                (ext or torch.cuda.current_stream(plane_dev.device)).synchronize()
                hp = plane_dev.cpu(); hr = torch.empty_like(hp)
                self.dist.all_to_all_single(hr, hp, group=self.group)
                recv_dev.copy_(hr)
            else:
                self.dist.all_to_all_single(recv_dev, plane_dev, group=self.group)

    def process_device_hybrid(self, x_dev, plane_dev, recv_dev, out_dev, stream=0, substripes=1, overlap=False):
        """GPU-resident hybrid step (bench): x_dev = this rank's stretch of the step (interleaved CF32 as float32, nf frames, state carried
        from the rank's previous stretch), plane_dev / recv_dev = [M][nf][2] float32 scratch, out_dev = the output elements.

        The step's time order is S = `substripes` rounds of G sub-stripes: round i holds sub-stripe i of rank 0, 1, ..., G - 1
        (S = 1: one stripe per rank, round 4's layout).  Per round: front end on the rank's sub-stripe -> ONE all-to-all of that plane
        (RCCL over xGMI: csdr_hybrid_exchange) -> the tail walks the round's G stretches in time order (it is streaming-stateful).
        out_dev = [S][G][M / G][nf / S] elements.

        overlap=True (needs S >= 2 to matter): the exchanges run on a second stream, so that round i's plane crosses the links while
        the front end of round i + 1 runs, and round i + 1's exchange runs under round i's tail; the tail calls stay in time order on
        `stream`.  Same kernels, same order per handle: bit-identical to overlap=False."""
        import torch
        G, M, S = self.world, self.cfg.channels, int(substripes)
        nf = x_dev.numel() // 2 // M
        if nf % S or (nf // S) % 32:
            raise ValueError("hybrid step: %d frames do not split into %d sub-stripes of whole 32-frame lines" % (nf, S))
        nfs = nf // S
        w = 1 if self.cfg.demod == "fm" else 2
        dev = plane_dev.device
        on_gpu = plane_dev.is_cuda
        s_main = int(stream or 0)
        if on_gpu and not s_main:
            s_main = int(torch.cuda.current_stream(dev).cuda_stream)
        if overlap and on_gpu:
            if self._s_comm is None:
                self._s_comm = torch.cuda.Stream(device=dev)
            s_x = int(self._s_comm.cuda_stream)
        else:
            s_x = s_main
        main = torch.cuda.ExternalStream(s_main, device=dev) if (on_gpu and s_main) else (torch.cuda.current_stream(dev) if on_gpu else None)
        xch = torch.cuda.ExternalStream(s_x, device=dev) if (on_gpu and s_x) else main
        xf = x_dev.view(-1)
        pf, rf, of = plane_dev.view(-1), recv_dev.view(-1), out_dev.view(-1)
        done = []
        for i in range(S):
            pl = pf[i * M * nfs * 2:(i + 1) * M * nfs * 2]
            rv = rf[i * M * nfs * 2:(i + 1) * M * nfs * 2]
            self.chain.process_device(xf[i * M * nfs * 2:].data_ptr(), M * nfs, pl.data_ptr(), s_main)
            if on_gpu and s_x != s_main:
                ev = torch.cuda.Event(); ev.record(main); xch.wait_event(ev)
            self._exchange_device(pl, rv, nfs, s_x)
            if on_gpu and s_x != s_main:
                ev = torch.cuda.Event(); ev.record(xch); done.append(ev)
            else:
                done.append(None)
            if not overlap:
                self._hybrid_tail_round(rv, of, i, nfs, w, s_main, main, done[i])
        if overlap:
            for i in range(S):
                rv = rf[i * M * nfs * 2:(i + 1) * M * nfs * 2]
                self._hybrid_tail_round(rv, of, i, nfs, w, s_main, main, done[i])
        return S * G * self.cn * nfs

    def _hybrid_tail_round(self, rv, of, i, nfs, w, s_main, main, ev):
        G = self.world
        if ev is not None:
            main.wait_event(ev)
        per = self.cn * nfs * 2                                        # float32 elements of one stretch of the plane
        for k in range(G):
            o = (i * G + k) * self.cn * nfs * w
            self.tail.process_device(rv[per * k:].data_ptr(), self.cn * nfs, of[o:].data_ptr(), s_main)

    # ------------------------------------------------------------------ time stripes
    def process_stream(self, x):
        """x: the WHOLE stream (host array of nf*M CF32).  Returns this rank's output:
        time mode -> [C][t1-t0] (or [t1-t0] mixed) for its stripe; channel mode -> [cn][nf]."""
        if self.mode == "hybrid":
            return self.process_stream_hybrid(x)
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
        if self.comm is not None:
            # the C ABI's entry point: chain + ncclAllReduce on the caller's stream (what a non-Python host calls)
            n_out = C.c_uint32()
            check(lib().csdr_chain_process_device_mix(self.chain.h, self.comm.h, C.c_void_p(x_dev.data_ptr()), x_dev.numel() // 2,
                                                      C.c_void_p(out_dev.data_ptr()), C.byref(n_out), C.c_void_p(int(stream or 0))))
            return n_out.value
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


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
