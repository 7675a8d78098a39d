#!/usr/bin/env python3
"""Headline benchmark: MS/s of CF32 input through the 256-channel PFB + FM chain on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic CF32 that is already
resident in HBM: dcBlocker -> NCO pre-mix -> firpfbch(256, m=7, 80 dB) -> per-channel
freqdem(kf) -> channel-major F32 [256][nf]  (BASELINE.json configs[2] with `-a 0`, the
reference's own "no AGC" setting, apps/SoapySDR.hs:195-198; the AGC-enabled variant is
measured beside it and reported under "agc_variant" -- see DESIGN.md section 6).

Multi-GPU (--gpus N): one process per GPU; every rank runs the chain on its own time
stripe of the stream (no data-path collective: FM/DeNo outputs of a stripe depend only
on that stripe plus a 13-frame halo), so scaling is weak and value = total samples / max time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured float4-copy ceiling (SURVEY 8d asks for both fractions)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--frames", type=int, default=262144,
                    help="frames per step (64 reference chunks of 4096 frames: 512 MiB of CF32 at M=256)")
    ap.add_argument("--demod", default="fm", choices=["fm", "none"])
    ap.add_argument("--kf", type=float, default=0.3)
    ap.add_argument("--agc", type=float, default=0.0, help="squelch threshold dB (0 = AGC off)")
    ap.add_argument("--mix", action="store_true", help="--mix: sum the channels (configs[4] shape)")
    ap.add_argument("--shard", default="time", choices=["time", "channel"],
                    help="multi-GPU partition: time stripes (weak scaling, no collective; default) or interleaved channel "
                         "ownership k = rank (mod N) with a pruned DFT per rank (strong scaling; --mix adds one RCCL all-reduce per step)")
    ap.add_argument("--chan-stride", type=int, default=0,
                    help="N = 1 only: run ONE rank's handle of an N = chan-stride channel-shard run (chan_first = --chan-first, default 0) alone on "
                         "this GPU -- the per-rank kernel of BASELINE configs[3] for --channels 1024 --chan-stride 8; `value` is then the rate at which "
                         "that rank gets through the common input stream (what profiles/rNN_shard_* are collected with)")
    ap.add_argument("--chan-first", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-agc-variant", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short side measurements of the other BASELINE shapes (\"other_configs\"; the default N = 1 run only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--preheat-ms", type=float, default=2500.0,
                    help="back-to-back steps in front of the W warm-up steps, for at least this long, with one synchronisation at the end: "
                         "reported as `sustained_long` (with the board's sclk / socket power sampled from rocm-smi meanwhile) and at the same "
                         "time what brings the board into the power state a long stream sees (the cap's averaging window is hundreds of ms: "
                         "profiles/r03_short_run_preheat.txt), so that the K timed steps behind it measure THAT state; 0 = off "
                         "(the K-step figure from an idle board is always reported as `cold_window`)")
    return ap.parse_args()


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo; None when it cannot be told"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except Exception:
        return None


def cpu_baseline(M, demod, kf, agc, x_host, seconds, mix=False):
    """The oracle (CPU restatement, 1 thread like the reference's non-threaded RTS) timed on a
    bounded sample of the same workload: reference-sized chunks (4096 frames) of the same
    synthetic stream, repeated until ~`seconds` of CPU work."""
    import numpy as np
    import oracle_lib as O
    chain = O.Chain(M, dc_block=True, agc_db=agc, demod=demod, kf=kf, mix=mix)
    chunk = 4096 * M
    nchunks = x_host.size // chunk
    done, t0 = 0, time.perf_counter()
    while True:
        for i in range(nchunks):
            chain.process(x_host[i * chunk:(i + 1) * chunk])
            done += chunk
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    res = {"value": round(done / dt / 1e6, 3), "unit": "MS/s", "cores": 1, "kind": "port",
           "sample": f"{done // chunk} chunks of 4096 frames x {M} ch ({done / 1e6:.1f} MS) of the same synthetic stream, "
                     f"oracle/csdr_oracle.c single thread, {dt:.1f} s"}
    # the same restatement on all host cores -- one PROCESS per core (children of this one; each builds its own
    # chain and runs reference-sized chunks of the same signal family for `budget` seconds of its own clock, after a common start
    # time), the fair-hardware figure; the reference itself is single-threaded
    import subprocess
    ncores = os.cpu_count() or 1
    if ncores > 1 and nchunks >= 1:
        budget = max(2.0, seconds / 3)
        worker = (
            "import sys,time,os;sys.path.insert(0,%r);sys.path.insert(0,%r);import numpy as np,oracle_lib as O;from synth import synth_cf32;"
            "M=%d;Ms=min(M,256);x=np.tile(synth_cf32(4096*Ms,Ms,seed=1000+int(sys.argv[1])),M//Ms) if M%%Ms==0 else synth_cf32(4096*M,M,seed=1000+int(sys.argv[1]));c=O.Chain(M,dc_block=True,agc_db=%r,demod=%r,kf=%r,mix=%r);"
            "t_go=float(sys.argv[2]);n=0\n"
            "while time.time()<t_go: time.sleep(0.01)\n"
            "t0=time.perf_counter()\n"
            "while time.perf_counter()-t0<%r:\n c.process(x);n+=x.size\n"
            "print(n,time.perf_counter()-t0)" % (ROOT, os.path.join(ROOT, "tests"), M, agc, demod, kf, mix, budget))
        t_go = time.time() + 4.0 + 0.02 * ncores                     # every child has imported numpy and built its chain by then
        procs = [subprocess.Popen([sys.executable, "-c", worker, str(k), repr(t_go)], stdout=subprocess.PIPE, text=True,
                                  env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")) for k in range(ncores)]
        tot, tmax, ok = 0, 0.0, 0
        for pr in procs:
            o, _ = pr.communicate()
            try:
                n, t = o.split()
                tot += int(n); tmax = max(tmax, float(t)); ok += 1
            except Exception:
                pass
        if ok:
            res["all_cores"] = {"value": round(tot / tmax / 1e6, 3), "unit": "MS/s", "cores": ok,
                                "sample": f"{ok} processes (one per host core, os.cpu_count() = {ncores}), one independent chain each on its own reference-sized "
                                          f"chunk, started together, {tot / 1e6:.1f} MS in {tmax:.1f} s",
                                "sched_affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                                "physical_cores": physical_cores()}
    return res


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the kernel sources of libcsdr_hip.so (csrc/*.hip, *.h, sorted by name): what a
    profiles/traffic.json entry must have been collected with to be quoted"""
    import hashlib
    d = os.path.join(ROOT, "composable_sdr_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class SmiSampler:
    """sclk / socket power from rocm-smi, sampled by a side thread while a run is in flight (each call is a child process)."""

    def __init__(self, device=0):
        import threading
        self.samples, self.stop, self.device = [], False, device
        self.t0 = time.perf_counter()
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        import subprocess
        while not self.stop:
            t = time.perf_counter() - self.t0
            try:
                o = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", str(self.device), "--showpower", "--showclocks", "--json"],
                                   capture_output=True, text=True, timeout=20).stdout
                for v in json.loads(o).values():
                    sclk = pw = None
                    for k, val in v.items():
                        kl = k.lower()
                        if "sclk clock speed" in kl:
                            sclk = float(str(val).strip("()").lower().replace("mhz", ""))
                        elif "power" in kl and "(w)" in kl:
                            pw = float(val)
                    self.samples.append((t, time.perf_counter() - self.t0, sclk, pw))
            except Exception:
                pass
            time.sleep(0.05)

    def finish(self, t_lo, t_hi):
        """mean over the samples taken entirely inside [t_lo, t_hi] (seconds since the sampler started)"""
        self.stop = True                         # (set by the caller right after the run already: the join below happens later)
        self.th.join(timeout=30)
        ok = [x for x in self.samples if x[0] >= t_lo and x[1] <= t_hi]
        sc = [x[2] for x in ok if x[2] is not None]
        pw = [x[3] for x in ok if x[3] is not None]
        return {"samples": len(ok), "sclk_mhz_mean": round(sum(sc) / len(sc), 1) if sc else None,
                "socket_power_w_mean": round(sum(pw) / len(pw), 1) if pw else None}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ        # under torchrun even N=1 goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libcsdr_hip has no CPU fallback)")
    # test hook (tests/test_gpu_parity.py::test_bench_channel_shard_two_ranks_one_gpu): all ranks on device 0 over gloo,
    # which exercises the N > 1 code path on a one-GPU box; the driver's runs use one GPU per rank over RCCL ("nccl")
    one_gpu = os.environ.get("CSDR_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        dist.init_process_group(backend="gloo" if one_gpu else "nccl", rank=rank, world_size=world)
        if not one_gpu:
            # the first collective creates torch's own RCCL communicator, whose version banner goes to stdout: keep stdout for the JSON line
            from composable_sdr_amd.sharded import _stdout_to_stderr
            with _stdout_to_stderr():
                dist.barrier()
                torch.cuda.synchronize()

    import composable_sdr_amd as cs
    from composable_sdr_amd import _lib
    from synth import synth_cf32_torch

    M, nf = a.channels, a.frames
    nx = M * nf
    out_elem = 4 if a.demod == "fm" else 8
    # two input buffers (> the 256 MiB Infinity Cache together) alternate between steps;
    # rank r's stripe is a different stretch of the stream (different seed offset)
    # --shard channel: north_star's partition.  Also accepted on ONE GPU (a world of one rank): the step then still goes through the
    # C ABI's collective entry points (csdr_comm_* / csdr_chain_process_device_mix), which is how a one-GPU box exercises them
    chan = a.shard == "channel"
    per_rank = a.chan_stride if (a.chan_stride > 1 and world == 1 and not chan) else 0      # one rank's shard handle, alone on this GPU
    if per_rank and (M % per_rank or not 0 <= a.chan_first < per_rank):
        raise SystemExit("--chan-stride must divide --channels and --chan-first must be below it")
    if chan and M % world:
        raise SystemExit(f"--shard channel needs --gpus | --channels ({world} does not divide {M})")
    # time stripes: rank r's stripe is a different stretch of the stream (different seed offset);
    # channel shards: every rank sees the SAME stream (in production: broadcast over xGMI) and owns channels rank, rank + N, ...
    xs = [synth_cf32_torch(nx, M, dev, seed=20260101 + 7919 * (2 * (0 if chan else rank) + i)) for i in range(2)]
    out = torch.empty(M * nf * out_elem // 4, dtype=torch.float32, device=dev)
    # the timed region carries ONE hipEvent in front of its first launch of the dominant kernel and one behind its last
    # (CSDR_FLAG_TIME_REGION: total / launches, on the launch stream); the per-launch duration of the kernel itself comes from a
    # separate pass of K launches with an event pair around every launch (`roofline.launch_ms`): the pairs cost the stream a few us
    # per launch and stay out of `value`
    flags = _lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS | _lib.FLAG_TIME_REGION
    comm = None
    if chan:
        # the data-path collectives run under the C ABI (include/csdr.h csdr_comm: RCCL over xGMI); torch.distributed only carries
        # the bootstrap id, the barriers and the max-over-ranks of the clock.  (gloo test hook: several ranks on one GPU cannot
        # form an RCCL communicator -- there ShardedChain falls back to torch.distributed.)
        # Time stripes (the default) have no data-path collective: their communicator is only created for the side measurements,
        # BEHIND the contract's numbers and under a watchdog (see run_sides), so that nothing a collective does can cost the line.
        from composable_sdr_amd.sharded import Comm
        if not one_gpu:
            comm = Comm.from_process_group(dist, None, local) if use_dist else Comm(0, 1, Comm.unique_id(), local)
    if chan:
        from composable_sdr_amd.pipes import ChainConfig
        from composable_sdr_amd.sharded import ShardedChain
        sc = ShardedChain(ChainConfig(channels=M, demod=a.demod, kf=a.kf, agc=a.agc, mix=a.mix, max_frames=nf, device=local, flags=flags),
                          mode="channel", interleave=True, rank=rank, world=world, comm=comm)
        chain = sc.chain
    else:
        chain = cs.Chain(channels=M, demod=a.demod, kf=a.kf, agc=a.agc, mix=a.mix, max_frames=nf, device=local, flags=flags,
                         chan_first=(a.chan_first if per_rank else 0), chan_stride=per_rank)
        chain.seek_frames(rank * a.steps * nf)      # rank r's stripe of one long stream
    stream = torch.cuda.current_stream().cuda_stream
    xv = [x.view(-1) for x in xs]
    # second handle for the per-launch kernel timing pass behind the timed region (created now: allocations between the two
    # would be an idle gap)
    kch = cs.Chain(channels=M, demod=a.demod, kf=a.kf, agc=a.agc, mix=a.mix, max_frames=nf, device=local, flags=_lib.FLAG_QUIET | _lib.FLAG_TIME_KERNELS,
                   chan_first=(rank if chan and world > 1 else (a.chan_first if per_rank else 0)), chan_stride=(world if chan and world > 1 else per_rank))

    def step(i):
        if chan and a.mix:
            sc.process_device_mix(xv[i & 1], out[: nf * out_elem // 4], stream)    # partial mix + one RCCL all-reduce(SUM) on the tensor
        else:
            chain.process_device(xs[i & 1].data_ptr(), nx, out.data_ptr(), stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, w):
        for i in range(w):
            step(i)
        barrier()
        chain.kernel_time()
        t0 = time.perf_counter()
        for i in range(k):
            step(i)
        barrier()
        d = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d, chain.kernel_time()

    # channel shards work on ONE stream: the chunk has to reach every rank (8 B per input sample over xGMI).  The bench synthesises
    # the identical chunk on every rank, so the broadcast is timed here once, on its own (csdr_comm_broadcast = ncclBroadcast)
    bcast = None
    if chan and comm is not None:
        barrier()
        for r in range(3):
            if r == 1:
                barrier(); tb0 = time.perf_counter()
            comm.broadcast(xs[1].data_ptr(), nx * 8, 0, stream)
        barrier()
        tbc = (time.perf_counter() - tb0) / 2
        bcast = {"ms": round(tbc * 1e3, 4), "bytes": nx * 8, "gb_per_s": round(nx * 8 / tbc / 1e9, 1), "ranks": world,
                 "note": "csdr_comm_broadcast (ncclBroadcast) of one step's input from rank 0, timed alone; not inside `value` "
                         "(every rank synthesises the same chunk)"}
    # (0) the contract's W + K steps on an idle board: reported as `cold_window`, never `value`
    barrier()
    dt_cold, _ = timed(a.steps, a.warmup)
    # (1) >= preheat_ms of back-to-back steps, one synchronisation at the end: `sustained_long`, and the state `value` is measured in
    preheat_steps, sus_long, smi = 0, None, None
    kms_l, klaunches_l = 0.0, 0
    if a.preheat_ms > 0:
        est = max(dt_cold / a.steps, 1e-5)
        preheat_steps = max(10, int(a.preheat_ms * 1e-3 / est) + 1)
        # (under rocprofv3 the preloaded tool has initialised the GPU in this process: the pool refuses a child that execs, which rocm-smi does)
        smi = SmiSampler(local) if rank == 0 and "rocprof" not in os.environ.get("LD_PRELOAD", "") else None
        for attempt in range(4):                # the estimate comes from a K-step window: re-run longer until the run lasts preheat_ms
            if use_dist:                        # every rank the SAME number of steps (a step may contain a collective)
                t = torch.tensor([preheat_steps], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                preheat_steps = int(t.item())
            barrier()
            chain.kernel_time()                 # (reset: the region events below bracket exactly this run's launches)
            t_a = time.perf_counter()
            for i in range(preheat_steps):
                step(i)
            torch.cuda.synchronize()
            t_b = time.perf_counter()
            _, kms_l, klaunches_l = chain.kernel_time()     # ONE hipEvent in front of the run's first launch, one behind its last, on the launch stream
            el = t_b - t_a
            if use_dist:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                el = float(t.item())
            if el >= 0.9 * a.preheat_ms * 1e-3:
                break
            preheat_steps = int(preheat_steps * min(50.0, 1.15 * a.preheat_ms * 1e-3 / max(el, 1e-6))) + 1
        d_l = (t_b - t_a) / preheat_steps
        if smi:
            smi.stop = True                     # no join here: an idle gap of 0.1 s in front of the timed region would put the board back
                                                # into its idle state (first run of this file: 295 us per step behind a 0.3 s join)
        sus_long = {"steps": preheat_steps, "seconds": round(t_b - t_a, 3), "ms_per_step": round(d_l * 1e3, 4),
                    "value": round(nx * (1 if chan else world) / d_l / 1e6, 1), "unit": "MS/s",
                    "launch_ms_events": round(kms_l / klaunches_l, 4) if klaunches_l else None, "launches": klaunches_l}
    # (2) the contract: W warm-up steps, then EXACTLY K timed steps between barrier + synchronize
    dt, (kname_r, kms_r, klaunches_r) = timed(a.steps, a.warmup)
    # (3) the same step over 400 further launches (`sustained`), still back to back
    d_s = None
    if world == 1:
        n_sus = 400
        t_s = time.perf_counter()
        for i in range(n_sus):
            step(i)
        torch.cuda.synchronize()
        d_s = (time.perf_counter() - t_s) / n_sus

    # the dominant kernel's own launch duration: a separate pass of K launches with a hipEvent pair around every launch, on the
    # launch stream, right behind the timed region (same board state); what rocprofv3's average for the kernel must agree with.
    # Every rank runs it (no collective inside), rank 0 reports its own.
    for i in range(max(a.warmup, 2)):
        kch.process_device(xs[i & 1].data_ptr(), nx, out.data_ptr(), stream)
    torch.cuda.synchronize()
    kch.kernel_time()
    for i in range(a.steps):
        kch.process_device(xs[i & 1].data_ptr(), nx, out.data_ptr(), stream)
    torch.cuda.synchronize()
    kname, kms, klaunches = kch.kernel_time()
    kch.close()
    if sus_long is not None and smi is not None:
        # samples from 0.5 s into the long run (the power state has settled) to its end
        sus_long["board"] = smi.finish(t_a - smi.t0 + min(0.5, 0.25 * (t_b - t_a)), t_b - smi.t0)

    # ---- side measurements of the N > 1 run (north_star's channel partition, the hybrid partition): behind the contract's numbers,
    # under a watchdog.  They are the only part of a default run that issues data-path collectives (csdr_comm / RCCL), and no N > 1
    # hardware was available to try them on: if one of them raises or does not come back within CSDR_BENCH_SIDE_TIMEOUT seconds, the
    # line is printed without them ("side_error") and every rank leaves with status 0.
    chan2, hyb = None, None
    side_state = {"error": None}
    side_done = threading.Event()

    def run_sides():
        nonlocal comm, chan2, hyb
        if world > 1 and not chan and not one_gpu and comm is None:
            from composable_sdr_amd.sharded import Comm
            comm = Comm.from_process_group(dist, None, local)
        # N > 1, time stripes (the default): north_star's own partition -- interleaved channel ownership, every rank on the SAME
        # stream -- measured beside it in the same run and reported under "channel_shard" (strong scaling: the samples are counted once)
        if world > 1 and not chan and M % world == 0:
            from composable_sdr_amd.pipes import ChainConfig
            from composable_sdr_amd.sharded import ShardedChain
            xc = [synth_cf32_torch(nx, M, dev, seed=20260101 + 7919 * i) for i in range(2)]
            sc2 = ShardedChain(ChainConfig(channels=M, demod=a.demod, kf=a.kf, agc=a.agc, mix=a.mix, max_frames=nf, device=local,
                                           flags=_lib.FLAG_QUIET), mode="channel", interleave=True, comm=comm)
            xcv = [x.view(-1) for x in xc]

            def step2(i):
                if a.mix:
                    sc2.process_device_mix(xcv[i & 1], out[: nf * out_elem // 4], stream)
                else:
                    sc2.chain.process_device(xc[i & 1].data_ptr(), nx, out.data_ptr(), stream)
            reps2 = max(3, a.steps // 2)
            for i in range(2):
                step2(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(reps2):
                step2(i)
            barrier()
            d2 = time.perf_counter() - t1
            t = torch.tensor([d2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d2 = float(t.item())
            chan2 = {"value": round(nx * reps2 / d2 / 1e6, 1), "unit": "MS/s", "ms_per_step": round(d2 / reps2 * 1e3, 4), "steps": reps2, "scaling": "strong",
                     "path": sc2.chain.path,
                     "sharding": f"channel-interleaved: rank g owns channels g + {world} m (every rank reads the whole stream; DC blocker, pre-mix and FIR are not divided)",
                     "collective": (f"RCCL all-reduce(SUM) of {nf} {'F32' if a.demod == 'fm' else 'CF32'} per step" if a.mix else "none"),
                     "collective_api": ("csdr_chain_process_device_mix (C ABI, RCCL)" if comm is not None else "torch.distributed (gloo test hook)") if a.mix else None,
                     "rccl_ranks": dist.get_world_size()}
            if comm is not None:
                barrier(); tb0 = time.perf_counter()
                comm.broadcast(xc[1].data_ptr(), nx * 8, 0, stream)
                barrier()
                chan2["input_broadcast_ms"] = round((time.perf_counter() - tb0) * 1e3, 4)
            sc2.chain.close()

        # N > 1: SURVEY 8e(B) for the AGC configuration -- linear front end on time stripes, ONE all-to-all of the channel-major CF32
        # plane (RCCL over xGMI), AGC + squelch + freqdem tails on channel blocks -- measured in the same run, reported under "hybrid"
        # (weak scaling like the time stripes: every rank brings its own stripe; `-a 10` whatever --agc says: without the AGC the time
        # stripes need no exchange at all)
        if world > 1 and M % world == 0 and not a.mix and a.demod in ("fm", "none"):
            from composable_sdr_amd.pipes import ChainConfig
            from composable_sdr_amd.sharded import ShardedChain
            agc_h = a.agc if a.agc != 0.0 else 10.0
            sch = ShardedChain(ChainConfig(channels=M, demod=a.demod, kf=a.kf, agc=agc_h, max_frames=nf, device=local, flags=_lib.FLAG_QUIET), mode="hybrid",
                               comm=comm)
            plane = torch.empty(M * nf * 2, dtype=torch.float32, device=dev)
            recv = torch.empty_like(plane)
            reps3 = max(3, a.steps // 2)

            def run3(sub, ov):
                for i in range(2):
                    sch.process_device_hybrid(xv[i & 1], plane, recv, out, stream, substripes=sub, overlap=ov)
                barrier()
                t1 = time.perf_counter()
                for i in range(reps3):
                    sch.process_device_hybrid(xv[i & 1], plane, recv, out, stream, substripes=sub, overlap=ov)
                barrier()
                d = time.perf_counter() - t1
                t = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())
            d3s = run3(2, False)            # the same two rounds of sub-stripes, exchanges in line with the kernels
            d3 = run3(2, True)              # exchanges on a second stream: under the next round's front end / the previous round's tail
            hyb = {"value": round(nx * world * reps3 / d3 / 1e6, 1), "unit": "MS/s", "ms_per_step": round(d3 / reps3 * 1e3, 4), "steps": reps3, "scaling": "weak",
                   "overlap": {"substripes": 2, "exchange_stream": "second stream (events both ways)", "ms_per_step_serial": round(d3s / reps3 * 1e3, 4),
                               "ms_per_step_overlapped": round(d3 / reps3 * 1e3, 4)},
                   "collective_api": "csdr_hybrid_exchange (C ABI: grouped ncclSend / ncclRecv)" if comm is not None else "torch.distributed (gloo test hook)",
                   "agc_db": agc_h, "path": sch.chain.path + " -> all_to_all -> " + sch.tail.path,
                   "sharding": f"hybrid (SURVEY 8e(B)): DC blocker + pre-mix + firpfbch on time stripes (one per rank), all-to-all of the [{M}][{nf}] CF32 plane, "
                               f"AGC + squelch{' + freqdem' if a.demod == 'fm' else ''} on channel blocks of {M // world}",
                   "collective": f"RCCL all_to_all_single of {M * nf * 8 / 2**20:.0f} MiB per rank and step ({world - 1}/{world} of it crosses xGMI)",
                   "rccl_ranks": dist.get_world_size()}
            sch.chain.close(); sch.tail.close()
            del plane, recv


    def side_exit():
        """Leave a run whose SIDE measurements hung or raised (the other ranks may sit in a collective: no barrier with them).  The contract's
        line has been printed by then and carries `side_error`; the failure also goes to stderr.  Exit status: 0 by default -- the driver's
        N > 1 runs are the only multi-GPU numbers there are, and a launcher that sees a non-zero child discards the line with them -- and
        3 under CSDR_BENCH_SIDE_STRICT=1, for a CI that wants a hung RCCL side path to fail the job."""
        print(f"bench.py: rank {rank}: side measurements failed: {side_state['error']}", file=sys.stderr, flush=True)
        os._exit(3 if os.environ.get("CSDR_BENCH_SIDE_STRICT") == "1" else 0)

    def watchdog(limit):
        if side_done.wait(limit):
            return
        side_state["error"] = f"the side measurements (channel_shard / hybrid) did not finish within {limit:.0f} s: left out"
        try:
            if rank == 0:
                finish(True)
        finally:
            sys.stdout.flush()
            side_exit()

    finish_lock = threading.Lock()
    finished = [False]

    def finish(from_watchdog=False):
        nonlocal chan2, hyb
        with finish_lock:                       # the line is printed once: by the main thread, or by the watchdog while the main thread is stuck
            if finished[0]:
                return
            finished[0] = True
            if from_watchdog:
                chan2, hyb = None, None
            _finish(from_watchdog)

    def _finish(no_collectives):
        if rank == 0:
            return _assemble(no_collectives)
        if not (no_collectives or side_state["error"]):
            dist.barrier()
            dist.destroy_process_group()

    def _assemble(no_collectives):
        nonlocal chan2, hyb

        total_samples = float(nx) * a.steps * (1 if chan else world)     # channel shards: every rank works on the same samples
        value = total_samples / dt / 1e6
        alg_bytes_per_sample = 8 + (out_elem / M if a.mix else out_elem)   # read CF32 once + write W
        if per_rank:
            # one rank of a channel-shard run: the whole stream in, its 1 / G of the rows out (--mix: its partial mix, what the all-reduce sums)
            alg_bytes_per_sample = 8 + (out_elem / M if a.mix else out_elem / per_rank)
        kavg_ms = kms / max(klaunches, 1)
        achieved_pairs = (nx * alg_bytes_per_sample) / (kavg_ms * 1e-3) / 1e9 if klaunches else None
        kreg_ms = kms_r / max(klaunches_r, 1)
        # `roofline.achieved` / `frac` are what a STREAM sees: the launch cadence of the dominant kernel (+ its correction kernel) over the long
        # back-to-back run (`sustained_long`: >= preheat_ms at the board's power cap), from one hipEvent in front of that run's first launch
        # and one behind its last on the launch stream.  The K-step window behind a barrier (`frac_k_step_window`) and the K event-paired
        # launches (`frac_event_pairs`) run 2-3 % faster -- the board has just idled for the barrier -- and are reported beside it.
        ksus_ms = kms_l / klaunches_l if klaunches_l else None
        roof_ms = ksus_ms if ksus_ms else kavg_ms
        achieved = (nx * alg_bytes_per_sample) / (roof_ms * 1e-3) / 1e9 if (ksus_ms or klaunches) else None
        # HBM bytes per launch from the committed PMC passes of this very configuration (tools/profile_all.sh +
        # tools/collect_all.sh: FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc runs); null when it has not been profiled
        tj = {}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
            except Exception:
                tj = {}

        # a traffic.json entry is only as good as the kernels it was counted on: every entry carries the hash of the kernel sources
        # (csrc/*.hip, *.h) it was collected with (tools/collect_profile.py); a different hash today -> traffic is reported as null
        src_sha = kernel_sources_sha16()
        stale = []

        def traffic_of(kernel):
            e = tj.get(f"{kernel}|M={M}|nf={nf}", {})
            if e and e.get("src_sha16") != src_sha:
                stale.append(kernel)
                return None
            return e.get("hbm_bytes_per_launch")
        traffic = traffic_of(kname)
        cfg_name = {(64, "none", False): "cfg2", (256, "fm", False): "cfg3", (1024, "fm", False): "cfg4 shape (one GPU)" if world == 1 else "cfg4",
                    (4096, "none", True): "cfg5 shape (one GPU)" if world == 1 else "cfg5"}.get((M, a.demod, bool(a.mix)), "custom")
        if per_rank:
            cfg_name = (f"cfg4 per rank (rank {a.chan_first} of {per_rank})" if (M, a.demod, bool(a.mix)) == (1024, "fm", False) else
                        f"cfg5 per rank (rank {a.chan_first} of {per_rank}: its partial mix)" if (M, a.demod, bool(a.mix)) == (4096, "none", True) else
                        f"rank {a.chan_first} of {per_rank} of a channel-shard run")
        # which partition `value` is -- first key of `config`, so that a truncated copy of the line still says it
        if world == 1:
            sharding_txt = ((f"ONE RANK (chan_first {a.chan_first}) of an N = {per_rank} channel-shard run, alone on this GPU: owns channels {a.chan_first} + {per_rank} m; "
                             "`value` = the rate it gets through the common input stream") if per_rank else "none (1 GPU)")
        elif chan:
            sharding_txt = (f"channel shards (north_star's partition): rank g owns channels g + {world} m of the SAME stream, samples counted once, pruned DFT; "
                            f"collective: {'one RCCL all-reduce per step (--mix)' if a.mix else 'none'}")
        else:
            sharding_txt = (f"independent time stripes, no collective: each of the {world} ranks runs the whole {M}-channel chain on its own stretch of the stream "
                            "(weak scaling: linear by construction); north_star's channel-shard partition of ONE stream is `value_channel_shard` (strong scaling)")
        res = {
            "metric": f"MS/s CF32 throughput, {M}-ch PFB{'+FM' if a.demod == 'fm' else ''}{'+AGC' if a.agc else ''}{' --mix' if a.mix else ''} pipeline" + (f", one rank of {per_rank} (channel shard)" if per_rank else ""), "value": round(value, 1), "unit": "MS/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if chan else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"sharding": sharding_txt, "workload": f"{cfg_name}: {M}-ch firpfbch(m=7,As=80)+dcBlocker{f'+freqdem(kf={a.kf})' if a.demod == 'fm' else ' (DeNo)'}{' --mix' if a.mix else ''} on synthetic CF32, "
                                   f"AGC {'off (-a 0)' if a.agc == 0 else a.agc}, {nf} frames/step "
                                   f"({nx * 8 / 2**20:.0f} MiB in, {M * nf * out_elem / (per_rank or 1) / 2**20:.0f} MiB out), HBM-resident",
                       "channels": M, "frames_per_step": nf, "demod": a.demod, "kf": a.kf, "agc_db": a.agc, "mix": bool(a.mix),
                       "path": f"{chain.path.split('|')[0]}|{kname}", "route": chain.path, "preheat_steps": preheat_steps,
                       "collective": (f"RCCL all-reduce(SUM) of {nf} {'F32' if a.demod == 'fm' else 'CF32'} per step" if (chan and a.mix) else "none"),
                       "collective_api": (("csdr_chain_process_device_mix (C ABI, RCCL)" if comm is not None else "torch.distributed (gloo test hook)")
                                          if (chan and a.mix) else None),
                       "rccl_ranks": (comm.world if comm is not None else (dist.get_world_size() if use_dist else 1))},
            "hbm_roofline_frac_whole_step": round(value * 1e6 * alg_bytes_per_sample / 1e9 / (HBM_PEAK_GBS * (1 if chan else world)), 4),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "frac_of_measured_copy_ceiling": round(achieved / HBM_COPY_GBS, 4) if achieved else None,
                         "traffic": traffic, "launch_ms": round(roof_ms, 4), "launches": klaunches_l if ksus_ms else klaunches,
                         "launch_ms_source": ("hipEvents around the sustained_long run's launches on the launch stream (region / launches): the steady state at the power cap"
                                              if ksus_ms else "hipEvent pairs around K launches (no sustained run: --preheat-ms 0)"),
                         "launch_ms_event_pairs": round(kavg_ms, 4), "launches_event_pairs": klaunches,
                         "frac_event_pairs": round(achieved_pairs / HBM_PEAK_GBS, 4) if achieved_pairs else None,
                         "traffic_source": ("profiles/traffic.json: rocprofv3 --pmc passes of this configuration (FETCH_SIZE x 2 + WRITE_SIZE per launch, "
                                            "tools/profile_all.sh + tools/collect_all.sh), not re-measured in this run; kernel sources unchanged since "
                                            f"(src_sha16 {src_sha})") if not stale else
                                           f"null: profiles/traffic.json was collected on other kernel sources than today's (src_sha16 {src_sha}); re-run tools/profile_all.sh",
                         "launch_ms_k_step_window": round(kreg_ms, 4) if klaunches_r else None,
                         "frac_k_step_window": round(nx * alg_bytes_per_sample / (kreg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if klaunches_r else None,
                         "alg_bytes_per_sample": alg_bytes_per_sample, "samples_per_launch": nx},
            "cold_window": {"steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt_cold / a.steps * 1e3, 4),
                            "value": round(total_samples / dt_cold / 1e6, 1), "unit": "MS/s",
                            "note": "the same W + K steps started on an idle board (no pre-heat): inside the board's power transient"},
        }
        if sus_long:
            res["value_sustained"] = sus_long["value"]      # the same metric over the long run: what a stream sees (`value` is the contract's K-step window)
            sus_long["hbm_roofline_frac_whole_step"] = round(sus_long["value"] * 1e6 * alg_bytes_per_sample / 1e9 / (HBM_PEAK_GBS * (1 if chan else world)), 4)
            res["sustained_long"] = sus_long

        if d_s is not None:
            # beside the contract's K-step window: the same step over 400 further launches, right behind it
            res["sustained"] = {"steps": 400, "ms_per_step": round(d_s * 1e3, 4), "value": round(nx / d_s / 1e6, 1), "unit": "MS/s",
                                "hbm_roofline_frac_whole_step": round(nx * alg_bytes_per_sample / d_s / 1e9 / HBM_PEAK_GBS, 4)}
        if bcast:
            res["input_broadcast"] = bcast
        if chan2:
            res["channel_shard"] = chan2
            res["value_channel_shard"] = chan2["value"]      # north_star's partition (strong scaling), beside `value` (time stripes)
        if hyb:
            res["hybrid"] = hyb
            res["value_hybrid"] = hyb["value"]               # SURVEY 8e(B) with the AGC on (weak scaling)
        if world == 1 and not per_rank and not a.no_agc_variant and a.agc == 0.0 and M == 256 and not a.mix:
            # cfg3 with the AGC on (squelch threshold -a 10 between the tone and the noise channels): the PFB kernel
            # writes channel-major CF32, the time-parallel verified AGC tail (bit-identical to the sequential
            # recurrence) adds squelch + freqdem
            ch2 = cs.Chain(channels=M, demod=a.demod, kf=a.kf, agc=10.0, max_frames=nf, device=local, flags=_lib.FLAG_QUIET)
            for i in range(3 + (40 if a.preheat_ms > 0 else 0)):
                ch2.process_device(xs[i & 1].data_ptr(), nx, out.data_ptr(), stream)
            torch.cuda.synchronize()
            c0, r0 = ch2.agc_stats()
            t1 = time.perf_counter()
            reps = max(3, a.steps // 2) if a.preheat_ms <= 0 else max(a.steps, int(min(a.preheat_ms, 1000.0) * 1e-3 / 0.0006))   # ~1 s of steps
            for i in range(reps):
                ch2.process_device(xs[(i + 1) & 1].data_ptr(), nx, out.data_ptr(), stream)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            c1, r1 = ch2.agc_stats()
            v2 = nx * reps / d2 / 1e6
            res["agc_variant"] = {"value": round(v2, 1), "unit": "MS/s", "agc_db": 10.0, "ms_per_step": round(d2 / reps * 1e3, 4),
                                  "path": ch2.path, "frames_per_step": nf, "steps": reps,
                                  "strategy": "time-parallel AGC+squelch+freqdem tail on a tile-major CF32 plane: one lane per (channel, segment) with a warm-up, "
                                              "segment boundaries verified bitwise, failing segments recomputed (up to the checkpoint where they meet the "
                                              "speculative trajectory) in parallel rounds until all hold (exact)",
                                  "tile_major_calls": ch2.agc_tile_major_calls(),
                                  "segments_checked": c1 - c0, "segments_recomputed": r1 - r0,
                                  "hbm_roofline_frac_whole_step": round(v2 * 1e6 * alg_bytes_per_sample / 1e9 / HBM_PEAK_GBS, 4)}
            # the AGC step is two kernels (channelizer to channel-major CF32 scratch, then the AGC + freqdem tail): achieved
            # = algorithmic bytes of the STEP over the step time; traffic = counter bytes of both launches
            def traffic_tm(kernel):
                e = tj.get(f"{kernel}|M={M}|nf={nf}|tm", {})
                return e.get("hbm_bytes_per_launch") if e.get("src_sha16") == src_sha else None
            ta = traffic_tm("k_run256v2<CF32>") or traffic_of("k_run256v2<CF32>")
            tb = traffic_of("k_agc_spec_tm") or traffic_of("k_agc_spec")
            res["agc_variant"]["roofline"] = {"bound": "hbm", "kernel": "k_run256v2<CF32> (tile-major plane) + k_agc_spec_tm (+ k_agc_fix)",
                                              "achieved": round(nx * alg_bytes_per_sample / (d2 / reps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(nx * alg_bytes_per_sample / (d2 / reps) / 1e9 / HBM_PEAK_GBS, 4),
                                              "traffic": (ta + tb) if (ta and tb) else None}
            ch2.close()

        if world == 1 and not per_rank and not a.no_other_configs and not a.no_agc_variant and a.agc == 0.0 and M == 256 and a.demod == "fm" and not a.mix and nf == 262144:
            # The other BASELINE shapes, each for ~0.5 s on the same (hot) board and the same 67.1 M input samples per step, so that a driver-run line
            # carries them too (verdict r04: "none of cfg2/cfg4/cfg5 figures is driver-run").  Whole-step figures (host clock around a back-to-back
            # loop); the kernels behind them are profiled under profiles/rNN_*.  Not the headline: `value` above is.
            out2 = torch.empty(nx * 2, dtype=torch.float32, device=dev)
            others = []
            def side(tag, workload, M2, nf2, demod2, agc2, mix2, bps, G2=0, g2=0):
                try:
                    c2 = cs.Chain(channels=M2, demod=demod2, kf=a.kf, agc=agc2, mix=mix2, max_frames=nf2, device=local, flags=_lib.FLAG_QUIET,
                                  chan_first=g2, chan_stride=G2)
                    n2 = M2 * nf2
                    for i in range(6):
                        c2.process_device(xs[i & 1].data_ptr(), n2, out2.data_ptr(), stream)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    c2.process_device(xs[0].data_ptr(), n2, out2.data_ptr(), stream)
                    torch.cuda.synchronize()
                    one = max(time.perf_counter() - t1, 1e-5)
                    reps = max(5, int(0.5 / one))
                    t1 = time.perf_counter()
                    for i in range(reps):
                        c2.process_device(xs[(i + 1) & 1].data_ptr(), n2, out2.data_ptr(), stream)
                    torch.cuda.synchronize()
                    d = (time.perf_counter() - t1) / reps
                    e = {"tag": tag, "workload": workload, "channels": M2, "frames_per_step": nf2, "demod": demod2, "agc_db": agc2, "mix": bool(mix2),
                         "route": c2.path, "kernel": c2.kernel_time()[0], "steps": reps, "ms_per_step": round(d * 1e3, 4), "value": round(n2 / d / 1e6, 1), "unit": "MS/s",
                         "alg_bytes_per_sample": bps}
                    if G2 > 1:
                        # ONE rank of an N = G2 channel-shard run, timed alone on this GPU: `value` is the rate at which that rank gets through the
                        # COMMON input stream (every rank reads all of it: 8 B per sample, plus its 1 / G2 of the output), which is also the whole
                        # job's rate when the G2 ranks run side by side on G2 GPUs (the samples are counted once)
                        e.update({"chan_stride": G2, "chan_first": g2, "owned_channels": M2 // G2, "per_rank": True, "us_per_step": round(d * 1e6, 2),
                                  "input_gs_per_s": round(n2 / d / 1e9, 2)})
                    if n2 * bps >= (64 << 20):
                        e["hbm_roofline_frac_whole_step"] = round(n2 * bps / d / 1e9 / HBM_PEAK_GBS, 4)
                    c2.close()
                except Exception as ex:        # a side measurement never takes the line down
                    e = {"tag": tag, "workload": workload, "error": str(ex)[:200]}
                others.append(e)
            side("cfg2", "64-ch PFB, DeNo (BASELINE configs[1] shape)", 64, 1048576, "none", 0.0, False, 16)
            side("cfg3_deno", "256-ch PFB, DeNo", 256, 262144, "none", 0.0, False, 16)
            side("cfg4_shape_1gpu", "1024-ch PFB + FM, all channels on one GPU (BASELINE configs[3] shape)", 1024, 65536, "fm", 0.0, False, 12)
            side("m1024_deno", "1024-ch PFB, DeNo", 1024, 65536, "none", 0.0, False, 16)
            side("cfg5_shape_1gpu", "4096-ch PFB, DeNo --mix over all channels = the mix identity (BASELINE configs[4] shape)", 4096, 16384, "none", 0.0, True, 8)
            side("m4096_deno", "4096-ch PFB, per-channel DeNo (fused 4096 route)", 4096, 16384, "none", 0.0, False, 16)
            side("m4096_fm", "4096-ch PFB + FM per channel (fused 4096 route)", 4096, 16384, "fm", 0.0, False, 12)
            side("m4096_fm_mix", "4096-ch PFB + FM --mix (fused 4096 route)", 4096, 16384, "fm", 0.0, True, 8)
            # BASELINE configs[3] / north_star's partition, per rank: rank 0 of 8 (channels 0, 8, 16, ...: Trans.hs:124-129, SoapySDR.hs:223-225), alone on this GPU
            side("shard_g8_m256_fm", "rank 0 of 8, interleaved channel shard of the 256-ch PFB + FM chain (32 owned channels)", 256, 262144, "fm", 0.0, False, 8 + 4 / 8, 8, 0)
            side("shard_g8_m256_fm_agc", "rank 0 of 8, interleaved channel shard of the 256-ch PFB + AGC (-a 10) + FM chain", 256, 262144, "fm", 10.0, False, 8 + 4 / 8, 8, 0)
            side("shard_g8_m1024_fm", "rank 0 of 8, interleaved channel shard of the 1024-ch PFB + FM chain (128 owned channels): BASELINE configs[3] per rank", 1024, 65536, "fm", 0.0, False, 8 + 4 / 8, 8, 0)
            side("shard_g8_m1024_fm_agc", "rank 0 of 8, interleaved channel shard of the 1024-ch PFB + AGC (-a 10) + FM chain", 1024, 65536, "fm", 10.0, False, 8 + 4 / 8, 8, 0)
            side("shard_g2_m1024_fm", "rank 0 of 2, interleaved channel shard of the 1024-ch PFB + FM chain", 1024, 65536, "fm", 0.0, False, 8 + 4 / 2, 2, 0)
            side("shard_g8_m4096_deno_mix", "rank 0 of 8, interleaved channel shard of the 4096-ch PFB, DeNo --mix (the shard's mix identity: the 8 surviving polyphase "
                 "branches; its partial mix is what the all-reduce sums): BASELINE configs[4] per rank", 4096, 16384, "none", 0.0, True, 8 + 8 / 4096, 8, 0)
            side("cfg2_agc", "64-ch PFB + AGC (-a 10), DeNo (configs[1] shape with the AGC on)", 64, 1048576, "none", 10.0, False, 16)
            side("cfg4_shape_1gpu_agc", "1024-ch PFB + AGC (-a 10) + FM, all channels on one GPU", 1024, 65536, "fm", 10.0, False, 12)
            side("ref_chunk_m256_fm", "the reference's own chunk: 256 ch x 4096 frames per call, FM", 256, 4096, "fm", 0.0, False, 12)
            side("ref_chunk_m256_fm_agc", "the reference's own chunk: 256 ch x 4096 frames per call, AGC (-a 10) + FM", 256, 4096, "fm", 10.0, False, 12)
            side("ref_chunk_m4096_deno", "the reference's own chunk: 4096 ch x 4096 frames per call, DeNo", 4096, 4096, "none", 0.0, False, 16)
            res["other_configs"] = others
            del out2

        if world == 1 and not per_rank and not a.no_cpu_baseline:
            x_host = xs[0][: 4096 * M * 4].cpu().numpy().view(np.complex64).reshape(-1)
            res["cpu_baseline"] = cpu_baseline(M, a.demod, a.kf, a.agc, x_host, a.cpu_seconds, a.mix)
        if side_state["error"]:
            res["side_error"] = side_state["error"]
        print(json.dumps(res), flush=True)
        if use_dist and not (no_collectives or side_state["error"]):
            dist.barrier()
            dist.destroy_process_group()

    if world > 1:
        limit = float(os.environ.get("CSDR_BENCH_SIDE_TIMEOUT", "300"))
        threading.Thread(target=watchdog, args=(limit,), daemon=True).start()
        try:
            run_sides()
        except Exception as ex:                 # (the other ranks may be inside a collective now: they leave through their watchdogs)
            side_state["error"] = f"{type(ex).__name__}: {ex}"[:300]
            chan2, hyb = None, None
    side_done.set()

    finish(False)
    if side_state["error"] and use_dist:
        sys.stdout.flush()
        side_exit()                             # the other ranks may be stuck in a collective: no barrier with them


if __name__ == "__main__":
    main()
