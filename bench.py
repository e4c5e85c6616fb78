#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the continuous wavelet transform.

    python bench.py --gpus N --steps K --warmup W [--config 3|5]

--config 3 (default; BASELINE.json configs[2], and configs[3] at N = 8): 128 channels per
GPU x 1e6 samples @ 1 kHz x 100 Morse scales (log-spaced 200..2 Hz), amplitude output,
input and output resident in HBM.  A "step" is one gcwt_execute over the rank's block.
--config 5 (BASELINE.json configs[4]): 48 channels per GPU (384 over 8) x 18e6 samples
@ 30 kHz (one 10-minute epoch) x 200 scales 500..1 Hz, streamed: the plan's overlapping
time blocks are computed one after the other, 24 channels at a time, into a ring of two
72 GB device buffers (the 691 GB of output per GPU never exist at once).  A step is one pass over
the rank's channels and the whole epoch.

N > 1: one process per GPU.  Under ``python -m torch.distributed.run`` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) this process is one rank; run bare with
``--gpus N`` it starts the N ranks itself, as fresh child processes, before anything here
touches HIP.  Each rank transforms its own channel block (weak scaling, no data-path
collective); rank 0's filter bank is broadcast once with RCCL.  Rank 0 prints one JSON
line on the real stdout; everything else any library prints goes to stderr.
No torch anywhere in this file.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E data sheet (guides/MI355X_MICROARCH.md)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # guides/MI355X_MICROARCH.md: Peak FP32 (vector), spec

# The result line goes to the stdout this process was started with; fd 1 itself is pointed
# at stderr for the whole run, so that nothing a library prints (RCCL's banner) can land
# in front of the line -- and no call ever has to redirect around itself.
_RESULT = os.fdopen(os.dup(1), "w")
os.dup2(2, 1)
sys.stdout = sys.stderr


def emit(obj):
    _RESULT.write(json.dumps(obj) + "\n")
    _RESULT.flush()


# ----------------------------------------------------------------------------
# launcher: N fresh ranks, no HIP call in this process
# ----------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, deadline_s=3600.0):
    """Starts the n ranks, relays rank 0's result line.  Rank 0's pipe is drained by a reader
    thread and every rank is polled in one loop under one deadline: the first rank that exits
    non-zero (or the deadline) ends the others, and the launcher exits non-zero."""
    import threading
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + deadline_s
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = "rank %d exited with %d" % (r, codes[r])
        if failed is None and time.time() > deadline:
            failed = "no result after %.0f s" % deadline_s
        if failed is not None:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.kill()
                    codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(5.0)
    if failed is not None:
        sys.stderr.write("bench.py: %s; ranks ended with %s\n" % (failed, codes))
        return 1
    _RESULT.write(b"".join(chunks).decode())
    _RESULT.flush()
    return 0


# ----------------------------------------------------------------------------
# checkers (rank 0, N = 1, outside the timed region): the oracle is test infrastructure
# ----------------------------------------------------------------------------
def cpu_baseline(fs, freqs, budget_s=25.0):
    """The oracle's literal path (port of transforms.py:187-224 + convolution.py:16-87) on this
    box's cores, the legs BASELINE.md plans: config 2's shape (1 ch x 1e6 samples, the bench's
    scales) with ThreadPool over scales like ``parallel=True`` (transforms.py:206-218) -- the
    headline `value` -- and serially (:219-224), plus config 1's shape (1 ch x 16 384 samples,
    32 scales 200..5.57 Hz) serially.  About 25 s of CPU work in all."""
    from oracle import ghost_oracle as orc
    from ghost_amd.synthetic import lfp_channel
    affinity = len(os.sched_getaffinity(0))
    cores = min(affinity, 16)                       # a 1-GPU box's CPU share is 16 cores

    def leg(n, f, threads, min_reps, max_s):
        x = lfp_channel(n, fs, 0).astype(np.float64)
        best, reps, t_start = None, 0, time.time()
        while reps < min_reps or (time.time() - t_start < max_s and reps < 60):
            t0 = time.time()
            orc.cwt_amplitude(x, fs, f, n_threads=threads)
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
            reps += 1
            if time.time() - t_start > max_s:
                break
        return {"value": round(n / best / 1e6, 4), "unit": "Msamples/s", "threads": threads,
                "shape": "1 ch x %d samples x %d scales" % (n, len(f)), "best_s": round(best, 3), "runs": reps}

    f1 = 200.0 / 2.0 ** (np.arange(32) / 6.0)          # config 1: 32 scales, 6 voices per octave
    pool = leg(1000000, freqs, cores, 2, budget_s * 0.3)
    # what the reference's parallel=True does on this box: ThreadPool(cpu_count()) (transforms.py:210) -- every
    # core the process may run on, at most one thread per scale
    every = max(1, min(affinity, len(freqs)))
    pool_all = leg(1000000, freqs, every, 2, budget_s * 0.2) if every != cores else dict(pool)
    serial = leg(1000000, freqs, 1, 1, budget_s * 0.4)
    m1 = leg(16384, f1, 1, 3, budget_s * 0.1)
    return {"value": pool["value"], "unit": "Msamples/s", "cores": cores,
            "os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "kind": "port",
            "sample": "1 ch x 1000000 samples x %d scales (config 2 of the same fs and scales), float64, "
                      "scipy.fft overlap-add, ThreadPool(%d) over scales, best of %d runs (%.2f s each)"
                      % (len(freqs), cores, pool["runs"], pool["best_s"]),
            "legs": {"config2_threadpool": pool, "config2_threadpool_all_cores": pool_all, "config2_serial": serial,
                     "config1_serial": m1}}


def oracle_rows_window(x_full, fs, freqs, a, b):
    """|W| of samples [a, b) of one channel at a few scales, from the oracle's own pieces
    (transforms.py:142-143 global mean, morse.py:108-122 lengths, morseutils.py:93-198 kernel,
    convolution.py:16-87 overlap-add) on a window of the recording: every output needs the
    input within (L-1)/2 of it, so [a - L, b + L) gives the same numbers as the whole array."""
    from oracle import ghost_oracle as orc
    xc = np.asarray(x_full, dtype=np.float64)
    xc = xc - xc.mean()
    n = xc.size
    om = orc.hz_to_rad(freqs, fs)
    out = []
    for w, L in zip(om, orc.morse_lengths(om)):
        psi, _ = orc.morse_kernel(int(L), w)
        w0, w1 = max(0, a - int(L)), min(n, b + int(L))
        y = orc.overlap_add_convolve(xc[w0:w1], psi)
        out.append(np.abs(y[a - w0:b - w0]))
    return np.array(out)


class PowerSampler:
    """Package power and shader clock of the busiest AMD GPU while the timed steps run: a side
    thread that only reads sysfs (hwmon power1_average / power1_input, freq1_input; no HIP call,
    no subprocess of a GPU tool)."""

    def __init__(self, period=0.02, pci_bus_id=None):
        import glob
        self.period, self.rows, self._stop, self._th = period, [], False, None
        self.cards = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f))]
            if pw:
                self.cards.append({"hw": hw, "power": os.path.join(hw, pw[0]),
                                   "sclk": os.path.join(hw, "freq1_input"),
                                   "cap": os.path.join(hw, "power1_cap"),
                                   "pci": os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))).lower()})
        # this rank's own GPU when sysfs names it (a host has eight); else the busiest card is reported
        mine = [c for c in self.cards if pci_bus_id and c["pci"] == pci_bus_id.lower()]
        self.matched = bool(mine)
        if mine:
            self.cards = mine

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            self.rows.append([(self._read(c["power"]), self._read(c["sclk"])) for c in self.cards])
            time.sleep(self.period)

    def start(self):
        if self.cards:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join(1.0)
        if not self.rows:
            return None
        best = None
        for i, c in enumerate(self.cards):
            pw = [r[i][0] for r in self.rows if r[i][0] is not None]
            ck = [r[i][1] for r in self.rows if r[i][1] is not None]
            if not pw:
                continue
            mean_w = sum(pw) / len(pw) / 1e6
            if best is None or mean_w > best["power_w"]:
                cap = self._read(c["cap"])
                best = {"power_w": round(mean_w, 1), "power_w_max": round(max(pw) / 1e6, 1),
                        "power_cap_w": round(cap / 1e6, 1) if cap else None,
                        "sclk_ghz": round(sum(ck) / len(ck) / 1e9, 3) if ck else None,
                        "sclk_ghz_min": round(min(ck) / 1e9, 3) if ck else None,
                        "samples": len(pw), "source": c["hw"],
                        "card": "the rank's device (PCI %s)" % c["pci"] if self.matched else "busiest card seen"}
        return best


def check_scales(S, dec=None):
    """The rows of a channel the bench compares with the oracle: the first and the last scale and, when the plan's
    decimations are given, the middle scale of every decimation level (each level is its own kernel launch
    geometry: seven levels on the headline grid)."""
    scales = {0, (57 * S) // 100, S - 1}
    if dec is not None:
        dec = np.asarray(dec)
        for r in sorted(set(dec.tolist())):
            idx = np.nonzero(dec == r)[0]
            scales.add(int(idx[len(idx) // 2]))
    return sorted(scales)


def spot_check(obuf, base, fs, freqs, C, N, distinct, output="amplitude", dec=None):
    """Looks at what the timed steps wrote: rows (channel, scale) of the device result
    against the oracle (transforms.py:187-204), and the tiled channels c and c + distinct
    (same input) bit for bit.  Returns (ok, worst relative error)."""
    from oracle import ghost_oracle as orc
    S = len(freqs)
    chans = sorted({0, min(7, C - 1), C - 1})
    scales = check_scales(S, dec)
    dtype, width = (np.complex64, 8) if output == "complex" else (np.float32, 4)
    worst, same = 0.0, True
    for c in chans:
        xc = base[c % distinct].astype(np.float64)
        if output == "complex":
            ref = orc.cwt_complex(xc, fs, freqs[scales])
        else:
            ref = orc.cwt_amplitude(xc, fs, freqs[scales])
            if output == "power":
                ref = ref ** 2
        for i, s in enumerate(scales):
            row = obuf.download((N,), dtype, offset_bytes=width * (c * S + s) * N)
            worst = max(worst, float(np.abs(row - ref[i]).max() / np.abs(ref[i]).max()))
            twin = c + distinct if c + distinct < C else c - distinct
            if 0 <= twin < C and twin != c:
                other = obuf.download((N,), dtype, offset_bytes=width * (twin * S + s) * N)
                same = same and np.array_equal(row, other)
    gate = 2e-5 if output == "power" else 1e-5
    return bool(worst <= gate and same), worst


def full_output_check(obuf, C, N, S, distinct, output="amplitude"):
    """Every value of the device result, on the device (gcwt_debug_check_output): none may be Inf / NaN,
    and every row of a channel c >= distinct must equal the same row of channel c % distinct bit for bit
    (the bench tiles `distinct` recordings over its channels) -- all C x S rows, not a sample of them."""
    import ctypes
    from ghost_amd._lib import lib, check
    w = 2 if output == "complex" else 1
    bad, diff = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.gcwt_debug_check_output(obuf.ptr, w * N, w * N, S, C, distinct, ctypes.byref(bad), ctypes.byref(diff)))
    return {"rows": C * S, "values": C * S * N * w, "nonfinite": bad.value, "tiled_twin_mismatches": diff.value,
            "ok": bad.value == 0 and diff.value == 0}


# ----------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------
def run_rank(args):
    from ghost_amd.dist import Comm, env_rank, shard_channels
    from ghost_amd.engine import CwtPlan, DeviceBuffer, device_count, device_name
    from ghost_amd._lib import lib, check
    from ghost_amd.synthetic import lfp
    import ctypes

    rank, world, local = env_rank()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if args.dry_run:                                # launcher rehearsal: no HIP call at all
        fail = os.environ.get("GHOSTCWT_BENCH_FAIL_RANK")   # (tests: one rank dies, the others would wait)
        if fail is not None:
            if int(fail) == rank:
                return 3
            time.sleep(120)
        path = os.path.join(args.dry_run, "rank%d.json" % rank)
        json.dump({"rank": rank, "local_rank": local, "world": world,
                   "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))},
                  open(path, "w"))
        if rank == 0:
            emit({"dry_run": True, "n_gpus": world})
        return 0
    ndev = device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an AMD GPU: libghostcwt has no CPU path")
    if ndev < world and not os.environ.get("GHOSTCWT_ALLOW_SHARED_GPU"):
        raise SystemExit("bench.py --gpus %d: only %d device(s) visible; ranks never share a GPU "
                         "(GHOSTCWT_ALLOW_SHARED_GPU=1 to rehearse on one)" % (world, ndev))
    dev = local % ndev
    check(lib.gcwt_set_device(dev))
    numa, n_pinned = (None, 0)
    if world > 1:                                    # each rank beside its own GPU (best effort)
        from ghost_amd.dist import pin_to_device_numa
        numa, n_pinned = pin_to_device_numa(dev)

    cfg5 = args.config == 5
    fs = 30000.0 if cfg5 else 1000.0
    C = args.channels or (48 if cfg5 else 128)
    N = args.samples or (18000000 if cfg5 else 1000000)
    S = args.scales or (200 if cfg5 else 100)
    freqs = np.geomspace(500.0, 1.0, S) if cfg5 else np.geomspace(200.0, 2.0, S)
    group = C                                        # channels per plan execution
    if cfg5:
        group = min(C, args.group or 24)
        while C % group:                             # the largest divisor of C within the request
            group -= 1
    time_shard = args.shard == "time"
    if time_shard and not cfg5:
        raise SystemExit("--shard time is config 5's split (few channels, long recording): add --config 5")
    if time_shard:
        total_channels = C                           # strong scaling: every rank holds all C channels and takes
    else:                                            # its share of the recording's time blocks
        total_channels = C * world                   # weak scaling: fixed channels per GPU
        c0, c1 = shard_channels(total_channels, rank, world)
        assert c1 - c0 == C

    comm = Comm(rank, world, device=dev)
    t_plan = time.perf_counter()
    plan = CwtPlan(N, group, fs, freqs, output=args.output, device=dev, max_fft_log2=args.max_fft_log2)
    plan.upload()                                    # workspace, filter bank, FFT tables
    check(lib.gcwt_device_synchronize())
    plan_ms = (time.perf_counter() - t_plan) * 1e3   # reported apart from the timed steps
    bank_via = comm.broadcast_bank(plan, root=0)
    # The timed steps carry HIP events around the synthesis kernels only (level 2: `roofline` comes from them, measured
    # live on the kernels' own stream).  Events around EVERY stage cost the step they measure 0.19 ms of 13.5
    # (tools/step_gap.py: 25 spans, each an event record between two dependent launches), so the stage breakdown
    # (`stages_ms`) is taken from further steps after the timed region, outside it.
    plan.set_profiling(2)
    info = plan.info

    # what this box's HBM delivers (measured once, before anything is timed)
    ceilings = {}
    if rank == 0 and not args.no_ceilings:
        # store_pattern_ceiling: k_synthi's own pattern (1 KB runs per wave, 53 KB visits; round 4) -- the kernel that
        # writes 63 of the 100 scales; store_pattern_ceiling_synth7: k_synth7's 128-byte runs (rounds 1-3's probe)
        for key, pat in (("peak_measured_copy", 1), ("peak_measured_fill", 0), ("store_pattern_ceiling", 3),
                         ("store_pattern_ceiling_synth7", 2)):
            g = ctypes.c_double(0)
            check(lib.gcwt_debug_bandwidth(pat, 12 << 30 if pat == 1 else 48 << 30, ctypes.byref(g)))
            ceilings[key] = round(g.value, 1)
    comm.barrier()

    # synthetic LFP: 8 distinct generated channels per rank, tiled over the block
    distinct = min(C, 8 if not cfg5 else 2)
    base = lfp(distinct, N, fs, seed=1234 + (0 if time_shard else 1000 * rank))   # (time shards: one recording)
    xbuf = DeviceBuffer(4 * C * N)
    for c in range(C):
        xbuf.upload(base[c % distinct], offset_bytes=4 * c * N)
    b_out = 8 if args.output == "complex" else 4
    segs = plan.segments()
    if time_shard:
        # BASELINE.json north_star: "independent channels/epochs shard embarrassingly" -- for few channels the
        # independent units are the time blocks (each carries its own halo: no exchange)
        from ghost_amd.dist import shard_time_blocks
        lo, hi = shard_time_blocks(segs, rank, world)
        all_segs = segs
        segs = [sg for sg in segs if lo <= sg[0] < hi]
        if not segs:
            raise SystemExit("rank %d of %d has no time block: %d blocks in the recording" % (rank, world, len(all_segs)))
    if cfg5:
        core = max(b - a for a, b, _ in segs)
        ring = [DeviceBuffer(b_out * group * S * core) for _ in range(2)]

        def step(stats):
            k = 0
            for g in range(C // group):
                xg = ctypes.c_void_p(xbuf.ptr.value + 4 * g * group * N)
                for i, (a, b, _) in enumerate(segs):
                    plan.execute_block_device(xg, ring[k & 1], a, b - a, reuse_means=i > 0)
                    k += 1
                    tm = plan.timings()
                    for key, v in tm.items():
                        stats[key] = stats.get(key, 0) + v
    else:
        obuf = DeviceBuffer(info["out_bytes"])

        def step(stats):
            plan.execute_device(xbuf, obuf)          # returns after the stream has drained
            for key, v in plan.timings().items():
                stats[key] = stats.get(key, 0) + v

    for _ in range(args.warmup):
        step({})
    check(lib.gcwt_device_synchronize())
    bus = None
    try:
        buf = ctypes.create_string_buffer(64)
        check(lib.gcwt_device_pci_bus_id(dev, buf, 64))
        bus = buf.value.decode()
    except Exception:
        pass
    sampler = PowerSampler(pci_bus_id=bus) if rank == 0 else None
    comm.barrier()
    stats = {}
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(stats)
    check(lib.gcwt_device_synchronize())
    own = time.perf_counter() - t0                   # this rank's own K steps, before it waits for the others
    comm.barrier()
    elapsed = time.perf_counter() - t0
    power = sampler.stop() if sampler else None
    elapsed = comm.allreduce_max(elapsed)
    sustained = None
    if rank == 0 and world == 1 and power and args.sustain > 0:
        # hwmon's power reading is a moving average that lags a 0.1 s burst: the same step is run
        # for `--sustain` seconds more, AFTER and OUTSIDE the timed region, and sampled again
        s2 = PowerSampler(pci_bus_id=bus)
        t_s = time.perf_counter()
        time.sleep(0.0)
        n_s = 0
        while time.perf_counter() - t_s < args.sustain:
            if n_s == 3:
                s2.start()                               # (the first steps bring the average up)
            step({})
            n_s += 1
        check(lib.gcwt_device_synchronize())
        sustained = s2.stop()
        if sustained:
            sustained["steps"], sustained["seconds"] = n_s, round(time.perf_counter() - t_s, 2)
    # the stage breakdown: the same step again with every stage between events, after and outside the timed region
    stage_stats, n_stage_steps = {}, min(args.steps, 10)
    plan.set_profiling(True)
    for i in range(n_stage_steps + 1):
        step(stage_stats if i > 0 else {})               # (the first one after the switch is not counted)
    check(lib.gcwt_device_synchronize())
    plan.set_profiling(2)
    per_rank = comm.allgather(own / args.steps * 1e3)
    devices = comm.allgather(dev)
    numas = comm.allgather(-1 if numa is None else numa)

    if rank == 0:
        units = (1 if time_shard else world) * C * N * args.steps
        value = units / elapsed / 1e6
        launches = max(1, stats["synth_launches"])
        k_ms = stats["synth_ms"] / launches
        ki_ms = stats.get("interp_ms", 0.0) / launches
        alg_step = C * N * (4 + S * b_out)           # per channel-sample 4 + S*b_out
        if time_shard:                               # this rank's share of the recording
            alg_step = C * sum(b - a for a, b, _ in segs) * (4 + S * b_out)
        alg_launch = alg_step * args.steps / launches
        achieved = alg_launch / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (C, N, S, args.output, args.config) == (128, 1000000, 100, "amplitude", 3):
            tj = json.load(open(tpath))
            traffic = tj.get("k_synth_hbm_bytes_per_launch")
            traffic_src = "profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
                          "this command (both synthesis kernels), not measured in this run" % tj.get("round", "r01")
        if cfg5:
            workload = ("config 5: %d ch/GPU x %d samples @ 30 kHz x %d Morse scales 500..1 Hz, %s f32 out, "
                        "streamed in %d time blocks x %d channel groups into a ring of 2 device buffers"
                        % (C, N, S, args.output, len(segs), C // group))
        else:
            workload = ("%d ch/GPU x %d samples @ 1 kHz x %d Morse scales 200..2 Hz, %s f32 out, "
                        "device-resident" % (C, N, S, args.output))
        line = {
            "metric": "Msamples/s CWT (128ch x 1e6 samp x 100 scales); % HBM roofline; 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if time_shard else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic %g kHz LFP (pink noise + 8 Hz rhythm + 40 Hz bursts), "
                    "%d generated channels tiled over each GPU's block" % (fs / 1e3, distinct),
            "config": {"workload": workload, "channels_total": total_channels,
                       "parallelism": ("time-block-sharded x%d (every rank: all %d channels, %d of the recording's "
                                       "%d time blocks)" % (world, C, len(segs), len(all_segs))) if time_shard
                                      else "channel-sharded x%d" % world,
                       "bank": bank_via, "comm": comm.backend, "device": device_name(dev),
                       "plan_create_ms": round(plan_ms, 2),
                       "scales": {"spectral": info["n_spectral"], "direct": info["n_direct"],
                                  "fullband": info["n_fullband"]}},
            "roofline": {"bound": "hbm",
                         "kernel": "synthesis: k_synthi (%d interpolated scales, R >= 16) + k_synth7 (%d scales: R = 4, 8 in "
                                   "its 32-column instantiation, R = 2 in the 16-column one), launched once each per step, "
                                   "back to back" % (info["n_interp"], info["n_spectral"] - info["n_interp"])
                                   if info["n_interp"] else "k_synth7",
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": round(k_ms, 4), "launches_per_step": launches // args.steps,
                         "algorithmic_bytes": int(alg_launch), **ceilings},
            # every stage between HIP events, over `stages_steps` further steps after the timed region (their events
            # slow a step by ~0.19 ms: `total_ms` here is longer than `ms_per_step`); the timed steps themselves
            # bracket the synthesis kernels only (`roofline.kernel_ms`)
            "stages_ms": {k: round(float(v) / n_stage_steps, 4) for k, v in stage_stats.items() if k.endswith("_ms")},
            "stages_steps": n_stage_steps,
            "timed_region_events": "synthesis kernels only (gcwt_plan_set_profiling level 2)",
            "whole_job_frac_of_hbm_peak": round(alg_step * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 4),
        }
        rl = line["roofline"]
        if info["n_interp"]:
            # the two synthesis kernels apart: rows x samples x 4 B each wrote, over its own launch time
            # (per launch, like kernel_ms: config 5 makes several launches per step)
            rows_launch = C * N * b_out * args.steps / launches        # bytes of one scale's rows per launch
            b_i = rows_launch * info["n_interp"]
            b_7 = rows_launch * (S - info["n_interp"])
            k7_ms = max(1e-9, k_ms - ki_ms)
            rl["kernels"] = {
                "k_synthi": {"ms": round(ki_ms, 4), "scales": info["n_interp"],
                             "achieved": round(b_i / (ki_ms * 1e-3) / 1e9, 1) if ki_ms > 0 else None},
                "k_synth7": {"ms": round(k7_ms, 4), "scales": S - info["n_interp"],
                             "achieved": round(b_7 / (k7_ms * 1e-3) / 1e9, 1)}}
        if power:
            # measured in this run (sysfs, 20 ms period): during the timed steps, and -- because the
            # power reading is a slow moving average -- over a longer run of the same step after them
            rl["power_w"], rl["sclk_ghz"] = power["power_w"], power["sclk_ghz"]
            rl["power_detail"] = {"timed_steps": power, "sustained": sustained}
            ref = sustained or power
            cap = ref.get("power_cap_w")
            if sustained:
                rl["power_w_sustained"], rl["sclk_ghz_sustained"] = sustained["power_w"], sustained["sclk_ghz"]
            # What was measured beside the kernels, not a verdict: the package's peak and mean power against its cap,
            # the clock the card reported, and how close the launch came to this box's own store-pattern ceiling.  What
            # these mean for the synthesis is profiles/r06_bound.md's CU-mask sweep (a fixed number of CU-cycles run at
            # the clock the cap allows; the R = 64 / 128 levels at the store path's rate).
            rl["limits_measured"] = {
                "power_w_max": ref["power_w_max"], "power_w_mean": ref["power_w"], "power_cap_w": cap,
                "power_frac_of_cap": round(ref["power_w_max"] / cap, 3) if cap else None,
                "sclk_ghz_mean": ref["sclk_ghz"], "sclk_ghz_max": 2.4,
                "frac_of_store_pattern_ceiling": round(achieved / ceilings["store_pattern_ceiling"], 4) if ceilings else None,
                "study": "profiles/r06_bound.md"}
        else:
            rl["power_w"] = rl["sclk_ghz"] = rl["limits_measured"] = None   # no readable hwmon
        if not cfg5:
            # SURVEY.md 8d's secondary ceiling, fp32 vector flops.  `reference_method_*`: what the
            # reference's method implies -- one real P-point FFT per channel, one P-point inverse
            # FFT + multiply + magnitude per (channel, scale); NOT what this engine executes.
            # `executed_*`: the engine's own arithmetic, counted from its kernels:
            # 46 flops per stored sample on the FFT-per-sample levels, 2 x 8 x 2 + 3 = 35 on the
            # interpolated ones plus their q / R share of the transform.
            P = float(info["fft_length"])
            nominal = C * (2.5 * P * np.log2(P) + S * (5.0 * P * np.log2(P) + 3.0 * P + 4.0 * N))
            executed = C * N * ((S - info["n_interp"]) * 46.0 + info["n_interp"] * (35.0 + 46.0 * 0.1))
            rl["fp32_vector"] = {
                "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                "executed_flops_per_step": float("%.4g" % executed),
                "executed_tflops": round(executed * args.steps / elapsed / 1e12, 1),
                "executed_frac": round(executed * args.steps / elapsed / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4),
                "reference_method_flops_per_step": float("%.4g" % nominal),
                "reference_method_tflops": round(nominal * args.steps / elapsed / 1e12, 1)}
        if ceilings:
            line["roofline"]["frac_of_store_pattern_ceiling"] = round(
                achieved / ceilings["store_pattern_ceiling"], 4)
        if comm.rccl_error:
            line["config"]["rccl_error"] = comm.rccl_error[:200]
        if world > 1:
            # where the skew is: every rank's own ms per step (before the closing barrier) and device
            slow = int(np.argmax(per_rank))
            line["ranks"] = {"ms_per_step": [round(v, 4) for v in per_rank],
                             "min": round(min(per_rank), 4), "max": round(max(per_rank), 4), "rank_of_max": slow,
                             "devices": [int(d) for d in devices],
                             "numa_nodes": [int(v) for v in numas],   # -1: not pinned (sysfs silent or outside the affinity)
                             "pinned_cpus_rank0": n_pinned}
        if cfg5 and not args.no_check:
            line["checked"], line["check"] = check_config5(plan, xbuf, ring[0], base, distinct, fs, freqs, N, S,
                                                           group, segs)
        if not cfg5 and not args.no_check:
            dec = plan.scale_info()["decimation"]
            ok, worst = spot_check(obuf, base, fs, freqs, C, N, distinct, output=args.output, dec=dec)
            full = full_output_check(obuf, C, N, S, distinct, output=args.output)
            line["checked"] = bool(ok and full["ok"])
            line["check"] = {"rows": "channels {0,7,C-1} x scales %s (first, last, 57 %% and the middle one of every "
                                     "decimation level) vs the oracle, tiled channels c / c+%d bit-equal"
                                     % (check_scales(S, dec), distinct),
                             "worst_rel_err": float("%.3g" % worst), "full_output": full}
        if world == 1 and not cfg5 and args.output == "amplitude" and not args.no_other_modes:
            # Secondary measurement, after and outside the timed region: the same workload with the
            # complex coefficients stored (8 B per coefficient: SURVEY.md 8d's second target).
            obuf.free()
            plan.close()
            line["other_modes"] = {"complex": other_mode("complex", N, C, fs, freqs, S, xbuf, dev, lib, check,
                                                         steps=args.steps, warmup=args.warmup,
                                                         check_against=None if args.no_check else (base, distinct))}
        if world == 1 and not cfg5 and not args.no_other_configs:
            # BASELINE.json's configs 2 and 5, after and outside the timed region, so that the driver's run
            # carries them too (round 3: builder-run only).  Everything of the headline is released first.
            for b in (locals().get("obuf"), xbuf):
                try:
                    b.free()
                except Exception:
                    pass
            try:
                plan.close()
            except Exception:
                pass
            line["other_configs"] = {"config1": config1_leg(fs, dev, lib, check, no_check=args.no_check),
                                     "config2": config2_leg(fs, freqs, dev, lib, check, no_check=args.no_check, cooldown=not args.no_cooldown),
                                     "heavy_tailed_wavelet": heavy_tail_leg(N, C, fs, freqs, dev, lib, check,
                                                                            no_check=args.no_check),
                                     "config5": config5_leg(args)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fs, freqs)
        emit(line)
    comm.close()
    return 0


def check_config5(plan, xbuf, ring_buf, base, distinct, fs, freqs, N, S, group, segs):
    """After the timed steps of --config 5: one (channel group, time block) is computed again
    into a ring buffer and rows channels {0, group-1} x scales {0, S/2, S-1 = 1 Hz} are compared
    with the oracle over that block's window (transforms.py:529-597 is the streaming design,
    :187-204 the numbers).  The block in the middle of the recording: both its edges are seams."""
    import ctypes
    i_mid = len(segs) // 2
    a, b, _ = segs[i_mid]
    xg = ctypes.c_void_p(xbuf.ptr.value)              # channel group 0
    plan.execute_block_device(xg, ring_buf, a, b - a, reuse_means=False)
    scales = sorted({0, S // 2, S - 1})
    worst = 0.0
    for c in sorted({0, group - 1}):
        ref = oracle_rows_window(base[c % distinct], fs, freqs[scales], a, b)
        for i, sc in enumerate(scales):
            row = ring_buf.download((b - a,), np.float32, offset_bytes=4 * (c * S + sc) * (b - a))
            worst = max(worst, float(np.abs(row - ref[i]).max() / np.abs(ref[i]).max()))
    return bool(worst <= 1e-5), {
        "rows": "time block %d of %d (samples %d..%d), channels {0, %d} x scales {0, %d, %d} vs the oracle"
                % (i_mid, len(segs), a, b, group - 1, S // 2, S - 1),
        "worst_rel_err": float("%.3g" % worst)}


def config1_leg(fs, dev, lib, check, no_check=False, steps=200, warmup=10):
    """BASELINE.json config 1 -- 1 channel x 16 384 samples x 32 scales (6 voices per octave, 200 Hz down), the
    reference's own CPU-sized case -- as a latency: device-resident execute of a prebuilt plan, and the public call
    end to end (plan lookup, host array in, float64 result out)."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp_channel
    from ghost_amd.wave import ContinuousWaveletTransform
    N = 16384
    f1 = 200.0 / 2.0 ** (np.arange(32) / 6.0)
    x = lfp_channel(N, fs, channel=0, seed=99)
    plan = CwtPlan(N, 1, fs, f1, output="amplitude", device=dev)
    plan.upload()
    xb, ob = DeviceBuffer(4 * N), DeviceBuffer(plan.info["out_bytes"])
    xb.upload(x)
    for _ in range(warmup):
        plan.execute_device(xb, ob)
    check(lib.gcwt_device_synchronize())
    wall = []
    for _ in range(steps):
        t0 = time.perf_counter()
        plan.execute_device(xb, ob)
        wall.append(time.perf_counter() - t0)
    el = float(np.median(wall))
    res = {"workload": "1 ch x %d samples @ 1 kHz x 32 Morse scales 200..5.6 Hz, amplitude f32" % N,
           "device_resident": {"us_per_call": round(el * 1e6, 1), "value": round(N / el / 1e6, 2), "unit": "Msamples/s",
                               "steps": steps, "warmup": warmup, "statistic": "median"}}
    if not no_check:
        from oracle import ghost_oracle as orc
        ref = orc.cwt_amplitude(x.astype(np.float64), fs, f1)
        got = ob.download((32, N), np.float32)
        worst = float((np.abs(got - ref).max(axis=1) / ref.max(axis=1)).max())
        res["checked"], res["worst_rel_err"] = bool(worst <= 1e-5), float("%.3g" % worst)
        res["check"] = "all 32 rows vs the oracle"
    xb.free(); ob.free(); plan.close()
    cwt = ContinuousWaveletTransform()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        cwt.transform(x, fs=fs, freqs=f1[::-1].copy())
        ts.append(time.perf_counter() - t0)
    res["transform_end_to_end"] = {"us_per_call": round(float(np.median(ts[2:])) * 1e6, 1), "first_call_ms": round(ts[0] * 1e3, 2),
                                   "note": "host in, float64 host out, plan cached after the first call"}
    return res


def heavy_tail_leg(N, C, fs, freqs, dev, lib, check, no_check=False, steps=5, warmup=2, gamma=3.0, beta=2.0):
    """The headline shape with Morse(3, 2): a wavelet whose truncated kernels answer at every frequency, so that no
    scale takes the decimated path -- time domain up to 48 taps, block convolution (overlap-save over 4096-sample
    blocks) above.  Round 3's review asked for 4 x the default wavelet's step or better."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp
    S, distinct = len(freqs), 8
    plan = CwtPlan(N, C, fs, freqs, gamma=gamma, beta=beta, output="amplitude", device=dev)
    plan.upload()
    plan.set_profiling(True)
    base = lfp(distinct, N, fs, seed=1234)
    xb, ob = DeviceBuffer(4 * C * N), DeviceBuffer(plan.info["out_bytes"])
    for c in range(C):
        xb.upload(base[c % distinct], offset_bytes=4 * c * N)
    for _ in range(warmup):
        plan.execute_device(xb, ob)
    check(lib.gcwt_device_synchronize())
    wall, stages = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        plan.execute_device(xb, ob)
        wall.append(time.perf_counter() - t0)
        stages.append(plan.timings())
    el, info = float(np.median(wall)), plan.info
    res = {"workload": "%d ch x %d samples @ 1 kHz x %d Morse(%g, %g) scales 200..2 Hz, amplitude f32" % (C, N, S, gamma, beta),
           "ms_per_step": round(el * 1e3, 3), "value": round(C * N / el / 1e6, 2), "unit": "Msamples/s",
           "steps": steps, "warmup": warmup, "statistic": "median",
           "scales": {"decimated": info["n_spectral"], "time_domain": info["n_direct"],
                      "block_convolution": info["n_blockconv"], "full_band": info["n_fullband"]},
           "stage_ms": {k: round(float(np.median([t[k] for t in stages])), 3)
                        for k in ("direct_ms", "blockconv_ms", "fullband_ms", "synth_ms", "total_ms")}}
    if not no_check:
        from oracle import ghost_oracle as orc
        rows = [0, S // 2, S - 1]
        worst = 0.0
        for c in (0, C - 1):
            ref = orc.cwt_amplitude(base[c % distinct].astype(np.float64), fs, freqs[rows], gamma=gamma, beta=beta)
            for i, sc in enumerate(rows):
                row = ob.download((N,), np.float32, offset_bytes=4 * (c * S + sc) * N)
                worst = max(worst, float(np.abs(row - ref[i]).max() / ref[i].max()))
        res["checked"], res["worst_rel_err"] = bool(worst <= 1e-5), float("%.3g" % worst)
        res["check"] = "channels {0, %d} x scales {0, %d, %d} vs the oracle" % (C - 1, S // 2, S - 1)
    xb.free(); ob.free(); plan.close()
    return res


def config2_leg(fs, freqs, dev, lib, check, no_check=False, steps=20, warmup=3, cooldown=True):
    """BASELINE.json config 2: 1 channel x 1e6 samples x 100 scales 200..2 Hz on one GPU.  (a) device-resident
    execute, plan prebuilt (the metric's definition: SURVEY.md 8d); (b) the public call
    ContinuousWaveletTransform.transform() end to end -- host array in, host result out over PCIe, the
    reference's float64 result and dtype=float32 -- which is what a user of the reference's class sees."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    from ghost_amd.synthetic import lfp_channel
    from ghost_amd.wave import ContinuousWaveletTransform
    N, S = 1000000, len(freqs)
    x = lfp_channel(N, fs, channel=0, seed=4321)
    plan = CwtPlan(N, 1, fs, freqs, output="amplitude", device=dev)
    plan.upload()
    plan.set_profiling(True)
    xb, ob = DeviceBuffer(4 * N), DeviceBuffer(plan.info["out_bytes"])
    xb.upload(x)
    for _ in range(warmup):
        plan.execute_device(xb, ob)
    check(lib.gcwt_device_synchronize())
    wall, dev_ms, synth = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        plan.execute_device(xb, ob)
        wall.append(time.perf_counter() - t0)
        tm = plan.timings()
        dev_ms.append(tm["total_ms"])
        synth.append(tm["synth_ms"])
    alg = N * (4 + 4 * S)
    el, k = float(np.median(wall)), float(np.median(synth))
    res = {"workload": "1 ch x %d samples @ 1 kHz x %d Morse scales 200..2 Hz, amplitude f32" % (N, S),
           "device_resident": {"ms_per_step": round(el * 1e3, 4), "device_ms": round(float(np.median(dev_ms)), 4),
                               "value": round(N / el / 1e6, 2), "unit": "Msamples/s", "steps": steps, "warmup": warmup,
                               "statistic": "median"},
           "roofline": {"bound": "hbm", "kernel": "synthesis (k_synthi + k_synth7)", "kernel_ms": round(k, 4),
                        "algorithmic_bytes": alg, "achieved": round(alg / (k * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "one channel is 404 MB of rows: a launch of ~0.3 ms that does not fill the chip for long"}}
    if not no_check:
        from oracle import ghost_oracle as orc
        rows = [0, S // 2, S - 1]
        ref = orc.cwt_amplitude(x.astype(np.float64), fs, freqs[rows])
        worst = 0.0
        for i, sc in enumerate(rows):
            row = ob.download((N,), np.float32, offset_bytes=4 * sc * N)
            worst = max(worst, float(np.abs(row - ref[i]).max() / ref[i].max()))
        res["checked"], res["worst_rel_err"] = bool(worst <= 1e-5), float("%.3g" % worst)
    xb.free(); ob.free(); plan.close()
    # The link is half as fast for some twenty seconds after the card has run at its power limit (measured: a 400 MB
    # float32 result 7.6 ms before the headline steps, 13.9 ms right after them, 7.6 ms again after 20 s of idling): the
    # public call's leg waits for it to come back, up to 30 s, and says what the link gave before and after the wait.
    api = {}
    import ctypes as C
    from ghost_amd import hostmem
    probe_dst = hostmem.empty((64 << 20,), np.float32)
    probe_src = DeviceBuffer(256 << 20)

    def link_gbps():
        if probe_dst is None:
            return None
        t0 = time.perf_counter()
        check(lib.gcwt_rows_to_host(probe_src.ptr, 64 << 20, 1, 64 << 20, probe_dst.ctypes.data_as(C.c_void_p), 64 << 20, 16))
        return round(0.268435456 / (time.perf_counter() - t0), 1)
    link_gbps()
    link_before = link_gbps()
    waited = 0.0
    link_after = link_before
    while cooldown and link_after is not None and link_after < 45.0 and waited < 30.0:
        time.sleep(2.0)
        waited += 2.0
        link_after = link_gbps()
    probe_src.free()
    del probe_dst
    for name, kw in (("float64", {}), ("float32", {"dtype": np.float32})):
        cwt = ContinuousWaveletTransform()
        ts, t_call, t_slice = [], [], []
        for _ in range(5):
            t0 = time.perf_counter()
            cwt.transform(x, fs=fs, freqs=freqs[::-1].copy(), **kw)
            t1 = time.perf_counter()
            amp = cwt.amplitude                  # the whole result on the host, as the reference leaves it
            ts.append(time.perf_counter() - t0)
            t_call.append(t1 - t0)
            nbytes = int(amp.nbytes)
            del amp
            # what a caller who looks at part of the result pays: transform() + one second of every scale
            t0 = time.perf_counter()
            cwt.transform(x, fs=fs, freqs=freqs[::-1].copy(), **kw)
            piece = cwt.fetch(start=500000, stop=501000)
            t_slice.append(time.perf_counter() - t0)
            assert piece.shape == (S, 1000)
        api[name] = {"ms_per_call": round(float(np.median(ts[1:])) * 1e3, 2), "first_call_ms": round(ts[0] * 1e3, 2),
                     "value": round(N / float(np.median(ts[1:])) / 1e6, 2), "unit": "Msamples/s",
                     "result_bytes": nbytes,
                     "transform_returns_ms": round(float(np.median(t_call[1:])) * 1e3, 2),
                     "transform_plus_1s_slice_ms": round(float(np.median(t_slice[1:])) * 1e3, 2)}
        del cwt
    api["link"] = {"gbps_after_the_timed_steps": link_before, "waited_s": waited, "gbps_when_measured": link_after}
    res["transform_end_to_end"] = dict(api, note="host array in, transform() and then the whole `amplitude` on the host (page-locked result from the pool of "
                                            "ghost_amd.hostmem, float64 sent as float32 and widened by host threads as it lands; `link`: what a 256 MB copy gave right after the timed steps and when this leg ran -- the card's link is half as fast for some 20 s after a run at the power limit); transform_returns_ms: the call alone (rows left on the "
                                            "device); transform_plus_1s_slice_ms: transform() + fetch() of 1000 samples of every scale.  PCIe-inclusive: never the headline value")
    return res


def config5_leg(args):
    """BASELINE.json config 5 per GPU (48 of its 384 channels), 3 timed steps and the config-5 check, as a
    child process that runs after this one has released the device: its JSON line, trimmed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "5", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-ceilings", "--sustain", "0"]
    if args.no_check:
        cmd.append("--no-check")
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"error": "exit %d: %s" % (out.returncode, out.stderr[-300:])}
        j = json.loads(lines[-1])
    except Exception as e:                                  # the headline line must not depend on this leg
        return {"error": repr(e)[:300]}
    keep = {k: j.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "checked", "check",
                                  "whole_job_frac_of_hbm_peak", "stages_ms")}
    keep["workload"] = j["config"]["workload"]
    keep["roofline"] = {k: j["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                          "kernel_ms", "launches_per_step", "algorithmic_bytes", "kernels")}
    keep["wall_s"] = round(time.perf_counter() - t0, 1)
    return keep


def other_mode(output, N, C, fs, freqs, S, xbuf, dev, lib, check, steps=10, warmup=2, check_against=None):
    """The same workload with another output mode (device-resident, plan prebuilt): `warmup`
    untimed executions -- the fresh result buffer's pages are touched for the first time there --
    then `steps` executions timed one by one: minimum and median."""
    from ghost_amd.engine import CwtPlan, DeviceBuffer
    plan = CwtPlan(N, C, fs, freqs, output=output, device=dev)
    plan.upload()
    plan.set_profiling(True)
    obuf = DeviceBuffer(plan.info["out_bytes"])
    for _ in range(warmup):
        plan.execute_device(xbuf, obuf)
    check(lib.gcwt_device_synchronize())
    synth, wall = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        plan.execute_device(xbuf, obuf)               # returns after the stream has drained
        wall.append(time.perf_counter() - t0)
        synth.append(plan.timings()["synth_ms"])
    b_out = 8 if output == "complex" else 4
    alg = C * N * (4 + S * b_out)
    el, el_min = float(np.median(wall)), min(wall)
    k_med, k_min = float(np.median(synth)), min(synth)
    res = {"ms_per_step": round(el * 1e3, 4), "ms_per_step_min": round(el_min * 1e3, 4),
           "value": round(C * N / el / 1e6, 2), "value_best": round(C * N / el_min / 1e6, 2), "unit": "Msamples/s",
           "steps": steps, "warmup": warmup, "statistic": "median of the timed steps (and the best one)",
           "algorithmic_bytes": int(alg),
           "whole_job_frac_of_hbm_peak": round(alg / el / 1e9 / HBM_PEAK_GBS, 4),
           "kernel_ms": round(k_med, 4), "kernel_ms_min": round(k_min, 4),
           "kernel_frac_of_hbm_peak": round(alg / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if check_against is not None:
        base, distinct = check_against
        ok, worst = spot_check(obuf, base, fs, freqs, C, N, distinct, output=output, dec=plan.scale_info()["decimation"])
        full = full_output_check(obuf, C, N, S, distinct, output=output)
        res["checked"], res["worst_rel_err"], res["full_output"] = bool(ok and full["ok"]), float("%.3g" % worst), full
    obuf.free()
    plan.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=[3, 5])
    ap.add_argument("--channels", type=int, default=0, help="channels per GPU (default 128; 48 for config 5)")
    ap.add_argument("--samples", type=int, default=0)
    ap.add_argument("--scales", type=int, default=0)
    ap.add_argument("--output", default="amplitude", choices=["amplitude", "power", "complex"])
    ap.add_argument("--shard", default="channels", choices=["channels", "time"],
                    help="how N > 1 ranks split the job: contiguous channel blocks (weak scaling, the default) or, "
                         "with --config 5, the time blocks of one recording that every rank holds whole (strong scaling)")
    ap.add_argument("--group", type=int, default=0, help="config 5: channels per plan execution (default 24)")
    ap.add_argument("--max-fft-log2", type=int, default=0, help="longest FFT of the plan (time-block size), 0 = library default")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="seconds the step is repeated after the timed region to read sustained power / sclk (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-2 and config-5 legs after the timed steps")
    ap.add_argument("--no-cooldown", action="store_true", help="config-2 public-call leg: do not wait for the link to recover after the timed steps")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the complex-output measurement after the timed steps")
    ap.add_argument("--dry-run", default="", metavar="DIR",
                    help="launcher rehearsal: every rank writes DIR/rank<r>.json and exits (no GPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
