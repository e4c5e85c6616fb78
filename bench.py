#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the continuous wavelet transform,
BASELINE.json config 3 -- 128 channels x 1e6 samples @ 1 kHz x 100 Morse scales
(log-spaced 200..2 Hz), amplitude output -- per GPU, device-resident.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by ``python -m torch.distributed.run`` (one rank per GPU); each
rank transforms its own 128-channel block (weak scaling, no data-path
collective); rank 0's filter bank is broadcast once with RCCL.  Rank 0 prints one
JSON line.  A "step" is one gcwt_execute over the rank's block with input and
output resident in HBM.  No torch anywhere in this file.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); measured copy 6290


def cpu_baseline(fs, n_samples, freqs, budget_s=25.0):
    """The oracle's literal path (port of transforms.py:187-224 + convolution.py:16-87)
    on this box's cores: config 2 shape, ThreadPool over scales like parallel=True."""
    from oracle import ghost_oracle as orc
    from ghost_amd.synthetic import lfp_channel
    cores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores
    x = lfp_channel(n_samples, fs, 0).astype(np.float64)
    best, reps = None, 0
    t_start = time.time()
    while reps < 3 or (time.time() - t_start < 10.0 and reps < 60):   # ~10 s of CPU work
        t0 = time.time()
        orc.cwt_amplitude(x, fs, freqs, n_threads=cores)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
        reps += 1
        if time.time() - t_start > budget_s:
            break
    return {"value": round(n_samples / best / 1e6, 4), "unit": "Msamples/s", "cores": cores,
            "kind": "port",
            "sample": "1 ch x %d samples x %d scales (config 2), float64, scipy.fft overlap-add, "
                      "ThreadPool(%d) over scales, best of %d runs (%.2f s each)" %
                      (n_samples, len(freqs), cores, reps, best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=128, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--scales", type=int, default=100)
    ap.add_argument("--output", default="amplitude", choices=["amplitude", "power", "complex"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from ghost_amd.dist import Comm, env_rank, shard_channels
    from ghost_amd.engine import CwtPlan, DeviceBuffer, device_count, device_name
    from ghost_amd._lib import lib, check
    from ghost_amd.synthetic import lfp

    rank, world, local = env_rank()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    ndev = device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs an AMD GPU: libghostcwt has no CPU path")
    dev = local % ndev
    check(lib.gcwt_set_device(dev))

    fs = 1000.0
    C, N, S = args.channels, args.samples, args.scales
    freqs = np.geomspace(200.0, 2.0, S)
    total_channels = C * world                       # weak scaling: 128 channels per GPU
    c0, c1 = shard_channels(total_channels, rank, world)
    assert c1 - c0 == C

    comm = Comm(rank, world, device=dev)
    t_plan = time.perf_counter()
    plan = CwtPlan(N, C, fs, freqs, output=args.output, device=dev)
    plan.upload()                                    # workspace, filter bank, FFT tables
    check(lib.gcwt_device_synchronize())
    plan_ms = (time.perf_counter() - t_plan) * 1e3   # reported apart from the timed steps
    bank_via = comm.broadcast_bank(plan, root=0)
    plan.set_profiling(True)

    # synthetic 1 kHz LFP: 8 distinct generated channels per rank, tiled over the block
    distinct = min(C, 8)
    base = lfp(distinct, N, fs, seed=1234 + 1000 * rank)
    xbuf = DeviceBuffer(4 * C * N)
    for c in range(C):
        xbuf.upload(base[c % distinct], offset_bytes=4 * c * N)
    obuf = DeviceBuffer(plan.info["out_bytes"])

    for _ in range(args.warmup):
        plan.execute_device(xbuf, obuf)
    check(lib.gcwt_device_synchronize())
    comm.barrier()
    synth_ms, stage_ms = [], None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.execute_device(xbuf, obuf)              # returns after the stream has drained
        tm = plan.timings()
        synth_ms.append(tm["synth_ms"] / max(1, tm["synth_launches"]))
        stage_ms = tm
    check(lib.gcwt_device_synchronize())
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.allreduce_max(elapsed)

    if rank == 0:
        b_out = 8 if args.output == "complex" else 4
        units = world * C * N * args.steps
        value = units / elapsed / 1e6
        k_ms = float(np.mean(synth_ms))
        alg_bytes = C * N * (4 + S * b_out)          # per channel-sample 4 + S*b_out
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (C, N, S, args.output) == (128, 1000000, 100, "amplitude"):
            traffic = json.load(open(tpath)).get("k_synth_hbm_bytes_per_launch")
        line = {
            "metric": "Msamples/s CWT (128ch x 1e6 samp x 100 scales); % HBM roofline; 1/2/4/8 GPU",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic 1 kHz LFP (pink noise + 8 Hz rhythm + 40 Hz bursts), "
                    "%d generated channels tiled over each GPU's block" % distinct,
            "config": {"workload": "%d ch/GPU x %d samples @ 1 kHz x %d Morse scales 200..2 Hz, "
                                   "%s f32 out, device-resident" % (C, N, S, args.output),
                       "channels_total": total_channels, "parallelism": "channel-sharded x%d" % world,
                       "bank": bank_via, "comm": comm.backend, "device": device_name(dev),
                       "plan_create_ms": round(plan_ms, 2)},
            "roofline": {"bound": "hbm", "kernel": "k_synth7", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel_ms": round(k_ms, 4),
                         "algorithmic_bytes": alg_bytes,
                         "limited_by": "package power (1.3 of 1.4 kW, sclk ~1.9 GHz with the HBM "
                                       "writes on: DESIGN.md 5, profiles/r01_power.txt)"},
            "stages_ms": {k: round(float(v), 4) for k, v in stage_ms.items() if k.endswith("_ms")},
            "whole_job_frac_of_hbm_peak": round(world and alg_bytes * args.steps / elapsed / 1e9
                                                / HBM_PEAK_GBS, 4),
        }
        if comm.rccl_error:
            line["config"]["rccl_error"] = comm.rccl_error[:200]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(fs, N, freqs)
        print(json.dumps(line), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
