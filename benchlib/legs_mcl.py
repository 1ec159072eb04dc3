"""MCL legs: the headline (configs[1], 1e6 x 32), the multinomial twin, configs[4], the sharded legs and their last-resort replicas line."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

from .common import BENCH_PY, ROOT  # noqa: F401
from .common import DEVICE_WARMUP_MCL, EXTRA_WARMUP, FP64_VALU_PEAK, HBM_PEAK, K1_BYTES, free_port, log, make_scene, mcl_instruction_budget, measured_traffic
from .cpu_baselines import cpu_baseline, index_parity


def replicas_fallback(ctx, n, L, K, W, obs_list, scheme, lik, reason):
    import torch

    import rust_robotics_amd.localization as loc

    dist = ctx.dist
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
    pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1 + ctx.rank, device=ctx.local_rank,
                                                    resample_scheme=scheme, likelihood_mode=lik)
    u = [1.0, 0.1]

    def fence():
        pf.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        pf.synchronize()
        torch.cuda.synchronize()

    W0 = max(len(obs_list) - 2 * K, W)  # W + 76: host-runtime warm-up of a process that has torch's HIP context loaded
    for t in range(W0):
        pf.step_async(u, obs_list[t])
    fence()
    t0 = time.perf_counter()
    for t in range(W0, W0 + K):
        pf.step_async(u, obs_list[t])
    fence()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    pf.profile_enable(1)
    pf.profile_reset()
    t1 = time.perf_counter()
    for t in range(W0 + K, W0 + 2 * K):
        pf.step_async(u, obs_list[t])
    pf.synchronize()
    dt_instr = time.perf_counter() - t1
    prof = pf.profile_read()
    pf.profile_enable(0)
    est = pf.estimate()
    dist.barrier()
    return dict(seconds=float(tmax.item()), seconds_instrumented=dt_instr, kernels=prof, estimate=[float(a) for a in est],
                dominant=None, transport="NONE -- independent replicas, no exchange",
                transport_note="SHARDING FAILED: " + reason, p2p_timed_out=False, migrated_particles_last_step=0)


def leg_mcl(args, ctx, n, L, K, W, with_cpu, breakdown=True, label="configs[1]"):
    """Fixed-N MCL: n particles per GPU x L landmarks, propagate + weight + resample every step."""
    world = ctx.world
    # time only moves forward for every filter: W warm-up + K timed + K estimate-every-step + K dispatch-stamped + K breakdown
    # steps; the sharded legs also validate (12 steps) and warm up 64 steps longer
    # (the sharded legs warm up EXTRA_WARMUP steps + settle blocks inside bench_sharded; ~50 ms of work is what the device needs)
    D = 0 if ctx.sharded else (DEVICE_WARMUP_MCL if n <= 2_000_000 else 60)
    obs_list = make_scene(L, D + W + 5 * K + 8 + (EXTRA_WARMUP if ctx.sharded else 0), seed=1)
    scheme = 1 if args.scheme == "systematic" else 0
    lik = 0 if args.likelihood == "fused" else 1
    extra = {}
    if ctx.sharded:
        from rust_robotics_amd import sharded

        try:
            res = sharded.bench_sharded(ctx.rank, world, ctx.local_rank, n, L, K, W, obs_list[:W + 2 * K + EXTRA_WARMUP], scheme, lik, args.transport)
        except RuntimeError as e:
            # no sharded transport works on this machine.  Last resort so that the run still leaves a line:
            # every rank steps its own, independent filter (NO exchange, NOT one sharded filter) and the line
            # says so in config.sharding -- the number is an upper bound for the sharded step, not a measurement of it.
            res = replicas_fallback(ctx, n, L, K, W, obs_list[:W + 2 * K + EXTRA_WARMUP], scheme, lik, str(e))
        extra["headline_step"] = (
            "sharded rr_pf_shard_step_p2p / rr_pf_shard_step + rr_pf_shard_want_estimate: propagate + weight + global resample, every shard leaving its part of the mean "
            "try_step returns every step (the sums over the sources of its own slots, added up by the kernel that moves the particles; one all-reduce of "
            "four doubles when the value is read) -- the counterpart of the N = 1 line's `value`"
            if res.get("estimate_every_step") else
            "sharded rr_pf_shard_step: propagate + weight + global resample -- the PLAIN step (the mean is formed when an accessor asks: local moments + "
            "one all-reduce); the N = 1 line's `plain_async_step` is its single-GPU counterpart")
    else:
        import rust_robotics_amd.localization as loc

        cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)
        pf = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=ctx.local_rank,
                                                        resample_scheme=scheme, likelihood_mode=lik)
        u = [1.0, 0.1]
        # The reference's try_step returns the refreshed mean EVERY step (particle_filter.rs:299,332,343,496), so the headline
        # step is the one that produces it: rr_pf_step_async_estimate -- the mean of the resampled set accumulated inside the
        # step's own plan kernel and kept on the device (one synchronisation at the end of the K steps).  The multinomial
        # scheme's estimate is the deferred form: the resampled set's mean is summed by the kernel that draws, searches and
        # gathers the sources -- the next step's k_step_lazy (the last step's by rr_pf_last_step_estimate's gather).
        with_est = n <= 8_388_608  # (the in-step estimate's limit, rr_pf.h)
        step_fn = pf.step_async_estimate if with_est else pf.step_async
        if D and getattr(args, "cold_first", False):
            # COLD: the same W + K steps with nothing but the command line's warm-up before them -- the first work this process
            # gives the device (a filter of its own, so that the hot measurement below starts from the same state as ever)
            # `ms_per_step_cold_unwarmed`: exactly that.  `ms_per_step_cold`: a caller that follows include/rr_pf.h -- rr_pf_warm right
            # after create (50 ms of step-shaped work on the filter's stream, round 6), then only the command line's W warm-up steps
            for key, warm in (("ms_per_step_cold_unwarmed", False), ("ms_per_step_cold", True)):
                pc = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=ctx.local_rank,
                                                                resample_scheme=scheme, likelihood_mode=lik)
                fn_c = pc.step_async_estimate if with_est else pc.step_async
                if warm:
                    pc.warm()
                for t in range(W):
                    fn_c(u, obs_list[t])
                pc.synchronize()
                t0 = time.perf_counter()
                for t in range(W, W + K):
                    fn_c(u, obs_list[t])
                pc.synchronize()
                extra[key] = (time.perf_counter() - t0) / K * 1e3
                del fn_c, pc  # (the bound method holds the filter too: both, or it lives on beside the hot one)
                if not warm:
                    time.sleep(0.3)  # let the clocks fall again: the warmed measurement must not inherit the unwarmed one's 25 steps
        for t in range(D):  # device warm-up (see DEVICE_WARMUP_MCL), then time moves on
            step_fn(u, obs_list[t])
            # in blocks with a synchronisation in between, the shape of the timed region: a thousand steps enqueued in one go leave
            # the host ~40 ms ahead of the device, and one run in five then paid a ~0.45 ms stall of the runtime somewhere in the
            # 20 steps that follow (measured with the driver's --steps 20: 75 instead of 53 us/step); in blocks: none in 300 blocks
            if (t + 1) % 50 == 0:
                pf.synchronize()
        obs_list = obs_list[D:]
        for t in range(W):
            step_fn(u, obs_list[t])
        pf.synchronize()
        t0 = time.perf_counter()
        for t in range(W, W + K):
            step_fn(u, obs_list[t])
        pf.synchronize()
        dt = time.perf_counter() - t0
        extra["headline_step"] = ("rr_pf_step_async_estimate: propagate + weight + resample + the mean try_step returns, every step"
                                  + (" (summed by the next step's kernel as it moves the particles; RR_PF_EST_DEFER=0: inside the plan kernel)"
                                     if args.scheme == "systematic" else " (summed by the next step's draw-and-gather kernel)")
                                  if with_est else
                                  "rr_pf_step_async: propagate + weight + resample (the in-step estimate serves up to 8 388 608 particles)")
        if with_est:
            extra["last_step_estimate"] = [float(a) for a in pf.last_step_estimate()]
        est = pf.estimate()
        # the same K steps again WITHOUT the per-step estimate (what a node that publishes every k-th estimate runs)
        if with_est:
            for t in range(W + K, W + K + min(W, 10)):
                pf.step_async(u, obs_list[t])
            pf.synchronize()
            t1 = time.perf_counter()
            for t in range(W + K + min(W, 10), W + 2 * K):
                pf.step_async(u, obs_list[t])
            pf.synchronize()
            t_plain = (time.perf_counter() - t1) / max(K - min(W, 10), 1)
            extra["plain_async_step"] = {"ms_per_step": t_plain * 1e3, "value": float(n) * L / t_plain,
                                         "note": "rr_pf_step_async: the step without the per-step estimate"}
        # roofline kernel: the NEXT K steps (the filter resamples every step, so the work per step is stationary) in
        # which ONLY the propagate+weight kernel is timed, by the start/stop timestamps of its own dispatch packets
        # (hipExtLaunchKernelGGL on the filter's stream): no event packets in the stream, the kernel runs as in the
        # timed loop.  Kept out of the timed region because the stamped launch costs the host ~3 us per step.
        pf.profile_enable(2)
        pf.profile_reset()
        for t in range(W + 2 * K, W + 3 * K):
            step_fn(u, obs_list[t])
        pf.synchronize()
        dominant = pf.profile_read()["k_propagate_weight"]
        # per-kernel breakdown: an instrumented re-run of K steps with HIP events around every launch (adds
        # ~3 us per launch; informational, kept out of `value` and of `roofline`)
        prof, dt_instr = {"k_propagate_weight": dominant}, 0.0
        if breakdown:
            pf.profile_enable(1)
            pf.profile_reset()
            t1 = time.perf_counter()
            for t in range(W + 3 * K, W + 4 * K):
                step_fn(u, obs_list[t])
            pf.synchronize()
            dt_instr = time.perf_counter() - t1
            prof = pf.profile_read()
        pf.profile_enable(0)
        # the SYNCHRONOUS try_step (rr_pf_step: the estimate comes back to the host every step -- what the reference's callers do,
        # particle_filter.rs:488-497 / monte_carlo_localization.rs:291-300), over the same K steps' worth of inputs, outside `value`
        if n <= 4_000_000:
            for t in range(5):
                pf.step(u, obs_list[W + 4 * K + t])
            per = []
            for t in range(K):  # (every step is a host round trip of its own, so each one is timed by itself)
                t1 = time.perf_counter()
                pf.step(u, obs_list[W + 4 * K + 5 + t])
                per.append(time.perf_counter() - t1)
            t_sync = float(np.mean(per))
            extra["synchronous_try_step"] = {"ms_per_step": t_sync * 1e3, "value": float(n) * L / t_sync,
                                             "median_ms": float(np.median(per)) * 1e3, "max_ms": float(np.max(per)) * 1e3,
                                             "note": "rr_pf_step: one host round trip per step, the mean of the resampled set returned every step" +
                                                     ("" if with_est else " (multinomial: the pending draws are searched, gathered and averaged by extra launches)") +
                                                     "; mean of the K steps (median_ms / max_ms beside it: the HIP runtime stalls ONCE for ~0.45 ms at some launch "
                                                     "count of a process -- tools/stall_probe.py: one step of 600 --, and with K = 20 that one step is 22 us of the mean "
                                                     "when it falls into this window)"}
        if with_cpu and n <= 4_000_000:  # (checker use of the oracle: part of the cpu_baseline leg)
            pf2 = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=ctx.local_rank,
                                                             resample_scheme=scheme, likelihood_mode=lik, record_indices=True)
            for t in range(3):
                pf2.step_async(u, obs_list[t])
            extra["index_parity"] = index_parity(pf2, n, L, args.scheme, obs_list[3])
            del pf2
        del pf
        res = dict(seconds=dt, seconds_instrumented=dt_instr, kernels=prof, estimate=[float(a) for a in est], dominant=dominant)

    if ctx.rank != 0:
        return None
    total_updates = float(n) * world * L * K
    value = total_updates / res["seconds"]
    kern = res["kernels"]
    dominant = res.get("dominant")
    if dominant and not dominant[0]:
        dominant = None  # this path does not stamp its dispatches (multinomial): fall back to the instrumented re-run
    k1_n, k1_ms = dominant or kern["k_propagate_weight"]
    k1_avg_s = (k1_ms / max(k1_n, 1)) * 1e-3
    # the systematic headline's step kernel is the EST build when the estimate is deferred (the default): it also reads the sources' v
    est_build = (not ctx.sharded) and args.scheme == "systematic" and n <= 8_388_608 and os.environ.get("RR_PF_EST_DEFER", "1") != "0"
    k1_bytes = (K1_BYTES[args.scheme] + (8.0 if est_build else 0.0)) if not ctx.sharded else 64.0
    achieved = k1_bytes * n / k1_avg_s if k1_avg_s > 0 else 0.0
    step_kernel_ms = {k: (v[1] / max(v[0], 1)) for k, v in kern.items() if v[0]}
    traffic, traffic_src = measured_traffic("k_step_lazy", getattr(args, "traffic_key", "mcl" if (n, L, args.scheme) == (1_000_000, 32, "systematic") else
                                                                   f"mcl_{n}x{L}_{args.scheme}"),
                                              est=est_build if args.scheme == "systematic" else None)
    # FP64-VALU side of the same kernel: f64-rate lane-instructions per particle (a per-pair count
    # times L plus a per-particle count, both read off the ISA and checked against SQ_INSTS_VALU) over the kernel time
    pair_i, part_i = mcl_instruction_budget(est=est_build)
    valu_rate = (pair_i * L + part_i) * n / k1_avg_s if k1_avg_s > 0 else 0.0
    # the multinomial kernel moves whole 128-byte lines for its 8-byte guide pairs and 32-byte source records (iid draws have no
    # locality): what binds it is the MEASURED line traffic (PMC), not the algorithmic bytes and not the FP64 pipe
    line_rate = (traffic / k1_avg_s) if (traffic and k1_avg_s > 0 and args.scheme == "multinomial") else 0.0
    fracs = {"fp64_valu": valu_rate / FP64_VALU_PEAK, "hbm": achieved / HBM_PEAK, "line_traffic": line_rate / HBM_PEAK}
    bound = max(fracs, key=fracs.get)
    out = {
        "metric": "particle-landmark updates/sec",
        "value": value,
        "unit": "particle-landmark updates/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": res["seconds"] / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"fixed-N MCL (BASELINE.json {label}): {n} particles/GPU x {L} landmarks, "
                        f"propagate+weight+{args.scheme} resample every step, likelihood={args.likelihood}",
            "particles_per_gpu": n,
            "landmarks": L,
            "resample": args.scheme,
            "sharding": "none" if not ctx.sharded else
                        (f"{res.get('transport')} ({res.get('transport_note')})" if str(res.get("transport", "")).startswith("NONE") else
                         f"contiguous particle blocks over {world} GPUs; transport {res.get('transport')} ({res.get('transport_note')})"),
        },
        "roofline": {
            # the binding resource of THIS kernel at THIS L: the FP64 vector pipe once its fraction of the issue peak exceeds the
            # HBM fraction (L >= ~16), HBM below that.  achieved / peak / frac stay the HBM figures the contract asks for;
            # binding_frac is the fraction of the binding resource's peak
            "bound": bound,
            "binding_frac": fracs[bound],
            "hbm_frac": achieved / HBM_PEAK,
            "kernel": ("k_step_lazy<EST> (propagate + weight + folded resample gather + the mean of the resampled set it moves)" if k1_bytes == 80.0 else
                       "k_step_lazy (propagate + weight + folded resample gather)" if k1_bytes == 72.0 else
                       "k_step_lazy<kSrcDraw> (multinomial draws + guide-table search + source gather + propagate + weight)"
                       if (args.scheme == "multinomial" and not ctx.sharded) else "k_propagate_weight"),
            "achieved": achieved / 1e9,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "traffic": traffic,
            "traffic_source": (traffic_src + (" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)" if traffic is not None else "")) if traffic_src else None,
            "avg_kernel_ms": k1_avg_s * 1e3,
            "timed_launches": k1_n,
            "timing": "dispatch timestamps (hipExtLaunchKernelGGL) of K launches that follow the timed region" if dominant else
                      "HIP events in an instrumented re-run of the K steps",
            "algorithmic_bytes_per_launch": k1_bytes * n,
            "fp64_valu": {"lane_instr_per_pair": pair_i, "lane_instr_per_particle": part_i, "achieved_lane_instr_per_s": valu_rate,
                          "peak_lane_instr_per_s": FP64_VALU_PEAK, "frac": valu_rate / FP64_VALU_PEAK},
            "line_traffic": ({"bytes_per_launch": traffic, "rate_GBps": line_rate / 1e9, "frac_of_hbm_peak": line_rate / HBM_PEAK,
                              "note": "PMC bytes per launch / kernel time: 10^6 random guide pairs and 10^6 random 32-byte records move a 128-byte line each "
                                      "(served by L2 and the Infinity Cache; priced against the 8 TB/s HBM peak); profiles/r04_multinomial_ab.md"}
                             if line_rate else None),
            "note": (("bound by cache-line traffic: see line_traffic.  " if bound == "line_traffic" else "") +
                     "FP64-VALU bound at L >= ~16 (fp64_valu.frac is the binding fraction).  " +
                     ("The working set (~90 B/particle) of 1e6 particles is Infinity-Cache resident, so `traffic` is fabric traffic, not DRAM traffic; "
                      if n <= 2_000_000 else "At this size the working set is several times the 256 MB Infinity Cache: `achieved` is a DRAM rate; ") +
                     "the HBM-bound workload of this line is the `fastslam` leg; profiles/r04_mcl_L_sweep.json shows where MCL turns from HBM- to VALU-bound"),
        },
        "kernel_ms_avg": step_kernel_ms,
        "device_warmup_steps": (EXTRA_WARMUP if ctx.sharded else D),
        "ms_per_step_instrumented": res.get("seconds_instrumented", 0.0) / K * 1e3,
        "estimate": res.get("estimate"),
    }
    out.update(extra)
    if with_cpu:
        n_cpu = getattr(args, "cpu_particles", None) or n
        out["cpu_baseline"] = cpu_baseline(n_cpu, L, obs_list, max_seconds=getattr(args, "cpu_seconds", 12.0), scheme=args.scheme,
                                           brief=getattr(args, "cpu_brief", False))
        if n_cpu != n:
            out["cpu_baseline"]["sample"] += f" -- a BOUNDED SAMPLE: {n_cpu} of the leg's {n} particles (the per-particle work and the serial scan both scale linearly)"
    if ctx.sharded:
        out["sharded"] = {k: res.get(k) for k in ("transport", "transport_note", "p2p_timed_out", "migrated_particles_last_step", "ranks_seen")}
        if world > 1 and not str(res.get("transport", "")).startswith("p2p"):
            # RCCL is the CORRECTNESS fallback (README): its step returns to the host between phases and tops out at
            # ~5x on 8 GPUs by world-1 arithmetic.  A line measured over it must not pass for the engine's scaling number.
            out["transport_fallback"] = (f"TIMED OVER {res.get('transport')}, NOT the peer-to-peer transport: {res.get('transport_note')} -- "
                                         "a correctness fallback (host round trips inside the step; ceiling ~5x at 8 GPUs), not the engine's scaling path")
            sys.stderr.write("\n" + "!" * 100 + "\nbench.py: " + out["transport_fallback"] + "\n" + "!" * 100 + "\n\n")
            if os.environ.get("RR_BENCH_REQUIRE_P2P"):
                raise SystemExit("RR_BENCH_REQUIRE_P2P is set and the peer-to-peer transport did not validate: " + str(res.get("transport_note")))
        seen = res.get("ranks_seen") or {}
        out["ranks_seen"] = seen.get("ranks")
        if seen and seen.get("distinct_devices", world) < world:
            # (RR_BENCH_SHARE_DEVICE: a rig that executes the N-rank code on fewer devices -- every rank's kernels run on the SAME GPU)
            out["shared_device"] = (f"{world} ranks on {seen['distinct_devices']} device(s): the {world}-rank code path executed, "
                                    f"NOT a scaling number -- `value` is the aggregate of ranks that time-share one GPU")
    return out


def leg_sharded_world1(args, n, L, K, W, transports=(("p2p", "p2p-only"), ("rccl", "rccl")), what=None, scheme=None):
    """The sharded MCL step with ONE rank, once per transport: the peer-to-peer transport (validated against the unsharded
    filter first, as in the multi-GPU run) and the native RCCL transport (a one-rank communicator: RCCL really called).
    Each in a process of its own (`bench.py --force-sharded --transport ...`): the sharded legs need torch.distributed, and
    torch's bundled HIP runtime has to be the first one a process loads."""
    out = {}
    for name, transport in transports:
        log(f"extra leg sharded_world1 / {name} ({n} x {L})")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR")}
        env["MASTER_PORT"] = str(free_port())
        cmd = [sys.executable, BENCH_PY, "--gpus", "1", "--force-sharded", "--transport", transport, "--no-extra-legs",
               "--no-cpu-baseline", "--steps", str(K), "--warmup", str(W), "--particles", str(n), "--landmarks", str(L), "--likelihood", args.likelihood]
        if scheme:
            cmd += ["--scheme", scheme]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {r.stderr[-400:]}")
            d = json.loads(lines[-1])
            out[name] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "sharding": d["config"]["sharding"],
                         "kernel_ms_avg": d.get("kernel_ms_avg"), "steps": d["steps"], "warmup": d["warmup"], "roofline": d.get("roofline")}
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    if what:
        out["workload"] = what
    out["note"] = ("world size 1 on this GPU: every exchange talks to itself, so this is the per-rank cost of the sharded step before any "
                   "cross-device latency (weak-scaling ceiling at 8 GPUs = 8 x plain_async_step / this)")
    return out
