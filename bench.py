#!/usr/bin/env python3
"""bench.py -- particle-landmark updates/s of the sampling-based localization hot path on N MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the driver launches it
under torch.distributed.run with one rank per GPU; started WITHOUT a launcher (``python bench.py --gpus 8``
from a bare shell) it re-executes itself under ``python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`` and forwards the ranks' output.  One "step" = one full pass of
the hot path over the whole particle set.  Prints ONE JSON line on rank 0.

What the line holds
  headline      BASELINE.json configs[1]: fixed-N MCL, 1 000 000 particles per GPU x 32 landmarks, propagate +
                weight + resample every step (weak scaling over N GPUs: contiguous particle blocks, RCCL / xGMI
                peer-to-peer exchange of the weight maximum, the integer sums and the migrating particles).
  "fastslam"    N = 1: BASELINE.json configs[2], FastSLAM 1.0, 100 000 particles x 200 landmarks (the HBM-bound
                workload) -- value, ms_per_step, roofline{}, cpu_baseline{} of its own, measured in the same run.
  "fastslam_sharded", "mcl_config5"
                N = 8 (or --all-legs): BASELINE.json configs[3] (1e6 x 200 over 8 GPUs = 125 000 per GPU) and
                configs[4] (1.6e7 x 64 over 8 GPUs = 2e6 per GPU).
``--workload fastslam|fastslam2`` prints that workload's line alone (rocprofv3 runs use this).

stdout carries that one line and nothing else: once the arguments are parsed, file descriptor 1 is handed to stderr (gloo
and RCCL print banners there from C++) and the line goes to a private copy of the real stdout.  Progress
(``[bench rank r +t s] ...``) goes to stderr.  ``RR_BENCH_DEADLINE_S`` (default 600; 0 = none): a stalled extra leg or a dead
rank costs at most this long -- rank 0 then prints the headline leg it already has, flagged ``deadline_exceeded``.

The legs live in benchlib/ (round 6: this file used to be one 1 600-line module); everything they define is re-exported here, so
`import bench; bench.leg_small_n(...)` keeps working for tools/ and tests/.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.common import _OUT, _TRANSPORT, BENCH_PY  # noqa: E402,F401
from benchlib.cpu_baselines import *  # noqa: E402,F401,F403
from benchlib.legs_fastslam import *  # noqa: E402,F401,F403
from benchlib.legs_mcl import *  # noqa: E402,F401,F403
from benchlib.legs_small import *  # noqa: E402,F401,F403
from benchlib.line import *  # noqa: E402,F401,F403
from benchlib.line import _compact_cpu, _compact_roofline, _leg_row, _num, _short  # noqa: E402,F401


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: re-execute under torch.distributed.run, one rank per GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: hipIpcGetMemHandle and RCCL need it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), BENCH_PY] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher detected (WORLD_SIZE unset); starting " + " ".join(cmd[1:8]) + " ...\n")
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["mcl", "fastslam", "fastslam2"], default="mcl")
    ap.add_argument("--particles", type=int, default=None, help="particles PER GPU (default 1e6 for mcl, 1e5 for fastslam)")
    ap.add_argument("--landmarks", type=int, default=None, help="default 32 for mcl, 200 for fastslam")
    ap.add_argument("--scheme", choices=["systematic", "multinomial"], default="systematic")
    ap.add_argument("--likelihood", choices=["fused", "product"], default="fused")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true",
                    help="skip the instrumented per-kernel re-run (use under rocprofv3 so that its averages cover the timed launches only)")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline workload only (no fastslam / configs[3] / configs[4] legs)")
    ap.add_argument("--all-legs", action="store_true", help="run the configs[3] / configs[4] legs at any --gpus (default: only at 8)")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded path even at --gpus 1")
    ap.add_argument("--no-sharded-world1", action="store_true", help="skip the world-size-1 sharded legs of the default line")
    ap.add_argument("--transport", choices=["auto", "p2p", "rccl", "torch", "p2p-only"], default="auto",
                    help="sharded exchange: auto = peer-to-peer over xGMI if it validates at run time, else native RCCL, else "
                         "torch.distributed NCCL; p2p / rccl / torch restrict the ladder; p2p-only validates against the unsharded filter")
    args = ap.parse_args()
    _TRANSPORT["value"] = args.transport
    # dmabuf IPC: hipIpcGetMemHandle and RCCL need it on this host driver.  Set before the first HIP / RCCL load of THIS
    # process, so that ranks started by somebody else's launcher (the driver's torchrun) get it too, not only self-launched ones.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    claim_stdout()
    start_deadline(int(os.environ.get("RANK", "0")))
    if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
        os.environ["NCCL_DEBUG"] = "WARN"  # keep RCCL's version banner off stdout
    ctx = Ctx(args)
    if ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or unset WORLD_SIZE and let bench.py launch itself)")
    ctx.init_group()
    require_devices(ctx)
    if ctx.sharded:
        import torch

        torch.cuda.set_device(ctx.local_rank)

    fast = args.workload in ("fastslam", "fastslam2")
    if args.particles is None:
        args.particles = 100_000 if fast else 1_000_000
    if args.landmarks is None:
        args.landmarks = 200 if fast else 32
    n, L, K, W = args.particles, args.landmarks, args.steps, args.warmup
    with_cpu = not args.no_cpu_baseline and ctx.world == 1 and not args.force_sharded  # rank 0 at N = 1 only

    if fast:
        if K == 200 and W == 20:
            K, W = 50, 5
        if ctx.sharded:
            if args.workload == "fastslam2":
                raise SystemExit("the sharded bench leg times FastSLAM 1.0 (the sharded update itself also serves FastSLAM 2.0 handles)")
            if n == 100_000 and ctx.world == 8:
                n = 125_000  # configs[3]: 1e6 particles over 8 GPUs
            out = leg_fastslam_sharded(ctx, n, L, K, W)
            ctx.close()
            if ctx.rank == 0:
                if "error" in out:
                    raise SystemExit(out["error"] + " (" + out.get("transport_note", "") + ")")
                emit(out)
            return
        emit(leg_fastslam(args, n, L, K, W, v2=args.workload == "fastslam2", with_cpu=with_cpu, breakdown=not args.no_breakdown))
        return

    log(f"headline leg: MCL {n} particles/GPU x {L} landmarks, world {ctx.world}")
    args.cold_first = os.environ.get("RR_BENCH_NO_COLD") is None  # the headline leg also reports the cold number (device_warmup_steps: 0) beside the hot one
    out = leg_mcl(args, ctx, n, L, K, W, with_cpu, breakdown=not args.no_breakdown)
    args.cold_first = False
    if ctx.rank == 0:
        _OUT["partial"] = out
    log("headline leg done")
    if not args.no_extra_legs and args.scheme == "systematic" and (n, L) == (1_000_000, 32):
        if not ctx.sharded:
            # configs[2], the HBM-bound workload, in the same run: its roofline fraction is the one the HBM target is about
            try:
                log("extra leg fastslam (configs[2])")
                leg = leg_fastslam(args, 100_000, 200, 50, 5, with_cpu=with_cpu, breakdown=not args.no_breakdown)
                out["fastslam"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg",
                                                       "obs_chunks", "device_warmup_steps") if k in leg}
                if "cpu_baseline" in leg:
                    out["fastslam"]["cpu_baseline"] = leg["cpu_baseline"]
            except Exception as e:  # noqa: BLE001 -- the headline line survives a failing extra leg
                out["fastslam"] = {"error": f"{type(e).__name__}: {e}"}
            # the resampler the reference's ParticleFilterLocalizer / MonteCarloLocalizer really use (multinomial draws,
            # particle_filter.rs:441-473, monte_carlo_localization.rs:322-365): the same workload with it, own roofline and CPU baseline
            try:
                log("extra leg mcl_multinomial")
                import copy

                a2 = copy.copy(args)
                a2.scheme = "multinomial"
                a2.cpu_seconds = 6.0
                leg = leg_mcl(a2, ctx, n, L, K, W, with_cpu, breakdown=not args.no_breakdown)
                out["mcl_multinomial"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg",
                                                              "headline_step", "synchronous_try_step", "index_parity", "cpu_baseline", "device_warmup_steps") if k in leg}
            except Exception as e:  # noqa: BLE001
                out["mcl_multinomial"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                log("extra leg small_n (configs[0])")
                out["small_n"] = leg_small_n(with_cpu)
            except Exception as e:  # noqa: BLE001
                out["small_n"] = {"error": f"{type(e).__name__}: {e}"}
            # FastSLAM 2.0 at the configs[2] shape
            try:
                log("extra leg fastslam2")
                leg = leg_fastslam(args, 100_000, 200, 50, 5, v2=True, with_cpu=with_cpu, breakdown=not args.no_breakdown)
                out["fastslam2"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg", "obs_chunks",
                                                        "device_warmup_steps", "cpu_baseline") if k in leg}
            except Exception as e:  # noqa: BLE001
                out["fastslam2"] = {"error": f"{type(e).__name__}: {e}"}
            # configs[4] at FULL size on this one GPU (1.6e7 x 64: 1.15 GB per launch, several times the Infinity Cache -- the
            # MCL measurement that is a DRAM measurement), then its per-GPU shape and configs[3]'s through the sharded step at world size 1
            try:
                log("extra leg mcl_config5_full (configs[4] unsharded)")
                import copy

                a5 = copy.copy(args)
                a5.cpu_seconds, a5.cpu_particles, a5.cpu_brief = 5.0, 2_000_000, True
                leg = leg_mcl(a5, ctx, 16_000_000, 64, 20, 3, with_cpu, breakdown=False, label="configs[4], all 1.6e7 particles on ONE GPU")
                out["mcl_config5_full"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg",
                                                               "headline_step", "plain_async_step", "cpu_baseline", "device_warmup_steps") if k in leg}
            except Exception as e:  # noqa: BLE001
                out["mcl_config5_full"] = {"error": f"{type(e).__name__}: {e}"}
            # configs[3] at FULL size on this one GPU (1e6 x 200: 19.3 GB of maps in two sets): the denominator of the FastSLAM
            # strong-scaling arithmetic (">= 6x at 8 GPUs")
            try:
                log("extra leg fastslam_config4_full (configs[3] unsharded)")
                leg = leg_fastslam(args, 1_000_000, 200, 10, 2, with_cpu=False, breakdown=False, device_warmup=10,
                                   label="configs[3], all 1e6 particles on ONE GPU")
                out["fastslam_config4_full"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline", "kernel_ms_avg",
                                                                    "obs_chunks", "device_warmup_steps") if k in leg}
            except Exception as e:  # noqa: BLE001
                out["fastslam_config4_full"] = {"error": f"{type(e).__name__}: {e}"}
            # the sharded step at world size 1, both transports: what a rank of the 8-GPU run pays before any cross-device latency
            # (weak-scaling ceiling = 8 x unsharded step / this)
            if not args.no_sharded_world1:
                out["sharded_world1"] = leg_sharded_world1(args, n, L, K, W)
                out["sharded_world1_config5_shape"] = leg_sharded_world1(args, 2_000_000, 64, K, W, transports=(("p2p", "p2p-only"),),
                                                                         what="BASELINE.json configs[4] per-GPU shape: 2e6 particles x 64 landmarks")
                out["fastslam_sharded_world1_config4_shape"] = leg_fastslam_sharded_world1(125_000, 200)
                # the reference's own PF / MCL resampler, sharded: peer-to-peer (round 6: every served slot stored straight into its
                # owner's slab) against the RCCL all-to-all form
                out["sharded_world1_multinomial"] = leg_sharded_world1(args, n, L, K, W, scheme="multinomial",
                                                                       what="configs[1] shape, MULTINOMIAL resampling (particle_filter.rs:441-473)")

                def ms(*path):
                    d = out
                    for k in path:
                        d = d.get(k) if isinstance(d, dict) else None
                    return d.get("ms_per_step") if isinstance(d, dict) else None

                def ratio(a, b, scale=1.0):
                    return round(scale * a / b, 3) if a and b else None

                # what 8 GPUs can reach BEFORE any cross-device latency (everything measured on this one GPU, world size 1):
                # weak: 8 x the unsharded step / the sharded step of the same per-GPU shape; strong: the full problem on one GPU / the
                # sharded step of its 1/8 shape
                out["weak_scaling_ceiling"] = {"configs[1] p2p": ratio(out.get("ms_per_step"), ms("sharded_world1", "p2p"), 8.0),
                                               "configs[1] rccl": ratio(out.get("ms_per_step"), ms("sharded_world1", "rccl"), 8.0)}
                out["strong_scaling_ceiling"] = {"configs[3] p2p": ratio(ms("fastslam_config4_full"), ms("fastslam_sharded_world1_config4_shape")),
                                                 "configs[4] p2p": ratio(ms("mcl_config5_full"), ms("sharded_world1_config5_shape", "p2p"))}
        elif ctx.world == 8 or args.all_legs:
            per_gpu = 1_000_000 // 8 if ctx.world == 8 else 125_000
            # the extra legs never take the headline down with them: an exception becomes an "error" entry
            def guarded(name, fn):
                log(f"extra leg {name}")
                try:
                    return fn()
                except BaseException as e:  # noqa: BLE001 -- SystemExit from a leg included
                    sys.stderr.write(f"bench.py: leg {name} failed on rank {ctx.rank}: {type(e).__name__}: {e}\n")
                    return {"error": f"{type(e).__name__}: {e}"}

            leg3 = guarded("fastslam_sharded", lambda: leg_fastslam_sharded(ctx, per_gpu, 200, 50, 5))
            leg5 = guarded("mcl_config5", lambda: leg_mcl(args, ctx, 2_000_000, 64, min(K, 100), W, False, breakdown=not args.no_breakdown,
                                                           label="configs[4]"))
            if ctx.rank == 0:
                out["fastslam_sharded"] = leg3
                out["mcl_config5"] = (leg5 if leg5 is None or "error" in leg5 else
                                      {k: leg5[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "config", "roofline",
                                                            "kernel_ms_avg", "sharded") if k in leg5})
    ctx.close()
    if ctx.rank == 0:
        emit(out)


if __name__ == "__main__":
    main()
