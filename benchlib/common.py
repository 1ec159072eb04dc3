"""bench.py's shared pieces: the peaks the rooflines are priced against, the one-line stdout discipline (claim_stdout / log / deadline),
the scene, the rank context, the PMC-traffic lookup tied to the loaded library, host CPU calibration."""
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


FP64_VALU_PEAK = 256 * 4 * 16 * 2.4e9  # FP64 FMA lane-instructions / s: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 3.93e13


# algorithmic HBM bytes per particle of the dominant kernel: read x,y,yaw (24 B) + write
# x,y,yaw,v,w (40 B); the systematic path's k_step_lazy also reads and clears the 4-byte resample marker
K1_BYTES = {"systematic": 72.0, "multinomial": 64.0}


# sharded legs: a process that has torch's HIP context loaded stalls once on the host (~40 ms) somewhere in its first few
# hundred launches; this many extra untimed steps keep that out of the timed region
# (RR_BENCH_EXTRA_WARMUP: the shared-device rig of tests/test_gpu_world8.py, where a step of eight processes time-slicing one GPU
# takes tens of milliseconds and nothing about the rate is being measured, asks for fewer)
EXTRA_WARMUP = max(0, int(os.environ.get("RR_BENCH_EXTRA_WARMUP", "1000")))


# ---- one JSON line on stdout, whatever the native libraries print; progress on stderr; a deadline
_T0 = time.time()


_OUT = {"fd": None, "partial": None, "emitted": False}


def log(msg):
    """progress on stderr (the driver keeps it): which leg a rank is in when something takes long"""
    sys.stderr.write(f"[bench rank {os.environ.get('RANK', '0')} +{time.time() - _T0:6.1f}s] {msg}\n")
    sys.stderr.flush()


def claim_stdout():
    """gloo and RCCL print banners and warnings on file descriptor 1 from C++: from here on descriptor 1 IS stderr, and
    the one JSON line goes to a private copy of the real stdout (emit)."""
    if _OUT["fd"] is None:
        sys.stdout.flush()
        _OUT["fd"] = os.dup(1)
        os.dup2(2, 1)


# every leg: the device itself needs ~50 ms of this work before it runs at its steady rate -- measured, MCL 1e6 x 32 with
# --steps 20: 51.2 us/step after 5 warm-up steps, 48.3 after 300, 46.5 after 1000, 47.0 after 3000 (`k_step_lazy` 35.8 ->
# 31.2 us); FastSLAM 1e5 x 200: 408 us/update after 5 warm-up updates, 397 after 50.  These untimed steps run BEFORE the W
# warm-up steps of the command line and are reported as `device_warmup_steps`; the timed region is still exactly K steps.
DEVICE_WARMUP_MCL = 1000


DEVICE_WARMUP_FS = 60


FS1_BYTES_PER_UPDATE = 96.0  # k_fs1_observe: read 48 B + write 48 B per (particle, observed landmark), EKF branch


def library_sha16():
    """What identifies the build of the engine library this process has loaded (what tools/collect_profiles.sh stamps into the PMC
    summaries it writes): the first 16 hex digits of the SHA-256 over the library's SOURCES, which the library itself reports
    (rr_version: "... sources <hash>", csrc/Makefile) -- two builds of the same sources are not the same bytes, but they are the
    same kernels.  A library that does not say (an older build loaded for an A/B): the hash of the file."""
    import hashlib
    import re

    from rust_robotics_amd import _ffi

    try:
        m = re.search(r"sources ([0-9a-f]{16})\)", _ffi.lib().rr_version().decode())
        if m:
            return m.group(1)
    except Exception:  # noqa: BLE001
        pass
    try:
        with open(_ffi.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def measured_traffic(kernel_prefix, workload, est=None):
    """HBM bytes per launch of `kernel_prefix` from the committed PMC passes (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).
    Newest round first.  `est`: of k_step_lazy's instantiations the one whose last template argument (EST: the build that also
    adds up the deferred estimate) is this.
    The bytes are only returned when the summary was collected ON THE LIBRARY THAT IS LOADED NOW (its `library_sha16` column,
    written by tools/collect_profiles.sh, equals library_sha16()): a number measured on another build is not this build's traffic
    (VERDICT r5 weak 12).  Otherwise (None, why)."""
    import csv
    import glob

    sha = library_sha16()
    stale = None
    newest = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.csv")))[::-1]
    for name in newest:
        try:
            rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))) if r["workload"] == workload and r["kernel"].startswith(kernel_prefix)]
            if est is not None and any(r["kernel"].endswith((",true>", ",false>")) for r in rows):
                rows = [r for r in rows if r["kernel"].endswith(",true>" if est else ",false>")]
            if rows:  # several instantiations of one kernel in a run (a warm-up variant): the one that did the timed launches
                r = max(rows, key=lambda q: int(q["dispatches"]))
                have = r.get("library_sha16")
                if have and sha and have == sha:
                    return (float(r["read_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6, f"profiles/{name} [library sources sha256 {sha}: the loaded build's]"
                if stale is None:
                    mb = float(r["read_MB_corrected_x2"]) + float(r["write_MB"])
                    stale = (f"profiles/{name} holds {mb:.1f} MB per launch for this kernel, measured on " +
                             (f"library {have}" if have else "a build that left no hash") + f"; the loaded library is {sha}: not reported as this build's traffic")
        except Exception:
            pass
    return None, stale


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


_THREADS = {}


def pick_threads():
    """Thread count for the OpenMP CPU baseline: the fastest of {1, 2, 4, ...} up to the CPUs this process may
    use (scheduler affinity and cgroup quota -- a container usually sees far fewer than /proc/cpuinfo lists), found
    by a short calibration on the weight kernel of the literal restatement.  Returns (threads, {threads: updates/s})."""
    if _THREADS:
        return _THREADS["best"], _THREADS["table"]
    import oracle
    from oracle import dp

    ref = oracle.ref()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    n, L = 200_000, 32
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=n), rng.normal(size=n)
    w = np.empty(n)
    obs = np.ascontiguousarray(np.column_stack([rng.uniform(5, 20, L), rng.uniform(-20, 20, L), rng.uniform(-20, 20, L)]))
    table, t = {}, 1
    cands = []
    while t < avail:
        cands.append(t)
        t *= 2
    cands.append(avail)
    for th in cands:
        ref.ref_set_threads(th)
        ref.ref_pf_update_raw(n, dp(x), dp(y), dp(w), dp(obs), L, 0.2)  # creates the team
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ref.ref_pf_update_raw(n, dp(x), dp(y), dp(w), dp(obs), L, 0.2)
            best = min(best, time.perf_counter() - t0)
        table[th] = n * L / best
    ref.ref_set_threads(1)
    _THREADS["best"] = max(table, key=table.get)
    _THREADS["table"] = {str(k): round(v) for k, v in table.items()}
    _THREADS["avail"] = avail
    return _THREADS["best"], _THREADS["table"]


def make_scene(L, steps, seed):
    from tests import helpers as H

    lms = H.landmarks_grid(L, seed)
    rng = np.random.default_rng(seed + 1)
    return [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(steps)]


# ------------------------------------------------------------------------------------------------------
class Ctx:
    """rank / world / device of this process and, for world > 1 (or --force-sharded), the gloo group that
    carries bootstrap data and the timing barrier (never particle data)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("RR_BENCH_SHARE_DEVICE"):  # development knob: every rank on device 0, so that the multi-process
            self.local_rank = 0                       # flow (launcher, gloo group, IPC hand-off, ladder) runs on a one-GPU box
        self.sharded = self.world > 1 or args.force_sharded
        self.dist = None

    def init_group(self):
        if self.dist is not None or not self.sharded:
            return
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")  # a lone rank started without a launcher
        import datetime

        # a rank that dies inside a leg must not leave the others waiting for half an hour in a gloo collective
        dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(seconds=240))
        self.dist = dist

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def require_devices(ctx):
    """The engine has no CPU fallback: say so once, clearly, instead of failing somewhere inside a rank."""
    from rust_robotics_amd import _ffi

    have = int(_ffi.lib().rr_device_count())
    need = ctx.local_rank + 1
    if have < need:
        msg = (f"bench.py: no HIP device available for rank {ctx.rank} (local rank {ctx.local_rank}; {have} device(s) visible) -- "
               f"the engine has no CPU fallback; run on an MI355X box")
        if ctx.dist is not None:
            try:
                ctx.close()
            except Exception:
                pass
        raise SystemExit(msg)


_TRANSPORT = {"value": "auto"}


def ctx_transport(ctx):
    t = _TRANSPORT["value"]
    return {"auto": "auto", "p2p": "p2p", "p2p-only": "p2p", "rccl": "rccl", "torch": "rccl"}[t]


def pin_to_gpu_numa_node(device=0):
    """The synchronous step is two trips over the host link: keep this process on the NUMA node the GPU hangs off (what
    nodes/pf_localizer_node does at start).  Returns the cpulist it pinned to, or None."""
    import ctypes as C

    from rust_robotics_amd import _ffi

    try:
        buf = C.create_string_buffer(64)
        if _ffi.lib().rr_device_pci_bus_id(device, buf, 64) != 0:
            return None
        cpulist = open(f"/sys/bus/pci/devices/{buf.value.decode()}/local_cpulist").read().strip()
        cpus = set()
        for part in cpulist.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpulist
    except Exception:  # noqa: BLE001 -- a measurement nicety, never a failure
        return None


def mcl_instruction_budget(est=False):
    """(f64-rate lane-instructions per particle-landmark pair, per particle) of k_step_lazy, maintained next to the kernel
    (rust_robotics_amd/csrc/INSTRUCTION_BUDGET.json, written from the ISA dump by tools/count_isa.py); est: the EST build (the
    step kernel that also adds up the deferred in-step estimate of the step before -- the headline's since round 5)."""
    try:
        d = json.load(open(os.path.join(ROOT, "rust_robotics_amd", "csrc", "INSTRUCTION_BUDGET.json")))
        if est and isinstance(d.get("est"), dict) and d["est"].get("per_particle"):
            d = d["est"]
        return float(d["per_pair"]), float(d["per_particle"])
    except Exception:
        return 19.0, 530.0  # round-1 ISA count


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
