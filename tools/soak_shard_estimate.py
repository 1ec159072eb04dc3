#!/usr/bin/env python3
"""Soak of the sharded in-step estimate against the unsharded filter's, the two stepping CONCURRENTLY on one device
(python tools/soak_shard_estimate.py [trials] [n_local] [steps]; GPU_MAX_HW_QUEUES=8 is set here).

Round 4 ended with a ~1 % disagreement of the two means in 1 of ~60 runs of tests/test_gpu_p2p.py::
test_in_process_shards_leave_their_part_of_the_mean (world size 1) and a serialising synchronize() in the test.  This tool is
the hunt: it repeats that test's trajectory `trials` times inside one process WITHOUT the serialisation and, instead of
stopping at the first mismatch, says for each one
  * which side is wrong: both means against the mean of the materialised particle set (read-at-once steps),
  * whether a second read of the same sums gives another value (a copy that overtook its kernel) or the same (wrong in memory),
  * whether the two particle sets are still identical,
  * which (tile, wave) sums differ from the host's own sums over the same slots, and whether those hold the step before's values.
Exit status 0 iff no trial disagreed.  RR_SOAK_SYNC=1 puts the serialisation back (the control experiment)."""
import ctypes as C
import json
import math
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests import helpers as H  # noqa: E402
import rust_robotics_amd.localization as loc  # noqa: E402
from rust_robotics_amd import _ffi  # noqa: E402
from rust_robotics_amd.sharded import P2PShard  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_local = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
SYNC = os.environ.get("RR_SOAK_SYNC", "0") != "0"
BUDGET_S = float(os.environ.get("RR_SOAK_BUDGET_S", "240"))
os.environ["RR_PF_EST_DEFER"] = "1"
L = _ffi.lib()
L.rr_pf_debug_est_slot_partials.restype = C.c_int
L.rr_pf_debug_est_slot_partials.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64)]
TRACE = hasattr(L, "rr_pf_debug_trace")  # the instrumented build (make -C rust_robotics_amd/csrc trace; RR_AMD_LIBRARY=...)
PLAIN = os.environ.get("RR_SOAK_PLAIN", "0") != "0"  # no estimates anywhere: does the sharded step itself diverge?
DBG_WORDS, DBG_CAP = 48, 32
if TRACE:
    L.rr_pf_debug_trace.restype = C.c_int
    L.rr_pf_debug_trace.argtypes = [C.c_void_p, C.c_uint32]
    L.rr_pf_debug_trace_read.restype = C.c_int
    L.rr_pf_debug_trace_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64]


def trace_of(handle):
    out = np.zeros(DBG_CAP * DBG_WORDS, dtype=np.uint64)
    assert L.rr_pf_debug_trace_read(handle, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size) == 0, _ffi.last_error()
    return out.reshape(DBG_CAP, DBG_WORDS)


def f64(u):
    return float(np.array([u], dtype=np.uint64).view(np.float64)[0])


def show_trace(sh, rf, upto):
    """shard: what k_step_lazy saw ([0..9]), what k_shard_plan_mark's workgroup 0 ([10..13]) and last arrival ([14..28]) left, what
    tiles 0..2 read back ([32..43]); unsharded: k_step_lazy's view ([0..9]) and the first 112 bytes of Ctl after the plan ([14..27])"""
    for st in range(max(0, upto - 6), upto + 1):
        a, b = sh[st % DBG_CAP], rf[st % DBG_CAP]
        ints = b[14:18].view(np.int32)  # cur, weights_uniform, usable, image_mode, shift, fired, pending, grid_timeout
        print(f"  step {st}: SHARD L saw step={a[0]} pending={a[1]} cur={a[2]} served=[{a[3]},+{a[4]}) wmax_bits_at_entry={f64(a[5]):.6g} wait_seq={a[6]} est={a[7]} fired={a[8]}"
              f" | P wg0: local_wmax={f64(a[10]):.17g} global={f64(a[11]):.17g} seq={a[12]} epoch={a[13]}"
              f" | P last(wg {a[24]}): T_local={a[14]} T={a[15]} fired={a[16]} pending={a[17]} cur={a[18]} base={a[19]} q2lo={a[20]} served=[{a[21]},+{a[22]}) shift={a[23]} wmax={f64(a[25]):.17g} rho={f64(a[26]):.17g} mode={a[27]}"
              f" | tiles read (pre, base, T, fired|shift|mode): {[(int(a[32+4*k]), int(a[33+4*k]), int(a[34+4*k]), hex(int(a[35+4*k]))) for k in range(3)]}")
        print(f"  step {st}: REF   L saw step={b[0]} pending={b[1]} cur={b[2]} wmax_bits_at_entry={f64(b[5]):.6g} est={b[7]} fired={b[8]}"
              f" | Ctl after plan: cur={ints[0]} uniform={ints[1]} usable={ints[2]} mode={ints[3]} shift={ints[4]} fired={ints[5]} pending={ints[6]}"
              f" wmax_bits={b[18]} T={b[19]} T_local={b[20]} base={b[21]} q2lo={b[23]} wmax={f64(b[24]):.17g} rho={f64(b[27]):.17g}", flush=True)


SLOTS, WAVES = 512, 4  # rr::kResolveSlots, kBlock / 64
n_tiles = (n_local + SLOTS - 1) // SLOTS


def partials(handle):
    out = np.empty(n_tiles * WAVES * 4)
    n = C.c_uint64()
    st = L.rr_pf_debug_est_slot_partials(handle, out.ctypes.data_as(C.POINTER(C.c_double)), out.size, C.byref(n))
    assert st == 0 and n.value == out.size, (st, _ffi.last_error())
    return out.reshape(n_tiles, WAVES, 4)


def host_partials(p):
    """the sums the kernels form: slot k = tile * 512 + r * 256 + tid, wave = tid // 64, over rows r = 0, 1 (any order: 1e-11 is asked)"""
    f = np.zeros((n_tiles * SLOTS, 4))
    f[:n_local] = p[:, :4]
    return f.reshape(n_tiles, 2, WAVES, 64, 4).sum(axis=(1, 3))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


kw = dict(range_noise=0.5, velocity_noise=0.3, yaw_rate_noise=math.radians(5.0))
bad = []
t_start = time.time()
done = 0
for trial in range(trials):
    if time.time() - t_start > BUDGET_S:
        break
    shard = P2PShard(0, 1, 0, n_local, seed=42 + trial, **kw)
    P2PShard.link_local([shard])
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n_local, max_particles=n_local, **kw)
    ref = loc.MonteCarloLocalizer(cfg, seed=42 + trial, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
    rng = np.random.default_rng(43 + trial)
    prev_sp = prev_rp = None
    if TRACE:
        assert L.rr_pf_debug_trace(shard.h, DBG_CAP) == 0 and L.rr_pf_debug_trace(ref._h, DBG_CAP) == 0
    shown = False
    if PLAIN:  # the bare steps, two at a time without a host wait in between, the particle sets compared after every pair
        for t in range(0, steps, 2):
            for q in range(2):
                obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + q + 1), 0.5, rng)
                shard.step([1.0, 0.1], obs)
                if SYNC:
                    shard.synchronize()
                ref.step_async([1.0, 0.1], obs)
            if not np.array_equal(bits(shard.particles()), bits(ref.get_particles_array())):
                bad.append({"trial": trial, "plain_pair_ending_at_step": t + 2})
                print(f"MISMATCH trial {trial}: particle sets differ after the pair of steps ending at {t + 2}", flush=True)
                if TRACE:
                    show_trace(trace_of(shard.h), trace_of(ref._h), t + 1)
                break
        shard.close()
        del ref
        done += 1
        continue

    def step_shard(obs, want):
        shard.want_estimate(want)
        shard.step([1.0, 0.1], obs)
        if SYNC:
            shard.synchronize()

    t = 0
    while t < steps:
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        step_shard(obs, True)
        ref.step_async_estimate([1.0, 0.1], obs)
        want = np.array(ref.last_step_estimate())
        truth_set = ref.get_particles_array()  # the resampled set, materialised by the read above, not yet propagated
        truth = truth_set[:, :4].mean(axis=0)
        rp = partials(ref._h)
        t += 1
        later = t % 3 == 0
        if later:
            obs2 = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
            step_shard(obs2, False)
            ref.step_async([1.0, 0.1], obs2)
            t += 1
        sums, den = shard.estimate_sums()
        got = sums / den
        if not np.array_equal(bits(got), bits(want)):
            sums2, _ = shard.estimate_sums()
            sp = partials(shard.h)
            hp = host_partials(truth_set)
            rec = {"trial": trial, "step": t, "read": "a step later" if later else "at once", "shard": got.tolist(), "ref": want.tolist(),
                   "truth_mean_of_ref_set": truth.tolist(),
                   "shard_rel_err": float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), 1e-300))),
                   "ref_rel_err": float(np.max(np.abs(want - truth) / np.maximum(np.abs(truth), 1e-300))),
                   "second_read_of_shard": (sums2 / den).tolist(), "second_read_differs": bool(not np.array_equal(bits(sums2), bits(sums)))}
            for name, dev, prev in (("shard", sp, prev_sp), ("ref", rp, prev_rp)):
                d = np.abs(dev - hp) > 1e-9 * np.maximum(1.0, np.abs(hp))
                wrong = np.argwhere(d.any(axis=2))
                rec[name + "_wrong_tile_wave"] = wrong[:40].tolist()
                rec[name + "_n_wrong"] = int(len(wrong))
                if prev is not None and len(wrong):
                    rec[name + "_wrong_equal_previous_steps"] = int(sum(np.array_equal(bits(dev[a, b]), bits(prev[a, b])) for a, b in wrong))
                if len(wrong):
                    a, b = wrong[0]
                    rec[name + "_first_wrong"] = {"device": dev[a, b].tolist(), "host": hp[a, b].tolist()}
            if not later:  # both sets are materialised: identical?
                rec["particles_identical"] = bool(np.array_equal(bits(shard.particles()), bits(ref.get_particles_array())))
            bad.append(rec)
            print("MISMATCH " + json.dumps(rec), flush=True)
            if TRACE and not shown:
                shown = True
                show_trace(trace_of(shard.h), trace_of(ref._h), t - 1)
        prev_sp = partials(shard.h)
        prev_rp = rp
    ps, pr = shard.particles(), ref.get_particles_array()
    if not np.array_equal(bits(ps), bits(pr)):
        bad.append({"trial": trial, "particles_differ_at_end": True})
        print(f"MISMATCH trial {trial}: particle sets differ at the end", flush=True)
    assert not shard.timed_out()
    shard.close()
    del ref
    done += 1
print(json.dumps({"trials": done, "n_local": n_local, "steps": steps, "serialised": SYNC, "mismatches": len(bad),
                  "seconds": round(time.time() - t_start, 1), "fused_plan_env": os.environ.get("RR_PF_FUSED_PLAN"), "plain": PLAIN, "trace_build": TRACE}), flush=True)
sys.exit(1 if bad else 0)
