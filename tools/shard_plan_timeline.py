#!/usr/bin/env python3
"""In-kernel timeline of a shard's one-launch resample plan (k_shard_plan_mark, peer-to-peer transport, world size 1) at the
headline size, from the wall-clock stamps of the instrumented build (make -C rust_robotics_amd/csrc timeline):

    RR_AMD_LIBRARY=rust_robotics_amd/librust_robotics_amd_timeline.so python tools/shard_plan_timeline.py [out.json]

Stations per workgroup (thread 0): 0 start, 1 the global weight maximum known, 2 tile record stored, 3 ticket taken,
4 prefix + totals seen, 5 markers written; the last arrival alone: 6 records scanned, 7 sums traded with the peers.  Microseconds
after the first workgroup's start: min / median / max over the workgroups, median over the sampled launches.  Both senders of the
weight maximum (RR_P2P_WMAX_EARLY = 1: the step kernel's last workgroup; 0: the plan kernel's first)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import P2PShard
    from tests import helpers as H

    L_ = _ffi.lib()
    if not hasattr(L_, "rr_pf_debug_plan_timeline"):
        sys.exit("load the instrumented library: RR_AMD_LIBRARY=rust_robotics_amd/librust_robotics_amd_timeline.so")
    L_.rr_pf_debug_plan_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    n, L = 1_000_000, 32
    n_tiles, words = (n + 2047) // 2048, 8
    lms = H.landmarks_grid(L, 1)
    names = ["start", "maximum known", "record stored", "ticket taken", "prefix + totals seen", "markers written"]
    out = {"workload": f"one shard (world size 1, p2p transport) of MCL {n} x {L}, systematic, resample every step", "n_tiles": n_tiles,
           "clock": "wall_clock64, 100 MHz", "stations": names + ["last arrival: records scanned", "last arrival: sums traded"]}
    for early in ("1", "0"):
        os.environ["RR_P2P_WMAX_EARLY"] = early
        sh = P2PShard(0, 1, 0, n, seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
        P2PShard.link_local([sh])
        rng = np.random.default_rng(2)
        for t in range(1500):
            sh.step([1.0, 0.1], H.observations(lms, H.true_pose(t + 1), 0.2, rng))
        sh.synchronize()
        samples, last = [], []
        for t in range(40):
            sh.step([1.0, 0.1], H.observations(lms, H.true_pose(1501 + t), 0.2, rng))
            buf = np.zeros(n_tiles * words, dtype=np.uint64)
            assert L_.rr_pf_debug_plan_timeline(sh.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), buf.size) == 0
            st = buf.reshape(n_tiles, words).astype(np.int64)
            t0 = st[:, 0].min()
            samples.append((st[:, :6] - t0) / 100.0)
            who = int(np.argmax(st[:, 3]))  # the last ticket
            last.append(((st[who, 6] - t0) / 100.0, (st[who, 7] - t0) / 100.0))
        a = np.stack(samples)
        rows = {}
        for k in range(6):
            v = a[:, :, k]
            rows[names[k]] = {"min_us": float(np.median(v.min(axis=1))), "median_us": float(np.median(np.median(v, axis=1))),
                              "max_us": float(np.median(v.max(axis=1)))}
        la = np.array(last)
        rows["last arrival: records scanned"] = {"median_us": float(np.median(la[:, 0]))}
        rows["last arrival: sums traded"] = {"median_us": float(np.median(la[:, 1]))}
        out[f"RR_P2P_WMAX_EARLY={early}"] = rows
        assert not sh.timed_out()
        sh.close()
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text + "\n")


if __name__ == "__main__":
    main()
