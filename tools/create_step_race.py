#!/usr/bin/env python3
"""Does a filter that is stepped IMMEDIATELY after it was created give the same particles as one whose creation was left to settle
first?  (Round 6: the markers of the lazy systematic resample were cleared with a hipMemset -- the null stream, which may return
before it has run and which a non-blocking stream does not wait for -- so the first plan could mark into an earlier tenant's bytes.)
Run with RR_DEBUG_POISON_ALLOC=0x01 to make "an earlier tenant's bytes" certain; RR_AMD_LIBRARY=<older build> for the A/B.

    python tools/create_step_race.py [particles] [rounds]            one JSON line: how many immediate filters differed from the settled one"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rust_robotics_amd.localization as loc
    from rust_robotics_amd import _ffi
    from tests import helpers as H

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lms = H.landmarks_grid(32, 1)
    rng = np.random.default_rng(5)
    obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(4)]
    cfg = loc.MonteCarloLocalizationConfig(min_particles=n, max_particles=n)

    def run(settle):
        f = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=0, resample_scheme=_ffi.RR_RESAMPLE_SYSTEMATIC)
        if settle:
            f.synchronize()
            time.sleep(0.05)
        for o in obs:
            f.step_async([1.0, 0.1], o)
        d = hashlib.blake2b(memoryview(np.ascontiguousarray(f.get_particles_array())).cast("B"), digest_size=10).hexdigest()
        del f
        return d

    want = run(True)
    again = run(True)
    bad = [i for i in range(rounds) if run(False) != want]
    print(json.dumps(dict(library=os.environ.get("RR_AMD_LIBRARY", "in-tree"), poison=os.environ.get("RR_DEBUG_POISON_ALLOC", "0"), particles=n, rounds=rounds,
                          settled_repeatable=again == want, immediate_differs=len(bad), which=bad[:10])), flush=True)


if __name__ == "__main__":
    main()
