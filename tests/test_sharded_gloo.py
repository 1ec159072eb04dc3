"""N > 1 path on CPU: world_size-2 (and 3, and 8: the BASELINE's rank count) gloo runs of rust_robotics_amd.sharded.ShardedLocalizer
over the oracle-backed stand-in must reproduce the single-shard D-spec trajectory bit for bit
(the integer CDF makes the particle set independent of the number of shards)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(tmp_path, world, n_local, steps, gate_always, port, scheme=1):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_worker.py"), str(tmp_path),
           str(n_local), str(steps), str(int(gate_always)), str(scheme)]
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return [np.load(os.path.join(tmp_path, f"rank{g}.npz")) for g in range(world)]


@pytest.mark.parametrize("world,gate_always,port,scheme", [(2, True, 29611, 1), (3, True, 29612, 1), (2, False, 29613, 1),
                                                          (2, True, 29614, 0), (3, False, 29615, 0),
                                                          (8, True, 29616, 1), (8, False, 29617, 1), (8, True, 29618, 0)])
def test_sharded_equals_single_shard(tmp_path, det, world, gate_always, port, scheme):
    """scheme 1 = systematic (contiguous served slots, segment matrix), 0 = multinomial (the resampler
    MonteCarloLocalizer uses, monte_carlo_localization.rs:322-365,387-392: scattered served slots, count matrix)"""
    n_local, steps = 600, 12
    n = n_local * world
    ranks = run_world(tmp_path, world, n_local, steps, gate_always, port, scheme)
    z = np.zeros(n)
    d = H.DetPF(det, z, z, z, z, dt=0.1, sigma=0.5, sigma_v=0.3, sigma_w=math.radians(5.0), threshold=1.0 if gate_always else 0.9,
                gate=1 if gate_always else 0, scheme=scheme, lik=0, seed=42)
    rng = np.random.default_rng(43)
    fired = []
    for t in range(steps):
        obs = H.observations(H.REF_SCENE_LANDMARKS, H.true_pose(t + 1), 0.5, rng)
        fired.append(int(d.step([1.0, 0.1], obs)))
    for g, r in enumerate(ranks):
        assert r["fired"].tolist() == fired
        sl = slice(g * n_local, (g + 1) * n_local)
        for name, e in (("x", d.x), ("y", d.y), ("yaw", d.yaw), ("v", d.v)):
            assert np.array_equal(r[name].view(np.uint64), e[sl].view(np.uint64)), f"rank {g} {name}"
        if not bool(r["uniform"]):
            assert np.array_equal(r["w"].view(np.uint64), d.w[sl].view(np.uint64))
    assert any(fired)
    if gate_always:
        assert int(ranks[0]["moved"]) > 0, "expected some particles to migrate between ranks"
    est, cov = d.moments()
    for r in ranks:
        np.testing.assert_allclose(r["est"], est, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["cov"], cov, rtol=1e-7, atol=1e-9)


def test_segment_matrix_properties():
    from rust_robotics_amd.sharded import first_slot_above, segment_matrix

    rng = np.random.default_rng(1)
    for _ in range(200):
        G = int(rng.integers(1, 9))
        n_local = int(rng.integers(1, 5000))
        n = G * n_local
        totals = [int(v) for v in rng.integers(0, 2**40, G)]
        if rng.random() < 0.3:
            totals[int(rng.integers(0, G))] = 0
        if sum(totals) == 0:
            totals[-1] = 1
        rho = float(np.floor(rng.random() * 2**53) / 2**53)
        M, first = segment_matrix(rho, totals, n, n_local, 0)
        assert first == 0
        assert M.sum() == n
        assert np.all(M.sum(axis=0) == n_local)  # every destination slot has exactly one source
        assert first_slot_above(rho, sum(totals), n, sum(totals)) == n
        assert first_slot_above(rho, sum(totals), n, 0) == 0
