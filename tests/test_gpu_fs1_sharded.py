"""Sharded FastSLAM 1.0 (BASELINE config 4 shape) over the peer-to-peer transport.
Contiguous particle blocks, each with its whole map; the resample stores every served slot's
3 + 6L planes straight into the owning shard's slab.  Every shard's poses, weights and maps must
equal the unsharded engine's bit for bit (same seed, same obs_chunks), over steps where the
N_eff gate fires and steps where it does not."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 77


def engines(variant):
    """(unsharded class, sharded class, params with the gate at n / 1.5) of FastSLAM 1.0 or 2.0"""
    from rust_robotics_amd.slam import fastslam1 as fs
    from rust_robotics_amd.slam import fastslam2 as fs2

    if variant == 2:
        return fs2.FastSlam2, fs2.ShardedFastSlam2, fs2.default_params()
    return fs.FastSlam1, fs.ShardedFastSlam1, fs.default_params()


def scenario(n, L, steps, variant=1):
    """initial state, controls and observations shared by the sharded and unsharded runs"""
    from rust_robotics_amd.slam import fastslam1 as fs
    from tests.test_gpu_fs1_parity import make_state, scene

    lms = scene(L, 5)
    poses, maps = make_state(n, L, lms, 6)
    prm = engines(variant)[2]
    base = prm.base if variant == 2 else prm
    base.nth = n / 1.5
    base.initial_weight = 1.0 / n
    xt = np.zeros(3)
    zs = []
    for t in range(steps):
        xt = np.array([xt[0] + 0.1 * np.cos(xt[2]), xt[1] + 0.1 * np.sin(xt[2]), xt[2] + 0.01])
        z = np.array(fs.get_observations(xt, [tuple(p) for p in lms], seed=SEED, step=t)).reshape(-1, 3)
        if t == 3:
            z = z[:0]  # a step without observations
        zs.append(z)
    return prm, poses, maps, zs


def unsharded(n, L, steps, chunks, variant=1):
    prm, poses, maps, zs = scenario(n, L, steps, variant)
    f = engines(variant)[0](n, L, params=prm, seed=SEED, obs_chunks=chunks)
    f.set_state(poses, maps)
    fired = []
    for z in zs:
        f.update([1.0, 0.1], z)
        fired.append(f.last_resample_fired())
    return f.get_state(), fired


def check(shard_states, n_local, L, steps, chunks, variant=1):
    (ep, em), fired = unsharded(n_local * len(shard_states), L, steps, chunks, variant)
    assert any(fired) and not all(fired), fired  # both branches of the gate were exercised
    for g, (p, m) in enumerate(shard_states):
        sl = slice(g * n_local, (g + 1) * n_local)
        assert np.array_equal(p.view(np.uint64), ep[sl].view(np.uint64)), f"rank {g}: poses / weights differ"
        assert np.array_equal(m.view(np.uint64), em[sl].view(np.uint64)), f"rank {g}: maps differ"


def run_in_process(world, n_local, L=7, steps=8, chunks=2, variant=1):
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1

    n = world * n_local
    prm, poses, maps, zs = scenario(n, L, steps, variant)
    shards = [engines(variant)[1](g, world, n_local, L, params=prm, seed=SEED, obs_chunks=chunks) for g in range(world)]
    for g, s in enumerate(shards):
        s.set_state(poses[g * n_local:(g + 1) * n_local], maps[g * n_local:(g + 1) * n_local])
    ShardedFastSlam1.link_local(shards)
    for t, z in enumerate(zs):
        for s in shards:  # only enqueued; the device-side waits pair the shards up
            s.update_async([1.0, 0.1], z)
        if t == 5:  # an accessor in mid-run has to make the pending (lazy) resample real
            for s in shards:
                s.poses()
    states = []
    for g, s in enumerate(shards):
        assert not s.timed_out(), f"rank {g}: a peer wait timed out"
        states.append(s.get_state())
    check(states, n_local, L, steps, chunks, variant)
    print("FS1_P2P_LOCAL_OK")


@pytest.mark.parametrize("world,n_local,variant", [(1, 3000, 1), (2, 2500, 1), (3, 1300, 1), (2, 2500, 2), (3, 1300, 2)])
def test_in_process_shards_equal_unsharded(world, n_local, variant):
    """fresh interpreter with enough hardware queues: see tests/test_gpu_p2p.py; variant 2 = FastSLAM 2.0
    (the proposal kernel reads the first observation's landmark through the lazy resample indices)"""
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_fs1_sharded import run_in_process; "
            f"run_in_process({world}, {n_local}, variant={variant})")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and "FS1_P2P_LOCAL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_two_processes_over_ipc_handles():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "tests", "_gpu_fs1_p2p_worker.py"), "3000", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and r.stdout.count("FS1_P2P_OK") == 2, (r.stdout[-2000:], r.stderr[-4000:])


def run_rccl_world1(n_local=3000, L=7, steps=8, chunks=2):
    """the RCCL transport with a one-rank communicator: all-reduce / all-gather really run, nothing migrates"""
    n = n_local
    prm, poses, maps, zs = scenario(n, L, steps)
    s = engines(1)[1](0, 1, n_local, L, params=prm, seed=SEED, obs_chunks=chunks)
    s.set_state(poses, maps)
    s.connect_rccl(lambda raw: raw)
    for t, z in enumerate(zs):
        s.update_async([1.0, 0.1], z)
        if t == 5:
            s.poses()
    assert s.migrated() == 0
    check([s.get_state()], n_local, L, steps, chunks)
    s.close()
    print("FS1_RCCL_W1_OK")


def test_rccl_transport_world1_equals_unsharded():
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_fs1_sharded import run_rccl_world1; run_rccl_world1()"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "FS1_RCCL_W1_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("world,n_local,variant", [(2, 2500, 1), (3, 1300, 1), (3, 1300, 2)])
def test_rccl_phases_in_process_equal_unsharded(world, n_local, variant):
    """The phases rr_fs1_shard_update is made of (local / quantize / plan / pack / unpack), for `world` shards
    living on ONE device, with the three collectives done by hand on the host (max, concatenation, and the
    per-pair [plane][count] blocks copied from the senders' buffers into the receivers') -- everything of the
    RCCL transport except the RCCL calls themselves, whole particles crossing shards included.  Fresh interpreter
    with torch imported FIRST: torch (device buffers here) brings its own HIP runtime, which has to be the one
    the engine library binds to."""
    code = (f"import torch, sys; sys.path.insert(0, {ROOT!r}); from tests.test_gpu_fs1_sharded import run_rccl_phases; "
            f"run_rccl_phases({world}, {n_local}, {variant})")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0 and "FS1_RCCL_PHASES_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def run_rccl_phases(world, n_local, variant):
    import ctypes as C

    import torch

    from rust_robotics_amd import _ffi
    from rust_robotics_amd.sharded import segment_matrix

    L, steps, chunks = 7, 8, 2
    n = world * n_local
    lib = _ffi.lib()
    prm, poses, maps, zs = scenario(n, L, steps, variant)
    shards = [engines(variant)[1](g, world, n_local, L, params=prm, seed=SEED, obs_chunks=chunks) for g in range(world)]
    for g, s in enumerate(shards):
        s.set_state(poses[g * n_local:(g + 1) * n_local], maps[g * n_local:(g + 1) * n_local])
    dev = torch.device("cuda", 0)
    n_planes = 3 + 6 * L
    dp = C.POINTER(C.c_double)
    V = C.c_void_p
    u = np.array([1.0, 0.1])
    moved_total = 0

    def ok(st):
        assert st == _ffi.RR_OK, _ffi.last_error()

    for t, z in enumerate(zs):
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, 3)
        zp = z.ctypes.data_as(dp) if z.size else None
        wmax = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in shards]
        for g, s in enumerate(shards):
            ok(lib.rr_fs1_shard_local(s._h, u.ctypes.data_as(dp), zp, z.shape[0], V(wmax[g].data_ptr())))
            s.synchronize()
        gmax = torch.stack(wmax).max().reshape(1).contiguous()  # "all-reduce MAX"
        sums = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in shards]
        torch.cuda.synchronize()
        for g, s in enumerate(shards):
            ok(lib.rr_fs1_shard_quantize(s._h, V(gmax.data_ptr()), V(sums[g].data_ptr())))
            s.synchronize()
        allv = torch.cat(sums).contiguous()  # "all-gather"
        torch.cuda.synchronize()
        plans = []
        for g, s in enumerate(shards):
            ok(lib.rr_fs1_shard_plan(s._h, V(allv.data_ptr()), world, g))
            pl = _ffi.PfShardPlan()
            ok(lib.rr_fs1_shard_get_plan(s._h, C.byref(pl)))
            plans.append(pl)
        assert len({(p.fired, p.rho, p.total_global) for p in plans}) == 1, "every shard must derive the same plan"
        if not plans[0].fired:
            continue
        totals = [int(v) for v in allv.cpu().numpy().view(np.uint64).reshape(world, 3)[:, 0]]
        M, _ = segment_matrix(plans[0].rho, totals, n, n_local, 0)
        Mc = np.ascontiguousarray(M, dtype=np.int64)
        mp = Mc.ctypes.data_as(C.POINTER(C.c_int64))
        send = []
        for g, s in enumerate(shards):
            cnt = int(M[g].sum() - M[g, g])
            buf = torch.full((max(cnt * n_planes, 1),), float("nan"), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            ok(lib.rr_fs1_shard_pack(s._h, mp, world, g, V(buf.data_ptr())))
            s.synchronize()
            send.append(buf)
        for g, s in enumerate(shards):  # "grouped send / recv": block (src -> g) out of src's buffer, sources ascending
            blocks = []
            for src in range(world):
                if src == g or M[src, g] == 0:
                    continue
                off = int(sum(M[src, d] for d in range(g) if d != src)) * n_planes
                blocks.append(send[src][off:off + int(M[src, g]) * n_planes])
            moved_total += sum(b.numel() for b in blocks) // n_planes
            recv = torch.cat(blocks).contiguous() if blocks else torch.zeros(1, dtype=torch.float64, device=dev)
            assert not torch.isnan(recv).any()
            torch.cuda.synchronize()
            ok(lib.rr_fs1_shard_unpack(s._h, mp, world, g, V(recv.data_ptr())))
            s.synchronize()
        if t == 5:
            for s in shards:
                s.poses()
    assert moved_total > 0, "expected whole particles to cross shards"
    check([s.get_state() for s in shards], n_local, L, steps, chunks, variant)
    print("FS1_RCCL_PHASES_OK")


def test_shard_geometry_is_checked():
    from rust_robotics_amd.core import RoboticsError
    from rust_robotics_amd.slam.fastslam1 import FastSlam1, ShardedFastSlam1

    with pytest.raises(RoboticsError):
        FastSlam1(100, 2, first_global_index=50, n_global=120)  # block sticks out of the global range
    s = ShardedFastSlam1(0, 2, 100, 2)
    with pytest.raises(RoboticsError):
        s.update([1.0, 0.0], [])  # not connected
    with pytest.raises(RoboticsError):
        ShardedFastSlam1.link_local([s])  # world 2 declared, one shard linked
