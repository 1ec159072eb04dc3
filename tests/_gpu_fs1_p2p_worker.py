"""worker of tests/test_gpu_fs1_sharded.py::test_two_processes_over_ipc_handles and tests/test_gpu_world8.py: one process per
shard, all on device 0, peers mapped through hipIpc handles exchanged over gloo.
    argv: n_local steps [L = 7].  Small L: the host-made state of tests/test_gpu_fs1_sharded.py::scenario, every shard's state
    gathered on rank 0 and compared there.  L >= 100 (BASELINE configs[3]: 125 000 x 200 per rank): the engines initialise
    themselves (first_obs_cov = 0.5), rank 0 alone runs the unsharded filter of all particles AFTER the shards are done, and the
    comparison goes by a BLAKE2 digest of every block's poses and maps (no rank ever holds more than its own block, rank 0 the
    unsharded state besides)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(*arrays):
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    return h.hexdigest()


def main():
    import torch.distributed as dist

    from rust_robotics_amd.sharded import gloo_allgather
    from rust_robotics_amd.slam import fastslam1 as fs
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1
    from tests import helpers as H
    from tests.test_gpu_fs1_sharded import SEED, check, scenario

    import datetime
    import faulthandler
    import time

    n_local, steps = int(sys.argv[1]), int(sys.argv[2])
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    big = L >= 100
    chunks = 0 if big else 2
    # a rank that stops must say WHERE, and must not keep the others in a collective for gloo's default half hour
    faulthandler.dump_traceback_later(int(os.environ.get("RR_WORKER_DUMP_AFTER_S", "420")), exit=True)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("RR_WORKER_GLOO_TIMEOUT_S", "480"))))
    rank, world = dist.get_rank(), dist.get_world_size()
    t_start = time.time()

    def log(msg):
        sys.stderr.write(f"[fs1 p2p worker rank {rank} +{time.time() - t_start:6.1f}s] {msg}\n")
        sys.stderr.flush()

    n = n_local * world
    if big:
        lms = np.random.default_rng(61).uniform(-13.0, 13.0, size=(L, 2))

        def params():
            prm = fs.default_params()
            prm.first_obs_cov, prm.nth, prm.initial_weight = 0.5, n / 1.5, 1.0 / n
            return prm

        zs = [np.array(fs.get_observations(H.true_pose(t + 1, v=0.5), [tuple(p) for p in lms], seed=8, step=t)).reshape(-1, 3) for t in range(steps)]
        shard = ShardedFastSlam1(rank, world, n_local, L, params=params(), seed=8, obs_chunks=chunks)
        u = [0.5, 0.1]
    else:
        prm, poses, maps, zs = scenario(n, L, steps)
        sl = slice(rank * n_local, (rank + 1) * n_local)
        shard = ShardedFastSlam1(rank, world, n_local, L, params=prm, seed=SEED, obs_chunks=chunks)
        shard.set_state(poses[sl], maps[sl])
        u = [1.0, 0.1]
    log("shard created")
    shard.connect_ipc(gloo_allgather(dist))
    log("peers mapped")
    dist.barrier()
    for z in zs:
        shard.update_async(u, z)
    log("updates enqueued")
    assert not shard.timed_out(), "a peer wait timed out"
    log("updates done, no give-up")
    got = shard.get_state()
    log("state read back")
    if big:
        mine = digest(*got)
        del got
        dist.barrier()
        log("closing the shard")
        shard.close()  # (its 3.6 GB back before rank 0 puts 19.3 GB beside the other shards)
        shard = None
        log("shard closed")
        digests = [None] * world
        dist.all_gather_object(digests, mine)
        if rank == 0:
            whole = fs.FastSlam1(n, L, params=params(), seed=8, obs_chunks=chunks)
            log("unsharded filter created")
            fired = []
            for z in zs:
                whole.update(u, z)
                fired.append(bool(whole.last_resample_fired()))
            ep, em = whole.get_state()
            whole.close()
            log("unsharded filter done")
            assert any(fired), fired
            for g in range(world):
                sl = slice(g * n_local, (g + 1) * n_local)
                assert digests[g] == digest(ep[sl], em[sl]), f"rank {g}: poses / weights / maps differ from the unsharded filter"
            print(f"FS1_P2P_CONFIG4 world {world} n_local {n_local} L {L} steps {steps} gate_fired {fired}: every block equals the unsharded filter", flush=True)
    else:
        states = [None] * world
        dist.all_gather_object(states, got)
        if rank == 0:
            check(states, n_local, L, steps, chunks)
    dist.barrier()
    del shard
    dist.destroy_process_group()
    print("FS1_P2P_OK")


if __name__ == "__main__":
    main()
