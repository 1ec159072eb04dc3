"""debug: sharded p2p steps followed by an accessor that materialises the pending window resample (two processes on one
device: RR_BENCH_SHARE_DEVICE-style, or in one process with --local)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers as H
from rust_robotics_amd.sharded import P2PShard, gloo_allgather

n_local, steps, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
local = len(sys.argv) > 4 and sys.argv[4] == "--local"
lms = H.landmarks_grid(L, 1)
rng = np.random.default_rng(2)
obs = [H.observations(lms, H.true_pose(t + 1), 0.2, rng) for t in range(steps)]
kw = dict(seed=1, initial_state=[0.0, 0.0, 0.0, 1.0])
if local:
    world = 2
    shards = [P2PShard(g, world, 0, n_local, **kw) for g in range(world)]
    P2PShard.link_local(shards)
    for t in range(steps):
        for s in shards: s.step([1.0, 0.1], obs[t])
    for s in shards: s.synchronize()
    print("local: timed_out after steps", [s.timed_out() for s in shards], flush=True)
    for s in shards:
        try:
            print("local: moments", s.local_moments()[0], flush=True)
        except Exception as e:
            print("local: moments failed:", str(e)[:90], flush=True)
else:
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s = P2PShard(rank, world, 0, n_local, **kw)
    s.connect_ipc(gloo_allgather(dist))
    dist.barrier()
    for t in range(steps):
        s.step([1.0, 0.1], obs[t])
        if t == 11 and os.environ.get("DBG_ACCESSOR"):
            got = s.particles()
            print(f"rank {rank}: accessor at step 12 ok, timed_out {s.timed_out()}", flush=True)
            dist.barrier()
    s.synchronize()
    dist.barrier()
    print(f"rank {rank}: timed_out after steps {s.timed_out()}", flush=True)
    dist.barrier()
    try:
        print(f"rank {rank}: moments", s.local_moments()[0], flush=True)
    except Exception as e:
        print(f"rank {rank}: moments failed:", str(e)[:90], flush=True)
    dist.barrier()
    s.close()
    dist.destroy_process_group()
