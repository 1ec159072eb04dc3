#!/usr/bin/env python3
"""How long does the peer-to-peer transport's connect (hipIpcOpenMemHandle of every peer's state slab, mailbox and inbox) take as the
slabs grow?  Round 6: eight PROCESSES of FastSLAM 125 000 x 200 (2.4 GB slab + 1.2 GB inbox each) did not come out of
rr_fs1_p2p_connect within 25 minutes on the one-GPU box, while 8 x 3 000 x 7 and MCL 8 x 2 000 000 (128 MB slabs) connect at once.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=W --master-addr 127.0.0.1 --master-port P tools/ipc_connect_probe.py n_local L
prints one line per rank: seconds in create / connect; a stack dump and exit after RR_PROBE_DUMP_S (default 150) if it does not return."""
import datetime
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist

    from rust_robotics_amd.sharded import gloo_allgather
    from rust_robotics_amd.slam.fastslam1 import ShardedFastSlam1

    n_local, L = int(sys.argv[1]), int(sys.argv[2])
    faulthandler.dump_traceback_later(int(os.environ.get("RR_PROBE_DUMP_S", "150")), exit=True)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=200))
    rank, world = dist.get_rank(), dist.get_world_size()
    t0 = time.perf_counter()
    shard = ShardedFastSlam1(rank, world, n_local, L, seed=1)
    t1 = time.perf_counter()
    dist.barrier()
    t2 = time.perf_counter()
    shard.connect_ipc(gloo_allgather(dist))
    t3 = time.perf_counter()
    gb = (3 + 6 * L) * n_local * 8 / 1e9
    print(f"IPC_PROBE world {world} rank {rank} n_local {n_local} L {L}: slab {2 * gb:.2f} GB + inbox {gb:.2f} GB; create {t1 - t0:.2f} s, connect {t3 - t2:.2f} s", flush=True)
    dist.barrier()
    shard.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
