"""worker of tests/test_fs1_sharded_gloo.py: one rank of a gloo world running the sharded FastSLAM
protocol of include/rr_fastslam1.h on the CPU -- the D-spec oracle stands in for the kernels, the
exchanges are torch.distributed collectives, and the resample segments come from the product's
rr_sys_segment_matrix.  Writes its final block to argv[1]."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import dp, u32p, u64p  # noqa: E402
from rust_robotics_amd.sharded import segment_matrix  # noqa: E402
from tests.test_fs1_sharded_gloo import SEED, scenario  # noqa: E402


def main():
    out_dir, n_local, L, steps, chunks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, N = n_local, n_local * world
    det = oracle.det()
    md, nth, poses, maps, zs = scenario(N, L, steps)
    sl = slice(rank * n, (rank + 1) * n)
    pw, px, py, pyaw = (np.ascontiguousarray(poses[sl, k]) for k in range(4))
    planes = oracle.maps_aos_to_planes(np.ascontiguousarray(maps[sl]), n, L)  # [L][6][n]
    fired_log, moved = [], 0
    rstep = 0
    for t, z in enumerate(zs):
        det.det_fs1_predict(n, dp(px), dp(py), dp(pyaw), 1.0, 0.1, None, None, SEED, t, rank * n, C.byref(md))
        det.det_fs1_observe(n, dp(px), dp(py), dp(pyaw), dp(pw), dp(planes), dp(z) if len(z) else None, len(z), C.byref(md), chunks)
        # exchange 1: global weight maximum
        wmax = torch.tensor([det.det_wmax(n, dp(pw))], dtype=torch.float64)
        dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
        sh, tot, qh, ql = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        usable = det.det_fix_reduce(n, dp(pw), float(wmax[0]), N, C.byref(sh), C.byref(tot), C.byref(qh), C.byref(ql))
        # exchange 2: every shard's integer sums
        mine = torch.from_numpy(np.array([tot.value, qh.value, ql.value], dtype=np.uint64).view(np.int64))
        allv = torch.zeros(3 * world, dtype=torch.int64)
        dist.all_gather_into_tensor(allv, mine)
        a = allv.numpy().view(np.uint64).reshape(world, 3)
        totals = [int(v) for v in a[:, 0]]
        total = sum(totals)
        q2 = sum((int(a[g, 1]) << 64) + int(a[g, 2]) for g in range(world))
        usable = bool(usable) and total > 0
        neff = det.det_fix_neff(total, q2 >> 64, q2 & ((1 << 64) - 1)) if usable else 0.0
        fire = neff < nth
        fired_log.append(int(fire))
        this_rstep = rstep
        rstep += 1
        if not fire:
            if usable:
                pw = pw / det.det_fix_total_to_double(total, sh.value)
            continue
        assert usable, "the scenario keeps the weights positive"
        rho = det.det_resample_rho(SEED, this_rstep)
        cdf = np.empty(n, np.uint64)
        det.det_fix_cdf(n, dp(pw), 1, sh.value, sum(totals[:rank]), u64p(cdf))
        M, first = segment_matrix(rho, totals, N, n, rank)
        n_send = int(M[rank].sum())
        idx = np.empty(max(n_send, 1), np.uint32)
        if n_send:
            det.det_indices_systematic(n, u64p(cdf), total, N, first, n_send, rho, u32p(idx))
        idx = idx[:n_send]
        P = planes.reshape(6 * L, n)
        send = np.ascontiguousarray(np.vstack([px[idx], py[idx], pyaw[idx], P[:, idx]]).T)  # one row per served slot
        recv = torch.empty((n, 3 + 6 * L), dtype=torch.float64)
        dist.all_to_all_single(recv, torch.from_numpy(send), output_split_sizes=[int(v) for v in M[:, rank]],
                               input_split_sizes=[int(v) for v in M[rank]])
        moved += int(M[rank].sum() - M[rank, rank])
        r = recv.numpy()
        px, py, pyaw = (np.ascontiguousarray(r[:, k]) for k in range(3))
        planes = np.ascontiguousarray(r[:, 3:].T).reshape(-1)
        pw = np.full(n, 1.0 / N)  # fastslam1.rs:228
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pw=pw, px=px, py=py, pyaw=pyaw, planes=planes, fired=np.array(fired_log),
             moved=moved)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
