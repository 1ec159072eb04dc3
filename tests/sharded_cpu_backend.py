"""CPU stand-in for rust_robotics_amd.sharded.HipShard, built on the D-spec oracle, so that the
N > 1 orchestration (collective order, segment matrix, all-to-all splits) runs under gloo
without a GPU.  TEST CODE: the product never imports this."""
import ctypes as C
import math

import numpy as np
import torch

import oracle
from oracle import dp, u32p, u64p
from rust_robotics_amd.sharded import ShardPlan


class CpuShard:
    torch = torch

    def __init__(self, rank, world, n_local, *, seed, sigma=0.2, sigma_v=2.0, sigma_w=math.radians(40.0), dt=0.1,
                 gate_always=True, threshold=1.0, lik=0, initial_state=None, scheme=1):
        self.det = oracle.det()
        self.scheme = scheme  # 1 = systematic, 0 = multinomial (rust_robotics_amd._ffi.RR_RESAMPLE_*)
        self.rank, self.world, self.n_local = rank, world, n_local
        self.n_global = n_local * world
        self.gid0 = rank * n_local
        self.p = dict(seed=seed, sigma=sigma, sv=sigma_v, sw=sigma_w, dt=dt, gate=gate_always, thr=threshold, lik=lik)
        n = n_local
        self.x, self.y, self.yaw, self.v = (np.zeros(n) for _ in range(4))
        if initial_state is not None:
            st = np.ascontiguousarray(initial_state, dtype=np.float64)
            self.det.det_pf_init(n, seed, self.gid0, dp(st), dp(self.x), dp(self.y), dp(self.yaw), dp(self.v))
        self.w = np.full(n, 1.0 / self.n_global)
        self.uniform = True
        self.step_ctr = 0
        self.rstep_ctr = 0
        self.wmax = torch.zeros(1, dtype=torch.float64)
        self.sums = torch.zeros(3, dtype=torch.int64)
        self.all_sums = torch.zeros(world * 3, dtype=torch.int64)
        self.recv_buf = torch.empty((n, 4 if scheme == 1 else 5), dtype=torch.float64)
        self.counts = torch.zeros(world, dtype=torch.int64)
        self.all_counts = torch.zeros(world * world, dtype=torch.int64)
        self._plan = None

    def propagate_weight(self, u, obs):
        p, n = self.p, self.n_local
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        self.det.det_pf_predict(n, dp(self.x), dp(self.y), dp(self.yaw), dp(self.v), u[0], u[1], p["dt"], None, None,
                                p["seed"], self.step_ctr, self.gid0, p["sv"], p["sw"])
        self.det.det_pf_weights(n, dp(self.x), dp(self.y), dp(self.w), dp(obs), obs.shape[0], p["sigma"], p["lik"])
        self.uniform = False
        self.step_ctr += 1
        self.wmax[0] = self.det.det_wmax(n, dp(self.w))

    def quantize(self):
        n = self.n_local
        wmax = float(self.wmax[0])
        sh, tot, qh, ql = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        usable = 0 if self.uniform else self.det.det_fix_reduce(n, dp(self.w), wmax, self.n_global, C.byref(sh), C.byref(tot),
                                                                C.byref(qh), C.byref(ql))
        self.usable, self.shift = usable, sh.value
        vals = (tot.value, qh.value, ql.value) if usable else (n, 0, n)  # uniform image q_i = 1
        self.sums[:] = torch.from_numpy(np.array(vals, dtype=np.uint64).view(np.int64))

    def cdf(self):
        a = self.all_sums.numpy().view(np.uint64).reshape(self.world, 3)
        totals = [int(v) for v in a[:, 0]]
        total = sum(totals)
        base = sum(totals[: self.rank])
        q2 = sum((int(a[g, 1]) << 64) + int(a[g, 2]) for g in range(self.world))
        if self.usable and total > 0:
            neff = self.det.det_fix_neff(total, q2 >> 64, q2 & ((1 << 64) - 1))
        else:
            neff = float(self.n_global)
        fired = True if self.p["gate"] else (neff < self.n_global * self.p["thr"])
        rho = self.det.det_resample_rho(self.p["seed"], self.rstep_ctr)
        self.rstep_ctr += 1
        self._totals = totals
        self._cdf = np.empty(self.n_local, np.uint64)
        self.det.det_fix_cdf(self.n_local, dp(self.w), self.usable, self.shift, base, u64p(self._cdf))
        self._plan = ShardPlan(fired, bool(self.usable), total, base, totals[self.rank], rho)

    def plan(self):
        return self._plan

    def totals(self):
        return self._totals

    def gather_slots(self, first_slot, n_slots):
        idx = np.empty(n_slots, np.uint32)
        if n_slots:
            self.det.det_indices_systematic(self.n_local, u64p(self._cdf), self._plan.total_global, self.n_global, first_slot,
                                            n_slots, self._plan.rho, u32p(idx))
        rows = np.column_stack([self.x[idx], self.y[idx], self.yaw[idx], self.v[idx]]) if n_slots else np.zeros((0, 4))
        return torch.from_numpy(np.ascontiguousarray(rows))

    # ---- multinomial shards: the slots this shard serves are those whose draw lands in its CDF interval
    def select(self):
        p = self._plan
        targets = np.empty(self.n_global, np.uint64)
        self.det.det_targets_multinomial(p.total_global, 0, self.n_global, self.p["seed"], self.rstep_ctr - 1, u64p(targets))
        t = targets.astype(object)  # exact integer comparison (values up to 2^63)
        lo, hi = p.base, p.base + p.total_local
        mine = np.array([lo < int(v) <= hi for v in t])
        self._served = np.nonzero(mine)[0]  # ascending global slot = grouped by destination
        self._served_targets = targets[self._served]
        cnt = np.bincount(self._served // self.n_local, minlength=self.world)
        self.counts[:] = torch.from_numpy(cnt.astype(np.int64))

    def pack_selected(self, n_send):
        assert n_send == self._served.size
        src = np.searchsorted(self._cdf, self._served_targets, side="left")  # first j with C_j >= target
        rows = np.column_stack([self.x[src], self.y[src], self.yaw[src], self.v[src], (self._served % self.n_local).astype(np.float64)])
        return torch.from_numpy(np.ascontiguousarray(rows.reshape(-1, 5)))

    def adopt_records(self, recv):
        a = recv.numpy()
        k = a[:, 4].astype(np.int64)
        assert np.array_equal(np.sort(k), np.arange(self.n_local)), "every slot exactly once"
        x, y, yaw, v = (np.empty(self.n_local) for _ in range(4))
        x[k], y[k], yaw[k], v[k] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        self.x, self.y, self.yaw, self.v = x, y, yaw, v
        self.w = np.full(self.n_local, 1.0 / self.n_global)
        self.uniform = True

    def adopt(self, recv):
        a = recv.numpy()
        self.x, self.y, self.yaw, self.v = (np.ascontiguousarray(a[:, k]) for k in range(4))
        self.w = np.full(self.n_local, 1.0 / self.n_global)
        self.uniform = True

    def local_moments(self):
        est = np.empty(4)
        cov = np.empty(16)
        w = np.full(self.n_local, 1.0) if self.uniform else self.w
        self.det.det_pf_moments(self.n_local, dp(self.x), dp(self.y), dp(self.yaw), dp(self.v), dp(w), float(w.sum()), dp(est), dp(cov))
        return est, cov.reshape(4, 4)
