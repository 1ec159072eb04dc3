"""Particle set sharded over the GPUs of one node: one process per GPU, RCCL between the phases.

SURVEY.md section 8(e): particles are independent in propagate / weight; the coupling is (i) the
global weight maximum and integer sums and (ii) the resample permutation.  Rank g owns the
contiguous global index block [g*N/G, (g+1)*N/G).  Per step (systematic resampling,
fastslam1.rs:205-234 semantics on the MCL step of monte_carlo_localization.rs:291-300):

  A  propagate + weight (local kernel)               -> local max weight
     all-reduce(MAX) of one double                   -- RCCL, 8 B
  B  integer image under the GLOBAL max (local)      -> local (T, sum q^2)
     all-gather of 3 x u64 per rank                  -- RCCL, 24 B * G
  C  global totals, gate, local slice of the global CDF (local kernels)
     host reads the G totals (one small D2H) and derives, with integer arithmetic only, which
     contiguous run of global output slots every rank serves
  D  every rank gathers the particles its slots' owners need into one contiguous send buffer
     all-to-all of contiguous 32-byte-per-particle segments -- RCCL; in steady state only
     neighbouring ranks exchange the few particles by which the cumulative weights drift
  E  the received block becomes the new particle set (w = 1/N)

Because every quantity that feeds back into the particle state is an exact integer sum, the
sharded filter produces bit-identical particles for any G (tests/test_sharded_gloo.py checks
G = 2 against G = 1 on CPU with the oracle standing in for the kernels).

The orchestration below is backend-agnostic: ``HipShard`` drives the C ABI (include/rr_pf.h
"sharded operation"); tests inject a CPU stand-in built on the oracle to cover the N > 1 logic
with the gloo backend.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .core import RoboticsError


@dataclass
class ShardPlan:
    fired: bool
    usable: bool
    total_global: int
    base: int
    total_local: int
    rho: float


def first_slot_above(rho: float, total_global: int, n_global: int, bound: int) -> int:
    """Host-only integer helper of the C ABI (no GPU needed)."""
    return int(_ffi.lib().rr_sys_first_slot_above(rho, total_global, n_global, bound))


def segment_matrix(rho: float, totals: Sequence[int], n_global: int, n_local: int, rank: int = 0):
    """(M, first): M[src][dst] = number of global output slots owned by rank dst whose source particle
    lives on rank src, for systematic positions (i + rho) / N over the integer CDF with per-rank totals
    ``totals``; first = first global slot ``rank`` serves.  Pure integer arithmetic in the C ABI
    (``rr_sys_segment_matrix``); every rank computes the same matrix."""
    G = len(totals)
    t = np.ascontiguousarray(totals, dtype=np.uint64)
    M = np.zeros((G, G), dtype=np.int64)
    first = _ffi.lib().rr_sys_segment_matrix(rho, t.ctypes.data_as(C.POINTER(C.c_uint64)), G, n_global, n_local, rank,
                                             M.ctypes.data_as(C.POINTER(C.c_int64)))
    return M, int(first)


class HipShard:
    """One shard of the MCL engine on one GPU, driven through the C ABI with torch tensors as
    the device buffers the collectives operate on (torch is plumbing here: memory + streams)."""

    def __init__(self, rank: int, world: int, device: int, n_local: int, *, seed: int, range_noise=0.2,
                 velocity_noise=2.0, yaw_rate_noise=math.radians(40.0), dt=0.1, gate=_ffi.RR_GATE_ALWAYS,
                 resample_threshold=1.0, likelihood_mode=_ffi.RR_LIK_FUSED, initial_state=None,
                 scheme=_ffi.RR_RESAMPLE_SYSTEMATIC):
        import torch

        self.torch = torch
        self.scheme = scheme
        self.rank, self.world, self.n_local = rank, world, n_local
        self.n_global = n_local * world
        self.device = torch.device("cuda", device)
        L = _ffi.lib()
        self.L = L
        cfg = _ffi.PfConfig(n_local, resample_threshold, range_noise, velocity_noise, yaw_rate_noise, dt)
        opt = _ffi.PfOptions()
        L.rr_pf_options_default(C.byref(opt))
        opt.device = device
        opt.seed = seed
        opt.resample_scheme = scheme
        opt.resample_gate = gate
        opt.likelihood_mode = likelihood_mode
        opt.first_global_index = rank * n_local
        opt.n_global = self.n_global
        self.h = C.c_void_p()
        if initial_state is None:
            self._check(L.rr_pf_create(C.byref(cfg), C.byref(opt), C.byref(self.h)))
        else:
            st = np.ascontiguousarray(initial_state, dtype=np.float64)
            self._check(L.rr_pf_create_with_state(C.byref(cfg), C.byref(opt), st.ctypes.data_as(C.POINTER(C.c_double)),
                                                  C.byref(self.h)))
        with torch.cuda.device(self.device):
            self.wmax = torch.zeros(1, dtype=torch.float64, device=self.device)
            self.sums = torch.zeros(3, dtype=torch.int64, device=self.device)
            self.all_sums = torch.zeros(world * 3, dtype=torch.int64, device=self.device)
            self.all_sums_host = torch.zeros(world * 3, dtype=torch.int64).pin_memory()
            width = 4 if scheme == _ffi.RR_RESAMPLE_SYSTEMATIC else 5  # multinomial records carry the destination slot
            self.send_buf = torch.empty((n_local * 2, width), dtype=torch.float64, device=self.device)
            self.recv_buf = torch.empty((n_local, width), dtype=torch.float64, device=self.device)
            self.counts = torch.zeros(world, dtype=torch.int64, device=self.device)
            self.all_counts = torch.zeros(world * world, dtype=torch.int64, device=self.device)
            stream = torch.cuda.current_stream(self.device)
            if stream.cuda_stream == 0:
                # the legacy default stream has handle 0, which rr_pf_set_stream reads as "use your own stream" -- the
                # library's kernels and torch's tensor ops / collectives would then run unordered.  Give the shard a
                # real stream and make it torch's current one for this device.
                stream = torch.cuda.Stream(device=self.device)
                torch.cuda.set_stream(stream)
            self.stream = stream
        self._check(L.rr_pf_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def _check(self, status: int) -> None:
        if status != _ffi.RR_OK:
            kind = RoboticsError.invalid_parameter if status == _ffi.RR_INVALID_PARAMETER else RoboticsError.runtime
            raise kind(_ffi.last_error())

    def close(self) -> None:
        if self.h:
            self.L.rr_pf_destroy(self.h)
            self.h = None

    # ---- phases
    def propagate_weight(self, u, obs: np.ndarray) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_propagate_weight(self.h, u.ctypes.data_as(dp),
                                                        obs.ctypes.data_as(dp) if obs.size else None, obs.shape[0],
                                                        C.c_void_p(self.wmax.data_ptr())))

    def quantize(self) -> None:
        self._check(self.L.rr_pf_shard_quantize(self.h, C.c_void_p(self.wmax.data_ptr()), C.c_void_p(self.sums.data_ptr())))

    def cdf(self) -> None:
        self._check(self.L.rr_pf_shard_cdf(self.h, C.c_void_p(self.all_sums.data_ptr()), self.world, self.rank))
        self.all_sums_host.copy_(self.all_sums, non_blocking=True)

    def plan(self) -> ShardPlan:
        p = _ffi.PfShardPlan()
        self._check(self.L.rr_pf_shard_get_plan(self.h, C.byref(p)))  # synchronises the stream
        return ShardPlan(bool(p.fired), bool(p.usable), int(p.total_global), int(p.base), int(p.total_local), float(p.rho))

    def totals(self) -> List[int]:
        a = self.all_sums_host.numpy().view(np.uint64).reshape(self.world, 3)
        return [int(v) for v in a[:, 0]]

    def gather_slots(self, first_slot: int, n_slots: int):
        if n_slots > self.send_buf.shape[0]:
            self.send_buf = self.torch.empty((n_slots, 4), dtype=self.torch.float64, device=self.device)
        out = self.send_buf[:n_slots]
        if n_slots:
            self._check(self.L.rr_pf_shard_gather_slots(self.h, first_slot, n_slots, C.c_void_p(out.data_ptr())))
        return out

    def adopt(self, recv) -> None:
        self._check(self.L.rr_pf_shard_adopt(self.h, C.c_void_p(recv.data_ptr())))

    # ---- multinomial shards: scattered served slots (include/rr_pf.h "Multinomial shards")
    def select(self) -> None:
        """counts of the slots this shard serves per destination -> self.counts"""
        self._check(self.L.rr_pf_shard_select(self.h, self.world, C.c_void_p(self.counts.data_ptr())))

    def pack_selected(self, n_send: int):
        if n_send > self.send_buf.shape[0]:
            self.send_buf = self.torch.empty((n_send + n_send // 4, 5), dtype=self.torch.float64, device=self.device)
        out = self.send_buf[:n_send]
        self._check(self.L.rr_pf_shard_pack_selected(self.h, self.world, C.c_void_p(self.send_buf.data_ptr())))
        return out

    def adopt_records(self, recv) -> None:
        self._check(self.L.rr_pf_shard_adopt_records(self.h, C.c_void_p(recv.data_ptr()), recv.shape[0]))

    # ---- read-out (local)
    def particles(self) -> np.ndarray:
        out = np.empty((self.n_local, 5))
        self._check(self.L.rr_pf_get_particles(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def local_moments(self):
        e = np.empty(4)
        c = np.empty(16)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_estimate(self.h, e.ctypes.data_as(dp)))
        self._check(self.L.rr_pf_covariance(self.h, c.ctypes.data_as(dp)))
        return e, c.reshape(4, 4)

    def synchronize(self) -> None:
        self._check(self.L.rr_pf_synchronize(self.h))

    def profile(self, on: bool) -> None:
        self._check(self.L.rr_pf_profile_enable(self.h, int(on)))
        if on:
            self._check(self.L.rr_pf_profile_reset(self.h))

    def profile_read(self) -> dict:
        out = {}
        for k in range(_ffi.RR_K_COUNT):
            n, ms = C.c_uint64(), C.c_double()
            self._check(self.L.rr_pf_profile_read(self.h, k, C.byref(n), C.byref(ms)))
            out[self.L.rr_pf_kernel_name(k).decode()] = (n.value, ms.value)
        return out


class ShardedLocalizer:
    """The step of one rank; ``backend`` is a HipShard (or a stand-in with the same phase methods
    and ``wmax`` / ``sums`` / ``all_sums`` tensors); ``dist`` is torch.distributed."""

    def __init__(self, backend, dist, group=None):
        self.b = backend
        self.dist = dist
        self.group = group
        self.last_matrix: Optional[np.ndarray] = None
        self.weight_share = 1.0 / backend.world  # this rank's share of the total weight

    def step(self, u, obs) -> bool:
        b, dist = self.b, self.dist
        b.propagate_weight(u, obs)
        dist.all_reduce(b.wmax, op=dist.ReduceOp.MAX, group=self.group)
        b.quantize()
        dist.all_gather_into_tensor(b.all_sums, b.sums, group=self.group)
        b.cdf()
        plan = b.plan()
        if not plan.fired:
            self.weight_share = plan.total_local / plan.total_global if plan.total_global else 1.0 / b.world
            return False
        r = b.rank
        if getattr(b, "scheme", _ffi.RR_RESAMPLE_SYSTEMATIC) == _ffi.RR_RESAMPLE_MULTINOMIAL:
            # iid draws: the slots a shard serves are scattered over all ranks.  Every rank counts what it serves per
            # destination, the counts are all-gathered into the exchange matrix, the records travel in one all-to-all
            b.select()
            dist.all_gather_into_tensor(b.all_counts, b.counts, group=self.group)
            M = b.all_counts.cpu().numpy().reshape(b.world, b.world).astype(np.int64)
            self.last_matrix = M
            assert int(M[:, r].sum()) == b.n_local, "every output slot of this rank must have exactly one source"
            send = b.pack_selected(int(M[r].sum()))
            recv = b.recv_buf
            dist.all_to_all_single(recv, send, output_split_sizes=[int(v) for v in M[:, r]],
                                   input_split_sizes=[int(v) for v in M[r]], group=self.group)
            b.adopt_records(recv)
            self.weight_share = 1.0 / b.world
            return True
        totals = b.totals()
        M, first = segment_matrix(plan.rho, totals, b.n_global, b.n_local, r)
        self.last_matrix = M
        n_send = int(M[r].sum())
        send = b.gather_slots(first, n_send)
        recv = b.recv_buf
        assert int(M[:, r].sum()) == b.n_local, "every output slot of this rank must have exactly one source"
        dist.all_to_all_single(recv, send, output_split_sizes=[int(v) for v in M[:, r]],
                               input_split_sizes=[int(v) for v in M[r]], group=self.group)
        b.adopt(recv)
        self.weight_share = 1.0 / b.world
        return True

    def estimate(self):
        """Global weighted mean / covariance (particle_filter.rs:382-413) from per-rank moments:
        all-gather of (W_g, mean_g, cov_g), combined in rank order on every rank."""
        torch = self.b.torch
        e, c = self.b.local_moments()
        rec = np.concatenate([[self.weight_share], e, c.reshape(-1)])
        t = torch.from_numpy(rec).to(self.b.wmax.device)
        allr = torch.empty(self.b.world * rec.size, dtype=torch.float64, device=t.device)
        self.dist.all_gather_into_tensor(allr, t, group=self.group)
        a = allr.cpu().numpy().reshape(self.b.world, -1)
        W = a[:, 0].sum()
        mean = (a[:, 0:1] * a[:, 1:5]).sum(0) / W
        cov = np.zeros((4, 4))
        for g in range(self.b.world):
            d = a[g, 1:5] - mean
            cov += a[g, 0] * (a[g, 5:].reshape(4, 4) + np.outer(d, d))
        return mean, cov / W


class NativeShard:
    """One shard driven entirely from inside the library: ``rr_pf_shard_step`` runs the phases
    and calls RCCL itself (include/rr_pf.h "native sharded step").  Python only creates the
    communicator -- the 128-byte unique id is created on rank 0 and handed over by ``exchange``,
    any callable that broadcasts a bytes object from rank 0 (bench.py uses torch.distributed/gloo)."""

    def __init__(self, rank: int, world: int, device: int, n_local: int, exchange, *, seed: int, range_noise=0.2,
                 velocity_noise=2.0, yaw_rate_noise=math.radians(40.0), dt=0.1, gate=_ffi.RR_GATE_ALWAYS,
                 resample_threshold=1.0, likelihood_mode=_ffi.RR_LIK_FUSED, initial_state=None,
                 scheme=_ffi.RR_RESAMPLE_SYSTEMATIC):
        L = _ffi.lib()
        self.L, self.rank, self.world, self.n_local = L, rank, world, n_local
        uid = (C.c_uint8 * 128)()
        if rank == 0:
            self._check(L.rr_comm_unique_id(uid))
        raw = exchange(bytes(uid))
        uid = (C.c_uint8 * 128).from_buffer_copy(raw)
        self.comm = C.c_void_p()
        self._check(L.rr_comm_create(uid, rank, world, device, C.byref(self.comm)))
        cfg = _ffi.PfConfig(n_local, resample_threshold, range_noise, velocity_noise, yaw_rate_noise, dt)
        opt = _ffi.PfOptions()
        L.rr_pf_options_default(C.byref(opt))
        opt.device, opt.seed = device, seed
        opt.resample_scheme, opt.resample_gate, opt.likelihood_mode = scheme, gate, likelihood_mode
        opt.first_global_index, opt.n_global = rank * n_local, n_local * world
        self.h = C.c_void_p()
        if initial_state is None:
            self._check(L.rr_pf_create(C.byref(cfg), C.byref(opt), C.byref(self.h)))
        else:
            st = np.ascontiguousarray(initial_state, dtype=np.float64)
            self._check(L.rr_pf_create_with_state(C.byref(cfg), C.byref(opt), st.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.h)))

    def _check(self, status: int) -> None:
        if status != _ffi.RR_OK:
            kind = RoboticsError.invalid_parameter if status == _ffi.RR_INVALID_PARAMETER else RoboticsError.runtime
            raise kind(_ffi.last_error())

    def step(self, u, obs) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_step(self.h, self.comm, u.ctypes.data_as(dp), obs.ctypes.data_as(dp) if obs.size else None,
                                            obs.shape[0]))

    def estimate(self):
        e, c = np.empty(4), np.empty(16)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_estimate(self.h, self.comm, e.ctypes.data_as(dp), c.ctypes.data_as(dp)))
        return e, c.reshape(4, 4)

    def particles(self) -> np.ndarray:
        out = np.empty((self.n_local, 5))
        self._check(self.L.rr_pf_get_particles(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def migrated(self) -> int:
        return int(self.L.rr_pf_shard_last_migrated(self.h))

    def want_estimate(self, on: bool = True) -> None:
        """rr_pf_shard_want_estimate (systematic shards): every step leaves this shard's part of the mean try_step returns."""
        self._check(self.L.rr_pf_shard_want_estimate(self.h, 1 if on else 0))

    def estimate_sums(self):
        """(sums[4], N) of the last step: the mean is the sum of every shard's sums over N."""
        sums, den = np.empty(4), C.c_double()
        self._check(self.L.rr_pf_shard_last_estimate_sums(self.h, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(den)))
        return sums, den.value

    def synchronize(self) -> None:
        self._check(self.L.rr_pf_synchronize(self.h))

    def profile(self, on: bool) -> None:
        self._check(self.L.rr_pf_profile_enable(self.h, int(on)))
        if on:
            self._check(self.L.rr_pf_profile_reset(self.h))

    def profile_read(self) -> dict:
        out = {}
        for k in range(_ffi.RR_K_COUNT):
            n, ms = C.c_uint64(), C.c_double()
            self._check(self.L.rr_pf_profile_read(self.h, k, C.byref(n), C.byref(ms)))
            out[self.L.rr_pf_kernel_name(k).decode()] = (n.value, ms.value)
        return out

    def close(self) -> None:
        if self.h:
            self.L.rr_pf_destroy(self.h)
            self.h = None
        if self.comm:
            self.L.rr_comm_destroy(self.comm)
            self.comm = None


class LocalWindowShards:
    """Bring-up / test seam (``rr_pf_shard_step_local``): the RCCL transport's sharded step for ``world`` shards that all live
    in this process on one device, the three exchanges done as device copies.  Every kernel and all of the host's segment
    arithmetic are the RCCL transport's own."""

    def __init__(self, world: int, n_local: int, device: int = 0, *, seed: int, range_noise=0.2, velocity_noise=2.0,
                 yaw_rate_noise=math.radians(40.0), dt=0.1, gate=_ffi.RR_GATE_ALWAYS, resample_threshold=1.0,
                 likelihood_mode=_ffi.RR_LIK_FUSED, initial_state=None):
        L = _ffi.lib()
        self.L, self.world, self.n_local = L, world, n_local
        self.hs = (C.c_void_p * world)()
        self.cs = (C.c_void_p * world)()
        for g in range(world):
            cfg = _ffi.PfConfig(n_local, resample_threshold, range_noise, velocity_noise, yaw_rate_noise, dt)
            opt = _ffi.PfOptions()
            L.rr_pf_options_default(C.byref(opt))
            opt.device, opt.seed = device, seed
            opt.resample_scheme, opt.resample_gate, opt.likelihood_mode = _ffi.RR_RESAMPLE_SYSTEMATIC, gate, likelihood_mode
            opt.first_global_index, opt.n_global = g * n_local, n_local * world
            h = C.c_void_p()
            if initial_state is None:
                self._check(L.rr_pf_create(C.byref(cfg), C.byref(opt), C.byref(h)))
            else:
                st = np.ascontiguousarray(initial_state, dtype=np.float64)
                self._check(L.rr_pf_create_with_state(C.byref(cfg), C.byref(opt), st.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h)))
            self.hs[g] = h.value
            c = C.c_void_p()
            self._check(L.rr_comm_create_local(g, world, device, C.byref(c)))
            self.cs[g] = c.value

    def _check(self, status: int) -> None:
        if status != _ffi.RR_OK:
            kind = RoboticsError.invalid_parameter if status == _ffi.RR_INVALID_PARAMETER else RoboticsError.runtime
            raise kind(_ffi.last_error())

    def step(self, u, obs) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_step_local(self.hs, self.cs, self.world, u.ctypes.data_as(dp), obs.ctypes.data_as(dp) if obs.size else None,
                                                  obs.shape[0]))

    def particles(self, g: int) -> np.ndarray:
        out = np.empty((self.n_local, 5))
        self._check(self.L.rr_pf_get_particles(C.c_void_p(self.hs[g]), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def synchronize(self) -> None:
        for g in range(self.world):
            self._check(self.L.rr_pf_synchronize(C.c_void_p(self.hs[g])))

    def want_estimate(self, on: bool = True) -> None:
        for g in range(self.world):
            self._check(self.L.rr_pf_shard_want_estimate(C.c_void_p(self.hs[g]), 1 if on else 0))

    def estimate(self) -> np.ndarray:
        """the mean of the last step's resampled set: the shards' sums (rr_pf_shard_last_estimate_sums) over N"""
        total, den = np.zeros(4), C.c_double()
        for g in range(self.world):
            sums = np.empty(4)
            self._check(self.L.rr_pf_shard_last_estimate_sums(C.c_void_p(self.hs[g]), sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(den)))
            total += sums
        return total / den.value

    def migrated(self) -> int:
        return int(self.L.rr_pf_shard_last_migrated(C.c_void_p(self.hs[0])))

    def close(self) -> None:
        for g in range(self.world):
            if self.hs[g]:
                self.L.rr_pf_destroy(C.c_void_p(self.hs[g]))
                self.hs[g] = None
            if self.cs[g]:
                self.L.rr_comm_destroy(C.c_void_p(self.cs[g]))
                self.cs[g] = None


class P2PShard:
    """One shard using the peer-to-peer transport (include/rr_pf.h "peer-to-peer transport"):
    no host code and no collective library inside a step.  ``connect_ipc`` is for one process per
    GPU (handles travel through ``allgather``: a callable returning every rank's bytes in rank
    order); ``link_local`` is for several shards inside one process."""

    def __init__(self, rank: int, world: int, device: int, n_local: int, *, seed: int, range_noise=0.2,
                 velocity_noise=2.0, yaw_rate_noise=math.radians(40.0), dt=0.1, gate=_ffi.RR_GATE_ALWAYS,
                 resample_threshold=1.0, likelihood_mode=_ffi.RR_LIK_FUSED, initial_state=None,
                 scheme=_ffi.RR_RESAMPLE_SYSTEMATIC):
        """scheme: RR_RESAMPLE_SYSTEMATIC (the window step) or RR_RESAMPLE_MULTINOMIAL (the reference's own PF / MCL resampler:
        every draw searched by the shard whose CDF interval holds it, the source stored straight into the owner's slab)"""
        L = _ffi.lib()
        self.L, self.rank, self.world, self.n_local = L, rank, world, n_local
        cfg = _ffi.PfConfig(n_local, resample_threshold, range_noise, velocity_noise, yaw_rate_noise, dt)
        opt = _ffi.PfOptions()
        L.rr_pf_options_default(C.byref(opt))
        opt.device, opt.seed = device, seed
        opt.resample_scheme, opt.resample_gate, opt.likelihood_mode = scheme, gate, likelihood_mode
        opt.first_global_index, opt.n_global = rank * n_local, n_local * world
        self.h = C.c_void_p()
        if initial_state is None:
            self._check(L.rr_pf_create(C.byref(cfg), C.byref(opt), C.byref(self.h)))
        else:
            st = np.ascontiguousarray(initial_state, dtype=np.float64)
            self._check(L.rr_pf_create_with_state(C.byref(cfg), C.byref(opt), st.ctypes.data_as(C.POINTER(C.c_double)), C.byref(self.h)))

    def _check(self, status: int) -> None:
        if status != _ffi.RR_OK:
            kind = RoboticsError.invalid_parameter if status == _ffi.RR_INVALID_PARAMETER else RoboticsError.runtime
            raise kind(_ffi.last_error())

    def connect_ipc(self, allgather) -> None:
        blob = (C.c_uint8 * _ffi.RR_P2P_HANDLE_BYTES)()
        self._check(self.L.rr_pf_p2p_export(self.h, blob))
        parts = allgather(bytes(blob))
        allb = (C.c_uint8 * (_ffi.RR_P2P_HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(parts))
        self._check(self.L.rr_pf_p2p_connect(self.h, allb, self.world, self.rank))

    @staticmethod
    def link_local(shards) -> None:
        L = _ffi.lib()
        arr = (C.c_void_p * len(shards))(*[s.h for s in shards])
        status = L.rr_pf_p2p_connect_local(arr, len(shards))
        if status != _ffi.RR_OK:
            raise RoboticsError.runtime(_ffi.last_error())

    def step(self, u, obs) -> None:
        u = np.ascontiguousarray(u, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_step_p2p(self.h, u.ctypes.data_as(dp), obs.ctypes.data_as(dp) if obs.size else None, obs.shape[0]))

    def step_unfused(self, u, obs) -> None:
        """the same step with every phase as its own launch (A/B measurement)"""
        u = np.ascontiguousarray(u, dtype=np.float64)
        obs = np.ascontiguousarray(obs, dtype=np.float64).reshape(-1, 3)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_shard_step_p2p_unfused(self.h, u.ctypes.data_as(dp), obs.ctypes.data_as(dp) if obs.size else None, obs.shape[0]))

    def timed_out(self) -> bool:
        out = C.c_int32()
        self._check(self.L.rr_pf_p2p_status(self.h, C.byref(out)))
        return bool(out.value)

    def topology(self) -> dict:
        """rr_pf_p2p_topology: ranks, ranks sharing this device, CUs the stream is confined to (0: all), form of the last step"""
        out = (C.c_int32 * 4)()
        self._check(self.L.rr_pf_p2p_topology(self.h, out))
        return dict(n_ranks=out[0], n_sharing=out[1], cu_partition_cus=out[2], last_step={0: "none", 1: "lazy", 2: "eager"}[out[3]])

    def want_estimate(self, on: bool = True) -> None:
        """rr_pf_shard_want_estimate: every step leaves this shard's part of the mean try_step returns (the sums of x, y, yaw, v
        over the sources of the shard's own output slots, added up by the kernel that moves the particles)."""
        self._check(self.L.rr_pf_shard_want_estimate(self.h, 1 if on else 0))

    def estimate_sums(self):
        """(sums[4], N) of the last step: the mean is the sum of every shard's sums over N (one all-reduce of four doubles)."""
        sums, den = np.empty(4), C.c_double()
        self._check(self.L.rr_pf_shard_last_estimate_sums(self.h, sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(den)))
        return sums, den.value

    def particles(self) -> np.ndarray:
        out = np.empty((self.n_local, 5))
        self._check(self.L.rr_pf_get_particles(self.h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def set_particles(self, aos: np.ndarray) -> None:
        """rr_pf_set_particles: this shard's block [n_local][x, y, yaw, v, w] of an injected cloud (parity seam)"""
        a = np.ascontiguousarray(aos, dtype=np.float64)
        if a.shape != (self.n_local, 5):
            raise RoboticsError.invalid_parameter(f"expected an array of shape ({self.n_local}, 5)")
        self._check(self.L.rr_pf_set_particles(self.h, a.ctypes.data_as(C.POINTER(C.c_double))))

    def local_moments(self):
        e, c = np.empty(4), np.empty(16)
        dp = C.POINTER(C.c_double)
        self._check(self.L.rr_pf_estimate(self.h, e.ctypes.data_as(dp)))
        self._check(self.L.rr_pf_covariance(self.h, c.ctypes.data_as(dp)))
        return e, c.reshape(4, 4)

    def synchronize(self) -> None:
        self._check(self.L.rr_pf_synchronize(self.h))

    def profile(self, on: bool) -> None:
        self._check(self.L.rr_pf_profile_enable(self.h, int(on)))
        if on:
            self._check(self.L.rr_pf_profile_reset(self.h))

    def profile_read(self) -> dict:
        out = {}
        for k in range(_ffi.RR_K_COUNT):
            n, ms = C.c_uint64(), C.c_double()
            self._check(self.L.rr_pf_profile_read(self.h, k, C.byref(n), C.byref(ms)))
            out[self.L.rr_pf_kernel_name(k).decode()] = (n.value, ms.value)
        return out

    def close(self) -> None:
        if self.h:
            self.L.rr_pf_destroy(self.h)
            self.h = None


def gloo_allgather(dist):
    def allgather(raw: bytes):
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, raw)
        return out

    return allgather


def gloo_exchange(dist):
    """broadcast a bytes object from rank 0 over an initialised torch.distributed group"""

    def exchange(raw: bytes) -> bytes:
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    return exchange


class TorchShard:
    """Third transport: the host-orchestrated ``ShardedLocalizer`` over ``HipShard`` with
    torch.distributed's NCCL (= RCCL) backend doing the collectives.  Slowest of the three (every
    phase returns to Python); bench.py only uses it if the native RCCL transport cannot be set up."""

    def __init__(self, rank, world, device, n_local, dist, **kw):
        self.hs = HipShard(rank, world, device, n_local, **kw)
        try:
            self.group = dist.new_group(backend="nccl")  # collective: every rank constructs its TorchShard together
            # the communicator is only made by the first collective: run one now, so that a machine on which RCCL cannot
            # connect the ranks fails HERE (bench_sharded's ladder then moves on) and not in the middle of a step
            torch = self.hs.torch
            probe = torch.ones(1, dtype=torch.float64, device=f"cuda:{device}")
            dist.all_reduce(probe, group=self.group)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"trial all-reduce over {world} ranks returned {probe.item()}")
        except Exception:
            self.hs.close()
            raise
        self.loc = ShardedLocalizer(self.hs, dist, group=self.group)

    def step(self, u, obs) -> None:
        self.loc.step(u, obs)

    def particles(self) -> np.ndarray:
        return self.hs.particles()

    def synchronize(self) -> None:
        self.hs.synchronize()
        self.hs.torch.cuda.synchronize()

    def profile(self, on) -> None:
        self.hs.profile(on)

    def profile_read(self) -> dict:
        return self.hs.profile_read()

    def estimate(self):
        return self.loc.estimate()

    def migrated(self) -> int:
        M = self.loc.last_matrix
        return int(M.sum() - np.trace(M)) if M is not None else 0

    def close(self) -> None:
        self.hs.close()


def _diagnose_reference_split(rank, world, dist, digests, odd, whole, whole_all, cfg, scheme, lik, device, u, obs, log):
    """The same unsharded filter, the same inputs, different particles on some ranks (round 6: eight processes on one device):
    say what differs -- which rows, which counters -- and replay the filter step by step, alone, to see whether it repeats."""
    import hashlib

    import rust_robotics_amd.localization as loc

    major = max(set(digests), key=digests.count)
    src = digests.index(major)
    path = f"/tmp/rr_reference_split_{os.environ.get('MASTER_PORT', '0')}.npy"
    if rank == src:
        np.save(path, whole_all)
    dist.barrier()
    try:
        fs = whole.fixed_sums()
        finger = dict(counters=list(whole.counters()), n_eff=whole.n_eff(), plan=list(whole.plan_stats()),
                      sums={k: int(getattr(fs, k)) for k, _ in fs._fields_ if isinstance(getattr(fs, k), int)})
    except Exception as e:  # noqa: BLE001
        finger = dict(error=str(e))
    what = dict(rank=rank, odd=rank in odd, finger=finger)
    if rank in odd:
        ref = np.load(path)
        a, b = whole_all.view(np.uint64), ref.view(np.uint64)
        bad = np.flatnonzero((a != b).any(axis=1))
        what.update(differing=int(bad.size), of=int(a.shape[0]), first=int(bad[0]), last=int(bad[-1]),
                    runs=int(np.flatnonzero(np.diff(bad) > 1).size + 1),
                    per_column=[int(np.count_nonzero(a[:, k] != b[:, k])) for k in range(a.shape[1])],
                    same_multiset_of_v=bool(np.array_equal(np.sort(whole_all[:, 3]), np.sort(ref[:, 3]))),
                    max_abs=[float(np.nanmax(np.abs(whole_all[:, k] - ref[:, k]))) for k in range(a.shape[1])])
    dist.barrier()
    if rank == src:
        os.unlink(path)
    # the replay: alone (rank after rank would take long: all ranks at once, but synchronised after every step)
    per_step = []
    replay = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=device, resample_scheme=scheme, likelihood_mode=lik)
    for t in range(len(obs)):
        replay.step_async(u, obs[t])
        per_step.append(hashlib.blake2b(memoryview(np.ascontiguousarray(replay.get_particles_array())).cast("B"), digest_size=6).hexdigest())
    final = hashlib.blake2b(memoryview(np.ascontiguousarray(replay.get_particles_array())).cast("B"), digest_size=12).hexdigest()
    what.update(replay_equals_majority=final == major, replay_equals_own_first_pass=final == digests[rank], replay_per_step=per_step)
    del replay
    everything = [None] * world
    dist.all_gather_object(everything, what)
    if rank == 0:
        import json

        for w in everything:
            log("REFERENCE SPLIT " + json.dumps(w))


def bench_sharded(rank, world, local_rank, n_local, L, K, W, obs_list, scheme, lik, transport="auto"):
    """bench.py's N > 1 leg: weak scaling, n_local particles per GPU, barrier + synchronize on
    both sides of the K timed steps, MAX over ranks.  torch.distributed (gloo) only bootstraps
    (communicator id, IPC handles) and provides the timing barrier; the step runs inside the library.

    Transport ladder (every decision is agreed on by all ranks):
      reference transport = native RCCL (``rr_pf_shard_step``), else torch.distributed/NCCL
        (``TorchShard``), else none;
      peer-to-peer (``rr_pf_shard_step_p2p``) is timed iff it connects on every rank AND reproduces,
        bit for bit and without a timed-out wait, a dozen steps of the reference transport -- or, when
        there is no reference transport, of an UNSHARDED filter of all n_local * world particles run on
        every rank -- on THIS machine: a run-time proof of the cross-GPU hand-off before anything is timed;
      otherwise the reference transport is timed; if a peer wait gives up inside the timed region the
        region is repeated on the reference transport.
    With no working transport at all the function raises (bench.py reports that instead of a number)."""
    import torch
    import torch.distributed as dist

    multinomial = scheme != _ffi.RR_RESAMPLE_SYSTEMATIC  # (iid draws: the peer-to-peer transport stores every served slot straight
    # into its owner's slab -- rr_pf_shard_step_p2p of multinomial shards, round 6; RCCL / torch move them in an all-to-all)
    own_group = not dist.is_initialized()  # bench.py keeps one gloo group for all its legs
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")  # a lone rank started without a launcher
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    u = [1.0, 0.1]
    kw = dict(seed=1, likelihood_mode=lik, initial_state=[0.0, 0.0, 0.0, 1.0])
    if multinomial:
        kw["scheme"] = _ffi.RR_RESAMPLE_MULTINOMIAL
    notes = []
    # observation schedule (time only moves forward for every filter): [0, V) validation, [V, T0) warm-up (W steps plus
    # whatever surplus bench.py handed over), [T0, T0 + K) timed region, [T0 + K, T0 + 2K) instrumented continuation
    V = min(12, len(obs_list))
    T0 = max(len(obs_list) - 2 * K, V + W)
    assert T0 + 2 * K <= len(obs_list), "bench.py must hand over at least 12 + W + 2K steps of observations"

    t_start = time.time()

    def log(msg):  # progress on stderr: where a rank is when something takes long
        sys.stderr.write(f"[sharded rank {rank} +{time.time() - t_start:6.1f}s] {msg}\n")
        sys.stderr.flush()

    def agree(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def attempt(name, make):
        """construct a transport on every rank; keep it only if every rank succeeded"""
        obj, err = None, None
        log(f"trying the {name}")
        try:
            obj = make()
        except Exception as e:  # noqa: BLE001 -- any failure means "next rung of the ladder"
            err = f"{type(e).__name__}: {e}"
        if agree(obj is not None):
            return obj
        notes.append(f"{name} unavailable" + (f" ({err})" if err else " (failed on another rank)"))
        if obj is not None:
            obj.close()
        return None

    if os.environ.get("RR_BENCH_SIMULATE_NO_TRANSPORT"):  # test hook for bench.py's last-resort line
        raise RuntimeError("no working sharded transport on this machine: simulated")
    ref, ref_kind = None, None
    if transport in ("auto", "p2p", "rccl"):  # "p2p-only" skips the reference transports (validation against the unsharded filter)
        # librccl must load on EVERY rank before anybody enters the collective communicator set-up
        probe = (C.c_uint8 * 128)()
        if agree(_ffi.lib().rr_comm_unique_id(probe) == _ffi.RR_OK):
            ref = attempt("native RCCL transport", lambda: NativeShard(rank, world, local_rank, n_local, gloo_exchange(dist), **kw))
        else:
            notes.append("librccl could not be loaded on every rank")
        ref_kind = "rccl" if ref else None
    if ref is None and transport in ("auto", "p2p", "rccl", "torch"):
        ref = attempt("torch.distributed NCCL transport", lambda: TorchShard(rank, world, local_rank, n_local, dist, **kw))
        ref_kind = "torch.distributed nccl (host-orchestrated phases)" if ref else None

    p2p, use_p2p = None, False
    if transport in ("auto", "p2p", "p2p-only"):
        def make_p2p():
            s = P2PShard(rank, world, local_rank, n_local, **kw)  # (kw carries the scheme of multinomial shards)
            s.connect_ipc(gloo_allgather(dist))
            return s

        p2p = attempt("peer-to-peer transport", make_p2p)
    # ---- run-time proof on THIS machine, before anything is timed: V steps of every transport that could be set up, next to the
    # UNSHARDED filter of all n_local * world particles on every rank (systematic shards; it fits one GPU up to tens of millions
    # of particles) -- both native transports share the window kernels, so agreeing with each other would not be enough
    validated_ref = False
    if (ref is not None or p2p is not None) and n_local * world <= (1 << 25):
        import rust_robotics_amd.localization as loc

        cfg = loc.MonteCarloLocalizationConfig(min_particles=n_local * world, max_particles=n_local * world)
        whole = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=local_rank,
                                                           resample_scheme=scheme, likelihood_mode=lik)
        alive = True
        for t in range(V):
            if p2p is not None and alive:
                p2p.step(u, obs_list[t])
            if ref is not None:
                ref.step(u, obs_list[t])
            whole.step_async(u, obs_list[t])
            if t == 0 and p2p is not None:  # a dead transport shows on the first exchange: stop before it costs more
                alive = agree(not p2p.timed_out())
        whole_all = np.ascontiguousarray(whole.get_particles_array())
        exp = np.ascontiguousarray(whole_all[rank * n_local:(rank + 1) * n_local]).view(np.uint64)
        # every rank ran the SAME unsharded filter: if their results differ from each other, the reference is at fault, not a transport
        import hashlib

        digests = [None] * world
        dist.all_gather_object(digests, hashlib.blake2b(memoryview(whole_all).cast("B"), digest_size=12).hexdigest())
        if len(set(digests)) > 1:
            odd = [g for g in range(world) if digests[g] != max(set(digests), key=digests.count)]
            notes.append(f"THE UNSHARDED REFERENCE FILTER DIFFERS BETWEEN RANKS (ranks {odd} against the majority): a validation failure on those ranks is the reference's")
            log(notes[-1])
            _diagnose_reference_split(rank, world, dist, digests, odd, whole, whole_all, cfg, scheme, lik, local_rank, u, obs_list[:V], log)
        reps = int(os.environ.get("RR_BENCH_VALIDATE_REPEAT", "0"))  # the hunt: the same unsharded filter again, on every rank at once
        for rep in range(reps):
            again = loc.MonteCarloLocalizer.with_initial_state([0.0, 0.0, 0.0, 1.0], cfg, seed=1, device=local_rank, resample_scheme=scheme, likelihood_mode=lik)
            dist.barrier()
            for t in range(V):
                again.step_async(u, obs_list[t])
            arr = np.ascontiguousarray(again.get_particles_array())
            ds = [None] * world
            dist.all_gather_object(ds, hashlib.blake2b(memoryview(arr).cast("B"), digest_size=12).hexdigest())
            if rank == 0:
                log(f"VALIDATE_REPEAT {rep}: " + ("all ranks equal" if len(set(ds)) == 1 else f"SPLIT {ds}") + f"; equal to the first pass's majority: {ds.count(max(set(digests), key=digests.count))} of {world}")
            if len(set(ds)) > 1:
                _diagnose_reference_split(rank, world, dist, ds, [g for g in range(world) if ds[g] != max(set(ds), key=ds.count)], again, arr, cfg, scheme, lik,
                                          local_rank, u, obs_list[:V], log)
            del again, arr
        del whole, whole_all

        def check(name, shard, dead):
            if dead:
                why, same = f"rank {rank}: a wait for a peer's flag gave up", False
            else:
                got = np.ascontiguousarray(shard.particles()).view(np.uint64)
                same = np.array_equal(got, exp)
                why = ""
                if not same:
                    bad = np.nonzero(np.any(got != exp, axis=1))[0]
                    cols = [int(np.count_nonzero(got[:, k] != exp[:, k])) for k in range(got.shape[1])]
                    why = (f"rank {rank}: {bad.size} of {n_local} particles differ; rows first {bad[:4].tolist()} last {bad[-4:].tolist()}, "
                           f"per column {cols}")
                    try:  # is what was read stable?  (a second read after a full synchronisation: the same / now equal to the reference)
                        shard.synchronize()
                        again = np.ascontiguousarray(shard.particles()).view(np.uint64)
                        why += f"; a second read {'repeats the first' if np.array_equal(again, got) else 'DIFFERS from the first'}" + \
                               (" and now EQUALS the reference" if np.array_equal(again, exp) else "")
                    except Exception as e:  # noqa: BLE001
                        why += f"; second read failed: {e}"
                    log(f"VALIDATION MISMATCH ({name}): " + why)  # every rank says what it saw (stderr)
            ok = agree(same)
            notes.append(f"{name} validated bit-identical to the unsharded filter of all particles over {V} steps" if ok else
                         f"{name} FAILED validation against the unsharded filter of all particles" + (f" ({why})" if why else " (on another rank)"))
            log(f"{name} " + ("validated" if ok else "failed validation"))
            return ok

        if ref is not None:
            validated_ref = check(f"{ref_kind} transport", ref, False)
            if not validated_ref:
                ref.close()
                ref, ref_kind = None, None
        if p2p is not None:
            dead = (not alive) or p2p.timed_out()
            use_p2p = check("peer-to-peer transport", p2p, dead)
    elif p2p is not None and ref is not None:
        # (multinomial shards / a filter too large for one GPU: the peer-to-peer transport against the reference transport)
        alive = True
        for t in range(V):
            p2p.step(u, obs_list[t])
            ref.step(u, obs_list[t])
            if t == 0:
                alive = agree(not p2p.timed_out())
                if not alive:
                    break
        same = False
        if alive and not p2p.timed_out():
            same = np.array_equal(np.ascontiguousarray(p2p.particles()).view(np.uint64), np.ascontiguousarray(ref.particles()).view(np.uint64))
        use_p2p = agree(same)
        validated_ref = True  # (its V steps are done)
        notes.append(f"peer-to-peer transport " + (f"validated bit-identical to the {ref_kind} transport over {V} steps" if use_p2p else
                                                   f"FAILED validation against the {ref_kind} transport"))
    if not use_p2p and ref is None:
        if p2p is not None:  # leave nothing behind on the device: its stream drained, the spin permit and the IPC mappings released
            try:
                p2p.close()
            except RoboticsError:
                pass
        dist.barrier()
        raise RuntimeError("no working sharded transport on this machine: " + "; ".join(notes))


    trouble = []  # errors of THIS rank inside a region: remembered, not raised -- every rank keeps to the same sequence of
                  # collectives and the ranks decide together afterwards (a rank that left early would leave the others in a barrier)

    def quiet_sync(shard):
        try:
            shard.synchronize()
        except RoboticsError:  # a latched peer-wait timeout (rr_pf_synchronize reports it): the ladder asks
            pass               # p2p.timed_out() on every rank right after the region and decides collectively

    def quiet_step(shard, obs):
        try:
            shard.step(u, obs)
        except RoboticsError as e:
            if not trouble:
                trouble.append(f"rank {rank}: {e}")
                log(f"a step failed on this rank: {e}")

    def fence(shard):
        quiet_sync(shard)
        torch.cuda.synchronize()
        dist.barrier()
        quiet_sync(shard)
        torch.cuda.synchronize()

    def timed_region(shard):
        # W warm-up steps plus 64 more: in a process that has torch's HIP context loaded the host enqueues the
        # first ~50 steps of a large filter an order of magnitude slower than later ones
        for t in range(V, T0):
            quiet_step(shard, obs_list[t])
        fence(shard)
        # ... and that stall lands at an unpredictable launch count (a few milliseconds, once): keep warming up in blocks
        # of 40 steps (replaying the last warm-up observations: a tracking filter does not mind) until two blocks in a
        # row run at the pace of the fastest one seen, at most 8 blocks
        best, good = None, 0
        for _ in range(8):
            tb = time.perf_counter()
            for t in range(max(V, T0 - 40), T0):
                quiet_step(shard, obs_list[t])
            fence(shard)
            blk = torch.tensor([time.perf_counter() - tb], dtype=torch.float64)
            dist.all_reduce(blk, op=dist.ReduceOp.MAX)
            blk = float(blk.item())
            best = blk if best is None else min(best, blk)
            good = good + 1 if blk <= 1.25 * best else 0
            if good >= 2:
                break
        # the sharded step is launch-rate sensitive (7 launches in ~85 us): nothing is instrumented inside the
        # timed region; the kernel times of the instrumented re-run below feed `roofline`
        t0 = time.perf_counter()
        for t in range(T0, T0 + K):
            quiet_step(shard, obs_list[t])
        fence(shard)
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    shard = p2p if use_p2p else ref
    log("timed region on " + ("the peer-to-peer transport" if use_p2p else f"the {ref_kind} transport"))
    # the reference's try_step returns the mean every step: over the peer-to-peer transport every step leaves this shard's part of
    # it (the sums over the sources of its own slots, added up by the kernel that moves the particles); the all-reduce of the four
    # doubles is only done when somebody wants the value -- below, once, after the timed steps
    est_every_step = False
    if not multinomial and hasattr(shard, "want_estimate"):  # (the peer-to-peer and the native RCCL transport)
        ok = True
        try:
            shard.want_estimate(True)
        except RoboticsError as e:
            ok = False
            log(f"rr_pf_shard_want_estimate failed on this rank: {e}")
        est_every_step = agree(ok)
        if not est_every_step:
            try:
                shard.want_estimate(False)
            except RoboticsError:
                pass
    if not validated_ref and not use_p2p:  # no validation ran on it: the reference transport has not seen the first V steps yet
        for t in range(V):
            ref.step(u, obs_list[t])
    seconds = timed_region(shard)
    timed_out = False
    if use_p2p and not agree(not p2p.timed_out() and not trouble):
        timed_out = True
        notes.append("a peer wait gave up inside the timed region" + (f" ({trouble[0]})" if trouble else ""))
        log(notes[-1])
        del trouble[:]
        if ref is not None:  # repeat the region on the reference transport
            use_p2p, shard = False, ref
            seconds = timed_region(shard)
            notes.append(f"timed region repeated on the {ref_kind} transport")
    est, ok = None, False
    if est_every_step and not timed_out:
        try:
            sums, den = shard.estimate_sums()
            ok = True
        except RoboticsError as e:
            log(f"rr_pf_shard_last_estimate_sums failed on this rank: {e}")
            sums, den = np.zeros(4), 1.0
        if agree(ok):
            tot = torch.tensor(sums, dtype=torch.float64)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            est = (tot / den).numpy()
        else:
            est_every_step = False
        try:
            shard.want_estimate(False)  # (the instrumented continuation below times the kernels of the plain step)
        except RoboticsError:
            pass
    elif est_every_step:
        est_every_step = False  # (the region was repeated on the reference transport without it)
    if use_p2p:
        if est is None:
            est, _ = p2p.local_moments()  # local estimate of this rank's block (all blocks are samples of the same posterior)
        moved = -1
    else:
        if est is None:
            est, _ = shard.estimate()
        moved = shard.migrated()
    shard.profile(True)
    t1 = time.perf_counter()
    for t in range(T0 + K, T0 + 2 * K):
        shard.step(u, obs_list[t])
    shard.synchronize()
    dt_instr = time.perf_counter() - t1
    prof = shard.profile_read()
    shard.profile(False)
    dist.barrier()
    # who took part, as seen from rank 0: every rank reports the device it ran on (PCI address) and how many peers its transport
    # had connected (peer-to-peer: ranks whose mailbox / inbox are mapped, this one included; RCCL: the communicator's size)
    bus = C.create_string_buffer(64)
    if _ffi.lib().rr_device_pci_bus_id(local_rank, bus, 64) != _ffi.RR_OK:
        bus.value = b"?"
    peers = world if (use_p2p or ref is not None) else 1
    seen = [None] * world
    dist.all_gather_object(seen, (rank, bus.value.decode(), peers))
    devices = sorted({b for _, b, _ in seen})
    ranks_seen = dict(ranks=len({r for r, _, _ in seen}), peers_connected_min=min(p for _, _, p in seen), distinct_devices=len(devices),
                      devices=devices)
    if p2p is not None:
        p2p.close()
    if ref is not None:
        ref.close()
    if own_group:
        dist.destroy_process_group()
    return dict(seconds=seconds, seconds_instrumented=dt_instr, kernels=prof, estimate=[float(a) for a in est], dominant=None,
                migrated_particles_last_step=moved, transport="p2p (xGMI, device-initiated)" if use_p2p else ref_kind,
                transport_note="; ".join(notes), p2p_timed_out=bool(timed_out), estimate_every_step=bool(est_every_step),
                ranks_seen=ranks_seen)
