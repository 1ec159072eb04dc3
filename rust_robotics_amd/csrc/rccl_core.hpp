// rccl_core.hpp -- RCCL loaded at run time (dlopen: the library links against nothing it may not find) and
// the communicator object of include/rr_pf.h "sharded operation", shared by the PF/MCL engine
// (rr_pf_shard_step) and the FastSLAM engine (rr_fs1_shard_update).
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "rr_common.hpp"

namespace rr {

// layout of ncclUniqueId (rccl.h:43): passed by value to ncclCommInitRank
struct ncclUniqueIdPod {
  char internal[128];
};

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ncclUniqueIdPod, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
// nccl.h enum values (stable ABI): ncclUint64 = 5, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2
constexpr int kNcclUint64 = 5, kNcclFloat64 = 8, kNcclMax = 2;

inline Rccl& rccl() {
  static Rccl r;
  return r;
}

inline rr_status rccl_load() {
  Rccl& r = rccl();
  if (r.lib) return RR_OK;
  // librccl.so.1 already mapped by the process (e.g. by torch) is reused; otherwise /opt/rocm/lib's
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!r.lib) return fail(RR_RUNTIME_ERROR, std::string("cannot load librccl: ") + dlerror());
  auto sym = [&](const char* n) { return dlsym(r.lib, n); };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.Send || !r.Recv ||
      !r.GroupStart || !r.GroupEnd) {
    r.lib = nullptr;
    return fail(RR_RUNTIME_ERROR, "librccl is missing a required symbol");
  }
  return RR_OK;
}

#define RR_NCCL_TRY(expr)                                                                                             \
  do {                                                                                                                \
    int _e = (expr);                                                                                                  \
    if (_e != 0)                                                                                                      \
      return ::rr::fail(RR_RUNTIME_ERROR, std::string(#expr) + ": " +                                                \
                                              (::rr::rccl().GetErrorString ? ::rr::rccl().GetErrorString(_e) : "rccl error")); \
  } while (0)

}  // namespace rr

struct rr_comm {
  void* comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
  // device scratch for the collectives
  double* d_wmax = nullptr;      // [1]
  uint64_t* d_sums = nullptr;    // [3]
  uint64_t* d_all = nullptr;     // [n_ranks][3] + 1: every rank's sums, then the bits of the global weight maximum
  uint64_t* h_all = nullptr;     // pinned
  hipEvent_t ev_plan = nullptr;  // "the sums are on the host": the host sizes the segments while the device marks and packs
  double* d_send = nullptr;      // MCL: [cap_send][4]
  double* d_recv = nullptr;      // MCL: [n_local][4]
  size_t cap_send = 0, cap_recv = 0;
  double* d_fsend = nullptr;     // FastSLAM: whole particles, per destination a [plane][count] block
  double* d_frecv = nullptr;
  size_t cap_fsend = 0, cap_frecv = 0;  // in doubles
  uint64_t* d_cnt = nullptr;     // multinomial: [n_ranks] own send counts + [n_ranks][n_ranks] everybody's
  uint64_t* h_cnt = nullptr;     // pinned mirror of the matrix
  double* d_mom = nullptr;       // moments all-gather: [n_ranks][21]
  double* h_mom = nullptr;       // pinned
  std::vector<int64_t> matrix;
};
