// RcclComm.h -- the three communication callbacks of nalu_gpu_comm (nalu_gpu.h) on RCCL, for a
// C++ host with one process per GPU (header only; link librccl and libamdhip64).
//
// In the reference these are Tpetra's Import inside CrsMatrix::apply and the all-reduce inside
// dot / norm2 (call sites src/LinearSolver.C:164,172,224; src/TpetraLinearSystem.C:1756).  The
// library packs / unpacks / reduces on the device; this class only moves the bytes:
//   allreduce_sum   ncclAllReduce, in place on the library's reduce buffer
//   exchange_begin  one grouped ncclSend / ncclRecv pair per neighbour, on a side stream that
//                   waits for the pack kernel, so that the interior SpMV overlaps it
//   exchange_end    makes the library's stream wait for the exchange
// xGMI is point to point: a z-slab partition talks to two neighbours, one link each.
#ifndef RcclComm_h
#define RcclComm_h

#include "nalu_gpu.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>
#include <vector>

namespace sierra {
namespace nalu {
namespace gpu {

class RcclComm
{
public:
  // comm: an initialised communicator whose rank numbering is the one of the halo plan.
  // Call after finalizeLinearSystem() and nalu_gpu_set_halo_plan (the buffers exist by then).
  RcclComm(nalu_gpu_linsys *h, ncclComm_t comm, const std::vector<int> &neighborRanks,
           const std::vector<int> &sendOffsets, const std::vector<int> &recvOffsets)
    : comm_(comm), nbr_(neighborRanks), so_(sendOffsets), ro_(recvOffsets), side_(nullptr), packed_(nullptr),
      arrived_(nullptr)
  {
    if (so_.size() != nbr_.size() + 1 || ro_.size() != nbr_.size() + 1)
      throw std::runtime_error("RcclComm: offsets need one entry per neighbour plus one");
    int64_t sc = 0, rc = 0;
    int cap = 0;
    if (nalu_gpu_get_comm_buffers(h, &send_, &sc, &recv_, &rc, &reduce_, &cap) != NALU_GPU_OK)
      throw std::runtime_error(std::string("RcclComm: ") + nalu_gpu_last_error());
    if (!nbr_.empty() && (so_.back() > sc || ro_.back() > rc))
      throw std::runtime_error("RcclComm: the halo plan is larger than the library's buffers");
    hip(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking), "hipStreamCreate");
    hip(hipEventCreateWithFlags(&packed_, hipEventDisableTiming), "hipEventCreate");
    hip(hipEventCreateWithFlags(&arrived_, hipEventDisableTiming), "hipEventCreate");
    table_.ctx = this;
    table_.allreduce_sum = &RcclComm::allreduce;
    table_.exchange_begin = &RcclComm::begin;
    table_.exchange_end = &RcclComm::end;
    table_.exchange_rows = &RcclComm::rows;
    if (nalu_gpu_set_comm(h, &table_) != NALU_GPU_OK)
      throw std::runtime_error(std::string("RcclComm: ") + nalu_gpu_last_error());
  }
  ~RcclComm()
  {
    if (side_) {
      (void)hipStreamSynchronize(side_);
      (void)hipStreamDestroy(side_);
    }
    if (packed_) (void)hipEventDestroy(packed_);
    if (arrived_) (void)hipEventDestroy(arrived_);
  }
  RcclComm(const RcclComm &) = delete;
  RcclComm &operator=(const RcclComm &) = delete;

private:
  static void hip(hipError_t e, const char *what)
  {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  // callbacks must not throw through the C ABI: report with a non-zero return
  static int allreduce(void *ctx, int count, void *stream)
  {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    return ncclAllReduce(c->reduce_, c->reduce_, (size_t)count, ncclDouble, ncclSum, c->comm_,
                         (hipStream_t)stream) == ncclSuccess ? 0 : 1;
  }
  static int begin(void *ctx, void *stream)
  {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (c->nbr_.empty()) return 0;
    if (hipEventRecord(c->packed_, (hipStream_t)stream) != hipSuccess) return 1;
    if (hipStreamWaitEvent(c->side_, c->packed_, 0) != hipSuccess) return 1;
    bool ok = ncclGroupStart() == ncclSuccess;
    for (size_t p = 0; ok && p < c->nbr_.size(); ++p) {
      const size_t ns = (size_t)(c->so_[p + 1] - c->so_[p]), nr = (size_t)(c->ro_[p + 1] - c->ro_[p]);
      if (ns) ok = ncclSend(c->send_ + c->so_[p], ns, ncclDouble, c->nbr_[p], c->comm_, c->side_) == ncclSuccess;
      if (ok && nr) ok = ncclRecv(c->recv_ + c->ro_[p], nr, ncclDouble, c->nbr_[p], c->comm_, c->side_) == ncclSuccess;
    }
    ok = (ncclGroupEnd() == ncclSuccess) && ok;
    if (!ok) return 1;
    return hipEventRecord(c->arrived_, c->side_) == hipSuccess ? 0 : 1;
  }
  static int end(void *ctx, void *stream)
  {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (c->nbr_.empty()) return 0;
    return hipStreamWaitEvent((hipStream_t)stream, c->arrived_, 0) == hipSuccess ? 0 : 1;
  }

  // loadComplete's row export (export mode): grouped send / recv of caller-given buffers, on the
  // library's own stream (nothing to overlap with)
  static int rows(void *ctx, const double *send, const int64_t *so, double *recv, const int64_t *ro, void *stream)
  {
    RcclComm *c = static_cast<RcclComm *>(ctx);
    if (c->nbr_.empty()) return 0;
    bool ok = ncclGroupStart() == ncclSuccess;
    for (size_t p = 0; ok && p < c->nbr_.size(); ++p) {
      const size_t ns = (size_t)(so[p + 1] - so[p]), nr = (size_t)(ro[p + 1] - ro[p]);
      if (ns) ok = ncclSend(send + so[p], ns, ncclDouble, c->nbr_[p], c->comm_, (hipStream_t)stream) == ncclSuccess;
      if (ok && nr) ok = ncclRecv(recv + ro[p], nr, ncclDouble, c->nbr_[p], c->comm_, (hipStream_t)stream) == ncclSuccess;
    }
    return (ncclGroupEnd() == ncclSuccess) && ok ? 0 : 1;
  }

  ncclComm_t comm_;
  std::vector<int> nbr_, so_, ro_;
  double *send_ = nullptr, *recv_ = nullptr, *reduce_ = nullptr;
  hipStream_t side_;
  hipEvent_t packed_, arrived_;
  nalu_gpu_comm table_;
};

}  // namespace gpu
}  // namespace nalu
}  // namespace sierra

#endif
