// rccl_comm.cpp -- libnalu_gpu_rccl.so: nalu_gpu_comm callbacks on RCCL (include/nalu_gpu_rccl.h).
// Host-only translation unit; the callbacks themselves are include/RcclComm.h.
#include "nalu_gpu_rccl.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "RcclComm.h"

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

struct nalu_gpu_rccl_world {
  ncclComm_t comm = nullptr;
  int rank = 0, nRanks = 1, device = 0;
};
struct nalu_gpu_rccl_comm {
  sierra::nalu::gpu::RcclComm *impl = nullptr;
  nalu_gpu_linsys *h = nullptr;
};

static_assert(sizeof(ncclUniqueId) <= NALU_GPU_RCCL_ID_BYTES, "unique id does not fit");

namespace {
struct DeviceScope {  // the caller's current device is restored on the way out
  int prev = -1;
  explicit DeviceScope(int d)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(d);
  }
  ~DeviceScope()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace

extern "C" const char *nalu_gpu_rccl_last_error(void) { return g_err.c_str(); }

extern "C" int nalu_gpu_rccl_get_unique_id(void *id_out)
{
  if (!id_out) return fail(NALU_GPU_EINVAL, "id_out is null");
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail(NALU_GPU_EHIP, "ncclGetUniqueId: %s", ncclGetErrorString(r));
  std::memset(id_out, 0, NALU_GPU_RCCL_ID_BYTES);
  std::memcpy(id_out, &id, sizeof(id));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_rccl_world_create(const void *unique_id, int rank, int n_ranks, int device,
                                          nalu_gpu_rccl_world **out)
{
  if (!unique_id || !out) return fail(NALU_GPU_EINVAL, "null argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(NALU_GPU_EINVAL, "rank %d of %d", rank, n_ranks);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(NALU_GPU_EINVAL, "device %d of %d", device, ndev);
  DeviceScope scope(device);
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  nalu_gpu_rccl_world *w = new nalu_gpu_rccl_world();
  w->rank = rank;
  w->nRanks = n_ranks;
  w->device = device;
  const ncclResult_t r = ncclCommInitRank(&w->comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    delete w;
    return fail(NALU_GPU_EHIP, "ncclCommInitRank: %s", ncclGetErrorString(r));
  }
  *out = w;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_rccl_world_destroy(nalu_gpu_rccl_world *w)
{
  if (!w) return NALU_GPU_OK;
  DeviceScope scope(w->device);
  if (w->comm) (void)ncclCommDestroy(w->comm);
  delete w;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_rccl_attach(nalu_gpu_rccl_world *w, nalu_gpu_linsys *h, nalu_gpu_rccl_comm **out)
{
  return nalu_gpu_rccl_attach_flags(w, h, 0, out);
}

extern "C" int nalu_gpu_rccl_attach_flags(nalu_gpu_rccl_world *w, nalu_gpu_linsys *h, int flags,
                                          nalu_gpu_rccl_comm **out)
{
  if (!w || !h || !out) return fail(NALU_GPU_EINVAL, "null argument");
  const bool selfOk = (flags & NALU_GPU_RCCL_ALLOW_SELF_NEIGHBOUR) != 0;
  int nNbr = 0;
  const int *ranks = nullptr, *so = nullptr, *ro = nullptr;
  if (nalu_gpu_get_halo_plan(h, &nNbr, &ranks, &so, &ro) != NALU_GPU_OK)
    return fail(NALU_GPU_ESTATE, "%s", nalu_gpu_last_error());
  for (int p = 0; p < nNbr; ++p)
    if (ranks[p] < 0 || ranks[p] >= w->nRanks || (ranks[p] == w->rank && !selfOk))
      return fail(NALU_GPU_EINVAL, "halo neighbour %d is rank %d: not another rank of this world of %d "
                  "(NALU_GPU_RCCL_ALLOW_SELF_NEIGHBOUR admits the rank itself)", p, ranks[p], w->nRanks);
  DeviceScope scope(w->device);
  nalu_gpu_rccl_comm *c = new nalu_gpu_rccl_comm();
  c->h = h;
  try {
    const std::vector<int> nb(ranks, ranks + nNbr);
    const std::vector<int> s(so, so + (nNbr ? nNbr + 1 : 0)), r(ro, ro + (nNbr ? nNbr + 1 : 0));
    c->impl = new sierra::nalu::gpu::RcclComm(h, w->comm, nb, nNbr ? s : std::vector<int>(1, 0),
                                              nNbr ? r : std::vector<int>(1, 0));
  } catch (const std::exception &e) {
    delete c;
    return fail(NALU_GPU_EHIP, "%s", e.what());
  }
  *out = c;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_rccl_detach(nalu_gpu_rccl_comm *c)
{
  if (!c) return NALU_GPU_OK;
  (void)nalu_gpu_set_comm(c->h, nullptr);
  delete c->impl;
  delete c;
  return NALU_GPU_OK;
}
