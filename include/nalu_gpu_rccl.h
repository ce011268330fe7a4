/*
 * nalu_gpu_rccl.h -- C ABI of libnalu_gpu_rccl.so: the three communication callbacks of
 * nalu_gpu_comm (nalu_gpu.h) implemented natively on RCCL over xGMI, one process per GPU.
 *
 * Replaces, for the hot path, what Tpetra does over MPI in the reference: the Import inside
 * CrsMatrix::apply (src/LinearSolver.C:164,224), stk::mesh::copy_owned_to_shared
 * (src/LinearSystem.C:115) and the all-reduce inside dot / norm2 (src/LinearSolver.C:172,
 * src/TpetraLinearSystem.C:1756).  No host language runs per Krylov iteration: the library calls
 * straight into ncclAllReduce / grouped ncclSend+ncclRecv (include/RcclComm.h holds the
 * callbacks; this library owns the communicator and the side stream).
 *
 * Bootstrap (MPI_Init's role): rank 0 calls nalu_gpu_rccl_get_unique_id and hands the 128 bytes
 * to every rank by any means (MPI_Bcast, torch.distributed, a file); every rank then calls
 * nalu_gpu_rccl_world_create -- a collective -- with the same id.  A world serves any number of
 * linear systems on its device.
 */
#ifndef NALU_GPU_RCCL_H
#define NALU_GPU_RCCL_H
#include "nalu_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NALU_GPU_RCCL_ID_BYTES 128

typedef struct nalu_gpu_rccl_world nalu_gpu_rccl_world;
typedef struct nalu_gpu_rccl_comm nalu_gpu_rccl_comm;

/* ncclGetUniqueId into id_out[NALU_GPU_RCCL_ID_BYTES] */
int nalu_gpu_rccl_get_unique_id(void *id_out);
/* ncclCommInitRank on `device` (collective over the n_ranks processes) */
int nalu_gpu_rccl_world_create(const void *unique_id, int rank, int n_ranks, int device,
                               nalu_gpu_rccl_world **out);
int nalu_gpu_rccl_world_destroy(nalu_gpu_rccl_world *w);
/* Installs the callbacks on a finalized system whose halo plan is set (nalu_gpu_set_halo_plan;
 * neighbour ranks and offsets are read back from the handle).  The comm must be destroyed before
 * the system and the world. */
int nalu_gpu_rccl_attach(nalu_gpu_rccl_world *w, nalu_gpu_linsys *h, nalu_gpu_rccl_comm **out);
/* The same with flags.  NALU_GPU_RCCL_ALLOW_SELF_NEIGHBOUR: a halo neighbour may be the rank itself
 * (ncclSend / ncclRecv to the own rank inside one group; the k-th send to oneself is matched with
 * the k-th receive from oneself).  That is what a mesh that is periodic across the rank's own
 * boundary needs -- and the way the one-GPU test box makes the grouped send / recv, the side
 * stream and its event hand-over really run on RCCL (tests/test_distributed.py, loop-back). */
enum { NALU_GPU_RCCL_ALLOW_SELF_NEIGHBOUR = 1 };
int nalu_gpu_rccl_attach_flags(nalu_gpu_rccl_world *w, nalu_gpu_linsys *h, int flags, nalu_gpu_rccl_comm **out);
int nalu_gpu_rccl_detach(nalu_gpu_rccl_comm *c);
/* message of the calling thread's last failure in this library */
const char *nalu_gpu_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
