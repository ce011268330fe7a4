// geometry.hip -- Hex8 CVFEM geometry on the device (gfx950): dual nodal volume and edge
// area vectors, ComputeGeometryInteriorAlgorithm::execute
// (src/ComputeGeometryInteriorAlgorithm.C:78-216) with the Hex8 sub-control-volume /
// sub-control-surface kernels of src/master_element/MasterElementWork.F (hex_scv_det,
// hex_scs_det), CalculateQuadArea.F, CalculateHexVolume.F + CalculatePolyhedralVolume.F.
//
// The reference loops over elements and adds into nodes / edges.  Here every node and
// every edge GATHERS its contributions in ascending element order (the order of that
// loop), so the sums are deterministic and -- with FP contraction off -- bit-identical to
// the CPU oracle, which is itself pinned bit for bit against the reference's F77 kernels.
// The node->element and edge->element adjacency is built on the device (count, host
// prefix sum, fill); a contribution computes only the one SCV / SCS it needs.  Setup-time
// work, gather-bound; no MFMA.
#include <algorithm>
#include <vector>

#include "internal.hpp"

using namespace nalu_gpu;

namespace {

// 27-point subdivision (MasterElementWork.F:61-210, :283-432); index = F77 index - 1
__device__ __forceinline__ void subdivide_hex(const double (&c)[8][3], double (&p)[27][3])
{
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) p[j][k] = c[j][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p[8][k] = 0.5 * (c[0][k] + c[1][k]);
    p[9][k] = 0.5 * (c[1][k] + c[2][k]);
    p[10][k] = 0.5 * (c[2][k] + c[3][k]);
    p[11][k] = 0.5 * (c[3][k] + c[0][k]);
    p[12][k] = 0.25 * (c[0][k] + c[1][k] + c[2][k] + c[3][k]);
    p[13][k] = 0.5 * (c[4][k] + c[5][k]);
    p[14][k] = 0.5 * (c[5][k] + c[6][k]);
    p[15][k] = 0.5 * (c[6][k] + c[7][k]);
    p[16][k] = 0.5 * (c[7][k] + c[4][k]);
    p[17][k] = 0.25 * (c[4][k] + c[5][k] + c[6][k] + c[7][k]);
    p[18][k] = 0.5 * (c[1][k] + c[5][k]);
    p[19][k] = 0.5 * (c[0][k] + c[4][k]);
    p[20][k] = 0.25 * (c[0][k] + c[1][k] + c[4][k] + c[5][k]);
    p[21][k] = 0.5 * (c[3][k] + c[7][k]);
    p[22][k] = 0.5 * (c[2][k] + c[6][k]);
    p[23][k] = 0.25 * (c[2][k] + c[3][k] + c[6][k] + c[7][k]);
    p[24][k] = 0.25 * (c[1][k] + c[2][k] + c[5][k] + c[6][k]);
    p[25][k] = 0.25 * (c[0][k] + c[3][k] + c[4][k] + c[7][k]);
    p[26][k] = 0.125 * (c[0][k] + c[1][k] + c[2][k] + c[3][k] + c[4][k] + c[5][k] + c[6][k] +
                        c[7][k]);
  }
}

// CalculateQuadArea.F:33-70 (triangle fan around the centroid)
__device__ __forceinline__ void quad_area(const double (&ac)[4][3], double (&area)[3])
{
  constexpr int third[4] = {1, 2, 3, 0};
  double xmid[3], r1[3], r2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) xmid[k] = 0.25 * (ac[0][k] + ac[1][k] + ac[2][k] + ac[3][k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { area[k] = 0.0; r2[k] = ac[0][k] - xmid[k]; }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int iq = third[t];
#pragma unroll
    for (int k = 0; k < 3; ++k) r1[k] = r2[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) r2[k] = ac[iq][k] - xmid[k];
    area[0] = area[0] + r1[1] * r2[2] - r2[1] * r1[2];
    area[1] = area[1] + r1[2] * r2[0] - r2[2] * r1[0];
    area[2] = area[2] + r1[0] * r2[1] - r2[0] * r1[1];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) area[k] = 0.5 * area[k];
}

// CalculateHexVolume.F:27-98 + CalculatePolyhedralVolume.F:41-73 (24 triangular facets)
__device__ __forceinline__ double hex_volume(const double (&vc)[8][3])
{
  constexpr int tri[24][3] = {
      {0, 8, 1},  {8, 2, 1},  {3, 2, 8},  {3, 8, 0},  {6, 9, 5},  {7, 9, 6},
      {4, 9, 7},  {4, 5, 9},  {10, 0, 1}, {5, 10, 1}, {4, 10, 5}, {4, 0, 10},
      {7, 6, 11}, {6, 2, 11}, {2, 3, 11}, {3, 7, 11}, {6, 12, 2}, {5, 12, 6},
      {5, 1, 12}, {1, 2, 12}, {0, 4, 13}, {4, 7, 13}, {7, 3, 13}, {3, 0, 13}};
  double c[14][3];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) c[j][k] = vc[j][k];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c[8][k] = 0.25 * (vc[0][k] + vc[1][k] + vc[2][k] + vc[3][k]);
    c[9][k] = 0.25 * (vc[4][k] + vc[5][k] + vc[6][k] + vc[7][k]);
    c[10][k] = 0.25 * (vc[0][k] + vc[1][k] + vc[5][k] + vc[4][k]);
    c[11][k] = 0.25 * (vc[3][k] + vc[2][k] + vc[6][k] + vc[7][k]);
    c[12][k] = 0.25 * (vc[1][k] + vc[2][k] + vc[6][k] + vc[5][k]);
    c[13][k] = 0.25 * (vc[0][k] + vc[3][k] + vc[7][k] + vc[4][k]);
  }
  double volume = 0.0;
#pragma unroll
  for (int t = 0; t < 24; ++t) {
    const int ip = tri[t][0], iq = tri[t][1], ir = tri[t][2];
    double xf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) xf[k] = c[ip][k] + c[iq][k] + c[ir][k];
    volume = volume +
             xf[0] * ((c[iq][1] - c[ip][1]) * (c[ir][2] - c[ip][2]) -
                      (c[ir][1] - c[ip][1]) * (c[iq][2] - c[ip][2])) -
             xf[1] * ((c[iq][0] - c[ip][0]) * (c[ir][2] - c[ip][2]) -
                      (c[ir][0] - c[ip][0]) * (c[iq][2] - c[ip][2])) +
             xf[2] * ((c[iq][0] - c[ip][0]) * (c[ir][1] - c[ip][1]) -
                      (c[ir][0] - c[ip][0]) * (c[iq][1] - c[ip][1]));
  }
  return volume / 18.0;
}

// one sub-control volume (MasterElementWork.F:271-280,434-452); ICV is a compile-time
// constant so every index into p is static (no scratch memory)
template <int ICV>
__device__ __forceinline__ double scv_volume(const double (&p)[27][3])
{
  constexpr int tab[8][8] = {
      {0, 8, 12, 11, 19, 20, 26, 25},  {8, 1, 9, 12, 20, 18, 24, 26},
      {12, 9, 2, 10, 26, 24, 22, 23},  {11, 12, 10, 3, 25, 26, 23, 21},
      {19, 20, 26, 25, 4, 13, 17, 16}, {20, 18, 24, 26, 13, 5, 14, 17},
      {26, 24, 22, 23, 17, 14, 6, 15}, {25, 26, 23, 21, 16, 17, 15, 7}};
  double s[8][3];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int k = 0; k < 3; ++k) s[n][k] = p[tab[ICV][n]][k];
  return hex_volume(s);
}

// one sub-control surface (MasterElementWork.F:41-54,212-226; the table is 1-based there)
template <int ICS>
__device__ __forceinline__ void scs_area(const double (&p)[27][3], double (&area)[3])
{
  constexpr int tab[12][4] = {{21, 9, 13, 27},  {25, 10, 13, 27}, {11, 13, 27, 24},
                              {12, 26, 27, 13}, {14, 21, 27, 18}, {18, 15, 25, 27},
                              {18, 16, 24, 27}, {17, 18, 27, 26}, {20, 21, 27, 26},
                              {21, 19, 25, 27}, {23, 24, 27, 25}, {22, 26, 27, 24}};
  double s[4][3];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int k = 0; k < 3; ++k) s[n][k] = p[tab[ICS][n] - 1][k];
  quad_area(s, area);
}

__device__ __forceinline__ void load_hex(const int *__restrict__ hex, long long el,
                                         const nalu_gpu_vec3 &x, double (&p)[27][3])
{
  double c[8][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = hex[el * 8 + i];
    c[i][0] = x.x[n]; c[i][1] = x.y[n]; c[i][2] = x.z[n];
  }
  subdivide_hex(c, p);
}

// lrscv_ (src/master_element/Hex8CVFEM.C:263-276): the two element nodes of SCS ne
__constant__ int c_lrscv[24] = {0, 1, 1, 2, 2, 3, 0, 3, 4, 5, 5, 6,
                                6, 7, 4, 7, 0, 4, 1, 5, 2, 6, 3, 7};

// edge {a, b} -> edge id | (bit 31: a is the edge's node 1), or 0xFFFFFFFF when the pair is
// not a registered edge.  Looks the column up in an owned row and reads its tag.
__device__ __forceinline__ unsigned find_edge(const SellView &A, int a, int b)
{
  const bool useA = a < A.nRows;
  const int row = useA ? a : b, col = useA ? b : a;
  if (row >= A.nRows) return TAG_PAD;
  const int slice = row / SELL_C, r = row % SELL_C;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    if (A.cols[pos] == col) {
      const unsigned tag = A.tags[pos];
      if (tag >= TAG_OTHER) return TAG_PAD;
      // tag bit 31: the ROW node is the edge's node 1
      const bool rowIsNode1 = (tag >> 31) != 0;
      const bool aIsNode1 = useA ? rowIsNode1 : !rowIsNode1;
      return (tag & 0x7FFFFFFFu) | (aIsNode1 ? 0x80000000u : 0u);
    }
  }
  return TAG_PAD;
}

// pass 1 (FILL = false): count the elements of every node / edge; pass 2 (FILL = true):
// write the adjacency (element * 8 + local node; (element * 12 + local scs) with the sign
// bit) at cursor positions.  The order inside a list is arbitrary; the gathers sort it.
template <bool FILL>
__global__ __launch_bounds__(256) void k_geom_adjacency(long long nElem, const int *__restrict__ hex,
                                                         SellView A, int nNodes,
                                                         int *__restrict__ nodeCur,
                                                         int *__restrict__ edgeCur,
                                                         unsigned *__restrict__ nodeAdj,
                                                         unsigned *__restrict__ edgeAdj,
                                                         int *__restrict__ err)
{
  const long long el = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (el >= nElem) return;
  int n[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    n[i] = hex[el * 8 + i];
    if (n[i] < 0 || n[i] >= nNodes) {
      atomicAdd(err, 1);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pos = atomicAdd(nodeCur + n[i], 1);
    if (FILL) nodeAdj[pos] = (unsigned)(el * 8 + i);
  }
#pragma unroll
  for (int ne = 0; ne < 12; ++ne) {
    const unsigned t = find_edge(A, n[c_lrscv[2 * ne]], n[c_lrscv[2 * ne + 1]]);
    if (t == TAG_PAD) continue;  // not a local edge (e.g. inside another rank's plane)
    const int pos = atomicAdd(edgeCur + (t & 0x7FFFFFFFu), 1);
    if (FILL) edgeAdj[pos] = (unsigned)(el * 12 + ne) | (t & 0x80000000u);
  }
}

// next list entry in ascending order of the low 31 bits, strictly after `last` (-1 first)
__device__ __forceinline__ bool next_sorted(const unsigned *__restrict__ list, int n, long long last,
                                            unsigned &out)
{
  long long best = -1;
  for (int i = 0; i < n; ++i) {
    const unsigned v = list[i];
    const long long key = v & 0x7FFFFFFFu;
    if (key > last && (best < 0 || key < best)) {
      best = key;
      out = v;
    }
  }
  return best >= 0;
}

// dual_nodal_volume[node] = sum over its elements (ascending) of the node's SCV volume
// (src/ComputeGeometryInteriorAlgorithm.C:78-129; ipNodeMap is the identity for Hex8 SCVs)
__global__ __launch_bounds__(256) void k_geom_nodes(int nNodes, const int *__restrict__ off,
                                                     const unsigned *__restrict__ adj,
                                                     const int *__restrict__ hex, nalu_gpu_vec3 x,
                                                     double *__restrict__ dualVolume)
{
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= nNodes) return;
  const unsigned *list = adj + off[node];
  const int n = off[node + 1] - off[node];
  double sum = 0.0;
  long long last = -1;
  unsigned v = 0;
  while (next_sorted(list, n, last, v)) {
    last = v & 0x7FFFFFFFu;
    double p[27][3];
    load_hex(hex, last / 8, x, p);
    double vol = 0.0;
    switch (last % 8) {
      case 0: vol = scv_volume<0>(p); break;
      case 1: vol = scv_volume<1>(p); break;
      case 2: vol = scv_volume<2>(p); break;
      case 3: vol = scv_volume<3>(p); break;
      case 4: vol = scv_volume<4>(p); break;
      case 5: vol = scv_volume<5>(p); break;
      case 6: vol = scv_volume<6>(p); break;
      default: vol = scv_volume<7>(p); break;
    }
    sum += vol;
  }
  dualVolume[node] = sum;
}

// edge_area_vector[edge] = sum over its elements (ascending) of sign * SCS area, sign = +1
// when the SCS's left node is the edge's node 0 (:134-216, sign convention :196-212)
__global__ __launch_bounds__(256) void k_geom_edges(long long nEdges, const int *__restrict__ off,
                                                     const unsigned *__restrict__ adj,
                                                     const int *__restrict__ hex, nalu_gpu_vec3 x,
                                                     double *__restrict__ ax, double *__restrict__ ay,
                                                     double *__restrict__ az)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nEdges) return;
  const unsigned *list = adj + off[e];
  const int n = off[e + 1] - off[e];
  double s[3] = {0.0, 0.0, 0.0};
  long long last = -1;
  unsigned v = 0;
  while (next_sorted(list, n, last, v)) {
    last = v & 0x7FFFFFFFu;
    const double sign = (v >> 31) ? -1.0 : 1.0;
    double p[27][3], a[3];
    load_hex(hex, last / 12, x, p);
    switch (last % 12) {
      case 0: scs_area<0>(p, a); break;
      case 1: scs_area<1>(p, a); break;
      case 2: scs_area<2>(p, a); break;
      case 3: scs_area<3>(p, a); break;
      case 4: scs_area<4>(p, a); break;
      case 5: scs_area<5>(p, a); break;
      case 6: scs_area<6>(p, a); break;
      case 7: scs_area<7>(p, a); break;
      case 8: scs_area<8>(p, a); break;
      case 9: scs_area<9>(p, a); break;
      case 10: scs_area<10>(p, a); break;
      default: scs_area<11>(p, a); break;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) s[j] += a[j] * sign;
  }
  ax[e] = s[0];
  ay[e] = s[1];
  az[e] = s[2];
}

// counts (device) -> offsets (device), through the host: setup-time, n ints each way
int scan_counts(nalu_gpu_linsys *h, int *dCnt, long long n, int *dOff, long long *total)
{
  std::vector<int> c((size_t)n + 1);
  NG_HIP(hipMemcpyAsync(c.data(), dCnt, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  long long run = 0;
  for (long long i = 0; i < n; ++i) {
    const int v = c[i];
    c[i] = (int)run;
    run += v;
  }
  c[n] = (int)run;
  NG_REQUIRE(run < 0x7FFFFFFFLL, NALU_GPU_EINVAL, "adjacency too large for 32-bit offsets");
  *total = run;
  NG_HIP(hipMemcpyAsync(dOff, c.data(), sizeof(int) * (size_t)(n + 1), hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipMemcpyAsync(dCnt, c.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));  // c goes out of scope
  return NALU_GPU_OK;
}

struct DevBuf {  // frees on scope exit (error paths included)
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)) == hipSuccess ? 0 : 1; }
  template <class T> T *as() { return static_cast<T *>(p); }
};

}  // namespace

extern "C" int nalu_gpu_compute_geometry_hex8(nalu_gpu_linsys *h, int64_t n_elem,
                                              const int *hex_nodes_dev,
                                              nalu_gpu_vec3 coordinates_dev,
                                              double *dual_volume_dev, double *area_x_dev,
                                              double *area_y_dev, double *area_z_dev)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state == 2, NALU_GPU_ESTATE, "linear system is not finalized");
  NG_DEVICE(h->device);
  NG_REQUIRE(n_elem >= 0 && n_elem < (1LL << 31) / 12, NALU_GPU_EINVAL,
             "element count %lld out of range", (long long)n_elem);
  NG_REQUIRE(coordinates_dev.x && coordinates_dev.y && coordinates_dev.z && dual_volume_dev,
             NALU_GPU_EINVAL, "compute_geometry: null device pointer");
  NG_REQUIRE(n_elem == 0 || hex_nodes_dev, NALU_GPU_EINVAL, "compute_geometry: null connectivity");
  NG_REQUIRE(h->nEdges == 0 || (area_x_dev && area_y_dev && area_z_dev), NALU_GPU_EINVAL,
             "compute_geometry: null edge area output");
  const int nN = h->nCols;
  const long long nE = h->nEdges;
  DevBuf nodeCur, edgeCur, nodeOff, edgeOff, nodeAdj, edgeAdj, err;
  NG_REQUIRE(!nodeCur.alloc(sizeof(int) * (size_t)nN) && !edgeCur.alloc(sizeof(int) * (size_t)nE) &&
                 !nodeOff.alloc(sizeof(int) * ((size_t)nN + 1)) &&
                 !edgeOff.alloc(sizeof(int) * ((size_t)nE + 1)) && !err.alloc(sizeof(int)),
             NALU_GPU_EHIP, "compute_geometry: out of device memory");
  NG_HIP(hipMemsetAsync(nodeCur.p, 0, sizeof(int) * (size_t)std::max(nN, 1), h->stream));
  NG_HIP(hipMemsetAsync(edgeCur.p, 0, sizeof(int) * (size_t)std::max<long long>(nE, 1), h->stream));
  NG_HIP(hipMemsetAsync(err.p, 0, sizeof(int), h->stream));
  const unsigned nbEl = (unsigned)((n_elem + 255) / 256);
  // (every stored row: in export mode also the rows of the shared-not-owned nodes, so that the edges
  // between two shared nodes -- part of this rank's elements, owned by a lower rank -- are found)
  const SellView A = sell_view_all(h);
  if (n_elem)
    k_geom_adjacency<false><<<nbEl, 256, 0, h->stream>>>(n_elem, hex_nodes_dev, A, nN, nodeCur.as<int>(),
                                                         edgeCur.as<int>(), nullptr, nullptr,
                                                         err.as<int>());
  NG_HIP(hipGetLastError());
  int nerr = 0;
  NG_HIP(hipMemcpyAsync(&nerr, err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  NG_REQUIRE(nerr == 0, NALU_GPU_EINVAL, "%d elements reference nodes outside [0, %d)", nerr, nN);
  long long nNodeAdj = 0, nEdgeAdj = 0;
  NG_TRY(scan_counts(h, nodeCur.as<int>(), nN, nodeOff.as<int>(), &nNodeAdj));
  NG_TRY(scan_counts(h, edgeCur.as<int>(), nE, edgeOff.as<int>(), &nEdgeAdj));
  NG_REQUIRE(!nodeAdj.alloc(sizeof(unsigned) * (size_t)nNodeAdj) &&
                 !edgeAdj.alloc(sizeof(unsigned) * (size_t)nEdgeAdj),
             NALU_GPU_EHIP, "compute_geometry: out of device memory");
  if (n_elem)
    k_geom_adjacency<true><<<nbEl, 256, 0, h->stream>>>(n_elem, hex_nodes_dev, A, nN, nodeCur.as<int>(),
                                                        edgeCur.as<int>(), nodeAdj.as<unsigned>(),
                                                        edgeAdj.as<unsigned>(), err.as<int>());
  if (nN)
    k_geom_nodes<<<(unsigned)((nN + 255) / 256), 256, 0, h->stream>>>(
        nN, nodeOff.as<int>(), nodeAdj.as<unsigned>(), hex_nodes_dev, coordinates_dev, dual_volume_dev);
  if (nE)
    k_geom_edges<<<(unsigned)((nE + 255) / 256), 256, 0, h->stream>>>(
        nE, edgeOff.as<int>(), edgeAdj.as<unsigned>(), hex_nodes_dev, coordinates_dev, area_x_dev,
        area_y_dev, area_z_dev);
  NG_HIP(hipGetLastError());
  NG_HIP(hipStreamSynchronize(h->stream));  // the adjacency buffers are freed on return
  return NALU_GPU_OK;
}
