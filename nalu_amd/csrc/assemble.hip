// assemble.hip -- bulk edge assembly kernels (continuity, scalar transport), the adjacent
// edge algorithms (mdot, nodal gradient) and the nodal mass term, for gfx950.
//
// Every kernel here is HBM-bound gather/scatter work in fp64: no MFMA.  The per-edge
// arithmetic is written operation by operation like the reference loops so that, with FP
// contraction off, results are bit-identical to the CPU oracle on identically ordered
// meshes.
#include <cstring>

#include "internal.hpp"

using namespace nalu_gpu;

namespace {

// ------------------------------------------------------------------ continuity
struct ContParams {
  nalu_gpu_continuity_fields f;
  double tau, noc, interp;
};

struct ContOp {
  using Params = ContParams;
  static constexpr int kOrderedWaves = 5;  // waves per SIMD asked of the ordered row kernel
  struct Node {
    double x[3], g[3], u[3], p, rho;
  };
  __device__ static __forceinline__ Node load(const Params &P, int n)
  {
    Node a;
    a.x[0] = P.f.coordinates.x[n]; a.x[1] = P.f.coordinates.y[n]; a.x[2] = P.f.coordinates.z[n];
    a.g[0] = P.f.dpdx.x[n]; a.g[1] = P.f.dpdx.y[n]; a.g[2] = P.f.dpdx.z[n];
    a.u[0] = P.f.velocity.x[n]; a.u[1] = P.f.velocity.y[n]; a.u[2] = P.f.velocity.z[n];
    a.p = P.f.pressure[n];
    a.rho = P.f.density[n];
    return a;
  }
  // (a node assembled from its raw fields -- the prefetching kernel's LDS copies -- is finished like a loaded one)
  __device__ static __forceinline__ void finish(Node &) {}
  // the f-th nodal / c-th edge field in the order of Node / Edge (the prefetching kernel addresses them by index)
  __device__ static __forceinline__ const double *node_field(const Params &P, int f)
  {
    const double *const F[11] = {P.f.coordinates.x, P.f.coordinates.y, P.f.coordinates.z, P.f.dpdx.x, P.f.dpdx.y, P.f.dpdx.z,
                                 P.f.velocity.x, P.f.velocity.y, P.f.velocity.z, P.f.pressure, P.f.density};
    const double *r = F[0];
#pragma unroll
    for (int i = 1; i < 11; ++i) r = f == i ? F[i] : r;
    return r;
  }
  __device__ static __forceinline__ const double *edge_field(const Params &P, int c)
  {
    return c == 0 ? P.f.edge_area.x : c == 1 ? P.f.edge_area.y : P.f.edge_area.z;
  }
  struct Edge {
    double av[3];
  };
  __device__ static __forceinline__ Edge load_edge(const Params &P, long long e)
  {
    Edge E;
    E.av[0] = P.f.edge_area.x[e]; E.av[1] = P.f.edge_area.y[e]; E.av[2] = P.f.edge_area.z[e];
    return E;
  }
  // the edge seen from its R node: area vector reversed
  __device__ static __forceinline__ Edge mirrored(const Edge &E, bool flip)
  {
    Edge M;
    M.av[0] = flip ? -E.av[0] : E.av[0]; M.av[1] = flip ? -E.av[1] : E.av[1]; M.av[2] = flip ? -E.av[2] : E.av[2];
    return M;
  }
  // src/AssembleContinuityEdgeSolverAlgorithm.C:163-203; returns tmdot
  __device__ static __forceinline__ double flux(const Params &P, const Edge &E, const Node &L,
                                                const Node &R, double &lhsfac)
  {
    const double *av = E.av;
    const double om_interp = 1.0 - P.interp;
    double axdx = 0.0, asq = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double axj = av[j];
      const double dxj = R.x[j] - L.x[j];
      asq += axj * axj;
      axdx += axj * dxj;
    }
    const double inv_axdx = 1.0 / axdx;
    const double rhoIp = 0.5 * (R.rho + L.rho);
    double tmdot = -P.tau * (R.p - L.p) * asq * inv_axdx;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double axj = av[j];
      const double dxj = R.x[j] - L.x[j];
      const double kxj = axj - asq * inv_axdx * dxj;
      const double rhoUjIp = 0.5 * (R.rho * R.u[j] + L.rho * L.u[j]);
      const double ujIp = 0.5 * (R.u[j] + L.u[j]);
      const double GjIp = 0.5 * (R.g[j] + L.g[j]);
      tmdot += (P.interp * rhoUjIp + om_interp * rhoIp * ujIp + P.tau * GjIp) * axj -
               P.tau * kxj * GjIp * P.noc;
    }
    lhsfac = -asq * inv_axdx;
    return tmdot;
  }
  __device__ static __forceinline__ void edge(const Params &P, const Edge &E, const Node &L,
                                              const Node &R, double lhs[4], double rhs[2])
  {
    double lhsfac;
    const double tmdot = flux(P, E, L, R, lhsfac);
    lhs[0] = -lhsfac;
    lhs[1] = +lhsfac;
    rhs[0] = -tmdot / P.tau;
    lhs[2] = +lhsfac;
    lhs[3] = -lhsfac;
    rhs[1] = tmdot / P.tau;
  }
};

// ---------------------------------------------------------------------- scalar
struct ScalParams {
  nalu_gpu_scalar_fields f;
  nalu_gpu_scalar_options o;
};

struct ScalOp {
  using Params = ScalParams;
  static constexpr int kOrderedWaves = 1;  // (no constraint: 12 nodal fields do not fit 96 VGPRs)
  // gor: the density as loaded, then (finish) Gamma / rho -- the only thing the edge formula wants of the density,
  // a node's own quantity: divided once per node instead of twice per edge evaluation (same operation, same bits)
  struct Node {
    double x[3], g[3], u[3], q, gor, gam;
  };
  __device__ static __forceinline__ void finish(Node &a) { a.gor = a.gam / a.gor; }
  __device__ static __forceinline__ Node load(const Params &P, int n)
  {
    Node a;
    a.x[0] = P.f.coordinates.x[n]; a.x[1] = P.f.coordinates.y[n]; a.x[2] = P.f.coordinates.z[n];
    a.g[0] = P.f.dqdx.x[n]; a.g[1] = P.f.dqdx.y[n]; a.g[2] = P.f.dqdx.z[n];
    a.u[0] = P.f.velocity.x[n]; a.u[1] = P.f.velocity.y[n]; a.u[2] = P.f.velocity.z[n];
    a.q = P.f.scalar_q[n];
    a.gor = P.f.density[n];
    a.gam = P.f.diff_flux_coeff[n];
    finish(a);
    return a;
  }
  // src/AssembleScalarEdgeSolverAlgorithm.C:320-329
  __device__ static __forceinline__ double van_leer(double dqm, double dqp, double small)
  {
    return (2.0 * (dqm * dqp + fabs(dqm * dqp))) / ((dqm + dqp) * (dqm + dqp) + small);
  }
  __device__ static __forceinline__ const double *node_field(const Params &P, int f)
  {
    const double *const F[12] = {P.f.coordinates.x, P.f.coordinates.y, P.f.coordinates.z, P.f.dqdx.x, P.f.dqdx.y, P.f.dqdx.z,
                                 P.f.velocity.x, P.f.velocity.y, P.f.velocity.z, P.f.scalar_q, P.f.density,
                                 P.f.diff_flux_coeff};
    const double *r = F[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) r = f == i ? F[i] : r;
    return r;
  }
  __device__ static __forceinline__ const double *edge_field(const Params &P, int c)
  {
    return c == 0 ? P.f.edge_area.x : c == 1 ? P.f.edge_area.y : c == 2 ? P.f.edge_area.z : P.f.mass_flow_rate;
  }
  struct Edge {
    double av[3], mdot;
  };
  __device__ static __forceinline__ Edge load_edge(const Params &P, long long e)
  {
    Edge E;
    E.av[0] = P.f.edge_area.x[e]; E.av[1] = P.f.edge_area.y[e]; E.av[2] = P.f.edge_area.z[e];
    E.mdot = P.f.mass_flow_rate[e];
    return E;
  }
  // the edge seen from its R node: area vector and mass flow reversed
  __device__ static __forceinline__ Edge mirrored(const Edge &E, bool flip)
  {
    Edge M;
    M.av[0] = flip ? -E.av[0] : E.av[0]; M.av[1] = flip ? -E.av[1] : E.av[1]; M.av[2] = flip ? -E.av[2] : E.av[2];
    M.mdot = flip ? -E.mdot : E.mdot;
    return M;
  }
  // src/AssembleScalarEdgeSolverAlgorithm.C:199-309
  __device__ static __forceinline__ void edge(const Params &P, const Edge &E, const Node &L,
                                              const Node &R, double lhs[4], double rhs[2])
  {
    const double small = 1.0e-16;
    const double *av = E.av;
    const double tmdot = E.mdot;
    const double alpha = P.o.alpha, alphaUpw = P.o.alpha_upw, hoUpwind = P.o.ho_upwind;
    const double om_alpha = 1.0 - alpha, om_alphaUpw = 1.0 - alphaUpw;
    double axdx = 0.0, asq = 0.0, udotx = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double axj = av[j];
      const double dxj = R.x[j] - L.x[j];
      asq += axj * axj;
      axdx += axj * dxj;
      udotx += 0.5 * dxj * (L.u[j] + R.u[j]);
    }
    const double inv_axdx = 1.0 / axdx;
    const double viscIp = 0.5 * (L.gam + R.gam);
    const double diffIp = 0.5 * (L.gor + R.gor);  // (gor = Gamma / rho, see Node)
    const double pe = fabs(udotx) / (diffIp + small);
    double pecfac;
    if (P.o.peclet_kind == NALU_GPU_PECLET_CLASSIC) {
      // src/PecletFunction.C:70-75 (A is hard-wired to 5.0 there)
      const double modPeclet = P.o.hybrid_factor * pe;
      pecfac = modPeclet * modPeclet / (5.0 + modPeclet * modPeclet);
    } else {
      // src/PecletFunction.C:105-109
      pecfac = 0.50 * (1.0 + tanh((pe - P.o.tanh_c1) / P.o.tanh_c2));
    }
    const double om_pecfac = 1.0 - pecfac;
    double dqL = 0.0, dqR = 0.0, nonOrth = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double dxj = R.x[j] - L.x[j];
      dqL += 0.5 * dxj * L.g[j];
      dqR += 0.5 * dxj * R.g[j];
      const double axj = av[j];
      const double kxj = axj - asq * inv_axdx * dxj;
      const double GjIp = 0.5 * (L.g[j] + R.g[j]);
      nonOrth += -viscIp * kxj * GjIp;
    }
    double limitL = 1.0, limitR = 1.0;
    const double dq = R.q - L.q;
    if (P.o.use_limiter) {
      const double dqMl = 2.0 * 2.0 * dqL - dq;
      const double dqMr = 2.0 * 2.0 * dqR - dq;
      limitL = van_leer(dqMl, dq, small);
      limitR = van_leer(dqMr, dq, small);
    }
    const double qIpL = L.q + dqL * hoUpwind * limitL;
    const double qIpR = R.q - dqR * hoUpwind * limitR;

    const double lhsfac = -viscIp * asq * inv_axdx;
    const double diffFlux = lhsfac * (R.q - L.q) + nonOrth;
    lhs[0] = -lhsfac;
    lhs[1] = +lhsfac;
    rhs[0] = -diffFlux;
    lhs[2] = +lhsfac;
    lhs[3] = -lhsfac;
    rhs[1] = diffFlux;

    const double qIp = 0.5 * (L.q + R.q);
    const double qUpwind = (tmdot > 0) ? alphaUpw * qIpL + om_alphaUpw * qIp
                                       : alphaUpw * qIpR + om_alphaUpw * qIp;
    const double qHatL = alpha * qIpL + om_alpha * qIp;
    const double qHatR = alpha * qIpR + om_alpha * qIp;
    const double qCds = 0.5 * (qHatL + qHatR);
    const double aflux = tmdot * (pecfac * qUpwind + om_pecfac * qCds);

    double alhsfac = 0.5 * (tmdot + fabs(tmdot)) * pecfac * alphaUpw + 0.5 * alpha * om_pecfac * tmdot;
    lhs[0] += alhsfac;
    lhs[2] -= alhsfac;
    alhsfac = 0.5 * (tmdot - fabs(tmdot)) * pecfac * alphaUpw + 0.5 * alpha * om_pecfac * tmdot;
    lhs[3] -= alhsfac;
    lhs[1] += alhsfac;
    alhsfac = 0.5 * tmdot * (pecfac * om_alphaUpw + om_pecfac * om_alpha);
    lhs[0] += alhsfac;
    lhs[1] += alhsfac;
    lhs[2] -= alhsfac;
    lhs[3] -= alhsfac;
    rhs[0] -= aflux;
    rhs[1] += aflux;
  }
};

// --------------------------------------------------------------------------------------
// Row-gather ("owner computes") assembly: one thread per matrix row walks the row's
// entries, recomputes the flux of each incident edge and keeps only its own row's share.
// Every matrix value and rhs entry has exactly one writer: deterministic, no atomics, no
// colouring, and with SET (matrix logically zero) no read of the old values at all.
// All accesses are lane-contiguous in the SELL layout; neighbour gathers are contiguous
// too on lattice-ordered meshes.
// Row-gather with the slice's own nodal data shared through LDS: neighbours inside the
// slice (on lattice-ordered meshes the two x neighbours) are read from LDS instead of being
// gathered again from memory.  (With ~96 workgroups per XCD streaming ~70 KB each through a
// 4 MiB L2, a line loaded at the start of a workgroup is often gone when a later entry
// touches it again: re-gathers are HBM traffic, measured with FETCH_SIZE.)
template <class Op, bool SET, bool ORDERED>
__global__ __launch_bounds__(SELL_C, ORDERED ? Op::kOrderedWaves : 1) void k_assemble_rows(SellView A, typename Op::Params P,
                                                           double *__restrict__ rhs)
{
  using Node = typename Op::Node;
  constexpr int NF = sizeof(Node) / sizeof(double);
  __shared__ double lds[NF * SELL_C];
  __shared__ double ldsStash[ORDERED ? 4 * SELL_C : 1];
  const int slice = sched_slice(A, blockIdx.x);
  const int r = threadIdx.x;
  const int sBase = slice * SELL_C;
  const int row = sBase + r;
  const bool live = row < A.nRows;
  Node own;
  if (live) {
    own = Op::load(P, row);
    const double *po = reinterpret_cast<const double *>(&own);
#pragma unroll
    for (int f = 0; f < NF; ++f) lds[f * SELL_C + r] = po[f];
  }
  __syncthreads();
  if (!live) return;
  const int nLive = min(SELL_C, A.nRows - sBase);
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  // computed columns for the rows whose k-th column is row + kOff[k] (see finalize)
  const int *kOff = A.kOff + off / SELL_C;
  const bool ex = (A.exMask[(size_t)slice * 4 + (r >> 6)] >> (r & 63)) & 1ULL;
  const unsigned *tBase = A.tBase + off / SELL_C, *tMul = A.tMul + off / SELL_C;
  const bool exT = (A.exMaskT[(size_t)slice * 4 + (r >> 6)] >> (r & 63)) & 1ULL;
  // A wave whose only irregular row is described by the slice descriptor (BurstSlice::irr: on a
  // lattice the first / last node of an x-line) takes that row's tags and columns from the
  // descriptor (scalar registers) instead of loading them entry by entry: two dependent round trips
  // less per entry for the whole wave.
  const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
  const BurstRow *X = nullptr;
  if (!ORDERED && A.burst) {
    const BurstRow *cand = &A.burst[slice].irr[wv];
    if (cand->lane < 64u && (cand->info >> 16) == 0u) X = cand;  // (one such row, no edges or entries off the list)
  }
  const bool useX = X != nullptr;
  const bool isXLane = useX && (unsigned)(r & 63) == X->lane;
  double diag = 0.0, b = 0.0;
  long long diagPos = -1;
  if (ORDERED) {
    // entries are visited in the slice's order (k_slice_order: out-of-slice neighbours first); an
    // "early" entry parks its shares in a stash slot until every smaller entry has been added, so
    // the sums see the additions in CSR order whatever the visiting order.  All of this control
    // flow is wave-uniform.
    const unsigned ow = A.ordWord[slice];
    int nSt = 0, nFl = 0, stK0 = 0, stK1 = 0;
    bool sV0 = false, sV1 = false;
    double *const st = ldsStash + r;  // [slot][diag, rhs][SELL_C]: registers are what is scarce
    for (int i = 0; i < w; ++i) {
      const int k = i < 8 ? (int)((ow >> (3 * i)) & 7u) : i;
      const bool early = i < 8 && ((ow >> (24 + i)) & 1u);
      const long long pos = off + (long long)k * SELL_C + r;
      const unsigned tag = exT ? A.tags[pos] : tBase[k] + tMul[k] * (unsigned)row;
      const bool isEdge = tag < TAG_OTHER && (!A.edgeOwned || A.edgeOwned[tag & 0x7FFFFFFFu]);
      typename Op::Edge E;
      int c = row;
      bool inS = false;
      Node nb;
      if (isEdge) {
        c = ex ? A.cols[pos] : row + kOff[k];
        inS = (unsigned)(c - sBase) < (unsigned)nLive;
        E = Op::load_edge(P, (long long)(tag & 0x7FFFFFFFu));
        if (!inS) nb = Op::load(P, c);
      }
      if (!early) {  // every stashed entry smaller than k goes in first
        if (nFl == 0 && nSt > 0 && stK0 < k) {
          if (sV0) { diag += st[0]; b += st[SELL_C]; }
          nFl = 1;
        }
        if (nFl == 1 && nSt > 1 && stK1 < k) {
          if (sV1) { diag += st[2 * SELL_C]; b += st[3 * SELL_C]; }
          nFl = 2;
        }
      }
      if (isEdge) {
        if (inS) {
          double *pn = reinterpret_cast<double *>(&nb);
#pragma unroll
          for (int f = 0; f < NF; ++f) pn[f] = lds[f * SELL_C + (c - sBase)];
        }
        const bool isR = (tag >> 31) != 0;
        double lhs[4], rh[2];
        Op::edge(P, Op::mirrored(E, isR), own, nb, lhs, rh);  // (the row's node plays L, see below)
        const double offd = lhs[1];
        if (SET) A.vals[pos] = offd;
        else A.vals[pos] += offd;
        if (early) {
          st[2 * nSt * SELL_C] = lhs[0];
          st[(2 * nSt + 1) * SELL_C] = rh[0];
        } else {
          diag += lhs[0];
          b += rh[0];
        }
      } else if (tag == TAG_DIAG) {
        diagPos = pos;
      } else if (SET) {
        A.vals[pos] = 0.0;  // (padding, a non-edge graph entry, or an edge another rank assembles)
      }
      if (early) {
        if (nSt == 0) { stK0 = k; sV0 = isEdge; }
        else { stK1 = k; sV1 = isEdge; }
        ++nSt;
      }
    }
    if (nFl == 0 && nSt > 0) {
      if (sV0) { diag += st[0]; b += st[SELL_C]; }
      nFl = 1;
    }
    if (nFl == 1 && nSt > 1 && sV1) { diag += st[2 * SELL_C]; b += st[3 * SELL_C]; }
  } else {
    for (int k = 0; k < w; ++k) {
      const long long pos = off + (long long)k * SELL_C + r;
      unsigned tag;
      int c;
      if (useX) {
        // (wave-uniform) entry k of the irregular row: an edge of its list, the diagonal, or padding
        unsigned tagX = ((int)((X->info >> 4) & 15u) - 1 == k) ? TAG_DIAG : TAG_PAD;
        int colX = row;
#pragma unroll
        for (int e = 0; e < BURST_NE; ++e)
          if ((unsigned)e < (X->info & 15u) && ((X->kl >> (4 * e)) & 15u) == (unsigned)k) {
            tagX = X->tb[e];
            colX = X->ko[e];
          }
        tag = isXLane ? tagX : tBase[k] + tMul[k] * (unsigned)row;
        c = isXLane ? colX : row + kOff[k];
      } else {
        tag = exT ? A.tags[pos] : tBase[k] + tMul[k] * (unsigned)row;
        c = ex ? A.cols[pos] : row + kOff[k];
      }
      if (tag >= TAG_OTHER || (A.edgeOwned && !A.edgeOwned[tag & 0x7FFFFFFFu])) {
        if (tag == TAG_DIAG) diagPos = pos;
        else if (SET) A.vals[pos] = 0.0;  // (an edge another rank assembles counts as no edge)
        continue;
      }
      const long long e = tag & 0x7FFFFFFFu;
      const bool isR = (tag >> 31) != 0;
      Node nb;
      if ((unsigned)(c - sBase) < (unsigned)nLive) {
        double *pn = reinterpret_cast<double *>(&nb);
  #pragma unroll
        for (int f = 0; f < NF; ++f) pn[f] = lds[f * SELL_C + (c - sBase)];
      } else {
        nb = Op::load(P, c);
      }
      double lhs[4], rh[2];
      // Both edge kernels are exactly invariant under swapping L and R together with the sign of
      // the area vector (and of the mass flow): every term changes sign or stays, bit for bit
      // (the oracle's mirror test).  So the row's node always plays L, and
      // its shares are lhs[0], lhs[1], rhs[0] -- no per-field selects.
      Op::edge(P, Op::mirrored(Op::load_edge(P, e), isR), own, nb, lhs, rh);
      const double offd = lhs[1];
      diag += lhs[0];
      b += rh[0];
      if (SET) A.vals[pos] = offd;
      else A.vals[pos] += offd;
    }
  }
  if (diagPos >= 0) {
    if (SET) A.vals[diagPos] = diag;
    else A.vals[diagPos] += diag;
  }
  if (SET) rhs[row] = b;
  else rhs[row] += b;
}

// Burst row gather (slices of at most BURST_W entries, default when it applies): the same rows,
// fluxes and summation order as k_assemble_rows<..., false>, bit for bit, but a row's loads are
// issued together instead of one dependent round trip per entry.  Each lane first lists its edge
// entries in CSR order (computed from the slice's tag / column offsets; the irregular rows load
// their tags and columns); the lists are the order of evaluation, so the diagonal / rhs sums see
// the same additions in the same order.  Then NB edges at a time: edge data and neighbour nodes
// of all NB in flight at once, NB fluxes, NB stores.  Neighbours inside the slice come from LDS
// (the slice's own data, staged behind the first burst's loads), the others from memory.  The row
// kernel is bound by its ~8 dependent round trips per row with 20 waves per CU (DESIGN.md section
// 3); this one needs 1-2 with 8 waves.
// edges per burst / waves per SIMD asked of the compiler (measured: 6 / 2 beats 3 / 3, 2 / 4 and 1 / 4 for the
// continuity operator; the scalar operator's 12 + 4 doubles per node / edge allow 3 per burst)
template <class Op> struct BurstCfg { static constexpr int NB = 6, WAVES = 2; };
template <> struct BurstCfg<ScalOp> { static constexpr int NB = 3, WAVES = 2; };

// (the body: one 256-thread workgroup assembles slice `slice`; lds: NF * SELL_C doubles.  Also called by the tail
// workgroups of k_assemble_rows_lagpf, which take the slices outside the runs in the same launch)
template <class Op, bool SET>
__device__ __forceinline__ void burst_slice(const SellView &A, const typename Op::Params &P, double *__restrict__ rhs,
                                            int slice, double *lds)
{
  using Node = typename Op::Node;
  using Edge = typename Op::Edge;
  constexpr int NF = sizeof(Node) / sizeof(double);
  constexpr int NB = BurstCfg<Op>::NB;
  constexpr int NE = BURST_NE;  // edges per row taken by the bursts; further ones go through the tail loop
  const int r = threadIdx.x;
  const int sBase = slice * SELL_C;
  const int row = sBase + r;
  const bool live = row < A.nRows;
  // (wave-uniform addresses: three 64-byte scalar loads, issued together with the row's own data)
  const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
  const BurstSlice *const S = A.burst + slice;
  const BurstRow D = S->reg, X = S->irr[wv];
  const long long off = S->off;
  const int w = S->w;
  const unsigned long long mI = S->irrMask[wv];
  const long long pos0 = off + r;
  Node own;
  if (live) own = Op::load(P, row);
  const bool irregular = (mI >> (r & 63)) & 1ULL;  // (only rows of the matrix are flagged)
  // ---- the lane's entries: edges in CSR order (id + R bit, column, entry index), the diagonal,
  // and the entries a SET pass has to zero.  Regular rows: the slice's list; a wave's one
  // irregular row: its list comes with the descriptor; several: they read their tags and columns.
  const bool isX = irregular && X.lane < 64u;
  unsigned eid[NE], kk[NE];
  int cl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    kk[e] = ((isX ? X.kl : D.kl) >> (4 * e)) & 15u;
    eid[e] = isX ? X.tb[e] : D.tb[e] + (unsigned)row;
    cl[e] = isX ? X.ko[e] : row + D.ko[e];
  }
  const unsigned info = isX ? X.info : D.info;
  int ne = live ? (int)(info & 15u) : 0;
  int kDiag = (int)((info >> 4) & 15u) - 1;
  unsigned zeroMask = (info >> 8) & 255u, lateMask = (info >> 16) & 255u;
  if (X.lane == BURST_IRR_MANY) {
    // irregular rows read their tags and columns and compact them (all loads first)
    unsigned tg[BURST_W];
    int cc[BURST_W];
#pragma unroll
    for (int k = 0; k < BURST_W; ++k) {
      tg[k] = TAG_PAD;
      cc[k] = row;
      if (k < w && irregular) {
        const long long pos = pos0 + (long long)k * SELL_C;
        tg[k] = A.tags[pos];
        cc[k] = A.cols[pos];
      }
    }
    if (irregular) {
      ne = 0;
      kDiag = -1;
      zeroMask = lateMask = 0;
#pragma unroll
      for (int k = 0; k < BURST_W; ++k) {
        const bool isE = tg[k] < TAG_OTHER;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const bool take = isE && ne == e;
          eid[e] = take ? tg[k] : eid[e];
          cl[e] = take ? cc[k] : cl[e];
          kk[e] = take ? (unsigned)k : kk[e];
        }
        if (isE && ne >= NE) lateMask |= 1u << k;
        ne += isE ? 1 : 0;
        if (tg[k] == TAG_DIAG) kDiag = k;
        else if (!isE && k < w) zeroMask |= 1u << k;
      }
    }
  }
  double diag = 0.0, b = 0.0;
  // ---- bursts of NB edges
  const int nLive = min(SELL_C, A.nRows - sBase);
#pragma unroll
  for (int e0 = 0; e0 < NE; e0 += NB) {
    const bool any = __ballot(ne > e0) != 0ULL;  // (wave-uniform)
    Edge E[NB];
    Node nb[NB];
    bool inS[NB];
    if (any) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = e0 + i;
        const bool act = e < ne;
        inS[i] = act && (unsigned)(cl[e] - sBase) < (unsigned)nLive;  // a row of this slice: from LDS
        if (act) {
          E[i] = Op::load_edge(P, (long long)(eid[e] & 0x7FFFFFFFu));
          if (!inS[i]) nb[i] = Op::load(P, cl[e]);
        }
      }
    }
    if (e0 == 0) {
      // the slice's own nodal data through LDS, behind the loads of the first burst (the barrier
      // waits for the own data only: the burst stays in flight)
      if (live) {
        const double *po = reinterpret_cast<const double *>(&own);
#pragma unroll
        for (int f = 0; f < NF; ++f) lds[f * SELL_C + r] = po[f];
      }
      __syncthreads();
    }
    if (any) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int e = e0 + i;
        if (e < ne) {
          if (inS[i]) {
            double *pn = reinterpret_cast<double *>(&nb[i]);
#pragma unroll
            for (int f = 0; f < NF; ++f) pn[f] = lds[f * SELL_C + (cl[e] - sBase)];
          }
          double lhs[4], rh[2];
          Op::edge(P, Op::mirrored(E[i], (eid[e] >> 31) != 0), own, nb[i], lhs, rh);
          diag += lhs[0];
          b += rh[0];
          const long long pos = pos0 + (long long)kk[e] * SELL_C;
          if (SET) A.vals[pos] = lhs[1];
          else A.vals[pos] += lhs[1];
        }
      }
    }
  }
  // ---- rows with more than NE edges: the remaining ones one at a time
  if (__ballot(lateMask != 0u) != 0ULL) {
    for (int k = 0; k < w; ++k) {
      if (!((lateMask >> k) & 1u)) continue;
      const long long pos = pos0 + (long long)k * SELL_C;
      // (the tag / column arrays hold every entry, also those of the regular rows)
      const unsigned tag = A.tags[pos];
      const int c = A.cols[pos];
      double lhs[4], rh[2];
      Op::edge(P, Op::mirrored(Op::load_edge(P, (long long)(tag & 0x7FFFFFFFu)), (tag >> 31) != 0), own,
               Op::load(P, c), lhs, rh);
      diag += lhs[0];
      b += rh[0];
      if (SET) A.vals[pos] = lhs[1];
      else A.vals[pos] += lhs[1];
    }
  }
  if (!live) return;
  if (SET) {
    for (int k = 0; k < w; ++k)
      if ((zeroMask >> k) & 1u) A.vals[pos0 + (long long)k * SELL_C] = 0.0;
  }
  if (kDiag >= 0) {
    const long long pos = pos0 + (long long)kDiag * SELL_C;
    if (SET) A.vals[pos] = diag;
    else A.vals[pos] += diag;
  }
  if (SET) rhs[row] = b;
  else rhs[row] += b;
}

template <class Op, bool SET>
__global__ __launch_bounds__(SELL_C, BurstCfg<Op>::WAVES) void k_assemble_rows_burst(SellView A, typename Op::Params P,
                                                                                 double *__restrict__ rhs, int listBase)
{
  __shared__ double lds[sizeof(typename Op::Node) / sizeof(double) * SELL_C];
  // (listBase >= 0: the single-slice runs A.march[listBase ...] that the marching kernel leaves)
  const int slice = listBase < 0 ? sched_slice(A, blockIdx.x) : A.march[listBase + blockIdx.x].start;
  burst_slice<Op, SET>(A, P, rhs, slice, lds);
}

// Marching burst row gather (NG_ASM_MARCH=1): a workgroup walks a run of slices whose rows continue
// each other (MarchSeg: row + rowStep per step; on a lattice the lines above each other) and keeps
// in registers what two consecutive rows share -- the upper row's node (the lower row's last
// neighbour becomes the next own row) and the flux of the edge between them, which is evaluated
// once: its shares for the upper row (lhs[3], lhs[2], rhs[1]) are exactly what that row's own
// evaluation gives (the mirror invariance of both edge kernels), so the
// result stays bit-identical to k_assemble_rows.  Per row: 48 instead of 73 eight-byte loads
// through the L2 -> L1 fabric.  Only slices with the MARCH_NEAR slot pattern (k_march_links): slot 0
// = the row below (carried), slot 5 = the row above, slots 2, 3 = rows of the slice (LDS), slots
// 1, 4 = other slices; a wave's irregular row has its edges in the slots of the regular offsets
// they match.  The edge ids of the next step are read from the tag array one step ahead.  Slices
// without the pattern are single-slice runs of the burst kernel (second launch).
// (the scalar operator's 12 + 4 doubles per node / edge spill at 256 registers: 1.69 against 1.51 ms)
template <class Op> struct MarchCfg { static constexpr bool kDefault = true; };
template <> struct MarchCfg<ScalOp> { static constexpr bool kDefault = false; };

template <class Op, bool SET>
__global__ __launch_bounds__(SELL_C, 2) void k_assemble_rows_march(SellView A, typename Op::Params P,
                                                                  double *__restrict__ rhs)
{
  using Node = typename Op::Node;
  using Edge = typename Op::Edge;
  constexpr int NF = sizeof(Node) / sizeof(double);
  constexpr int NE = BURST_NE;
  constexpr int LW = SELL_C + 2;  // the slice's rows and the node just below / above them
  __shared__ double lds[2][NF * LW];
  const MarchSeg G = A.march[xcd_run(blockIdx.x, A.nMarchPat, A.marchXcdRun)];
  const int r = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
  int sBase = G.start * SELL_C;
  int row = sBase + r;  // (pattern slices are full: every row is a row of the matrix)
  const BurstSlice *const S = A.burst + G.start;
  const BurstRow D = S->reg, X = S->irr[wv];
  const int w = S->w;
  const unsigned long long mI = S->irrMask[wv];
  long long pos0 = S->off + r;
  Node own = Op::load(P, row);
  // the node just below the slice (lane 0) / just above it (last lane), when some row reads it
  // through a near slot; like the own row it is loaded one step ahead
  const bool haloLane = (r == 0 && (G.halo & 1)) || (r == SELL_C - 1 && (G.halo & 2));
  const int haloOff = r == 0 ? -1 : 1;
  Node hal;
  if (haloLane) hal = Op::load(P, row + haloOff);
  const bool irregular = (mI >> (r & 63)) & 1ULL;  // the wave's one irregular row
  // its edges, moved to the slots of the regular offsets they have (wave-uniform arithmetic)
  unsigned xe[NE], xk[NE], xact = 0;
  {
    const int rowX = sBase + wv * 64 + (int)(X.lane & 63u), nx = (int)(X.info & 15u);
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      xe[q] = 0;
      xk[q] = 0;
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e < nx && X.ko[e] - rowX == D.ko[q]) {
          xe[q] = X.tb[e];
          xk[q] = (X.kl >> (4 * e)) & 15u;
          xact |= 1u << q;
        }
    }
  }
  unsigned eid[NE], kk[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    eid[q] = irregular ? xe[q] : D.tb[q] + (unsigned)row;
    kk[q] = irregular ? xk[q] : (D.kl >> (4 * q)) & 15u;
  }
  const unsigned act = irregular ? xact : (1u << NE) - 1u;
  const unsigned info = irregular ? X.info : D.info;
  const int kDiag = (int)((info >> 4) & 15u) - 1;
  const unsigned zeroMask = (info >> 8) & 255u;
  // slot 0: its shares are carried from the row below; the first row of a run evaluates them here,
  // on its own, so that the steps below never hold slot 0's operands
  double cDiag = 0.0, cOff = 0.0, cRhs = 0.0;
  if (act & 1u) {
    double lhs[4], rh[2];
    Op::edge(P, Op::mirrored(Op::load_edge(P, (long long)(eid[0] & 0x7FFFFFFFu)), (eid[0] >> 31) != 0), own,
             Op::load(P, row + D.ko[0]), lhs, rh);
    cDiag = lhs[0]; cOff = lhs[1]; cRhs = rh[0];
  }
  for (int t = 0; t < G.len; ++t) {
    Edge E[NE];
    Node nb[NE];
#pragma unroll
    for (int e = 1; e < NE; ++e) {
      if ((act >> e) & 1u) {
        E[e] = Op::load_edge(P, (long long)(eid[e] & 0x7FFFFFFFu));
        if (!((MARCH_NEAR >> e) & 1u)) nb[e] = Op::load(P, row + D.ko[e]);
      }
    }
    // the next step's edge ids, one step ahead (the tag array holds every entry)
    unsigned tgN[NE];
    if (t + 1 < G.len) {
#pragma unroll
      for (int e = 1; e < NE; ++e)
        if ((act >> e) & 1u) tgN[e] = A.tags[pos0 + G.offStep + (long long)kk[e] * SELL_C];
    }
    double *const buf = lds[t & 1];
    {
      const double *po = reinterpret_cast<const double *>(&own);
#pragma unroll
      for (int f = 0; f < NF; ++f) buf[f * LW + r + 1] = po[f];
      if (haloLane) {
        const double *ph = reinterpret_cast<const double *>(&hal);
#pragma unroll
        for (int f = 0; f < NF; ++f) buf[f * LW + r + 1 + haloOff] = ph[f];
        if (t + 1 < G.len) hal = Op::load(P, row + G.rowStep + haloOff);  // (the next step's)
      }
    }
    __syncthreads();
    double diag = 0.0, b = 0.0;
    double nDiag = 0.0, nOff = 0.0, nRhs = 0.0;
    if (act & 1u) {
      const long long pos = pos0 + (long long)kk[0] * SELL_C;
      diag += cDiag;
      b += cRhs;
      if (SET) A.vals[pos] = cOff;
      else A.vals[pos] += cOff;
    }
#pragma unroll
    for (int e = 1; e < NE; ++e) {
      if (!((act >> e) & 1u)) continue;
      const long long pos = pos0 + (long long)kk[e] * SELL_C;
      if ((MARCH_NEAR >> e) & 1u) {
        double *pn = reinterpret_cast<double *>(&nb[e]);
#pragma unroll
        for (int f = 0; f < NF; ++f) pn[f] = buf[f * LW + r + 1 + D.ko[e]];
      }
      double lhs[4], rh[2];
      Op::edge(P, Op::mirrored(E[e], (eid[e] >> 31) != 0), own, nb[e], lhs, rh);
      diag += lhs[0];
      b += rh[0];
      if (SET) A.vals[pos] = lhs[1];
      else A.vals[pos] += lhs[1];
      if (e == NE - 1) { nDiag = lhs[3]; nOff = lhs[2]; nRhs = rh[1]; }
    }
    if (SET) {
      for (int k = 0; k < w; ++k)
        if ((zeroMask >> k) & 1u) A.vals[pos0 + (long long)k * SELL_C] = 0.0;
    }
    if (kDiag >= 0) {
      const long long pos = pos0 + (long long)kDiag * SELL_C;
      if (SET) A.vals[pos] = diag;
      else A.vals[pos] += diag;
    }
    if (SET) rhs[row] = b;
    else rhs[row] += b;
    // ---- step: the last neighbour is the next row
    if (t + 1 < G.len) {
      own = nb[NE - 1];
      cDiag = nDiag; cOff = nOff; cRhs = nRhs;
#pragma unroll
      for (int e = 1; e < NE; ++e)
        if ((act >> e) & 1u) eid[e] = tgN[e];
      sBase += G.rowStep;
      row += G.rowStep;
      pos0 += G.offStep;
    }
  }
}

// Lagged marching row gather, barrier-free (round 3; the default where every run qualifies, NG_ASM_LAG=0/1):
// the same runs of slices as k_assemble_rows_march, the same fluxes and the same summation order per row,
// bit for bit -- but arranged so that ALL users of a nodal line ask for it in the SAME step, and without
// LDS or barriers:
//  * a step loads its OWN row together with the rows' y neighbours (far slots 1, 4); the row below is
//    the previous step's own row, kept in registers.  The run next door loads this line as its far
//    neighbour in the very same step (runs of adjacent lines are dispatched together and advance in
//    step), so the 4 MiB L2 serves two of the three requests: the marching kernel asked for its next
//    own row one step AHEAD, and a step of an XCD's resident workgroups moves more bytes than its L2
//    holds (5.5 GB of traffic per launch at 256^3 against 4.2 GB compulsory).
//  * the edge between two consecutive rows is evaluated by the UPPER row (slot 0, with the lower row
//    from registers); the lower row's shares of it -- lhs[3], lhs[2], rhs[1], bit for bit what its own
//    evaluation of slot 5 gives (mirror invariance) -- complete that row one step late: its last edge
//    is also the last addition of its CSR-ordered sums, so nothing is reordered.
//  * the in-slice neighbours (near slots 2, 3: the rows next door, k_march_links checks it) come from
//    the neighbouring LANES by full-wave DPP shifts; a wave's first / last lane loads its neighbour in
//    the next wave itself (one two-lane load per field).  No LDS, no barrier: the four waves of a
//    workgroup run free.
//  * edge ids advance by a per-slot constant per step (MarchSeg::eStep for the regular rows, ::xStep for
//    every wave's irregular row; runs are cut where the constants change): no tag loads.
// Per row and step: 56-59 eight-byte load instructions (11 of them two lanes wide) against 48 + 5 tag
// loads, no LDS traffic, no barrier.  HBM traffic at 256^3: 4.87 GB per launch (1.17 x algorithmic) against
// 5.54 GB; time: continuity 1.00 -> 0.975 ms, scalar 1.40 (row kernel) -> 1.25 ms.  What bounds it now
// (profiles/r3_lag_ablation.txt): without the flux arithmetic 0.96 ms, without the y gathers (L2 hits)
// 0.875, without the stores 0.82: the kernel sits ~15 % above the mixed read / write HBM roof of its bytes.
// kDppArea: the area vector of the edge to the lane below is that lane's edge to the lane above, shifted in
// instead of loaded (continuity: 1.016 -> 0.985 ms at 256^3; the scalar operator's 4-double edges then spill:
// 1.26 -> 1.37 ms); stores are nontemporal for both (-1.5 %: the written lines are not read again)
// (kYShare: which y-share form of the prefetching kernel the operator takes -- 1: two share buffers per wave, 1.5 KiB
// of LDS more per wave; 2: one buffer and a second barrier, for the scalar operator's 80 KiB workgroups)
template <class Op> struct LagCfg { static constexpr bool kDefault = true, kDppArea = true; static constexpr int kYShare = 1; };
template <> struct LagCfg<ScalOp> { static constexpr bool kDefault = true, kDppArea = false; static constexpr int kYShare = 2; };
// (development aids: ablations of the lagged kernel, never built into the shipped library)
#ifdef NG_LAG_NOFLUX
template <class Op>
__device__ __forceinline__ void lag_fake_edge(const typename Op::Params &, const typename Op::Edge &E,
                                              const typename Op::Node &L, const typename Op::Node &R, double lhs[4],
                                              double rh[2])
{
  const double *pe = reinterpret_cast<const double *>(&E), *pl = reinterpret_cast<const double *>(&L),
               *pr = reinterpret_cast<const double *>(&R);
  double s = 0.0;
  for (int f = 0; f < (int)(sizeof(typename Op::Edge) / 8); ++f) s += pe[f];
  for (int f = 0; f < (int)(sizeof(typename Op::Node) / 8); ++f) s += pl[f] + pr[f];
  lhs[0] = s; lhs[1] = -s; lhs[2] = s + 1.0; lhs[3] = s - 1.0; rh[0] = s * 0.5; rh[1] = -s * 0.5;
}
#define LAG_EDGE(...) lag_fake_edge<Op>(__VA_ARGS__)
#else
#define LAG_EDGE(...) Op::edge(__VA_ARGS__)
#endif
#ifdef NG_LAG_NOFAR
#define LAG_FAR(r) prev
#else
#define LAG_FAR(r) Op::load(P, (r))
#endif
#if defined(NG_LAG_NOSTORE)
#define LAG_ST(p, v) do { if ((v) == 1.2345e300) *(p) = (v); } while (0)
#else
#define LAG_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif

__device__ __forceinline__ double dpp_from_lower_lane(double v, double lane0)
{  // lane i gets lane i - 1's v; lane 0 keeps lane0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138, 0xF, 0xF, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_upper_lane(double v, double lane63)
{  // lane i gets lane i + 1's v; lane 63 keeps lane63
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(lane63), lo, 0x130, 0xF, 0xF, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(lane63), hi, 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// GY = 1: a workgroup = the four 64-row waves of one run (list: the runs to take, or null = all pattern runs).
// GY = 4: a workgroup = the same 64-row quarter of FOUR runs of adjacent lines (list: 4 run indices per
// group, item = 4 * group + quarter): wave v walks line v of the group, every wave leaves its own row in LDS
// (double-buffered, one barrier per step, placed where every wave waits for its loads anyway) and takes
// the far neighbours that are lines of the group from there: 2 far gathers per 4 lines instead of 8.
template <class Op, bool SET, int GY>
__global__ __launch_bounds__(SELL_C, 2) void k_assemble_rows_lag(SellView A, typename Op::Params P,
                                                                double *__restrict__ rhs,
                                                                const int *__restrict__ list, int nItems)
{
  using Node = typename Op::Node;
  using Edge = typename Op::Edge;
  constexpr int NF = sizeof(Node) / sizeof(double);
  constexpr int NE = BURST_NE;
  __shared__ double ldsY[GY > 1 ? 2 * GY * NF * 64 : 1];
  const int r = threadIdx.x, lane = r & 63;
  const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
  const int item = xcd_run(blockIdx.x, nItems, GY > 1 ? A.marchXcdRun * GY : A.marchXcdRun);
  // xw: which 64-row quarter of its slices the wave walks
  const int xw = GY > 1 ? (item & (GY - 1)) : wv;
  const MarchSeg &G = A.march[GY > 1 ? list[(item & ~(GY - 1)) + wv] : (list ? list[item] : item)];
  const int gStart = G.start, gLen = G.len, gRowStep = G.rowStep;
  const long long gOffStep = G.offStep;
  int row = gStart * SELL_C + xw * 64 + lane;  // (pattern slices are full: every row is a row of the matrix)
  const BurstSlice *const S = A.burst + gStart;
  const BurstRow D = S->reg, X = S->irr[xw];
  const int w = S->w;
  const unsigned long long mI = S->irrMask[xw];
  long long pos0 = S->off + xw * 64 + lane;
  const bool irregular = (mI >> lane) & 1ULL;  // the wave's one irregular row
  unsigned eStep[NE];  // what the lane's edge ids advance by per step (slot 5's id is only needed after the last one)
#pragma unroll
  for (int q = 0; q < NE; ++q) eStep[q] = irregular ? G.xStep[xw][q] : G.eStep[q];
  const unsigned eLast5 = eStep[NE - 1] * (unsigned)(gLen - 1);
  // its edges, moved to the slots of the regular offsets they have (wave-uniform arithmetic)
  unsigned xe[NE], xk[NE], xact = 0;
  {
    const int rowX = gStart * SELL_C + xw * 64 + (int)(X.lane & 63u), nx = (int)(X.info & 15u);
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      xe[q] = 0;
      xk[q] = 0;
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e < nx && X.ko[e] - rowX == D.ko[q]) {
          xe[q] = X.tb[e];
          xk[q] = (X.kl >> (4 * e)) & 15u;
          xact |= 1u << q;
        }
    }
  }
  unsigned eid[NE], kkAll = 0;  // (the entry indices of the six slots, four bits each: registers are what is scarce)
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    eid[q] = irregular ? xe[q] : D.tb[q] + (unsigned)row;
    kkAll |= (irregular ? xk[q] : (D.kl >> (4 * q)) & 15u) << (4 * q);
  }
  auto kk = [&](int q) { return (long long)((kkAll >> (4 * q)) & 15u); };
  const unsigned act = irregular ? xact : (1u << NE) - 1u;
  const unsigned info = irregular ? X.info : D.info;
  const int kDiag = (int)((info >> 4) & 15u) - 1;
  const unsigned zeroMask = (info >> 8) & 255u;
  // the lanes at the ends of the wave take their in-slice neighbour in the next wave from memory
  const bool endLane = (lane == 0 && (act & 4u)) || (lane == 63 && (act & 8u));
  const int endOff = lane == 0 ? -1 : 1;
  Node prev;       // the row below: the previous step's own row (first step: loaded)
  double pDiag = 0.0, pRhs = 0.0;  // its sums, complete but for its last edge
  if (act & 1u) prev = Op::load(P, row + D.ko[0]);
  for (int t = 0; t < gLen; ++t) {
    // ---- every load of the step at once
    Node own = Op::load(P, row);
    Edge E[NE - 1];
    Node ym, yp, en;
#pragma unroll
    for (int e = 0; e < NE - 1; ++e) {
      if (LagCfg<Op>::kDppArea && e == 2) continue;  // (the edge to the lane below is that lane's edge to the lane above)
      if ((act >> e) & 1u) E[e] = Op::load_edge(P, (long long)(eid[e] & 0x7FFFFFFFu));
    }
    if (LagCfg<Op>::kDppArea) {
      // the edge to the lane below is that lane's edge to the lane above: shifted, not loaded (the wave's
      // first lane loads it)
      Edge e0;
      if (lane == 0 && (act & 4u)) e0 = Op::load_edge(P, (long long)(eid[2] & 0x7FFFFFFFu));
      const double *ps = reinterpret_cast<const double *>(&E[3]), *p0 = reinterpret_cast<const double *>(&e0);
      double *pd = reinterpret_cast<double *>(&E[2]);
#pragma unroll
      for (int f = 0; f < (int)(sizeof(Edge) / sizeof(double)); ++f) pd[f] = dpp_from_lower_lane(ps[f], p0[f]);
    }
    // (GY > 1: only the group's first / last line has a far neighbour outside the group)
    const bool farM = GY == 1 || wv == 0, farP = GY == 1 || wv == GY - 1;
#ifdef NG_LAG_NOY
    ym = own;
    yp = own;
#else
    if (farM && (act & 2u)) ym = Op::load(P, row + D.ko[1]);
    if (farP && (act & 16u)) yp = Op::load(P, row + D.ko[4]);
#endif
    double *const ldsT = ldsY + (GY > 1 ? (t & 1) * GY * NF * 64 : 0);
    if (GY > 1) {
      const double *po = reinterpret_cast<const double *>(&own);
#pragma unroll
      for (int f = 0; f < NF; ++f) ldsT[(wv * NF + f) * 64 + lane] = po[f];
      __syncthreads();
    }
    if (endLane) en = Op::load(P, row + endOff);
    double diag = 0.0, b = 0.0;
    double lhs[4], rh[2];
    // slot 0: the edge to the row below; its other half completes that row
    if (act & 1u) {
      LAG_EDGE(P, Op::mirrored(E[0], (eid[0] >> 31) != 0), own, prev, lhs, rh);
      diag += lhs[0];
      b += rh[0];
      const long long pos = pos0 + kk(0) * SELL_C;
      if (SET) LAG_ST(A.vals + pos, lhs[1]);
      else A.vals[pos] += lhs[1];
    }
    if (t > 0) {
      const long long posP = pos0 - gOffStep;
      if (act & 32u) {
        pDiag += lhs[3];
        pRhs += rh[1];
        const long long pos = posP + kk(NE - 1) * SELL_C;
        if (SET) LAG_ST(A.vals + pos, lhs[2]);
        else A.vals[pos] += lhs[2];
      }
      if (kDiag >= 0) {
        const long long pos = posP + (long long)kDiag * SELL_C;
        if (SET) LAG_ST(A.vals + pos, pDiag);
        else A.vals[pos] += pDiag;
      }
      if (SET) LAG_ST(rhs + row - gRowStep, pRhs);
      else rhs[row - gRowStep] += pRhs;
    }
    // slots 1 .. 4 in CSR order.  The rows next door (slots 2, 3) come out of the neighbouring lanes by
    // full-wave DPP shifts, formed right before their use and outside the per-lane conditions (every lane
    // takes part in a shift); the lanes at the ends of the wave keep what they loaded themselves.
    auto slot = [&](int e, const Node &nb) {
      if ((act >> e) & 1u) {
        LAG_EDGE(P, Op::mirrored(E[e], (eid[e] >> 31) != 0), own, nb, lhs, rh);
        diag += lhs[0];
        b += rh[0];
        const long long pos = pos0 + kk(e) * SELL_C;
        if (SET) LAG_ST(A.vals + pos, lhs[1]);
        else A.vals[pos] += lhs[1];
      }
    };
    if (GY > 1 && !farM) {
      double *pn = reinterpret_cast<double *>(&ym);
#pragma unroll
      for (int f = 0; f < NF; ++f) pn[f] = ldsT[((wv - 1) * NF + f) * 64 + lane];
    }
    slot(1, ym);
    {
      Node xm;
      const double *po = reinterpret_cast<const double *>(&own), *pe = reinterpret_cast<const double *>(&en);
      double *pm = reinterpret_cast<double *>(&xm);
#pragma unroll
      for (int f = 0; f < NF; ++f) pm[f] = dpp_from_lower_lane(po[f], pe[f]);
      slot(2, xm);
    }
    {
      Node xp;
      const double *po = reinterpret_cast<const double *>(&own), *pe = reinterpret_cast<const double *>(&en);
      double *pp = reinterpret_cast<double *>(&xp);
#pragma unroll
      for (int f = 0; f < NF; ++f) pp[f] = dpp_from_upper_lane(po[f], pe[f]);
      slot(3, xp);
    }
    if (GY > 1 && !farP) {
      double *pn = reinterpret_cast<double *>(&yp);
#pragma unroll
      for (int f = 0; f < NF; ++f) pn[f] = ldsT[((wv + 1) * NF + f) * 64 + lane];
    }
    slot(4, yp);
    if (SET) {
      for (int k = 0; k < w; ++k)
        if ((zeroMask >> k) & 1u) A.vals[pos0 + (long long)k * SELL_C] = 0.0;
    }
    // ---- step
    prev = own;
    pDiag = diag;
    pRhs = b;
    if (t + 1 < gLen) {
#pragma unroll
      for (int q = 0; q < NE - 1; ++q) eid[q] += eStep[q];
      row += gRowStep;
      pos0 += gOffStep;
    }
  }
  // ---- the last row of the run: its edge to the row above
  if (act & 32u) {
    double lhs[4], rh[2];
    eid[NE - 1] += eLast5;
    LAG_EDGE(P, Op::mirrored(Op::load_edge(P, (long long)(eid[NE - 1] & 0x7FFFFFFFu)), (eid[NE - 1] >> 31) != 0), prev,
             Op::load(P, row + D.ko[NE - 1]), lhs, rh);
    pDiag += lhs[0];
    pRhs += rh[0];
    const long long pos = pos0 + kk(NE - 1) * SELL_C;
    if (SET) LAG_ST(A.vals + pos, lhs[1]);
    else A.vals[pos] += lhs[1];
  }
  if (kDiag >= 0) {
    const long long pos = pos0 + (long long)kDiag * SELL_C;
    if (SET) LAG_ST(A.vals + pos, pDiag);
    else A.vals[pos] += pDiag;
  }
  if (SET) LAG_ST(rhs + row, pRhs);
  else rhs[row] += pRhs;
}

// The lagged kernel with its streamed operands PREFETCHED INTO LDS (round 3, NG_ASM_LAG_PF): what a step reads
// from HBM -- the wave's own 64 rows of every nodal field and the edge data of slots 0, 1, 3, 4 -- is requested one
// step ahead by direct-to-LDS loads (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16 l, no register
// is held, scripts/ldsdma_probe.hip), lanes 0-31 fetching one field and lanes 32-63 the next.  A step
// (A) waits for its prefetch (the wave's one in-order memory counter: everything issued before is done) and
// copies it to registers, (E) issues the prefetch of the NEXT step at once, evaluates its edges while that is in
// flight, and (B) at its end issues the few loads of the next step that are L2 hits (far rows, the end lanes' x
// neighbours) into registers it no longer needs.  Same fluxes in the same order as k_assemble_rows_lag:
// bit-identical.
// GY = 1: a workgroup = the four quarters of one run; each wave owns its LDS region: no barrier, no flags; the
// y neighbours' rows are loads (B).
// GY = 4 (the default; 8: 512-thread workgroups, NG_ASM_LAG_GY): a workgroup = the same quarter of GY runs of
// adjacent lines (finalize: dMarchGroupsP, groups of 2 .. GY lines; single lines take GY = 1); the nodal part of
// a wave's region is double-buffered and the waves read their y neighbours' rows out of the NEIGHBOURING WAVE'S
// region (one barrier per step, right behind the wait (A): every wave's prefetch has landed and nobody reads
// the buffer the next prefetch overwrites).  A row is then fetched from HBM once per group and by the two
// groups next door, and the y neighbours hold no registers.  The far row of a wave at an end of its group --
// the one load that misses the L2 -- is issued right after the edge that used the previous one (see early4),
// not with (B).  The workgroups behind the last item assemble the slices outside the runs (burst_slice).
// Edge ids of the prefetch follow the regular rows' pattern (id = tb + row); a wave's irregular row whose ids
// do not, and a wave at the very end of the edge array (a 16-byte load would leave it), load those edges
// themselves.
template <class Op, int GY> struct PfLayout {
  static constexpr int NF = sizeof(typename Op::Node) / sizeof(double), NEF = sizeof(typename Op::Edge) / sizeof(double);
  static constexpr int NFP = (NF + 1) / 2 * 2;      // nodal fields, padded to pairs (one load = two fields)
  static constexpr int NS = 4;                      // prefetched edge slots: 0, 1, 3, 4
  static constexpr int NB = GY > 1 ? 2 : 1;         // nodal buffers
  static constexpr int kDoubles = (NB * NFP + NS * NEF) * 64;  // per wave
  __host__ __device__ static constexpr int slot(int k) { return k < 2 ? k : k + 1; }
};

// YS (GY = 4; the default since the end of round 3, NG_ASM_LAG_YSHARE=0 turns it off: 256^3 continuity 0.88 -> 0.84 ms,
// scalar 1.10 -> 1.08 ms): an edge between two lines of a group is evaluated ONCE, by the upper line's wave as
// its slot 1; the lower line's shares (the R-row entries of the same evaluation: lhs[2], lhs[3], rhs[1], bit for bit
// what its own mirrored evaluation gives -- the invariance the lagged z edge already rests on) travel through a
// double-buffered LDS slot and are added one step later, in their place of the row's CSR-ordered sums (behind
// slots 0-3, before the z edge above).  17 instead of 20 edge evaluations per group and step; the receiving waves
// need neither the upper line's row nor the edge data of their slot 4 (whose LDS region is one of the two buffers).
// YS = 1: two buffers (the step's parity), one barrier per step: written in step t behind the barrier, read in step
// t + 1 behind the next.  YS = 2: ONE buffer (the slot-4 region: no LDS beyond the plain kernel's) and a second
// barrier right behind the first: the receivers take their shares between the two, the writers write behind the
// second (the waves have just met at the first barrier: the second one finds no stragglers).
template <class Op, bool SET, int GY, int YS = 0>
__global__ __launch_bounds__(GY > 1 ? 64 * GY : SELL_C, GY == 8 ? 1 : 2) void k_assemble_rows_lagpf(SellView A, typename Op::Params P,
                                                                  double *__restrict__ rhs,
                                                                  const int *__restrict__ list, int nItems,
                                                                  long long nEdges, int burstBase)
{
  using Node = typename Op::Node;
  using Edge = typename Op::Edge;
  using L = PfLayout<Op, GY>;
  constexpr int NF = L::NF, NEF = L::NEF, NFP = L::NFP, NS = L::NS;
  constexpr int NE = BURST_NE;
  static_assert((NS * NEF) % 2 == 0, "edge fields are fetched in pairs");
  static_assert(GY == 1 || GY == 4 || GY == 8, "a group is one line per wave");
  constexpr int NW = GY > 1 ? GY : SELL_C / 64;  // waves per workgroup
  constexpr int QS = SELL_C / 64;                // quarters per slice: a group is QS consecutive items
  static_assert(!YS || GY == 4, "y-share: groups of four lines");
  constexpr int SH = 3 * 64;  // a share buffer: three doubles per lane
  static_assert(!YS || NEF * 64 >= SH, "the slot-4 edge region holds one share buffer");
  __shared__ double ldsAll[NW * L::kDoubles + (YS == 1 ? NW * SH : 0)];
  // The workgroups behind the last item take the slices outside the runs (the mesh's boundary planes and lines) in
  // their burst form, one slice each: in the same launch they fill the CUs that the last runs leave one by one,
  // instead of waiting for the whole grid to drain (a launch of their own: 26 of 875 us at 256^3).
  if (GY <= 4 && (int)blockIdx.x >= nItems) {
    static_assert(GY > 4 || NW * L::kDoubles >= NF * SELL_C, "the burst form stages a slice's nodal data in LDS");
    burst_slice<Op, SET>(A, P, rhs, A.march[burstBase + ((int)blockIdx.x - nItems)].start, ldsAll);
    return;
  }
  const int r = threadIdx.x, lane = r & 63;
  const int wv = __builtin_amdgcn_readfirstlane(r >> 6);
  const int item = xcd_run(blockIdx.x, nItems, GY > 1 ? A.marchXcdRun * (SELL_C / 64) : A.marchXcdRun);
  const int xw = GY > 1 ? (item & (QS - 1)) : wv;  // which 64-row quarter of its slices the wave walks
  const int gBase = GY > 1 ? (item / QS) * GY : 0;  // the group's first entry in the list
  // (GY > 1: a group may have fewer than GY lines: run index -1; such a wave only keeps the barriers company)
  const int ridx = GY > 1 ? list[gBase + wv] : (list ? list[item] : item);
  if (GY > 1 && ridx < 0) {
    const int n = A.march[list[gBase]].len * (YS == 2 ? 2 : 1) + (YS ? 1 : 0);
    for (int t = 0; t < n; ++t) __syncthreads();
    return;
  }
  const MarchSeg &G = A.march[ridx];
  const int gStart = G.start, gLen = G.len, gRowStep = G.rowStep;
  const long long gOffStep = G.offStep;
  const int rowW = gStart * SELL_C + xw * 64;  // the wave's first row
  int row = rowW + lane;
  const BurstSlice *const S = A.burst + gStart;
  const BurstRow D = S->reg, X = S->irr[xw];
  const int w = S->w;
  const unsigned long long mI = S->irrMask[xw];
  long long pos0 = S->off + xw * 64 + lane;
  const bool irregular = (mI >> lane) & 1ULL;
  unsigned eStep[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) eStep[q] = irregular ? G.xStep[xw][q] : G.eStep[q];
  const unsigned eLast5 = eStep[NE - 1] * (unsigned)(gLen - 1);
  unsigned xe[NE], xk[NE], xact = 0, xown = 0;  // xown: slots in which the irregular row's ids leave the pattern
  {
    const int rowX = rowW + (int)(X.lane & 63u), nx = (int)(X.info & 15u);
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      xe[q] = 0;
      xk[q] = 0;
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e < nx && X.ko[e] - rowX == D.ko[q]) {
          xe[q] = X.tb[e];
          xk[q] = (X.kl >> (4 * e)) & 15u;
          xact |= 1u << q;
          if (X.tb[e] != D.tb[q] + (unsigned)rowX || G.xStep[xw][q] != G.eStep[q]) xown |= 1u << q;
        }
    }
  }
  unsigned eid[NE], kkAll = 0;
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    eid[q] = irregular ? xe[q] : D.tb[q] + (unsigned)row;
    kkAll |= (irregular ? xk[q] : (D.kl >> (4 * q)) & 15u) << (4 * q);
  }
  auto kk = [&](int q) { return (long long)((kkAll >> (4 * q)) & 15u); };
  const unsigned act = irregular ? xact : (1u << NE) - 1u;
  const unsigned own_ids = irregular ? xown : 0u;
  const bool ownAny = __builtin_amdgcn_readfirstlane((int)(__ballot(own_ids != 0u) != 0ULL)) != 0;
  const unsigned info = irregular ? X.info : D.info;
  const int kDiag = (int)((info >> 4) & 15u) - 1;
  const unsigned zeroMask = (info >> 8) & 255u;
  const bool endLane = (lane == 0 && (act & 4u)) || (lane == 63 && (act & 8u));
  const int endOff = lane == 0 ? -1 : 1;
  // (GY > 1: only the group's first / last line has a far neighbour outside the group)
  const bool farM = GY == 1 || wv == 0,
             farP = GY == 1 || wv == GY - 1 ||
                    __builtin_amdgcn_readfirstlane(list[gBase + min(wv + 1, GY - 1)]) < 0;
  // (GY > 1) a wave at the upper end of its group evaluates its far edge EARLY in the step (its shares wait in
  // two registers for their turn in the CSR-ordered sums), so that the far row's registers are free for the
  // NEXT step's far row for most of the step: the one load of the kernel that misses the L2 is then as far
  // ahead as the prefetch.  A wave at the lower end reloads right after slot 1.  (Groups have two lines or more:
  // finalize gives single lines to the GY = 1 form.)
  const bool early4 = GY > 1 && farP && !farM;
  // ---- the prefetch: ids of the wave's first row in the prefetched slots (wave-uniform, one step ahead)
  double *const ldsW = ldsAll + wv * L::kDoubles;
  double *const ldsE = ldsW + L::NB * NFP * 64;
  const int half = lane >> 5, l2 = 2 * (lane & 31);
  unsigned pfId[NS], pfStep[NS];  // (scalar registers: a vector load here would make the loop wait for its own prefetch)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    pfId[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)(D.tb[L::slot(k)] + (unsigned)rowW));
    pfStep[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)G.eStep[L::slot(k)]);
  }
  auto prefetch = [&](int rowN, int buf) {  // rowN: the wave's first row of the step to fetch; pfId: that step's ids
    double *const dstN = ldsW + buf * NFP * 64;
#pragma unroll
    for (int i = 0; i < NFP / 2; ++i) {
      const double *src = (half ? Op::node_field(P, min(2 * i + 1, NF - 1)) : Op::node_field(P, 2 * i)) + rowN + l2;
      if (2 * i + 1 < NF || !half)
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(dstN + i * 128), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NS * NEF / 2; ++i) {
      const int g0 = 2 * i, g1 = 2 * i + 1;
      const long long id = (long long)((half ? pfId[g1 / NEF] : pfId[g0 / NEF]) & 0x7FFFFFFFu) + l2;
      const double *src = (half ? Op::edge_field(P, g1 % NEF) : Op::edge_field(P, g0 % NEF)) + id;
      // (y-share: a receiving wave's slot-4 region is a share buffer, and it has no use for those edges)
      const bool shareRegion = YS && !farP && (half ? g1 / NEF : g0 / NEF) == NS - 1;
      if (id + 1 < nEdges && !shareRegion)
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(ldsE + i * 128), 16, 0, 0);
    }
  };
  double pDiag = 0.0, pRhs = 0.0;  // the sums of the last step's row, complete but for its last edge(s)
  // (y-share) the two share buffers of wave v: its slot-4 edge region and its slot behind the waves' regions
  auto share_buf = [&](int v, int parity) {
    return YS == 1 && parity ? ldsAll + NW * L::kDoubles + v * SH
                             : ldsAll + v * L::kDoubles + L::NB * NFP * 64 + (NS - 1) * NEF * 64;
  };
  // the receiving wave's slot 4 of the last step: behind slots 0-3 in the row's sums, before the z edge above
  auto take_share = [&](int parity, long long posRow) {
    if (!farP && (act & 16u)) {
      const double *sh = share_buf(wv, parity);
      const double s2 = sh[lane], s3 = sh[64 + lane], sr = sh[128 + lane];
      pDiag += s3;
      pRhs += sr;
      const long long pos = posRow + kk(4) * SELL_C;
      if (SET) LAG_ST(A.vals + pos, s2);
      else A.vals[pos] += s2;
    }
  };
  prefetch(rowW, 0);
  Node prev;
  if (act & 1u) prev = Op::load(P, row + D.ko[0]);
  // (B) the loads that hit the L2 -- far rows, the end lanes' x neighbours, the first lane's edge to the lane
  // below -- are issued at the END of the previous step, into registers that step no longer needs
  Node ym, yp, en;  // (GY > 1: ym doubles as the one far row a wave at the end of its group has)
  Edge e0, E2;      // (kDppArea: e0 = the first lane's edge to the lane below, the others shift in the lower lane's slot 3)
  auto loadB = [&]() {
    if (LagCfg<Op>::kDppArea) {
      if (lane == 0 && (act & 4u)) e0 = Op::load_edge(P, (long long)(eid[2] & 0x7FFFFFFFu));
    } else if (act & 4u) {
      E2 = Op::load_edge(P, (long long)(eid[2] & 0x7FFFFFFFu));
    }
    if (GY == 1) {
      if (act & 2u) ym = LAG_FAR(row + D.ko[1]);
      if (act & 16u) yp = LAG_FAR(row + D.ko[4]);
    }
#ifdef NG_LAG_NOEND
    en = prev;
#else
    if (endLane) en = Op::load(P, row + endOff);
#endif
  };
  loadB();
  if (GY > 1) {  // the first step's far row
    if (farM) { if (act & 2u) ym = LAG_FAR(row + D.ko[1]); }
    else if (farP) { if (act & 16u) ym = LAG_FAR(row + D.ko[4]); }
  }
  for (int t = 0; t < gLen; ++t) {
    // ---- (A) this step's prefetch and (B) have landed (and the last step's stores are acknowledged): LDS -> registers
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    asm volatile("" ::: "memory");
#ifndef NG_LAG_NOBAR
    if (GY > 1) __syncthreads();
#endif
    const int cur = GY > 1 ? (t & 1) : 0;
    const double *const ldsN = ldsW + cur * NFP * 64;
    Node own;
    Edge E[NE - 1];
    {
      double *po = reinterpret_cast<double *>(&own);
#pragma unroll
      for (int f = 0; f < NF; ++f) po[f] = ldsN[f * 64 + lane];
      Op::finish(own);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        double *pe = reinterpret_cast<double *>(&E[L::slot(k)]);
#pragma unroll
        for (int c = 0; c < NEF; ++c) pe[c] = ldsE[(k * NEF + c) * 64 + lane];
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own region may be overwritten from here on
    {
      // the rare wave whose prefetch did not bring every lane's edges: ids off the pattern (ownAny), or the
      // 16-byte loads would have left the edge array.  A wave-uniform branch that ends with a full wait.
      bool slow = ownAny;
#pragma unroll
      for (int k = 0; k < NS; ++k) slow = slow || (long long)(pfId[k] & 0x7FFFFFFFu) + 64 > nEdges;
      if (slow) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
          const int e = L::slot(k);
          const bool mine = ((own_ids >> e) & 1u) || (long long)(pfId[k] & 0x7FFFFFFFu) + 64 > nEdges;
          if (((act >> e) & 1u) && mine) E[e] = Op::load_edge(P, (long long)(eid[e] & 0x7FFFFFFFu));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
    }
    // ---- (E) the next step's prefetch, as early as the region is free
    auto issue_prefetch = [&]() {
      if (t + 1 < gLen) {
#pragma unroll
        for (int k = 0; k < NS; ++k) pfId[k] += pfStep[k];
        prefetch(rowW + (t + 1) * gRowStep, GY > 1 ? (cur ^ 1) : 0);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    issue_prefetch();
    double diag = 0.0, b = 0.0;
    double lhs[4], rh[2];
    if (act & 1u) LAG_EDGE(P, Op::mirrored(E[0], (eid[0] >> 31) != 0), own, prev, lhs, rh);
    if (LagCfg<Op>::kDppArea) {
      const double *ps = reinterpret_cast<const double *>(&E[3]), *p0 = reinterpret_cast<const double *>(&e0);
      double *pd = reinterpret_cast<double *>(&E[2]);
#pragma unroll
      for (int f = 0; f < NEF; ++f) pd[f] = dpp_from_lower_lane(ps[f], p0[f]);
    } else {
      E[2] = E2;
    }
    if (act & 1u) {
      diag += lhs[0];
      b += rh[0];
      const long long pos = pos0 + kk(0) * SELL_C;
      if (SET) LAG_ST(A.vals + pos, lhs[1]);
      else A.vals[pos] += lhs[1];
    }
    if (t > 0) {
      const long long posP = pos0 - gOffStep;
      if (YS) take_share(YS == 1 ? (t - 1) & 1 : 0, posP);  // the last step's slot 4, evaluated by the wave above
      if (act & 32u) {
        pDiag += lhs[3];
        pRhs += rh[1];
        const long long pos = posP + kk(NE - 1) * SELL_C;
        if (SET) LAG_ST(A.vals + pos, lhs[2]);
        else A.vals[pos] += lhs[2];
      }
      if (kDiag >= 0) {
        const long long pos = posP + (long long)kDiag * SELL_C;
        if (SET) LAG_ST(A.vals + pos, pDiag);
        else A.vals[pos] += pDiag;
      }
      if (SET) LAG_ST(rhs + row - gRowStep, pRhs);
      else rhs[row - gRowStep] += pRhs;
    }
    if (YS == 2) {  // the one share buffer: read in front of this barrier, written behind it
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the shares are in registers
      __syncthreads();
    }
    auto slot = [&](int e, const Node &nb) {
      if ((act >> e) & 1u) {
        LAG_EDGE(P, Op::mirrored(E[e], (eid[e] >> 31) != 0), own, nb, lhs, rh);
        diag += lhs[0];
        b += rh[0];
        const long long pos = pos0 + kk(e) * SELL_C;
        if (SET) LAG_ST(A.vals + pos, lhs[1]);
        else A.vals[pos] += lhs[1];
      }
    };
    // a y neighbour inside the group: the neighbouring wave's row of this step, out of its LDS region
    auto group_row = [&](int v) {
      Node nb;
      double *pn = reinterpret_cast<double *>(&nb);
      const double *src = ldsAll + v * L::kDoubles + cur * NFP * 64;
#pragma unroll
      for (int f = 0; f < NF; ++f) pn[f] = src[f * 64 + lane];
      Op::finish(nb);
      return nb;
    };
    // (GY > 1) the far edge of a wave at an end of its group, then at once the NEXT step's far row into the
    // same registers.  ym is not read again in this step on any path: the compiler's wait for it would be a
    // wait for everything in flight, the prefetch included.
    double d4 = 0.0, r4 = 0.0;
    if (GY > 1 && farM) {
      slot(1, ym);
      if (t + 1 < gLen && (act & 2u)) ym = LAG_FAR(row + gRowStep + D.ko[1]);
    } else if (early4) {
      if (act & 16u) {
#ifdef NG_LAG_NOFAREDGE  // (ablation, timing only: the last wave of a group without its fifth evaluation)
        lhs[0] = own.x[0]; lhs[1] = ym.x[0]; rh[0] = E[4].av[0];
#else
        LAG_EDGE(P, Op::mirrored(E[4], (eid[4] >> 31) != 0), own, ym, lhs, rh);
#endif
        d4 = lhs[0];
        r4 = rh[0];
        const long long pos = pos0 + kk(4) * SELL_C;
        if (SET) LAG_ST(A.vals + pos, lhs[1]);
        else A.vals[pos] += lhs[1];
        if (t + 1 < gLen) ym = LAG_FAR(row + gRowStep + D.ko[4]);
      }
    }
    if (GY == 1) slot(1, ym);
    else if (!farM) {
      slot(1, group_row(wv - 1));
      if (YS && (act & 2u)) {  // the lower line's shares of this edge: its slot 4 of this step
        double *sh = share_buf(wv - 1, t & 1);
        sh[lane] = lhs[2];
        sh[64 + lane] = lhs[3];
        sh[128 + lane] = rh[1];
      }
    }
    {
      Node xm;
      const double *po = reinterpret_cast<const double *>(&own), *pe = reinterpret_cast<const double *>(&en);
      double *pm = reinterpret_cast<double *>(&xm);
#pragma unroll
      for (int f = 0; f < NF; ++f) pm[f] = dpp_from_lower_lane(po[f], pe[f]);
      slot(2, xm);
    }
    {
      Node xp;
      const double *po = reinterpret_cast<const double *>(&own), *pe = reinterpret_cast<const double *>(&en);
      double *pp = reinterpret_cast<double *>(&xp);
#pragma unroll
      for (int f = 0; f < NF; ++f) pp[f] = dpp_from_upper_lane(po[f], pe[f]);
      slot(3, xp);
    }
    if (GY == 1) slot(4, yp);
    else if (early4) {
      if (act & 16u) { diag += d4; b += r4; }
    } else if (!farP && !YS) slot(4, group_row(wv + 1));  // (farM && farP: no such group, finalize keeps single lines out)
    if (SET) {
      for (int k = 0; k < w; ++k)
        if ((zeroMask >> k) & 1u) A.vals[pos0 + (long long)k * SELL_C] = 0.0;
    }
    // ---- step
    prev = own;
    pDiag = diag;
    pRhs = b;
    if (t + 1 < gLen) {
#pragma unroll
      for (int q = 0; q < NE - 1; ++q) eid[q] += eStep[q];
      row += gRowStep;
      pos0 += gOffStep;
      loadB();
    }
  }
  if (YS) {  // the last step's slot 4 of the receiving waves
    __syncthreads();
    take_share((gLen - 1) & 1, pos0);
  }
  if (act & 32u) {
    double lhs[4], rh[2];
    eid[NE - 1] += eLast5;
    LAG_EDGE(P, Op::mirrored(Op::load_edge(P, (long long)(eid[NE - 1] & 0x7FFFFFFFu)), (eid[NE - 1] >> 31) != 0), prev,
             Op::load(P, row + D.ko[NE - 1]), lhs, rh);
    pDiag += lhs[0];
    pRhs += rh[0];
    const long long pos = pos0 + kk(NE - 1) * SELL_C;
    if (SET) LAG_ST(A.vals + pos, lhs[1]);
    else A.vals[pos] += lhs[1];
  }
  if (kDiag >= 0) {
    const long long pos = pos0 + (long long)kDiag * SELL_C;
    if (SET) LAG_ST(A.vals + pos, pDiag);
    else A.vals[pos] += pDiag;
  }
  if (SET) LAG_ST(rhs + row, pRhs);
  else rhs[row] += pRhs;
}

// Entry-parallel row gather, persistent: a workgroup of NW waves assembles one QUARTER slice
// (64 rows) per iteration; wave k evaluates the k-th entry of those 64 rows (for w > NW:
// entries k, k + NW, ...), so the loads of a row's edges -- edge data and out-of-quarter
// neighbours -- are all in flight at once instead of one dependent round trip per entry,
// without one thread having to hold them in registers.  For the same number of resident waves
// the rows in flight per XCD shrink ~NW-fold, and the neighbour lines gathered by adjacent
// slices are still in the 4 MiB L2 when they are asked for again (DESIGN.md section 3:
// FETCH_SIZE 6.1 -> 3.9 GB per launch at 256^3).  The workgroups are persistent (one launch
// overhead and one argument fetch per workgroup, the next quarter's slice metadata requested
// while the current one computes) and walk the quarters in the XCD-aware order of the row
// kernels.  The quarter's own nodal data goes through LDS (each wave fetches 1-2 of its
// fields); every wave leaves its rows' shares of diagonal and rhs in LDS and the last wave adds
// them in entry order: the same additions in the same order as k_assemble_rows, bit for bit.
constexpr int QROWS = 64;
template <class Op, bool SET, int NW, int WMAX>
__global__ __launch_bounds__(QROWS * NW) void k_assemble_entries(SellView A, typename Op::Params P,
                                                                  double *__restrict__ rhs)
{
  using Node = typename Op::Node;
  using Edge = typename Op::Edge;
  constexpr int NF = sizeof(Node) / sizeof(double);
  constexpr int QPS = SELL_C / QROWS;  // quarters per slice
  __shared__ double ldsOwn[NF * QROWS];
  __shared__ double ldsD[2][WMAX * QROWS], ldsB[2][WMAX * QROWS];
  __shared__ unsigned char ldsKind[2][WMAX * QROWS];  // 0 nothing, 1 edge share, 2 the diagonal
  const int nQuarters = A.nSlices * QPS;
  const int lane = threadIdx.x & (QROWS - 1), wave = threadIdx.x / QROWS;
  const double *const *fp = reinterpret_cast<const double *const *>(&P.f);
  int buf = 0;
  for (int t = blockIdx.x; t < nQuarters; t += gridDim.x, buf ^= 1) {
    const int qid = xcd_run(t, nQuarters, A.xcdRun * QPS);
    const int slice = qid / QPS, q = qid % QPS;
    const int qBase = slice * SELL_C + q * QROWS;
    const int row = qBase + lane;
    const bool live = row < A.nRows;
    const int nLive = min(QROWS, A.nRows - qBase);  // <= 0: the quarter is beyond the matrix
    const long long off = A.sliceOff[slice];
    const int w = A.sliceWidth[slice];
    const int *kOff = A.kOff + off / SELL_C;
    const unsigned *tBase = A.tBase + off / SELL_C, *tMul = A.tMul + off / SELL_C;
    bool ex = false, exT = false;
    if (live) {
      ex = (A.exMask[(size_t)slice * 4 + q] >> lane) & 1ULL;
      exT = (A.exMaskT[(size_t)slice * 4 + q] >> lane) & 1ULL;
    }
    // first entry of this wave: decode and request everything it needs from memory
    int k = wave;
    unsigned tag = TAG_PAD;
    int c = row;
    long long pos = 0;
    bool isEdge = false, inQ = false;
    Edge E;
    Node nb;
    if (live && k < w) {
      pos = off + (long long)k * SELL_C + q * QROWS + lane;
      tag = exT ? A.tags[pos] : tBase[k] + tMul[k] * (unsigned)row;
      isEdge = tag < TAG_OTHER && (!A.edgeOwned || A.edgeOwned[tag & 0x7FFFFFFFu]);
      if (isEdge) {
        c = ex ? A.cols[pos] : row + kOff[k];
        inQ = (unsigned)(c - qBase) < (unsigned)nLive;
        E = Op::load_edge(P, (long long)(tag & 0x7FFFFFFFu));
        if (!inQ) nb = Op::load(P, c);
      }
    }
    // the quarter's own nodal data: wave v fetches fields v, v + NW, ...
    if (live) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (f % NW == wave) ldsOwn[f * QROWS + lane] = fp[f][row];
    }
    __syncthreads();
    for (; k < w; k += NW) {
      if (k != wave) {  // entries beyond the first (slices wider than NW): one round trip each
        tag = TAG_PAD;
        isEdge = inQ = false;
        if (live) {
          pos = off + (long long)k * SELL_C + q * QROWS + lane;
          tag = exT ? A.tags[pos] : tBase[k] + tMul[k] * (unsigned)row;
          isEdge = tag < TAG_OTHER && (!A.edgeOwned || A.edgeOwned[tag & 0x7FFFFFFFu]);
          if (isEdge) {
            c = ex ? A.cols[pos] : row + kOff[k];
            inQ = (unsigned)(c - qBase) < (unsigned)nLive;
            E = Op::load_edge(P, (long long)(tag & 0x7FFFFFFFu));
            if (!inQ) nb = Op::load(P, c);
          }
        }
      }
      unsigned char kind = 0;
      if (live) {
        if (isEdge) {
          Node own;
          double *po = reinterpret_cast<double *>(&own);
#pragma unroll
          for (int f = 0; f < NF; ++f) po[f] = ldsOwn[f * QROWS + lane];
          Op::finish(own);  // (ldsOwn holds the fields as they are in memory)
          if (inQ) {
            double *pn = reinterpret_cast<double *>(&nb);
#pragma unroll
            for (int f = 0; f < NF; ++f) pn[f] = ldsOwn[f * QROWS + (c - qBase)];
            Op::finish(nb);
          }
          const bool isR = (tag >> 31) != 0;
          double lhs[4], rh[2];
          Op::edge(P, Op::mirrored(E, isR), own, nb, lhs, rh);  // (the row's node plays L)
          const double offd = lhs[1];
          if (SET) A.vals[pos] = offd;
          else A.vals[pos] += offd;
          ldsD[buf][k * QROWS + lane] = lhs[0];
          ldsB[buf][k * QROWS + lane] = rh[0];
          kind = 1;
        } else if (tag == TAG_DIAG) {
          kind = 2;
        } else if (SET) {
          A.vals[pos] = 0.0;
        }
      }
      ldsKind[buf][k * QROWS + lane] = kind;
    }
    // after this barrier ldsOwn may be overwritten (next iteration) and the shares of `buf` are
    // complete; they are next written two iterations on, behind two more barriers
    __syncthreads();
    if (wave == NW - 1 && live) {
      double diag = 0.0, b = 0.0;
      long long diagPos = -1;
      for (int kk = 0; kk < w; ++kk) {
        const unsigned char kind = ldsKind[buf][kk * QROWS + lane];
        if (kind == 1) {
          diag += ldsD[buf][kk * QROWS + lane];
          b += ldsB[buf][kk * QROWS + lane];
        } else if (kind == 2) {
          diagPos = off + (long long)kk * SELL_C + q * QROWS + lane;
        }
      }
      if (diagPos >= 0) {
        if (SET) A.vals[diagPos] = diag;
        else A.vals[diagPos] += diag;
      }
      if (SET) rhs[row] = b;
      else rhs[row] += b;
    }
  }
}

__device__ __forceinline__ void add_plain(double *p, double v) { *p += v; }
__device__ __forceinline__ void add_atomic(double *p, double v) { unsafeAtomicAdd(p, v); }

// Edge-parallel assembly (the reference's loop shape): one thread per edge, pre-computed
// matrix slots.  The two off-diagonal slots of an edge have a single writer; diagonal and
// rhs are shared between the edges of a node and need either colouring (COLORED: launched
// once per colour, plain adds, deterministic) or fp64 atomics (ATOMIC).
template <class Op, bool ATOMIC>
__global__ __launch_bounds__(256) void k_assemble_edges(long long n, long long first,
                                                         const int *__restrict__ perm,
                                                         const int *__restrict__ eL,
                                                         const int *__restrict__ eR,
                                                         const int *__restrict__ slots,
                                                         long long nEdges, typename Op::Params P,
                                                         double *vals, double *rhs,
                                                         const unsigned char *__restrict__ edgeOwned)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long e = perm ? perm[first + t] : first + t;
  if (edgeOwned && !edgeOwned[e]) return;  // export mode: another rank assembles this edge
  const int L = eL[e], R = eR[e];
  const typename Op::Node nL = Op::load(P, L), nR = Op::load(P, R);
  double lhs[4], rh[2];
  Op::edge(P, Op::load_edge(P, e), nL, nR, lhs, rh);
  const int sLL = slots[e], sLR = slots[nEdges + e], sRL = slots[2 * nEdges + e],
            sRR = slots[3 * nEdges + e];
  if (sLL >= 0) {
    if (ATOMIC) { add_atomic(vals + sLL, lhs[0]); add_atomic(rhs + L, rh[0]); }
    else { add_plain(vals + sLL, lhs[0]); add_plain(rhs + L, rh[0]); }
    add_plain(vals + sLR, lhs[1]);
  }
  if (sRR >= 0) {
    if (ATOMIC) { add_atomic(vals + sRR, lhs[3]); add_atomic(rhs + R, rh[1]); }
    else { add_plain(vals + sRR, lhs[3]); add_plain(rhs + R, rh[1]); }
    add_plain(vals + sRL, lhs[2]);
  }
}

template <class Op>
int assemble(nalu_gpu_linsys *h, const typename Op::Params &P, int mode)
{
  if (h->nEdges == 0 || h->nOwned == 0) return materialize_zero(h);
  if (mode == NALU_GPU_SCATTER_ROWS) {
    const bool set = h->lazyZero;
    h->lazyZero = false;
    // NG_ASM_ENTRIES=1 selects the entry-parallel persistent kernel (slices of up to 32
    // entries) instead of the row kernel: 1.6x less HBM traffic but latency-bound and 2x slower
    // as it stands, kept as the starting point for a pipelined version.
    // Grid = what fits the chip at once, a multiple of 8 so that a workgroup stays on its XCD's
    // share of the schedule.
    const bool entries = h->asmVariant == 1;
    const int nQ = h->nSlices * (SELL_C / QROWS);
    timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
    if (entries && h->maxSliceWidth <= 32) {
      void (*kern)(SellView, typename Op::Params, double *) =
          h->maxSliceWidth <= 8 ? (set ? k_assemble_entries<Op, true, 8, 8> : k_assemble_entries<Op, false, 8, 8>)
                                : (set ? k_assemble_entries<Op, true, 8, 32> : k_assemble_entries<Op, false, 8, 32>);
      int perCu = 0, cus = 0;
      NG_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kern, QROWS * 8, 0));
      NG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
      int grid = std::max(1, perCu) * std::max(1, cus);
      grid = std::min((grid + 7) / 8 * 8, (nQ + 7) / 8 * 8);
      kern<<<grid, QROWS * 8, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
    } else if ((h->asmVariant == 4 || ((MarchCfg<Op>::kDefault || (LagCfg<Op>::kDefault && lag_kernel(h))) &&
                                       march_by_default(h))) &&
               h->dBurst && h->dMarch && !h->dEdgeOwned) {
      const int nPat = h->nMarchPat, nRest = h->nMarch - h->nMarchPat;
      if (nPat > 0 && lag_kernel(h)) {
        // The default (round 3): the prefetching kernel on groups of four adjacent lines (k_assemble_rows_lagpf<.., 4>:
        // 256^3 continuity 0.97 -> 0.89 ms, scalar 1.27 -> 1.15 ms; 512^3 7.55 -> 6.90 / 10.0 -> 9.03 ms).
        // NG_ASM_LAG_PF=0: no LDS prefetch (k_assemble_rows_lag); NG_ASM_LAG_GROUPS=0: single runs instead of
        // groups; NG_ASM_LAG_GY=8: groups of eight lines in 512-thread workgroups (measured: slower, 0.93 ms).
        const bool grouped = lag_grouped();
        const bool pf = lag_prefetch(h);
        const int nG = grouped ? h->nMarchGroups : 0, nS = grouped ? h->nMarchSingles : nPat;
        const int *singles = grouped ? h->dMarchSingles : nullptr;
        bool restDone = false;  // the slices outside the runs were taken by the tail workgroups of a launch
        if (grouped && pf) {  // (every pattern run in one launch: incomplete groups have idle waves)
          const int nP = h->nMarchGroupsP, nI = (SELL_C / 64) * nP;  // items: the four quarters of every group
          if (nI == 0) {
          } else if (h->marchGY == 8) {
            if (set) k_assemble_rows_lagpf<Op, true, 8><<<nI, 512, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, 0);
            else k_assemble_rows_lagpf<Op, false, 8><<<nI, 512, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, 0);
          } else {
            const int nT = env_not_zero("NG_ASM_LAG_TAIL") ? nRest : 0;  // (NG_ASM_LAG_TAIL=0: the burst kernel in a launch of its own)
            bool launched = false;
            if (env_not_zero("NG_ASM_LAG_YSHARE")) {  // (the default; NG_ASM_LAG_YSHARE=0: every wave evaluates its own slot 4)
              constexpr int YS = LagCfg<Op>::kYShare;
              if (set) k_assemble_rows_lagpf<Op, true, 4, YS><<<nI + nT, 256, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, nPat);
              else k_assemble_rows_lagpf<Op, false, 4, YS><<<nI + nT, 256, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, nPat);
              launched = true;
            }
            if (launched) {
            } else if (set) k_assemble_rows_lagpf<Op, true, 4><<<nI + nT, 256, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, nPat);
            else k_assemble_rows_lagpf<Op, false, 4><<<nI + nT, 256, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroupsP, nI, h->nEdges, nPat);
            restDone = nT == nRest;
          }
          if (const int n1 = h->nMarchSinglesP) {  // (lines without a neighbour to group with)
            if (set) k_assemble_rows_lagpf<Op, true, 1><<<n1, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchSinglesP, n1, h->nEdges, 0);
            else k_assemble_rows_lagpf<Op, false, 1><<<n1, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchSinglesP, n1, h->nEdges, 0);
          }
        } else if (nG > 0) {
          if (set) k_assemble_rows_lag<Op, true, 4><<<4 * nG, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroups, 4 * nG);
          else k_assemble_rows_lag<Op, false, 4><<<4 * nG, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, h->dMarchGroups, 4 * nG);
        }
        if (grouped && pf) {
        } else if (nS > 0 && pf) {
          if (set) k_assemble_rows_lagpf<Op, true, 1><<<nS, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, singles, nS, h->nEdges, 0);
          else k_assemble_rows_lagpf<Op, false, 1><<<nS, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, singles, nS, h->nEdges, 0);
        } else if (nS > 0) {
          if (set) k_assemble_rows_lag<Op, true, 1><<<nS, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, singles, nS);
          else k_assemble_rows_lag<Op, false, 1><<<nS, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, singles, nS);
        }
        if (nRest > 0 && !restDone) {
          if (set) k_assemble_rows_burst<Op, true><<<nRest, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, nPat);
          else k_assemble_rows_burst<Op, false><<<nRest, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, nPat);
        }
      } else {
        if (nPat > 0) {
          if (set) k_assemble_rows_march<Op, true><<<nPat, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
          else k_assemble_rows_march<Op, false><<<nPat, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
        }
        if (nRest > 0) {
          if (set) k_assemble_rows_burst<Op, true><<<nRest, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, nPat);
          else k_assemble_rows_burst<Op, false><<<nRest, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, nPat);
        }
      }
    } else if (h->asmVariant == 3 && h->dBurst && !h->dEdgeOwned) {
      if (set) k_assemble_rows_burst<Op, true><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, -1);
      else k_assemble_rows_burst<Op, false><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs, -1);
    } else if (h->asmVariant == 2) {
      // NG_ASM_ORDER=1: the slice's out-of-slice entries first (k_slice_order).  HBM traffic falls
      // from 7.3 to 4.7 GB per launch at 256^3, but the 18 extra registers cost a wave per SIMD
      // and the kernel then sits on the L2 -> L1 bandwidth: 1.48 against 1.25 ms.
      if (set) k_assemble_rows<Op, true, true><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
      else k_assemble_rows<Op, false, true><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
    } else if (set) k_assemble_rows<Op, true, false><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
    else k_assemble_rows<Op, false, false><<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), P, h->dRhs);
    timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  NG_TRY(materialize_zero(h));
  NG_TRY(ensure_edge_slots(h));
  if (mode == NALU_GPU_SCATTER_ATOMIC) {
    timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
    k_assemble_edges<Op, true><<<(unsigned)((h->nEdges + 255) / 256), 256, 0, h->stream>>>(
        h->nEdges, 0, nullptr, h->dEdgeL, h->dEdgeR, h->dSlots, h->nEdges, P, h->dVals, h->dRhs, h->dEdgeOwned);
    timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  NG_REQUIRE(mode == NALU_GPU_SCATTER_COLORED, NALU_GPU_EINVAL, "unknown scatter mode %d", mode);
  NG_TRY(ensure_colors(h));
  timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
  for (int c = 0; c < h->nColors; ++c) {
    const long long n = h->colorOff[c + 1] - h->colorOff[c];
    if (n == 0) continue;
    k_assemble_edges<Op, false><<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(
        n, h->colorOff[c], h->dColorPerm, h->dEdgeL, h->dEdgeR, h->dSlots, h->nEdges, P, h->dVals,
        h->dRhs, h->dEdgeOwned);
  }
  timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

bool vec3_ok(const nalu_gpu_vec3 &v) { return v.x && v.y && v.z; }

// ------------------------------------------------------------------ mdot / gradient
__global__ __launch_bounds__(256) void k_mdot(long long nEdges, const int *__restrict__ eL,
                                               const int *__restrict__ eR, ContParams P,
                                               double *__restrict__ mdot,
                                               const unsigned char *__restrict__ edgeOwned)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nEdges) return;
  if (edgeOwned && !edgeOwned[e]) return;  // export mode: the edge's owner computes its mass flow
  double lhsfac;
  mdot[e] = ContOp::flux(P, ContOp::load_edge(P, e), ContOp::load(P, eL[e]), ContOp::load(P, eR[e]), lhsfac);
}

// src/AssembleNodalGradEdgeAlgorithm.C:87-114 as a row gather (same summation order as the
// edge loop on (L,R)-sorted edges)
__global__ __launch_bounds__(SELL_C) void k_nodal_grad(SellView A, nalu_gpu_vec3 area,
                                                        const double *__restrict__ vol,
                                                        const double *__restrict__ q,
                                                        double *__restrict__ gx,
                                                        double *__restrict__ gy,
                                                        double *__restrict__ gz)
{
  const int slice = sched_slice(A, blockIdx.x);
  const int r = threadIdx.x;
  const int row = slice * SELL_C + r;
  if (row >= A.nRows) return;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  const double qi = q[row];
  const double invVol = 1.0 / vol[row];
  double g[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    const unsigned tag = A.tags[pos];
    if (tag >= TAG_OTHER) continue;
    const long long e = tag & 0x7FFFFFFFu;
    if (A.edgeOwned && !A.edgeOwned[e]) continue;  // export mode: only the edges this rank owns (the reference's selector)
    const bool isR = (tag >> 31) != 0;
    const double qn = q[A.cols[pos]];
    const double qip = isR ? 0.5 * (qn + qi) : 0.5 * (qi + qn);
    const double a[3] = {area.x[e], area.y[e], area.z[e]};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double ajQip = a[j] * qip;
      if (isR) g[j] -= ajQip * invVol;
      else g[j] += ajQip * invVol;
    }
  }
  gx[row] = g[0];
  gy[row] = g[1];
  gz[row] = g[2];
}

// src/AssembleNodeSolverAlgorithm.C:84-110 + src/ScalarMassBackwardEulerNodeSuppAlg.C:72-79
__global__ __launch_bounds__(SELL_C) void k_mass_be(SellView A, const double *__restrict__ qN,
                                                     const double *__restrict__ qNp1,
                                                     const double *__restrict__ rhoN,
                                                     const double *__restrict__ rhoNp1,
                                                     const double *__restrict__ vol, double dt,
                                                     double *__restrict__ rhs)
{
  const int slice = blockIdx.x;
  const int r = threadIdx.x;
  const int row = slice * SELL_C + r;
  if (row >= A.nRows) return;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  const double lhsTime = rhoNp1[row] * vol[row] / dt;
  const double res = (rhoNp1[row] * qNp1[row] - qN[row] * rhoN[row]) * vol[row] / dt;
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    if (A.tags[pos] == TAG_DIAG) {
      A.vals[pos] += lhsTime;
      break;
    }
  }
  rhs[row] += (0.0 - res);
}

// src/FieldFunctions.C:24-58
__global__ __launch_bounds__(256) void k_axpby(long long n, double alpha, const double *__restrict__ x,
                                                double beta, double *__restrict__ y)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = alpha * x[i] + beta * y[i];
}

// src/LowMachEquationSystem.C:842-853
__global__ __launch_bounds__(256) void k_project_velocity(long long n, double projTimeScale,
                                                           nalu_gpu_vec3 gNew, nalu_gpu_vec3 gOld,
                                                           const double *__restrict__ rho,
                                                           const unsigned char *__restrict__ projected,
                                                           nalu_gpu_vec3 u)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (projected && !projected[i])) return;
  const double fac = projTimeScale / rho[i];
  const_cast<double *>(u.x)[i] -= fac * (gNew.x[i] - gOld.x[i]);
  const_cast<double *>(u.y)[i] -= fac * (gNew.y[i] - gOld.y[i]);
  const_cast<double *>(u.z)[i] -= fac * (gNew.z[i] - gOld.z[i]);
}

// src/MixtureFractionEquationSystem.C:966-986.  diag (null: no diagnostics): [0], [1] nodes clipped
// below / above, [2], [3] the smallest / largest un-clipped value as order-preserving 64-bit keys
__device__ __forceinline__ unsigned long long ord_key(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__global__ __launch_bounds__(256) void k_update_and_clip(long long n, const double *__restrict__ dz,
                                                          double *__restrict__ z, double *__restrict__ zUF,
                                                          double lowBound, double highBound,
                                                          unsigned long long *__restrict__ diag)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = z[i] + dz[i];
  if (zUF) zUF[i] = v;
  if (v < lowBound) {
    if (diag) { atomicAdd(diag, 1ULL); atomicMin(diag + 2, ord_key(v)); }
    v = lowBound;
  } else if (v > highBound) {
    if (diag) { atomicAdd(diag + 1, 1ULL); atomicMax(diag + 3, ord_key(v)); }
    v = highBound;
  }
  z[i] = v;
}

}  // namespace


extern "C" int nalu_gpu_update_and_clip(nalu_gpu_linsys *h, int64_t n, const double *dz_dev, double *z_dev,
                                        double *z_unfiltered_dev, double delta_clip, double clip_info[4])
{
  NG_FINALIZED(h);
  NG_REQUIRE(n >= 0 && (n == 0 || (dz_dev && z_dev)), NALU_GPU_EINVAL, "update_and_clip: bad arguments");
  const double lowBound = 0.0 - delta_clip, highBound = 1.0 + delta_clip;  // (:946-947)
  unsigned long long *dDiag = nullptr;
  if (clip_info) {  // (the head of the reduce buffer: nothing of the solver is in flight between solves)
    dDiag = (unsigned long long *)h->dReduce;
    unsigned long long *pin = (unsigned long long *)((char *)h->hPinned + 3584);
    pin[0] = pin[1] = 0ULL;
    pin[2] = ~0ULL;
    pin[3] = 0ULL;
    NG_HIP(hipMemcpyAsync(dDiag, pin, 4 * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
  }
  if (n)
    k_update_and_clip<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(n, dz_dev, z_dev, z_unfiltered_dev, lowBound,
                                                                          highBound, dDiag);
  NG_HIP(hipGetLastError());
  if (clip_info) {
    unsigned long long *pin = (unsigned long long *)((char *)h->hPinned + 3584);
    NG_HIP(hipMemcpyAsync(pin, dDiag, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    NG_HIP(hipStreamSynchronize(h->stream));
    auto unkey = [](unsigned long long k) {
      const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
      double v;
      memcpy(&v, &b, sizeof(v));
      return v;
    };
    clip_info[0] = (double)pin[0];
    clip_info[1] = (double)pin[1];
    clip_info[2] = pin[0] ? unkey(pin[2]) : +1.0e16;  // (:949-950)
    clip_info[3] = pin[1] ? unkey(pin[3]) : -1.0e16;
  }
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_assemble_continuity_edges(nalu_gpu_linsys *h,
                                                  const nalu_gpu_continuity_fields *f, double dt,
                                                  double gamma1, double noc_fac,
                                                  double interp_together, int mode)
{
  NG_FINALIZED(h);
  NG_REQUIRE(f && vec3_ok(f->coordinates) && vec3_ok(f->dpdx) && vec3_ok(f->velocity) &&
                 f->pressure && f->density && vec3_ok(f->edge_area),
             NALU_GPU_EINVAL, "continuity fields: null device pointer");
  NG_REQUIRE(gamma1 != 0.0 && dt != 0.0, NALU_GPU_EINVAL, "dt and gamma1 must be non-zero");
  ContParams P;
  P.f = *f;
  P.tau = dt / gamma1;  // projTimeScale (:90)
  P.noc = noc_fac;
  P.interp = interp_together;
  return assemble<ContOp>(h, P, mode);
}

extern "C" int nalu_gpu_assemble_scalar_edges(nalu_gpu_linsys *h, const nalu_gpu_scalar_fields *f,
                                              const nalu_gpu_scalar_options *o, int mode)
{
  NG_FINALIZED(h);
  NG_REQUIRE(f && o && vec3_ok(f->coordinates) && vec3_ok(f->dqdx) && vec3_ok(f->velocity) &&
                 f->scalar_q && f->density && f->diff_flux_coeff && vec3_ok(f->edge_area) &&
                 f->mass_flow_rate,
             NALU_GPU_EINVAL, "scalar fields: null device pointer");
  NG_REQUIRE(o->peclet_kind == NALU_GPU_PECLET_CLASSIC || o->peclet_kind == NALU_GPU_PECLET_TANH,
             NALU_GPU_EINVAL, "unknown peclet function %d", o->peclet_kind);
  ScalParams P;
  P.f = *f;
  P.o = *o;
  return assemble<ScalOp>(h, P, mode);
}

extern "C" int nalu_gpu_assemble_scalar_mass_be(nalu_gpu_linsys *h, const double *qN,
                                                const double *qNp1, const double *rhoN,
                                                const double *rhoNp1, const double *vol, double dt)
{
  NG_FINALIZED(h);
  NG_REQUIRE(qN && qNp1 && rhoN && rhoNp1 && vol && dt != 0.0, NALU_GPU_EINVAL, "bad mass-term input");
  NG_TRY(materialize_zero(h));
  if (h->nSlicesOwned)
    k_mass_be<<<h->nSlicesOwned, SELL_C, 0, h->stream>>>(sell_view(h), qN, qNp1, rhoN, rhoNp1, vol, dt, h->dRhs);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_compute_mdot_edges(nalu_gpu_linsys *h, const nalu_gpu_continuity_fields *f,
                                           double dt, double gamma1, double noc_fac,
                                           double interp_together, double *mdot_dev)
{
  NG_FINALIZED(h);
  NG_REQUIRE(f && mdot_dev && vec3_ok(f->coordinates) && vec3_ok(f->dpdx) && vec3_ok(f->velocity) &&
                 f->pressure && f->density && vec3_ok(f->edge_area),
             NALU_GPU_EINVAL, "mdot: null device pointer");
  ContParams P;
  P.f = *f;
  P.tau = dt / gamma1;
  P.noc = noc_fac;
  P.interp = interp_together;
  if (h->nEdges)
    k_mdot<<<(unsigned)((h->nEdges + 255) / 256), 256, 0, h->stream>>>(h->nEdges, h->dEdgeL, h->dEdgeR, P, mdot_dev,
                                                                       h->dEdgeOwned);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_nodal_grad_edges(nalu_gpu_linsys *h, nalu_gpu_vec3 area, const double *vol,
                                         const double *q, double *gx, double *gy, double *gz)
{
  NG_FINALIZED(h);
  NG_REQUIRE(vec3_ok(area) && vol && q && gx && gy && gz, NALU_GPU_EINVAL, "nodal grad: null pointer");
  // owner-computes mode: the complete sums of the owned nodes (every edge that touches them is local).
  // Export mode: the reference's edge loop -- this rank's own edges only, into the owned AND the
  // shared-not-owned nodes; nalu_gpu_parallel_sum(h, NULL, 3, ...) then plays the driver's part.
  if (h->nSlices)
    k_nodal_grad<<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), area, vol, q, gx, gy, gz);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_field_axpby(nalu_gpu_linsys *h, int64_t n, double alpha, const double *x,
                                    double beta, double *y)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(n >= 0 && (n == 0 || (x && y)), NALU_GPU_EINVAL, "field_axpby: bad arguments");
  NG_DEVICE(h->device);
  if (n) k_axpby<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(n, alpha, x, beta, y);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_project_nodal_velocity(nalu_gpu_linsys *h, int64_t n, double dt,
                                               double gamma1, nalu_gpu_vec3 dpdx_new,
                                               nalu_gpu_vec3 dpdx_old, const double *density,
                                               const unsigned char *projected,
                                               nalu_gpu_vec3 velocity)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(n >= 0 && gamma1 != 0.0, NALU_GPU_EINVAL, "project_nodal_velocity: bad arguments");
  NG_REQUIRE(n == 0 || (vec3_ok(dpdx_new) && vec3_ok(dpdx_old) && density && vec3_ok(velocity)),
             NALU_GPU_EINVAL, "project_nodal_velocity: null device pointer");
  NG_DEVICE(h->device);
  if (n)
    k_project_velocity<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(
        n, dt / gamma1, dpdx_new, dpdx_old, density, projected, velocity);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}
