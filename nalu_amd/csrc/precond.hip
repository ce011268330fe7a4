// precond.hip -- symmetric Gauss-Seidel preconditioner (gfx950).
//
// Ifpack2 RELAXATION "Symmetric Gauss-Seidel", 1 sweep, zero starting solution, damping 1
// (src/LinearSolverConfig.C:71-75; Ifpack2 itself is Trilinos code, not in the tree):
//     z = (D + U)^-1 D (D + L)^-1 r     on the rank's owned block (off-rank couplings are
// ignored, as Ifpack2's processor-local sweeps do).  The sweeps are sequential in the row
// number.  They are reproduced EXACTLY -- same operations in the same order for every row,
// so results are bit-identical to the CPU oracle's sequential loops -- by level scheduling:
// row i goes to level 1 + max(level of its lower-numbered neighbours), rows of one level are
// independent, and one kernel launch per level walks the levels in order (the reverse
// construction for the backward sweep).  A multicolour ordering would be faster but is a
// different preconditioner (different residual history); this one is the reference's.
// Lattice 256^3: 766 levels of ~22k rows each way.
//
// Level-major storage.  The rows of a level are far apart in the SELL layout (255 rows on the
// lattice), so sweeping them in place costs a 128-byte line per 8-byte value.  compute() (once
// per solve) therefore gathers the strictly lower / strictly upper entries into level-major ELL
// blocks -- level l holds cnt_l rows x w_l entries, entry-major, with the column stored as the
// POSITION of that row in the same level order -- and apply() works on vectors permuted to that
// order: r is gathered once, the level kernels stream contiguous values and read t / z of the
// previous levels nearly contiguously, z is scattered back once.  Same operations in the same
// order per row as before: bit-identical results.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "internal.hpp"
#include "sgs_flow.hpp"

using namespace nalu_gpu;

namespace {

// forward: t_i = (r_i - sum_{c<i} a_ic t_c) / d_i        (entries in column order)
__global__ __launch_bounds__(256) void k_sgs_forward(SellView A, const int *__restrict__ rows, int n,
                                                      const SolverScalars *__restrict__ sc,
                                                      const double *__restrict__ dinv,
                                                      const double *__restrict__ r, double *t)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off = A.sliceOff[slice] + lane;
  const int w = A.sliceWidth[slice];
  double s = r[i];
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C;
    const int c = A.cols[pos];
    if (c < i) s -= A.vals[pos] * t[c];
  }
  t[i] = s * dinv[i];
}

// backward: z_i = t_i - (sum_{i<c<n} a_ic z_c) / d_i
__global__ __launch_bounds__(256) void k_sgs_backward(SellView A, const int *__restrict__ rows, int n,
                                                       const SolverScalars *__restrict__ sc,
                                                       const double *__restrict__ dinv,
                                                       const double *__restrict__ t, double *z)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off = A.sliceOff[slice] + lane;
  const int w = A.sliceWidth[slice];
  double s = 0.0;
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C;
    const int c = A.cols[pos];
    if (c > i && c < A.nRows) s += A.vals[pos] * z[c];
  }
  z[i] = t[i] - s * dinv[i];
}

// ---- level-major sweeps
// values of the level-major blocks from the SELL matrix (compute(), once per solve)
// the matrix values in the order of a dataflow schedule: one workgroup per chunk; entry k of compact row q is the
// (kFirst + k)-th entry of its SELL row (FLOW_PAD: no entry, 0.0)
__global__ __launch_bounds__(256) void k_flow_gather_vals(const FlowChunk *__restrict__ chunks, const int *__restrict__ rowOf,
                                                           const unsigned char *__restrict__ kFirst,
                                                           const int *__restrict__ code,
                                                           const long long *__restrict__ sliceOff,
                                                           const double *__restrict__ vals, double *__restrict__ out)
{
  const FlowChunk C = chunks[blockIdx.x];
  for (int q = threadIdx.x; q < C.m; q += blockDim.x) {
    const int i = rowOf[(size_t)C.rowOff + q];
    const long long p0 = sliceOff[i / SELL_C] + i % SELL_C + (long long)(kFirst ? kFirst[(size_t)C.rowOff + q] : 0) * SELL_C;
    for (int k = 0; k < C.w; ++k) {
      const size_t idx = (size_t)(C.entOff + (long long)k * C.m + q);
      out[idx] = code[idx] != FLOW_PAD ? vals[p0 + (long long)k * SELL_C] : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_lm_gather_vals(long long n, const long long *__restrict__ src,
                                                         const double *__restrict__ vals, double *__restrict__ out)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long p = src[i];
  out[i] = p >= 0 ? vals[p] : 0.0;
}

// out[q] = v[rows[q]] (gather) or out[rows[q]] = v[q] (scatter)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_lm_permute(int n, const int *__restrict__ rows, const int *__restrict__ done,
                                                     const double *__restrict__ v, double *__restrict__ out)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  if (SCATTER) out[rows[q]] = v[q];
  else out[q] = v[rows[q]];
}

// one level of the forward sweep: t_q = (r_q - sum_k a_k t[col_k]) * dinv_q, k ascending (= column order)
__global__ __launch_bounds__(256) void k_lm_forward(int cnt, int w, const double *__restrict__ av,
                                                     const int *__restrict__ ac, const int *__restrict__ done,
                                                     const double *__restrict__ rq, const double *__restrict__ dinvq,
                                                     double *tq, int o)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  double s = rq[o + q];
  for (int k = 0; k < w; ++k) {
    const int c = ac[(size_t)k * cnt + q];
    if (c >= 0) s -= av[(size_t)k * cnt + q] * tq[c];
  }
  tq[o + q] = s * dinvq[o + q];
}

// one level of the backward sweep: z_q = t - (sum_k a_k z[col_k]) * dinv_q
__global__ __launch_bounds__(256) void k_lm_backward(int cnt, int w, const double *__restrict__ av,
                                                      const int *__restrict__ ac, const int *__restrict__ done,
                                                      const int *__restrict__ tpos, const double *__restrict__ tq,
                                                      const double *__restrict__ dinvq, double *zq, int o)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  double s = 0.0;
  for (int k = 0; k < w; ++k) {
    const int c = ac[(size_t)k * cnt + q];
    if (c >= 0) s += av[(size_t)k * cnt + q] * zq[c];
  }
  zq[o + q] = tq[tpos[o + q]] - s * dinvq[o + q];
}

// Multicolour SGS ("mt_sgs"): rows of one colour are not coupled, so a sweep is one launch per colour, over a
// colour-major copy of the entries that colour's rows read: those of lower colours in the forward sweep (colours
// ascending), of higher colours in the backward sweep (descending), in the row's column order.  A colour's rows
// are every other row or so of the matrix: read in place they would drag the whole matrix through the caches once
// per colour and sweep (round 2: 1.2 ms per application at 256^3).
//   forward:  t_i = (r_i - sum_{colour(c) < colour(i)} a_ic t_c) dinv_i
//   backward: z_i = t_i - (sum_{colour(c) > colour(i)} a_ic z_c) dinv_i
template <bool FWD>
__global__ __launch_bounds__(256) void k_mc_sweep(int cnt, int w, const double *__restrict__ av, const int *__restrict__ ac,
                                                   const int *__restrict__ rows, const SolverScalars *__restrict__ sc,
                                                   const double *__restrict__ dinv, const double *__restrict__ r,
                                                   const double *__restrict__ t, double *out)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  const int i = rows[q];
  double s = FWD ? r[i] : 0.0;
  for (int k = 0; k < w; ++k) {
    const int c = ac[(size_t)k * cnt + q];
    if (c >= 0) {
      if (FWD) s -= av[(size_t)k * cnt + q] * out[c];
      else s += av[(size_t)k * cnt + q] * out[c];
    }
  }
  out[i] = FWD ? s * dinv[i] : t[i] - s * dinv[i];
}

// set-up on the device: how many entries a row reads in either sweep (the widest row of a colour sets the colour's
// width), then the colour-major columns and entry numbers, in the row's column order
__device__ __forceinline__ int mc_col(const SellView &A, int i, int k, long long off, const int *kOff, bool ex)
{
  return ex ? A.cols[off + (long long)k * SELL_C] : i + kOff[k];
}

__global__ __launch_bounds__(256) void k_mc_count(SellView A, const int *__restrict__ color, int *wF, int *wB)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.nRows) return;
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off0 = A.sliceOff[slice];
  const int *kOff = A.kOff + off0 / SELL_C;
  const bool ex = (A.exMask[(size_t)slice * 4 + (lane >> 6)] >> (lane & 63)) & 1ULL;
  const int ci = color[i], w = A.sliceWidth[slice];
  int cf = 0, cb = 0;
  for (int k = 0; k < w; ++k) {
    const int c = mc_col(A, i, k, off0 + lane, kOff, ex);
    if (c != i && c < A.nRows) {  // (padding: col = row)
      cf += color[c] < ci;
      cb += color[c] > ci;
    }
  }
  // (a stale read only costs an atomic that changes nothing; without the test 16 M atomics queue on a few addresses)
  if (cf > *(volatile int *)(wF + ci)) atomicMax(wF + ci, cf);
  if (cb > *(volatile int *)(wB + ci)) atomicMax(wB + ci, cb);
}

template <bool FWD>
__global__ __launch_bounds__(256) void k_mc_fill(SellView A, const int *__restrict__ color, const int *__restrict__ rows,
                                                  int cnt, int wc, int *__restrict__ ac, unsigned short *__restrict__ ak)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off0 = A.sliceOff[slice];
  const int *kOff = A.kOff + off0 / SELL_C;
  const bool ex = (A.exMask[(size_t)slice * 4 + (lane >> 6)] >> (lane & 63)) & 1ULL;
  const int ci = color[i], w = A.sliceWidth[slice];
  int kk = 0;
  for (int k = 0; k < w; ++k) {
    const int c = mc_col(A, i, k, off0 + lane, kOff, ex);
    if (c != i && c < A.nRows && (FWD ? color[c] < ci : color[c] > ci) && kk < wc) {
      ac[(size_t)kk * cnt + q] = c;
      ak[(size_t)kk * cnt + q] = (unsigned short)k;
      ++kk;
    }
  }
  for (; kk < wc; ++kk) {
    ac[(size_t)kk * cnt + q] = -1;
    ak[(size_t)kk * cnt + q] = 0;
  }
}

// the matrix values of a colour's entries: entry kIdx of the row's SELL storage
__global__ __launch_bounds__(256) void k_mc_gather_vals(int cnt, int w, const int *__restrict__ rows, const int *__restrict__ ac,
                                                         const unsigned short *__restrict__ ak,
                                                         const long long *__restrict__ sliceOff,
                                                         const double *__restrict__ vals, double *__restrict__ av)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  const int i = rows[q];
  const long long p0 = sliceOff[i / SELL_C] + i % SELL_C;
  for (int k = 0; k < w; ++k) {
    const size_t idx = (size_t)k * cnt + q;
    av[idx] = ac[idx] >= 0 ? vals[p0 + (long long)ak[idx] * SELL_C] : 0.0;
  }
}

// ---- dataflow sweeps (schedule: sgs_flow.hpp)
// One launch per sweep, one workgroup per chunk of rows, dispatched in the schedule's order.  Wave 0
// (the consumer) walks the chunk's steps: at most 64 independent rows each, same operations in
// the same order per row as the sequential sweep, so the result is bit-identical to it.  Values of
// the chunk's own earlier rows come from an LDS ring; values of other chunks are read from an
// exchange buffer (the rows other chunks read, in schedule order), which starts as all-ones NaNs
// ("not computed yet", data-as-flag: an 8-byte agent-scope store publishes a value, ~0.6 us from
// store to a polling load on MI355X, scripts/flowbench.hip) -- the consumer asks for them D steps
// ahead and only waits when one is still missing at its step.  Wave 1 (the publisher) stores the
// computed values from the LDS ring to memory; waves 2-4 (the loaders) stream the chunk's rows,
// D + 2 steps or more ahead, into an LDS staging ring: matrix values, ring slots of the entries, r,
// 1/d, row numbers, exchange slots.
// Every spin is bounded; a wave that gives up sets *err, after which no wave waits any more, the
// grid drains and nalu_gpu_solve reports the failure.  Workgroups only wait for chunks earlier in
// the dispatch order (flow_build checks it), which are running or finished.
constexpr unsigned long long FLOW_SENT = ~0ull;
constexpr int FLOW_LOADERS = 3;       // waves 2 to 4 (wave 0: consumer, wave 1: publisher)
constexpr int FLOW_PUB_WINDOW = 256;  // rows the publisher may lag behind the consumer
constexpr int FLOW_MAX_POLL = 1 << 20;      // ~0.6 s of polling one value
constexpr int FLOW_MAX_LDS_SPIN = 1 << 24;  // loaders <-> consumer inside one workgroup

struct FlowView {
  const FlowChunk *chunks;
  const int *rowOf, *pubOf, *code, *steps;
  const double *val;
  int ringMask;  // LDS ring of in-chunk values: ring - 1, ring >= largest lookback + 64
  int maxSteps;  // most steps of any chunk (the step offsets of a chunk are copied to LDS)
  int nVisible;  // slots of the exchange buffer
  int recoverAll;  // a recovery wait ends when the values of all D look-ahead steps are there (one consumer) / of its own step (two)
  int maxPoll;   // polls of one value before a wave gives up (FLOW_MAX_POLL; NG_SGS_MAX_POLL: the tests' way to provoke it)
  unsigned long long *ticket;  // chunks handed out so far, over all launches of this sweep direction
  long long *trace;  // development aid (NG_SGS_TRACE): per chunk, clock at start / after the first step / at the end
  long long *stepTrace;  // development aid (NG_SGS_TRACE_STEPS): per chunk and step: clock after the step, ticks in recover, in wait_prev
};

// D: steps of look-ahead for values of other chunks; W: entries per row at most (a multiple of 4);
// GB: batches of 64 rows a loader wave has in flight; NSR: rows of the staging ring (a power of two:
// D steps of look-ahead + a loader group + what the publisher may still hold)
constexpr int flow_pow2(int v) { int p = 64; while (p < v) p *= 2; return p; }
// (development aid, -DNG_FLOW_STEP_TRACE: NG_SGS_TRACE_STEPS=<path> dumps every chunk's step timeline, scripts/flowsteps.py)
#ifdef NG_FLOW_STEP_TRACE
constexpr bool kStepTrace = true;
#else
constexpr bool kStepTrace = false;
#endif
template <int D, int W, int GB>
struct FlowCfg {
  static constexpr int GROUP = GB * 64;
  static constexpr int NSR = flow_pow2(D * 64 + GROUP + 64);
  static constexpr int NG = NSR / GROUP;
  // per staged row: W values, W/4 packed ring slots, r, 1/d, row, exchange slot, meta, two slots of other chunks
  static constexpr int ROW_BYTES = 8 * W + 2 * W + 8 + 8 + 4 + 4 + 4 + 8;
  static size_t lds_bytes(int ring, int maxSteps)
  {
    return 8u * (size_t)(ring + 2 + 2 * NSR) + (size_t)NSR * ROW_BYTES + 4u * (size_t)(NG + 4 + maxSteps + 2);
  }
};

__device__ __forceinline__ double flow_poll(const double *p, int *err, int maxPoll)
{
  unsigned long long bits;
  int spins = 0;
  for (;;) {
    bits = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits != FLOW_SENT) break;
    ++spins;
    if ((spins & 15) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (spins >= maxPoll) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    // back off: a chunk that starts long before its turn must not flood the memory fabric with polls
    if (spins < 4) __builtin_amdgcn_s_sleep(1);
    else if (spins < 16) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(32);
  }
  return __longlong_as_double((long long)bits);
}

// FWD: out_i = (in_i - sum_{c<i} a_ic out_c) dinv_i              (in = r, out = t)
// else: out_i = in_i - (sum_{i<c<n} a_ic out_c) dinv_i           (in = t, out = z)
// xch: the exchange buffer of this sweep, all-ones NaNs at launch: the rows other chunks read are also
// stored there (agent scope), in their slot of the schedule's numbering
//
// The consumer's step is the unit of the sweep's critical path (~3 n steps on an n^3 lattice, plus
// the hops between chunks), so everything that can be is done by the other waves: the loaders turn
// the entry codes into LDS ring slots (a slot that always holds 0.0 for entries of other chunks and
// for padding), count and locate the entries of other chunks, and pad every row to W entries, so that
// the consumer's usual path has no data-dependent branch.
// NC = 2 (round 3): TWO consumer waves take the chunk's steps in turn (wave 0 the even steps, the last wave the odd
// ones).  A step's chain of dependent operations -- the previous step's values out of the LDS ring, W
// multiply-subtracts, the new values back into the ring -- is a third of the ~150 instructions a step costs one
// wave; the rest (the step's rows out of the staging ring, the look-ahead loads of values of other chunks, the
// bookkeeping) does not depend on the previous step, and the other wave now does it for ITS next step meanwhile.
// The hand-over is the `consumed` counter the publisher already follows: a wave computes step s when the rows
// before off(s) are in the ring (acquire), and releases off(s + 1) after it.  Same operations in the same order
// per row: bit-identical to NC = 1 and to the sequential sweep.
template <bool FWD, int D, int W, int GB, int NC>
__global__ __launch_bounds__(64 * (2 + FLOW_LOADERS + NC - 1)) void k_flow_sweep(FlowView F, const int *__restrict__ done,
                                                                        const double *in,
                                                                        const double *__restrict__ dinv, double *xch,
                                                                        double *out, int *err)
{
  using Cfg = FlowCfg<D, W, GB>;
  constexpr int NSR = Cfg::NSR, NG = Cfg::NG, GROUP = Cfg::GROUP, WQ = W / 4;
  static_assert(W % 4 == 0 && D % 2 == 0, "ring slots are packed four to a word; two stages alternate");
  static_assert(NC == 1 || NC == 2, "one or two consumer waves");
  // (a consumer asks for the staged rows of its D-th next step: NC D + 1 steps ahead of the one it computes; the
  // loaders can only stage them while the publisher is at most NSR - GROUP rows behind that)
  static_assert((NC * D + 1) * 64 + GROUP <= NSR + 1 || NC == 1, "the staging ring is too small for two consumers");
  extern __shared__ double flowLds[];
  // The chunk of this workgroup: the next one of the schedule's order at the moment the workgroup
  // STARTS (a ticket drawn from a counter that every workgroup of every launch draws from exactly
  // once), not blockIdx: a workgroup only ever waits for chunks earlier in that order, and those were
  // drawn by workgroups that are running or finished, whatever order the hardware dispatches in.
  __shared__ int sChunk;
  if (threadIdx.x == 0) sChunk = (int)(atomicAdd(F.ticket, 1ULL) % (unsigned long long)gridDim.x);
  __syncthreads();
  const int chunkId = sChunk;
  if (done && *done) return;
  const int ring = F.ringMask + 1;
  double *tring = flowLds;             // [ring] values of the chunk's last rows, [ring] = 0.0, then [NSR][2]: the
                                       // values of other chunks a staged row needs (put there by the consumer at
                                       // the row's step, so that all W entries of a row are read the same way)
  const int exBase = ring + 2;
  double *sA = tring + exBase + 2 * NSR;  // [W][NSR] matrix values (0.0: padding)
  double *sR = sA + W * NSR;           // r_i (forward) / t_i (backward)
  double *sDV = sR + NSR;              // 1 / d_i
  unsigned long long *sT = (unsigned long long *)(sDV + NSR);  // [WQ][NSR] ring slots of four entries, 16 bits each
  int2 *sG = (int2 *)(sT + WQ * NSR);  // exchange slots of the first two entries of other chunks (-1: none)
  int *sRow = (int *)(sG + NSR);       // matrix row
  int *sPub = sRow + NSR;              // its slot in the exchange buffer (-1: no other chunk reads it)
  int *sM = sPub + NSR;                // k0 | k1 << 8 | next << 16: entries of other chunks (first, second: 255 = none; count)
  int *ready = sM + NSR;               // [NG]: group g staged <=> ready[g % NG] == g + 1
  int *consumed = ready + NG;          // rows the consumer has computed (their values are in tring)
  int *abortFlag = consumed + 1;
  int *published = abortFlag + 1;      // rows the publisher has stored: their staging rows and ring slots are free
  int *sSteps = abortFlag + 3;         // [nSteps + 1]: first row of every step
  long long *sTs = (long long *)(((size_t)(sSteps + F.maxSteps + 2) + 7) & ~(size_t)7);  // (NG_SGS_TRACE_STEPS only)
  const FlowChunk C = F.chunks[chunkId];
  if (threadIdx.x < NG + 4) ready[threadIdx.x] = 0;
  if (threadIdx.x < 2) tring[ring + threadIdx.x] = 0.0;
  for (int i = threadIdx.x; i <= C.nSteps; i += blockDim.x) sSteps[i] = F.steps[C.stepOff + i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = C.m, w = C.w;

  // rows without an entry of another chunk ask for this slot instead (any valid address; one per
  // chunk, or every workgroup would hit the same memory channel twice per step)
  const int dummyRow = F.nVisible ? (int)((unsigned)chunkId % (unsigned)F.nVisible) : 0;

  if (wave == 1) {  // ---- publisher: the computed values out of the LDS ring to memory
    // (the consumer itself issues no store: vector-memory operations retire in order, and a wait for
    // its look-ahead loads would also wait for every older store to be acknowledged by memory)
    int pub = 0, spins = 0;
    while (pub < m) {
      const int have = __hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (have <= pub) {
        if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
        if (++spins >= FLOW_MAX_LDS_SPIN) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          return;
        }
        __builtin_amdgcn_s_sleep(1);  // (leave the issue slots to the consumers that share the SIMD)
        continue;
      }
      spins = 0;
      for (int q0 = pub; q0 < have; q0 += 64) {
        const int q = q0 + lane;
        if (q < have) {
          const double v = tring[q & F.ringMask];
          const int pub = sPub[q & (NSR - 1)];
          out[sRow[q & (NSR - 1)]] = v;  // (only needed after the sweep)
          // rows that other chunks read are published at once.  A computed value whose bits happen to be
          // the "not computed yet" pattern (an all-ones NaN, e.g. from a 0xFF-filled rhs) goes out as
          // another NaN: a genuine NaN must propagate, not stall the chunks that wait for this slot.
          if (pub >= 0) {
            unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (bits == FLOW_SENT) bits = 0x7FF8000000000000ULL;
            __hip_atomic_store((unsigned long long *)(xch + pub), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      pub = have;
      __hip_atomic_store(published, pub, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  if (wave > 1 && wave < 2 + FLOW_LOADERS) {  // ---- loaders
    // One memory round trip per group of GB x 64 rows: the row numbers of a wave's next group are
    // loaded with the data of the current one, and all loads are issued before the wave waits for
    // the ring slots to become free.
    const int nGroups = (m + GROUP - 1) / GROUP;
    int rawNext[GB];
#pragma unroll
    for (int b = 0; b < GB; ++b) {
      const int q = (wave - 2) * GROUP + b * 64 + lane;
      rawNext[b] = q < m ? F.rowOf[(size_t)C.rowOff + q] : 0;
    }
    for (int g = wave - 2; g < nGroups; g += FLOW_LOADERS) {
      int row[GB], pub[GB], cd[GB][W];
      double av[GB][W], rr[GB], dd[GB];
#pragma unroll
      for (int b = 0; b < GB; ++b) {
        const int q = g * GROUP + b * 64 + lane;
        const bool ok = q < m;
        row[b] = ok ? rawNext[b] : -1;
        pub[b] = ok ? F.pubOf[(size_t)C.rowOff + q] : -1;
#pragma unroll
        for (int k = 0; k < W; ++k) {
          cd[b][k] = FLOW_PAD;
          av[b][k] = 0.0;
          if (k < w && ok) {
            const size_t idx = (size_t)(C.entOff + (long long)k * m + q);
            cd[b][k] = F.code[idx];
            av[b][k] = F.val[idx];
          }
        }
        if (ok) {
          rr[b] = in[row[b]];
          dd[b] = dinv[row[b]];
        }
        const int qn = q + FLOW_LOADERS * GROUP;
        rawNext[b] = qn < m ? F.rowOf[(size_t)C.rowOff + qn] : 0;
      }
      const int need = GROUP * (g + 1) - NSR;  // rows the publisher must be done with
      if (need > 0) {
        int spins = 0;
        while (__hip_atomic_load(published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
          if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
          if (++spins >= FLOW_MAX_LDS_SPIN) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
#pragma unroll
      for (int b = 0; b < GB; ++b)
        if (row[b] >= 0) {
          const int q = g * GROUP + b * 64 + lane;
          const int slot = q & (NSR - 1);
          sR[slot] = rr[b];
          sDV[slot] = dd[b];
          sRow[slot] = row[b];
          sPub[slot] = pub[b];
          int g0 = -1, g1 = -1, ne = 0, k0 = 255, k1 = 255;
          unsigned long long packed[WQ];
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const int c = cd[b][k];
            if (c == FLOW_PAD) av[b][k] = 0.0;  // (the schedule's padding carries no value)
            sA[k * NSR + slot] = av[b][k];
            // ring slot of an in-chunk entry; the row's own two slots for the first two entries of other
            // chunks; the slot that holds 0.0 for padding (and for further entries of other chunks: such a row
            // is computed entry by entry)
            const bool ext = c < 0 && c != FLOW_PAD;
            const unsigned ts = c > 0 ? (unsigned)((q - c) & F.ringMask)
                                : ext && ne < 2 ? (unsigned)(exBase + 2 * slot + ne) : (unsigned)ring;
            if (k % 4 == 0) packed[k / 4] = 0;
            packed[k / 4] |= (unsigned long long)ts << (16 * (k % 4));
            if (ext) {
              if (ne == 0) { g0 = ~c; k0 = k; }
              if (ne == 1) { g1 = ~c; k1 = k; }
              ++ne;
            }
          }
#pragma unroll
          for (int j = 0; j < WQ; ++j) sT[j * NSR + slot] = packed[j];
          sG[slot] = make_int2(g0, g1);
          sM[slot] = k0 | (k1 << 8) | (ne << 16);
        }
      __hip_atomic_store(&ready[g % NG], g + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // ---- consumer(s): their instruction stream is the sweep's critical path, the other waves of the CU yield
  __builtin_amdgcn_s_setprio(3);
  const int nSteps = C.nSteps;
  const int cid = wave == 0 ? 0 : 1;                   // which consumer (NC = 2: the last wave is the second one)
  auto G = [&](int i) { return NC * i + cid; };        // the consumer's i-th step
  // first rows of steps vbase .. vbase + 63, one per lane (uniform reads without a memory access)
  int vbase = 0;
  int vstep = sSteps[min(lane, nSteps)];
  auto off = [&](int s) { return __builtin_amdgcn_readlane(vstep, min(s, nSteps) - vbase); };
  int readyUpTo = 0, readyRows = 0;
  bool dead = false;
  long long trPollTicks = 0, trPolls = 0, trEnsureTicks = 0, trRingTicks = 0;  // NG_SGS_TRACE only
  auto ensure = [&](int rowsEnd) {  // rows [0, rowsEnd) are in the staging ring
    if (__builtin_expect(rowsEnd <= readyRows, 1)) return;  // (the loops below stay off the usual path)
    const long long trT0 = F.trace ? wall_clock64() : 0;
    const int ng = (rowsEnd + GROUP - 1) / GROUP;
    while (readyUpTo < ng) {
      int spins = 0;
      while (__hip_atomic_load(&ready[readyUpTo % NG], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < readyUpTo + 1) {
        if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || ++spins >= FLOW_MAX_LDS_SPIN) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          dead = true;
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      ++readyUpTo;
    }
    readyRows = readyUpTo * GROUP;
    if (F.trace) trEnsureTicks += wall_clock64() - trT0;
  };
  auto is_sent = [](double v) { return (unsigned long long)__double_as_longlong(v) == FLOW_SENT; };
  // Values of other chunks for step s, asked for D steps ahead: the first two external entries of
  // every row.  The two loads are issued on EVERY path, steps past the end and rows without such
  // an entry included (they read a row of this chunk and never look at the result), and the loaded
  // registers are not touched before their step: the compiler counts what is in flight per path, and
  // would otherwise wait for all of it at every step.  For the same reason the consumer issues no
  // other vector-memory operation: step offsets and rows come from LDS, the publisher stores.
  auto prefetch = [&](int s, double &e0, double &e1) {
    const int p0 = off(s), p1 = off(s + 1);
    ensure(p1);
    int2 g = make_int2(-1, -1);
    const int q = p0 + lane;
    if (!dead && q < p1) g = sG[q & (NSR - 1)];
    // lanes with nothing to ask for ride on the step's first row's request (the first row of a step of an
    // interior tile has both entries): no memory request of their own
    int f0 = __builtin_amdgcn_readfirstlane(g.x), f1 = __builtin_amdgcn_readfirstlane(g.y);
    f0 = f0 < 0 ? dummyRow : f0;
    f1 = f1 < 0 ? dummyRow : f1;
    g.x = g.x < 0 ? f0 : g.x;
    g.y = g.y < 0 ? f1 : g.y;
    e0 = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)(xch + g.x),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    e1 = __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)(xch + g.y),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  // a step's rows out of the staging ring: read one step ahead, off the chain of dependent values
  struct Stage {
    unsigned long long t[WQ];
    double a[W], own, dv;
    int q, end, meta;
    bool act;
  };
  auto load_stage = [&](int s, Stage &S) {
    const int s0 = off(s), s1 = off(s + 1);
    ensure(s1);
    S.q = s0 + lane;
    S.end = s1;
    S.act = !dead && S.q < s1;
    if (S.act) {
      const int slot = S.q & (NSR - 1);
#pragma unroll
      for (int j = 0; j < WQ; ++j) S.t[j] = sT[j * NSR + slot];
#pragma unroll
      for (int k = 0; k < W; ++k) S.a[k] = sA[k * NSR + slot];
      S.own = sR[slot];
      S.dv = sDV[slot];
      S.meta = sM[slot];
    }
  };
  // does the row need a value of another chunk that its look-ahead loads did not bring?
  auto late = [&](const Stage &S, double e0, double e1) {
    const int k0 = S.meta & 255, k1 = (S.meta >> 8) & 255;
    return S.act && ((k0 != 255 && is_sent(e0)) || (k1 != 255 && is_sent(e1)));
  };
  auto compute = [&](const Stage &S, double e0, double e1) {
    if (S.act) {
      *(double2 *)(tring + exBase + 2 * (S.q & (NSR - 1))) = make_double2(e0, e1);
      double tv[W];
#pragma unroll
      for (int k = 0; k < W; ++k) tv[k] = tring[(S.t[k / 4] >> (16 * (k % 4))) & 0xffff];
      double acc = FWD ? S.own : 0.0;
#pragma unroll
      for (int k = 0; k < W; ++k)  // (padding: 0.0 * 0.0, which changes neither a forward nor a backward partial sum)
        acc = FWD ? acc - S.a[k] * tv[k] : acc + S.a[k] * tv[k];
      if (__builtin_expect(late(S, e0, e1) || (S.meta >> 16) > 2, 0)) {
        // a value that is still not there (or a third entry of another chunk): the row again, entry by
        // entry, polling.  One copy of this loop per step: the unrolled round has to stay inside the
        // instruction cache.
        // (Its codes come from memory: LDS only holds what the usual path needs, so that three
        // workgroups fit a CU.)
        const int slot = S.q & (NSR - 1);
        acc = FWD ? S.own : 0.0;
#pragma nounroll
        for (int k = 0; k < w; ++k) {
          const int c = F.code[(size_t)(C.entOff + (long long)k * m + S.q)];
          if (c == FLOW_PAD) break;
          const double v = c > 0 ? tring[(S.q - c) & F.ringMask] : flow_poll(xch + ~c, err, F.maxPoll);
          const double a = sA[k * NSR + slot];
          acc = FWD ? acc - a * v : acc + a * v;
        }
      }
      tring[S.q & F.ringMask] = FWD ? acc * S.dv : S.own - acc * S.dv;
    }
  };
  // A value asked for D steps ahead was not there yet: this chunk runs right behind a producer.  Going
  // on step by step would cost a memory round trip per step (and every chunk behind this one would
  // inherit that pace).  Instead ask again for the values of steps s .. s + D - 1, all of them at once,
  // until they are all there: one wait per D steps, and the loads that end it bring the values.
  // (Two consumers, F.recoverAll = 0: until those of step s are there -- the other wave computes meanwhile.)
  auto recover = [&](auto &&refetch_missing) {
    const long long trT0 = F.trace ? wall_clock64() : 0;
    int spins = 0;
    while (__builtin_amdgcn_ballot_w64(refetch_missing()) != 0) {
      ++spins;
      if ((spins & 15) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      if (spins >= F.maxPoll) break;  // (the rows' own polls report the failure)
      if (spins >= 16) __builtin_amdgcn_s_sleep(32);
      else if (spins >= 4) __builtin_amdgcn_s_sleep(8);
    }
    if (F.trace) { trPollTicks += wall_clock64() - trT0; ++trPolls; }
  };
  // does step s need a value that is not in (e0, e1)?
  auto missing = [&](int s, double e0, double e1) {
    const int p0 = off(s), p1 = off(s + 1);
    const int q = p0 + lane;
    if (dead || q >= p1) return false;
    const int meta = sM[q & (NSR - 1)];
    return ((meta & 255) != 255 && is_sent(e0)) || (((meta >> 8) & 255) != 255 && is_sent(e1));
  };
  // a ring slot is written again `ring` rows later: the publisher must have stored the old row by then
  int pubSeen = 0;
  auto ring_free = [&](int rowsEnd) {
    if (__builtin_expect(rowsEnd - ring <= pubSeen, 1)) return;
    const long long trT0 = F.trace ? wall_clock64() : 0;
    int spins = 0;
    while ((pubSeen = __hip_atomic_load(published, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < rowsEnd - ring) {
      if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || ++spins >= FLOW_MAX_LDS_SPIN) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        dead = true;
        return;
      }
    }
    if (F.trace) trRingTicks += wall_clock64() - trT0;
  };

  double ex0[D], ex1[D];
  Stage st[2];
  // (NC = 2) the rows before step s are in the ring: the other consumer has released them
  auto wait_prev = [&](int s) {
    if (NC == 1 || s == 0) return;
    const int need = off(s);
    // (polls back to back: a lower priority or a sleep between polls changes nothing, measured)
    int spins = 0;
    while (__hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
      if (__hip_atomic_load(abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || ++spins >= FLOW_MAX_LDS_SPIN) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(abortFlag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        dead = true;
        return;
      }
    }
  };
  const bool tracing = F.trace && lane == 0 && cid == 0;
  if (tracing) F.trace[8 * chunkId] = wall_clock64();
#pragma unroll
  for (int j = 0; j < D; ++j) prefetch(G(j), ex0[j], ex1[j]);
  load_stage(G(0), st[0]);
  for (int ib = 0; G(ib) < nSteps && !dead; ib += D) {
    if (G(ib + 2 * D) + 1 - vbase > 63) {  // this round reads the offsets of steps G(ib) .. G(ib + 2 D) + 1
      vbase = G(ib);
      vstep = sSteps[min(vbase + lane, nSteps)];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int i = ib + j, s = G(i);
      load_stage(G(i + 1), st[(j + 1) & 1]);
      ring_free(st[j & 1].end);
      if (!dead) {
        const Stage &S = st[j & 1];
        long long trRec = 0, trPrev = 0;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(late(S, ex0[j], ex1[j])) != 0, 0)) {
          if (kStepTrace && F.stepTrace) trRec = wall_clock64();
          recover([&]() {
#pragma unroll
            for (int jj = 0; jj < D; ++jj) prefetch(G(i + jj), ex0[(j + jj) % D], ex1[(j + jj) % D]);
            bool miss = false;
#pragma unroll
            for (int jj = 0; jj < D; ++jj)
              if (jj == 0 || F.recoverAll) miss |= missing(G(i + jj), ex0[(j + jj) % D], ex1[(j + jj) % D]);
            return miss;
          });
          if (kStepTrace && F.stepTrace) trRec = wall_clock64() - trRec;
        }
        if (kStepTrace && F.stepTrace) trPrev = wall_clock64();
        wait_prev(s);
        if (kStepTrace && F.stepTrace) trPrev = wall_clock64() - trPrev;
        if (!dead) compute(S, ex0[j], ex1[j]);
        if (kStepTrace && F.stepTrace && lane == 0 && s < nSteps) {
          sTs[3 * s] = wall_clock64();
          sTs[3 * s + 1] = trRec;
          sTs[3 * s + 2] = trPrev;
        }
      }
      if (s < nSteps && !dead)  // hands the step's rows to the publisher (and to the other consumer)
        __hip_atomic_store(consumed, st[j & 1].end, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      prefetch(G(i + D), ex0[j], ex1[j]);
    }
    if (tracing && ib == 0) F.trace[8 * chunkId + 1] = wall_clock64();
    if (tracing && G(ib) <= nSteps / 2) F.trace[8 * chunkId + 2] = wall_clock64();
  }
  if (kStepTrace && F.stepTrace)
    for (int i = lane; G(i) < nSteps; i += 64)
      for (int c = 0; c < 3; ++c) F.stepTrace[((size_t)chunkId * F.maxSteps + G(i)) * 3 + c] = sTs[3 * G(i) + c];
  if (tracing) {
    F.trace[8 * chunkId + 3] = wall_clock64();
    F.trace[8 * chunkId + 4] = trPollTicks;
    F.trace[8 * chunkId + 5] = trPolls;
    F.trace[8 * chunkId + 6] = trEnsureTicks;
    F.trace[8 * chunkId + 7] = trRingTicks;
  }
}

// rows sorted by level (ascending row inside a level) + level offsets
void bucket_by_level(const std::vector<int> &level, int nLevels, std::vector<int> &rows,
                     std::vector<int> &off)
{
  const int n = (int)level.size();
  off.assign((size_t)nLevels + 1, 0);
  for (int i = 0; i < n; ++i) ++off[level[i] + 1];
  for (int l = 0; l < nLevels; ++l) off[l + 1] += off[l];
  rows.resize((size_t)n);
  std::vector<int> cur(off.begin(), off.end() - 1);
  for (int i = 0; i < n; ++i) rows[cur[level[i]]++] = i;
}

}  // namespace

int nalu_gpu::ensure_sgs_levels(nalu_gpu_linsys *h)
{
  if (h->sgsReady) return NALU_GPU_OK;
  const int n = h->nOwned;
  std::vector<int> lf((size_t)n, 0), lb((size_t)n, 0);
  int nf = 0, nb = 0;
  for (int i = 0; i < n; ++i) {  // sequential by construction: a level depends on lower rows
    int l = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c < i) l = std::max(l, lf[c] + 1);
    }
    lf[i] = l;
    nf = std::max(nf, l + 1);
  }
  for (int i = n - 1; i >= 0; --i) {
    int l = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c > i && c < n) l = std::max(l, lb[c] + 1);
    }
    lb[i] = l;
    nb = std::max(nb, l + 1);
  }
  std::vector<int> rows;
  bucket_by_level(lf, nf, rows, h->sgsFwdOff);
  NG_HIP(hipMalloc(&h->dSgsFwdRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dSgsFwdRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  std::vector<int> rowsF = rows;
  bucket_by_level(lb, nb, rows, h->sgsBwdOff);
  NG_HIP(hipMalloc(&h->dSgsBwdRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dSgsBwdRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  const std::vector<int> &rowsB = rows;

  // level-major blocks (see the header comment)
  std::vector<int> posF((size_t)n), posB((size_t)n);
  for (int q = 0; q < n; ++q) { posF[rowsF[q]] = q; posB[rowsB[q]] = q; }
  auto build = [&](bool fwd, const std::vector<int> &order, const std::vector<int> &off,
                   const std::vector<int> &posOf, std::vector<long long> &entOff, std::vector<int> &width,
                   std::vector<long long> &src, std::vector<int> &col) {
    const int nl = (int)off.size() - 1;
    entOff.assign((size_t)nl + 1, 0);
    width.assign((size_t)std::max(nl, 1), 0);
    for (int l = 0; l < nl; ++l) {
      int w = 0;
      for (int q = off[l]; q < off[l + 1]; ++q) {
        const int i = order[q];
        int c0 = 0;
        for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
          const int c = h->colIdx[k];
          c0 += fwd ? (c < i) : (c > i && c < n);
        }
        w = std::max(w, c0);
      }
      width[l] = w;
      entOff[l + 1] = entOff[l] + (long long)w * (off[l + 1] - off[l]);
    }
    src.assign((size_t)std::max<long long>(entOff[nl], 1), -1);
    col.assign((size_t)std::max<long long>(entOff[nl], 1), -1);
    for (int l = 0; l < nl; ++l) {
      const int cnt = off[l + 1] - off[l];
      for (int q = 0; q < cnt; ++q) {
        const int i = order[off[l] + q];
        const long long so = h->hSliceOff[i / SELL_C] + i % SELL_C;
        int kk = 0;
        for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
          const int c = h->colIdx[k];
          if (fwd ? (c < i) : (c > i && c < n)) {
            const size_t idx = (size_t)entOff[l] + (size_t)kk * cnt + q;
            src[idx] = so + (long long)(k - h->rowPtr[i]) * SELL_C;
            col[idx] = posOf[c];
            ++kk;
          }
        }
      }
    }
  };
  std::vector<long long> srcF, srcB;
  std::vector<int> colF, colB, tposB((size_t)std::max(n, 1));
  build(true, rowsF, h->sgsFwdOff, posF, h->lmFwdEntOff, h->lmFwdW, srcF, colF);
  build(false, rowsB, h->sgsBwdOff, posB, h->lmBwdEntOff, h->lmBwdW, srcB, colB);
  for (int q = 0; q < n; ++q) tposB[q] = posF[rowsB[q]];
  auto up = [&](auto &vec, auto **dst) -> int {
    using T = typename std::remove_reference<decltype(vec[0])>::type;
    NG_HIP(hipMalloc(dst, sizeof(T) * vec.size()));
    NG_HIP(hipMemcpy(*dst, vec.data(), sizeof(T) * vec.size(), hipMemcpyHostToDevice));
    return NALU_GPU_OK;
  };
  NG_TRY(up(srcF, &h->dLmFwdSrc)); NG_TRY(up(colF, &h->dLmFwdCol));
  NG_TRY(up(srcB, &h->dLmBwdSrc)); NG_TRY(up(colB, &h->dLmBwdCol));
  NG_TRY(up(tposB, &h->dLmTpos));
  h->lmFwdN = (long long)srcF.size();
  h->lmBwdN = (long long)srcB.size();
  NG_HIP(hipMalloc(&h->dLmFwdVal, sizeof(double) * srcF.size()));
  NG_HIP(hipMalloc(&h->dLmBwdVal, sizeof(double) * srcB.size()));
  const size_t nv = sizeof(double) * (size_t)std::max(n, 1);
  NG_HIP(hipMalloc(&h->dLmDinvF, nv)); NG_HIP(hipMalloc(&h->dLmDinvB, nv));
  NG_HIP(hipMalloc(&h->dLmR, nv)); NG_HIP(hipMalloc(&h->dLmT, nv)); NG_HIP(hipMalloc(&h->dLmZ, nv));
  h->sgsReady = true;
  return NALU_GPU_OK;
}

// ---- which schedule runs the exact SGS sweeps
//   NG_SGS_MODE=flow (default): the dataflow sweeps, one launch each way (level launches if the
//   matrix has rows with more than 16 lower or upper entries); levels: one launch per level on the
//   level-major copy (round 2's first version); inplace (or NG_SGS_INPLACE=1): one launch per level
//   on the SELL matrix itself (round 1).  All three give bit-identical results.
namespace {
enum { SGS_FLOW = 0, SGS_LEVELS = 1, SGS_INPLACE = 2 };
int sgs_mode()
{
  const char *m = getenv("NG_SGS_MODE");
  if (m && !strcmp(m, "levels")) return SGS_LEVELS;
  if (m && !strcmp(m, "inplace")) return SGS_INPLACE;
  if (getenv("NG_SGS_INPLACE") && atoi(getenv("NG_SGS_INPLACE")) != 0) return SGS_INPLACE;
  return SGS_FLOW;
}
// rows of at most 4 entries a side (7-point graphs): look-ahead of 2 or 4 steps (NG_SGS_DEPTH);
// up to 16 (27-point graphs): 2 steps
using FlowA4 = FlowCfg<4, 4, 2>;
using FlowA2 = FlowCfg<2, 4, 2>;
using FlowB = FlowCfg<2, 16, 1>;
constexpr size_t FLOW_MAX_LDS = 96 * 1024;

template <class V, class T>
int flow_upload(const V &v, T **dst)
{
  NG_HIP(hipMalloc(dst, sizeof(T) * std::max<size_t>(v.size(), 1)));
  NG_HIP(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return NALU_GPU_OK;
}

template <bool FWD>
int flow_launch(nalu_gpu_linsys *h, const int *done, const double *in, double *out)
{
  const nalu_gpu_linsys::FlowDir &fd = h->flow[FWD ? 0 : 1];
  FlowView F;
  F.chunks = (const FlowChunk *)fd.dChunks;
  F.rowOf = fd.dRowOf;
  F.code = fd.dCode;
  F.steps = fd.dSteps;
  F.val = fd.dVal;
  F.ringMask = fd.ring - 1;
  F.maxSteps = fd.maxSteps;
  F.pubOf = fd.dPubOf;
  F.nVisible = fd.nVisible;
  // (read per launch: the tests shrink the bound to provoke a give-up)
  F.maxPoll = getenv("NG_SGS_MAX_POLL") ? std::max(1, atoi(getenv("NG_SGS_MAX_POLL"))) : FLOW_MAX_POLL;
  F.ticket = fd.dTicket;
  // the exchange buffer starts as "nothing computed yet"
  NG_HIP(hipMemsetAsync(fd.dXch, 0xFF, sizeof(double) * (size_t)std::max(fd.nVisible, 1), h->stream));
  F.trace = nullptr;
  const char *tracePath = getenv("NG_SGS_TRACE");
  if (tracePath) {
    NG_HIP(hipMalloc(&F.trace, sizeof(long long) * 8 * (size_t)fd.nChunks));
    NG_HIP(hipMemsetAsync(F.trace, 0, sizeof(long long) * 8 * (size_t)fd.nChunks, h->stream));
  }
  F.stepTrace = nullptr;
  const char *stepPath = kStepTrace ? getenv("NG_SGS_TRACE_STEPS") : nullptr;
  const size_t stepN = (size_t)fd.nChunks * (size_t)fd.maxSteps * 3;
  size_t ldsBytes = fd.ldsBytes;
  if (stepPath) {
    NG_HIP(hipMalloc(&F.stepTrace, sizeof(long long) * stepN));
    NG_HIP(hipMemsetAsync(F.stepTrace, 0, sizeof(long long) * stepN, h->stream));
    ldsBytes += 8 + 24 * (size_t)(fd.maxSteps + 1);
  }
  int *err = (int *)(h->dReduce + REDUCE_CAP) + 2;
  const dim3 block(64 * (2 + FLOW_LOADERS)), block2(64 * (3 + FLOW_LOADERS));
  // two consumer waves for the 7-point rows at the default depth (NG_SGS_CONSUMERS=1: one)
  const bool two = !(getenv("NG_SGS_CONSUMERS") && atoi(getenv("NG_SGS_CONSUMERS")) == 1);
  // One consumer: a recovery wait that only asked for its own step would cost a memory round trip per step.  Two: the
  // other wave computes meanwhile, and waiting for the second look-ahead step as well puts two more producer steps
  // (0.8 us) into every hop (profiles/r3b_sgs_steps.txt; NG_SGS_RECOVER_ALL=1: the old behaviour; -1.5 ... -3 %)
  F.recoverAll = getenv("NG_SGS_RECOVER_ALL") ? atoi(getenv("NG_SGS_RECOVER_ALL")) != 0 : !(two && fd.depth != 4 && fd.wMax <= 4);
  if (fd.depth == 4 && fd.wMax <= 4)
    k_flow_sweep<FWD, 4, 4, 2, 1><<<fd.nChunks, block, ldsBytes, h->stream>>>(F, done, in, h->dDinv, fd.dXch, out, err);
  else if (fd.wMax <= 4 && two)
    k_flow_sweep<FWD, 2, 4, 2, 2><<<fd.nChunks, block2, ldsBytes, h->stream>>>(F, done, in, h->dDinv, fd.dXch, out, err);
  else if (fd.wMax <= 4)
    k_flow_sweep<FWD, 2, 4, 2, 1><<<fd.nChunks, block, ldsBytes, h->stream>>>(F, done, in, h->dDinv, fd.dXch, out, err);
  else
    k_flow_sweep<FWD, 2, 16, 1, 1><<<fd.nChunks, block, ldsBytes, h->stream>>>(F, done, in, h->dDinv, fd.dXch, out, err);
  NG_HIP(hipGetLastError());
  if (tracePath) {  // development aid: the last sweep's per-chunk clocks (100 MHz ticks), one file per direction
    std::vector<long long> t((size_t)fd.nChunks * 8);
    NG_HIP(hipStreamSynchronize(h->stream));
    NG_HIP(hipMemcpy(t.data(), F.trace, sizeof(long long) * t.size(), hipMemcpyDeviceToHost));
    (void)hipFree(F.trace);
    FILE *f = t[3] ? fopen((std::string(tracePath) + (FWD ? ".fwd" : ".bwd")).c_str(), "w") : nullptr;  // (a sweep skipped on the device's done flag leaves zeros)
    if (f) {
      for (int c = 0; c < fd.nChunks; ++c)
        fprintf(f, "%d %lld %lld %lld %lld %lld %lld %lld %lld\n", c, t[8 * c], t[8 * c + 1], t[8 * c + 2], t[8 * c + 3],
                t[8 * c + 4], t[8 * c + 5], t[8 * c + 6], t[8 * c + 7]);
      fclose(f);

    }
  }
  if (stepPath) {
    std::vector<long long> t(stepN);
    NG_HIP(hipStreamSynchronize(h->stream));
    NG_HIP(hipMemcpy(t.data(), F.stepTrace, sizeof(long long) * stepN, hipMemcpyDeviceToHost));
    (void)hipFree(F.stepTrace);
    bool any = false;
    for (size_t i = 0; i < stepN && !any; i += 3) any = t[i] != 0;
    FILE *f = any ? fopen((std::string(stepPath) + (FWD ? ".fwd" : ".bwd")).c_str(), "wb") : nullptr;
    if (f) {
      const long long hdr[2] = {fd.nChunks, fd.maxSteps};
      fwrite(hdr, sizeof(long long), 2, f);
      fwrite(t.data(), sizeof(long long), stepN, f);
      fclose(f);
    }
  }
  return NALU_GPU_OK;
}

// builds the dataflow schedule of both sweeps (once per graph); flowState = -1 if it cannot be used
int ensure_sgs_flow(nalu_gpu_linsys *h)
{
  if (h->flowState != 0) return NALU_GPU_OK;
  const int n = h->nOwned;
  h->flowState = -1;
  if (n == 0) return NALU_GPU_OK;
  int chunkRows = getenv("NG_SGS_CHUNK") ? atoi(getenv("NG_SGS_CHUNK")) : 1024;
  chunkRows = std::min(std::max(chunkRows, 64), 2048);
  int By = 6, Bz = 6;  // (profiles/r2_sgsflow_tiles.json; r3b_sgs_steps.txt: 5 x 6 is 3 % ahead at 256^3, 3-4 % behind at 64^3 / 128^3)
  if (const char *t = getenv("NG_SGS_TILE")) {
    if (sscanf(t, "%d,%d", &By, &Bz) != 2) By = Bz = 0;  // anything else: contiguous chunks only
  }
  int hostThreads = getenv("NG_HOST_THREADS") ? atoi(getenv("NG_HOST_THREADS")) : (int)std::thread::hardware_concurrency();
  hostThreads = std::min(std::max(hostThreads, 2), 16);
  // (NG_SGS_VERBOSE: the phases of the set-up on stderr)
  const bool verbose = env_not_zero("NG_SGS_VERBOSE");
  auto tPrev = std::chrono::steady_clock::now();
  auto phase = [&](const char *name) {
    if (!verbose) return;
    const auto tNow = std::chrono::steady_clock::now();
    fprintf(stderr, "[nalu_gpu sgs set-up] %-28s %.3f s\n", name, std::chrono::duration<double>(tNow - tPrev).count());
    tPrev = tNow;
  };
  FlowBuf<int> chunkOf;
  FlowTiling T = flow_chunking(n, h->rowPtr.data(), h->colIdx.data(), chunkRows, By, Bz, chunkOf, hostThreads);
  phase("chunking");
  FlowSchedule S[2];
  for (int attempt = 0; attempt < 2; ++attempt) {
    // both sweeps at once, each on half of the host threads (16 at most: a GPU's share of the node's cores)
    std::string why[2];
    {
      auto build = [&](int dir) {  // (an exception must not leave a thread: it would end the process)
        try {
          why[dir] = flow_build(n, h->rowPtr.data(), h->colIdx.data(), chunkOf, T.nChunks, dir == 0, S[dir], hostThreads / 2);
        } catch (const std::exception &e) {
          why[dir] = std::string("exception: ") + e.what();
        }
      };
      std::thread other(build, 1);
      build(0);
      other.join();
      for (int dir = 0; dir < 2; ++dir)
        NG_REQUIRE(why[dir].compare(0, 10, "exception:") != 0, NALU_GPU_EHIP, "sgs schedule (host): %s", why[dir].c_str());
    }
    bool ok = why[0].empty() && why[1].empty();
    for (int dir = 0; dir < 2 && ok; ++dir) ok = S[dir].maxLookback + FLOW_STEP <= 4096;
    if (ok) break;
    if (!T.tiled) return NALU_GPU_OK;  // contiguous chunks are always a valid order: only the ring can fail
    T = flow_chunking(n, h->rowPtr.data(), h->colIdx.data(), chunkRows, 0, 0, chunkOf, hostThreads);
  }
  phase("schedules (both sweeps)");
  for (int dir = 0; dir < 2; ++dir) {
    nalu_gpu_linsys::FlowDir &fd = h->flow[dir];
    const FlowSchedule &s = S[dir];
    int ring = 64;
    while (ring < s.maxLookback + FLOW_STEP + FLOW_PUB_WINDOW) ring *= 2;
    if (ring > 8192 || s.wMax > 16) return NALU_GPU_OK;
    int depth = getenv("NG_SGS_DEPTH") ? atoi(getenv("NG_SGS_DEPTH")) : 2;
    depth = depth >= 4 ? 4 : 2;  // (8 steps of look-ahead no longer fit the instruction cache and the LDS)
    fd.depth = s.wMax <= 4 ? depth : 2;
    fd.ldsBytes = s.wMax > 4       ? FlowB::lds_bytes(ring, s.maxSteps)
                  : fd.depth == 4 ? FlowA4::lds_bytes(ring, s.maxSteps)
                                  : FlowA2::lds_bytes(ring, s.maxSteps);
    if (getenv("NG_SGS_LDS_PAD")) fd.ldsBytes += (size_t)atoi(getenv("NG_SGS_LDS_PAD"));  // development aid: fewer workgroups per CU
    if (fd.ldsBytes > FLOW_MAX_LDS) return NALU_GPU_OK;
    fd.ring = ring;
    fd.wMax = s.wMax;
    fd.maxSteps = s.maxSteps;
    fd.nChunks = (int)s.chunks.size();
    fd.nEnt = (long long)s.code.size();
  }
  for (int dir = 0; dir < 2; ++dir) {
    nalu_gpu_linsys::FlowDir &fd = h->flow[dir];
    FlowSchedule &s = S[dir];
    FlowChunk *dc = nullptr;
    NG_TRY(flow_upload(s.chunks, &dc));
    fd.dChunks = dc;
    NG_TRY(flow_upload(s.rowOf, &fd.dRowOf));
    NG_TRY(flow_upload(s.pubOf, &fd.dPubOf));
    fd.nVisible = s.nVisible;
    NG_HIP(hipMalloc(&fd.dXch, sizeof(double) * (size_t)std::max(s.nVisible, 1)));
    NG_HIP(hipMalloc(&fd.dTicket, sizeof(unsigned long long)));
    NG_HIP(hipMemset(fd.dTicket, 0, sizeof(unsigned long long)));
    NG_TRY(flow_upload(s.code, &fd.dCode));
    NG_TRY(flow_upload(s.steps, &fd.dSteps));
    if (dir == 1) NG_TRY(flow_upload(s.kFirst, &fd.dKFirst));
    NG_HIP(hipMalloc(&fd.dVal, sizeof(double) * (size_t)fd.nEnt));
    s = FlowSchedule();
  }
  phase("upload");
  NG_HIP(hipMalloc(&h->dFlowT, sizeof(double) * (size_t)n));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<true, 4, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<false, 4, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<true, 2, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<false, 2, 4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<true, 2, 4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<false, 2, 4, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<true, 2, 16, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  NG_HIP(hipFuncSetAttribute((const void *)k_flow_sweep<false, 2, 16, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLOW_MAX_LDS));
  h->flowTiled = T.tiled;
  h->flowTile[0] = T.Q; h->flowTile[1] = T.P; h->flowTile[2] = T.By; h->flowTile[3] = T.Bz;
  h->flowState = 1;
  return NALU_GPU_OK;
}

bool flow_active(const nalu_gpu_linsys *h) { return sgs_mode() == SGS_FLOW && h->flowState == 1; }
}  // namespace

// a dataflow sweep gave up (nalu_gpu_solve): this handle uses the level schedule from now on
void nalu_gpu::sgs_flow_disable(nalu_gpu_linsys *h)
{
  h->flowState = -1;
  h->flowGaveUp = true;
}

int nalu_gpu::sgs_prepare(nalu_gpu_linsys *h)
{
  if (sgs_mode() == SGS_FLOW) {
    NG_TRY(ensure_sgs_flow(h));
    if (h->flowState == 1) return NALU_GPU_OK;
  }
  return ensure_sgs_levels(h);
}

// Ifpack2 compute() for SGS: refresh the level-major copies of the matrix values and of dinv
// (the matrix changes between solves, the level structure does not).  dDinv must be ready.
int nalu_gpu::sgs_compute(nalu_gpu_linsys *h)
{
  NG_TRY(sgs_prepare(h));
  const int n = h->nOwned;
  if (n == 0) return NALU_GPU_OK;
  if (flow_active(h)) {  // the values in schedule order
    for (auto &fd : h->flow)
      k_flow_gather_vals<<<fd.nChunks, 256, 0, h->stream>>>((const FlowChunk *)fd.dChunks, fd.dRowOf, fd.dKFirst, fd.dCode,
                                                            h->dSliceOff, h->dVals, fd.dVal);
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  k_lm_gather_vals<<<(unsigned)((h->lmFwdN + 255) / 256), 256, 0, h->stream>>>(h->lmFwdN, h->dLmFwdSrc, h->dVals,
                                                                                h->dLmFwdVal);
  k_lm_gather_vals<<<(unsigned)((h->lmBwdN + 255) / 256), 256, 0, h->stream>>>(h->lmBwdN, h->dLmBwdSrc, h->dVals,
                                                                                h->dLmBwdVal);
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsFwdRows, nullptr, h->dDinv, h->dLmDinvF);
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsBwdRows, nullptr, h->dDinv, h->dLmDinvB);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// z = SGS(r) on the owned rows (level-major sweeps; sgs_compute must have run for the current
// matrix).  r and z may alias each other.  NG_SGS_INPLACE=1 selects the first implementation (the
// sweeps on the SELL matrix itself, 5x slower at 256^3), kept as the cross-check of the tests.
int nalu_gpu::apply_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone)
{
  NG_TRY(sgs_prepare(h));
  if (flow_active(h)) {
    if (h->nOwned == 0) return NALU_GPU_OK;
    const int *done = useDone ? &h->dScal->done : nullptr;
    // t = (D + L)^-1 r into dFlowT, then z = t - D^-1 U z (each launch clears its exchange buffer first)
    NG_TRY(flow_launch<true>(h, done, r, h->dFlowT));
    NG_TRY(flow_launch<false>(h, done, h->dFlowT, z));
    return NALU_GPU_OK;
  }
  const int nf = (int)h->sgsFwdOff.size() - 1, nb = (int)h->sgsBwdOff.size() - 1;
  const bool inplace = sgs_mode() == SGS_INPLACE;
  if (inplace) {
    const SellView A = sell_view(h);
    const SolverScalars *sc = useDone ? h->dScal : nullptr;
    for (int l = 0; l < nf; ++l) {
      const int o = h->sgsFwdOff[l], cnt = h->sgsFwdOff[l + 1] - o;
      k_sgs_forward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dSgsFwdRows + o, cnt, sc, h->dDinv, r,
                                                              h->dT);
    }
    for (int l = 0; l < nb; ++l) {
      const int o = h->sgsBwdOff[l], cnt = h->sgsBwdOff[l + 1] - o;
      k_sgs_backward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dSgsBwdRows + o, cnt, sc, h->dDinv,
                                                               h->dT, z);
    }
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  const int n = h->nOwned;
  if (n == 0) return NALU_GPU_OK;
  const int *done = useDone ? &h->dScal->done : nullptr;
  // ~2 x (number of levels) dependent launches (1532 at 256^3), each a kernel boundary plus two
  // dependent memory round trips (column positions -> t values): ~6 us per level, latency-bound.
  // (Replaying them from a HIP graph was measured and changes nothing: the time is on the device.)
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsFwdRows, done, r, h->dLmR);
  for (int l = 0; l < nf; ++l) {
    const int o = h->sgsFwdOff[l], cnt = h->sgsFwdOff[l + 1] - o;
    k_lm_forward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, h->lmFwdW[l], h->dLmFwdVal + h->lmFwdEntOff[l],
                                                           h->dLmFwdCol + h->lmFwdEntOff[l], done, h->dLmR,
                                                           h->dLmDinvF, h->dLmT, o);
  }
  for (int l = 0; l < nb; ++l) {
    const int o = h->sgsBwdOff[l], cnt = h->sgsBwdOff[l + 1] - o;
    k_lm_backward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, h->lmBwdW[l], h->dLmBwdVal + h->lmBwdEntOff[l],
                                                            h->dLmBwdCol + h->lmBwdEntOff[l], done, h->dLmTpos,
                                                            h->dLmT, h->dLmDinvB, h->dLmZ, o);
  }
  k_lm_permute<true><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsBwdRows, done, h->dLmZ, z);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// greedy colouring in row order (the oracle's orc_greedy_colors): colour of row i = smallest
// colour none of its lower-numbered owned neighbours has
int nalu_gpu::ensure_mc_colors(nalu_gpu_linsys *h)
{
  if (h->mcReady) return NALU_GPU_OK;
  const int n = h->nOwned;
  std::vector<int> color((size_t)n, 0);
  int nc = 0;
  for (int i = 0; i < n; ++i) {
    unsigned long long used = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c < i) used |= 1ull << color[c];
    }
    int col = 0;
    while (col < 63 && ((used >> col) & 1ull)) ++col;
    NG_REQUIRE(col < 63, NALU_GPU_EUNSUPPORTED, "mt_sgs: the graph needs more than 63 colours");
    color[i] = col;
    nc = std::max(nc, col + 1);
  }
  std::vector<int> rows;
  bucket_by_level(color, nc, rows, h->mcOff);
  NG_HIP(hipMalloc(&h->dMcRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dMcRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  // colour-major entries of both sweeps, built on the device from the SELL matrix's own columns
  NG_REQUIRE(h->maxSliceWidth <= 65535, NALU_GPU_EUNSUPPORTED, "mt_sgs: a row has more than 65535 entries");
  const SellView A = sell_view(h);
  int *dColor = nullptr, *dW = nullptr;
  NG_HIP(hipMalloc(&dColor, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMalloc(&dW, sizeof(int) * 2 * (size_t)std::max(nc, 1)));
  NG_HIP(hipMemcpyAsync(dColor, color.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipMemsetAsync(dW, 0, sizeof(int) * 2 * (size_t)std::max(nc, 1), h->stream));
  if (n) k_mc_count<<<(n + 255) / 256, 256, 0, h->stream>>>(A, dColor, dW, dW + nc);
  std::vector<int> wHost(2 * (size_t)std::max(nc, 1), 0);
  NG_HIP(hipMemcpyAsync(wHost.data(), dW, sizeof(int) * 2 * (size_t)std::max(nc, 1), hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  for (int dir = 0; dir < 2; ++dir) {
    nalu_gpu_linsys::McDir &M = h->mc[dir];
    M.entOff.assign((size_t)nc + 1, 0);
    M.width.assign((size_t)std::max(nc, 1), 0);
    for (int col = 0; col < nc; ++col) {
      M.width[(size_t)col] = wHost[(size_t)dir * nc + col];
      M.entOff[(size_t)col + 1] = M.entOff[(size_t)col] + (long long)M.width[(size_t)col] * (h->mcOff[col + 1] - h->mcOff[col]);
    }
    M.nEnt = M.entOff[(size_t)nc];
    const size_t ne = (size_t)std::max<long long>(M.nEnt, 1);
    NG_HIP(hipMalloc(&M.dCol, sizeof(int) * ne));
    NG_HIP(hipMalloc(&M.dK, sizeof(unsigned short) * ne));
    NG_HIP(hipMalloc(&M.dVal, sizeof(double) * ne));
    for (int col = 0; col < nc; ++col) {
      const int o = h->mcOff[col], cnt = h->mcOff[col + 1] - o, w = M.width[(size_t)col];
      if (!cnt || !w) continue;
      if (dir == 0)
        k_mc_fill<true><<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, dColor, h->dMcRows + o, cnt, w, M.dCol + M.entOff[(size_t)col],
                                                                  M.dK + M.entOff[(size_t)col]);
      else
        k_mc_fill<false><<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, dColor, h->dMcRows + o, cnt, w, M.dCol + M.entOff[(size_t)col],
                                                                   M.dK + M.entOff[(size_t)col]);
    }
  }
  NG_HIP(hipGetLastError());
  NG_HIP(hipStreamSynchronize(h->stream));
  (void)hipFree(dColor);
  (void)hipFree(dW);
  h->mcReady = true;
  return NALU_GPU_OK;
}

// Ifpack2 compute() for mt_sgs: the current matrix values in colour-major order
int nalu_gpu::mc_sgs_compute(nalu_gpu_linsys *h)
{
  NG_TRY(ensure_mc_colors(h));
  const int nc = (int)h->mcOff.size() - 1;
  for (const nalu_gpu_linsys::McDir &M : h->mc)
    for (int c = 0; c < nc; ++c) {
      const int o = h->mcOff[c], cnt = h->mcOff[c + 1] - o, w = M.width[(size_t)c];
      if (cnt && w)
        k_mc_gather_vals<<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, w, h->dMcRows + o, M.dCol + M.entOff[(size_t)c],
                                                                    M.dK + M.entOff[(size_t)c], h->dSliceOff, h->dVals,
                                                                    M.dVal + M.entOff[(size_t)c]);
    }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// z = multicolour SGS(r) on the owned rows (mc_sgs_compute must have run for the current matrix); uses h->dT as
// the intermediate vector.  Every row is written by its colour's launch before a later colour reads it: no zero fill.
int nalu_gpu::apply_mc_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone)
{
  NG_TRY(ensure_mc_colors(h));
  const SolverScalars *sc = useDone ? h->dScal : nullptr;
  const int nc = (int)h->mcOff.size() - 1;
  for (int c = 0; c < nc; ++c) {
    const nalu_gpu_linsys::McDir &M = h->mc[0];
    const int o = h->mcOff[c], cnt = h->mcOff[c + 1] - o;
    if (cnt)
      k_mc_sweep<true><<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, M.width[(size_t)c], M.dVal + M.entOff[(size_t)c],
                                                                 M.dCol + M.entOff[(size_t)c], h->dMcRows + o, sc, h->dDinv,
                                                                 r, nullptr, h->dT);
  }
  for (int c = nc - 1; c >= 0; --c) {
    const nalu_gpu_linsys::McDir &M = h->mc[1];
    const int o = h->mcOff[c], cnt = h->mcOff[c + 1] - o;
    if (cnt)
      k_mc_sweep<false><<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, M.width[(size_t)c], M.dVal + M.entOff[(size_t)c],
                                                                  M.dCol + M.entOff[(size_t)c], h->dMcRows + o, sc, h->dDinv,
                                                                  nullptr, h->dT, z);
  }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_sgs_schedule(nalu_gpu_linsys *h, int out[8])
{
  NG_REQUIRE(h && out, NALU_GPU_EINVAL, "null argument");
  const bool flow = h->flowState == 1 && sgs_mode() == SGS_FLOW;
  out[0] = flow ? 1 : h->flowGaveUp ? -2 : (h->flowState != 0 || h->sgsReady) ? -1 : 0;
  out[1] = flow && h->flowTiled;
  out[2] = flow ? h->flow[0].nChunks : 0;
  out[3] = flow ? std::max(h->flow[0].maxSteps, h->flow[1].maxSteps) : 0;
  out[4] = flow ? std::max(h->flow[0].wMax, h->flow[1].wMax) : 0;
  out[5] = flow ? std::max(h->flow[0].ring, h->flow[1].ring) : 0;
  out[6] = flow ? h->flowTile[2] : 0;
  out[7] = flow ? h->flowTile[3] : 0;
  return NALU_GPU_OK;
}
