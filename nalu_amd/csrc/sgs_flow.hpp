// sgs_flow.hpp -- schedule of the dataflow Gauss-Seidel sweeps (precond.hip), host side.
//
// The exact sequential sweep of Ifpack2's SGS (src/LinearSolverConfig.C:71-75) is a chain of
// 3n - 2 dependent levels on an n^3 lattice.  One launch per level costs a kernel boundary per
// level; here ONE launch per sweep runs every chunk of rows in its own workgroup and lets the
// chunks wait for each other through memory (data-as-flag, see precond.hip):
//   * a chunk is a set of rows; the chunks form a list in which every row only depends on rows
//     of its own chunk or of EARLIER chunks (checked here, so workgroups dispatched in list
//     order can never wait for one that is not running);
//   * inside a chunk the rows are sorted by their local level (counting only the dependencies
//     inside the chunk) and cut into steps of at most 64 independent rows: one wavefront walks
//     the steps, in-chunk values travel through an LDS ring, values of other chunks are polled;
//   * per compact row p: rowOf[p] (the matrix row) and w entries in ascending column order (the
//     order of the sequential sweep's sum): code > 0 = the value is p - code rows back in this
//     chunk, code < 0 = the value of a row of another chunk, FLOW_PAD = no entry;
//   * the rows other chunks read ("visible") are numbered in schedule order: pubOf[p] is the slot of
//     compact row p in the exchange buffer the chunks publish to and poll (-1: not visible), and
//     ~code of an entry of another chunk is that row's slot.  The rows a step publishes, and the
//     rows a step of a neighbouring tile needs, are then a few consecutive slots: a handful of
//     memory requests per step instead of one per row.
// Plain C++ (no HIP): scripts/flowcheck.cpp replays the schedule on the CPU against the
// sequential sweep.
#pragma once
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <type_traits>
#include <sys/mman.h>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace nalu_gpu {

constexpr int FLOW_PAD = INT_MIN;
constexpr int FLOW_STEP = 64;  // rows per step: one wavefront

struct FlowChunk {   // 32 bytes, read by the device
  long long entOff;  // entry k of compact row q of the chunk: entOff + k * m + q
  int rowOff;        // first compact row of the chunk (index into rowOf)
  int m;             // rows
  int w;             // entries per row (padded with FLOW_PAD)
  int stepOff;       // index of the chunk's step offsets in steps[]: nSteps + 1 chunk-relative row offsets
  int nSteps;
  int pad;
};

// an array whose elements start out uninitialised: the big arrays of the schedule are written exactly once, by the
// threads that build them (a std::vector would first fill -- and page in -- hundreds of MB on one thread)
// (large ones ask for transparent huge pages: the first touch of 4 KiB pages by 16 threads at once is what the build
// would otherwise spend most of its time in)
template <class T>
struct FlowBuf {
  struct Free { void operator()(T *q) const { std::free(q); } };
  std::unique_ptr<T[], Free> p;
  size_t n = 0;
  void alloc(size_t m)
  {
    static_assert(std::is_trivial<T>::value, "raw storage");
    const size_t bytes = std::max<size_t>(m, 1) * sizeof(T), huge = (size_t)2 << 20;
    void *q = nullptr;
    if (bytes >= 2 * huge) {
      if (posix_memalign(&q, huge, (bytes + huge - 1) / huge * huge) != 0) q = nullptr;
#ifdef MADV_HUGEPAGE
      if (q) (void)madvise(q, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
#endif
    }
    if (!q) q = std::malloc(bytes);
    if (!q) throw std::bad_alloc();
    p.reset((T *)q);
    n = m;
  }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

struct FlowSchedule {
  std::vector<FlowChunk> chunks;  // in processing (= dispatch) order
  FlowBuf<int> rowOf;             // n: matrix row of every compact row
  FlowBuf<int> pubOf;             // n: slot of the row in the exchange buffer, -1 if no other chunk reads it
  int nVisible = 0;               // slots
  FlowBuf<int> code;              // entries
  // Entry k of compact row p takes its value from the (kFirst[p] + k)-th entry of the matrix row (columns
  // ascending: the lower entries are a row's first ones, kFirst = 0 and the array stays empty; the upper ones
  // follow the diagonal)
  FlowBuf<unsigned char> kFirst;
  std::vector<int> steps;
  int wMax = 0;                   // widest chunk
  int maxLookback = 0;            // largest in-chunk distance p - p(dep)
  int maxSteps = 0;
  long long nExternal = 0;
};

// runs f(0) .. f(count - 1) on up to `threads` host threads (dynamic distribution)
template <class F>
inline void flow_parallel_for(int count, int threads, F f)
{
  threads = std::max(1, std::min(threads, count));
  if (threads == 1) {
    for (int i = 0; i < count; ++i) f(i);
    return;
  }
  std::atomic<int> next(0);
  std::atomic<bool> failed(false);
  std::exception_ptr error;  // the first exception of any thread, rethrown by the caller's thread
  auto work = [&]() {
    try {
      for (int i = next.fetch_add(1); i < count && !failed.load(); i = next.fetch_add(1)) f(i);
    } catch (...) {
      bool expected = false;
      if (failed.compare_exchange_strong(expected, true)) error = std::current_exception();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (std::thread &t : pool) t.join();
  if (error) std::rethrow_exception(error);
}

// chunkOf[i] = position of row i's chunk in the FORWARD processing order (0 .. nChunks-1); the
// backward sweep processes the same chunks in reverse.  Returns an empty string, or the reason
// this chunking cannot be used (a dependency on a later chunk).  The chunks are independent of
// each other until the exchange slots are numbered: two parallel passes over the chunks (levels
// and order; entries) around a prefix sum of their sizes.
inline std::string flow_build(int n, const int64_t *rowPtr, const int *colIdx, const FlowBuf<int> &chunkOf,
                              int nChunks, bool fwd, FlowSchedule &S, int threads = 1)
{
  S = FlowSchedule();
  threads = std::max(1, threads);
  // rows of every chunk, ascending: a counting sort, the rows dealt to the threads in contiguous blocks
  std::vector<int> cnt((size_t)nChunks + 1, 0);
  FlowBuf<int> rows;
  rows.alloc((size_t)std::max(n, 1));
  {
    const int nb = (int)std::min<long long>(threads, std::max(1, n / 65536));
    auto blk = [&](int b) { return (int)((long long)n * b / nb); };
    std::vector<std::vector<int>> local((size_t)nb, std::vector<int>((size_t)nChunks, 0));
    flow_parallel_for(nb, threads, [&](int b) {
      std::vector<int> &c = local[(size_t)b];
      for (int i = blk(b); i < blk(b + 1); ++i) ++c[(size_t)chunkOf[i]];
    });
    for (int c = 0, run = 0; c < nChunks; ++c) {
      for (int b = 0; b < nb; ++b) {
        const int k = local[(size_t)b][(size_t)c];
        local[(size_t)b][(size_t)c] = run;
        run += k;
      }
      cnt[(size_t)c + 1] = run;
    }
    flow_parallel_for(nb, threads, [&](int b) {
      std::vector<int> &c = local[(size_t)b];
      for (int i = blk(b); i < blk(b + 1); ++i) rows[(size_t)c[(size_t)chunkOf[i]]++] = i;
    });
  }
  auto is_dep = [&](int i, int c) { return fwd ? c < i : (c > i && c < n); };
  FlowBuf<int> lvl, pos;  // (written per chunk before they are read)
  lvl.alloc((size_t)std::max(n, 1));
  pos.alloc((size_t)std::max(n, 1));
  S.rowOf.alloc((size_t)std::max(n, 1));
  S.pubOf.alloc((size_t)std::max(n, 1));
  if (!fwd) S.kFirst.alloc((size_t)std::max(n, 1));
  // the non-empty chunks in processing order, and where their rows start
  std::vector<int> order;
  std::vector<FlowChunk> &chunks = S.chunks;
  for (int oc = 0, rowOff = 0; oc < nChunks; ++oc) {
    const int c = fwd ? oc : nChunks - 1 - oc;
    const int m = cnt[c + 1] - cnt[c];
    if (m == 0) continue;
    FlowChunk C{};
    C.rowOff = rowOff;
    C.m = m;
    rowOff += m;
    order.push_back(c);
    chunks.push_back(C);
  }
  const int nc = (int)chunks.size();
  std::vector<std::vector<int>> stepsOf((size_t)nc);
  std::vector<int> badRow((size_t)nc, -1), badCol((size_t)nc, -1);
  // pass 1: local levels (only the dependencies inside the chunk count), order of the rows, steps
  flow_parallel_for(nc, threads, [&](int ic) {
    const int c = order[ic];
    FlowChunk &C = chunks[ic];
    const int b = cnt[c], e = cnt[c + 1], m = e - b;
    int nl = 0, w = 0;
    for (int t = 0; t < m; ++t) {
      const int i = fwd ? rows[(size_t)(b + t)] : rows[(size_t)(e - 1 - t)];
      int l = 0, deps = 0;
      for (int64_t k = rowPtr[i]; k < rowPtr[i + 1]; ++k) {
        const int col = colIdx[k];
        if (!is_dep(i, col)) continue;
        ++deps;
        if (chunkOf[col] == c) l = std::max(l, lvl[(size_t)col] + 1);
        else if ((fwd ? chunkOf[col] > c : chunkOf[col] < c) && badRow[ic] < 0) { badRow[ic] = i; badCol[ic] = col; }
      }
      lvl[(size_t)i] = l;
      nl = std::max(nl, l + 1);
      w = std::max(w, deps);
    }
    // counting sort by level, ascending row inside a level
    std::vector<int> lcount((size_t)nl + 1, 0);
    for (int t = b; t < e; ++t) ++lcount[(size_t)lvl[(size_t)rows[(size_t)t]] + 1];
    for (int l = 0; l < nl; ++l) lcount[l + 1] += lcount[l];
    C.w = w;
    std::vector<int> &st = stepsOf[ic];
    for (int l = 0; l < nl; ++l)
      for (int q = lcount[l]; q < lcount[l + 1]; q += FLOW_STEP) st.push_back(q);
    st.push_back(m);
    C.nSteps = (int)st.size() - 1;
    for (int t = b; t < e; ++t) {
      const int i = rows[(size_t)t];
      const int q = lcount[lvl[(size_t)i]]++;
      pos[(size_t)i] = q;
      S.rowOf[(size_t)C.rowOff + q] = i;
    }
  });
  for (int ic = 0; ic < nc; ++ic)
    if (badRow[ic] >= 0)
      return "row " + std::to_string(badRow[ic]) + " depends on row " + std::to_string(badCol[ic]) + " of a later chunk";
  long long entTotal = 0;
  for (int ic = 0; ic < nc; ++ic) {
    FlowChunk &C = chunks[ic];
    C.entOff = entTotal;
    entTotal += (long long)C.w * C.m;
    C.stepOff = (int)S.steps.size();
    S.steps.insert(S.steps.end(), stepsOf[ic].begin(), stepsOf[ic].end());
    S.maxSteps = std::max(S.maxSteps, C.nSteps);
    S.wMax = std::max(S.wMax, C.w);
  }
  S.code.alloc((size_t)std::max<long long>(entTotal, 1));
  S.code[0] = FLOW_PAD;
  // pass 2: the entries (every one of a chunk's w x m, padding included); the rows other chunks read are marked
  std::vector<unsigned char> visible((size_t)std::max(n, 1), 0);
  std::vector<int> lookOf((size_t)nc, 0), scattered((size_t)nc, -1);
  std::vector<long long> extOf((size_t)nc, 0);
  flow_parallel_for(nc, threads, [&](int ic) {
    const int c = order[ic];
    const FlowChunk &C = chunks[ic];
    const int b = cnt[c], e = cnt[c + 1], m = C.m;
    int look = 0, scat = -1;  // (locals: the per-chunk slots of neighbouring chunks share cache lines)
    long long ext = 0;
    for (int t = b; t < e; ++t) {
      const int i = rows[(size_t)t], q = pos[(size_t)i];
      int kk = 0, first = -1;
      for (int64_t k = rowPtr[i]; k < rowPtr[i + 1]; ++k) {
        const int col = colIdx[k];
        if (!is_dep(i, col)) continue;
        if (first < 0) first = (int)(k - rowPtr[i]);
        if ((int)(k - rowPtr[i]) != first + kk || first > 255) scat = i;  // (kFirst + k would be wrong)
        const size_t idx = (size_t)(C.entOff + (long long)kk * m + q);
        if (chunkOf[col] == c) {
          const int back = q - pos[(size_t)col];  // > 0: a lower level sorts first
          S.code[idx] = back;
          look = std::max(look, back);
        } else {
          S.code[idx] = ~col;
          __atomic_store_n(&visible[(size_t)col], (unsigned char)1, __ATOMIC_RELAXED);  // (several chunks may mark a row)
          ++ext;
        }
        ++kk;
      }
      for (; kk < C.w; ++kk) S.code[(size_t)(C.entOff + (long long)kk * m + q)] = FLOW_PAD;
      if (!fwd) S.kFirst[(size_t)C.rowOff + q] = (unsigned char)std::min(std::max(first, 0), 255);
    }
    lookOf[ic] = look;
    extOf[ic] = ext;
    scattered[ic] = scat;
  });
  for (int ic = 0; ic < nc; ++ic) {
    if (scattered[ic] >= 0)
      return "the entries row " + std::to_string(scattered[ic]) + " depends on are not consecutive (columns not ascending?)";
    S.maxLookback = std::max(S.maxLookback, lookOf[ic]);
    S.nExternal += extOf[ic];
  }
  // exchange slots in schedule order, then the entries of other chunks point at them
  std::vector<int> visOf((size_t)nc + 1, 0);
  flow_parallel_for(nc, threads, [&](int ic) {
    const FlowChunk &C = chunks[ic];
    int k = 0;
    for (int p = C.rowOff; p < C.rowOff + C.m; ++p) k += visible[(size_t)S.rowOf[(size_t)p]];
    visOf[(size_t)ic + 1] = k;
  });
  for (int ic = 0; ic < nc; ++ic) visOf[(size_t)ic + 1] += visOf[(size_t)ic];
  S.nVisible = visOf[(size_t)nc];
  FlowBuf<int> slotOfRow;  // (only the visible rows' entries are written and read)
  slotOfRow.alloc((size_t)std::max(n, 1));
  flow_parallel_for(nc, threads, [&](int ic) {
    const FlowChunk &C = chunks[ic];
    int next = visOf[(size_t)ic];
    for (int p = C.rowOff; p < C.rowOff + C.m; ++p) {
      const int i = S.rowOf[(size_t)p];
      S.pubOf[(size_t)p] = visible[(size_t)i] ? (slotOfRow[(size_t)i] = next++) : -1;
    }
  });
  flow_parallel_for(nc, threads, [&](int ic) {
    const FlowChunk &C = chunks[ic];
    for (long long x = C.entOff, xe = C.entOff + (long long)C.w * C.m; x < xe; ++x) {
      int &c = S.code[(size_t)x];
      if (c < 0 && c != FLOW_PAD) c = ~slotOfRow[(size_t)~c];
    }
  });
  return std::string();
}

// Chunking.  Contiguous runs of `chunkRows` rows are always valid (a row only depends on lower
// rows in the forward sweep, on higher rows in the backward one).  On a lattice-numbered mesh
// (row = x + Q y + P z: the three dominant lower offsets are 1 < Q < P with Q | P) tiles of
// By lines x Bz planes put most y and z dependencies inside the chunk and leave
// ny/By + nz/Bz chunk-to-chunk hops on the critical path instead of ny + nz.  (Some tiles may be
// empty when n is not a multiple of P: flow_build skips them.)
struct FlowTiling {
  bool tiled = false;
  int Q = 0, P = 0, By = 0, Bz = 0;
  int nChunks = 0;
};

inline FlowTiling flow_chunking(int n, const int64_t *rowPtr, const int *colIdx, int chunkRows, int By, int Bz,
                                FlowBuf<int> &chunkOf, int threads = 1)
{
  FlowTiling T;
  chunkOf.alloc((size_t)std::max(n, 1));
  threads = std::max(1, threads);
  const int nb = (int)std::min<long long>(threads, std::max(1, n / 65536));
  auto blk = [&](int b) { return (int)((long long)n * b / nb); };
  // dominant lower offsets
  int Q = 0, P = 0;
  if (n > 0 && By > 0 && Bz > 0) {
    // the most frequent offset i - c over first (farthest) lower entries = P, over rows' second = Q
    typedef std::vector<std::pair<int, long long>> Hist;  // small maps: offset -> count
    auto bump = [](Hist &h, int off, long long by) {
      for (auto &p : h)
        if (p.first == off) { p.second += by; return; }
      if (h.size() < 64) h.push_back({off, by});
    };
    std::vector<Hist> h1b((size_t)nb), h2b((size_t)nb);
    std::vector<long long> rows3b((size_t)nb, 0);
    flow_parallel_for(nb, threads, [&](int b) {
      for (int i = blk(b); i < blk(b + 1); ++i) {
        int lower[3], nlow = 0;
        for (int64_t k = rowPtr[i]; k < rowPtr[i + 1] && colIdx[k] < i; ++k) {
          if (nlow < 3) lower[nlow] = colIdx[k];
          ++nlow;
        }
        if (nlow != 3 || lower[2] != i - 1) continue;
        ++rows3b[(size_t)b];
        bump(h1b[(size_t)b], i - lower[0], 1);
        bump(h2b[(size_t)b], i - lower[1], 1);
      }
    });
    Hist h1, h2;
    long long rows3 = 0;
    for (int b = 0; b < nb; ++b) {
      rows3 += rows3b[(size_t)b];
      for (auto &p : h1b[(size_t)b]) bump(h1, p.first, p.second);
      for (auto &p : h2b[(size_t)b]) bump(h2, p.first, p.second);
    }
    auto top = [](const Hist &h, long long &count) {
      int best = 0;
      count = 0;
      for (auto &p : h)
        if (p.second > count) { count = p.second; best = p.first; }
      return best;
    };
    long long c1 = 0, c2 = 0;
    const int p1 = top(h1, c1), q1 = top(h2, c2);
    if (rows3 * 2 >= n && c1 * 10 >= rows3 * 9 && c2 * 10 >= rows3 * 9 && q1 > 1 && p1 > q1 && p1 % q1 == 0) {
      P = p1;
      Q = q1;
    }
  }
  if (P > 0) {
    const long long blockRows = (long long)By * Q;
    const int nYb = (int)((P + blockRows - 1) / blockRows);
    // tiles in anti-diagonal order (yb + zb, then zb): a tile's producers (yb - 1, zb) and
    // (yb, zb - 1) come earlier, and the workgroups resident at any time are the ones whose turn
    // is next (a z-major order would fill the chip with tiles that wait for most of the sweep)
    const int nZb = (n - 1) / P / Bz + 1;
    std::vector<int> order((size_t)nZb * nYb);
    {
      int next = 0;
      for (int d = 0; d <= nZb + nYb - 2; ++d)
        for (int zb = std::max(0, d - (nYb - 1)); zb <= std::min(d, nZb - 1); ++zb) order[(size_t)zb * nYb + (d - zb)] = next++;
    }
    flow_parallel_for(nb, threads, [&](int b) {
      for (int i = blk(b); i < blk(b + 1); ++i) {
        const int zb = i / P / Bz, within = i % P;
        chunkOf[(size_t)i] = order[(size_t)zb * nYb + (int)(within / blockRows)];
      }
    });
    T.tiled = true;
    T.Q = Q; T.P = P; T.By = By; T.Bz = Bz;
    T.nChunks = nZb * nYb;
  } else {
    flow_parallel_for(nb, threads, [&](int b) {
      for (int i = blk(b); i < blk(b + 1); ++i) chunkOf[(size_t)i] = i / chunkRows;
    });
    T.nChunks = n ? (n - 1) / chunkRows + 1 : 0;
  }
  return T;
}

}  // namespace nalu_gpu
