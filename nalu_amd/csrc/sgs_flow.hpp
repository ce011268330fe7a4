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
#include <exception>
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

struct FlowSchedule {
  std::vector<FlowChunk> chunks;  // in processing (= dispatch) order
  std::vector<int> rowOf;         // n: matrix row of every compact row
  std::vector<int> pubOf;         // n: slot of the row in the exchange buffer, -1 if no other chunk reads it
  int nVisible = 0;               // slots
  std::vector<int> code;          // entries
  std::vector<long long> src;     // entries: CSR entry index the value comes from (-1: pad)
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
inline std::string flow_build(int n, const int64_t *rowPtr, const int *colIdx, const std::vector<int> &chunkOf,
                              int nChunks, bool fwd, FlowSchedule &S, int threads = 1)
{
  S = FlowSchedule();
  // rows of every chunk, ascending
  std::vector<int> cnt((size_t)nChunks + 1, 0);
  for (int i = 0; i < n; ++i) ++cnt[(size_t)chunkOf[i] + 1];
  for (int c = 0; c < nChunks; ++c) cnt[c + 1] += cnt[c];
  std::vector<int> rows((size_t)n);
  {
    std::vector<int> cur(cnt.begin(), cnt.end() - 1);
    for (int i = 0; i < n; ++i) rows[cur[chunkOf[i]]++] = i;
  }
  auto is_dep = [&](int i, int c) { return fwd ? c < i : (c > i && c < n); };
  std::vector<int> lvl((size_t)n, 0), pos((size_t)n, 0);
  S.rowOf.resize((size_t)n);
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
      const int i = fwd ? rows[b + t] : rows[e - 1 - t];
      int l = 0, deps = 0;
      for (int64_t k = rowPtr[i]; k < rowPtr[i + 1]; ++k) {
        const int col = colIdx[k];
        if (!is_dep(i, col)) continue;
        ++deps;
        if (chunkOf[col] == c) l = std::max(l, lvl[col] + 1);
        else if ((fwd ? chunkOf[col] > c : chunkOf[col] < c) && badRow[ic] < 0) { badRow[ic] = i; badCol[ic] = col; }
      }
      lvl[i] = l;
      nl = std::max(nl, l + 1);
      w = std::max(w, deps);
    }
    // counting sort by level, ascending row inside a level
    std::vector<int> lcount((size_t)nl + 1, 0);
    for (int t = b; t < e; ++t) ++lcount[(size_t)lvl[rows[t]] + 1];
    for (int l = 0; l < nl; ++l) lcount[l + 1] += lcount[l];
    C.w = w;
    std::vector<int> &st = stepsOf[ic];
    for (int l = 0; l < nl; ++l)
      for (int q = lcount[l]; q < lcount[l + 1]; q += FLOW_STEP) st.push_back(q);
    st.push_back(m);
    C.nSteps = (int)st.size() - 1;
    for (int t = b; t < e; ++t) {
      const int i = rows[t];
      const int q = lcount[lvl[i]]++;
      pos[i] = q;
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
  S.code.assign((size_t)std::max<long long>(entTotal, 1), FLOW_PAD);
  S.src.assign((size_t)std::max<long long>(entTotal, 1), -1);
  // pass 2: the entries
  std::vector<int> lookOf((size_t)nc, 0);
  std::vector<long long> extOf((size_t)nc, 0);
  flow_parallel_for(nc, threads, [&](int ic) {
    const int c = order[ic];
    const FlowChunk &C = chunks[ic];
    const int b = cnt[c], e = cnt[c + 1], m = C.m;
    for (int t = b; t < e; ++t) {
      const int i = rows[t], q = pos[i];
      int kk = 0;
      for (int64_t k = rowPtr[i]; k < rowPtr[i + 1]; ++k) {
        const int col = colIdx[k];
        if (!is_dep(i, col)) continue;
        const size_t idx = (size_t)(C.entOff + (long long)kk * m + q);
        S.src[idx] = k;
        if (chunkOf[col] == c) {
          const int back = q - pos[col];  // > 0: a lower level sorts first
          S.code[idx] = back;
          lookOf[ic] = std::max(lookOf[ic], back);
        } else {
          S.code[idx] = ~col;
          ++extOf[ic];
        }
        ++kk;
      }
    }
  });
  for (int ic = 0; ic < nc; ++ic) {
    S.maxLookback = std::max(S.maxLookback, lookOf[ic]);
    S.nExternal += extOf[ic];
  }
  // exchange slots in schedule order, then the entries of other chunks point at them
  std::vector<char> visible((size_t)n, 0);
  for (int c : S.code)
    if (c < 0 && c != FLOW_PAD) visible[(size_t)~c] = 1;
  std::vector<int> slotOfRow((size_t)n, -1);
  S.pubOf.assign((size_t)n, -1);
  for (int p = 0; p < n; ++p)
    if (visible[S.rowOf[p]]) slotOfRow[S.rowOf[p]] = S.pubOf[p] = S.nVisible++;
  flow_parallel_for(nc, threads, [&](int ic) {
    const FlowChunk &C = chunks[ic];
    for (long long x = C.entOff, xe = C.entOff + (long long)C.w * C.m; x < xe; ++x) {
      int &c = S.code[(size_t)x];
      if (c < 0 && c != FLOW_PAD) c = ~slotOfRow[~c];
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
                                std::vector<int> &chunkOf)
{
  FlowTiling T;
  chunkOf.resize((size_t)n);
  // dominant lower offsets
  int Q = 0, P = 0;
  if (n > 0 && By > 0 && Bz > 0) {
    // the most frequent offset i - c over first (farthest) lower entries = P, over rows' second = Q
    std::vector<std::pair<int, int>> h1, h2;  // small maps: offset -> count
    auto bump = [](std::vector<std::pair<int, int>> &h, int off) {
      for (auto &p : h)
        if (p.first == off) { ++p.second; return; }
      if (h.size() < 64) h.push_back({off, 1});
    };
    long long rows3 = 0;
    for (int i = 0; i < n; ++i) {
      int lower[3], nlow = 0;
      for (int64_t k = rowPtr[i]; k < rowPtr[i + 1] && colIdx[k] < i; ++k) {
        if (nlow < 3) lower[nlow] = colIdx[k];
        ++nlow;
      }
      if (nlow != 3 || lower[2] != i - 1) continue;
      ++rows3;
      bump(h1, i - lower[0]);
      bump(h2, i - lower[1]);
    }
    auto top = [](const std::vector<std::pair<int, int>> &h, long long &count) {
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
    for (int i = 0; i < n; ++i) {
      const int zb = i / P / Bz, within = i % P;
      chunkOf[i] = order[(size_t)zb * nYb + (int)(within / blockRows)];
    }
    T.tiled = true;
    T.Q = Q; T.P = P; T.By = By; T.Bz = Bz;
    T.nChunks = nZb * nYb;
  } else {
    for (int i = 0; i < n; ++i) chunkOf[i] = i / chunkRows;
    T.nChunks = n ? (n - 1) / chunkRows + 1 : 0;
  }
  return T;
}

}  // namespace nalu_gpu
