// flowcheck.cpp -- CPU replay of the dataflow SGS schedule (nalu_amd/csrc/sgs_flow.hpp) against the
// sequential sweeps, bit for bit.  Development aid: g++ -O2 -std=c++17 -pthread -I../nalu_amd/csrc flowcheck.cpp
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>

#include "sgs_flow.hpp"

using namespace nalu_gpu;

struct Csr {
  int n = 0;
  std::vector<int64_t> rowPtr;
  std::vector<int> col;
  std::vector<double> val;
};

static Csr lattice(int nx, int ny, int nz, std::mt19937 &rng)
{
  Csr A;
  A.n = nx * ny * nz;
  std::uniform_real_distribution<double> u(0.1, 1.0);
  A.rowPtr.push_back(0);
  for (int z = 0; z < nz; ++z)
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) {
        const int i = x + nx * (y + ny * z);
        auto add = [&](int c, double v) { A.col.push_back(c); A.val.push_back(v); };
        if (z > 0) add(i - nx * ny, -u(rng));
        if (y > 0) add(i - nx, -u(rng));
        if (x > 0) add(i - 1, -u(rng));
        add(i, 7.0 + u(rng));
        if (x + 1 < nx) add(i + 1, -u(rng));
        if (y + 1 < ny) add(i + nx, -u(rng));
        if (z + 1 < nz) add(i + nx * ny, -u(rng));
        A.rowPtr.push_back((int64_t)A.col.size());
      }
  return A;
}

static Csr random_graph(int n, int deg, std::mt19937 &rng)
{
  std::vector<std::set<int>> adj((size_t)n);
  std::uniform_int_distribution<int> pick(0, n - 1), near(-40, 40);
  for (int i = 0; i < n; ++i) {
    adj[i].insert(i);
    for (int d = 0; d < deg; ++d) {
      int c = (d & 1) ? pick(rng) : i + near(rng);
      if (c < 0 || c >= n) continue;
      adj[i].insert(c);
      adj[c].insert(i);
    }
  }
  Csr A;
  A.n = n;
  std::uniform_real_distribution<double> u(0.1, 1.0);
  A.rowPtr.push_back(0);
  for (int i = 0; i < n; ++i) {
    for (int c : adj[i]) { A.col.push_back(c); A.val.push_back(c == i ? 30.0 + u(rng) : -u(rng)); }
    A.rowPtr.push_back((int64_t)A.col.size());
  }
  return A;
}

static int check(const char *name, const Csr &A, int chunkRows, int By, int Bz)
{
  const int n = A.n;
  std::vector<double> r((size_t)n), dinv((size_t)n), tRef((size_t)n), zRef((size_t)n);
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int i = 0; i < n; ++i) {
    r[i] = u(rng);
    for (int64_t k = A.rowPtr[i]; k < A.rowPtr[i + 1]; ++k)
      if (A.col[k] == i) dinv[i] = 1.0 / A.val[k];
  }
  for (int i = 0; i < n; ++i) {
    double s = r[i];
    for (int64_t k = A.rowPtr[i]; k < A.rowPtr[i + 1]; ++k)
      if (A.col[k] < i) s -= A.val[k] * tRef[A.col[k]];
    tRef[i] = s * dinv[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = 0.0;
    for (int64_t k = A.rowPtr[i]; k < A.rowPtr[i + 1]; ++k)
      if (A.col[k] > i) s += A.val[k] * zRef[A.col[k]];
    zRef[i] = tRef[i] - s * dinv[i];
  }
  FlowBuf<int> chunkOf;
  FlowTiling T = flow_chunking(n, A.rowPtr.data(), A.col.data(), chunkRows, By, Bz, chunkOf, 3);
  int bad = 0;
  for (int dir = 0; dir < 2; ++dir) {
    const bool fwd = dir == 0;
    FlowSchedule S;
    std::string why = flow_build(n, A.rowPtr.data(), A.col.data(), chunkOf, T.nChunks, fwd, S, 1 + dir * 3);  // 1 and 4 threads
    if (!why.empty()) { printf("%s: %s chunking rejected: %s\n", name, fwd ? "fwd" : "bwd", why.c_str()); return 1; }
    int ring = 64;
    while (ring < S.maxLookback + FLOW_STEP) ring *= 2;
    const double NANV = std::nan("");
    std::vector<double> flow((size_t)n, NANV), all((size_t)n, NANV), ringBuf((size_t)ring);
    const std::vector<double> &in = fwd ? r : tRef;
    long long notReady = 0, rowsSeen = 0;
    for (const FlowChunk &C : S.chunks) {
      for (int s = 0; s < C.nSteps; ++s) {
        const int s0 = S.steps[C.stepOff + s], s1 = S.steps[C.stepOff + s + 1];
        if (s1 - s0 > FLOW_STEP || s1 <= s0) { printf("%s: bad step %d..%d\n", name, s0, s1); return 1; }
        std::vector<double> out((size_t)(s1 - s0));
        for (int q = s0; q < s1; ++q) {
          const int row = S.rowOf[(size_t)C.rowOff + q];
          double acc = fwd ? in[row] : 0.0;
          for (int k = 0; k < C.w; ++k) {
            const size_t idx = (size_t)(C.entOff + (long long)k * C.m + q);
            const int code = S.code[idx];
            if (code == FLOW_PAD) break;
            double v;
            if (code > 0) {
              if (code > S.maxLookback || q - code < 0) { printf("%s: bad lookback\n", name); return 1; }
              v = ringBuf[(size_t)((q - code) & (ring - 1))];
            } else {
              if (~code < 0 || ~code >= S.nVisible) { printf("%s: bad exchange slot\n", name); return 1; }
              v = flow[(size_t)~code];
              if (std::isnan(v)) ++notReady;
            }
            const double a = A.val[(size_t)(A.rowPtr[row] + (fwd ? 0 : S.kFirst[(size_t)C.rowOff + q]) + k)];
            acc = fwd ? acc - a * v : acc + a * v;
          }
          out[(size_t)(q - s0)] = fwd ? acc * dinv[row] : in[row] - acc * dinv[row];
          ++rowsSeen;
        }
        for (int q = s0; q < s1; ++q) {  // a step's rows are independent: written after all were read
          ringBuf[(size_t)(q & (ring - 1))] = out[(size_t)(q - s0)];
          // only the visible rows reach other chunks, through their slots of the exchange buffer
          if (S.pubOf[(size_t)C.rowOff + q] >= 0) flow[(size_t)S.pubOf[(size_t)C.rowOff + q]] = out[(size_t)(q - s0)];
          all[(size_t)S.rowOf[(size_t)C.rowOff + q]] = out[(size_t)(q - s0)];
        }
      }
    }
    const std::vector<double> &ref = fwd ? tRef : zRef;
    long long diff = 0;
    for (int i = 0; i < n; ++i) diff += std::memcmp(&all[i], &ref[i], 8) != 0;
    printf("%-22s %s: %s chunks %zu wMax %d lookback %d ring %d maxSteps %d external %lld  rows %lld notReady %lld "
           "mismatches %lld\n", name, fwd ? "fwd" : "bwd", T.tiled ? "tiled" : "contiguous", S.chunks.size(), S.wMax,
           S.maxLookback, ring, S.maxSteps, S.nExternal, rowsSeen, notReady, diff);
    if (T.tiled && fwd) printf("   lattice Q %d P %d tile %d x %d\n", T.Q, T.P, T.By, T.Bz);
    bad += diff != 0 || notReady != 0 || rowsSeen != n;
  }
  return bad;
}

int main()
{
  std::mt19937 rng(1);
  int bad = 0;
  bad += check("lattice 20^3 tiled", lattice(20, 20, 20, rng), 1024, 8, 8);
  bad += check("lattice 20^3 contig", lattice(20, 20, 20, rng), 1024, 0, 0);
  bad += check("lattice 256x5x4", lattice(256, 5, 4, rng), 1024, 8, 8);
  bad += check("lattice 64x48x40 4x4", lattice(64, 48, 40, rng), 512, 4, 4);
  bad += check("lattice 33x7x19 8x16", lattice(33, 7, 19, rng), 256, 8, 16);
  bad += check("random 5000", random_graph(5000, 6, rng), 1024, 8, 8);
  bad += check("random 3000 chunk 256", random_graph(3000, 12, rng), 256, 8, 8);
  bad += check("tiny 1", lattice(1, 1, 1, rng), 1024, 8, 8);
  bad += check("line 300", lattice(300, 1, 1, rng), 128, 8, 8);
  {  // a row whose columns are not ascending: the schedule locates an entry's value by its position after the
     // row's first dependency, so such a graph must be refused (the library then runs the level launches)
    Csr A = lattice(12, 6, 5, rng);
    const int i = 200;
    std::swap(A.col[(size_t)A.rowPtr[i]], A.col[(size_t)A.rowPtr[i] + 4]);  // (a lower and an upper entry trade places)
    FlowBuf<int> chunkOf;
    FlowTiling T = flow_chunking(A.n, A.rowPtr.data(), A.col.data(), 256, 0, 0, chunkOf, 2);
    int refused = 0;
    for (int dir = 0; dir < 2; ++dir) {
      FlowSchedule S;
      const std::string why = flow_build(A.n, A.rowPtr.data(), A.col.data(), chunkOf, T.nChunks, dir == 0, S, 2);
      refused += !why.empty();
      printf("unsorted row           %s: %s\n", dir == 0 ? "fwd" : "bwd", why.empty() ? "accepted" : why.c_str());
    }
    bad += refused == 0;
  }
  printf(bad ? "FAILED\n" : "all schedules replay the sequential sweeps bit for bit\n");
  return bad != 0;
}
