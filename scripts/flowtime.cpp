// flowtime.cpp -- host time of the dataflow SGS schedule build (nalu_amd/csrc/sgs_flow.hpp) on an n^3 7-point lattice,
// phase by phase.  Development aid: g++ -O2 -std=c++17 -pthread -I../nalu_amd/csrc flowtime.cpp -o flowtime; ./flowtime 256 16
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "sgs_flow.hpp"

using namespace nalu_gpu;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
  const int nx = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 16;
  const int n = nx * nx * nx;
  std::vector<int64_t> rowPtr(1, 0);
  std::vector<int> col;
  rowPtr.reserve((size_t)n + 1);
  col.reserve((size_t)n * 7);
  for (int z = 0; z < nx; ++z)
    for (int y = 0; y < nx; ++y)
      for (int x = 0; x < nx; ++x) {
        const int i = x + nx * (y + nx * z);
        if (z > 0) col.push_back(i - nx * nx);
        if (y > 0) col.push_back(i - nx);
        if (x > 0) col.push_back(i - 1);
        col.push_back(i);
        if (x + 1 < nx) col.push_back(i + 1);
        if (y + 1 < nx) col.push_back(i + nx);
        if (z + 1 < nx) col.push_back(i + nx * nx);
        rowPtr.push_back((int64_t)col.size());
      }
  double t0 = now();
  FlowBuf<int> chunkOf;
  FlowTiling T = flow_chunking(n, rowPtr.data(), col.data(), 1024, 6, 6, chunkOf, threads);
  double t1 = now();
  FlowSchedule S[2];
  std::string why[2];
  auto build = [&](int dir) { why[dir] = flow_build(n, rowPtr.data(), col.data(), chunkOf, T.nChunks, dir == 0, S[dir], threads / 2); };
  std::thread other(build, 1);
  build(0);
  other.join();
  double t2 = now();
  printf("n %d threads %d: chunking %.3f s, both schedules %.3f s (chunks %zu, steps %d) %s%s\n", n, threads, t1 - t0, t2 - t1,
         S[0].chunks.size(), S[0].maxSteps, why[0].c_str(), why[1].c_str());
  return 0;
}
