// gather_bw -- what the memory system of one MI355X delivers for GATHERS of single doubles (the access pattern of the band kernels'
// stems and item operands: one double per lane from a row that is somewhere else for every cell), as opposed to streaming.
// Each lane issues K independent 8-byte loads per iteration; the lanes of a wave form groups of G neighbours that read G consecutive
// doubles at a random position of the table (G = 1: every lane its own line; G = 64: one coalesced 512-byte row segment per wave).
// Prints lanes / s, useful GB/s (8 B per lane) and line GB/s (64 B per distinct line touched) for table sizes that sit in the L2s,
// in the Infinity Cache and in HBM (and 16 KB: every load a hit of the vector L1 -- its tag rate).   build: hipcc --offload-arch=gfx950 -O3 -o build/gather_bw tools/bench_micro/gather_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int K = 8;
__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
template <int G>
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ t, unsigned long long n_mask, int iters, double* out) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long grp = gid / G, in_grp = gid % G;
  double acc = 0.;
  for (int it = 0; it < iters; ++it) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned long long r = mix(grp * 0x9e3779b97f4a7c15ULL + (unsigned long long)(it * K + k) * 0xbf58476d1ce4e5b9ULL);
      const unsigned long long idx = ((r & n_mask) & ~(unsigned long long)(G - 1)) + in_grp;     // G consecutive doubles, aligned to G
      v[k] = t[idx];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc += v[k];
  }
  if (acc == 123.456) out[gid & 1023] = acc;
}
template <int G> static float run(const double* t, unsigned long long n, int blocks, int iters, double* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_gather<G>, dim3(blocks), dim3(256), 0, 0, t, n - 1, 2, out);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k_gather<G>, dim3(blocks), dim3(256), 0, 0, t, n - 1, iters, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  const unsigned long long n_max = 1ull << 31;      // 16 GB of doubles
  double *t = nullptr, *out = nullptr;
  OK(hipMalloc(&t, n_max * 8));
  OK(hipMalloc(&out, 1024 * 8));
  OK(hipMemset(t, 0, n_max * 8));
  const int blocks = 256 * 8 * 4, iters = 64;
  const double lanes = (double)blocks * 256 * iters * K;
  printf("%-28s %6s %14s %12s %12s\n", "table", "G", "lanes/s", "useful GB/s", "line GB/s");
  for (unsigned long long n : {1ull << 11 /* 16 KB: the vector L1 */, 1ull << 18 /* 2 MB: the L2s */, 1ull << 24 /* 128 MB: the Infinity Cache */, 1ull << 31 /* 16 GB: HBM */}) {
    char name[64];
    snprintf(name, sizeof name, "%llu KB", n * 8 >> 10);
    float ms;
#define ROW(G) ms = run<G>(t, n, blocks, iters, out); \
    printf("%-28s %6d %14.3e %12.1f %12.1f\n", name, G, lanes / (ms * 1e-3), lanes * 8 / (ms * 1e-3) / 1e9, lanes / G * (G * 8 < 64 ? 64 : G * 8) / (ms * 1e-3) / 1e9);
    ROW(1) ROW(2) ROW(4) ROW(8) ROW(16) ROW(64)
  }
  return 0;
}
