// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths of this library (MI355X_MICROARCH.md, HBM
// section: only 16 B-per-lane streaming reads are calibrated there).  Four kernels over a buffer far larger than the
// Infinity Cache, each reading (or writing) a KNOWN byte count:
//   rd8   : 8 B per lane, a wave reads 512 contiguous bytes          (our staged operand segments)
//   rd16  : 16 B per lane, a wave reads 1024 contiguous bytes         (the guide's calibrated case)
//   seg8  : 8 B per lane, segments of 242 doubles at a stride of 4422 doubles (one diagonal row block of the band tables)
//   wr8   : 8 B per lane stores
// prints bytes, ms and GB/s of every kernel; the counters come from `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` runs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rd8(const double* p, size_t n, double* sink) {
  double acc = 0.;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 1.2345e-300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void rd16(const double2* p, size_t n2, double* sink) {
  double acc = 0.;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 1.2345e-300) sink[0] = acc;
}
// block b reads `nseg` segments of 242 doubles: segment k of block b starts at (b * nseg + k) * stride
__global__ __launch_bounds__(256) void seg8(const double* p, int nseg, size_t stride, double* sink) {
  double acc = 0.;
  const size_t b0 = (size_t)blockIdx.x * nseg;
  if (threadIdx.x < 242)
    for (int k = 0; k < nseg; ++k) acc += p[(b0 + k) * stride + threadIdx.x];
  if (acc == 1.2345e-300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wr8(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.;
}

int main() {
  const size_t n = (size_t)1 << 29;   // 4 GiB of doubles
  double *buf, *sink;
  CK(hipMalloc(&buf, n * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 0, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0)); rd8<<<grid, 256>>>(buf, n, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rd8   bytes %zu ms %.3f GB/s %.1f\n", n * 8, ms, n * 8 / ms / 1e6);
    CK(hipEventRecord(e0)); rd16<<<grid, 256>>>(reinterpret_cast<const double2*>(buf), n / 2, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rd16  bytes %zu ms %.3f GB/s %.1f\n", n * 8, ms, n * 8 / ms / 1e6);
    const size_t stride = 4422;
    const int nseg = 16;
    const size_t nblk = n / stride / nseg - 1;
    CK(hipEventRecord(e0)); seg8<<<(unsigned)nblk, 256>>>(buf, nseg, stride, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("seg8  bytes %zu ms %.3f GB/s %.1f\n", nblk * nseg * 242 * 8, ms, nblk * nseg * 242 * 8 / ms / 1e6);
    CK(hipEventRecord(e0)); wr8<<<grid, 256>>>(buf, n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("wr8   bytes %zu ms %.3f GB/s %.1f\n", n * 8, ms, n * 8 / ms / 1e6);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
