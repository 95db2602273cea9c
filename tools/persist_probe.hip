// persist_probe.hip -- what would a per-diagonal hand-off between the workgroups of one sequence cost?
//
// Model of "persistent workgroups per cell block" (DESIGN section 9): NB blocks per sequence, every block owned by one
// workgroup for all NSTEP diagonals; before step d a block needs step d-1 of its K right-hand neighbours.  Work items
// (sequence, block) are taken from ONE ticket counter in dependency order (right-most block first), so a waiting
// workgroup only ever waits for workgroups that already run: no deadlock for any grid size; every spin is bounded.
// Hand-off = the placement-independent recipe of the CDNA guide (Guideline 16): payload stores, every wave drains
// (s_waitcnt vmcnt(0)), barrier, one lane agent-scope release + flag store; the consumer polls relaxed (agent scope), one
// acquire fence, barrier, plain loads.  Payload per step and block: ROW doubles written, K * ROW read back and checked.
//
// Prints: errors (stale payload), timeouts, total time, and the time per step of a workgroup -- to be compared with the
// same loop WITHOUT hand-offs (mode 0: same loads and stores, no waiting, payload unchecked), i.e. the price of the
// protocol per workgroup-diagonal.
//
//   hipcc --offload-arch=gfx950 -O3 -o build/persist_probe tools/persist_probe.hip && build/persist_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kThreads = 256;
constexpr int ROW = 2048;          // doubles per (block, step): ~ 11 cells x 23 states x 7 planes
constexpr int kSpinMax = 1 << 22;

struct Args {
  int n_seq, nb, nstep, K, mode;   // mode 0: no hand-off, 1: agent release/acquire, 2: + busy work between
  int* ticket;                     // [1]
  int* prog;                       // [n_seq * nb] steps completed
  int* err;                        // [0] stale payloads, [1] timeouts
  double* data;                    // [n_seq][nb][nstep][ROW]
  int work;                        // dependent FMA chain length per step (stands for the phases of a diagonal)
};

__device__ __forceinline__ double expected(int g, int b, int d, int t) { return (double)(((g * 31 + b) * 131 + d) * 257 + t); }

__global__ __launch_bounds__(kThreads) void probe(Args a) {
  __shared__ int s_item;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  for (;;) {
    if (tid == 0) s_item = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int item = s_item;
    __syncthreads();
    if (item >= a.n_seq * a.nb) return;
    const int g = item / a.nb, b = a.nb - 1 - item % a.nb;     // right-most block of a sequence first
    double* mine = a.data + ((size_t)g * a.nb + b) * a.nstep * ROW;
    for (int d = 0; d < a.nstep; ++d) {
      double acc = 0.;
      if (d > 0) {
        // wait for step d-1 of the K neighbours to the right (mode 0: the same reads without waiting -- payload unchecked)
        const int nn = (a.nb - 1 - b < a.K) ? a.nb - 1 - b : a.K;
        if (a.mode && tid < 64) {
          bool ok = true;
          int spins = 0;
          do {
            ok = true;
            if (tid < nn) ok = __hip_atomic_load(&a.prog[g * a.nb + b + 1 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= d;
            if (++spins > kSpinMax) break;
            if (__hip_atomic_load(&a.err[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          } while (!__all(ok));
          if (tid == 0) {
            s_ok = __all(ok) ? 1 : 0;
            if (!s_ok) atomicAdd(&a.err[1], 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        __syncthreads();
        if (a.mode && !s_ok) return;                           // (bounded: give up, the host sees err[1])
        for (int n = 0; n < nn; ++n) {
          const double* nb_row = a.data + (((size_t)g * a.nb + b + 1 + n) * a.nstep + (d - 1)) * ROW;
          for (int t = tid; t < ROW; t += kThreads) {
            const double v = nb_row[t];
            if (a.mode && v != expected(g, b + 1 + n, d - 1, t)) atomicAdd(&a.err[0], 1);
            acc += v;
          }
        }
      }
      // the work of a diagonal: a dependent chain (latency, like the phases of k4_in / k4_out)
      double x = acc * 1e-30 + 1.0;
      for (int k = 0; k < a.work; ++k) x = fma(x, 1.0000001, 1e-9);
      double* row = mine + (size_t)d * ROW;
      for (int t = tid; t < ROW; t += kThreads) row[t] = expected(g, b, d, t) + (x > 1e300 ? 1. : 0.);
      if (a.mode) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
        __syncthreads();
        if (tid == 0) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&a.prog[g * a.nb + b], d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  Args a;
  a.n_seq = argc > 1 ? atoi(argv[1]) : 1024;
  a.nb = 19; a.nstep = 51; a.K = 6;
  const int grid = argc > 2 ? atoi(argv[2]) : 1024;
  a.work = argc > 3 ? atoi(argv[3]) : 2000;
  CK(hipMalloc(&a.ticket, 16));
  CK(hipMalloc(&a.prog, sizeof(int) * a.n_seq * a.nb));
  CK(hipMalloc(&a.err, 16));
  CK(hipMalloc(&a.data, sizeof(double) * (size_t)a.n_seq * a.nb * a.nstep * ROW));
  CK(hipMemset(a.data, 0xff, sizeof(double) * (size_t)a.n_seq * a.nb * a.nstep * ROW));   // (NaNs: a row read before it is written shows)
  printf("persist_probe: %d sequences x %d blocks x %d steps, K = %d neighbours, %d workgroups, work %d\n", a.n_seq, a.nb, a.nstep, a.K, grid, a.work);
  for (int mode = 0; mode <= 1; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      a.mode = mode;
      CK(hipMemset(a.ticket, 0, 16));
      CK(hipMemset(a.prog, 0, sizeof(int) * a.n_seq * a.nb));
      CK(hipMemset(a.err, 0, 16));
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(probe, dim3(grid), dim3(kThreads), 0, 0, a);
      CK(hipDeviceSynchronize());
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      int err[4];
      CK(hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost));
      const double steps = (double)a.n_seq * a.nb * a.nstep;
      printf("mode %d rep %d: %.2f ms, stale %d, timeouts %d -> %.2f us per workgroup-step at %d workgroups in flight\n", mode, rep, ms, err[0],
             err[1], ms * 1e3 * grid / steps, grid);
      fflush(stdout);
    }
  }
  return 0;
}
