// Cost of neighbour synchronisation through L2 flags between persistent workgroups (one per CU):
// each of N rounds a workgroup writes `bytes` of data, publishes its round number, and waits until
// its two neighbours and itself have published that round.  Every wait is bounded.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rounds(unsigned* flags, float4* data, int n_rounds, int vec_per_thread,
                                               int work_sleep, unsigned* timeouts) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  float4* mine = data + (size_t)wg * 256 * vec_per_thread;
  for (int it = 1; it <= n_rounds; ++it) {
    if (__hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;    // someone gave up: everyone leaves
    for (int s = 0; s < work_sleep; ++s) __builtin_amdgcn_s_sleep(8);     // stand-in for the layer's compute
    for (int v = 0; v < vec_per_thread; ++v) mine[v * 256 + threadIdx.x] = make_float4(it, wg, v, 0.f);
    __syncthreads();                                                      // all waves' stores issued ...
    if (threadIdx.x == 0)
      __hip_atomic_store(&flags[wg], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ... and released
    if (threadIdx.x < 3) {
      const int nb = (wg + (int)threadIdx.x - 1 + nwg) % nwg;
      int spins = 0;
      while (__hip_atomic_load(&flags[nb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
        if (++spins > 2000000 || __hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(timeouts, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  const int n_rounds = 2000;
  unsigned *flags, *timeouts;
  float4* data;
  CHECK(hipMalloc(&flags, nwg * sizeof(unsigned)));
  CHECK(hipMalloc(&timeouts, sizeof(unsigned)));
  CHECK(hipMalloc(&data, (size_t)nwg * 256 * 8 * sizeof(float4)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int cfg[][2] = {{0, 0}, {1, 0}, {4, 0}, {4, 12}, {0, 12}};       // {16-byte vectors per thread, sleeps}
  for (auto& c : cfg) {
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(flags, 0, nwg * sizeof(unsigned)));
      CHECK(hipMemset(timeouts, 0, sizeof(unsigned)));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(rounds, dim3(nwg), dim3(256), 0, 0, flags, data, n_rounds, c[0], c[1], timeouts);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned t; CHECK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost));
      if (rep) printf("wgs %d  %d KB stored/round/wg  sleeps %d: %.2f us per round  (timeouts %u)\n", nwg, c[0] * 4, c[1], ms * 1e3 / n_rounds, t);
    }
  }
  return 0;
}
