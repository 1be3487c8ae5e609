// Bare MFMA loops, one wave per SIMD on every CU: v_mfma_f32_32x32x16_bf16 against
// v_mfma_f32_16x16x32_bf16 on random operands — FLOP/s, cycles per instruction and the clock the chip holds.
// Build:  hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form mfmabench.hip -o mfmabench
// The flag matters: without it hipcc keeps the eight 4-register accumulators of the 16x16x32 loop in
// AGPRs and rotates them through VGPRs every iteration (about sixty v_accvgpr_read / _write / _mov per
// eight MFMAs), and the loop then measures those moves (23 cycles per instruction) instead of the matrix pipe.
// Check the loop with -save-temps: it must be eight (four) back-to-back MFMAs, two scalar
// instructions and the branch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FORM>
__global__ __launch_bounds__(256) void loop(const bf16x8* in, float* out, int iters, unsigned long long* clk) {
  bf16x8 a = in[threadIdx.x], b = in[256 + threadIdx.x];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if (FORM == 32) {
    f32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
    }
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  } else {
    f32x4 acc[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    for (int k = 0; k < 8; ++k) for (int e = 0; e < 4; ++e) s += acc[k][e];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main() {
  bf16x8* in; float* out; unsigned long long* clk;
  CHECK(hipMalloc(&in, 512 * 16)); CHECK(hipMalloc(&out, 256 * 256 * 4)); CHECK(hipMalloc(&clk, 16));
  unsigned short h[512 * 8];
  srand(1);
  for (int i = 0; i < 512 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));   // +-[0.0078, 0.0156)
  CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 40000;      // x4 (32x32x16) or x8 (16x16x32) MFMAs per wave: the same FLOPs
  for (int rep = 0; rep < 3; ++rep)
    for (int form : {32, 16}) {
      CHECK(hipEventRecord(e0));
      if (form == 32) hipLaunchKernelGGL(loop<32>, dim3(256), dim3(256), 0, 0, in, out, iters, clk);
      else hipLaunchKernelGGL(loop<16>, dim3(256), dim3(256), 0, 0, in, out, iters, clk);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2]; CHECK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      const double flops = 256.0 * 4 * iters * 4 * 32768.0;
      const double per_iter = form == 32 ? 4.0 : 8.0;     // MFMA instructions per loop iteration
      printf("%s: %.2f ms  %.0f TFLOP/s  shader clock %.2f GHz  cycles per instruction %.1f  (per 32K FLOP %.1f)\n",
             form == 32 ? "32x32x16" : "16x16x32", ms, flops / ms / 1e9, (double)c[0] / ((double)c[1] * 10.0),
             (double)c[0] / (iters * per_iter), (double)c[0] / (iters * 4.0));
    }
  return 0;
}
