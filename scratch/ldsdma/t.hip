// Does an out-of-range buffer_load ... lds deposit zeros in LDS, or leave LDS untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned* x, unsigned* y, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned sm[];
  for (int i = threadIdx.x; i < 512; i += 64) sm[i] = 0xAAAAAAAAu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned off = (threadIdx.x & 1) ? threadIdx.x * 16u : 0xFFFFFFF0u;   // even lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)sm, 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) y[i] = sm[i];
}
int main() {
  unsigned *x, *y, h[256];
  hipMalloc(&x, 4096); hipMalloc(&y, 1024);
  unsigned hx[1024]; for (int i = 0; i < 1024; ++i) hx[i] = 0x1000 + i;
  hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, x, y, 1024);
  hipMemcpy(h, y, 1024, hipMemcpyDeviceToHost);
  printf("lane0 (OOB) words: %x %x %x %x | lane1 (valid): %x %x %x %x | lane2 (OOB): %x | lane3: %x\n",
         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[12]);
  return 0;
}
