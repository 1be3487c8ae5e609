"""Build scratch/libsrx_STAMP.so: conv.hip with s_memrealtime stamps at phase boundaries."""
import subprocess
s = open('super-resolution_amd/csrc/conv.hip').read()
def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old[:60])
    s = s.replace(old, new)
rep('''namespace {

struct ConvParams {''', '''__device__ unsigned long long g_stamps[1024 * 8];
extern "C" int srx_dbg_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 1024 * 8);
}
__device__ unsigned long long g_cyc[1024 * 8];
extern "C" int srx_dbg_cycles(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_cyc), sizeof(unsigned long long) * 1024 * 8);
}
#define STAMP(k) do { if (threadIdx.x == 0) { const int w_ = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 8 + (k); g_stamps[w_] = __builtin_amdgcn_s_memrealtime(); g_cyc[w_] = __builtin_amdgcn_s_memtime(); } } while (0)

namespace {

struct ConvParams {''')
rep('''  __builtin_amdgcn_s_setprio(3);
''', '''  __builtin_amdgcn_s_setprio(3);
  STAMP(0);
''')
rep('''  const T* xb = reinterpret_cast<const T*>(p.x) + (long)b * p.x_is;''', '''  STAMP(1);
  const T* xb = reinterpret_cast<const T*>(p.x) + (long)b * p.x_is;''')
rep('''  // first chunks on their way before anything else is set up
''', '''  STAMP(2);
''')
rep('''  int stage = 0;
  for (int chunk''', '''  int stage = 0;
  STAMP(3);
  for (int chunk''')
rep('''    __builtin_amdgcn_s_barrier();     // ... in every wave; and every wave is done with the previous stage
''', '''    __builtin_amdgcn_s_barrier();     // ... in every wave; and every wave is done with the previous stage
    if (chunk == 0) STAMP(4);
    wait_sum += __builtin_amdgcn_s_memtime() - t_top;
''')
rep('''    // this chunk's DMA has landed (the DIST-1 younger chunks may stay in flight) ...
''', '''    const unsigned long long t_top = __builtin_amdgcn_s_memtime();
''')
rep('''  int stage = 0;
  STAMP(3);
  for (int chunk''', '''  unsigned long long wait_sum = 0;
  int stage = 0;
  STAMP(3);
  for (int chunk''')
rep('''  // ---- epilogue ----
''', '''  STAMP(5);
  if (threadIdx.x == 0) g_cyc[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 8 + 1] = wait_sum;
''')
rep('''    T* yb = reinterpret_cast<T*>(p.y) + (long)b * p.y_is;
''', '''    STAMP(6);
    T* yb = reinterpret_cast<T*>(p.y) + (long)b * p.y_is;
''')
rep('''    return;
  }
  // general form''', '''    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);
    return;
  }
  // general form''')
i = s.index("template <typename T, int STRIDE, int NB, int RW, int LTW, int KS, int WAVES = 4, int CW = 1>\nint launch(")
j = s.rindex("}", 0, i)
s = s[:j] + '  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n  STAMP(7);\n' + s[j:]
open('scratch/conv_dbg.hip', 'w').write(s)
c = "cd scratch && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../include -I../super-resolution_amd/csrc -c conv_dbg.hip -o conv_STAMP.o && hipcc -shared -fPIC --offload-arch=gfx950 ../super-resolution_amd/csrc/api.o conv_STAMP.o ../super-resolution_amd/csrc/conv_pair.o ../super-resolution_amd/csrc/wgrad.o ../super-resolution_amd/csrc/elementwise.o ../super-resolution_amd/csrc/tiles.o -o libsrx_STAMP.so"
subprocess.check_call(c, shell=True)
print("built")
