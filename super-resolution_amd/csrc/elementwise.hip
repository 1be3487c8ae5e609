// Bandwidth-bound kernels of libsrx: layout conversion, weight packing, pooling,
// pixel (un)shuffle helpers, reductions, losses, spectral norm, Adam.
// All are wave64, one 16/32-byte octet per lane per access, grid-strided.
#include <mutex>
#include "srx_common.h"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(long n, int per_block = TPB, long cap = 256L * 16) {
  long g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// ------------------------------------------------------------------ layout
template <typename T>
__global__ void nchw_to_c8_kernel(const float* __restrict__ src, T* __restrict__ dst, long dst_is,
                                  int B, int C, int H, int W) {
  const int C8 = (C + 7) / 8;
  const long HW = (long)H * W;
  const long n = (long)B * C8 * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long pix = i % HW;
    int oct = (int)((i / HW) % C8);
    int b = (int)(i / (HW * C8));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = oct * 8 + e;
      v[e] = c < C ? src[((long)b * C + c) * HW + pix] : 0.f;
    }
    store_oct(dst + (long)b * dst_is + ((long)oct * HW + pix) * 8, v);
  }
}

template <typename T>
__global__ void c8_to_nchw_kernel(const T* __restrict__ src, long src_is, float* __restrict__ dst,
                                  int B, int C, int H, int W) {
  const int C8 = (C + 7) / 8;
  const long HW = (long)H * W;
  const long n = (long)B * C8 * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long pix = i % HW;
    int oct = (int)((i / HW) % C8);
    int b = (int)(i / (HW * C8));
    float v[8];
    load_oct(src + (long)b * src_is + ((long)oct * HW + pix) * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = oct * 8 + e;
      if (c < C) dst[((long)b * C + c) * HW + pix] = v[e];
    }
  }
}

// ----------------------------------------------------------------- packing
// One (32-row block, 16-channel chunk) unit per workgroup pass: the unit's taps*512 source values are
// read in SOURCE order (runs of 16*taps floats per output channel, or of 32*taps per input channel for
// transposed segments), dropped into an LDS image of the packed layout, and written out as whole
// 16/32-byte octets — both sides of the gather are coalesced.
constexpr int PACK_PLANE = 520;   // floats per tap plane in LDS (512 + a bank offset of 8)
struct PackCol { const float* base; int nstep; float scale; int flip; };
constexpr int PACK_MAX_SEGS = 8;     // segments of a layer cached in LDS (a dense block's backward has 5)
template <typename T, int taps>      // taps is a compile-time constant: the index arithmetic is all divisions by it
__device__ __forceinline__ void pack_units(const srx_pack_layer& L, const srx_pack_seg* __restrict__ segs,
                                           const float* const* __restrict__ seg_scale_ptr, float* tile,
                                           srx_pack_seg* ssegs, PackCol* cols, int u_begin, int u_step,
                                           int u_end) {
  const int nunits = u_end < L.nblk * L.nchunk ? u_end : L.nblk * L.nchunk;
  static_assert(512 % TPB == 0, "pairs per thread");
  T* dst = reinterpret_cast<T*>(L.dst_w);
  // the layer's segments, with the run-time scale folded in, next to the tile
  const int nseg = L.seg_count < PACK_MAX_SEGS ? L.seg_count : PACK_MAX_SEGS;
  if ((int)threadIdx.x < nseg) {
    srx_pack_seg S = segs[L.seg_begin + threadIdx.x];
    if (seg_scale_ptr && seg_scale_ptr[L.seg_begin + threadIdx.x]) S.scale *= *seg_scale_ptr[L.seg_begin + threadIdx.x];
    ssegs[threadIdx.x] = S;
  }
  __syncthreads();
  const bool tr = nseg > 0 && ssegs[0].transposed != 0;   // read order only
  for (int u = u_begin; u < nunits; u += u_step) {
    const int nb = u / L.nchunk, chunk = u - nb * L.nchunk;
    // per input channel of the chunk (16 of them): where its source run starts for output channel 0,
    // the step per output channel, the scale, and whether the taps are flipped — one segment
    // look-up per channel instead of one per (row, channel)
    if (threadIdx.x < 16) {
      const int c = chunk * 16 + (int)threadIdx.x;
      PackCol col = {nullptr, 0, 0.f, 0};
      if (c < L.cin) {
        for (int sI = 0; sI < L.seg_count; ++sI) {
          const srx_pack_seg& S = sI < PACK_MAX_SEGS ? ssegs[sI] : segs[L.seg_begin + sI];
          if (c >= S.c_begin && c < S.c_begin + S.c_count) {
            col.scale = S.scale;
            if (sI >= PACK_MAX_SEGS && seg_scale_ptr && seg_scale_ptr[L.seg_begin + sI])
              col.scale *= *seg_scale_ptr[L.seg_begin + sI];
            const int cc = c - S.c_begin;
            col.flip = S.transposed != 0;
            col.base = S.src + (col.flip ? ((long)(S.src_o_off + cc) * S.cin_src + S.src_off) * taps
                                         : (long)(S.src_off + cc) * taps);
            col.nstep = col.flip ? taps : S.cin_src * taps;
            break;
          }
        }
      }
      cols[threadIdx.x] = col;
    }
    __syncthreads();
    // a thread owns (row, channel) pairs: `taps` consecutive source floats each; neighbouring lanes
    // hold neighbouring runs of the source
#pragma unroll
    for (int k = 0; k < 512 / TPB; ++k) {
      const int pr = k * TPB + (int)threadIdx.x;
      const int row = tr ? (pr & 31) : (pr >> 4);
      const int cl = tr ? (pr >> 5) : (pr & 15);
      const int n = nb * 32 + srx_row_to_chan(row, L.perm);
      const PackCol col = cols[cl];
      const float* src = (n < L.cout && col.base) ? col.base + (long)n * col.nstep : nullptr;
      float v[taps];
#pragma unroll
      for (int t = 0; t < taps; ++t) v[t] = src ? src[t] * col.scale : 0.f;
      float* out = tile + ((cl >> 3) * 32 + row) * 8 + (cl & 7);
#pragma unroll
      for (int t = 0; t < taps; ++t) out[(col.flip ? taps - 1 - t : t) * PACK_PLANE] = v[t];
    }
    __syncthreads();
    T* d = dst + (long)u * taps * 512;
    for (int o = threadIdx.x; o < taps * 64; o += blockDim.x) {
      const int t = o >> 6, r = o & 63;
      store_oct(d + (long)o * 8, &tile[t * PACK_PLANE + r * 8]);
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void pack_bias(const srx_pack_layer& L) {
  for (int i = threadIdx.x; i < L.nblk * 32; i += blockDim.x) {
    int nb = i >> 5, hh = (i >> 4) & 1, reg = i & 15;
    int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
    int n = nb * 32 + srx_row_to_chan(row, L.perm);
    L.dst_bias[i] = (L.src_bias && n < L.cout) ? L.src_bias[n] : 0.f;
  }
}

// grid = (units, ceil(nlayers / PACK_LPB)): a workgroup serves its unit index in PACK_LPB consecutive
// layers.  Most layers have far fewer units than the largest one, and a launch of (max units x
// layers) workgroups that exit at once is bound by the dispatcher (45 000 workgroups: 150 us).
constexpr int PACK_LPB = 2;
template <typename T>
__global__ void pack_kernel(const srx_pack_layer* __restrict__ layers,
                            const srx_pack_seg* __restrict__ segs,
                            const float* const* __restrict__ seg_scale_ptr, int nlayers) {
  __shared__ __attribute__((aligned(16))) float tile[9 * PACK_PLANE];
  __shared__ srx_pack_seg ssegs[PACK_MAX_SEGS];
  __shared__ PackCol cols[16];
  for (int li = blockIdx.y * PACK_LPB; li < nlayers && li < (int)(blockIdx.y + 1) * PACK_LPB; ++li) {
    const srx_pack_layer L = layers[li];
    if ((int)blockIdx.x < L.nblk * L.nchunk) {
      if (L.taps == 9) pack_units<T, 9>(L, segs, seg_scale_ptr, tile, ssegs, cols, blockIdx.x, gridDim.x, 1 << 30);
      else if (L.taps == 1) pack_units<T, 1>(L, segs, seg_scale_ptr, tile, ssegs, cols, blockIdx.x, gridDim.x, 1 << 30);
    }
    if (blockIdx.x == 0 && L.dst_bias) pack_bias(L);
  }
}

// The same packing driven by a flat table of (layer, unit) pairs, one entry per unit of every layer
// (layer << 16 | unit): a grid of a few workgroups per CU walks the table.  The two-dimensional
// grid above launches (largest layer's units, capped) x layers workgroups, two thirds of which
// find nothing to do for a generator's 701 layers, and is bound by the rate at which workgroups
// are dispatched; here every workgroup that starts has work.
template <typename T>
__global__ void pack_flat_kernel(const srx_pack_layer* __restrict__ layers,
                                 const srx_pack_seg* __restrict__ segs,
                                 const float* const* __restrict__ seg_scale_ptr,
                                 const unsigned* __restrict__ unit_table, int total_units) {
  __shared__ __attribute__((aligned(16))) float tile[9 * PACK_PLANE];
  __shared__ srx_pack_seg ssegs[PACK_MAX_SEGS];
  __shared__ PackCol cols[16];
  for (int gu = blockIdx.x; gu < total_units; gu += gridDim.x) {
    const unsigned ent = unit_table[gu];
    const srx_pack_layer L = layers[ent >> 16];
    const int u = (int)(ent & 0xffffu);
    if (L.taps == 9) pack_units<T, 9>(L, segs, seg_scale_ptr, tile, ssegs, cols, u, 1, u + 1);
    else if (L.taps == 1) pack_units<T, 1>(L, segs, seg_scale_ptr, tile, ssegs, cols, u, 1, u + 1);
    if (u == 0 && L.dst_bias) pack_bias(L);
    __syncthreads();                 // ssegs / cols / tile are rewritten by the next unit
  }
}

__global__ void unpack_wgrad_kernel(const srx_unpack_item* __restrict__ items) {
  // source [tap][cout*cin] (tap-major staging), destination [cout*cin][tap] (nn.Conv2d layout).
  // A thread owns one (cout, cin) pair: its `taps` reads are coalesced across the wave (one per tap
  // plane), its writes are `taps` consecutive floats next to its neighbours'.
  // With nslices > 1 the source is nslices partial tensors slice_stride floats apart (the
  // deterministic weight-gradient form), added up in slice order: a fixed summation order.
  const srx_unpack_item it = items[blockIdx.y];
  const long plane = (long)it.cout * it.cin;
  const int ns = it.nslices > 1 ? it.nslices : 1;
  for (long oc = blockIdx.x * (long)blockDim.x + threadIdx.x; oc < plane; oc += (long)gridDim.x * blockDim.x) {
    float* d = it.dst + oc * it.taps;
    if (it.taps == 9) {
      float v[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = it.src[t * plane + oc];
      for (int sl = 1; sl < ns; ++sl) {
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] += it.src[sl * it.slice_stride + t * plane + oc];
      }
      if (it.accumulate) {
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] += d[t];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) d[t] = v[t];
    } else {
      for (int t = 0; t < it.taps; ++t) {
        float v = it.src[t * plane + oc];
        for (int sl = 1; sl < ns; ++sl) v += it.src[sl * it.slice_stride + t * plane + oc];
        d[t] = it.accumulate ? d[t] + v : v;
      }
    }
  }
}

// -------------------------------------------------------------- reductions
template <typename T>
__global__ void channel_sum_kernel(const T* __restrict__ x, long is, int B, long HW,
                                   float* __restrict__ out, float scale) {
  __shared__ float sh[4];
  const int oct = blockIdx.y;
  const long n = (long)B * HW;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long pix = i % HW;
    long b = i / HW;
    float v[8];
    load_oct(x + b * is + ((long)oct * HW + pix) * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = block_sum(acc[e], sh);
    if (threadIdx.x == 0) atomicAdd(out + oct * 8 + e, t * scale);
  }
}

// ------------------------------------------------------- shuffle / stuffing
template <typename T>
__global__ void pixel_unshuffle2_kernel(const T* __restrict__ x, long x_is, T* __restrict__ y,
                                        long y_is, int B, int C, int H, int W) {
  // x: C channels at 2H x 2W ; y: 4C channels at H x W ; y[4c+2i+j][h][w] = x[c][2h+i][2w+j].
  // A thread takes the 2x2 quad of one input octet (read once, 4 x 16 bytes) and writes the four
  // output octets it becomes (input channels 2k, 2k+1 -> output octet 4*ci8 + k).
  const int Ci8 = (C + 7) / 8;
  const int Co8 = (4 * C + 7) / 8;
  const long HW = (long)H * W;
  const long n = (long)B * Ci8 * HW;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    const long pix = idx % HW;
    const int ci8 = (int)((idx / HW) % Ci8);
    const int b = (int)(idx / (HW * Ci8));
    const int h = (int)(pix / W), w = (int)(pix % W);
    const T* xp = x + (long)b * x_is + (long)ci8 * (4 * HW * 8);
    float t[4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      load_oct(xp + ((long)(2 * h + (s >> 1)) * (2 * W) + 2 * w + (s & 1)) * 8, t[s]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oo = 4 * ci8 + k;
      if (oo >= Co8) break;
      const int c0 = ci8 * 8 + 2 * k;
      float v[8];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        v[s] = c0 < C ? t[s][2 * k] : 0.f;
        v[4 + s] = c0 + 1 < C ? t[s][2 * k + 1] : 0.f;
      }
      store_oct(y + (long)b * y_is + ((long)oo * HW + pix) * 8, v);
    }
  }
}

template <typename T>
__global__ void zero_stuff2_kernel(const T* __restrict__ x, long x_is, T* __restrict__ z, long z_is,
                                   int B, int noct, int H, int W) {
  const long HW2 = 4L * H * W;
  const long n = (long)B * noct * HW2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    long pix = idx % HW2;
    int oct = (int)((idx / HW2) % noct);
    int b = (int)(idx / (HW2 * noct));
    int Y = (int)(pix / (2 * W)), X = (int)(pix % (2 * W));
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!(Y & 1) && !(X & 1))
      load_oct(x + (long)b * x_is + ((long)oct * H * W + (long)(Y >> 1) * W + (X >> 1)) * 8, v);
    store_oct(z + (long)b * z_is + ((long)oct * HW2 + pix) * 8, v);
  }
}

// ------------------------------------------------------------------ pooling
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, long x_is, T* __restrict__ y, long y_is,
                                    int B, int noct, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long OHW = (long)OH * OW;
  const long n = (long)B * noct * OHW;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    long pix = idx % OHW;
    int oct = (int)((idx / OHW) % noct);
    int b = (int)(idx / (OHW * noct));
    int oy = (int)(pix / OW), ox = (int)(pix % OW);
    const T* xp = x + (long)b * x_is + (long)oct * H * W * 8;
    float m[8], t[8];
    load_oct(xp + ((long)(2 * oy) * W + 2 * ox) * 8, m);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
      load_oct(xp + ((long)(2 * oy + (s >> 1)) * W + 2 * ox + (s & 1)) * 8, t);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = t[e] > m[e] ? t[e] : m[e];
    }
    store_oct(y + (long)b * y_is + ((long)oct * OHW + pix) * 8, m);
  }
}

template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, long x_is, const T* __restrict__ gy,
                                    long gy_is, T* __restrict__ gx, long gx_is,
                                    const T* __restrict__ mask, long m_is, float mask_slope,
                                    int B, int noct, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const long OHW = (long)OH * OW;
  const long n = (long)B * noct * OHW;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    long pix = idx % OHW;
    int oct = (int)((idx / OHW) % noct);
    int b = (int)(idx / (OHW * noct));
    int oy = (int)(pix / OW), ox = (int)(pix % OW);
    const long pbase = (long)oct * H * W * 8;
    float t[4][8], g[8];
    int arg[8];
    float m[8];
#pragma unroll
    for (int s = 0; s < 4; ++s)
      load_oct(x + (long)b * x_is + pbase + ((long)(2 * oy + (s >> 1)) * W + 2 * ox + (s & 1)) * 8, t[s]);
    load_oct(gy + (long)b * gy_is + ((long)oct * OHW + pix) * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m[e] = t[0][e]; arg[e] = 0;
#pragma unroll
      for (int s = 1; s < 4; ++s)
        if (t[s][e] > m[e]) { m[e] = t[s][e]; arg[e] = s; }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long off = pbase + ((long)(2 * oy + (s >> 1)) * W + 2 * ox + (s & 1)) * 8;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = arg[e] == s ? g[e] : 0.f;
      if (mask) {
        float mk[8];
        load_oct(mask + (long)b * m_is + off, mk);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = mk[e] > 0.f ? o[e] : o[e] * mask_slope;
      }
      store_oct(gx + (long)b * gx_is + off, o);
    }
  }
}

template <typename T>
__global__ void gap_fwd_kernel(const T* __restrict__ x, long x_is, T* __restrict__ y, int B, int noct, long HW) {
  __shared__ float sh[4];
  const int oct = blockIdx.x % noct, b = blockIdx.x / noct;
  const T* xp = x + (long)b * x_is + (long)oct * HW * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = threadIdx.x; i < HW; i += blockDim.x) {
    float v[8];
    load_oct(xp + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = block_sum(acc[e], sh) / (float)HW;
  if (threadIdx.x == 0) store_oct(y + ((long)oct * B + b) * 8, o);
}

template <typename T>
__global__ void gap_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, long gx_is,
                               const T* __restrict__ mask, long m_is, float mask_slope,
                               int B, int noct, long HW) {
  const long n = (long)B * noct * HW;
  const float inv = 1.f / (float)HW;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
    long pix = idx % HW;
    int oct = (int)((idx / HW) % noct);
    int b = (int)(idx / (HW * noct));
    float g[8];
    load_oct(gy + ((long)oct * B + b) * 8, g);
    const long off = ((long)oct * HW + pix) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] *= inv;
    if (mask) {
      float mk[8];
      load_oct(mask + (long)b * m_is + off, mk);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = mk[e] > 0.f ? g[e] : g[e] * mask_slope;
    }
    store_oct(gx + (long)b * gx_is + off, g);
  }
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, T* __restrict__ y, float a, float bb, long n) {
  const long nocts = (n + 7) / 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nocts; i += (long)gridDim.x * blockDim.x) {
    if (i * 8 + 8 > n) {   // ragged tail: element by element
      for (long e = i * 8; e < n; ++e)
        y[e] = (T)(a * (float)x[e] + (bb != 0.f ? bb * (float)y[e] : 0.f));
      continue;
    }
    float u[8], v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load_oct(x + i * 8, u);
    if (bb != 0.f) load_oct(y + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a * u[e] + bb * v[e];
    store_oct(y + i * 8, v);
  }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ g, const T* __restrict__ m, T* __restrict__ out,
                               float slope, long nocts) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nocts; i += (long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    load_oct(g + i * 8, u);
    load_oct(m + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = v[e] > 0.f ? u[e] : u[e] * slope;
    store_oct(out + i * 8, u);
  }
}

// ------------------------------------------------------------------- losses
template <typename T, int MODE>   // MODE 0: L1 (+grad), 1: squared error sum
__global__ void pix_loss_c8_nchw_kernel(const T* __restrict__ pred, long p_is,
                                        const float* __restrict__ target, float* __restrict__ loss,
                                        T* __restrict__ g, long g_is, float gscale,
                                        int B, int C, int H, int W) {
  __shared__ float sh[4];
  const int C8 = (C + 7) / 8;
  const long HW = (long)H * W;
  const long n = (long)B * C8 * HW;
  const float invN = 1.f / ((float)B * C * HW);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long pix = i % HW;
    int oct = (int)((i / HW) % C8);
    int b = (int)(i / (HW * C8));
    float v[8], go[8];
    const long off = ((long)oct * HW + pix) * 8;
    load_oct(pred + (long)b * p_is + off, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = oct * 8 + e;
      float d = c < C ? v[e] - target[((long)b * C + c) * HW + pix] : 0.f;
      if (MODE == 0) {
        acc += fabsf(d);
        go[e] = d > 0.f ? gscale * invN : (d < 0.f ? -gscale * invN : 0.f);
      } else {
        acc += d * d;
      }
    }
    if (MODE == 0 && g) store_oct(g + (long)b * g_is + off, go);
  }
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss, MODE == 0 ? t * invN : t);
}

template <typename T>
__global__ void l1_c8_kernel(const T* __restrict__ a, long a_is, const T* __restrict__ bq, long b_is,
                             float* __restrict__ loss, T* __restrict__ ga, long ga_is, float gscale,
                             float mask_slope, int B, int C8, long HW) {
  __shared__ float sh[4];
  const long n = (long)B * C8 * HW;
  const float invN = 1.f / ((float)B * C8 * 8 * HW);
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i % ((long)C8 * HW);
    int b = (int)(i / ((long)C8 * HW));
    float u[8], v[8], go[8];
    load_oct(a + (long)b * a_is + r * 8, u);
    load_oct(bq + (long)b * b_is + r * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float d = u[e] - v[e];
      acc += fabsf(d);
      float gg = d > 0.f ? gscale * invN : (d < 0.f ? -gscale * invN : 0.f);
      go[e] = u[e] > 0.f ? gg : gg * mask_slope;
    }
    if (ga) store_oct(ga + (long)b * ga_is + r * 8, go);
  }
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss, t * invN);
}

template <typename T>
__global__ void bce_logits_kernel(const T* __restrict__ logits, int n, float target, float weight,
                                  float* __restrict__ loss, T* __restrict__ g, float gscale) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float v[8];
    load_oct(logits + (long)i * 8, v);
    float x = v[0];
    acc += fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
    if (g) {
      float sg = 1.f / (1.f + expf(-x));
      float o[8] = {gscale * weight * (sg - target) / (float)n, 0, 0, 0, 0, 0, 0, 0};
      store_oct(g + (long)i * 8, o);
    }
  }
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss, weight * t / (float)n);
}

// ------------------------------------------------------------ spectral norm
// v = W^T u (un-normalised; normalised in sn_finish_kernel).  A workgroup owns 32 columns: lane
// group g of 8 walks rows g, g+8, ... (each row access is one 128-byte segment), four loads in
// flight per lane, then the 8 partial sums meet in LDS in a fixed order (deterministic, no atomics).
constexpr int SN_COLS = 32;
__global__ void sn_wt_u_kernel(const srx_sn_layer* __restrict__ layers) {
  __shared__ float part[TPB / SN_COLS][SN_COLS];
  const srx_sn_layer L = layers[blockIdx.y];
  if ((int)blockIdx.x * SN_COLS >= L.cols) return;
  constexpr int PH = TPB / SN_COLS;
  const int cl = threadIdx.x % SN_COLS, ph = threadIdx.x / SN_COLS;
  const int c = blockIdx.x * SN_COLS + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < L.cols) {
    const float* w = L.w + c;
    int r = ph;
    for (; r + 3 * PH < L.rows; r += 4 * PH) {
      a0 += w[(long)r * L.cols] * L.u[r];
      a1 += w[(long)(r + PH) * L.cols] * L.u[r + PH];
      a2 += w[(long)(r + 2 * PH) * L.cols] * L.u[r + 2 * PH];
      a3 += w[(long)(r + 3 * PH) * L.cols] * L.u[r + 3 * PH];
    }
    for (; r < L.rows; r += PH) a0 += w[(long)r * L.cols] * L.u[r];
  }
  part[ph][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ph == 0 && c < L.cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < PH; ++k) t += part[k][cl];
    L.v[c] = t;
  }
}

__device__ __forceinline__ float vec_inv_norm(const float* v, int n, float eps, float* sh) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += v[i] * v[i];
  float t = block_sum(a, sh);
  return 1.f / fmaxf(sqrtf(t), eps);
}

// wv_base != nullptr: layer l's W v rows live at wv_base + l * wv_stride instead of at layers[l].wv
__global__ void sn_w_v_kernel(const srx_sn_layer* __restrict__ layers, int do_pi, float* wv_base, int wv_stride) {
  __shared__ float sh[4];
  const srx_sn_layer L = layers[blockIdx.y];
  float* const wv = wv_base ? wv_base + (long)blockIdx.y * wv_stride : L.wv;
  if ((int)blockIdx.x * 4 >= L.rows) return;
  const float sc = do_pi ? vec_inv_norm(L.v, L.cols, 1e-12f, sh) : 1.f;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= L.rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int c = lane; c < L.cols; c += 64) acc += L.w[(long)row * L.cols + c] * L.v[c];
  acc = wave_sum(acc);
  if (lane == 0) wv[row] = acc * sc;
}

__global__ void sn_finish_kernel(const srx_sn_layer* __restrict__ layers, int do_pi, const float* wv_base,
                                 int wv_stride) {
  __shared__ float sh[4];
  const srx_sn_layer L = layers[blockIdx.x];
  const float* const wv = wv_base ? wv_base + (long)blockIdx.x * wv_stride : L.wv;
  if (do_pi) {
    const float sv = vec_inv_norm(L.v, L.cols, 1e-12f, sh);
    __syncthreads();
    for (int i = threadIdx.x; i < L.cols; i += blockDim.x) L.v[i] *= sv;
    const float su = vec_inv_norm(wv, L.rows, 1e-12f, sh);
    for (int i = threadIdx.x; i < L.rows; i += blockDim.x) L.u[i] = wv[i] * su;
    __syncthreads();
  }
  float a = 0.f;
  for (int i = threadIdx.x; i < L.rows; i += blockDim.x) a += L.u[i] * wv[i];
  float sigma = block_sum(a, sh);
  if (threadIdx.x == 0) L.inv_sigma[0] = 1.f / sigma;
}

// ---- static scratch of the entry point that allocates nothing ----
// srx_sn_power_iter_noalloc keeps the W v rows of all layers here instead of in caller-provided
// buffers.  One copy per GPU the library runs on, held until the process ends.  In host code the
// array's name is only a handle for hipGetSymbolAddress: the device address it returns (one per
// device) is what kernels and memory calls are given.
constexpr int SN_SCRATCH_FLOATS = 64 * 1024;                     // 256 KB
static __device__ __align__(256) float sn_scratch[SN_SCRATCH_FLOATS];

constexpr int SN_MAX_DEVICES = 64;
static float* g_sn_scratch_ptr[SN_MAX_DEVICES];
static std::mutex g_sn_scratch_mutex;

// Device address of sn_scratch on the CURRENT device, resolved once per device.
static int sn_scratch_ptr(float** out) {
  int dev = -1;
  SRX_CHECK_HIP(hipGetDevice(&dev));
  SRX_CHECK_ARG(dev >= 0 && dev < SN_MAX_DEVICES, "device ordinal %d outside the library's table of %d", dev,
                SN_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_sn_scratch_mutex);
  if (g_sn_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    SRX_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(sn_scratch)));
    SRX_CHECK_ARG(p != nullptr, "hipGetSymbolAddress returned a null address for the static scratch");
    g_sn_scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = g_sn_scratch_ptr[dev];
  return SRX_OK;
}

__global__ void sn_bwd_dot_kernel(const srx_sn_bwd_layer* __restrict__ layers) {
  __shared__ float sh[4];
  const srx_sn_bwd_layer L = layers[blockIdx.y];
  const long n = (long)L.rows * L.cols;
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    a += L.dw[i] * L.w[i];
  float t = block_sum(a, sh);
  if (threadIdx.x == 0 && (long)blockIdx.x * blockDim.x < n) atomicAdd(L.dot, t);
}
__global__ void sn_bwd_zero_kernel(const srx_sn_bwd_layer* __restrict__ layers, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) layers[i].dot[0] = 0.f;
}
__global__ void sn_bwd_apply_kernel(const srx_sn_bwd_layer* __restrict__ layers) {
  const srx_sn_bwd_layer L = layers[blockIdx.y];
  const long n = (long)L.rows * L.cols;
  const float is = L.inv_sigma[0];
  const float k = L.dot[0] * is * is;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / L.cols), c = (int)(i % L.cols);
    L.dw[i] = L.dw[i] * is - k * L.u[r] * L.v[c];
  }
}

// --------------------------------------------------------------------- Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

#define STREAM reinterpret_cast<hipStream_t>(stream)
#define DISPATCH_T(dtype, CALL_BF16, CALL_F32)                       \
  do {                                                               \
    if ((dtype) == SRX_BF16) { CALL_BF16; }                          \
    else if ((dtype) == SRX_F32) { CALL_F32; }                       \
    else { srx_set_error("bad dtype %d", (dtype)); return SRX_ERR_INVALID; } \
  } while (0)

extern "C" {

int srx_nchw_to_c8(int dtype, const float* src, void* dst, int64_t dst_is, int B, int C, int H, int W, void* stream) {
  SRX_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srx_nchw_to_c8: bad argument");
  long n = (long)B * ((C + 7) / 8) * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(nchw_to_c8_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, src, (bf16*)dst, dst_is, B, C, H, W),
             hipLaunchKernelGGL(nchw_to_c8_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, src, (float*)dst, dst_is, B, C, H, W));
  return srx_check_launch("nchw_to_c8");
}

int srx_c8_to_nchw(int dtype, const void* src, int64_t src_is, float* dst, int B, int C, int H, int W, void* stream) {
  SRX_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srx_c8_to_nchw: bad argument");
  long n = (long)B * ((C + 7) / 8) * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(c8_to_nchw_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)src, src_is, dst, B, C, H, W),
             hipLaunchKernelGGL(c8_to_nchw_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)src, src_is, dst, B, C, H, W));
  return srx_check_launch("c8_to_nchw");
}

int srx_pack_weights_scaled(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs,
                            const float* const* seg_scale_ptr, int nlayers, int64_t max_elems, void* stream) {
  SRX_CHECK_ARG(layers && segs && nlayers > 0 && max_elems > 0, "srx_pack_weights: bad argument");
  dim3 grid(grid_for(max_elems / 512, 1, 64), (nlayers + PACK_LPB - 1) / PACK_LPB);   // x: (block, chunk) units, at least taps*512 values each
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_kernel<bf16>, grid, dim3(TPB), 0, STREAM, layers, segs, seg_scale_ptr, nlayers),
             hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(TPB), 0, STREAM, layers, segs, seg_scale_ptr, nlayers));
  return srx_check_launch("pack_kernel");
}

int srx_pack_weights_flat(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs,
                          const float* const* seg_scale_ptr, const uint32_t* unit_table, int total_units,
                          void* stream) {
  SRX_CHECK_ARG(layers && segs && unit_table && total_units > 0, "srx_pack_weights_flat: bad argument");
  dim3 grid(grid_for(total_units, 1, 1L << 20));      // one workgroup per unit: the units hide each other's load latency
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_flat_kernel<bf16>, grid, dim3(TPB), 0, STREAM, layers, segs, seg_scale_ptr, unit_table, total_units),
             hipLaunchKernelGGL(pack_flat_kernel<float>, grid, dim3(TPB), 0, STREAM, layers, segs, seg_scale_ptr, unit_table, total_units));
  return srx_check_launch("pack_flat_kernel");
}

int srx_pack_weights(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs, int nlayers,
                     int64_t max_elems, void* stream) {
  return srx_pack_weights_scaled(dtype, layers, segs, nullptr, nlayers, max_elems, stream);
}

int srx_unpack_wgrad(const srx_unpack_item* items, int nitems, int64_t max_elems, void* stream) {
  SRX_CHECK_ARG(items && nitems > 0 && max_elems > 0, "srx_unpack_wgrad: bad argument");
  dim3 grid(grid_for(max_elems / 8, TPB, 64), nitems);     // max_elems counts taps too; a thread takes all taps of a pair
  hipLaunchKernelGGL(unpack_wgrad_kernel, grid, dim3(TPB), 0, STREAM, items);
  return srx_check_launch("unpack_wgrad");
}

int srx_channel_sum(int dtype, const void* x, int64_t is, int noct, int B, int H, int W, float* out,
                    float scale, void* stream) {
  SRX_CHECK_ARG(x && out && noct > 0 && B > 0 && H > 0 && W > 0, "srx_channel_sum: bad argument");
  long HW = (long)H * W;
  dim3 grid(grid_for((long)B * HW, TPB * 8, 64), noct);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(channel_sum_kernel<bf16>, grid, dim3(TPB), 0, STREAM, (const bf16*)x, is, B, HW, out, scale),
             hipLaunchKernelGGL(channel_sum_kernel<float>, grid, dim3(TPB), 0, STREAM, (const float*)x, is, B, HW, out, scale));
  return srx_check_launch("channel_sum");
}

int srx_pixel_unshuffle2(int dtype, const void* x, int64_t x_is, void* y, int64_t y_is, int B, int C,
                         int H, int W, void* stream) {
  SRX_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0 && C % 2 == 0, "srx_pixel_unshuffle2: bad argument");
  long n = (long)B * ((C + 7) / 8) * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pixel_unshuffle2_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)x, x_is, (bf16*)y, y_is, B, C, H, W),
             hipLaunchKernelGGL(pixel_unshuffle2_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)x, x_is, (float*)y, y_is, B, C, H, W));
  return srx_check_launch("pixel_unshuffle2");
}

int srx_zero_stuff2(int dtype, const void* x, int64_t x_is, void* z, int64_t z_is, int B, int noct,
                    int H, int W, void* stream) {
  SRX_CHECK_ARG(x && z && B > 0 && noct > 0 && H > 0 && W > 0, "srx_zero_stuff2: bad argument");
  long n = (long)B * noct * 4 * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(zero_stuff2_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)x, x_is, (bf16*)z, z_is, B, noct, H, W),
             hipLaunchKernelGGL(zero_stuff2_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)x, x_is, (float*)z, z_is, B, noct, H, W));
  return srx_check_launch("zero_stuff2");
}

int srx_maxpool2_fwd(int dtype, const void* x, int64_t x_is, void* y, int64_t y_is, int B, int noct,
                     int H, int W, void* stream) {
  SRX_CHECK_ARG(x && y && B > 0 && noct > 0 && H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0,
                "srx_maxpool2_fwd: bad argument (H, W must be even)");
  long n = (long)B * noct * (H / 2) * (W / 2);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)x, x_is, (bf16*)y, y_is, B, noct, H, W),
             hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)x, x_is, (float*)y, y_is, B, noct, H, W));
  return srx_check_launch("maxpool2_fwd");
}

int srx_maxpool2_bwd(int dtype, const void* x, int64_t x_is, const void* gy, int64_t gy_is, void* gx,
                     int64_t gx_is, const void* mask, int64_t m_is, float mask_slope, int B, int noct,
                     int H, int W, void* stream) {
  SRX_CHECK_ARG(x && gy && gx && B > 0 && noct > 0 && H > 1 && W > 1 && H % 2 == 0 && W % 2 == 0,
                "srx_maxpool2_bwd: bad argument (H, W must be even)");
  long n = (long)B * noct * (H / 2) * (W / 2);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)x, x_is, (const bf16*)gy, gy_is, (bf16*)gx, gx_is, (const bf16*)mask, m_is, mask_slope, B, noct, H, W),
             hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)x, x_is, (const float*)gy, gy_is, (float*)gx, gx_is, (const float*)mask, m_is, mask_slope, B, noct, H, W));
  return srx_check_launch("maxpool2_bwd");
}

int srx_gap_fwd(int dtype, const void* x, int64_t x_is, void* y, int B, int noct, int H, int W, void* stream) {
  SRX_CHECK_ARG(x && y && B > 0 && noct > 0 && H > 0 && W > 0, "srx_gap_fwd: bad argument");
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gap_fwd_kernel<bf16>, dim3(B * noct), dim3(TPB), 0, STREAM, (const bf16*)x, x_is, (bf16*)y, B, noct, (long)H * W),
             hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3(B * noct), dim3(TPB), 0, STREAM, (const float*)x, x_is, (float*)y, B, noct, (long)H * W));
  return srx_check_launch("gap_fwd");
}

int srx_gap_bwd(int dtype, const void* gy, void* gx, int64_t gx_is, const void* mask, int64_t m_is,
                float mask_slope, int B, int noct, int H, int W, void* stream) {
  SRX_CHECK_ARG(gy && gx && B > 0 && noct > 0 && H > 0 && W > 0, "srx_gap_bwd: bad argument");
  long n = (long)B * noct * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(gap_bwd_kernel<bf16>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const bf16*)gy, (bf16*)gx, gx_is, (const bf16*)mask, m_is, mask_slope, B, noct, (long)H * W),
             hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid_for(n)), dim3(TPB), 0, STREAM, (const float*)gy, (float*)gx, gx_is, (const float*)mask, m_is, mask_slope, B, noct, (long)H * W));
  return srx_check_launch("gap_bwd");
}

int srx_axpby(int dtype, const void* x, void* y, float a, float b, int64_t n, void* stream) {
  SRX_CHECK_ARG(x && y && n > 0, "srx_axpby: null pointer or n <= 0");
  SRX_CHECK_ARG(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "srx_axpby: x and y must be 16-byte aligned");
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(axpby_kernel<bf16>, dim3(grid_for((n + 7) / 8)), dim3(TPB), 0, STREAM, (const bf16*)x, (bf16*)y, a, b, (long)n),
             hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid_for((n + 7) / 8)), dim3(TPB), 0, STREAM, (const float*)x, (float*)y, a, b, (long)n));
  return srx_check_launch("axpby");
}

int srx_act_bwd(int dtype, const void* g, const void* act_out, void* out, float slope, int64_t n, void* stream) {
  SRX_CHECK_ARG(g && act_out && out && n > 0 && n % 8 == 0, "srx_act_bwd: n must be a positive multiple of 8");
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(grid_for(n / 8)), dim3(TPB), 0, STREAM, (const bf16*)g, (const bf16*)act_out, (bf16*)out, slope, (long)(n / 8)),
             hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(n / 8)), dim3(TPB), 0, STREAM, (const float*)g, (const float*)act_out, (float*)out, slope, (long)(n / 8)));
  return srx_check_launch("act_bwd");
}

int srx_l1_c8_nchw(int dtype, const void* pred, int64_t p_is, const float* target, float* loss, void* g,
                   int64_t g_is, float gscale, int B, int C, int H, int W, void* stream) {
  SRX_CHECK_ARG(pred && target && loss && B > 0 && C > 0 && H > 0 && W > 0, "srx_l1_c8_nchw: bad argument");
  long n = (long)B * ((C + 7) / 8) * H * W;
  auto kb = pix_loss_c8_nchw_kernel<bf16, 0>;
  auto kf = pix_loss_c8_nchw_kernel<float, 0>;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(kb, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const bf16*)pred, p_is, target, loss, (bf16*)g, g_is, gscale, B, C, H, W),
             hipLaunchKernelGGL(kf, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const float*)pred, p_is, target, loss, (float*)g, g_is, gscale, B, C, H, W));
  return srx_check_launch("l1_c8_nchw");
}

int srx_sqerr_c8_nchw(int dtype, const void* pred, int64_t p_is, const float* target, float* out, int B,
                      int C, int H, int W, void* stream) {
  SRX_CHECK_ARG(pred && target && out && B > 0 && C > 0 && H > 0 && W > 0, "srx_sqerr_c8_nchw: bad argument");
  long n = (long)B * ((C + 7) / 8) * H * W;
  auto kb = pix_loss_c8_nchw_kernel<bf16, 1>;
  auto kf = pix_loss_c8_nchw_kernel<float, 1>;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(kb, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const bf16*)pred, p_is, target, out, (bf16*)nullptr, 0L, 0.f, B, C, H, W),
             hipLaunchKernelGGL(kf, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const float*)pred, p_is, target, out, (float*)nullptr, 0L, 0.f, B, C, H, W));
  return srx_check_launch("sqerr_c8_nchw");
}

int srx_l1_c8(int dtype, const void* a, int64_t a_is, const void* b, int64_t b_is, float* loss, void* ga,
              int64_t ga_is, float gscale, float mask_slope, int B, int C, int H, int W, void* stream) {
  SRX_CHECK_ARG(a && b && loss && B > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0, "srx_l1_c8: bad argument (C %% 8 must be 0)");
  long n = (long)B * (C / 8) * H * W;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(l1_c8_kernel<bf16>, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const bf16*)a, a_is, (const bf16*)b, b_is, loss, (bf16*)ga, ga_is, gscale, mask_slope, B, C / 8, (long)H * W),
             hipLaunchKernelGGL(l1_c8_kernel<float>, dim3(grid_for(n, TPB, 1024)), dim3(TPB), 0, STREAM, (const float*)a, a_is, (const float*)b, b_is, loss, (float*)ga, ga_is, gscale, mask_slope, B, C / 8, (long)H * W));
  return srx_check_launch("l1_c8");
}

int srx_bce_logits(int dtype, const void* logits, int n, float target, float weight, float* loss, void* g,
                   float gscale, void* stream) {
  SRX_CHECK_ARG(logits && loss && n > 0, "srx_bce_logits: bad argument");
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bce_logits_kernel<bf16>, dim3(1), dim3(TPB), 0, STREAM, (const bf16*)logits, n, target, weight, loss, (bf16*)g, gscale),
             hipLaunchKernelGGL(bce_logits_kernel<float>, dim3(1), dim3(TPB), 0, STREAM, (const float*)logits, n, target, weight, loss, (float*)g, gscale));
  return srx_check_launch("bce_logits");
}

static int sn_power_iter_launch(const srx_sn_layer* layers, int n, int do_pi, int max_rows, int max_cols,
                                float* wv_base, int wv_stride, void* stream) {
  if (do_pi) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((max_cols + SN_COLS - 1) / SN_COLS, n), dim3(TPB), 0, STREAM, layers);
    if (int rc = srx_check_launch("sn_wt_u")) return rc;
  }
  hipLaunchKernelGGL(sn_w_v_kernel, dim3((max_rows + 3) / 4, n), dim3(TPB), 0, STREAM, layers, do_pi, wv_base,
                     wv_stride);
  if (int rc = srx_check_launch("sn_w_v")) return rc;
  hipLaunchKernelGGL(sn_finish_kernel, dim3(n), dim3(TPB), 0, STREAM, layers, do_pi, (const float*)wv_base,
                     wv_stride);
  return srx_check_launch("sn_finish");
}

int srx_sn_power_iter(const srx_sn_layer* layers, int n, int do_pi, int max_rows, int max_cols, void* stream) {
  SRX_CHECK_ARG(layers && n > 0 && max_rows > 0 && max_cols > 0, "srx_sn_power_iter: bad argument");
  return sn_power_iter_launch(layers, n, do_pi, max_rows, max_cols, nullptr, 0, stream);
}

int srx_sn_scratch_floats(void) {
  // also the set-up call of srx_sn_power_iter_noalloc: resolves the static array's address on the
  // current device (hipGetSymbolAddress may load the code object: not something to do for the first
  // time inside a stream capture).  Without a device it only reports the size.
  float* ptr = nullptr;
  (void)sn_scratch_ptr(&ptr);
  return SN_SCRATCH_FLOATS;
}

int srx_sn_power_iter_noalloc(const srx_sn_layer* layers, int n, int do_pi, int max_rows, int max_cols,
                              void* stream) {
  SRX_CHECK_ARG(layers && n > 0 && max_rows > 0 && max_cols > 0, "srx_sn_power_iter_noalloc: bad argument");
  SRX_CHECK_ARG((long)n * max_rows <= (long)SN_SCRATCH_FLOATS,
                "srx_sn_power_iter_noalloc: n * max_rows = %ld floats of scratch needed, the library's static "
                "scratch holds %d; use srx_sn_power_iter with caller-provided wv buffers",
                (long)n * max_rows, SN_SCRATCH_FLOATS);
  float* ptr = nullptr;
  if (int rc = sn_scratch_ptr(&ptr)) return rc;
  // every row the finishing kernel reads is written by the kernel before it in this same call:
  // nothing of an earlier call's contents is read, so the array needs no clearing
  return sn_power_iter_launch(layers, n, do_pi, max_rows, max_cols, ptr, max_rows, stream);
}

int srx_sn_backward(const srx_sn_bwd_layer* layers, int n, int max_elems, int deterministic, void* stream) {
  SRX_CHECK_ARG(layers && n > 0 && max_elems > 0, "srx_sn_backward: bad argument");
  hipLaunchKernelGGL(sn_bwd_zero_kernel, dim3(1), dim3(TPB), 0, STREAM, layers, n);
  dim3 grid(grid_for(max_elems, TPB * 4, 512), n);
  // deterministic: <dW, W> of a layer summed by ONE workgroup (a fixed order instead of atomic adds
  // from many); the elementwise update that follows has no order to keep and always takes the full grid
  hipLaunchKernelGGL(sn_bwd_dot_kernel, deterministic ? dim3(1, n) : grid, dim3(TPB), 0, STREAM, layers);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, grid, dim3(TPB), 0, STREAM, layers);
  return srx_check_launch("sn_backward");
}

int srx_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
             float eps, float weight_decay, int step, float gscale, void* stream) {
  SRX_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "srx_adam: bad argument");
  float bc1 = 1.f - powf(beta1, (float)step);
  float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(TPB), 0, STREAM, p, g, m, v, (long)n, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2s, gscale);
  return srx_check_launch("adam");
}

}  // extern "C"
