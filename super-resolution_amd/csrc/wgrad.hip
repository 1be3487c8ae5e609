// Weight gradient of a 3x3 / 1x1 convolution on 32x32 MFMA tiles (gfx950).
//
// GEMM view per tap:  dW[tap][cout][cin] = sum_pixels GY[cout][p] * X[cin][p + tap],
// i.e. the contraction index is the PIXEL.  Activations are pixel-major (C8
// layout), so both operands are transposed on the way out of LDS with
// ds_read_b64_tr_b16 (bf16) or gathered with ds_read_b32 (fp32).
//
// One workgroup owns a 64(cin) x 64(cout) block of one or several layers (a
// task); its 4 waves each hold the 9-tap accumulators of one 32x32 sub-block
// (144 registers) across all the pixel tiles of the workgroup's slice, then add
// them with fp32 atomics into tap-major gradient tensors ([tap][cout][cin]: the
// 32 lanes of a half-wave are 32 consecutive cin, i.e. 128 contiguous bytes).
// A dense block's five convolutions are one launch: the task table encodes
// which (cin block, cout block) pairs exist and where each lands.
#include "srx_common.h"

namespace {

struct WgradParams {
  const void* x; long x_is;
  const void* gy; long gy_is;
  const srx_wgrad_task* tasks;
  int H, W, OH, OW, B;
  int nslices, tiles_x, tiles_y;
};

template <typename T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };

__device__ __forceinline__ void mma(const bf16x8& a, const bf16x8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const f32x8& a, const f32x8& b, f32x16& c) {
#pragma unroll
  for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 8 consecutive-pixel bf16 values of one channel: two transposed 4x16 block reads.
// addr0/addr1: this lane's byte addresses (LDS) for pixel quads 0 and 1.
__device__ __forceinline__ bf16x8 tr_frag(const char* lds_base, int addr0, int addr1) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + addr0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + addr1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T, int STRIDE, int LTW, int KS, int TPIX>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  typedef typename Frag<T>::type frag_t;
  constexpr int TW = 1 << LTW;
  constexpr int TH = TPIX >> LTW;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int IW = (TW - 1) * STRIDE + KS;
  constexpr int NPX = IH * IW;
  constexpr int PPO = Oct<T>::PPO;
  constexpr int EPP = 16 / sizeof(T);
  constexpr int XP = 8 * NPX * PPO;
  constexpr int GP = 8 * TPIX * PPO;
  constexpr int NXS = (XP + 255) / 256;
  constexpr int NGS = (GP + 255) / 256;
  constexpr int KSTEPS = TPIX / 16;
  constexpr int OB = sizeof(T) * 8;         // bytes per octet slot

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  uint4* sx = smem;
  uint4* sg = smem + XP;
  const char* sxb = reinterpret_cast<const char*>(sx);
  const char* sgb = reinterpret_cast<const char*>(sg);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int task_id = blockIdx.x / p.nslices, slice = blockIdx.x % p.nslices;
  const srx_wgrad_task* __restrict__ tkp = p.tasks + task_id;
  const int sa = wave & 1, sb = wave >> 1;      // cin half, cout half
  float* const outp = tkp->out[sa + 2 * sb];
  const int x_noct = tkp->x_noct, gy_noct = tkp->gy_noct;
  const bool active = outp != nullptr;          // wave-uniform

  const long plane_x = (long)p.H * p.W * 8;
  const long plane_g = (long)p.OH * p.OW * 8;
  const T* xbase = reinterpret_cast<const T*>(p.x) + (long)tkp->x_oct * plane_x;
  const T* gbase = reinterpret_cast<const T*>(p.gy) + (long)tkp->gy_oct * plane_g;

  // ---- per-lane LDS read addresses (bytes), tap (0,0), k-step 0 ----
  int ga[2], xa[2];      // bf16: two transposed reads; fp32: [0] only (element address)
  if constexpr (sizeof(T) == 2) {
    const int gidx = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;
    const int h = gidx >> 1;
    const int osub = 2 * (gidx & 1) + (pq >> 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kp = 8 * h + 4 * t + q;        // pixel inside the 16-pixel k-step
      ga[t] = ((4 * sb + osub) * TPIX + kp) * OB + 8 * (pq & 1);
      const int py = kp >> LTW, px = kp & (TW - 1);   // for k-step 0; later steps add a constant
      xa[t] = ((4 * sa + osub) * NPX + py * STRIDE * IW + px * STRIDE) * OB + 8 * (pq & 1);
    }
  } else {
    const int r = lane & 31, h = lane >> 5;
    ga[0] = ((4 * sb + (r >> 3)) * TPIX + h) * OB + (r & 7) * 4;
    xa[0] = ((4 * sa + (r >> 3)) * NPX + h * STRIDE) * OB + (r & 7) * 4;   // TW >= 2
    ga[1] = xa[1] = 0;
  }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int ntiles = p.B * tiles_per_img;

  for (int tile = slice; tile < ntiles; tile += p.nslices) {
    const int b = tile / tiles_per_img;
    const int tr = tile % tiles_per_img;
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
    const T* xb = xbase + (long)b * p.x_is;
    const T* gb = gbase + (long)b * p.gy_is;

    uint4 xr[NXS], gr[NGS];
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      int piece = tid + i * 256;
      int q = piece % PPO;
      int pix = (piece / PPO) % NPX;
      int o = piece / (PPO * NPX);
      int gy_ = iy0 + pix / IW, gx_ = ix0 + pix % IW;
      bool ok = piece < XP && o < x_noct && gy_ >= 0 && gy_ < p.H && gx_ >= 0 && gx_ < p.W;
      xr[i] = ok ? *reinterpret_cast<const uint4*>(xb + (long)o * plane_x + ((long)gy_ * p.W + gx_) * 8 + q * EPP)
                 : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NGS; ++i) {
      int piece = tid + i * 256;
      int q = piece % PPO;
      int pix = (piece / PPO) % TPIX;
      int o = piece / (PPO * TPIX);
      int oy = oy0 + (pix >> LTW), ox = ox0 + (pix & (TW - 1));
      bool ok = piece < GP && o < gy_noct && oy < p.OH && ox < p.OW;
      gr[i] = ok ? *reinterpret_cast<const uint4*>(gb + (long)o * plane_g + ((long)oy * p.OW + ox) * 8 + q * EPP)
                 : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();          // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      int piece = tid + i * 256;
      if (piece < XP) sx[piece] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < NGS; ++i) {
      int piece = tid + i * 256;
      if (piece < GP) sg[piece] = gr[i];
    }
    __syncthreads();

    if (active) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        // pixel offset of k-step ks relative to k-step 0 (16 pixels further in tile order)
        constexpr int dummy = 0; (void)dummy;
        const int tp0 = ks * 16;
        int gofs, xofs;
        if constexpr (LTW >= 4) {     // 16 pixels stay inside one tile row
          const int py = tp0 >> LTW, px = tp0 & (TW - 1);
          gofs = tp0 * OB;
          xofs = (py * STRIDE * IW + px * STRIDE) * OB;
        } else {                      // TW == 8: a k-step spans two rows; k-step ks starts at row 2*ks
          gofs = tp0 * OB;
          xofs = (2 * ks) * STRIDE * IW * OB;
        }
        frag_t gf;
        if constexpr (sizeof(T) == 2) {
          gf = tr_frag(sgb, ga[0] + gofs, ga[1] + gofs);
        } else {
#pragma unroll
          for (int s = 0; s < 8; ++s)
            gf[s] = *reinterpret_cast<const float*>(sgb + ga[0] + gofs + 2 * s * OB);
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int dy = tap / KS, dx = tap % KS;
          const int tofs = (dy * IW + dx) * OB;
          frag_t xf;
          if constexpr (sizeof(T) == 2) {
            xf = tr_frag(sxb, xa[0] + xofs + tofs, xa[1] + xofs + tofs);
          } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
              // pixel 2s+h of the k-step: for TW >= 16 it is in the same row
              int po;
              if constexpr (LTW >= 4) po = 2 * s * STRIDE * OB;
              else po = (((2 * s) >> 3) * STRIDE * IW + ((2 * s) & 7) * STRIDE) * OB;
              xf[s] = *reinterpret_cast<const float*>(sxb + xa[0] + xofs + tofs + po);
            }
          }
          mma(gf, xf, acc[tap]);
        }
      }
    }
  }

  if (!active) return;
  const int col = lane & 31, hh = lane >> 5;
  const int ld = tkp->ld[sa + 2 * sb], co_n = tkp->co_n[sa + 2 * sb];
  const int ci = tkp->ci0[sa + 2 * sb] + col;
  const int co_base = tkp->co0[sa + 2 * sb];
  const float scale = tkp->scale[sa + 2 * sb];
  if (ci >= ld) return;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co_base + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (co < co_n) atomicAdd(outp + ((long)tap * co_n + co) * ld + ci, acc[tap][i] * scale);
    }
  }
}

template <typename T, int STRIDE, int LTW, int KS, int TPIX>
int launch(const srx_wgrad_desc* d, WgradParams p, hipStream_t stream) {
  constexpr int TW = 1 << LTW, TH = TPIX >> LTW;
  constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
  constexpr int PPO = Oct<T>::PPO;
  constexpr int LDS = (8 * IH * IW * PPO + 8 * TPIX * PPO) * 16;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(TH >= 1, "tile");
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + TH - 1) / TH;
  auto kern = wgrad_kernel<T, STRIDE, LTW, KS, TPIX>;
  static bool attr_set = false;
  if (!attr_set) {
    SRX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  long ntiles = (long)p.B * p.tiles_x * p.tiles_y;
  if (p.nslices > ntiles) p.nslices = (int)ntiles;
  long grid = (long)d->ntasks * p.nslices;
  SRX_CHECK_ARG(grid > 0 && grid < (1L << 31), "wgrad grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, p);
  return srx_check_launch("wgrad_kernel");
}

template <typename T>
int dispatch(const srx_wgrad_desc* d, const WgradParams& p, hipStream_t s) {
  constexpr bool F = sizeof(T) == 4;
  if (d->ksize == 1) return launch<T, 1, 5, 1, 128>(d, p, s);
  if (d->stride == 1) {
    if (p.OW >= 24) return launch<T, 1, 5, 3, 128>(d, p, s);
    if (p.OW >= 12) return launch<T, 1, 4, 3, 128>(d, p, s);
    return launch<T, 1, 3, 3, 128>(d, p, s);
  }
  if (p.OW >= 24) return launch<T, 2, 5, 3, F ? 32 : 128>(d, p, s);
  if (p.OW >= 12) return launch<T, 2, 4, 3, F ? 32 : 128>(d, p, s);
  return launch<T, 2, 3, 3, F ? 32 : 128>(d, p, s);
}

}  // namespace

extern "C" int srx_conv2d_wgrad(const srx_wgrad_desc* d, void* stream) {
  SRX_CHECK_ARG(d != nullptr, "srx_conv2d_wgrad: null descriptor");
  SRX_CHECK_ARG(d->dtype == SRX_BF16 || d->dtype == SRX_F32, "srx_conv2d_wgrad: bad dtype %d", d->dtype);
  SRX_CHECK_ARG(d->ksize == 3 || d->ksize == 1, "srx_conv2d_wgrad: ksize must be 3 or 1");
  SRX_CHECK_ARG(d->stride == 1 || (d->stride == 2 && d->ksize == 3), "srx_conv2d_wgrad: bad stride %d", d->stride);
  SRX_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->ntasks > 0 && d->nslices > 0,
                "srx_conv2d_wgrad: non-positive size");
  SRX_CHECK_ARG(d->x && d->gy && d->tasks, "srx_conv2d_wgrad: null pointer");
  WgradParams p;
  p.x = d->x; p.x_is = d->x_img_stride;
  p.gy = d->gy; p.gy_is = d->gy_img_stride;
  p.tasks = d->tasks;
  p.H = d->H; p.W = d->W; p.B = d->B;
  const int pad = d->ksize / 2;
  p.OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  p.OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  p.nslices = d->nslices;
  p.tiles_x = p.tiles_y = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == SRX_BF16 ? dispatch<bf16>(d, p, s) : dispatch<float>(d, p, s);
}
