// Two consecutive 3x3 convolutions of a dense block in ONE launch (bf16, gfx950).
//
// In a dense block (reference esrgan/generator.py:17-23) convolution k+1 reads what convolution
// k read plus k's 32 output channels; the data-gradient chain of the block has the same shape.
// Launched one by one, the pair stages the shared input twice and pays the fixed cost of a
// launch (boundary, set-up, epilogue: ~4.5 us against ~3-6 us of multiply at 64x64 patches)
// twice.  Here a workgroup owns a 32x8 output tile of the SECOND convolution B:
//   1. the shared input X (halo 2: 36x12 pixels) streams through a 2-stage LDS-DMA ring, 16
//      channels at a time, together with that chunk's weights of both convolutions; each chunk
//      feeds A's accumulators on the 34x10 region B needs (halo recompute: 11 groups of 32
//      pixels instead of 8) and the X part of B's accumulators;
//   2. A is finished (bias, derivative mask, LeakyReLU, zero outside the image), written to
//      global memory for the pixels this tile owns and, whole, to LDS;
//   3. B's remaining 32 input channels are read from that LDS tile; B is finished and stored.
// Neighbouring workgroups recompute the halo pixels of A with the same arithmetic, so no
// workgroup ever reads another's output: the launch needs no inter-workgroup synchronisation.
#include "srx_common.h"

namespace {

struct PairParams {
  const bf16* x; long x_is;
  bf16* ya; long ya_is;
  bf16* yb; long yb_is;
  const bf16* wa; const bf16* wb;
  const float* bias_a; const float* bias_b;
  const bf16* mask_a; long ma_is;
  const bf16* mask_b; long mb_is;
  float mask_slope, act_slope;
  int H, W, nchunk_a, sa_first;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long a = (unsigned long)base;
  u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}
// LDS-DMA as inline assembly (the ring protocol below orders it against the LDS reads; the
// compiler's own builtin would be drained before every LDS read that might alias it).
// 16 bytes per lane from rsrc + voffset (out of range: zeros) to LDS lds_addr + 16*lane.
__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  // the "s" constraint is not enforced for values the compiler holds in VGPRs: make them scalar
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  const u32x4 r = {(unsigned)__builtin_amdgcn_readfirstlane(rsrc[0]), (unsigned)__builtin_amdgcn_readfirstlane(rsrc[1]),
                   (unsigned)__builtin_amdgcn_readfirstlane(rsrc[2]), (unsigned)__builtin_amdgcn_readfirstlane(rsrc[3])};
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(a), "v"(voffset), "s"(r) : "memory");
}
__device__ __forceinline__ void wait_all_and_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ bool in_img(int H, int W, int y, int x) { return y >= 0 && y < H && x >= 0 && x < W; }
__device__ __forceinline__ void mma(const bf16x8& a, const bf16x8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int TW = 32, TH = 8;
constexpr int AW = TW + 2, AH = TH + 2, NA = AW * AH;        // region of A that B needs: 34 x 10
constexpr int XW = TW + 4, XH = TH + 4, NX = XW * XH;        // input tile: 36 x 12
constexpr int NGA = (NA + 31) / 32;                          // 11 groups of 32 pixels of A
constexpr int XS = 896;                                      // X slots per stage (2*432, padded to 64)
constexpr int WS = 9 * 64;                                   // weight slots per 16-channel chunk: 576
constexpr int STAGE = XS + 2 * WS;                           // 2048 slots = 32 KB
constexpr int NDMA = STAGE / 256;                            // 8 instructions per wave per chunk
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr unsigned WCHUNK_B = 9 * 512 * 2;                   // bytes of one packed 16-channel weight chunk

// One chunk (16 channels of X + that chunk's weights of A and of B) into ring stage `stage`;
// advances the per-lane offsets to the next chunk.  A free function: captured arrays in a
// lambda end up in scratch memory.
__device__ __forceinline__ void pair_issue_chunk(const u32x4& xr, const u32x4& war, const u32x4& wbr,
                                                 unsigned (&off)[NDMA], const unsigned (&cstep)[NDMA],
                                                 unsigned lds0, int stage, int wave_u) {
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int s0 = i * 256 + wave_u * 64;                 // wave-uniform: one region per instruction
    const u32x4 r = s0 < XS ? xr : (s0 < XS + WS ? war : wbr);
    lds_dma16(r, lds0 + (unsigned)((stage * STAGE + s0) * 16), off[i]);
    off[i] = off[i] == OOB ? OOB : off[i] + cstep[i];
  }
}

// B's weights for A's 32 channels (two chunks), the last unit of the ring
__device__ __forceinline__ void pair_issue_tail(const u32x4& wbr, unsigned tail0, unsigned lds0, int stage,
                                                int tid, int wave_u) {
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int s = tid + i * 256;
    lds_dma16(wbr, lds0 + (unsigned)((stage * STAGE + i * 256 + wave_u * 64) * 16),
              s < 2 * WS ? tail0 + (unsigned)s * 16u : OOB);
  }
}

// NSTG ring stages of 32 KB: three when the launch has the CU to itself (two units in flight hide
// the staging latency), two where a weight-gradient workgroup (84 KB) must fit beside it.
template <int NSTG>
__global__ __launch_bounds__(256, 2) void conv_pair_kernel(const PairParams p) {
  constexpr int DIST = NSTG - 1;
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];   // NSTG ring stages
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pl = lane & 31, hh = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b = blockIdx.z;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 - 2, ix0 = ox0 - 2;
  const unsigned plane_b = (unsigned)((long)p.H * p.W * 16);
  const int nca = p.nchunk_a;

  // bias = initial accumulator value (the packer's row order makes a lane's 16 registers the
  // packed floats [hh*16, +16) of the block)
  f32x4 bva[4], bvb[4];
  if (p.bias_a) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bva[k] = *reinterpret_cast<const f32x4*>(p.bias_a + hh * 16 + k * 4);
  }
  if (p.bias_b) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bvb[k] = *reinterpret_cast<const f32x4*>(p.bias_b + hh * 16 + k * 4);
  }

  // ---- staging plan ----
  const bf16* xb = p.x + (long)b * p.x_is;
  const u32x4 xr = make_rsrc(xb, (unsigned)(2 * nca) * plane_b);
  const u32x4 war = make_rsrc(p.wa, (unsigned)nca * WCHUNK_B);
  const u32x4 wbr = make_rsrc(p.wb, (unsigned)(nca + 2) * WCHUNK_B);
  unsigned off[NDMA], cstep[NDMA];     // byte offset of chunk 0, and its growth per chunk
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int s = tid + i * 256;
    if (s < XS) {
      const int h = s / NX, pix = s % NX, r = pix / XW, c = pix % XW;
      const int gy = iy0 + r, gx = ix0 + c;
      const bool ok = s < 2 * NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      off[i] = ok ? (unsigned)h * plane_b + (unsigned)((gy * p.W + gx) * 16) : OOB;
      cstep[i] = ok ? 2u * plane_b : 0u;
    } else if (s < XS + WS) {
      off[i] = (unsigned)(s - XS) * 16u;
      cstep[i] = WCHUNK_B;
    } else {
      off[i] = (unsigned)(s - XS - WS) * 16u + (p.sa_first ? 2u * WCHUNK_B : 0u);
      cstep[i] = WCHUNK_B;
    }
  }
  const unsigned lds0 = (unsigned)(unsigned long)smem;
  // ring units 0 .. nca-1 are the chunks of X, unit nca is the tail (B's weights for A's channels)
  const unsigned tail0 = (p.sa_first ? 0u : (unsigned)nca) * WCHUNK_B;
#pragma unroll
  for (int k = 0; k < DIST; ++k) {
    if (k < nca) pair_issue_chunk(xr, war, wbr, off, cstep, lds0, k, wave_u);
    else if (k == nca) pair_issue_tail(wbr, tail0, lds0, k, tid, wave_u);
  }

  // ---- per-lane fragment addresses (slots) ----
  int xsa[3];                         // A: pixel p = 32*(wave+4*ga)+pl of the 34x10 region
  bool live_a[3];
#pragma unroll
  for (int ga = 0; ga < 3; ++ga) {
    const int pa = 32 * (wave + 4 * ga) + pl;
    live_a[ga] = pa < NA;
    const int pc = pa < NA ? pa : NA - 1;
    xsa[ga] = hh * NX + (pc / AW) * XW + (pc % AW);
  }
  int xsb[2], ssb[2];                 // B: tile row 2*wave+rb, column pl; in X and in the A tile
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int row = 2 * wave + rb;
    xsb[rb] = hh * NX + (row + 1) * XW + (pl + 1);
    ssb[rb] = row * AW + pl;
  }
  const bool third_a = wave + 8 < NGA;           // wave 3 owns only two groups of A

  f32x16 acca[3], accb[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float va = p.bias_a ? bva[i >> 2][i & 3] : 0.f, vb = p.bias_b ? bvb[i >> 2][i & 3] : 0.f;
    acca[0][i] = acca[1][i] = acca[2][i] = va;
    accb[0][i] = accb[1][i] = vb;
  }

  // epilogue operands fetched while the last chunk is multiplied
  const long plane_e = (long)p.H * p.W * 8;      // elements per octet plane
  int pra[3], pca[3];                             // image coordinates of the A pixels
#pragma unroll
  for (int ga = 0; ga < 3; ++ga) {
    const int pa = 32 * (wave + 4 * ga) + pl;
    pra[ga] = oy0 - 1 + pa / AW;
    pca[ga] = ox0 - 1 + pa % AW;
  }
  bf16x8 mka[3][2], mkb[2][2];

  // ---- main loop over the chunks of X ----
  int st = 0, st_new = DIST % NSTG;              // stage of unit c, stage of unit c+DIST
  for (int c = 0; c < nca; ++c) {
    // unit c has landed (with three stages unit c+1 may stay in flight: 8 instructions per wave
    // for a chunk, 5 for the tail) ...
    if (DIST == 2 && c + 1 < nca) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (DIST == 2 && c + 1 == nca) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // ... in every wave, and everyone is done with unit c-1's stage
    if (c + DIST < nca) pair_issue_chunk(xr, war, wbr, off, cstep, lds0, st_new, wave_u);
    else if (c + DIST == nca) pair_issue_tail(wbr, tail0, lds0, st_new, tid, wave_u);
    if (c + 1 == nca) {
      // last chunk: the derivative masks are fetched while it is multiplied
      if (p.mask_a) {
#pragma unroll
        for (int ga = 0; ga < 3; ++ga)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (live_a[ga] && in_img(p.H, p.W, pra[ga], pca[ga]))
              mka[ga][j] = *reinterpret_cast<const bf16x8*>(p.mask_a + (long)b * p.ma_is + (2 * j + hh) * plane_e +
                                                           ((long)pra[ga] * p.W + pca[ga]) * 8);
      }
      if (p.mask_b) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int oy = oy0 + 2 * wave + rb, ox = ox0 + pl;
            if (in_img(p.H, p.W, oy, ox))
              mkb[rb][j] = *reinterpret_cast<const bf16x8*>(p.mask_b + (long)b * p.mb_is + (2 * j + hh) * plane_e +
                                                           ((long)oy * p.W + ox) * 8);
          }
      }
    }
    const bf16x8* sx = reinterpret_cast<const bf16x8*>(smem + st * STAGE);
    const bf16x8* swa = sx + XS;
    const bf16x8* swb = swa + WS;
    // fragments of tap t+1 are read while tap t is multiplied
    bf16x8 fwa[2], fwb[2], fxa[2][3], fxb[2][2];
#define SRX_READ_TAP(tap_, buf_)                                                              \
  do {                                                                                        \
    const int dy_ = (tap_) / 3, dx_ = (tap_) % 3;                                             \
    fwa[buf_] = swa[((tap_) * 2 + hh) * 32 + pl];                                             \
    fwb[buf_] = swb[((tap_) * 2 + hh) * 32 + pl];                                             \
    fxa[buf_][0] = sx[xsa[0] + dy_ * XW + dx_];                                               \
    fxa[buf_][1] = sx[xsa[1] + dy_ * XW + dx_];                                               \
    fxa[buf_][2] = sx[xsa[2] + dy_ * XW + dx_];                                               \
    fxb[buf_][0] = sx[xsb[0] + dy_ * XW + dx_];                                               \
    fxb[buf_][1] = sx[xsb[1] + dy_ * XW + dx_];                                               \
  } while (0)
    SRX_READ_TAP(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < 9) SRX_READ_TAP(tap + 1, cur ^ 1);
      mma(fwa[cur], fxa[cur][0], acca[0]);
      mma(fwa[cur], fxa[cur][1], acca[1]);
      if (third_a) mma(fwa[cur], fxa[cur][2], acca[2]);
      mma(fwb[cur], fxb[cur][0], accb[0]);
      mma(fwb[cur], fxb[cur][1], accb[1]);
    }
#undef SRX_READ_TAP
    st = st + 1 == NSTG ? 0 : st + 1;
    st_new = st_new + 1 == NSTG ? 0 : st_new + 1;
  }

  // ---- finish A: mask, activation, zero outside the image; to LDS (whole region) and to
  //      global memory (the pixels of this tile) ----
  __builtin_amdgcn_s_barrier();                  // every wave is done reading the last stage
  const int stl = (nca - 1) % NSTG, stt = nca % NSTG;      // stage of the last chunk, stage of the tail
  bf16x8* sa_tile = reinterpret_cast<bf16x8*>(smem + stl * STAGE);       // [4 octets][340 pixels]
  const float ms = p.mask_slope, as = p.act_slope;
#pragma unroll
  for (int ga = 0; ga < 3; ++ga) {
    if (ga == 2 && !third_a) continue;
    const int pa = 32 * (wave + 4 * ga) + pl;
    if (pa >= NA) continue;
    const bool inside = in_img(p.H, p.W, pra[ga], pca[ga]);
    const int ra = pa / AW, ca = pa % AW;
    const bool own = inside && ra >= 1 && ra <= TH && ca >= 1 && ca <= TW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = acca[ga][j * 8 + e];
        if (p.mask_a) t = (inside && (float)mka[ga][j][e] > 0.f) ? t : t * ms;
        if (as != 1.f) t = fmaxf(t, t * as);
        v[e] = inside ? t : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      sa_tile[(2 * j + hh) * NA + pa] = o;
      if (own)
        *reinterpret_cast<bf16x8*>(p.ya + (long)b * p.ya_is + (2 * j + hh) * plane_e +
                                   ((long)pra[ga] * p.W + pca[ga]) * 8) = o;
    }
  }
  wait_all_and_barrier();                        // A tile complete; B's tail weights have landed

  // ---- B over A's 32 channels, from LDS ----
  const bf16x8* swt = reinterpret_cast<const bf16x8*>(smem + stt * STAGE);
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int tap = dy * 3 + dx;
        const bf16x8 w = swt[k * WS + (tap * 2 + hh) * 32 + pl];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          mma(w, sa_tile[(2 * k + hh) * NA + ssb[rb] + dy * AW + dx], accb[rb]);
      }

  // ---- finish B ----
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int oy = oy0 + 2 * wave + rb, ox = ox0 + pl;
    if (!in_img(p.H, p.W, oy, ox)) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = accb[rb][j * 8 + e];
        if (p.mask_b) t = (float)mkb[rb][j][e] > 0.f ? t : t * ms;
        if (as != 1.f) t = fmaxf(t, t * as);
        v[e] = t;
      }
      store_oct(p.yb + (long)b * p.yb_is + (2 * j + hh) * plane_e + ((long)oy * p.W + ox) * 8, v);
    }
  }
}

}  // namespace

extern "C" int srx_conv2d_pair(const srx_conv_pair_desc* d, void* stream) {
  SRX_CHECK_ARG(d != nullptr, "srx_conv2d_pair: null descriptor");
  SRX_CHECK_ARG(d->dtype == SRX_BF16, "srx_conv2d_pair: bf16 only (got dtype %d)", d->dtype);
  SRX_CHECK_ARG(d->B > 0 && d->B <= 65535 && d->H > 0 && d->W > 0, "srx_conv2d_pair: bad size B=%d H=%d W=%d",
                d->B, d->H, d->W);
  SRX_CHECK_ARG(d->cin_a > 0 && d->cin_a % 16 == 0, "srx_conv2d_pair: cin_a must be a positive multiple of 16 (got %d)",
                d->cin_a);
  SRX_CHECK_ARG(d->x && d->ya && d->yb && d->wa && d->wb, "srx_conv2d_pair: null x/ya/yb/wa/wb pointer");
  SRX_CHECK_ARG(d->act_slope >= 0.f && d->act_slope <= 1.f, "srx_conv2d_pair: act_slope must be in [0,1]");
  SRX_CHECK_ARG(((uintptr_t)d->x | (uintptr_t)d->ya | (uintptr_t)d->yb | (uintptr_t)d->wa | (uintptr_t)d->wb) % 16 == 0,
                "srx_conv2d_pair: pointers must be 16-byte aligned");
  SRX_CHECK_ARG((long)d->H * d->W * 16 * (d->cin_a / 8) < (1L << 32), "srx_conv2d_pair: image too large");
  PairParams p;
  p.x = (const bf16*)d->x; p.x_is = d->x_img_stride;
  p.ya = (bf16*)d->ya; p.ya_is = d->ya_img_stride;
  p.yb = (bf16*)d->yb; p.yb_is = d->yb_img_stride;
  p.wa = (const bf16*)d->wa; p.wb = (const bf16*)d->wb;
  p.bias_a = d->bias_a; p.bias_b = d->bias_b;
  p.mask_a = (const bf16*)d->mask_a; p.ma_is = d->mask_a_img_stride;
  p.mask_b = (const bf16*)d->mask_b; p.mb_is = d->mask_b_img_stride;
  p.mask_slope = d->mask_slope; p.act_slope = d->act_slope;
  p.H = d->H; p.W = d->W; p.nchunk_a = d->cin_a / 16; p.sa_first = d->a_out_first;
  const int tiles_y = (d->H + TH - 1) / TH;
  SRX_CHECK_ARG(tiles_y <= 65535, "srx_conv2d_pair: image too tall");
  const dim3 grid((d->W + TW - 1) / TW, tiles_y, d->B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // The data-gradient form (masks) is the one that runs beside weight-gradient workgroups.
  if (d->mask_a || d->mask_b) {
    constexpr int LDS = 2 * STAGE * 16;
    static bool attr_set = false;
    if (!attr_set) {
      SRX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_kernel<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_set = true;
    }
    hipLaunchKernelGGL(conv_pair_kernel<2>, grid, dim3(256), LDS, s, p);
  } else {
    constexpr int LDS = 3 * STAGE * 16;
    static bool attr_set = false;
    if (!attr_set) {
      SRX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pair_kernel<3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_set = true;
    }
    hipLaunchKernelGGL(conv_pair_kernel<3>, grid, dim3(256), LDS, s, p);
  }
  return srx_check_launch("conv_pair_kernel");
}
