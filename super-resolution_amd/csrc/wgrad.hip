// Weight (and bias) gradient of a 3x3 / 1x1 convolution on 32x32 MFMA tiles (gfx950).
//
// GEMM view per tap:  dW[tap][cout][cin] = sum_pixels GY[cout][p] * X[cin][p + tap],
// i.e. the contraction index is the PIXEL.  Activations are pixel-major (C8
// layout), so both operands are transposed on the way out of LDS with
// ds_read_b64_tr_b16 (bf16) or gathered with ds_read_b32 (fp32).
//
// One workgroup owns a 64(cin) x 64(cout) block of one or several layers (a
// task) and a slice of the pixel tiles; its 4 waves each hold the 9-tap
// accumulators of one 32x32 sub-block (144 registers) across all tiles of the
// slice, then add them with fp32 atomics into tap-major gradient tensors
// ([tap][cout][cin]: the 32 lanes of a half-wave are 32 consecutive cin, i.e.
// 128 contiguous bytes per atomic instruction half).  The bias gradient is one more MFMA
// per k-step against an all-ones operand, in the wave that owns that bias slice.
//
// Pipeline: LDS-DMA double buffering (tile t+1 streams global->LDS behind tile t's
// MFMAs without touching VGPRs; one barrier per tile).  For stride 1 the X
// fragments of an input row are read once and used by the three taps dy that map
// it to an output row.  Octet planes in LDS are padded to 4 (mod 8) slots so the
// transposed reads of a 32-lane half fall on 32 distinct 8-byte banks.
#include <stdlib.h>
#include "srx_common.h"

namespace {

struct WgradParams {
  const void* x; long x_is;
  const void* gy; long gy_is;
  const srx_wgrad_task* tasks;
  int H, W, OH, OW, B;
  int nslices, tiles_x, tiles_y;
  int group, ntasks;     // tasks per group (share x/gy buffers) for XCD-local ordering; 0 = off
  long slice_stride;     // > 0: slice s stores its partial sums s * slice_stride floats further on
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
__device__ __forceinline__ bf16x8 tr_frag(const char* lds_base, int addr0, int addr1) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + addr0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_base + addr1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int pad4mod8(int n) { return n + ((4 - n % 8) + 8) % 8; }

// LDS-DMA issued as inline assembly.  With the compiler's own builtin, its wait-count pass
// treats every later ds_read_b64_tr_b16 as possibly reading what the DMA writes and puts an
// s_waitcnt vmcnt(0) in front of the first one: the next tile's staging would then complete
// before this tile's multiply starts, i.e. no overlap at all.  The ring protocol (which stage is
// written, which is read, the explicit vmcnt(0) + barrier at the end of a tile) carries the
// ordering instead.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long a = (unsigned long)base;
  u32x4 r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}
// 16 bytes per lane from rsrc + voffset (out of range: zeros) to LDS byte address lds_addr + 16*lane;
// lds_addr must be wave-uniform.
__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voffset) {
  // the "s" constraint is not enforced for values the compiler holds in VGPRs: make them scalar
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  const u32x4 r = {(unsigned)__builtin_amdgcn_readfirstlane(rsrc[0]), (unsigned)__builtin_amdgcn_readfirstlane(rsrc[1]),
                   (unsigned)__builtin_amdgcn_readfirstlane(rsrc[2]), (unsigned)__builtin_amdgcn_readfirstlane(rsrc[3])};
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(a), "v"(voffset), "s"(r) : "memory");
}

// The accumulators of one wave (9-tap 32x32 sub-block + bias column sums) into the task's
// gradient tensors: fp32 atomic adds, or plain stores into the slice's own copy (deterministic form).
template <int TAPS>
__device__ __forceinline__ void wgrad_write_out(const WgradParams& p, const srx_wgrad_task* __restrict__ tkp,
                                                int slice, int lane, int sa, int sb, float* outp, float* biasp,
                                                const f32x16 (&acc)[TAPS], const f32x16& accb) {
  const int col = lane & 31, hh = lane >> 5;
  // deterministic form: every (task, slice) owns its own copy of the elements it produces, so a
  // plain store replaces the atomic add and nothing depends on the order workgroups finish in
  const bool det = p.slice_stride > 0;
  const long sofs = det ? (long)slice * p.slice_stride : 0;
  if (biasp != nullptr && col == 0) {
    const int bn = tkp->bias_n[sb], b0 = tkp->bias_co0[sb];
    const float bs = tkp->bias_scale[sb];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = b0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (co < bn) {
        if (det) biasp[sofs + co] = accb[i] * bs;
        else atomicAdd(biasp + co, accb[i] * bs);
      }
    }
  }
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
      if (co < co_n) {
        float* const o = outp + sofs + ((long)tap * co_n + co) * ld + ci;
        if (det) *o = acc[tap][i] * scale;
        else atomicAdd(o, acc[tap][i] * scale);
      }
    }
  }
}

// Stage one tile (X halo tile + GY tile) into LDS stage `ls` by LDS-DMA.  A free function
// rather than a lambda: hipcc (ROCm 7.2) drops the host-side kernel stub when a by-reference
// captured array feeds this builtin's offset operand.
template <typename T, int STRIDE, int LTW, int KS, int TPIX, int NXS, int NGS, int XSL, int GSL>
__device__ __forceinline__ void wgrad_stage_tile(const WgradParams& p, const T* xbase, const T* gbase,
                                                 int tile, int tiles_per_img, char* ls, int tid, int wave_u,
                                                 int xbytes, int gbytes, const unsigned (&xoffs)[NXS],
                                                 const int (&xrc)[NXS], const unsigned (&goffs)[NGS],
                                                 const int (&grc)[NGS]) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  constexpr int TW = 1 << LTW, TH = TPIX >> LTW, PAD = KS / 2;

    const int b = tile / tiles_per_img;
    const int tr = tile % tiles_per_img;
    const int oy0 = (tr / p.tiles_x) * TH, ox0 = (tr % p.tiles_x) * TW;
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
    // descriptors based at the tile's first input / output pixel (may lie before the plane for
    // border tiles: those lanes are out of range by construction and never dereferenced)
    const long xoff_b = ((long)iy0 * p.W + ix0) * 8 * (long)sizeof(T);
    const long goff_b = ((long)oy0 * p.OW + ox0) * 8 * (long)sizeof(T);
    const u32x4 xrs = make_rsrc(reinterpret_cast<const char*>(xbase + (long)b * p.x_is) + xoff_b,
                                (unsigned)(xbytes - xoff_b));
    const u32x4 grs = make_rsrc(reinterpret_cast<const char*>(gbase + (long)b * p.gy_is) + goff_b,
                                (unsigned)(gbytes - goff_b));
    const unsigned lds0 = (unsigned)(unsigned long)ls;      // low 32 bits of a flat LDS address = LDS offset
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      const unsigned gy_ = (unsigned)(iy0 + (xrc[i] & 0xffff)), gx_ = (unsigned)(ix0 + (xrc[i] >> 16));
      const bool ok = xrc[i] >= 0 && gy_ < (unsigned)p.H && gx_ < (unsigned)p.W;
      if (NXS * 256 == XSL || tid + i * 256 < XSL)
        lds_dma16(xrs, lds0 + (i * 256 + wave_u * 64) * 16, ok ? xoffs[i] : OOB);
    }
    const unsigned lg0 = lds0 + XSL * 16;
#pragma unroll
    for (int i = 0; i < NGS; ++i) {
      const unsigned oy = (unsigned)(oy0 + (grc[i] & 0xffff)), ox = (unsigned)(ox0 + (grc[i] >> 16));
      const bool ok = grc[i] >= 0 && oy < (unsigned)p.OH && ox < (unsigned)p.OW;
      if (NGS * 256 == GSL || tid + i * 256 < GSL)
        lds_dma16(grs, lg0 + (i * 256 + wave_u * 64) * 16, ok ? goffs[i] : OOB);
    }
  }

template <typename T, int STRIDE, int LTW, int KS, int TPIX>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  typedef typename Frag<T>::type frag_t;
  constexpr int TW = 1 << LTW;
  constexpr int TH = TPIX >> LTW;
  constexpr int TAPS = KS * KS;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int IW = (TW - 1) * STRIDE + KS;
  constexpr int NPX = IH * IW;
  constexpr int PSX = pad4mod8(NPX);        // LDS octet-plane stride (slots) of the X tile
  constexpr int PSG = pad4mod8(TPIX);       // ... of the GY tile
  constexpr int PPO = Oct<T>::PPO;
  constexpr int EPP = 16 / sizeof(T);
  constexpr int KSTEPS = TPIX / 16;
  constexpr int OB = sizeof(T) * 8;         // bytes per octet slot
  constexpr int STAGE = (8 * PSX + 8 * PSG) * PPO;   // 16-byte pieces per stage
  constexpr bool ROWREUSE = (STRIDE == 1 && LTW >= 4 && KS == 3);

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Block -> (task, slice).  With groups, the `group` tasks that read the same buffers for the
  // same pixel slice get block ids that are equal mod 8 and adjacent in time, so (under the
  // observed round-robin placement) they run on one XCD and share its L2: each byte of X / GY
  // is then fetched from HBM/MALL about once instead of once per task.  Placement is a speed
  // matter only; any mapping that is a bijection is correct.
  int task_id, slice;
  if (p.group > 0) {
    const int gsz = p.group;
    const int ngrp = (p.ntasks / gsz) * p.nslices;           // (buffer set, slice) pairs
    const int super_ = blockIdx.x / (8 * gsz), r = blockIdx.x % (8 * gsz);
    const int g = super_ * 8 + (r % 8), t = r / 8;
    if (g >= ngrp) return;
    task_id = (g / p.nslices) * gsz + t;
    slice = g % p.nslices;
  } else {
    task_id = blockIdx.x / p.nslices;
    slice = blockIdx.x % p.nslices;
  }
  const srx_wgrad_task* __restrict__ tkp = p.tasks + task_id;
  const int sa = wave & 1, sb = wave >> 1;      // cin half, cout half
  float* const outp = tkp->out[sa + 2 * sb];
  float* const biasp = sa == 0 ? tkp->bias_out[sb] : nullptr;
  const int x_noct = tkp->x_noct, gy_noct = tkp->gy_noct;
  const int x_skip = tkp->x_skip, gy_skip = tkp->gy_skip;      // planes between octets 3 and 4 of the task
  const bool active = outp != nullptr;          // wave-uniform

  const long plane_x = (long)p.H * p.W * 8;
  const long plane_g = (long)p.OH * p.OW * 8;
  const T* xbase = reinterpret_cast<const T*>(tkp->x ? tkp->x : p.x) + (long)tkp->x_oct * plane_x;
  const T* gbase = reinterpret_cast<const T*>(tkp->gy ? tkp->gy : p.gy) + (long)tkp->gy_oct * plane_g;

  // ---- per-lane LDS read addresses (bytes), tap (0,0), k-step 0 ----
  int ga[2], xa[2];      // bf16: two transposed reads; fp32: [0] only (element address)
  if constexpr (sizeof(T) == 2) {
    const int gidx = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;
    const int h = gidx >> 1;
    const int osub = 2 * (gidx & 1) + (pq >> 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kp = 8 * h + 4 * t + q;        // pixel inside the 16-pixel k-step
      ga[t] = ((4 * sb + osub) * PSG + kp) * OB + 8 * (pq & 1);
      const int py = kp >> LTW, px = kp & (TW - 1);   // for k-step 0; later steps add a constant
      xa[t] = ((4 * sa + osub) * PSX + py * STRIDE * IW + px * STRIDE) * OB + 8 * (pq & 1);
    }
  } else {
    const int r = lane & 31, h = lane >> 5;
    ga[0] = ((4 * sb + (r >> 3)) * PSG + h) * OB + (r & 7) * 4;
    xa[0] = ((4 * sa + (r >> 3)) * PSX + h * STRIDE) * OB + (r & 7) * 4;   // TW >= 2
    ga[1] = xa[1] = 0;
  }

  auto load_g = [&](const char* sgb, int ks) -> frag_t {
    const int gofs = ks * 16 * OB;
    frag_t gf;
    if constexpr (sizeof(T) == 2) {
      gf = tr_frag(sgb, ga[0] + gofs, ga[1] + gofs);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) gf[s] = *reinterpret_cast<const float*>(sgb + ga[0] + gofs + 2 * s * OB);
    }
    return gf;
  };
  // X fragment whose 16 pixels start at tile-input position (row, col) [slots]
  auto load_x = [&](const char* sxb, int row, int col) -> frag_t {
    const int xofs = (row * IW + col) * OB;
    frag_t xf;
    if constexpr (sizeof(T) == 2) {
      xf = tr_frag(sxb, xa[0] + xofs, xa[1] + xofs);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        int po;
        if constexpr (LTW >= 4) po = 2 * s * STRIDE * OB;
        else po = (((2 * s) >> 3) * STRIDE * IW + ((2 * s) & 7) * STRIDE) * OB;
        xf[s] = *reinterpret_cast<const float*>(sxb + xa[0] + xofs + po);
      }
    }
    return xf;
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  // bias gradient: in the one wave per cout half that owns it, one extra MFMA per k-step
  // against an all-ones operand (every column of the result is the per-channel pixel sum)
  f32x16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
  frag_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (T)1.0f;

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int ntiles = p.B * tiles_per_img;

  // ---- staging: LDS-DMA through a buffer descriptor (buffer_load ... lds, 16 B per lane).
  // The LDS image of a wave instruction is lane-linear, so threads own consecutive 16-byte
  // slots of the padded tile image; the per-lane SOURCE offset carries the layout.  A lane
  // whose offset is out of the descriptor's range deposits zeros in LDS (measured on gfx950),
  // which serves the zero padding, octets beyond the task's valid range and the pad slots:
  // no load is conditional and nothing passes through VGPRs.
  constexpr int XSL = 8 * PSX * PPO, GSL = 8 * PSG * PPO;       // slots per stage (incl. pads)
  constexpr int NXS = (XSL + 255) / 256, NGS = (GSL + 255) / 256;
  const unsigned xplane_b = (unsigned)(plane_x * sizeof(T)), gplane_b = (unsigned)(plane_g * sizeof(T));
  const int xbytes = (int)((unsigned)(x_noct + (x_noct > 4 ? x_skip : 0)) * xplane_b);
  const int gbytes = (int)((unsigned)(gy_noct + (gy_noct > 4 ? gy_skip : 0)) * gplane_b);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // per-slot constants: byte offset from the tile's first input pixel, and (row, col) for the
  // bounds test; rc < 0 marks a slot that always reads zeros
  unsigned xoffs[NXS], goffs[NGS];
  int xrc[NXS], grc[NGS];
#pragma unroll
  for (int i = 0; i < NXS; ++i) {
    int slot = tid + i * 256;
    int q = slot % PPO, pix = (slot / PPO) % PSX, o = slot / (PPO * PSX);
    int r = pix / IW, c = pix % IW;
    bool live = slot < XSL && pix < NPX && o < x_noct;
    xoffs[i] = (unsigned)(o + (o >= 4 ? x_skip : 0)) * xplane_b + (unsigned)(((r * p.W + c) * 8 + q * EPP) * (int)sizeof(T));
    xrc[i] = live ? (r | (c << 16)) : -1;
  }
#pragma unroll
  for (int i = 0; i < NGS; ++i) {
    int slot = tid + i * 256;
    int q = slot % PPO, pix = (slot / PPO) % PSG, o = slot / (PPO * PSG);
    int r = pix >> LTW, c = pix & (TW - 1);
    bool live = slot < GSL && pix < TPIX && o < gy_noct;
    goffs[i] = (unsigned)(o + (o >= 4 ? gy_skip : 0)) * gplane_b + (unsigned)(((r * p.OW + c) * 8 + q * EPP) * (int)sizeof(T));
    grc[i] = live ? (r | (c << 16)) : -1;
  }
  // tiles of this slice: a contiguous range, walked with (image, tile row, tile col) counters
  const int per_slice = (ntiles + p.nslices - 1) / p.nslices;
  const int t_begin = slice * per_slice;
  const int t_end = t_begin + per_slice < ntiles ? t_begin + per_slice : ntiles;
  auto stage_tile = [&](int tile, int stage) {
    wgrad_stage_tile<T, STRIDE, LTW, KS, TPIX, NXS, NGS, XSL, GSL>(
        p, xbase, gbase, tile, tiles_per_img, reinterpret_cast<char*>(smem + stage * STAGE), tid, wave_u,
        xbytes, gbytes, xoffs, xrc, goffs, grc);
  };

  int tile = t_begin;
  if (tile < t_end) stage_tile(tile, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int stage = 0;
  for (; tile < t_end; ++tile) {
    const bool more = tile + 1 < t_end;
    if (more) stage_tile(tile + 1, stage ^ 1);
    const char* sxb = reinterpret_cast<const char*>(smem + stage * STAGE);
    const char* sgb = sxb + 8 * PSX * PPO * 16;
    if (active) {
      if constexpr (ROWREUSE) {
        constexpr int NC = TW / 16;           // k-steps per tile row
        frag_t gf[TH][NC];
#pragma unroll
        for (int y = 0; y < TH; ++y)
#pragma unroll
          for (int c = 0; c < NC; ++c) gf[y][c] = load_g(sgb, y * NC + c);
        if (biasp != nullptr) {               // wave-uniform: one wave in ~6 owns a bias slice
#pragma unroll
          for (int y = 0; y < TH; ++y)
#pragma unroll
            for (int c = 0; c < NC; ++c) mma(gf[y][c], ones, accb);
        }
#pragma unroll
        for (int ry = 0; ry < IH; ++ry)
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            frag_t xf[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) xf[dx] = load_x(sxb, ry, c * 16 + dx);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int y = ry - dy;
              if (y >= 0 && y < TH) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) mma(gf[y][c], xf[dx], acc[dy * 3 + dx]);
              }
            }
          }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const int tp0 = ks * 16;
          int row0, col0;
          if constexpr (LTW >= 4) { row0 = (tp0 >> LTW) * STRIDE; col0 = (tp0 & (TW - 1)) * STRIDE; }
          else { row0 = 2 * ks * STRIDE; col0 = 0; }     // TW == 8: a k-step is two tile rows
          frag_t gf = load_g(sgb, ks);
          if (biasp != nullptr) mma(gf, ones, accb);
#pragma unroll
          for (int tap = 0; tap < TAPS; ++tap) {
            frag_t xf = load_x(sxb, row0 + tap / KS, col0 + tap % KS);
            mma(gf, xf, acc[tap]);
          }
        }
      }
    }
    // the next tile has landed (every wave's share of it) and every wave is done reading this one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }

  if (!active) return;
  wgrad_write_out<TAPS>(p, tkp, slice, lane, sa, sb, outp, biasp, acc, accb);
}


// ---------------------------------------------------------------------------------------------
// Column walk: the 3x3 / stride-1 form for images at least 24 pixels wide.
//
// wgrad_kernel stages a whole halo tile per 128-pixel tile: (TH + 2) x 34 input pixels for TH x 32
// output pixels, of which two rows were already staged for the tile above.  Here a workgroup walks
// its tiles DOWN the image (tile order: image, tile column, tile row) and keeps the input rows in a
// ring of three row groups of TH rows each: group k of a column holds input rows k*TH-1 ..
// k*TH+TH-2; tile j multiplies with group j and the first two rows of group j+1, while group j+2
// streams in.  Per tile one group (TH rows) is staged instead of TH+2 rows — a third less
// activation traffic and staging instructions for TH = 4 — and the gradient tile goes in whole
// 64-slot runs (no pad slots): 8.4 LDS-DMA instructions per wave and tile instead of 12.
// A column costs one extra step (its first group has nothing to be multiplied with yet).
template <typename T, int TPIX>
__global__ __launch_bounds__(256) void wgrad_walk_kernel(const WgradParams p) {
  typedef typename Frag<T>::type frag_t;
  constexpr int TW = 32, TH = TPIX / 32, TAPS = 9;
  constexpr int IH = TH + 2, IW = TW + 2;
  constexpr int NPG = TH * IW;              // pixels of one row group
  constexpr int PG = pad4mod8(NPG);         // LDS octet-plane stride (slots) inside a group
  constexpr int PSG = pad4mod8(TPIX);       // ... of the GY tile
  constexpr int PPO = Oct<T>::PPO;
  constexpr int EPP = 16 / sizeof(T);
  constexpr int OB = sizeof(T) * 8;         // bytes per octet slot
  constexpr int XG = 8 * PG * PPO;          // 16-byte pieces per row group (incl. pads)
  constexpr int GST = 8 * PSG * PPO;        // ... per GY stage
  constexpr int NXS = (XG + 255) / 256;
  constexpr int RPO = TPIX * PPO / 64;      // 64-piece runs per GY octet
  constexpr int NGR = 8 * RPO / 4;          // GY runs per wave
  static_assert(TH >= 2 && TPIX * PPO % 64 == 0 && (8 * RPO) % 4 == 0, "tile shape");
  constexpr unsigned OOB = 0xFFFFFFF0u;

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [3 groups][2 GY stages]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int task_id, slice;
  if (p.group > 0) {
    const int gsz = p.group;
    const int ngrp = (p.ntasks / gsz) * p.nslices;
    const int super_ = blockIdx.x / (8 * gsz), r = blockIdx.x % (8 * gsz);
    const int g = super_ * 8 + (r % 8), t = r / 8;
    if (g >= ngrp) return;
    task_id = (g / p.nslices) * gsz + t;
    slice = g % p.nslices;
  } else {
    task_id = blockIdx.x / p.nslices;
    slice = blockIdx.x % p.nslices;
  }
  const srx_wgrad_task* __restrict__ tkp = p.tasks + task_id;
  const int sa = wave & 1, sb = wave >> 1;      // cin half, cout half
  float* const outp = tkp->out[sa + 2 * sb];
  float* const biasp = sa == 0 ? tkp->bias_out[sb] : nullptr;
  const int x_noct = tkp->x_noct, gy_noct = tkp->gy_noct;
  const int x_skip = tkp->x_skip, gy_skip = tkp->gy_skip;
  const bool active = outp != nullptr;          // wave-uniform

  const long plane_x = (long)p.H * p.W * 8;
  const long plane_g = (long)p.OH * p.OW * 8;
  const T* xbase = reinterpret_cast<const T*>(tkp->x ? tkp->x : p.x) + (long)tkp->x_oct * plane_x;
  const T* gbase = reinterpret_cast<const T*>(tkp->gy ? tkp->gy : p.gy) + (long)tkp->gy_oct * plane_g;

  // ---- per-lane LDS read addresses (bytes) relative to a group / GY stage, tap (0,0), k-step 0 ----
  int ga[2], xa[2];
  if constexpr (sizeof(T) == 2) {
    const int gidx = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;
    const int h = gidx >> 1;
    const int osub = 2 * (gidx & 1) + (pq >> 1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int kp = 8 * h + 4 * t + q;        // pixel inside the 16-pixel k-step (one tile row)
      ga[t] = ((4 * sb + osub) * PSG + kp) * OB + 8 * (pq & 1);
      xa[t] = ((4 * sa + osub) * PG + kp) * OB + 8 * (pq & 1);
    }
  } else {
    const int r = lane & 31, h = lane >> 5;
    ga[0] = ((4 * sb + (r >> 3)) * PSG + h) * OB + (r & 7) * 4;
    xa[0] = ((4 * sa + (r >> 3)) * PG + h) * OB + (r & 7) * 4;
    ga[1] = xa[1] = 0;
  }
  auto load_g = [&](const char* sgb, int ks) -> frag_t {
    const int gofs = ks * 16 * OB;
    frag_t gf;
    if constexpr (sizeof(T) == 2) {
      gf = tr_frag(sgb, ga[0] + gofs, ga[1] + gofs);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) gf[s] = *reinterpret_cast<const float*>(sgb + ga[0] + gofs + 2 * s * OB);
    }
    return gf;
  };
  // X fragment whose 16 pixels start at (row, col) of the row group at sxg
  auto load_x = [&](const char* sxg, int row, int col) -> frag_t {
    const int xofs = (row * IW + col) * OB;
    frag_t xf;
    if constexpr (sizeof(T) == 2) {
      xf = tr_frag(sxg, xa[0] + xofs, xa[1] + xofs);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) xf[s] = *reinterpret_cast<const float*>(sxg + xa[0] + xofs + 2 * s * OB);
    }
    return xf;
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  f32x16 accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) accb[i] = 0.f;
  frag_t ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (T)1.0f;

  // ---- staging constants ----
  const unsigned xplane_b = (unsigned)(plane_x * sizeof(T)), gplane_b = (unsigned)(plane_g * sizeof(T));
  const int xbytes = (int)((unsigned)(x_noct + (x_noct > 4 ? x_skip : 0)) * xplane_b);
  const int gbytes = (int)((unsigned)(gy_noct + (gy_noct > 4 ? gy_skip : 0)) * gplane_b);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned xoffs[NXS];
  int xrc[NXS];
#pragma unroll
  for (int i = 0; i < NXS; ++i) {
    const int slot = tid + i * 256;
    const int q = slot % PPO, pix = (slot / PPO) % PG, o = slot / (PPO * PG);
    const int r = pix / IW, c = pix % IW;
    const bool live = slot < XG && pix < NPG && o < x_noct;
    xoffs[i] = (unsigned)(o + (o >= 4 ? x_skip : 0)) * xplane_b + (unsigned)(((r * p.W + c) * 8 + q * EPP) * (int)sizeof(T));
    xrc[i] = live ? (r | (c << 16)) : -1;
  }
  // GY: wave w stages the runs w, w+4, ...: run = (octet, 64-piece part of its TPIX*PPO pieces)
  unsigned goffs[NGR];
  int grc[NGR];
#pragma unroll
  for (int i = 0; i < NGR; ++i) {
    const int run = wave + 4 * i;
    const int o = run / RPO, piece = (run % RPO) * 64 + lane;
    const int q = piece % PPO, pix = piece / PPO;
    const int r = pix >> 5, c = pix & 31;
    goffs[i] = (unsigned)(o + (o >= 4 ? gy_skip : 0)) * gplane_b + (unsigned)(((r * p.OW + c) * 8 + q * EPP) * (int)sizeof(T));
    grc[i] = o < gy_noct ? (r | (c << 16)) : -1;
  }
  const unsigned lds_x0 = (unsigned)(unsigned long)smem;
  const unsigned lds_g0 = lds_x0 + 3 * XG * 16;

  // row group k of tile column (b, tx) -> ring slot
  auto stage_group = [&](int b, int tx, int k, int ring) {
    const int iy0 = k * TH - 1, ix0 = tx * TW - 1;
    const long xoff_b = ((long)iy0 * p.W + ix0) * 8 * (long)sizeof(T);
    const u32x4 xrs = make_rsrc(reinterpret_cast<const char*>(xbase + (long)b * p.x_is) + xoff_b,
                                (unsigned)(xbytes - xoff_b));
    const unsigned l0 = lds_x0 + ring * (XG * 16);
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      const unsigned gy_ = (unsigned)(iy0 + (xrc[i] & 0xffff)), gx_ = (unsigned)(ix0 + (xrc[i] >> 16));
      const bool ok = xrc[i] >= 0 && gy_ < (unsigned)p.H && gx_ < (unsigned)p.W;
      if (NXS * 256 == XG || tid + i * 256 < XG)
        lds_dma16(xrs, l0 + (i * 256 + wave_u * 64) * 16, ok ? xoffs[i] : OOB);
    }
  };
  auto stage_gy = [&](int b, int tx, int ty, int gst) {
    const int oy0 = ty * TH, ox0 = tx * TW;
    const long goff_b = ((long)oy0 * p.OW + ox0) * 8 * (long)sizeof(T);
    const u32x4 grs = make_rsrc(reinterpret_cast<const char*>(gbase + (long)b * p.gy_is) + goff_b,
                                (unsigned)(gbytes - goff_b));
    const unsigned l0 = lds_g0 + gst * (GST * 16);
#pragma unroll
    for (int i = 0; i < NGR; ++i) {
      const int run = wave_u + 4 * i;
      const unsigned oy = (unsigned)(oy0 + (grc[i] & 0xffff)), ox = (unsigned)(ox0 + (grc[i] >> 16));
      const bool ok = grc[i] >= 0 && oy < (unsigned)p.OH && ox < (unsigned)p.OW;
      lds_dma16(grs, l0 + ((run / RPO) * PSG * PPO + (run % RPO) * 64) * 16, ok ? goffs[i] : OOB);
    }
  };
  // tile j of a column: row group ring slots ra (group j) and rb (group j+1), GY stage gst
  auto compute = [&](int ra, int rb, int gst) {
    const char* sga = reinterpret_cast<const char*>(smem) + ra * (XG * 16);
    const char* sgb_x = reinterpret_cast<const char*>(smem) + rb * (XG * 16);
    const char* sgy = reinterpret_cast<const char*>(smem) + 3 * XG * 16 + gst * (GST * 16);
    constexpr int NC = TW / 16;           // k-steps per tile row
    frag_t gf[TH][NC];
#pragma unroll
    for (int y = 0; y < TH; ++y)
#pragma unroll
      for (int c = 0; c < NC; ++c) gf[y][c] = load_g(sgy, y * NC + c);
    if (biasp != nullptr) {
#pragma unroll
      for (int y = 0; y < TH; ++y)
#pragma unroll
        for (int c = 0; c < NC; ++c) mma(gf[y][c], ones, accb);
    }
#pragma unroll
    for (int ry = 0; ry < IH; ++ry)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        frag_t xf[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          xf[dx] = ry < TH ? load_x(sga, ry, c * 16 + dx) : load_x(sgb_x, ry - TH, c * 16 + dx);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int y = ry - dy;
          if (y >= 0 && y < TH) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) mma(gf[y][c], xf[dx], acc[dy * 3 + dx]);
          }
        }
      }
  };

  // ---- the walk: tiles of this slice in (image, tile column, tile row) order ----
  const int ncols = p.B * p.tiles_x;
  const int ntiles = ncols * p.tiles_y;
  const int per_slice = (ntiles + p.nslices - 1) / p.nslices;
  int t = slice * per_slice;
  const int t_end = t + per_slice < ntiles ? t + per_slice : ntiles;
  int ring = 0;                              // ring slot of the group staged next
  if (t < t_end) {
    const int col = t / p.tiles_y;
    stage_group(col / p.tiles_x, col % p.tiles_x, t % p.tiles_y, ring);
  }
  while (t < t_end) {
    const int col = t / p.tiles_y, ty0 = t % p.tiles_y;
    const int b = col / p.tiles_x, tx = col % p.tiles_x;
    const int left = t_end - t;
    const int n = p.tiles_y - ty0 < left ? p.tiles_y - ty0 : left;     // tiles of this column in the slice
    // step j: group ty0+j has landed; stage group ty0+j+1 and the gradient tile of tile j; multiply
    // tile j-1.  The last step stages the first group of the slice's next column instead.
    // (step 0, which has nothing to multiply yet, stands in front of the loop: with the multiply
    // under a condition on j the accumulators pass through copies at every step)
    {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int nxt = ring == 2 ? 0 : ring + 1;
      stage_group(b, tx, ty0 + 1, nxt);
      stage_gy(b, tx, ty0, 0);
      ring = nxt;
    }
    for (int j = 1; j <= n; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // landed in every wave; every wave is done with the slot staged next
      const int nxt = ring == 2 ? 0 : ring + 1;
      if (j < n) {
        stage_group(b, tx, ty0 + j + 1, nxt);
        stage_gy(b, tx, ty0 + j, j & 1);
      } else if (t + n < t_end) {
        const int col2 = (t + n) / p.tiles_y;
        stage_group(col2 / p.tiles_x, col2 % p.tiles_x, (t + n) % p.tiles_y, nxt);
      }
      if (active) compute(ring == 0 ? 2 : ring - 1, ring, (j - 1) & 1);
      ring = nxt;
    }
    t += n;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!active) return;
  wgrad_write_out<TAPS>(p, tkp, slice, lane, sa, sb, outp, biasp, acc, accb);
}

template <typename T, int TPIX>
int launch_walk(const srx_wgrad_desc* d, WgradParams p, hipStream_t stream) {
  constexpr int TH = TPIX / 32;
  constexpr int PPO = Oct<T>::PPO;
  constexpr int LDS = (3 * 8 * pad4mod8(TH * 34) + 2 * 8 * pad4mod8(TPIX)) * PPO * 16;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  p.tiles_x = (p.OW + 31) / 32;
  p.tiles_y = (p.OH + TH - 1) / TH;
  auto kern = wgrad_walk_kernel<T, TPIX>;
  static srx_per_device_once attr_set;     // per instantiation, per device
  if (int rc = srx_ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(kern), LDS)) return rc;
  long ntiles = (long)p.B * p.tiles_x * p.tiles_y;
  if (p.nslices > ntiles) p.nslices = (int)ntiles;
  long grid = (long)d->ntasks * p.nslices;
  p.group = 0; p.ntasks = d->ntasks;
  if (d->group > 1 && d->ntasks % d->group == 0) {
    p.group = d->group;
    long ngrp = (long)(d->ntasks / d->group) * p.nslices;
    grid = ((ngrp + 7) / 8) * 8 * d->group;
  }
  SRX_CHECK_ARG(grid > 0 && grid < (1L << 31), "wgrad grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, p);
  return srx_check_launch("wgrad_walk_kernel");
}

template <typename T, int STRIDE, int LTW, int KS, int TPIX>
int launch(const srx_wgrad_desc* d, WgradParams p, hipStream_t stream) {
  constexpr int TW = 1 << LTW, TH = TPIX >> LTW;
  constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
  constexpr int PPO = Oct<T>::PPO;
  constexpr int LDS = 2 * (8 * pad4mod8(IH * IW) + 8 * pad4mod8(TPIX)) * PPO * 16;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(TH >= 1, "tile");
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + TH - 1) / TH;
  auto kern = wgrad_kernel<T, STRIDE, LTW, KS, TPIX>;
  static srx_per_device_once attr_set;     // per instantiation, per device
  if (int rc = srx_ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(kern), LDS)) return rc;
  long ntiles = (long)p.B * p.tiles_x * p.tiles_y;
  if (p.nslices > ntiles) p.nslices = (int)ntiles;
  long grid = (long)d->ntasks * p.nslices;
  p.group = 0; p.ntasks = d->ntasks;
  if (d->group > 1 && d->ntasks % d->group == 0) {
    p.group = d->group;
    long ngrp = (long)(d->ntasks / d->group) * p.nslices;
    grid = ((ngrp + 7) / 8) * 8 * d->group;
  }
  SRX_CHECK_ARG(grid > 0 && grid < (1L << 31), "wgrad grid out of range");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, stream, p);
  return srx_check_launch("wgrad_kernel");
}

// SRX_WGRAD_WALK=1 selects the column-walk kernel for every wide 3x3 stride-1 launch (A/B runs).
static bool wgrad_walk_forced() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SRX_WGRAD_WALK");
    v = e ? atoi(e) : 0;
  }
  return v != 0;
}

template <typename T>
int dispatch(const srx_wgrad_desc* d, const WgradParams& p, hipStream_t s) {
  constexpr bool F = sizeof(T) == 4;
  if (d->ksize == 1) return launch<T, 1, 5, 1, 128>(d, p, s);
  if (d->stride == 1) {
    if (p.OW >= 24 && (d->tile_walk || wgrad_walk_forced())) return launch_walk<T, F ? 64 : 128>(d, p, s);
    if (p.OW >= 24) return launch<T, 1, 5, 3, F ? 64 : 128>(d, p, s);
    if (p.OW >= 12) return launch<T, 1, 4, 3, F ? 64 : 128>(d, p, s);
    return launch<T, 1, 3, 3, F ? 64 : 128>(d, p, s);
  }
  if (p.OW >= 24) return launch<T, 2, 5, 3, F ? 32 : 64>(d, p, s);
  if (p.OW >= 12) return launch<T, 2, 4, 3, F ? 32 : 64>(d, p, s);
  return launch<T, 2, 3, 3, F ? 32 : 64>(d, p, s);
}

}  // namespace

extern "C" int srx_conv2d_wgrad(const srx_wgrad_desc* d, void* stream) {
  SRX_CHECK_ARG(d != nullptr, "srx_conv2d_wgrad: null descriptor");
  SRX_CHECK_ARG(d->dtype == SRX_BF16 || d->dtype == SRX_F32, "srx_conv2d_wgrad: bad dtype %d", d->dtype);
  SRX_CHECK_ARG(d->ksize == 3 || d->ksize == 1, "srx_conv2d_wgrad: ksize must be 3 or 1");
  SRX_CHECK_ARG(d->stride == 1 || (d->stride == 2 && d->ksize == 3), "srx_conv2d_wgrad: bad stride %d", d->stride);
  SRX_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->ntasks > 0 && d->nslices > 0,
                "srx_conv2d_wgrad: non-positive size");
  SRX_CHECK_ARG(d->tasks, "srx_conv2d_wgrad: null task table");
  // a task's channels of one image (64 of them, spread over at most 32 octet planes when the task
  // pairs two distant groups) are addressed through a 32-bit buffer descriptor with signed byte counts
  SRX_CHECK_ARG((long)d->H * d->W * 256 * (d->dtype == SRX_BF16 ? 2 : 4) < (1L << 31),
                "srx_conv2d_wgrad: image of %d x %d is too large for one launch; tile the image", d->H, d->W);
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
  p.group = 0; p.ntasks = d->ntasks;
  SRX_CHECK_ARG(d->slice_stride >= 0, "srx_conv2d_wgrad: negative slice_stride");
  p.slice_stride = d->slice_stride;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == SRX_BF16 ? dispatch<bf16>(d, p, s) : dispatch<float>(d, p, s);
}
