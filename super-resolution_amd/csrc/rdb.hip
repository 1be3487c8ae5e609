// ResidualDenseBlock forward (esrgan/generator.py:17-23, with the RRDB tail of :37) as ONE launch.
//
// The five convolutions of a dense block re-read the block input and every earlier growth slice:
// run layer by layer they are five dependent launches whose fixed cost (launch boundary, prologue,
// epilogue drain: about 4.5 us each) outweighs their arithmetic on small feature maps.  Here a
// workgroup owns a 16 x 8 pixel tile of the block's output and computes all five layers for it:
//   * the block input x (64 channels) is staged ONCE into LDS over the tile plus a 5-pixel halo;
//   * layer k (k = 1..4) computes its 32 channels over the tile plus a (5 - k)-pixel halo (the
//     halo is recomputed by the neighbouring tiles: 1.5 x the arithmetic of the layer-by-layer
//     form), applies bias + LeakyReLU, and leaves them in LDS for the layers behind it (and, for
//     training, stores the tile's own pixels to the block buffer for the backward pass);
//   * layer 5 computes the 64 output channels of the tile as two 32-channel passes and applies
//     alpha * y + beta1 * x + beta2 * res2.
// Only weights stream during the launch, and they never touch LDS: each lane reads its own 16-byte
// fragment of a 16-input-channel chunk (9 taps x 32 x 16, 9 KB, the same for the four waves: L1
// hits) into registers one chunk ahead of the multiply.  With the activations resident in LDS and
// the weights in registers a layer needs no synchronisation at all: one barrier per LAYER (five
// per tile), where the layer-by-layer kernel has one per 16-channel chunk (52).  A first version
// that staged the weights through an LDS ring (one counted wait + one barrier per chunk) took
// 61 us per block at 16 x 64 x 64 against 51 us for the five launches.
// The GEMM view, the fragment layouts, the packed weight format and the order of every
// floating-point operation are those of conv_kernel (conv.hip), so the results equal the
// layer-by-layer path bit for bit.
//
// The data gradient of a dense block has the same shape as its forward pass (BWD = true): with the
// gradient buffer laid out [gy1 | gy2 | gy3 | gy4 | g_out], the gradient of growth slice j is one
// convolution over the planes [gy_(j+1) .. g_out] with the stacked, transposed, tap-flipped
// weights of convolutions j+1 .. 5, times LeakyReLU'(x_j).  Layer L of the kernel then computes
// gy_(4-L) from g_out (source 0, staged with its 5-pixel halo) and the slices computed before it,
// newest first (the channel order of the gradient buffer); the derivative masks come from the
// block's activation buffer (a global read per lane and layer, issued in front of the K loop);
// layer 4 gives the gradient of the block input, plus beta1 * g_out + beta2 * res2.
//
// bf16, 64 block channels, 32 growth channels only (what the reference instantiates,
// esrgan/generator.py:8,27); other shapes take the layer-by-layer path.
#include <stdlib.h>
#include <utility>
#include "srx_common.h"

namespace {

constexpr int HALO = 5;
constexpr int NCHUNK_TOTAL = 4 + 6 + 8 + 10 + 2 * 12;          // weight chunks of a block
// A wave multiplies the chunks of layers 0..3 and those of ONE 32-channel half of layer 4 (waves
// split that layer 2 halves x 2 sets of pixel groups): 40 chunks stream through its registers.
constexpr int NCHUNK_WAVE = 4 + 6 + 8 + 10 + 12;
constexpr unsigned WCHUNK_BYTES = 9 * 512 * 2;
constexpr unsigned OOB = 0xFFFFFFF0u;
// which waves take the odd groups alternates between the layers that have them
constexpr int wave_rot(int L) { return L == 3 ? 2 : 0; }
constexpr int bias_off(int L) { return 32 * L; }                            // floats; conv5: + 32 * pass

struct RdbParams {
  bf16* buf; long buf_is;           // block buffer (x | x1..x4) or gradient buffer (gy1..gy4 | g_out): 24 octet planes per image
  bf16* y; long y_is;               // output view, 8 planes
  const bf16* w; const float* bias;
  const bf16* res2; long r2_is;
  const bf16* act; long act_is;     // BWD: the block's activation buffer (derivative masks)
  float slope, alpha, beta1, beta2;
  int H, W, tiles_x, tiles_y, total;
#ifdef RDB_STAMPS
  unsigned long long* stamps;       // timing build: shader-clock stamps of workgroup 8 ([wave][32], 256 values)
#endif
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
typedef __attribute__((address_space(3))) void lds_void;

template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void mma16(const bf16x8& a, const bf16x8& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// One tap row (three taps) of one 16-channel chunk for R pixel groups: A = the chunk's weight
// fragments of that row (in registers), B = the activation fragments at the three tap shifts, read
// from the LDS-resident regions one tap row ahead: behind every MFMA of this row goes one fragment
// read of the next row (next_read(dx, rg)), so that the reads never hold the instruction stream
// for longer than the gap between two MFMAs (nine reads issued as a block in front of the row cost
// the matrix pipe about 40 idle cycles per row).  Same order of accumulation as conv_kernel: tap
// rows, taps, groups.
template <int R, bool PREFETCH, typename NextRead>
__device__ __forceinline__ void row_mma(const u32x4_ (&wf)[9], u32x4_ (&wnext)[9], __amdgpu_buffer_rsrc_t wrsrc,
                                        unsigned wlane, unsigned wnext_off, int dy, const bf16x8 (&xf)[3][R], f32x16* acc,
                                        NextRead&& next_read) {
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
    for (int rg = 0; rg < R; ++rg) {
#ifdef RDB_ABLATE_NO_MMA
      if (dy + dx > 0) { asm volatile("" :: "v"(xf[dx][rg]), "v"(wf[dy * 3 + dx])); continue; }   // timing-only build
#endif
      mma16(__builtin_bit_cast(bf16x8, wf[dy * 3 + dx]), xf[dx][rg], acc[rg]);
      next_read(dx, rg);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the weight fragment of the same tap of a later chunk, one load per tap: nine loads issued
    // back to back by four waves at once queue up in the address unit (16 cycles each) and hold
    // the instruction stream, and with it the MFMAs, for their whole length
    if constexpr (PREFETCH) {
#ifndef RDB_ABLATE_NO_WLOAD
      wnext[dy * 3 + dx] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (unsigned)(dy * 3 + dx) * 1024u, wnext_off, 0);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Position of a chunk in the stream of weight chunks a wave multiplies: (layer, source, chunk of
// the source).  Forward: sources in the order block input, x1, x2, ...; backward: newest slice
// first, g_out last (the channel order of each buffer, hence of the packed weights).
constexpr int layer_chunk0(int L) { return L == 0 ? 0 : L == 1 ? 4 : L == 2 ? 10 : L == 3 ? 18 : 28; }
// chunk i (0 .. 4 + 2 L - 1) of layer L: its source and its index inside the source
constexpr int layer_nchunks(int L) { return 4 + 2 * L; }
template <bool BWD> constexpr int chunk_src(int L, int i) { return BWD ? (i < 2 * L ? L - i / 2 : 0) : (i < 4 ? 0 : 1 + (i - 4) / 2); }
template <bool BWD> constexpr int chunk_cc(int L, int i) { return BWD ? (i < 2 * L ? i % 2 : i - 2 * L) : (i < 4 ? i : (i - 4) % 2); }
// planes (in buf) of the kernel's input and of the slice layer L produces; BWD: plane (in act) of its mask
template <bool BWD> constexpr int in_plane0() { return BWD ? 16 : 0; }
template <bool BWD> constexpr int slice_plane(int L) { return BWD ? 4 * (3 - L) : 8 + 4 * L; }
constexpr int mask_plane(int L) { return 8 + 4 * (3 - L); }

// Geometry of a TW x TH-pixel output tile (TW * TH = 64 or 128 pixels) and the kernel body for it.
// The 16 x 8 tile recomputes 1.5 x the arithmetic of the layer-by-layer form in its halos and needs
// 132 KB of LDS; the 8 x 8 tile recomputes 1.85 x and needs 87 KB, and gives a small feature map
// twice the workgroups.  NH > 1: a workgroup computes NH tiles stacked vertically one after the
// other from ONE staged input region (TH * NH + 10 rows): a feature map of two rounds of 16 x 8 tiles
// becomes one round of workgroups that stage 28 % fewer bytes and wait for them once (155 KB of LDS).
template <int TW, int TH, int NH = 1>
struct Tile {
  static constexpr int reg_h(int halo) { return TH + 2 * halo; }
  static constexpr int reg_w(int halo) { return TW + 2 * halo; }
  static constexpr int reg_p(int halo) { return reg_h(halo) * reg_w(halo); }
  static constexpr int X_P = (TH * NH + 2 * HALO) * reg_w(HALO);      // pixels of the staged input region
  // pixels per plane of source s (the input region covers all NH tiles, the slices one tile)
  static constexpr int src_p(int s) { return s == 0 ? X_P : reg_p(HALO - s); }
  // source s of a layer's input: 0 = block input (8 octet planes, halo 5); s = 1..4 = growth slice s
  // (4 planes, halo 5 - s).  Regions are stored [plane][pixel of the region], 16-byte slots.
  static constexpr int src_halo(int s) { return HALO - s; }
  static constexpr int src_planes(int s) { return s == 0 ? 8 : 4; }
  static constexpr int src_base(int s) { return s == 0 ? 0 : src_base(s - 1) + src_planes(s - 1) * src_p(s - 1); }
  static constexpr int ACT_SLOTS = 8 * X_P + 4 * (reg_p(4) + reg_p(3) + reg_p(2) + reg_p(1));
  static constexpr int BIAS_BASE = ACT_SLOTS;              // packed biases: 4 x 32 + 64 floats = 48 slots
  static constexpr int LDS_SLOTS = BIAS_BASE + 64;
  static constexpr int LDS_BYTES = LDS_SLOTS * 16;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static constexpr int X_SLOTS = 8 * X_P;
  static constexpr int X_DMA = (X_SLOTS + 255) / 256;           // staging instructions per wave for x
  // layer L = 0..4 (conv1..conv5): output halo, pixel groups of 32, groups per wave
  static constexpr int out_halo(int L) { return L < 4 ? HALO - 1 - L : 0; }
  static constexpr int out_p(int L) { return reg_p(out_halo(L)); }
  static constexpr int out_groups(int L) { return (out_p(L) + 31) / 32; }      // 16 x 8: 12 10 8 6 4;  8 x 8: 8 7 5 4 2
  // layer 4: each wave takes half of the groups for one 32-channel half of the output
  static constexpr int NG5 = TW * TH / 32;
  static_assert(NG5 == 2 || NG5 == 4, "tile of 64 or 128 pixels");
  static constexpr int rw_max(int L) { return L == 4 ? NG5 / 2 : (out_groups(L) + 3) / 4; }

  template <bool BWD, bool STORE>
  static __device__ __forceinline__ void run(const RdbParams& p, u32x4_* smem) {
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, pl = lane & 31, hh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    // layer 4: this wave's 32-channel half of the output and its pair of pixel groups
    const int pass5 = wave_u & 1, pair5 = wave_u >> 1;
#ifdef RDB_STAMPS
    int stamp_k = 0;
    auto stamp = [&]() {
      const unsigned long long tnow = __builtin_readcyclecounter();
      if (blockIdx.x == 8 && lane == 0 && p.stamps && stamp_k < 32) p.stamps[wave_u * 32 + stamp_k] = tnow;
      ++stamp_k;
    };
    stamp();
#else
    auto stamp = [] {};
#endif

    // workgroup -> tile, one contiguous run of tiles per XCD (neighbouring tiles share most of their
    // halo: they should meet in one XCD's L2)
    unsigned t = blockIdx.x;
    {
      const unsigned chunk = (unsigned)p.total / 8u;
      if (t < 8u * chunk) t = (t & 7u) * chunk + (t >> 3);
    }
    const int tx = (int)(t % (unsigned)p.tiles_x);
    const unsigned tr = t / (unsigned)p.tiles_x;
    const int ty = (int)(tr % (unsigned)p.tiles_y), b = (int)(tr / (unsigned)p.tiles_y);
    const int oy_wg = ty * TH * NH, ox0 = tx * TW;
    const int H = p.H, W = p.W;
    const unsigned plane_b = (unsigned)H * (unsigned)W * 16u;          // bytes of one octet plane

    bf16* const img = p.buf + (long)b * p.buf_is;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        img + (long)in_plane0<BWD>() * H * W * 8, 0, (int)(8u * plane_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(img, 0, (int)(24u * plane_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (long)b * p.y_is, 0, (int)(8u * plane_b), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.w), 0, (int)(NCHUNK_TOTAL * WCHUNK_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16*>(BWD ? p.act + (long)b * p.act_is : p.w), 0, (int)(BWD ? 24u * plane_b : 0u), 0x00020000);

    // ---- the kernel's input (block input / g_out) over the tile + 5-pixel halo: X_DMA LDS-DMA
    // instructions per wave; pixels outside the image (the zero padding) and the slots past the region
    // get an out-of-range offset, for which the hardware deposits zeros (the few slots past the region
    // belong to the first slice, which is written later)
#pragma unroll
    for (int j = 0; j < X_DMA; ++j) {
      const unsigned slot = (unsigned)(j * 256 + tid);
      const unsigned q = slot / (unsigned)X_P, pix = slot - q * (unsigned)X_P;
      const unsigned r = pix / (unsigned)reg_w(HALO), c = pix - r * (unsigned)reg_w(HALO);
      const int gy = oy_wg - HALO + (int)r, gx = ox0 - HALO + (int)c;
      const bool ok = (slot < (unsigned)X_SLOTS) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
      const unsigned off = q * plane_b + (unsigned)(gy * W + gx) * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_void*)(smem + j * 256 + wave_u * 64), 16, ok ? off : OOB, 0, 0, 0);
    }
    // packed biases (the accumulators' initial values) -> LDS; the data gradient has none
    if (!BWD && tid < 48) *reinterpret_cast<f32x4*>(smem + BIAS_BASE + tid) = *reinterpret_cast<const f32x4*>(p.bias + tid * 4);

    // ---- weights: every wave multiplies all 32 output channels of a chunk, so every lane reads its own
    // 16-byte fragment of each of the nine taps straight into registers (1 KB per wave instruction,
    // served by the L1 after the first wave), WDEPTH chunks ahead of the multiply.  Nothing but the
    // activations lives in LDS, so no barrier is needed inside a layer.  Position n of a wave's chunk
    // stream is chunk n of the packed block for layers 0..3 and chunk n + 12 * (its half) for layer 4.
    const unsigned wlane = (unsigned)(hh * 32 + pl) * 16u;
    constexpr int WDEPTH = 3;                  // chunks in flight ahead of the multiply (L2 latency under load: about 1 us)
    u32x4_ wbuf[WDEPTH + 1][9];
    auto chunk_bytes = [&](int n) -> unsigned {          // n: compile-time after unrolling
      if (NH > 1 && n >= NCHUNK_WAVE) n -= NCHUNK_WAVE;    // the next tile's first chunks
      return n < layer_chunk0(4) ? (unsigned)n * WCHUNK_BYTES : (unsigned)(n + 12 * pass5) * WCHUNK_BYTES;
    };
    auto load_w = [&](int n, u32x4_ (&dst)[9]) {
#ifdef RDB_ABLATE_NO_WLOAD
      if (n >= 1) return;                       // timing-only build: weight fragments are never refreshed
#endif
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
        dst[tp] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (unsigned)tp * 1024u, chunk_bytes(n), 0);
    };
#pragma unroll
    for (int n0 = 0; n0 < WDEPTH; ++n0) load_w(n0, wbuf[n0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the input tile has landed (this wave's share) ...
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // ... and every other wave's, and the biases
    stamp();
#ifdef RDB_ABLATE_EXIT_AFTER_STAGE
    if (p.total > 0) return;                     // timing-only build: launch + input staging alone
#endif

    const bf16x8* const lds = reinterpret_cast<const bf16x8*>(smem);
    bf16x8* const ldsw = reinterpret_cast<bf16x8*>(smem);
    const float* const lbias = reinterpret_cast<const float*>(smem + BIAS_BASE);

    // Per-layer state of a wave (accumulators, the pixels of its groups), kept in two alternating sets:
    // the forward pass finishes layer L - 1's epilogue under the first MFMAs of layer L (below).
    f32x16 accs[2][3];
    int pys[2][3], pxs[2][3], pps[2][3];
    bool lvs[2] = {true, true};
    static_assert(NCHUNK_WAVE % (WDEPTH + 1) == 0, "the weight ring keeps its phase from tile to tile");

    // the NH tiles of this workgroup, top to bottom (a run-time loop: the layer code exists once)
#pragma clang loop unroll(disable)
    for (int half = 0; half < NH; ++half) {
    const int oy0 = oy_wg + half * TH;
    const int xoff = half * TH * reg_w(HALO);           // the tile's first row inside the staged input region
    static_for<5>([&](auto Lc) {
      constexpr int L = decltype(Lc)::value;
      constexpr int OH_ = out_halo(L), OWd = reg_w(OH_), OP = out_p(L), NG = out_groups(L), RW = rw_max(L);
      constexpr int P = L & 1;
      static_assert(RW <= 3, "state arrays");
#ifdef RDB_ABLATE_LAYERS
      if (L >= RDB_ABLATE_LAYERS) return;         // timing-only build: the first layers alone
#endif
      // this wave's pixel groups.  Layers 0..3: g = rg * 4 + ((wave + rot) & 3), only the last rg can be
      // missing (the wave then multiplies a dummy group whose results are dropped: the waves that have
      // the group set the layer's length anyway); layer 4: the pair of groups 2 * pair5 + rg.
      const int wsel = (wave_u + wave_rot(L)) & 3;
      const bool last_valid = L == 4 || (RW - 1) * 4 + wsel < NG;          // wave-uniform
      lvs[P] = last_valid;
      int (&py)[3] = pys[P], (&px)[3] = pxs[P], (&pp)[3] = pps[P];
#pragma unroll
      for (int rg = 0; rg < RW; ++rg) {
        pp[rg] = (L == 4 ? pair5 * (NG5 / 2) + rg : rg * 4 + wsel) * 32 + pl;
        const int pc = pp[rg] < OP ? pp[rg] : OP - 1;
        py[rg] = pc / OWd;
        px[rg] = pc - py[rg] * OWd;
      }
      // BWD: LeakyReLU'(x_j) of this wave's pixels, read from the activation buffer in front of the K
      // loop (two octets per lane and group); pixels outside the image read zeros and are zeroed anyway
      [[maybe_unused]] u32x4_ mk[RW][2];
      if constexpr (BWD && L < 4) {
#pragma unroll
        for (int rg = 0; rg < RW; ++rg) {
          const int gy = oy0 - OH_ + py[rg], gx = ox0 - OH_ + px[rg];
          const bool ok = (rg < RW - 1 || last_valid) && pp[rg] < OP && ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            mk[rg][j] = __builtin_amdgcn_raw_buffer_load_b128(
                arsrc, ok ? (unsigned)(mask_plane(L) + 2 * j + hh) * plane_b + (unsigned)(gy * W + gx) * 16u : OOB, 0, 0);
        }
      }

      // last layer: the second residual of this wave's pixels, likewise read in front of the K loop
      [[maybe_unused]] u32x4_ r2w[RW][2];
      if constexpr (L == 4) {
        if (p.res2 != nullptr) {
          const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<bf16*>(p.res2 + (long)b * p.r2_is), 0, (int)(8u * plane_b), 0x00020000);
#pragma unroll
          for (int rg = 0; rg < RW; ++rg) {
            const int gy = oy0 + py[rg], gx = ox0 + px[rg];
            const bool in_img = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              r2w[rg][j] = __builtin_amdgcn_raw_buffer_load_b128(
                  rrsrc, in_img ? (unsigned)(pass5 * 4 + 2 * j + hh) * plane_b + (unsigned)(gy * W + gx) * 16u : OOB, 0, 0);
          }
        }
      }

      f32x16 (&acc)[3] = accs[P];
      if constexpr (BWD) {
#pragma unroll
        for (int rg = 0; rg < RW; ++rg)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[rg][i] = 0.f;
      } else {
        const f32x4* bv = reinterpret_cast<const f32x4*>(lbias + bias_off(L) + (L == 4 ? 32 * pass5 : 0) + hh * 16);
        const f32x4 b0 = bv[0], b1 = bv[1], b2 = bv[2], b3 = bv[3];
#pragma unroll
        for (int rg = 0; rg < RW; ++rg)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[rg][i] = b0[i]; acc[rg][4 + i] = b1[i]; acc[rg][8 + i] = b2[i]; acc[rg][12 + i] = b3[i];
          }
      }
      // Forward: the epilogue of the layer before this one (LeakyReLU, bf16, into LDS and to the block
      // buffer) is straight VALU work, about a thousand cycles in which the matrix pipe would idle.
      // Its results are this layer's LAST source, so it runs in steps of four VALU instructions
      // behind the first MFMAs of this layer, one step per MFMA (an MFMA gap of 32 cycles hides about
      // five 4-cycle instructions, the fragment read included): accumulator reads / multiplies / maxima
      // in halves, pack, zero, store for each (group, octet) of the previous layer.
      constexpr bool DEFER = !BWD && L > 0;
      constexpr int Lp = L > 0 ? L - 1 : 0, Q = P ^ 1;
      constexpr int RWp = DEFER ? rw_max(Lp) : 0, NSTEP = 18 * RWp;
      [[maybe_unused]] int e_lds[3];
      [[maybe_unused]] unsigned e_goff[3];
      [[maybe_unused]] bool e_ok[3], e_img[3];
      [[maybe_unused]] float ev[8], et[8];
      if constexpr (DEFER) {
        constexpr int OHp = out_halo(Lp), OPp = out_p(Lp);
#pragma unroll
        for (int rg = 0; rg < RWp; ++rg) {
          const int qy = pys[Q][rg], qx = pxs[Q][rg];
          const int gy = oy0 - OHp + qy, gx = ox0 - OHp + qx;
          e_ok[rg] = (rg < RWp - 1 || lvs[Q]) && pps[Q][rg] < OPp;
          e_img[rg] = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
          const bool own = e_ok[rg] && e_img[rg] && qy >= OHp && qy < OHp + TH && qx >= OHp && qx < OHp + TW;
          e_lds[rg] = src_base(Lp + 1) + hh * OPp + pps[Q][rg];
          e_goff[rg] = own ? (unsigned)(slice_plane<BWD>(Lp) + hh) * plane_b + (unsigned)(gy * W + gx) * 16u : OOB;
        }
      }
      [[maybe_unused]] u32x4_ eow;
      auto epi_step = [&](int s) {                    // s: a constant after unrolling
        if constexpr (DEFER) {
          if (s >= NSTEP) return;
          constexpr int OPp = out_p(Lp);
          const int piece = s / 9, sub = s % 9, rg = piece >> 1, j = piece & 1, h4 = (sub & 1) * 4;
          if (sub < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ev[h4 + e] = accs[Q][rg][j * 8 + h4 + e];
          } else if (sub < 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) et[h4 + e] = ev[h4 + e] * p.slope;
          } else if (sub < 6) {
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, %1, %2" : "=v"(ev[h4 + e]) : "v"(ev[h4 + e]), "v"(et[h4 + e]));
          } else if (sub == 6) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)ev[e];
            eow = __builtin_bit_cast(u32x4_, o);
          } else if (sub == 7) {
#pragma unroll
            for (int k = 0; k < 4; ++k) eow[k] = e_img[rg] ? eow[k] : 0u;
          } else {
            if (e_ok[rg]) ldsw[e_lds[rg] + 2 * j * OPp] = __builtin_bit_cast(bf16x8, eow);
            if constexpr (STORE)
              __builtin_amdgcn_raw_buffer_store_b128(eow, grsrc, e_goff[rg] == OOB ? OOB : e_goff[rg] + (unsigned)(2 * j) * plane_b, 0, 0);
          }
        }
      };

      // K loop: 16 channels a step, sources in the channel order of the buffer.  The activation
      // fragments of a tap row are read while the row before it is multiplied, across chunk boundaries
      // too: with one wave per SIMD nothing else hides the LDS latency.
      constexpr int NC = layer_nchunks(L);
      int ps[L + 1][RW];                              // per source: this lane's pixel at tap (0, 0)
      static_for<L + 1>([&](auto Sc) {
        constexpr int S = decltype(Sc)::value;
        constexpr int SW = reg_w(src_halo(S)), O = L - S;   // O: offset of this layer's region inside the source's, minus the tap pad
#pragma unroll
        for (int rg = 0; rg < RW; ++rg) ps[S][rg] = (py[rg] + O) * SW + px[rg] + O;
      });
      bf16x8 xf[2][3][RW];
      // base of chunk i's fragments of tap row dy for this lane's k half (ps[S][rg] + dx is added per read)
      auto row_base = [&](auto Ic, int dy) -> const bf16x8* {
        constexpr int i = decltype(Ic)::value;
        constexpr int S = chunk_src<BWD>(L, i), cc = chunk_cc<BWD>(L, i);
        constexpr int SW = reg_w(src_halo(S)), SP = src_p(S);
        return lds + src_base(S) + (2 * cc + hh) * SP + dy * SW + (S == 0 ? xoff : 0);
      };
      {
        const bf16x8* sx = row_base(std::integral_constant<int, 0>{}, 0);
        constexpr int S0 = chunk_src<BWD>(L, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int rg = 0; rg < RW; ++rg) xf[0][dx][rg] = sx[ps[S0][rg] + dx];
      }
      static_for<NC>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        constexpr int n = layer_chunk0(L) + i;
#ifdef RDB_STAMPS_L0
        if constexpr (L == 0) stamp();               // timing build: a stamp in front of every chunk of the first layer
#endif
        constexpr bool PF = NH > 1 || n + WDEPTH < NCHUNK_WAVE;
        constexpr int NB_ = (n + WDEPTH) % (WDEPTH + 1);
        const unsigned noff = chunk_bytes(n + WDEPTH);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int cur = (3 * i + dy) & 1;
          // the row after this one: row dy + 1 of this chunk, or row 0 of the next chunk
          constexpr int inext = i + 1 < NC ? i + 1 : i;
          const bool more = dy < 2 || i + 1 < NC;
          const bf16x8* sxn = dy < 2 ? row_base(Ic, dy + 1) : row_base(std::integral_constant<int, inext>{}, 0);
          const int* psn = dy < 2 ? ps[chunk_src<BWD>(L, i)] : ps[chunk_src<BWD>(L, inext)];
          auto& dst = xf[cur ^ 1];
          // Forward: the slice the layer before this one produced is the LAST source of this layer, so
          // the waves do not meet at the end of a layer but here, in front of the first fragment read
          // of that slice (issued behind the MFMAs of this row): by now every wave has long left its
          // epilogue, and nobody waits.  (Backward reads the newest slice first: barrier at layer end.)
          if constexpr (!BWD && L > 0 && i == NC - 3) {
            if (dy == 2) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's slice is in LDS ...
              __builtin_amdgcn_s_barrier();                             // ... and every other wave's
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          row_mma<RW, PF>(wbuf[n % (WDEPTH + 1)], wbuf[NB_], wrsrc, wlane, noff, dy, xf[cur], acc,
                          [&](int dx, int rg) {
                            if (more) dst[dx][rg] = sxn[psn[rg] + dx];
                            epi_step(((i * 3 + dy) * 3 + dx) * RW + rg);
                          });
          __builtin_amdgcn_sched_barrier(0);
        }
      });

      stamp();
      if constexpr (L < 4 && !BWD) {
        // (its 18 steps per group must be done before the next layer's barrier, in chunk NC - 3 of that layer)
        static_assert(18 * RW <= 9 * rw_max(L + 1) * (layer_nchunks(L + 1) - 3), "room for the deferred epilogue");
        stamp();
      } else if constexpr (L < 4) {
        // backward: times LeakyReLU'(x_j) (the forward pass does this step under the next layer).  Zero outside
        // the image (the next layer's padding); into LDS for the layers behind, and the tile's own
        // pixels to the buffer (the backward pass / the weight gradients read them)
#pragma unroll
        for (int rg = 0; rg < RW; ++rg) {
          const bool has = rg < RW - 1 || last_valid;                 // wave-uniform
          const int gy = oy0 - OH_ + py[rg], gx = ox0 - OH_ + px[rg];
          const bool lane_ok = has && pp[rg] < OP;
          const bool in_img = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
          const bool own = lane_ok && in_img && py[rg] >= OH_ && py[rg] < OH_ + TH && px[rg] >= OH_ && px[rg] < OH_ + TW;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            // (this tail is straight VALU work in front of a barrier: instruction count is its price.
            //  The maximum is taken by v_max_f32 itself: fmaxf() first quiets signalling NaNs with a
            //  second v_max per value; the mask's sign is tested on the bf16 bit patterns; pixels
            //  outside the image are zeroed on the packed words)
            bf16x8 o;
            if constexpr (BWD) {
              const u32x4_ mw = mk[rg][j];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = acc[rg][j * 8 + e];
                const bool pos = (e & 1) ? (int)mw[e >> 1] > 0xFFFF : (short)(mw[e >> 1] & 0xFFFFu) > 0;
                o[e] = (bf16)(pos ? v : v * p.slope);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = acc[rg][j * 8 + e], t = v * p.slope;
                float r;
                asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(t));
                o[e] = (bf16)r;
              }
            }
            {
              u32x4_ ow = __builtin_bit_cast(u32x4_, o);
#pragma unroll
              for (int k = 0; k < 4; ++k) ow[k] = in_img ? ow[k] : 0u;
              o = __builtin_bit_cast(bf16x8, ow);
            }
            const int oct = 2 * j + hh;
            if (lane_ok) ldsw[src_base(L + 1) + oct * OP + pp[rg]] = o;
            if constexpr (STORE) {
              const unsigned off = own ? (unsigned)(slice_plane<BWD>(L) + oct) * plane_b + (unsigned)(gy * W + gx) * 16u : OOB;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, o), grsrc, off, 0, 0);
            }
          }
        }
        // the next layer reads what every wave has just written (forward: the barrier comes later)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (BWD) __builtin_amdgcn_s_barrier();
        stamp();
      } else {
        // last layer: alpha * y + beta1 * (the kernel's input) + beta2 * res2 for this wave's 32 channels
#pragma unroll
        for (int rg = 0; rg < RW; ++rg) {
          const int gy = oy0 + py[rg], gx = ox0 + px[rg];
          const bool in_img = ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
          const int xs = (py[rg] + HALO) * reg_w(HALO) + px[rg] + HALO;
          if (p.alpha != 1.f) acc[rg] *= p.alpha;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int oct = pass5 * 4 + 2 * j + hh;
            const bf16x8 xr = lds[src_base(0) + oct * X_P + xs + xoff];
            const float b1 = p.beta1;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[rg][j * 8 + e] += b1 * (float)xr[e];
          }
          if (p.res2 != nullptr) {
            const float b2 = p.beta2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const bf16x8 r2 = __builtin_bit_cast(bf16x8, r2w[rg][j]);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[rg][j * 8 + e] += b2 * (float)r2[e];
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int oct = pass5 * 4 + 2 * j + hh;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[rg][j * 8 + e];
            const unsigned off = in_img ? (unsigned)oct * plane_b + (unsigned)(gy * W + gx) * 16u : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, o), yrsrc, off, 0, 0);
          }
        }
        stamp();
      }
    });
    // the next tile's first layers write the slice regions this tile's last layer may still be reading
    if (NH > 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    }
  }
};

template <bool BWD, bool STORE, int TW, int TH, int NH>
__global__ __launch_bounds__(256, 1) void rdb_kernel(const RdbParams p) {
  extern __shared__ __attribute__((aligned(16))) u32x4_ smem[];
  Tile<TW, TH, NH>::template run<BWD, STORE>(p, smem);
}

}  // namespace

#ifdef RDB_STAMPS
static unsigned long long* g_rdb_stamps = nullptr;
extern "C" void srx_rdb_debug_stamps(void* buf) { g_rdb_stamps = static_cast<unsigned long long*>(buf); }
#endif

template <bool BWD, bool STORE, int TW, int TH, int NH>
static int rdb_launch_tile(RdbParams p, const srx_rdb_desc* d, hipStream_t s, const char* who) {
  constexpr int LDS_BYTES = Tile<TW, TH, NH>::LDS_BYTES;
  p.tiles_x = (d->W + TW - 1) / TW;
  p.tiles_y = (d->H + TH * NH - 1) / (TH * NH);
  const long total = (long)p.tiles_x * p.tiles_y * d->B;
  SRX_CHECK_ARG(total < (1L << 31), "%s: grid out of range", who);
  p.total = (int)total;
  static srx_per_device_once attr;         // per instantiation, per device
  auto kern = rdb_kernel<BWD, STORE, TW, TH, NH>;
  if (int rc = srx_ensure_dynamic_lds(&attr, reinterpret_cast<const void*>(kern), LDS_BYTES)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), LDS_BYTES, s, p);
  return srx_check_launch(BWD ? "rdb_kernel (backward)" : "rdb_kernel (forward)");
}

static int rdb_launch(const srx_rdb_desc* d, void* stream, bool bwd, const char* who) {
  SRX_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  SRX_CHECK_ARG(d->dtype == SRX_BF16, "%s: bf16 only (got dtype %d)", who, d->dtype);
  SRX_CHECK_ARG(d->nf == 64 && d->gc == 32, "%s: 64 block channels and 32 growth channels only (got %d, %d)", who,
                d->nf, d->gc);
  SRX_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0, "%s: non-positive size", who);
  SRX_CHECK_ARG(d->buf && d->y && d->w && (bwd ? d->act != nullptr : d->bias != nullptr),
                "%s: null buf / y / w / %s pointer", who, bwd ? "act" : "bias");
  SRX_CHECK_ARG(((uintptr_t)d->buf | (uintptr_t)d->y | (uintptr_t)d->w | (uintptr_t)d->bias | (uintptr_t)d->res2 |
                 (uintptr_t)d->act) % 16 == 0, "%s: pointers must be 16-byte aligned", who);
  // one image's 24 planes are addressed through a 32-bit buffer descriptor
  SRX_CHECK_ARG((long)d->H * d->W * 16 * 24 < 0x7FFFFFF0L, "%s: image of %d x %d is too large for one launch; "
                "tile the image", who, d->H, d->W);
  RdbParams p;
  p.buf = static_cast<bf16*>(d->buf); p.buf_is = d->buf_img_stride;
  p.y = static_cast<bf16*>(d->y); p.y_is = d->y_img_stride;
  p.w = static_cast<const bf16*>(d->w); p.bias = d->bias;
  p.res2 = static_cast<const bf16*>(d->res2); p.r2_is = d->res2_img_stride;
  p.act = static_cast<const bf16*>(d->act); p.act_is = d->act_img_stride;
  p.slope = d->act_slope; p.alpha = d->alpha; p.beta1 = d->beta1; p.beta2 = d->beta2;
  p.H = d->H; p.W = d->W;
  SRX_CHECK_ARG(d->tile >= 0 && d->tile <= 2, "%s: tile must be 0 (16 x 8), 1 (8 x 8) or 2 (16 x 8, two per workgroup), got %d",
                who, d->tile);
#ifdef RDB_STAMPS
  p.stamps = g_rdb_stamps;
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int mode = bwd ? 2 : (d->store_growth ? 1 : 0);
  if (d->tile == 0) {
    if (mode == 2) return rdb_launch_tile<true, true, 16, 8, 1>(p, d, s, who);
    return mode == 1 ? rdb_launch_tile<false, true, 16, 8, 1>(p, d, s, who) : rdb_launch_tile<false, false, 16, 8, 1>(p, d, s, who);
  }
  if (d->tile == 2) {
    if (mode == 2) return rdb_launch_tile<true, true, 16, 8, 2>(p, d, s, who);
    return mode == 1 ? rdb_launch_tile<false, true, 16, 8, 2>(p, d, s, who) : rdb_launch_tile<false, false, 16, 8, 2>(p, d, s, who);
  }
  if (mode == 2) return rdb_launch_tile<true, true, 8, 8, 1>(p, d, s, who);
  return mode == 1 ? rdb_launch_tile<false, true, 8, 8, 1>(p, d, s, who) : rdb_launch_tile<false, false, 8, 8, 1>(p, d, s, who);
}

extern "C" int srx_rdb_forward(const srx_rdb_desc* d, void* stream) { return rdb_launch(d, stream, false, "srx_rdb_forward"); }
extern "C" int srx_rdb_backward(const srx_rdb_desc* d, void* stream) { return rdb_launch(d, stream, true, "srx_rdb_backward"); }
