// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 on 32x32 MFMA tiles.
//
// GEMM view:  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],  k = (tap, cin).
// The A operand is the packed weight block (32 couts x 16 cin of one tap), the
// B operand 32 output pixels x the same 16 cin read from an LDS halo tile at the
// tap's (dy,dx) shift -- no im2col buffer exists.  Activations live in the C8
// layout ([B][C/8][H][W][8]), so a 16-channel K step is two octet planes and
// every LDS fragment read is one 16-byte slot per lane.
//
// One workgroup = 4 waves = (4*RW) groups of 32 output pixels x (32*NB) couts.
// K loop over 16-channel chunks through a 2- or 3-stage LDS ring filled by LDS-DMA
// (buffer_load ... lds): later chunks stream in while chunk c is multiplied; one
// counted s_waitcnt vmcnt + one raw s_barrier per chunk.
//
// Epilogue (bias, derivative mask, LeakyReLU/ReLU, residual scaling, optional
// PixelShuffle(2) scatter) runs on the accumulators; the weight packer orders
// the 32 rows of a block so each lane's registers 0..7 / 8..15 are two whole
// output octets, stored with 16-byte writes and no cross-lane traffic.
#include <stdlib.h>
#include "srx_common.h"

namespace {

struct ConvParams {
  const void* x; long x_is;
  void* y; long y_is;
  const void* w; const float* bias;
  const void* res1; long r1_is;
  const void* res2; long r2_is;
  const void* mask; long m_is;
  float alpha, beta1, beta2, act_slope, mask_slope;
  int H, W, OH, OW;
  int cin8, cout8, nchunk, nblk, ngroups;
  int tiles_x, tiles_y;
  int ps;
  // XCD-aware workgroup -> tile map (0: off): workgroup L runs tile (L % 8) * remap_chunk + L / 8 for
  // L < 8 * remap_chunk, decoded with the shifts lgx / lgy when the grid's x and y extents are
  // powers of two (lgx >= 0), else by division
  unsigned remap_chunk;
  int lgx, lgy;
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

// Issue one chunk's LDS-DMA: DPC instructions per wave.  Wave instruction g = j * WAVES + wave of
// the workgroup fills the 64 slots [64 g, 64 g + 64) of the stage: activation slots first (XPP of
// them, a multiple of 64), weight slots behind.  An instruction is all activation or all weight;
// only for the one j where the boundary falls between waves is the choice made at run time (a
// wave-uniform branch).
template <typename T, int DPC, int XPP, int THREADS>
__device__ __forceinline__ void conv_issue_chunk(__amdgpu_buffer_rsrc_t xrsrc, __amdgpu_buffer_rsrc_t wrsrc,
                                                 const unsigned* off, unsigned xcoff, unsigned wcoff,
                                                 char* ls, int wave_u) {
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr unsigned OOB = 0xFFFFFFF0u;
#pragma unroll
  for (int j = 0; j < DPC; ++j) {
    lds_void* dst = (lds_void*)(ls + (j * THREADS + wave_u * 64) * 16);
    // (j is a constant after unrolling: the first two tests fold away)
    if ((j + 1) * THREADS <= XPP) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, dst, 16, off[j] == OOB ? OOB : off[j] + xcoff, 0, 0, 0);
    } else if (j * THREADS >= XPP) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, dst, 16, off[j] == OOB ? OOB : off[j] + wcoff, 0, 0, 0);
    } else {
      if (j * THREADS + wave_u * 64 < XPP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, dst, 16, off[j] == OOB ? OOB : off[j] + xcoff, 0, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, dst, 16, off[j] == OOB ? OOB : off[j] + wcoff, 0, 0, 0);
    }
  }
}

// Stage geometry shared by the kernel and its launcher (16-byte slots).
template <typename T, int STRIDE, int NB, int RW, int LTW, int KS, int WAVES, int CW>
struct StageGeom {
  static constexpr int THREADS = 64 * WAVES;
  static constexpr int TH = (WAVES / CW) * RW * (32 >> LTW);
  static constexpr int IH = (TH - 1) * STRIDE + KS, IW = ((1 << LTW) - 1) * STRIDE + KS;
  static constexpr int PPO = Oct<T>::PPO;
  static constexpr int XP = 2 * IH * IW * PPO;                   // live activation slots
  static constexpr int WP = NB * CW * KS * KS * 64 * PPO;        // live weight slots
  // Regions padded to whole WORKGROUP instructions.  Padding them to whole wave instructions only
  // (64 slots: 5 instead of 6 staging instructions per wave and chunk and 12 KB less LDS for the
  // 32-channel dense-block shape) was built and measured: no faster stand-alone, and 1.3 ms per
  // training step slower beside the weight-gradient workgroups of the other stream.
  static constexpr int GRAN = THREADS;
  static constexpr int XPP = (XP + GRAN - 1) / GRAN * GRAN, WPP = (WP + GRAN - 1) / GRAN * GRAN;
  static constexpr int DPC = ((XPP + WPP) / 64 + WAVES - 1) / WAVES;           // instructions per wave and chunk
  static constexpr int STAGE = DPC * THREADS;                    // slots per ring stage
};

// Workgroup shape.  WAVES waves = PW pixel groups of waves x CW output-block groups: a wave owns RW
// groups of 32 pixels x NB blocks of 32 output channels; the workgroup stages NB*CW blocks of
// weights.  Four waves put one wave on each SIMD; the eight-wave shapes put two there, so that one
// wave's staging issue, fragment reads and barrier waits run under the other's MFMAs — what small
// problems (one workgroup per CU) need, since nothing else shares the CU with them.
template <typename T, int NB, int RW, int WAVES, int CW>
struct RingDepth {
  // fp32 tiles are twice as large; the two-block and four-row-group tiles stay at two stages so
  // that two workgroups (or one plus a weight-gradient workgroup) still fit a CU's 160 KB (a third
  // stage for the two-block shape was measured: 17.9 against 17.6 us at 192 -> 64 channels on
  // 16 x 64x64, and a quarter slower on large grids, where it halves the workgroups per CU)
  static constexpr int value = (sizeof(T) == 2 && NB * CW == 1 && RW * (WAVES / CW) <= 8) ? 3 : 2;
};

template <typename T, int STRIDE, int NB, int RW, int LTW, int KS, int WAVES = 4, int CW = 1>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 8 ? 4 : (sizeof(T) == 2 ? 2 : 1)))
void conv_kernel(const ConvParams p) {
  // (register caps: 4-wave bf16 shapes at most 256, 8-wave shapes at most 128 per lane, so a
  //  convolution workgroup and a weight-gradient workgroup of the other stream fit one SIMD's
  //  register file together)
  typedef typename Frag<T>::type frag_t;
  constexpr int THREADS = 64 * WAVES;
  constexpr int PW = WAVES / CW;            // pixel groups of waves
  constexpr int TW = 1 << LTW;
  constexpr int RPG = 32 >> LTW;            // rows per 32-pixel group
  constexpr int TH = PW * RW * RPG;
  constexpr int PAD = KS / 2;
  constexpr int TAPS = KS * KS;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int IW = (TW - 1) * STRIDE + KS;
  constexpr int NPIX = IH * IW;
  constexpr int PPO = Oct<T>::PPO;
  constexpr int EPP = 16 / sizeof(T);       // elements per 16-byte piece
  typedef StageGeom<T, STRIDE, NB, RW, LTW, KS, WAVES, CW> SG;
  constexpr int XP = SG::XP, WP = SG::WP, XPP = SG::XPP;
  constexpr int STAGE = SG::STAGE;          // 16-byte slots per ring stage
  constexpr int NSTG = RingDepth<T, NB, RW, WAVES, CW>::value;   // ring depth
  constexpr int DIST = NSTG - 1;            // chunks kept in flight ahead of the one being multiplied
  constexpr int DMA_PER_CHUNK = SG::DPC;    // LDS-DMA instructions each wave issues per chunk
  static_assert(DMA_PER_CHUNK <= 31, "vmcnt budget");
  // consecutive row groups of a wave are consecutive image rows: activation fragments are shared
  // between tap rows (see the K loop)
  constexpr bool XROWS = (STRIDE == 1 && LTW == 5 && KS == 3 && RW > 1);

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  extern __shared__ __attribute__((aligned(16))) u32x4_ smem[];   // NSTG ring stages

  // Convolutions sit on the critical path of a layer chain; weight-gradient workgroups that
  // share the CU (other stream) only fill the issue slots these waves leave idle.
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int pl = lane & 31;                 // pixel / row index inside the MFMA tile
  const int hh = lane >> 5;                 // k half

  // grid = (tiles_x * ngroups, tiles_y, B): no run-time divisions on the way to the first load
  // in the common single-group case
  int tx = blockIdx.x, grp = 0;
  int ty = blockIdx.y;
  int b = blockIdx.z;
  if (p.remap_chunk) {
    // Workgroups go to the 8 XCDs round-robin in launch order.  Give each XCD a contiguous run of
    // tiles instead of every eighth one: neighbouring tiles, which share halo rows and columns,
    // then meet in one XCD's L2 instead of each fetching the halo from memory.
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned t = L < 8u * p.remap_chunk ? (L & 7u) * p.remap_chunk + (L >> 3) : L;
    if (p.lgx >= 0) {
      tx = (int)(t & ((1u << p.lgx) - 1u));
      ty = (int)((t >> p.lgx) & ((1u << p.lgy) - 1u));
      b = (int)(t >> (p.lgx + p.lgy));
    } else {
      const unsigned r = t / gridDim.x;
      tx = (int)(t - r * gridDim.x);
      b = (int)(r / gridDim.y);
      ty = (int)(r - (unsigned)b * gridDim.y);
    }
  }
  if (p.ngroups > 1) { grp = tx % p.ngroups; tx /= p.ngroups; }
  const int pg = wave % PW;                 // this wave's pixel group ...
  const int bg = wave / PW;                 // ... and output-block group inside the workgroup
  const int nb0 = grp * (NB * CW);          // first block staged by the workgroup
  const int nbw0 = nb0 + bg * NB;           // first block of this wave
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
  const long plane_in = (long)p.H * p.W * 8;

  const T* xb = reinterpret_cast<const T*>(p.x) + (long)b * p.x_is;
  const T* wb = reinterpret_cast<const T*>(p.w);

  // ---- staging plan: LDS-DMA ring ----
  // Every chunk (16 input channels: the halo tile of two octet planes + the 9-tap weight block)
  // goes global -> LDS by buffer_load ... lds, 16 bytes per lane, into one of NSTG ring stages;
  // nothing passes through VGPRs.  Threads own consecutive 16-byte slots (the LDS image of a wave
  // instruction is lane-linear); the per-lane SOURCE offset carries the layout.  An out-of-image
  // pixel (zero padding), the second octet of the last chunk when Cin % 16 != 0, and the pad
  // slots get an out-of-range offset, for which the hardware deposits zeros in LDS.  No load is
  // conditional, so each wave issues exactly DMA_PER_CHUNK instructions per chunk and a counted
  // s_waitcnt vmcnt() retires exactly one chunk while the next stays in flight across the barrier.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned plane_bytes = (unsigned)(plane_in * sizeof(T));
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(xb), 0, (int)((unsigned)p.cin8 * plane_bytes), 0x00020000);
  const unsigned wchunk_bytes = TAPS * 512 * sizeof(T);
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(wb + (long)nb0 * p.nchunk * (TAPS * 512)), 0,
      (int)((unsigned)(NB * CW) * p.nchunk * wchunk_bytes), 0x00020000);
  // Set-up is ordered by what the first staging instructions need: a wave issues one instruction
  // every few cycles and nothing overlaps this stretch, so every instruction in front of the
  // first LDS-DMA is paid in full by every launch.  All selects are branch-free.
  unsigned off[DMA_PER_CHUNK];
  auto w_off = [&](unsigned piece) -> unsigned {      // piece: slot index inside the weight region
    const unsigned nb = piece / (TAPS * 64 * PPO);
    const unsigned rem = piece - nb * (TAPS * 64 * PPO);
    return piece < (unsigned)WP ? nb * (unsigned)p.nchunk * wchunk_bytes + rem * 16u : OOB;
  };
  auto x_off = [&](unsigned piece) -> unsigned {      // piece: slot index inside the activation region
    const unsigned q = piece % PPO;
    const unsigned hp = piece / PPO;                  // (k half, pixel)
    const unsigned h = hp / NPIX;
    const unsigned pix = hp - h * NPIX;
    const unsigned r = pix / IW, c = pix - r * IW;
    const int gy = iy0 + (int)r, gx = ix0 + (int)c;
    // unsigned compares fold the >= 0 tests in
    const bool ok = (piece < (unsigned)XP) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
    const unsigned o = h * plane_bytes + (unsigned)((gy * p.W + gx) * 8 + (int)(q * EPP)) * (unsigned)sizeof(T);
    // out-of-image slots stay out of range for every chunk: OOB minus the largest chunk offset
    return ok ? o : OOB;
  };
#pragma unroll
  for (int j = 0; j < DMA_PER_CHUNK; ++j) {
    const unsigned piece = (unsigned)tid + j * THREADS;
    if ((j + 1) * THREADS <= XPP) off[j] = x_off(piece);
    else if (j * THREADS >= XPP) off[j] = w_off(piece - XPP);
    else off[j] = piece < (unsigned)XPP ? x_off(piece) : w_off(piece - XPP);
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int chunk, int stage) {
    conv_issue_chunk<T, DMA_PER_CHUNK, XPP, THREADS>(xrsrc, wrsrc, off, (unsigned)chunk * 2u * plane_bytes,
                                                     (unsigned)chunk * wchunk_bytes,
                                                     reinterpret_cast<char*>(smem + stage * STAGE), wave_u);
  };

  // The bias is the accumulators' initial value: a lane's 16 accumulator registers of block n
  // are the packed-bias floats [(n*2+hh)*16, +16) (the packer's row order).  Issued in front of
  // the staging instructions, so they are the wave's oldest memory operations: the wait the
  // compiler puts in front of their first use then leaves every staged chunk in flight.
  f32x4 bias_v[NB][4];
  if (p.bias) {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        bias_v[n][k] = *reinterpret_cast<const f32x4*>(p.bias + ((nbw0 + n) * 2 + hh) * 16 + k * 4);
  }

  // first chunks on their way before anything else is set up
#pragma unroll
  for (int j = 0; j < DIST; ++j)
    if (j < p.nchunk) issue(j, j);

  // LDS slot (in octets) of this lane's pixel for row group rg at tap (0,0)
  int xslot[RW];
#pragma unroll
  for (int rg = 0; rg < RW; ++rg) {
    int g = pg * RW + rg;
    int py = g * RPG + (pl >> LTW), px = pl & (TW - 1);
    xslot[rg] = hh * NPIX + py * STRIDE * IW + px * STRIDE;
  }

  // ---- epilogue addressing and operand prefetch state ----
  // A convolution either masks its output by an activation derivative (data gradients inside
  // a chain) or adds residuals (block outputs), in practice never both: each of the two
  // common forms prefetches all its epilogue operands in one burst while the last chunk is
  // being multiplied; the combination takes the plain path after the loop.
  const long plane_out = (long)p.OH * p.OW * 8;
  const bool all_octs = (nbw0 + NB) * 4 <= p.cout8;
  const bool fast_mask = all_octs && p.mask && !p.res1 && !p.res2 && !p.ps;
  const bool fast_res = all_octs && !p.mask && p.res1 && !p.ps;
  typedef typename Oct<T>::vec oct_t;
  long eoff[RW];                       // element offset of (octet plane 0, this lane's pixel); -1: outside
#pragma unroll
  for (int rg = 0; rg < RW; ++rg) {
    const int g = pg * RW + rg;
    const int oy = oy0 + g * RPG + (pl >> LTW), ox = ox0 + (pl & (TW - 1));
    eoff[rg] = (oy < p.OH && ox < p.OW) ? ((long)oy * p.OW + ox) * 8 : -1;
  }
  oct_t pre_a[RW][NB][2];              // mask or res1; re-used for res2
  auto prefetch = [&](const T* base_a) {
#pragma unroll
    for (int rg = 0; rg < RW; ++rg)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int oct = (nbw0 + n) * 4 + 2 * j + hh;
          if (eoff[rg] < 0 || oct >= p.cout8) continue;
          const long off = (long)oct * plane_out + eoff[rg];
          pre_a[rg][n][j] = *reinterpret_cast<const oct_t*>(base_a + off);
        }
  };

  f32x16 acc[RW][NB];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[r][n][i] = p.bias ? bias_v[n][i >> 2][i & 3] : 0.f;

  int stage = 0;
  for (int chunk = 0; chunk < p.nchunk; ++chunk) {
    // this chunk's DMA has landed (the DIST-1 younger chunks may stay in flight) ...
    if (DIST > 1 && chunk + DIST - 1 < p.nchunk)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * DMA_PER_CHUNK) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // ... in every wave; and every wave is done with the previous stage
    if (chunk + DIST < p.nchunk) issue(chunk + DIST, stage == 0 ? NSTG - 1 : stage - 1);
    if (chunk + 1 == p.nchunk && (fast_mask || fast_res))   // this chunk waited for vmcnt(0): nothing counts after it
      prefetch(reinterpret_cast<const T*>(fast_mask ? p.mask : p.res1) + (long)b * (fast_mask ? p.m_is : p.r1_is));
    const frag_t* sx = reinterpret_cast<const frag_t*>(smem + stage * STAGE);
    const frag_t* sw = reinterpret_cast<const frag_t*>(smem + stage * STAGE + XPP);
    // Fragment reads run one tap ROW (KS taps) ahead of the MFMAs that consume them: with one
    // wave per SIMD nothing else hides the LDS latency.
    if constexpr (XROWS) {
      // A row group is one image row here (32-pixel-wide tiles, stride 1), so the input row that
      // tap row dy feeds to row group rg is row rg + dy of the wave's part of the halo tile: the
      // fragment of (dy, rg) is the fragment of (dy - 1, rg + 1).  Each of the RW + KS - 1 input
      // rows is read once per chunk into a register window and used by every tap row that
      // touches it: (RW + 2) * 3 activation reads per chunk instead of RW * 9.
      frag_t wf[2][KS][NB], xw[RW + KS - 1][KS];
      auto read_w = [&](int dy, int buf) {
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            wf[buf][dx][n] = sw[(((bg * NB + n) * TAPS + dy * KS + dx) * 2 + hh) * 32 + pl];
      };
      auto read_xrow = [&](int r) {
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) xw[r][dx] = sx[xslot[0] + r * IW + dx];
      };
      read_w(0, 0);
#pragma unroll
      for (int r = 0; r < RW; ++r) read_xrow(r);
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        if (dy + 1 < KS) {
          read_w(dy + 1, (dy + 1) & 1);
          read_xrow(dy + RW);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
          for (int rg = 0; rg < RW; ++rg)
#pragma unroll
            for (int n = 0; n < NB; ++n) mma(wf[dy & 1][dx][n], xw[rg + dy][dx], acc[rg][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      frag_t wf[2][KS][NB], xf[2][KS][RW];
      auto read_row = [&](int dy, int buf) {
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) {
          const int tap = dy * KS + dx;
#pragma unroll
          for (int n = 0; n < NB; ++n) wf[buf][dx][n] = sw[(((bg * NB + n) * TAPS + tap) * 2 + hh) * 32 + pl];
#pragma unroll
          for (int rg = 0; rg < RW; ++rg) xf[buf][dx][rg] = sx[xslot[rg] + dy * IW + dx];
        }
      };
      read_row(0, 0);
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        if (dy + 1 < KS) read_row(dy + 1, (dy + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
          for (int rg = 0; rg < RW; ++rg)
#pragma unroll
            for (int n = 0; n < NB; ++n) mma(wf[dy & 1][dx][n], xf[dy & 1][dx][rg], acc[rg][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stage = stage == NSTG - 1 ? 0 : stage + 1;
  }

  // ---- epilogue ----
  // One wave per SIMD issues an instruction every few cycles, so this straight-line tail is
  // priced by its instruction count: every run-time option is tested ONCE per wave (uniform
  // branches around whole passes over the accumulators), never per stored octet.
  const bool full = (nbw0 + NB) * 4 <= p.cout8 || p.ps;        // every octet of these blocks exists
  const bool fast = full && !(p.mask && !fast_mask) && !((p.res1 || p.res2) && !fast_res);
  if (fast) {
    if (fast_mask) {
      const float ms = p.mask_slope;
#pragma unroll
      for (int rg = 0; rg < RW; ++rg)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[rg][n][i] = (float)pre_a[rg][n][i >> 3][i & 7] > 0.f ? acc[rg][n][i] : acc[rg][n][i] * ms;
    }
    if (p.act_slope != 1.f) {          // LeakyReLU / ReLU, slope in [0,1): max(v, slope*v)
      const float as = p.act_slope;
#pragma unroll
      for (int rg = 0; rg < RW; ++rg)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[rg][n][i] = (as >= 0.f && as <= 1.f) ? fmaxf(acc[rg][n][i], acc[rg][n][i] * as)
                                                     : (acc[rg][n][i] > 0.f ? acc[rg][n][i] : acc[rg][n][i] * as);
    }
    if (p.alpha != 1.f) {
#pragma unroll
      for (int rg = 0; rg < RW; ++rg)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[rg][n] *= p.alpha;
    }
    if (fast_res) {
      const float b1 = p.beta1;
#pragma unroll
      for (int rg = 0; rg < RW; ++rg)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[rg][n][i] += b1 * (float)pre_a[rg][n][i >> 3][i & 7];
      if (p.res2) {                    // second residual: one more burst into the same registers
        prefetch(reinterpret_cast<const T*>(p.res2) + (long)b * p.r2_is);
        const float b2 = p.beta2;
#pragma unroll
        for (int rg = 0; rg < RW; ++rg)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rg][n][i] += b2 * (float)pre_a[rg][n][i >> 3][i & 7];
      }
    }
    T* yb = reinterpret_cast<T*>(p.y) + (long)b * p.y_is;
#pragma unroll
    for (int rg = 0; rg < RW; ++rg) {
      if (eoff[rg] < 0) continue;
      long yrow;                       // element offset inside the image's output
      long ostep;                      // element step between consecutive stored octets (j, then n)
      if (p.ps) {                      // PixelShuffle(2): sub-pixel (hh, j) of output channel block nb
        const int g = pg * RW + rg;
        const int oy = oy0 + g * RPG + (pl >> LTW), ox = ox0 + (pl & (TW - 1));
        yrow = (long)nbw0 * 4 * plane_out + ((long)(2 * oy + hh) * (2 * p.OW) + 2 * ox) * 8;
        ostep = 8;
      } else {
        yrow = (long)(nbw0 * 4 + hh) * plane_out + eoff[rg];
        ostep = 2 * plane_out;
      }
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[rg][n][j * 8 + e];
          store_oct(yb + yrow + (p.ps ? (long)n * 4 * plane_out + j * ostep : (long)(n * 2 + j) * ostep), v);
        }
    }
    return;
  }
  // general form (ragged channel counts, mask together with residuals): per stored octet
#pragma unroll
  for (int rg = 0; rg < RW; ++rg) {
    if (eoff[rg] < 0) continue;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int nb = nbw0 + n;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int oct = nb * 4 + 2 * j + hh;
        if (oct >= p.cout8) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[rg][n][j * 8 + e];
        const long off = (long)oct * plane_out + eoff[rg];
        if (p.mask) {
          float m[8];
          load_oct(reinterpret_cast<const T*>(p.mask) + (long)b * p.m_is + off, m);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * p.mask_slope;
        }
        if (p.act_slope != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.act_slope;
        }
        if (p.alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        }
        if (p.res1) {
          float r[8];
          load_oct(reinterpret_cast<const T*>(p.res1) + (long)b * p.r1_is + off, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += p.beta1 * r[e];
        }
        if (p.res2) {
          float r[8];
          load_oct(reinterpret_cast<const T*>(p.res2) + (long)b * p.r2_is + off, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += p.beta2 * r[e];
        }
        store_oct(reinterpret_cast<T*>(p.y) + (long)b * p.y_is + off, v);
      }
    }
  }
}

// SRX_CONV_REMAP=0 keeps the launch-order tile map (A/B timing).
static bool remap_off() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SRX_CONV_REMAP");
    v = (e && atoi(e) == 0) ? 1 : 0;
  }
  return v == 1;
}

template <typename T, int STRIDE, int NB, int RW, int LTW, int KS, int WAVES = 4, int CW = 1>
int launch(const ConvParams& p0, int B, hipStream_t stream) {
  typedef StageGeom<T, STRIDE, NB, RW, LTW, KS, WAVES, CW> SG;
  constexpr int THREADS = SG::THREADS;
  constexpr int TW = 1 << LTW;
  constexpr int TH = SG::TH;
  constexpr int LDS = RingDepth<T, NB, RW, WAVES, CW>::value * SG::STAGE * 16;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  ConvParams p = p0;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + TH - 1) / TH;
  p.ngroups = p.nblk / (NB * CW);
  auto kern = conv_kernel<T, STRIDE, NB, RW, LTW, KS, WAVES, CW>;
  static srx_per_device_once attr_set;     // per instantiation, per device
  if (int rc = srx_ensure_dynamic_lds(&attr_set, reinterpret_cast<const void*>(kern), LDS)) return rc;
  const long gx = (long)p.tiles_x * p.ngroups;
  p.remap_chunk = 0; p.lgx = p.lgy = 0;
  {
    auto lg2 = [](long v) { int l = 0; while ((1L << l) < v) ++l; return (1L << l) == v ? l : -1; };
    const int lx = lg2(gx), ly = lg2(p.tiles_y);
    const long total = gx * p.tiles_y * B;
    // (the division form costs every workgroup some sixty instructions in front of its first
    //  load: only where a workgroup's lifetime dwarfs that)
    if (total < (1L << 31) && !remap_off() && ((lx >= 0 && ly >= 0 && total >= 64) || total >= 1024)) {
      p.remap_chunk = (unsigned)(total / 8);
      p.lgx = (lx >= 0 && ly >= 0) ? lx : -1;
      p.lgy = ly;
    }
  }
  SRX_CHECK_ARG(gx > 0 && gx < (1L << 31) && p.tiles_y > 0 && p.tiles_y <= 65535 && B <= 65535,
                "conv grid out of range (%ld x %d x %d)", gx, p.tiles_y, B);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)p.tiles_y, (unsigned)B), dim3(THREADS), LDS, stream, p);
  return srx_check_launch("conv_kernel");
}

// SRX_CONV_WAVES=8 selects the eight-wave workgroup shape for the 32-wide bf16 3x3 stride-1 tiles (tuning).
static int forced_waves() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SRX_CONV_WAVES");
    v = e ? atoi(e) : 0;
  }
  return v;
}

template <typename T>
int dispatch(const srx_conv_desc* d, const ConvParams& p, hipStream_t s) {
  const bool nb2 = (p.nblk % 2) == 0;
  // Small feature maps (16 x 8 and 8 x 16-pixel tiles): 64 output channels per workgroup only while
  // that gives every CU a workgroup; otherwise 32, twice the workgroups (SRX_CONV_SMALL_NB1=0: always 64)
  auto nb2s = [&](int ltw) {
    if (!nb2) return false;
    static const bool off = [] { const char* e = getenv("SRX_CONV_SMALL_NB1"); return e && atoi(e) == 0; }();
    if (off) return true;
    const int tw = 1 << ltw, th = 4 * (32 >> ltw);
    const long wgs = (long)((p.OW + tw - 1) / tw) * ((p.OH + th - 1) / th) * d->B * (p.nblk / 2);
    return wgs >= 256;
  };
  if (d->ksize == 1) {
    return nb2 ? launch<T, 1, 2, 1, 5, 1>(p, d->B, s) : launch<T, 1, 1, 1, 5, 1>(p, d->B, s);
  }
  if (d->stride == 1) {
    if (p.OW >= 24) {
      if constexpr (sizeof(T) == 2) {
        // Eight waves per workgroup (two per SIMD) on the same 32x8 tile: measured no faster per
        // 16-channel chunk than four (the loop is bound by the L2 -> LDS staging rate, not by
        // issue), and slower next to a co-resident weight-gradient workgroup; kept selectable.
        if (forced_waves() == 8)
          return nb2 ? launch<T, 1, 1, 2, 5, 3, 8, 2>(p, d->B, s) : launch<T, 1, 1, 1, 5, 3, 8, 1>(p, d->B, s);
      }
      if (p.OH * p.OW >= 128 * 128 && !nb2) return launch<T, 1, 1, 4, 5, 3>(p, d->B, s);
      // too few 32x8 tiles to give every CU a workgroup: 32x4 tiles
      const long wgs8 = (long)((p.OW + 31) / 32) * ((p.OH + 7) / 8) * d->B * (p.nblk / (nb2 ? 2 : 1));
      if (wgs8 < 256) return nb2 ? launch<T, 1, 2, 1, 5, 3>(p, d->B, s) : launch<T, 1, 1, 1, 5, 3>(p, d->B, s);
      return nb2 ? launch<T, 1, 2, 2, 5, 3>(p, d->B, s) : launch<T, 1, 1, 2, 5, 3>(p, d->B, s);
    }
    if (p.OW >= 12) return nb2s(4) ? launch<T, 1, 2, 1, 4, 3>(p, d->B, s) : launch<T, 1, 1, 1, 4, 3>(p, d->B, s);
    return nb2s(3) ? launch<T, 1, 2, 1, 3, 3>(p, d->B, s) : launch<T, 1, 1, 1, 3, 3>(p, d->B, s);
  }
  if (p.OW >= 24) return nb2 ? launch<T, 2, 2, 1, 5, 3>(p, d->B, s) : launch<T, 2, 1, 1, 5, 3>(p, d->B, s);
  if (p.OW >= 12) return nb2s(4) ? launch<T, 2, 2, 1, 4, 3>(p, d->B, s) : launch<T, 2, 1, 1, 4, 3>(p, d->B, s);
  return nb2s(3) ? launch<T, 2, 2, 1, 3, 3>(p, d->B, s) : launch<T, 2, 1, 1, 3, 3>(p, d->B, s);
}

}  // namespace

extern "C" int srx_conv2d(const srx_conv_desc* d, void* stream) {
  SRX_CHECK_ARG(d != nullptr, "srx_conv2d: null descriptor");
  SRX_CHECK_ARG(d->dtype == SRX_BF16 || d->dtype == SRX_F32, "srx_conv2d: bad dtype %d", d->dtype);
  SRX_CHECK_ARG(d->ksize == 3 || d->ksize == 1, "srx_conv2d: ksize must be 3 or 1 (got %d)", d->ksize);
  SRX_CHECK_ARG(d->stride == 1 || (d->stride == 2 && d->ksize == 3),
                "srx_conv2d: stride must be 1, or 2 with ksize 3 (got %d)", d->stride);
  SRX_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->cin > 0 && d->cout > 0,
                "srx_conv2d: non-positive size B=%d H=%d W=%d cin=%d cout=%d", d->B, d->H, d->W, d->cin, d->cout);
  SRX_CHECK_ARG(d->x && d->y && d->w, "srx_conv2d: null x/y/w pointer");
  SRX_CHECK_ARG(!d->pixel_shuffle || (d->cout % 32 == 0 && !d->mask && !d->res1 && !d->res2 && d->alpha == 1.f),
                "srx_conv2d: pixel_shuffle needs cout %% 32 == 0 and no mask/residual/alpha");
  SRX_CHECK_ARG(((uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->w) % 16 == 0,
                "srx_conv2d: x/y/w must be 16-byte aligned");
  // one image's input planes are addressed through a 32-bit buffer descriptor (offsets at and above
  // 0xFFFFFFF0 are the "deposit zeros" marker): larger inputs must be tiled by the caller
  const long x_bytes = (long)((d->cin + 7) / 8) * d->H * d->W * 8 * (d->dtype == SRX_BF16 ? 2 : 4);
  SRX_CHECK_ARG(x_bytes < 0xFFFFFFF0L,
                "srx_conv2d: one image's input (%ld bytes) exceeds the 4 GB a buffer descriptor addresses; "
                "tile the image", x_bytes);
  ConvParams p;
  p.x = d->x; p.x_is = d->x_img_stride;
  p.y = d->y; p.y_is = d->y_img_stride;
  p.w = d->w; p.bias = d->bias;
  p.res1 = d->res1; p.r1_is = d->res1_img_stride;
  p.res2 = d->res2; p.r2_is = d->res2_img_stride;
  p.mask = d->mask; p.m_is = d->mask_img_stride;
  p.alpha = d->alpha; p.beta1 = d->beta1; p.beta2 = d->beta2;
  p.act_slope = d->act_slope; p.mask_slope = d->mask_slope;
  p.H = d->H; p.W = d->W;
  const int pad = d->ksize / 2;
  p.OH = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  p.OW = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  p.cin8 = (d->cin + 7) / 8;
  p.nchunk = (d->cin + 15) / 16;
  p.nblk = (d->cout + 31) / 32;
  p.cout8 = d->pixel_shuffle ? d->cout / 32 : (d->cout + 7) / 8;
  p.ps = d->pixel_shuffle;
  p.tiles_x = p.tiles_y = p.ngroups = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return d->dtype == SRX_BF16 ? dispatch<bf16>(d, p, s) : dispatch<float>(d, p, s);
}
