// Tiled inference around the generator (reference inference.py:187-230): cut overlapping LR
// tiles out of a reflect-padded image, and blend the super-resolved tiles back with a
// separable window, normalise, crop and clamp.  Both kernels are pure data movement
// (HBM-bound): one 16-byte octet per lane on the tile side, coalesced fp32 rows on the image side.
#include "srx_common.h"

namespace {

constexpr int TPB = 256;

// F.pad(..., 'reflect') on the bottom / right edge only (inference.py:199): padded index -> source.
__device__ __forceinline__ int reflect(int i, int n) { return i < n ? i : 2 * (n - 1) - i; }

template <typename T>
__global__ void tile_gather_kernel(const float* __restrict__ img, int C, int h, int w,
                                   const srx_tile* __restrict__ tiles, int th, int tw,
                                   T* __restrict__ dst, long dst_is) {
  const srx_tile t = tiles[blockIdx.y];
  const int npx = th * tw;
  T* out = dst + (long)blockIdx.y * dst_is;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    const int r = i / tw, c = i % tw;
    const int sy = reflect(t.y0 + r, h), sx = reflect(t.x0 + c, w);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < C; ++ch) v[ch] = img[((long)ch * h + sy) * w + sx];
    store_oct(out + (long)i * 8, v);
  }
}

// One thread per output pixel: sum over the (few) tiles that cover it, in the reference's
// row-major tile order, of value * window and of window; out = sum / (wsum + 1e-8).
template <typename T>
__global__ void tile_blend_kernel(const T* __restrict__ store, const srx_tile* __restrict__ tiles,
                                  int ny, int nx, int stride, int tile, int scale,
                                  const float* __restrict__ window, float* __restrict__ out,
                                  uint8_t* __restrict__ out_u8, int C, int OH, int OW, int clamp01) {
  const long n = (long)OH * OW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int Y = (int)(i / OW), X = (int)(i % OW);
    const int ly = Y / scale, lx = X / scale;
    const int ky1 = min(ny - 1, ly / stride), kx1 = min(nx - 1, lx / stride);
    const int ky0 = max(0, (ly - tile + stride) / stride), kx0 = max(0, (lx - tile + stride) / stride);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float wsum = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) {
        const srx_tile t = tiles[ky * nx + kx];
        const int ry = Y - t.y0 * scale, rx = X - t.x0 * scale;
        if (ry < 0 || rx < 0 || ry >= t.th * scale || rx >= t.tw * scale) continue;
        const float wgt = window[ry] * window[rx];
        float v[8];
        load_oct(store + t.offset + ((long)ry * (t.tw * scale) + rx) * 8, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += v[c] * wgt;
        wsum += wgt;
      }
    const float inv = 1.f / (wsum + 1e-8f);
    for (int c = 0; c < C; ++c) {
      float v = acc[c] * inv;
      if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
      if (out) out[(long)c * n + i] = v;
      if (out_u8) out_u8[(long)c * n + i] = (uint8_t)(fminf(fmaxf(v, 0.f), 1.f) * 255.f);
    }
  }
}

inline int grid_for(long n, int cap = 4096) {
  long g = (n + TPB - 1) / TPB;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int srx_tile_gather(int dtype, const float* img, int C, int h, int w, const srx_tile* tiles,
                               int count, int th, int tw, void* dst, int64_t dst_img_stride, void* stream) {
  SRX_CHECK_ARG(img && tiles && dst, "srx_tile_gather: null pointer");
  SRX_CHECK_ARG(C > 0 && C <= 8 && h > 1 && w > 1 && count > 0 && th > 0 && tw > 0,
                "srx_tile_gather: bad size C=%d h=%d w=%d count=%d th=%d tw=%d", C, h, w, count, th, tw);
  SRX_CHECK_ARG(dst_img_stride >= (int64_t)th * tw * 8, "srx_tile_gather: image stride too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for((long)th * tw, 1024), count);
  if (dtype == SRX_BF16)
    hipLaunchKernelGGL(tile_gather_kernel<bf16>, grid, dim3(TPB), 0, s, img, C, h, w, tiles, th, tw, (bf16*)dst,
                       (long)dst_img_stride);
  else if (dtype == SRX_F32)
    hipLaunchKernelGGL(tile_gather_kernel<float>, grid, dim3(TPB), 0, s, img, C, h, w, tiles, th, tw, (float*)dst,
                       (long)dst_img_stride);
  else
    SRX_CHECK_ARG(false, "srx_tile_gather: bad dtype %d", dtype);
  return srx_check_launch("tile_gather");
}

extern "C" int srx_tile_blend(int dtype, const void* store, const srx_tile* tiles, int ny, int nx, int stride,
                              int tile, int scale, const float* window, float* out, uint8_t* out_u8, int C,
                              int OH, int OW, int clamp01, void* stream) {
  SRX_CHECK_ARG(store && tiles && window && (out || out_u8), "srx_tile_blend: null pointer");
  SRX_CHECK_ARG(ny > 0 && nx > 0 && stride > 0 && tile >= stride && scale > 0 && C > 0 && C <= 8 && OH > 0 && OW > 0,
                "srx_tile_blend: bad size ny=%d nx=%d stride=%d tile=%d scale=%d C=%d OH=%d OW=%d", ny, nx,
                stride, tile, scale, C, OH, OW);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for((long)OH * OW, 16384));
  if (dtype == SRX_BF16)
    hipLaunchKernelGGL(tile_blend_kernel<bf16>, grid, dim3(TPB), 0, s, (const bf16*)store, tiles, ny, nx, stride,
                       tile, scale, window, out, out_u8, C, OH, OW, clamp01);
  else if (dtype == SRX_F32)
    hipLaunchKernelGGL(tile_blend_kernel<float>, grid, dim3(TPB), 0, s, (const float*)store, tiles, ny, nx, stride,
                       tile, scale, window, out, out_u8, C, OH, OW, clamp01);
  else
    SRX_CHECK_ARG(false, "srx_tile_blend: bad dtype %d", dtype);
  return srx_check_launch("tile_blend");
}

// ---------------------------------------------------------------------------------------------
// Training-batch former (reference utils/datasets.py:77-114): synchronized crop of a decoded
// uint8 HWC image, hflip, vflip, quarter-turn rotation, ToTensor (/255) and Normalize
// ((x-0.5)/0.5), written as one fp32 NCHW batch entry.  One thread per output pixel walks the
// augmentation chain backwards to its source pixel.
namespace {

__global__ void crop_augment_kernel(const srx_crop* __restrict__ crops, int S, float* __restrict__ out) {
  const srx_crop c = crops[blockIdx.y];
  const int n = S * S;
  float* o = out + (long)blockIdx.y * 3 * n;
  const int turns = (c.flags >> 2) & 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int y = i / S, x = i % S;
    for (int t = 0; t < turns; ++t) {      // undo one counter-clockwise quarter turn
      const int ny = x, nx = S - 1 - y;
      y = ny; x = nx;
    }
    if (c.flags & 2) y = S - 1 - y;        // vflip
    if (c.flags & 1) x = S - 1 - x;        // hflip
    const uint8_t* px = c.img + ((long)(c.y0 + y) * c.W + (c.x0 + x)) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float v = __fdiv_rn((float)px[ch], 255.0f);
      o[(long)ch * n + i] = __fdiv_rn(v - 0.5f, 0.5f);
    }
  }
}

}  // namespace

extern "C" int srx_crop_augment(const srx_crop* crops, int B, int size, float* out, void* stream) {
  SRX_CHECK_ARG(crops && out && B > 0 && size > 0, "srx_crop_augment: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(grid_for((long)size * size, 256), B);
  hipLaunchKernelGGL(crop_augment_kernel, grid, dim3(TPB), 0, s, crops, size, out);
  return srx_check_launch("crop_augment");
}
