/*
 * srx.h — C ABI of libsrx, the MI355X (gfx950) op layer for the ESRGAN hot path.
 *
 * Every entry point takes raw device pointers, explicit sizes and a hipStream_t
 * (passed as void*), launches asynchronously on that stream, allocates nothing
 * and returns an int status (SRX_OK == 0).  After a non-zero return,
 * srx_last_error() describes the failure.  No C++ or torch types cross this
 * boundary; the caller (PyTorch's caching allocator in the Python host layer)
 * owns every buffer, including workspaces.
 *
 * Activation layout ("C8"): a tensor with C channels is stored as
 *     [B][ceil(C/8)][H][W][8]   (channel c -> octet plane c/8, slot c%8)
 * in bf16 or fp32.  A *view* is a pointer to (image 0, first octet plane of the
 * view) plus the image stride in elements; the plane stride is always H*W*8.
 * Because a dense block's growth channels are appended planes, the reference's
 * torch.cat calls (esrgan/generator.py:19-22) need no copy in this layout.
 *
 * Each declaration cites the reference code it replaces.
 */
#ifndef SRX_H
#define SRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_OK 0
#define SRX_ERR_INVALID 1   /* bad argument / unsupported shape */
#define SRX_ERR_HIP 2       /* a HIP runtime call failed        */

#define SRX_BF16 0
#define SRX_F32 1

int srx_version(void);
const char* srx_last_error(void);
/* sizeof() of the ABI structs, for binding self-checks: 0 srx_conv_desc, 1 srx_wgrad_task,
 * 2 srx_wgrad_desc, 3 srx_pack_seg, 4 srx_pack_layer, 5 srx_unpack_item, 6 srx_sn_layer,
 * 7 srx_sn_bwd_layer, 8 srx_generic_args, 9 srx_op, 10 srx_tile, 11 srx_crop, 12 srx_rdb_desc; -1 for an
 * unknown id. */
int srx_sizeof(int which);
/* Host-side self-check (no GPU call): kernel attributes such as the dynamic LDS limit are set
 * once per DEVICE a kernel is launched on, not once per process.  0 = ok.  No counterpart in the
 * reference (single device, config.py:54). */
int srx_selftest_per_device_once(void);

/* ------------------------------------------------------------------ layout */
/* NCHW fp32 -> C8 (dtype), zero-filling the padding slots of the last octet.
 * Replaces nothing in the reference (it is NCHW throughout); this is the edge
 * of the drop-in boundary: GeneratorESRGAN.forward input, esrgan/generator.py:65. */
int srx_nchw_to_c8(int dtype, const float* src, void* dst, int64_t dst_img_stride,
                   int B, int C, int H, int W, void* stream);
/* C8 (dtype) -> NCHW fp32 (first C channels).  esrgan/generator.py:72 output. */
int srx_c8_to_nchw(int dtype, const void* src, int64_t src_img_stride, float* dst,
                   int B, int C, int H, int W, void* stream);

/* ----------------------------------------------------------- weight packing */
/* One virtual convolution layer to pack.  Its weight W_v[n][c][tap] is gathered
 * from one or more fp32 PyTorch-layout source tensors ([Cout][Cin][k][k]):
 * forward layers use one segment; the backward-data "reversed dense block"
 * layers stack the transposed, tap-flipped weights of several convolutions. */
typedef struct srx_pack_seg {
  const float* src;    /* source weight tensor [cout_src][cin_src][k*k]            */
  int32_t cin_src;
  int32_t cout_src;
  int32_t c_begin;     /* first virtual input channel covered by this segment      */
  int32_t c_count;
  int32_t src_off;     /* transposed: first source cin; else first source cin      */
  int32_t src_o_off;   /* transposed: first source cout (maps to c); else 0        */
  int32_t transposed;  /* 1: W_v[n][c][t] = src[src_o_off + c-c_begin][src_off+n][KK-1-t] */
  float scale;
} srx_pack_seg;

typedef struct srx_pack_layer {
  void* dst_w;          /* packed weights [nblk][nchunk][taps][2][32][8] (dtype)    */
  float* dst_bias;      /* packed bias [nblk][2][16] fp32, or NULL                  */
  const float* src_bias;/* fp32 [cout] or NULL (zeros)                              */
  int32_t cout;         /* real virtual Cout                                        */
  int32_t cin;          /* real virtual Cin                                         */
  int32_t nblk;         /* ceil(cout_rows/32)                                       */
  int32_t nchunk;       /* ceil(cin/16)                                             */
  int32_t taps;         /* 9 or 1                                                   */
  int32_t perm;         /* 0: plain octet order; 1: PixelShuffle(2) order           */
  int32_t seg_begin;
  int32_t seg_count;
} srx_pack_layer;

/* layers/segs are DEVICE arrays.  max_elems = max over layers of
 * nblk*nchunk*taps*512.  Packs nn.Conv2d parameters (esrgan/generator.py:10-14,
 * 43,54-62; esrgan/discriminator.py:12,30,32; VGG19 convs, utils/loss.py:15). */
int srx_pack_weights(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs,
                     int nlayers, int64_t max_elems, void* stream);

/* -------------------------------------------------------------- convolution */
typedef struct srx_conv_desc {
  int32_t dtype;        /* SRX_BF16 | SRX_F32 (activations and packed weights)      */
  int32_t ksize;        /* 3 (pad 1) or 1 (pad 0)                                   */
  int32_t stride;       /* 1 or 2                                                   */
  int32_t B, H, W;      /* input batch / spatial size                               */
  int32_t cin;          /* input channels (view holds ceil(cin/8) planes)           */
  int32_t cout;         /* output channels of the convolution                       */
  int32_t pixel_shuffle;/* 1: store PixelShuffle(2)(conv) — weights packed perm=1   */
  const void* x;  int64_t x_img_stride;
  void* y;        int64_t y_img_stride;
  const void* w;        /* packed by srx_pack_weights                               */
  const float* bias;    /* packed bias or NULL                                      */
  /* epilogue, applied in this order to v = acc + bias:
   *   v *= (mask > 0 ? 1 : mask_slope)          if mask != NULL
   *   v  = v > 0 ? v : v * act_slope             (act_slope == 1: identity)
   *   v  = alpha*v + beta1*res1 + beta2*res2     (NULL residuals skipped)      */
  const void* mask; int64_t mask_img_stride; float mask_slope;
  float act_slope;
  float alpha;
  const void* res1; int64_t res1_img_stride; float beta1;
  const void* res2; int64_t res2_img_stride; float beta2;
} srx_conv_desc;

/* Implicit-GEMM convolution on MFMA.  Forward of nn.Conv2d + LeakyReLU/ReLU +
 * residual scaling + PixelShuffle (esrgan/generator.py:18-23,37,48,66-72;
 * esrgan/discriminator.py:12-13,30-32; VGG conv+ReLU, utils/loss.py:15), and,
 * with transposed/flipped packed weights, the data gradient of the same. */
int srx_conv2d(const srx_conv_desc* d, void* stream);

/* ------------------------------------------------- fused dense block (forward) */
/* ResidualDenseBlock.forward (esrgan/generator.py:17-23) — five convolutions over the growing
 * channel stack, LeakyReLU(0.2) on the first four, conv5 * 0.2 + x — in ONE launch, optionally
 * with the enclosing RRDB's out * 0.2 + x (:37) folded into the same epilogue:
 *     y = alpha * conv5(cat(x, x1..x4)) + beta1 * x + beta2 * res2.
 * buf is the block's C8 buffer of nf + 4 gc channels: planes [0, nf/8) hold x on entry; with
 * store_growth the growth slices x1..x4 are written to the planes behind them (the backward pass
 * reads them), without it only y is written (inference).  w / bias: the five layers packed by
 * srx_pack_weights (perm 0), laid out back to back in conv1..conv5 order (52 chunks of 9 x 512
 * elements; 4 x 32 + 64 bias floats).  bf16, nf = 64, gc = 32 only; same results, bit for bit, as
 * five srx_conv2d calls. */
typedef struct srx_rdb_desc {
  int32_t dtype;         /* SRX_BF16                                                   */
  int32_t B, H, W;
  int32_t nf, gc;        /* 64, 32                                                     */
  int32_t store_growth;  /* 1: x1..x4 written to buf planes [nf/8, (nf+4gc)/8)           */
  int32_t tile;          /* output tile per workgroup: 0 = 16 x 8 pixels, 1 = 8 x 8 (twice the
                          * workgroups for a small feature map, a fifth more arithmetic),
                          * 2 = two 16 x 8 tiles stacked vertically, computed one after the
                          * other from one staged input region (half the workgroups)      */
  void* buf; int64_t buf_img_stride;
  void* y;   int64_t y_img_stride;
  const void* w;
  const float* bias;
  float act_slope;       /* 0.2                                                        */
  float alpha;
  float beta1;
  float beta2;
  const void* res2; int64_t res2_img_stride;   /* NULL: no second residual             */
  const void* act;  int64_t act_img_stride;    /* srx_rdb_backward only: the block's buffer as the
                                                * forward pass left it (x | x1..x4)      */
} srx_rdb_desc;
int srx_rdb_forward(const srx_rdb_desc* d, void* stream);
/* The data gradient of the same block (autograd's backward of esrgan/generator.py:17-23 with
 * respect to the activations) in ONE launch.  buf is the block's gradient buffer
 * [gy1 | gy2 | gy3 | gy4 | g_out] (4 x gc/8 + nf/8 planes): planes [4 gc/8, ...) hold the gradient of
 * the block output on entry; the kernel writes gy_j = LeakyReLU'(x_j) * (gradient reaching growth
 * slice j) into the planes in front of them (the weight-gradient launches read them) and
 *     y = (gradient reaching the block input through the five convolutions) + beta1 * g_out + beta2 * res2
 * (beta1: the block's identity path; res2: the gradient of the enclosing RRDB's skip path).
 * act: the block's activation buffer, whose growth slices give the derivative masks.  w: the five
 * data-gradient layers (stacked, transposed, tap-flipped weights of convolutions j+1..5 for
 * j = 4, 3, 2, 1, 0; the 0.2 of the residual scaling folded in) packed by srx_pack_weights back
 * to back in that order (4, 6, 8, 10, 2 x 12 chunks); bias, alpha and store_growth are not used.
 * Same results, bit for bit, as the five srx_conv2d calls it replaces. */
int srx_rdb_backward(const srx_rdb_desc* d, void* stream);

/* --------------------------------------------------------- weight gradients */
/* One 64x64 (cin x cout) block task of a weight-gradient launch: four 32x32
 * sub-blocks (cin half a, cout half b), each accumulated into a [taps][Cout][Cin]
 * fp32 "tap-major" gradient tensor with atomic adds. */
typedef struct srx_wgrad_task {
  const void* x;         /* image-0 base of this task's input view  (NULL: desc.x)  */
  const void* gy;        /* image-0 base of its gradient view       (NULL: desc.gy) */
  int32_t x_oct;         /* first input octet plane (relative to x view)            */
  int32_t gy_oct;        /* first output-gradient octet plane (relative to gy view) */
  int32_t x_noct;        /* valid x octets from x_oct (<= 8)                        */
  int32_t gy_noct;       /* valid gy octets from gy_oct (<= 8)                      */
  int32_t x_skip;        /* octet planes skipped between the task's first four x octets and its last
                            four (0: eight consecutive planes): lets one task pair two 32-channel
                            groups that are not neighbours in the buffer                         */
  int32_t gy_skip;       /* the same for gy                                         */
  float* out[4];         /* [a + 2*b]: tap-major grad tensor base or NULL to skip   */
  int32_t ld[4];         /* Cin of that tensor                                      */
  int32_t co_n[4];       /* Cout of that tensor                                     */
  int32_t ci0[4];        /* cin index of sub-block row 0 in that tensor             */
  int32_t co0[4];        /* cout index of sub-block col 0 in that tensor            */
  float scale[4];
  float* bias_out[2];    /* per cout half b: fp32 bias-gradient tensor or NULL      */
  int32_t bias_co0[2];   /* cout index of that half's first column                  */
  int32_t bias_n[2];     /* Cout of that tensor                                     */
  float bias_scale[2];
} srx_wgrad_task;

typedef struct srx_wgrad_desc {
  int32_t dtype, ksize, stride;
  int32_t B, H, W;       /* INPUT (x) spatial size; gy is H/stride x W/stride       */
  const void* x;  int64_t x_img_stride;   /* x/gy may be NULL if every task has its own */
  const void* gy; int64_t gy_img_stride;
  const srx_wgrad_task* tasks;   /* DEVICE array                                   */
  int32_t ntasks;
  int32_t nslices;       /* pixel-tile slices per task (grid = ntasks*nslices)      */
  int32_t group;         /* >1: tasks come in consecutive groups of this many that read the same
                            buffers; used to co-locate them on one XCD (speed only)  */
  int32_t tile_walk;     /* 3x3 / stride-1 launches on images at least 24 pixels wide: 1 selects the
                            column-walk kernel (tiles walked down the image, input rows kept in an
                            LDS ring: a third less activation staging, the faster one alone on
                            the chip and beside a convolution chain); 0 the whole-halo-tile kernel */
  int32_t pad_;
  int64_t slice_stride;  /* 0: the pixel slices of a task ADD into the same tensors with fp32 atomics
                            (the order of the adds, hence the last bits, varies from run to run).
                            > 0, the deterministic form: slice s of every task STORES its partial
                            sums at out[..] + s * slice_stride floats (bias_out likewise); the caller
                            provides nslices zero-filled copies of the gradient tensors that far
                            apart and srx_unpack_wgrad adds them up in slice order               */
} srx_wgrad_desc;

/* dW of nn.Conv2d (the autograd of esrgan/generator.py:10-14 etc.). */
int srx_conv2d_wgrad(const srx_wgrad_desc* d, void* stream);

/* tap-major fp32 [taps][Cout][Cin] -> PyTorch layout [Cout][Cin][taps], table driven. */
typedef struct srx_unpack_item {
  const float* src; float* dst; int32_t cout, cin, taps; int32_t accumulate;
  int32_t nslices;       /* > 1: src is the first of nslices partial tensors (srx_wgrad_desc.slice_stride */
  int64_t slice_stride;  /*      floats apart), summed in slice order before the layout change          */
} srx_unpack_item;
int srx_unpack_wgrad(const srx_unpack_item* items /*DEVICE*/, int nitems, int64_t max_elems,
                     void* stream);

/* per-channel sum over B,H,W of a C8 view: out[c] += scale * sum (fp32 atomics).
 * Bias gradient of nn.Conv2d. */
int srx_channel_sum(int dtype, const void* x, int64_t img_stride, int noct, int B, int H, int W,
                    float* out, float scale, void* stream);

/* ------------------------------------------------ bandwidth-bound helpers */
/* inverse of PixelShuffle(2) on a C8 tensor: y[4c+2i+j][h][w] = x[c][2h+i][2w+j]
 * (backward of esrgan/generator.py:44).  x: C ch at 2H x 2W, y: 4C ch at H x W. */
int srx_pixel_unshuffle2(int dtype, const void* x, int64_t x_img_stride, void* y,
                         int64_t y_img_stride, int B, int C, int H, int W, void* stream);
/* z[c][2h][2w] = x[c][h][w], zeros elsewhere (z is 2H x 2W): turns the data
 * gradient of a stride-2 convolution (esrgan/discriminator.py:19,21,23,25)
 * into a stride-1 convolution. */
int srx_zero_stuff2(int dtype, const void* x, int64_t x_img_stride, void* z,
                    int64_t z_img_stride, int B, int noct, int H, int W, void* stream);
/* nn.MaxPool2d(2,2) of VGG19 (utils/loss.py:15) and its backward (gradient to
 * the first maximal element in row-major window order, as ATen does). */
int srx_maxpool2_fwd(int dtype, const void* x, int64_t x_img_stride, void* y,
                     int64_t y_img_stride, int B, int noct, int H, int W, void* stream);
int srx_maxpool2_bwd(int dtype, const void* x, int64_t x_img_stride, const void* gy,
                     int64_t gy_img_stride, void* gx, int64_t gx_img_stride,
                     const void* mask, int64_t mask_img_stride, float mask_slope,
                     int B, int noct, int H, int W, void* stream);
/* nn.AdaptiveAvgPool2d(1) (esrgan/discriminator.py:29).  Output is a C8 tensor
 * of ONE image whose width is the batch: y[0][oct][0][b][8]; backward spreads
 * gy/(H*W), multiplied by (mask>0?1:mask_slope). */
int srx_gap_fwd(int dtype, const void* x, int64_t x_img_stride, void* y, int B, int noct,
                int H, int W, void* stream);
int srx_gap_bwd(int dtype, const void* gy, void* gx, int64_t gx_img_stride,
                const void* mask, int64_t mask_img_stride, float mask_slope,
                int B, int noct, int H, int W, void* stream);
/* out = g * (act_out > 0 ? 1 : slope) over n contiguous elements: backward of the in-place
 * nn.LeakyReLU(0.2) / ReLU given the activation's OUTPUT (esrgan/generator.py:15,45,63). */
int srx_act_bwd(int dtype, const void* g, const void* act_out, void* out, float slope, int64_t n,
                void* stream);
/* y = a*x + b*y over n elements of dtype, any n > 0, 16-byte aligned pointers (RRDB skip
 * add, esrgan/generator.py:37; sums of gradient buffers). */
int srx_axpby(int dtype, const void* x, void* y, float a, float b, int64_t n, void* stream);

/* ------------------------------------------------------------------ losses */
/* nn.L1Loss() mean (esrgan/train_esrgan.py:104) between a C8 prediction and an
 * NCHW fp32 target: loss[0] += mean|p-t| ; if g != NULL, g (C8, same dtype) =
 * gscale * sign(p-t)/N with zero in padding slots. */
int srx_l1_c8_nchw(int dtype, const void* pred, int64_t pred_img_stride, const float* target,
                   float* loss, void* g, int64_t g_img_stride, float gscale,
                   int B, int C, int H, int W, void* stream);
/* L1 between two C8 tensors (VGGLoss, utils/loss.py:22): loss[0] += mean|a-b|;
 * ga (optional) = gscale*sign(a-b)/N * (a>0 ? 1 : mask_slope_a). */
int srx_l1_c8(int dtype, const void* a, int64_t a_img_stride, const void* b,
              int64_t b_img_stride, float* loss, void* ga, int64_t ga_img_stride,
              float gscale, float mask_slope, int B, int C, int H, int W, void* stream);
/* nn.BCEWithLogitsLoss() mean against a constant target (esrgan/train_esrgan.py:
 * 91-92,103).  logits: C8 tensor of one image, width n (slot 0 of each pixel).
 * loss[0] += weight*mean; g (optional, same layout) = gscale*weight*(sigmoid(x)-t)/n. */
int srx_bce_logits(int dtype, const void* logits, int n, float target, float weight,
                   float* loss, void* g, float gscale, void* stream);
/* F.mse_loss sum for PSNR (esrgan/train_esrgan.py:59): out[0] += sum (p-t)^2. */
int srx_sqerr_c8_nchw(int dtype, const void* pred, int64_t pred_img_stride, const float* target,
                      float* out, int B, int C, int H, int W, void* stream);

/* ----------------------------------------------------------- spectral norm */
/* torch.nn.utils.spectral_norm hook (esrgan/discriminator.py:12,30,32; torch
 * spectral_norm.py:92-114) for `n` layers in one launch sequence.  Per layer,
 * W is fp32 [rows][cols]; when do_power_iteration: v = normalize(W^T u),
 * u = normalize(W v) in place (eps 1e-12); always sigma = u.(W v);
 * inv_sigma[l] = 1/sigma.  The normalised weight is never materialised: the
 * packer multiplies by inv_sigma (srx_pack_seg.scale is read from device when
 * scale_ptr is set — see srx_pack_weights_scaled). */
typedef struct srx_sn_layer {
  const float* w; float* u; float* v; float* inv_sigma; float* wv; /* wv: scratch [rows] */
  int32_t rows, cols;
} srx_sn_layer;
int srx_sn_power_iter(const srx_sn_layer* layers /*DEVICE*/, int n, int do_power_iteration,
                      int max_rows, int max_cols, void* stream);
/* srx_sn_power_iter for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the same computation and results, but the per-layer
 * W v scratch rows live in a static device array inside the library instead of in the caller's
 * `wv` buffers (srx_sn_layer.wv is ignored and may be NULL).  No counterpart in the reference.
 * Limit: n * max_rows <= srx_sn_scratch_floats() (65536 floats = 256 KB per GPU, held until the
 * process ends); larger problems are rejected with SRX_ERR_INVALID.  Calls no hipMalloc / hipFree
 * or any allocating routine.  Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; serialise them or use srx_sn_power_iter.
 * Set-up: the array's device address is looked up once per device (hipGetSymbolAddress, which may
 * load the library's code object).  srx_sn_scratch_floats() does that for the current device: call
 * it once per device before the first call that must not touch the runtime (e.g. before a stream
 * capture begins); otherwise the first srx_sn_power_iter_noalloc call on a device does it. */
int srx_sn_power_iter_noalloc(const srx_sn_layer* layers /*DEVICE*/, int n, int do_power_iteration,
                              int max_rows, int max_cols, void* stream);
int srx_sn_scratch_floats(void);
/* dW_orig = dW_sn*inv_sigma - (<dW_sn, W_orig> * inv_sigma^2) u v^T, in place on
 * dw (fp32 [rows][cols]).  Autograd of spectral_norm.py:112-113. */
typedef struct srx_sn_bwd_layer {
  const float* w; const float* u; const float* v; const float* inv_sigma; float* dw;
  float* dot; /* scratch [1] */
  int32_t rows, cols;
} srx_sn_bwd_layer;
/* deterministic != 0: each layer's <dW_sn, W_orig> is summed by one workgroup in a fixed order
 * (same inputs, same bits) instead of by atomic adds; the update itself is elementwise either way. */
int srx_sn_backward(const srx_sn_bwd_layer* layers /*DEVICE*/, int n, int max_elems, int deterministic,
                    void* stream);
/* Same as srx_pack_weights but every segment's scale is multiplied by
 * *scale_ptr[i] (device floats, e.g. inv_sigma) when scale_ptr[i] != NULL. */
int srx_pack_weights_scaled(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs,
                            const float* const* seg_scale_ptr /*DEVICE, per segment*/,
                            int nlayers, int64_t max_elems, void* stream);

/* The same packing driven by a flat DEVICE table with one entry per (layer, unit) pair,
 * entry = layer index << 16 | unit index, unit = block * nchunk + chunk of that layer
 * (0 <= unit < nblk * nchunk; at most 65535 layers and 65536 units per layer): every workgroup
 * that starts has work, which is what a table of many small layers (the generator's 701) needs.
 * The layer indices are relative to `layers`; seg_scale_ptr may be NULL. */
int srx_pack_weights_flat(int dtype, const srx_pack_layer* layers, const srx_pack_seg* segs,
                          const float* const* seg_scale_ptr /*DEVICE, per segment*/,
                          const uint32_t* unit_table /*DEVICE*/, int total_units, void* stream);

/* --------------------------------------------------------------- optimizer */
/* torch.optim.Adam step (esrgan/train_esrgan.py:134-135,97,110) over a flat
 * fp32 parameter range: p,g,m,v of n elements; bias corrections from step. */
int srx_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
             float beta2, float eps, float weight_decay, int step, float gscale, void* stream);

/* -------------------------------------------------------- tiled inference */
/* Tiling around the generator, reference inference.py:187-230 (_infer_tiled).
 * A tile is a th x tw window of the reflect-padded LR image at (y0,x0); its 4x
 * output lives at `offset` elements into a tile store, as one C8 octet plane
 * [th*scale][tw*scale][8]. */
typedef struct srx_tile {
  int32_t y0, x0, th, tw;
  int64_t offset;
} srx_tile;
/* F.pad(lr, [0,pad_w,0,pad_h], 'reflect') + the tile slices lr_padded[:, :, y:y+th, x:x+tw]
 * (inference.py:199,218) for `count` equally sized tiles of ONE fp32 [C][h][w] image, written
 * as a C8 batch [count][1][th][tw][8] (the generator's input layout).  tiles: DEVICE array
 * of the `count` tiles to cut. */
int srx_tile_gather(int dtype, const float* img, int C, int h, int w, const srx_tile* tiles /*DEVICE*/,
                    int count, int th, int tw, void* dst, int64_t dst_img_stride, void* stream);
/* The blend of inference.py:208-230 in gather form: for every pixel of the cropped output
 * [C][OH][OW], sum over the covering tiles (row-major tile order, as the reference's loop)
 * of tile_value*w and of w, w = window[ry]*window[rx]; out = sum/(wsum+1e-8), optionally
 * clamped to [0,1] (inference.py:163); out_u8 (optional) = floor(clamp(v)*255) as
 * to_pil_image does.  tiles: DEVICE [ny][nx] grid with tile (ky,kx) at LR origin
 * (ky*stride, kx*stride); window: DEVICE fp32 [tile*scale]. */
int srx_tile_blend(int dtype, const void* store, const srx_tile* tiles /*DEVICE*/, int ny, int nx,
                   int stride, int tile, int scale, const float* window, float* out, uint8_t* out_u8,
                   int C, int OH, int OW, int clamp01, void* stream);

/* ------------------------------------------------------------ batch former */
/* One sample of the training loader, reference utils/datasets.py:77-114 (TrainDataset.
 * __getitem__): crop [y0,y0+size) x [x0,x0+size) of a decoded uint8 RGB image (HWC, device
 * memory), then hflip (flags bit 0), vflip (bit 1), then (flags>>2)&3 counter-clockwise
 * quarter turns (PIL's rotate by 90/180/270 on a square patch), then ToTensor (/255) and
 * Normalize(mean 0.5, std 0.5).  The host draws the random decisions; the LR and HR sample
 * of a pair use the same flags and y0, x0 scaled by 4 (:92-96).  The crop must lie inside
 * the image (checked by the caller: the table is in device memory). */
typedef struct srx_crop {
  const uint8_t* img;
  int32_t H, W, y0, x0, flags, pad_;
} srx_crop;
/* out: fp32 [B][3][size][size]; crops: DEVICE array of B entries. */
int srx_crop_augment(const srx_crop* crops /*DEVICE*/, int B, int size, float* out, void* stream);

/* ------------------------------------------------------------ batched ops */
#define SRX_OP_CONV 1
#define SRX_OP_WGRAD 2
#define SRX_OP_PIXEL_UNSHUFFLE 3
#define SRX_OP_CHANNEL_SUM 4
#define SRX_OP_ZERO_STUFF 5
#define SRX_OP_MAXPOOL_FWD 6
#define SRX_OP_MAXPOOL_BWD 7
#define SRX_OP_MEMSET 8
#define SRX_OP_AXPBY 9
#define SRX_OP_GAP_FWD 10
#define SRX_OP_GAP_BWD 11
#define SRX_OP_ACT_BWD 12
#define SRX_OP_RDB_FWD 13
#define SRX_OP_RDB_BWD 14

typedef struct srx_generic_args {   /* for the bandwidth-bound helpers */
  int32_t dtype, B, C, H, W, noct;
  const void* a; int64_t a_img_stride;
  const void* b; int64_t b_img_stride;
  void* c;       int64_t c_img_stride;
  const void* mask; int64_t mask_img_stride;
  float f0, f1;
  int64_t n;
} srx_generic_args;

typedef struct srx_op {
  int32_t kind;
  int32_t pad_;
  union {
    srx_conv_desc conv;
    srx_wgrad_desc wgrad;
    srx_generic_args g;
    srx_rdb_desc rdb;
  } u;
} srx_op;

/* Launch a recorded sequence of ops back to back on one stream with a single
 * crossing of the language boundary (a whole generator forward or backward). */
int srx_run_ops(const srx_op* ops, int n, void* stream);
/* A non-blocking stream of the given HIP priority (1 low, 0 normal, -1 high) on the current
 * device, for launches that run beside the caller's chain (weight gradients beside data gradients).
 * The runtime keeps one pool of hardware queues per priority (four each by default) and gives a new
 * stream a queue of its own while its pool has room; later streams share queues, and two launch
 * chains on one queue run one after the other.
 * HAZARD (observed on gfx950 / ROCm 7.2): a stream that waits on the device for an event of a
 * LOWER-priority stream (hipStreamWaitEvent) can keep that stream from being scheduled, and then
 * neither finishes.  Let streams that wait for each other have the same priority, or join them on
 * the host.  The library's own side stream has normal priority for that reason.
 * No counterpart in the reference (single stream, esrgan/train_esrgan.py:80-118). */
int srx_stream_create(int priority, void** stream);
int srx_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
