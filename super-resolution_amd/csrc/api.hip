// libsrx: error reporting, version, and the batched op runner.
#include <stdarg.h>
#include <string.h>
#include "srx_common.h"

namespace {
thread_local char g_err[512] = "";
}

void srx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int srx_version(void) { return 100; }

const char* srx_last_error(void) { return g_err; }

int srx_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(srx_conv_desc);
    case 1: return (int)sizeof(srx_wgrad_task);
    case 2: return (int)sizeof(srx_wgrad_desc);
    case 3: return (int)sizeof(srx_pack_seg);
    case 4: return (int)sizeof(srx_pack_layer);
    case 5: return (int)sizeof(srx_unpack_item);
    case 6: return (int)sizeof(srx_sn_layer);
    case 7: return (int)sizeof(srx_sn_bwd_layer);
    case 8: return (int)sizeof(srx_generic_args);
    case 9: return (int)sizeof(srx_op);
    case 10: return (int)sizeof(srx_tile);
    case 11: return (int)sizeof(srx_crop);
    case 12: return (int)sizeof(srx_rdb_desc);
    default: return -1;
  }
}

namespace {
int selftest_calls = 0;
int selftest_set(void* arg) {
  ++selftest_calls;
  return *static_cast<int*>(arg);      // the status the "attribute call" is to return
}
}  // namespace

int srx_selftest_per_device_once(void) {
  // the bookkeeping behind hipFuncSetAttribute(MaxDynamicSharedMemorySize): the setter must run
  // once for EVERY device a kernel is launched on (not once per process), again after a failure,
  // and never for an out-of-range device index.  No GPU call is made.
  static srx_per_device_once flags;
  for (int d = 0; d < SRX_MAX_DEVICES; ++d) flags.done[d] = 0;
  selftest_calls = 0;
  int ok = SRX_OK, fail = SRX_ERR_HIP;
  const int seq[] = {0, 1, 0, 1, 7, 0, 7};
  for (int d : seq)
    if (srx_once_per_device(&flags, d, selftest_set, &ok) != SRX_OK) return -1;
  if (selftest_calls != 3) return -2;                          // devices 0, 1, 7: once each
  if (srx_once_per_device(&flags, 3, selftest_set, &fail) != SRX_ERR_HIP) return -3;
  if (srx_once_per_device(&flags, 3, selftest_set, &ok) != SRX_OK || selftest_calls != 5) return -4;   // retried
  if (srx_once_per_device(&flags, SRX_MAX_DEVICES, selftest_set, &ok) != SRX_ERR_INVALID) return -5;
  if (srx_once_per_device(&flags, -1, selftest_set, &ok) != SRX_ERR_INVALID || selftest_calls != 5) return -6;
  return 0;
}

int srx_run_ops(const srx_op* ops, int n, void* stream) {
  SRX_CHECK_ARG(ops != nullptr && n >= 0, "srx_run_ops: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < n; ++i) {
    const srx_op& op = ops[i];
    const srx_generic_args& g = op.u.g;
    int rc = SRX_OK;
    switch (op.kind) {
      case SRX_OP_CONV: rc = srx_conv2d(&op.u.conv, stream); break;
      case SRX_OP_WGRAD: rc = srx_conv2d_wgrad(&op.u.wgrad, stream); break;
      case SRX_OP_RDB_FWD: rc = srx_rdb_forward(&op.u.rdb, stream); break;
      case SRX_OP_RDB_BWD: rc = srx_rdb_backward(&op.u.rdb, stream); break;
      case SRX_OP_PIXEL_UNSHUFFLE:
        rc = srx_pixel_unshuffle2(g.dtype, g.a, g.a_img_stride, g.c, g.c_img_stride, g.B, g.C, g.H, g.W, stream);
        break;
      case SRX_OP_CHANNEL_SUM:
        rc = srx_channel_sum(g.dtype, g.a, g.a_img_stride, g.noct, g.B, g.H, g.W, (float*)g.c, g.f0, stream);
        break;
      case SRX_OP_ZERO_STUFF:
        rc = srx_zero_stuff2(g.dtype, g.a, g.a_img_stride, g.c, g.c_img_stride, g.B, g.noct, g.H, g.W, stream);
        break;
      case SRX_OP_MAXPOOL_FWD:
        rc = srx_maxpool2_fwd(g.dtype, g.a, g.a_img_stride, g.c, g.c_img_stride, g.B, g.noct, g.H, g.W, stream);
        break;
      case SRX_OP_MAXPOOL_BWD:
        rc = srx_maxpool2_bwd(g.dtype, g.a, g.a_img_stride, g.b, g.b_img_stride, g.c, g.c_img_stride, g.mask,
                              g.mask_img_stride, g.f0, g.B, g.noct, g.H, g.W, stream);
        break;
      case SRX_OP_MEMSET:
        if (hipMemsetAsync(g.c, 0, (size_t)g.n, s) != hipSuccess) {
          srx_set_error("srx_run_ops: hipMemsetAsync failed at op %d", i);
          rc = SRX_ERR_HIP;
        }
        break;
      case SRX_OP_AXPBY: rc = srx_axpby(g.dtype, g.a, g.c, g.f0, g.f1, g.n, stream); break;
      case SRX_OP_GAP_FWD: rc = srx_gap_fwd(g.dtype, g.a, g.a_img_stride, g.c, g.B, g.noct, g.H, g.W, stream); break;
      case SRX_OP_GAP_BWD:
        rc = srx_gap_bwd(g.dtype, g.a, g.c, g.c_img_stride, g.mask, g.mask_img_stride, g.f0, g.B, g.noct, g.H,
                         g.W, stream);
        break;
      case SRX_OP_ACT_BWD: rc = srx_act_bwd(g.dtype, g.a, g.mask, g.c, g.f0, g.n, stream); break;
      default:
        srx_set_error("srx_run_ops: unknown op kind %d at index %d", op.kind, i);
        rc = SRX_ERR_INVALID;
    }
    if (rc != SRX_OK) {
      char tmp[400];
      strncpy(tmp, g_err, sizeof(tmp) - 1);
      tmp[sizeof(tmp) - 1] = 0;
      srx_set_error("op %d (kind %d): %s", i, op.kind, tmp);
      return rc;
    }
  }
  return SRX_OK;
}

int srx_stream_create(int priority, void** stream) {
  SRX_CHECK_ARG(stream != nullptr, "srx_stream_create: null result pointer");
  int least = 0, greatest = 0;
  SRX_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  SRX_CHECK_ARG(priority <= least && priority >= greatest, "srx_stream_create: priority %d outside [%d, %d]",
                priority, greatest, least);
  hipStream_t s = nullptr;
  SRX_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
  *stream = s;
  return SRX_OK;
}

int srx_stream_destroy(void* stream) {
  if (stream) SRX_CHECK_HIP(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
  return SRX_OK;
}

}  // extern "C"
