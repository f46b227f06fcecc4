// Command-list executor, hipGraph capture/replay, timing and hardware layout probes.
// The host (Python mirror of the reference modules) builds a flat list of mi_cmd once; a training
// step is then issued from C++ in one call, or replayed as a captured hipGraph, so the ~10^3 kernel
// launches of a YOLOX step cost no per-launch interpreter time.
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"

__global__ __launch_bounds__(256) void fill16_kernel(u32x4* p, size_t n16, unsigned v) {
  const u32x4 w = {v, v, v, v};
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) p[k] = w;
}

thread_local char g_mi_err[512] = {0};

extern "C" int mi_version(void) { return 100; }
extern "C" const char* mi_last_error(void) { return g_mi_err; }

// device word added to every dropout seed (see mi_dropout_seed_offset in the header); NULL = none
const unsigned long long* g_mi_seed_off = nullptr;
extern "C" int mi_dropout_seed_offset(const uint64_t* dev_word) {
  g_mi_seed_off = (const unsigned long long*)dev_word;
  return MI_OK;
}
extern "C" int mi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int run_one(const mi_cmd& c, hipStream_t s) {
  mi_stream_t st = (mi_stream_t)s;
  const int32_t* i = c.i;
  void* const* p = c.p;
  switch (c.op) {
    case MI_OP_NOP: return MI_OK;
    case MI_OP_CONV: return mi_conv2d((const mi_conv_desc*)p[0], st);
    case MI_OP_WGRAD: return mi_conv2d_wgrad((const mi_wgrad_desc*)p[0], st);
    case MI_OP_PACK_W:
      return mi_pack_conv_weight((const float*)p[0], i[0], i[1], i[2], i[3], p[1], i[4], i[5], p[2], i[6], i[7], st);
    case MI_OP_BN_ACT_FWD:
      return mi_bn_act_fwd(p[0], i[0], (const double*)p[1], i[5], c.l[0], (const float*)p[2], (const float*)p[3], c.f[0],
                           c.f[1], (float*)p[4], (float*)p[5], (int64_t*)p[6], (float*)p[7], (float*)p[8],
                           (float*)p[9], (float*)p[10], p[11], i[1], p[12], i[2], c.l[1], i[3], i[4], st);
    case MI_OP_BN_BWD_REDUCE:
      return mi_bn_act_bwd_reduce(p[0], i[0], p[1], i[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                  (const float*)p[5], (double*)p[6], i[5], i[2], c.l[0], i[3], i[4], st);
    case MI_OP_BN_BWD_APPLY:
      return mi_bn_act_bwd_apply(p[0], i[0], p[1], i[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                 (const float*)p[5], (const float*)p[6], (const double*)p[7], i[7], c.l[1], (float*)p[8],
                                 (float*)p[9], p[10], i[2], p[11], i[3], i[4], c.l[0], i[5], i[6], st);
    case MI_OP_BN_BWD_FUSED:
      return mi_bn_act_bwd_fused(p[0], i[0], p[1], i[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                 (const float*)p[5], (const float*)p[6], (double*)p[7], i[7], c.l[1], (float*)p[8],
                                 (float*)p[9], p[10], i[2], p[11], i[3], i[4], c.l[0], i[5], i[6], (uint32_t*)p[12], st);
    case MI_OP_DWCONV_FWD:
      return mi_dwconv3x3_fwd(p[0], i[0], (const float*)p[1], p[2], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8],
                              (double*)p[3], i[9], st);
    case MI_OP_DWCONV_DGRAD:
      return mi_dwconv3x3_dgrad(p[0], i[0], (const float*)p[1], p[2], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], st);
    case MI_OP_DWCONV_WGRAD:
      return mi_dwconv3x3_wgrad(p[0], i[0], p[1], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], (float*)p[2], c.l[0],
                                (float*)p[3], st);
    case MI_OP_FOCUS:
      return i[4] ? mi_focus_pack_u8((const uint8_t*)p[0], i[0], i[1], i[2], p[1], i[3], st)
                  : mi_focus_pack((const float*)p[0], i[0], i[1], i[2], p[1], i[3], st);
    case MI_OP_UPSAMPLE_FWD: return mi_upsample2x_fwd(p[0], i[0], p[1], i[1], i[2], i[3], i[4], i[5], st);
    case MI_OP_UPSAMPLE_BWD: return mi_upsample2x_bwd(p[0], i[0], p[1], i[1], i[2], i[3], i[4], i[5], i[6], st);
    case MI_OP_SPP_FWD: return mi_spp_pool_fwd(p[0], i[0], p[1], p[2], p[3], i[1], (uint8_t*)p[4], i[2], i[3], i[4], i[5], st);
    case MI_OP_SPP_BWD:
      return mi_spp_pool_bwd(p[0], p[1], p[2], i[0], (const uint8_t*)p[3], p[4], i[1], i[2], i[3], i[4], i[5], i[6], st);
    case MI_OP_COPY: return mi_copy_bf16(p[0], i[0], p[1], i[1], i[2], c.l[0], i[3], st);
    case MI_OP_COLSUM: return mi_colsum_bf16(p[0], i[0], c.l[0], i[1], (float*)p[1], i[2], (float*)p[2], st);
    case MI_OP_BIAS_GRADS:
      return mi_yolox_bias_grads((const float*)p[1], i[0], i[1], i[2], (const mi_bias_job*)p[0], i[3], (float*)p[2], st);
    case MI_OP_WGRAD_GROUP: return mi_conv2d_wgrad_group_run((const mi_wgrad_group*)p[0], p[1], st);
    case MI_OP_CONV_GROUP: return mi_conv2d_group_run((const mi_conv_group*)p[0], p[1], st);
    case MI_OP_BN_GROUP: return mi_bn_group_run((const mi_bn_group*)p[0], p[1], st);
    case MI_OP_PACK_W_BATCH: return mi_pack_conv_weights_batch((const mi_pack_job*)p[0], i[0], i[1], i[2], st);
    case MI_OP_LOSS_FWD: return mi_yolox_loss_fwd((const mi_yolox_loss_desc*)p[0], st);
    case MI_OP_LOSS_BWD_FUSED:
      return mi_yolox_loss_bwd_fused((const mi_yolox_loss_desc*)p[0], (const float*)p[1], (float*)p[2], (const mi_split_job*)p[3], i[0],
                                     (const mi_bias_job*)p[4], i[1], (float*)p[5], c.l[0], st);
    case MI_OP_LOSS_BWD: return mi_yolox_loss_bwd((const mi_yolox_loss_desc*)p[0], (const float*)p[1], (float*)p[2], st);
    case MI_OP_SPLIT_DPREDS:
      return mi_yolox_split_dpreds((const float*)p[0], i[0], i[1], i[2], i[3], i[4], i[5], i[6], p[1], i[7], st);
    case MI_OP_SPLIT_DPREDS_BATCH:
      return mi_yolox_split_dpreds_batch((const float*)p[1], i[0], i[1], i[2], (const mi_split_job*)p[0], i[3], st);
    case MI_OP_MEMSET: {
      // a fill KERNEL for the aligned case (every use in the step plans: the fp64 BatchNorm accumulators, 256-byte
      // multiples): a kernel node is ordered like its neighbours inside a captured hipGraph; hipMemsetAsync nodes were the
      // one node kind whose effect went missing on later replays of a large captured graph (round 4, SparseInst step)
      const size_t n = (size_t)c.l[0];
      if (((uintptr_t)p[0] % 16) == 0 && n % 16 == 0 && n > 0 && n < ((size_t)1 << 40)) {
        const unsigned v = (unsigned)(i[0] & 0xff) * 0x01010101u;
        const size_t n16 = n / 16;
        size_t nb = (n16 + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(fill16_kernel, dim3((unsigned)nb), dim3(256), 0, s, (u32x4*)p[0], n16, v);
        MI_CHECK_LAUNCH("memset (fill kernel)");
        return MI_OK;
      }
      if (hipMemsetAsync(p[0], i[0], n, s) != hipSuccess) MI_FAIL(MI_ELAUNCH, "memset failed");
      return MI_OK;
    }
    case MI_OP_SGD:
      return mi_sgd_momentum_step((float*)p[0], (const float*)p[1], (float*)p[2], (const mi_sgd_seg*)p[3], i[0],
                                  c.f[0], c.f[1], i[1], st);
    case MI_OP_BN_EVAL_AFFINE:
      return mi_bn_eval_affine((const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], c.f[0],
                               i[0], (float*)p[4], (float*)p[5], st);
    case MI_OP_DECODE: return mi_yolox_decode((float*)p[0], (const float*)p[1], i[0], i[1], i[2], st);
    default: MI_FAIL(MI_EINVAL, "cmdlist: unknown op %d", c.op);
  }
}

// auxiliary streams / events for parallel chains (created once, never destroyed: process lifetime)
static hipStream_t g_aux[MI_MAX_AUX];
static hipEvent_t g_ev_fork[MI_MAX_AUX], g_ev_join[MI_MAX_AUX];
static bool g_aux_ready = false;
static hipStream_t g_aux_user[MI_MAX_AUX] = {nullptr, nullptr, nullptr, nullptr};   // caller-owned replacements (mi_aux_stream_set)
static inline hipStream_t aux_stream(int k) { return g_aux_user[k] ? g_aux_user[k] : g_aux[k]; }
static int ensure_aux() {
  if (g_aux_ready) return MI_OK;
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  for (int k = 0; k < MI_MAX_AUX; ++k) {
    // the LAST auxiliary stream carries background work (weight gradients beside the backward chain): lowest priority,
    // so that the chain's workgroups are dispatched first whenever both have blocks pending
    const hipError_t e = (k == MI_MAX_AUX - 1)
                             ? hipStreamCreateWithPriority(&g_aux[k], hipStreamNonBlocking, prio_least)
                             : hipStreamCreateWithFlags(&g_aux[k], hipStreamNonBlocking);
    if (e != hipSuccess) MI_FAIL(MI_ELAUNCH, "aux stream");
    if (hipEventCreateWithFlags(&g_ev_fork[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_ev_join[k], hipEventDisableTiming) != hipSuccess)
      MI_FAIL(MI_ELAUNCH, "aux event");
  }
  g_aux_ready = true;
  return MI_OK;
}

extern "C" int mi_cmdlist_run(const mi_cmd* cmds, int n, mi_stream_t st) {
  if (mi_device_count() <= 0) MI_FAIL(MI_ENODEV, "no HIP device");
  hipStream_t s = (hipStream_t)st;
  hipStream_t cur = s;
  for (int k = 0; k < n; ++k) {
    const int op = cmds[k].op;
    if (op == MI_OP_STREAM || op == MI_OP_FORK || op == MI_OP_JOIN) {
      const int sid = cmds[k].i[0];
      if (sid < 0 || sid > MI_MAX_AUX) MI_FAIL(MI_EINVAL, "cmd %d: stream id %d", k, sid);
      if (sid > 0 && ensure_aux() != MI_OK) return MI_ELAUNCH;
      if (op == MI_OP_STREAM) {
        cur = sid ? aux_stream(sid - 1) : s;
      } else if (sid > 0 && op == MI_OP_FORK) {
        if (hipEventRecord(g_ev_fork[sid - 1], s) != hipSuccess ||
            hipStreamWaitEvent(aux_stream(sid - 1), g_ev_fork[sid - 1], 0) != hipSuccess)
          MI_FAIL(MI_ELAUNCH, "cmd %d: fork", k);
      } else if (sid > 0) {
        if (hipEventRecord(g_ev_join[sid - 1], aux_stream(sid - 1)) != hipSuccess ||
            hipStreamWaitEvent(s, g_ev_join[sid - 1], 0) != hipSuccess)
          MI_FAIL(MI_ELAUNCH, "cmd %d: join", k);
      }
      continue;
    }
    const int rc = run_one(cmds[k], cur);
    if (rc != MI_OK) {
      char tmp[400];
      snprintf(tmp, sizeof(tmp), "%s", g_mi_err);
      snprintf(g_mi_err, sizeof(g_mi_err), "cmd %d (op %d): %s", k, cmds[k].op, tmp);
      return rc;
    }
  }
  return MI_OK;
}

extern "C" int mi_upload_async(void* dst_dev, const void* src_pinned, int64_t nbytes, mi_stream_t st) {
  MI_REQUIRE(dst_dev && src_pinned && nbytes > 0, "upload_async: args");
  const hipError_t e = hipMemcpyAsync(dst_dev, src_pinned, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)st);
  if (e != hipSuccess) MI_FAIL(MI_ELAUNCH, "upload_async: %s", hipGetErrorString(e));
  return MI_OK;
}

// ---- CU-masked streams (the weight-gradient side queue)
extern "C" int mi_stream_create_cu_mask(const uint32_t* mask, int nwords, mi_stream_t* out) {
  if (mi_device_count() <= 0) MI_FAIL(MI_ENODEV, "no HIP device");
  MI_REQUIRE(mask && out && nwords > 0 && nwords <= 32, "stream_create_cu_mask: args");
  bool any = false;
  for (int k = 0; k < nwords; ++k) any = any || mask[k] != 0;
  MI_REQUIRE(any, "stream_create_cu_mask: empty mask");
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask);
  if (e != hipSuccess || !s) MI_FAIL(MI_ELAUNCH, "hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
  *out = (mi_stream_t)s;
  return MI_OK;
}
extern "C" int mi_stream_destroy(mi_stream_t st) {
  MI_REQUIRE(st, "stream_destroy: null");
  for (int k = 0; k < MI_MAX_AUX; ++k)
    if (g_aux_user[k] == (hipStream_t)st) g_aux_user[k] = nullptr;
  if (hipStreamDestroy((hipStream_t)st) != hipSuccess) MI_FAIL(MI_ELAUNCH, "hipStreamDestroy failed");
  return MI_OK;
}
extern "C" int mi_aux_stream_set(int sid, mi_stream_t st) {
  MI_REQUIRE(sid >= 1 && sid <= MI_MAX_AUX, "aux_stream_set: stream id %d", sid);
  g_aux_user[sid - 1] = (hipStream_t)st;
  return MI_OK;
}

// ---- hipGraph capture / replay
static std::mutex g_graph_mu;
static std::unordered_map<int64_t, hipGraphExec_t> g_graphs;
static int64_t g_next_graph = 1;

extern "C" int64_t mi_graph_capture(const mi_cmd* cmds, int n, mi_stream_t st) {
  if (mi_device_count() <= 0) MI_FAIL(MI_ENODEV, "no HIP device");
  hipStream_t s = (hipStream_t)st;
  if (s == nullptr) MI_FAIL(MI_EINVAL, "graph capture needs a non-default stream");
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess)
    MI_FAIL(MI_ELAUNCH, "hipStreamBeginCapture failed");
  const int rc = mi_cmdlist_run(cmds, n, st);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != MI_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess || !graph) MI_FAIL(MI_ELAUNCH, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t exec = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) MI_FAIL(MI_ELAUNCH, "hipGraphInstantiate: %s", hipGetErrorString(e2));
  std::lock_guard<std::mutex> lk(g_graph_mu);
  const int64_t h = g_next_graph++;
  g_graphs[h] = exec;
  return h;
}
extern "C" int mi_graph_launch(int64_t handle, mi_stream_t st) {
  hipGraphExec_t exec;
  {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    auto it = g_graphs.find(handle);
    if (it == g_graphs.end()) MI_FAIL(MI_EINVAL, "graph: bad handle");
    exec = it->second;
  }
  const hipError_t e = hipGraphLaunch(exec, (hipStream_t)st);
  if (e != hipSuccess) MI_FAIL(MI_ELAUNCH, "hipGraphLaunch: %s", hipGetErrorString(e));
  return MI_OK;
}
extern "C" int mi_graph_destroy(int64_t handle) {
  std::lock_guard<std::mutex> lk(g_graph_mu);
  auto it = g_graphs.find(handle);
  if (it == g_graphs.end()) return MI_EINVAL;
  (void)hipGraphExecDestroy(it->second);
  g_graphs.erase(it);
  return MI_OK;
}

extern "C" int mi_cmdlist_time(const mi_cmd* cmds, int n, int iters, float* ms, float* per_cmd_ms, mi_stream_t st) {
  if (mi_device_count() <= 0) MI_FAIL(MI_ENODEV, "no HIP device");
  hipStream_t s = (hipStream_t)st;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) MI_FAIL(MI_ELAUNCH, "event create");
  int rc = MI_OK;
  if (ms) {
    (void)hipEventRecord(e0, s);
    for (int it = 0; it < iters && rc == MI_OK; ++it) rc = mi_cmdlist_run(cmds, n, st);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms = t / (float)(iters > 0 ? iters : 1);
  }
  if (per_cmd_ms && rc == MI_OK) {
    // events recorded in-stream between the commands, ONE host synchronisation per replay: the device runs the list
    // back to back (clocks and caches as in a real step) and the event deltas are the per-command durations
    for (int k = 0; k < n; ++k) per_cmd_ms[k] = 0.f;
    std::vector<hipEvent_t> ev((size_t)n + 1);
    for (auto& e : ev)
      if (hipEventCreate(&e) != hipSuccess) MI_FAIL(MI_ELAUNCH, "event create");
    for (int it = 0; it < iters && rc == MI_OK; ++it) {
      (void)hipEventRecord(ev[0], s);
      for (int k = 0; k < n && rc == MI_OK; ++k) {
        rc = mi_cmdlist_run(cmds + k, 1, st);
        (void)hipEventRecord(ev[k + 1], s);
      }
      (void)hipEventSynchronize(ev[n]);
      for (int k = 0; k < n && rc == MI_OK; ++k) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, ev[k], ev[k + 1]);
        per_cmd_ms[k] += t / (float)iters;
      }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

// ---- layout probes: D = A x B through the exact lane mappings the conv kernels assume
__global__ void probe_mfma32_kernel(const __bf16* A, const __bf16* B, float* D) {
  // A [32][16] row-major (M x K), B [16][32] row-major (K x N)
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[l31 * 16 + h * 8 + j];
    b[j] = B[(h * 8 + j) * 32 + l31];
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    D[row * 32 + l31] = acc[r];
  }
}
__global__ void probe_mfma16_kernel(const __bf16* A, const __bf16* B, float* D) {
  // A [16][32] row-major (M x K), B [32][16] row-major (K x N)
  const int lane = threadIdx.x, t = lane & 15, g = lane >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[t * 32 + g * 8 + j];
    b[j] = B[(g * 8 + j) * 16 + t];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + t] = acc[r];
}
extern "C" int mi_probe_mfma32(const void* a, const void* b, float* d, mi_stream_t st) {
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, (const __bf16*)a, (const __bf16*)b, d);
  MI_CHECK_LAUNCH("probe_mfma32");
  return MI_OK;
}
extern "C" int mi_probe_mfma16(const void* a, const void* b, float* d, mi_stream_t st) {
  hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)st, (const __bf16*)a, (const __bf16*)b, d);
  MI_CHECK_LAUNCH("probe_mfma16");
  return MI_OK;
}

// sizeof() of every public struct, so that a binding (ctypes, cgo, JNI ...) can verify its own layout against the
// library it loaded: 0 mi_conv_desc, 1 mi_wgrad_desc, 2 mi_wgrad_group, 3 mi_pack_job, 4 mi_bias_job,
// 5 mi_yolox_loss_desc, 6 mi_detr_loss_desc, 7 mi_sgd_seg, 8 mi_cmd, 9 mi_conv_group, 10 mi_bn_job, 11 mi_bn_group, 12 mi_pil_resize_job, 13 mi_jpeg_info, 14 mi_jpeg_job
extern "C" int mi_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(mi_conv_desc);
    case 1: return (int)sizeof(mi_wgrad_desc);
    case 2: return (int)sizeof(mi_wgrad_group);
    case 3: return (int)sizeof(mi_pack_job);
    case 4: return (int)sizeof(mi_bias_job);
    case 5: return (int)sizeof(mi_yolox_loss_desc);
    case 6: return (int)sizeof(mi_detr_loss_desc);
    case 7: return (int)sizeof(mi_sgd_seg);
    case 8: return (int)sizeof(mi_cmd);
    case 9: return (int)sizeof(mi_conv_group);
    case 10: return (int)sizeof(mi_bn_job);
    case 11: return (int)sizeof(mi_bn_group);
    case 12: return (int)sizeof(mi_pil_resize_job);
    case 13: return (int)sizeof(mi_jpeg_info);
    case 14: return (int)sizeof(mi_jpeg_job);
    case 15: return (int)sizeof(mi_bnx);
  }
  return -1;
}
