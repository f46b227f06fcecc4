// Diagnostic kernels -> libmi355dbg.so (include/mi355_debug.h; tools/ only, never on the product path).
//   mi_debug_code_polluter: a kernel whose body is `kb` KiB of straight-line scalar no-ops, one block per CU slot: walking it
//   replaces that much of every instruction cache (64 KiB per pair of CUs on gfx950) without touching data memory - the
//   probe interleaves it with a convolution to separate "the kernel's code is cold" from "the kernel's data is cold".
#include "common.h"
#include "../../include/mi355_debug.h"

template <int KB>
__global__ __launch_bounds__(64) void code_polluter_kernel(int* out) {
  // 4-byte s_nop x 256 = 1 KiB per repetition
#pragma unroll 1
  for (int r = 0; r < 1; ++r) {
    asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(KB * 256));
  }
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *out = 1;
}

#undef MI_FAIL
#define MI_FAIL(code, ...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return code; } while (0)
#undef MI_CHECK_LAUNCH
#define MI_CHECK_LAUNCH(what) do { if (hipGetLastError() != hipSuccess) MI_FAIL(MI_ELAUNCH, "%s: launch failed", what); } while (0)

extern "C" int mi_debug_code_polluter(int kb, int blocks, mi_dbg_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  if (blocks <= 0) blocks = 512;
  switch (kb) {
    case 8: hipLaunchKernelGGL(code_polluter_kernel<8>, dim3(blocks), dim3(64), 0, s, (int*)nullptr); break;
    case 16: hipLaunchKernelGGL(code_polluter_kernel<16>, dim3(blocks), dim3(64), 0, s, (int*)nullptr); break;
    case 32: hipLaunchKernelGGL(code_polluter_kernel<32>, dim3(blocks), dim3(64), 0, s, (int*)nullptr); break;
    case 64: hipLaunchKernelGGL(code_polluter_kernel<64>, dim3(blocks), dim3(64), 0, s, (int*)nullptr); break;
    case 128: hipLaunchKernelGGL(code_polluter_kernel<128>, dim3(blocks), dim3(64), 0, s, (int*)nullptr); break;
    default: MI_FAIL(MI_EINVAL, "code_polluter: kb %d (8, 16, 32, 64, 128)", kb);
  }
  MI_CHECK_LAUNCH("code_polluter");
  return MI_OK;
}

// ---- CU census: which (XCC, SE, CU) each block of a launch lands on
__global__ __launch_bounds__(256) void cu_census_kernel(uint32_t* out, long long spin_ticks) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc & 0xf;
    out[2 * blockIdx.x + 1] = hw;
  }
}
extern "C" int mi_debug_cu_census(uint32_t* out, int blocks, int spin_us, mi_dbg_stream_t st) {
  if (!out || blocks <= 0) MI_FAIL(MI_EINVAL, "cu_census: args");
  hipLaunchKernelGGL(cu_census_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)st, out, (long long)spin_us * 100);  // 100 MHz
  MI_CHECK_LAUNCH("cu_census");
  return MI_OK;
}
