// Diagnostic kernels (tools/icache_probe.py; never on the product path).
//   mi_debug_code_polluter: a kernel whose body is `kb` KiB of straight-line scalar no-ops, one block per CU slot: walking it
//   replaces that much of every instruction cache (64 KiB per pair of CUs on gfx950) without touching data memory - the
//   probe interleaves it with a convolution to separate "the kernel's code is cold" from "the kernel's data is cold".
#include "common.h"

template <int KB>
__global__ __launch_bounds__(64) void code_polluter_kernel(int* out) {
  // 4-byte s_nop x 256 = 1 KiB per repetition
#pragma unroll 1
  for (int r = 0; r < 1; ++r) {
    asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(KB * 256));
  }
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *out = 1;
}

extern "C" int mi_debug_code_polluter(int kb, int blocks, mi_stream_t st) {
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
