// How many bytes per second can a CU / the chip take in from L2-resident data?  Every block streams the same small buffer
// (SZ bytes, L2 / MALL resident after the first pass) with 16-byte loads per lane: (0) plain global_load_dwordx4 into
// registers, U loads in flight per wave, (1) LDS-DMA (global_load_lds_dwordx4) into a 64 KB ring, U instructions in flight.
// hipcc --offload-arch=gfx950 -O3 cu_ingest.hip -o cu_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_plain(const u32x4* __restrict__ src, size_t n16, int passes, unsigned* out) {
  const int tid = threadIdx.x;
  unsigned acc = 0;
  // block b starts at a different offset so that the blocks of an XCD do not all hit the same line at once
  size_t base = ((size_t)blockIdx.x * 4099 * 256) % n16;
  for (int p = 0; p < passes; ++p) {
    for (size_t i = 0; i + 256 * U <= n16; i += 256 * U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t j = base + i + u * 256 + tid;
        if (j >= n16) j -= n16;
        v[u] = src[j];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u][0] ^ v[u][3];
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_dma(const u32x4* __restrict__ src, size_t n16, int passes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  size_t base = ((size_t)blockIdx.x * 4099 * 256) % n16;
  unsigned acc = 0;
  for (int p = 0; p < passes; ++p) {
    for (size_t i = 0; i + 256 * U <= n16; i += 256 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        size_t j = base + i + u * 256 + tid;
        if (j >= n16) j -= n16;
        const void* g = src + j;
        const unsigned off = lds0 + ((u * 4 + wave) % 64) * 1024;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(off) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  acc = ((unsigned*)smem)[tid];
  out[blockIdx.x * 256 + tid] = acc;
}

template <class F>
static void run(const char* name, F launch, int blocks, size_t bytes, int passes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); CHECK(hipDeviceSynchronize());
  hipEventRecord(e0); launch(); hipEventRecord(e1); CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)blocks * bytes * passes;
  printf("%-34s blocks %4d x %5.1f MB x %2d: %8.1f us  %7.2f TB/s chip  %6.1f GB/s per block\n", name, blocks, bytes / 1e6, passes, ms * 1e3,
         total / ms / 1e9, total / blocks / ms / 1e6);
}

int main() {
  u32x4* src; unsigned* out;
  const size_t maxb = 64u << 20;
  CHECK(hipMalloc(&src, maxb)); CHECK(hipMemset(src, 1, maxb)); CHECK(hipMalloc(&out, 2048 * 256 * 4));
  CHECK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (size_t bytes : {(size_t)1 << 20, (size_t)8 << 20, (size_t)32 << 20}) {
    const size_t n16 = bytes / 16;
    const int passes = (int)((256u << 20) / bytes / 8) + 1;
    for (int blocks : {1, 32, 256, 512}) {
      run("plain 16 B loads, 8 in flight", [&] { hipLaunchKernelGGL(k_plain<8>, dim3(blocks), dim3(256), 0, 0, src, n16, passes, out); }, blocks, bytes, passes);
      run("plain 16 B loads, 16 in flight", [&] { hipLaunchKernelGGL(k_plain<16>, dim3(blocks), dim3(256), 0, 0, src, n16, passes, out); }, blocks, bytes, passes);
      run("LDS-DMA, 8 per wave in flight", [&] { hipLaunchKernelGGL(k_dma<8>, dim3(blocks), dim3(256), 65536, 0, src, n16, passes, out); }, blocks, bytes, passes);
      run("LDS-DMA, 16 per wave in flight", [&] { hipLaunchKernelGGL(k_dma<16>, dim3(blocks), dim3(256), 65536, 0, src, n16, passes, out); }, blocks, bytes, passes);
    }
  }
  return 0;
}
