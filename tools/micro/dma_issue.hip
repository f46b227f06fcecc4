// What does ONE global_load_lds_dwordx4 cost the issuing wave?  One block per CU (blocks = 256) or a few, 4 waves, every wave
// issues 8 DMA instructions (1 KB each) then waits for them; the stamp pair around the 8 issues (wall_clock64) gives the
// issue time, the pair around the wait the landing time.  Patterns: 0 a lane-linear 1 KB piece; 1 four 256-byte rows of a
// tensor with a 2 KB pixel pitch (the 1x1 weight gradient's piece of a 1024-channel map); source warm (64 MB shared, L2 / MALL)
// or cold (every block its own 8 MB region of a 2 GB buffer, streamed once).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t region, size_t stride_blk, int iters, long long* tl, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * stride_blk;
  long long t_issue = 0, t_land = 0;
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    const long long a = wall_clock64();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      size_t o;
      if (PAT == 0) o = off + ((size_t)(u * 4 + wave) * 1024) + lane * 16;
      else o = off + ((size_t)((u * 4 + wave) * 4 + (lane >> 4)) * 2048) + (lane & 15) * 16;   // 4 rows x 256 B, 2 KB pitch
      const void* g = base + (o % region);
      const unsigned l = lds0 + (u * 4 + wave) * 1024;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
    }
    const long long b = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c = wall_clock64();
    t_issue += b - a;
    t_land += c - b;
    off += PAT == 0 ? 32 * 1024 : 32 * 4 * 2048;
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0) { tl[blockIdx.x * 2] = t_issue; tl[blockIdx.x * 2 + 1] = t_land; }
  out[blockIdx.x * 256 + tid] = ((unsigned*)smem)[tid];
}

int main() {
  char* src; long long* tl; unsigned* out;
  const size_t big = (size_t)2 << 30;
  CHECK(hipMalloc(&src, big)); CHECK(hipMemset(src, 1, big)); CHECK(hipMalloc(&tl, 4096 * 16)); CHECK(hipMalloc(&out, 4096 * 256 * 4));
  CHECK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int iters = 200;
  for (int blocks : {32, 256, 512}) {
    for (int pat = 0; pat < 2; ++pat) {
      for (int cold = 0; cold < 2; ++cold) {
        const size_t region = cold ? (size_t)4 << 20 : (size_t)8 << 20;   // bytes a block walks through (wraps)
        const size_t sb = cold ? ((size_t)4 << 20) : 0;                   // cold: every block its own region (<= 512 x 4 MB = 2 GB)
        // cold runs touch each byte about twice over 200 steps of 32 KB (pattern 0) - the first pass is the cold one; flush between
        CHECK(hipMemset(src, 2, (size_t)1 << 30));
        if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 65536, 0, src, region, sb, iters, tl, out);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 65536, 0, src, region, sb, iters, tl, out);
        CHECK(hipDeviceSynchronize());
        long long h[2]; CHECK(hipMemcpy(h, tl, 16, hipMemcpyDeviceToHost));
        printf("blocks %3d  %-28s %-5s: per step (8 DMA per wave, 32 KB per block): issue %6.0f ns  landing wait %6.0f ns\n", blocks,
               pat ? "4 rows x 256 B, 2 KB pitch" : "lane-linear 1 KB", cold ? "cold" : "warm", h[0] * 10.0 / iters, h[1] * 10.0 / iters);
      }
    }
  }
  return 0;
}
