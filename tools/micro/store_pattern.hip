// how fast can 16-byte-per-lane stores run when a wave instruction covers SEG contiguous bytes per row of RS bytes?
// (the MFMA fragment epilogue writes 32-byte pieces of 32 different rows; a staged epilogue writes whole rows)
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// one "tile" = 32 rows x 256 B = 8 KB written by 8 instructions of one wave (SEG = 32) or fewer rows per instruction
template <int SEG>
__global__ __launch_bounds__(256) void k(char* out, long ntiles, int rs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  constexpr int LPR = SEG / 16;          // lanes per row segment
  constexpr int RPI = 64 / LPR;          // rows per instruction
  constexpr int NI = 32 / RPI * (256 / SEG);  // instructions per tile
  for (long t = gw; t < ntiles; t += nw) {
    char* base = out + t * 32 * (long)rs;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // instruction i: rows group (i % (32 / RPI)), column segment (i / (32 / RPI))
      const int rg = i % (32 / RPI), cs = i / (32 / RPI);
      const int row = rg * RPI + lane / LPR, col = cs * SEG + (lane % LPR) * 16;
      *(u32x4*)(base + (long)row * rs + col) = v;
    }
  }
}
template <int SEG>
void run(char* d, long bytes, int blocks) {
  const int rs = 256;
  const long ntiles = bytes / (32 * rs);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<SEG>, dim3(blocks), dim3(256), 0, 0, d, ntiles, rs);
  hipEventRecord(e0);
  const int it = 5;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL(k<SEG>, dim3(blocks), dim3(256), 0, 0, d, ntiles, rs);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("SEG %3d B/row/instr, %5d blocks, %4ld MB: %7.1f us  %6.0f GB/s\n", SEG, blocks, bytes >> 20, ms * 1e3 / it, bytes / (ms / it * 1e-3) / 1e9);
}
int main() {
  char* d; const long cap = 1L << 30; hipMalloc(&d, cap);
  for (long bytes : {26L << 20, 105L << 20, 1L << 30})
    for (int blocks : {256, 1024}) {
      run<32>(d, bytes, blocks); run<64>(d, bytes, blocks); run<128>(d, bytes, blocks); run<256>(d, bytes, blocks);
    }
  return 0;
}
