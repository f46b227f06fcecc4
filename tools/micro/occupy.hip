// A stand-in for a collective kernel's footprint on the chip (RCCL cannot run on this pool's 1-GPU boxes): G persistent
// blocks of 256 threads that stay resident for a given time, either sleeping (pure CU-slot footprint) or streaming a
// buffer (read + write: the HBM share a ring all-reduce takes).  Built by tools/ddp_footprint.sh into tools/micro/libocc.so;
// NOT part of libmi355det.so.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, float4* buf, long long n4, int stream_mem) {
  const long long t0 = wall_clock64();
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long step = (long long)gridDim.x * 256;
  while (wall_clock64() - t0 < ticks) {
    if (stream_mem) {
      for (int k = 0; k < 64; ++k) {
        float4 v = buf[i];
        v.x += 1.0f;
        buf[i] = v;
        i += step;
        if (i >= n4) i -= n4;
      }
    } else {
      __builtin_amdgcn_s_sleep(64);
    }
  }
}

extern "C" int occupy(int blocks, long long usec, void* buf, long long bytes, int stream_mem, void* stream) {
  // wall_clock64: constant 100 MHz on gfx9
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, usec * 100, (float4*)buf, bytes / 16, stream_mem);
  return (int)hipGetLastError();
}
