// Random-read characterisation of the memory system (development tool, not part of the product):
//   throughput of independent 32-byte reads and latency of a dependent chain, against the size of the region read.
// hipcc --offload-arch=gfx950 -O3 tools/randread.hip -o gpurun_out/randread && gpurun_out/randread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// every thread: `per` independent 32-byte reads at hashed slots
__global__ void k_tput(const uint4* __restrict__ buf, uint64_t nslots, int per, uint64_t* sink) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  for (int i = 0; i < per; ++i) {
    const uint64_t s = (mix(t * 1315423911ull + i) >> 11) % nslots;
    const uint4 a = buf[2 * s], b = buf[2 * s + 1];
    acc += a.x + b.y;
  }
  if (acc == 0x1234567) sink[0] = acc;
}
// one lane per wave follows a dependent chain (address from the value read + a hash)
__global__ void k_chase(const uint4* __restrict__ buf, uint64_t nslots, int steps, uint64_t* sink) {
  const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) != 0) return;
  uint64_t s = mix(w) % nslots, acc = 0;
  for (int i = 0; i < steps; ++i) {
    const uint4 a = buf[2 * s];
    acc += a.x;
    s = (mix(s + a.x + i) >> 7) % nslots;
  }
  if (acc == 0x1234567) sink[0] = acc;
}
int main() {
  uint64_t* sink; CK(hipMalloc(&sink, 8));
  const double gbs[] = {0.0625, 0.25, 1, 2, 3.5, 7, 14, 32, 64, 160};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (double gb : gbs) {
    const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
    uint4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%.2f GB: alloc failed\n", gb); continue; }
    CK(hipMemset(buf, 1, bytes));
    const uint64_t nslots = bytes / 32;
    const int blocks = 256 * 8 * 4, per = 16;
    hipLaunchKernelGGL(k_tput, dim3(blocks), dim3(256), 0, 0, buf, nslots, per, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_tput, dim3(blocks), dim3(256), 0, 0, buf, nslots, per, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double nreq = (double)blocks * 256 * per;
    // dependent chain: few waves (latency unloaded), then many (loaded)
    float msc[2];
    const int cb[2] = {64, 256 * 20};
    for (int v = 0; v < 2; ++v) {
      hipLaunchKernelGGL(k_chase, dim3(cb[v]), dim3(64), 0, 0, buf, nslots, 2000, sink);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chase, dim3(cb[v]), dim3(64), 0, 0, buf, nslots, 2000, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&msc[v], e0, e1));
    }
    printf("%7.2f GB: independent 32-B reads %6.1f G/s (%.2f ms for %.0f M)   chain: %6.0f ns/step with 64 waves, %6.0f ns/step with 5120 waves (= %.1f G/s)\n",
           gb, nreq / ms / 1e6, ms, nreq / 1e6, msc[0] * 1e6 / 2000, msc[1] * 1e6 / 2000, 5120.0 * 2000 / msc[1] / 1e6);
    fflush(stdout);
    CK(hipFree(buf));
  }
  return 0;
}
