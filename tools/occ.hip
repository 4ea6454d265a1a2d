// How many one-wave workgroups fit a CU for a given LDS size per workgroup (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ int dyn[];
__global__ void __launch_bounds__(64) k(int* out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); if (out) out[threadIdx.x] = dyn[63 - threadIdx.x]; }
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu sharedMemPerBlock %zu maxBlocksPerMultiProcessor %d\n", (size_t)p.maxSharedMemoryPerMultiProcessor, (size_t)p.sharedMemPerBlock, p.maxBlocksPerMultiProcessor);
  for (int b = 5120; b <= 9216; b += 256) { int n = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, b); printf("LDS %5d B per wave: %d workgroups per CU\n", b, n); }
  return 0;
}
