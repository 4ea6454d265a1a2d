// How long hipMalloc / hipFree / hipHostMalloc take on this box, by size (the first batch of a CLI worker pays for its 21 GB
// of search scratch and its page-locked buffers inside the correction phase).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main() {
  hipFree(nullptr);
  for (double gb : {21.0, 21.0, 28.0, 0.25, 4.0, 8.0, 16.0, 21.0}) {
    const size_t n = (size_t)(gb * (1ull << 30));
    char* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc((void**)&p, n);
    double t1 = now();
    hipLaunchKernelGGL(touch, dim3((unsigned)((n / 4096 + 255) / 256)), dim3(256), 0, 0, p, n);
    hipDeviceSynchronize();
    double t2 = now();
    hipLaunchKernelGGL(touch, dim3((unsigned)((n / 4096 + 255) / 256)), dim3(256), 0, 0, p, n);
    hipDeviceSynchronize();
    double t3 = now();
    hipFree(p);
    double t4 = now();
    printf("hipMalloc %6.2f GB: %.3f s (%s), first touch of every page %.3f s, second %.3f s, hipFree %.3f s\n", gb, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2, t4 - t3);
  }
  for (double gb : {0.064, 0.35, 1.0}) {
    const size_t n = (size_t)(gb * (1ull << 30));
    void* h = nullptr;
    double t0 = now();
    hipHostMalloc(&h, n);
    double t1 = now();
    hipHostFree(h);
    printf("hipHostMalloc %5.3f GB: %.3f s, free %.3f s\n", gb, t1 - t0, now() - t1);
  }
  return 0;
}
