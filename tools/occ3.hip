// How many one-wave workgroups with S bytes of private (scratch) memory per lane are resident on a CU at once (measured as
// in occ2.hip): is the search kernel's residency limited by its scratch size?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S>
__global__ void __launch_bounds__(64) k(int* out, int sel) {
  volatile int priv[S / 4];
  for (int i = 0; i < S / 4; i += 16) priv[i] = i + threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int acc = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < 20000ull) { acc += priv[(sel + acc) & (S / 4 - 1) & ~15]; }   // 200 us at 100 MHz
  if (out && acc == -12345) out[0] = 1;
}
template <int S> void run(int cus) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int fit = 0;
  for (int n = 8; n <= 32; ++n) {
    hipLaunchKernelGGL(k<S>, dim3(cus * n), dim3(64), 0, 0, nullptr, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<S>, dim3(cus * n), dim3(64), 0, 0, nullptr, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < 0.3f) fit = n;
  }
  printf("scratch %5d B per lane (%4d KB per wave): %d waves per CU resident\n", S, S * 64 / 1024, fit);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<256>(cus); run<512>(cus); run<1024>(cus); run<1536>(cus); run<2048>(cus); run<2560>(cus); run<4096>(cus);
  return 0;
}
