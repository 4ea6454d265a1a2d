// How many one-wave workgroups with S bytes of LDS are resident on a CU at once (measured, not asked): N x CUs workgroups
// that each spin 200 us finish in one round (~200 us) while N fits, in two (~400 us) when it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S>
__global__ void __launch_bounds__(64) k(int* out) {
  __shared__ int lds[S / 4];
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 20000ull) { lds[(threadIdx.x * 7) % (S / 4)] += 1; }   // 200 us at 100 MHz
  if (out && lds[threadIdx.x] == -12345) out[0] = 1;
}
template <int S> void run(int cus) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int fit = 0;
  for (int n = 12; n <= 32; ++n) {
    hipLaunchKernelGGL(k<S>, dim3(cus * n), dim3(64), 0, 0, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<S>, dim3(cus * n), dim3(64), 0, 0, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < 0.3f) fit = n;
  }
  printf("LDS %5d B per wave: %d waves per CU resident\n", S, fit);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<5120>(cus); run<6144>(cus); run<6400>(cus); run<6656>(cus); run<6912>(cus); run<7168>(cus); run<7680>(cus); run<8192>(cus); run<9216>(cus);
  return 0;
}
