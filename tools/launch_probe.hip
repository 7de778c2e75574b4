// Measurement aid: what does a launch of many early-exit workgroups cost, by block size / dynamic LDS / scratch?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int WPE, bool SCR>
__global__ __launch_bounds__(NT, WPE) void probe(const int *flag, double *out) {
  extern __shared__ unsigned char smem[];
  if (flag[blockIdx.x & 1023] != 0) return;
  double a[SCR ? 96 : 1];
  for (int i = 0; i < (SCR ? 96 : 1); i++) a[i] = out[i + threadIdx.x];
  __syncthreads();
  smem[threadIdx.x] = (unsigned char)a[flag[5] & (SCR ? 63 : 0)];
  out[threadIdx.x] = smem[threadIdx.x ^ 1] + a[flag[7] & (SCR ? 63 : 0)];
}
template <int NT, int WPE, bool SCR>
void run(const char *name, int grid, size_t lds, int *flag, double *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = probe<NT, WPE, SCR>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, flag, out);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, 0, flag, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s grid %6d lds %6zu: %8.3f us per launch\n", name, grid, lds, ms * 100.0);
}
int main() {
  int *flag; double *out;
  hipMalloc(&flag, 4096); hipMalloc(&out, 1 << 20);
  int h[1024]; for (int i = 0; i < 1024; i++) h[i] = 1;
  hipMemcpy(flag, h, 4096, hipMemcpyHostToDevice);
  run<64, 8, false>("64 thr, no scratch", 20000, 0, flag, out);
  run<64, 8, false>("64 thr, no scratch, lds 10K", 20000, 10240, flag, out);
  run<256, 4, false>("256 thr, no scratch", 20000, 0, flag, out);
  run<256, 4, false>("256 thr, no scratch, lds 40K", 20000, 40768, flag, out);
  run<256, 4, false>("256 thr, no scratch, lds 20K", 20000, 20480, flag, out);
  run<256, 4, false>("256 thr, no scratch, lds 80K", 20000, 81920, flag, out);
  run<256, 8, true>("256 thr, scratch (wpe 8)", 20000, 0, flag, out);
  run<256, 8, true>("256 thr, scratch, lds 40K", 20000, 40768, flag, out);
  run<64, 8, true>("64 thr, scratch", 80000, 0, flag, out);
  run<64, 8, false>("64 thr, no scratch, 80000", 80000, 0, flag, out);
  run<128, 4, false>("128 thr, lds 20K", 40000, 20480, flag, out);
  return 0;
}
