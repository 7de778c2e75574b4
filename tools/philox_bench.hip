// Microbenchmark (measurement aid): Philox4x32-10 blocks per second on one GPU, for the roofline of the sampler's
// fast path.  hipcc --offload-arch=gfx950 -O3 -o philox_bench tools/philox_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../mchap_amd/csrc/philox.hpp"
template <int ILP>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters) {
  uint32_t acc = 0;
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  for (int i = 0; i < iters; i += ILP) {
    uint32_t o[ILP][4];
#pragma unroll
    for (int j = 0; j < ILP; j++) mchap::philox4x32_10(i + j, 0, id, 7, 42, 43, o[j]);
#pragma unroll
    for (int j = 0; j < ILP; j++) acc ^= o[j][0] ^ o[j][1] ^ o[j][2] ^ o[j][3];
  }
  out[id] = acc;
}
template <int ILP>
void run(int waves, int iters, uint32_t *d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<ILP>, dim3(waves), dim3(64), 0, 0, d, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<ILP>, dim3(waves), dim3(64), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double blocks = (double)waves * 64 * iters;
  printf("ILP %d waves %5d iters %d: %.3f ms, %.3e blocks/s, %.1f cycles per wave-block at 2.4 GHz x 1024 SIMDs (waves/SIMD %.2f)\n", ILP, waves, iters, ms,
         blocks / (ms * 1e-3), (ms * 1e-3 * 2.4e9) / ((double)iters * ((waves + 1023) / 1024)), waves / 1024.0);
}
int main() {
  uint32_t *d; hipMalloc(&d, 64 * 65536 * 4);
  for (int waves : {512, 1024, 2048, 4096, 8192}) { run<1>(waves, 4096, d); run<2>(waves, 4096, d); run<4>(waves, 4096, d); }
  return 0;
}
