// PMC calibration (profiling aid, not part of the product): stream a buffer of known size with the sampler's
// access width (8 bytes per lane, 512 contiguous bytes per wavefront) so that FETCH_SIZE / WRITE_SIZE can be
// converted into bytes for this pattern (MI355X_MICROARCH.md "HBM": other widths than 16 B/lane are uncalibrated).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libpmc_calib.so tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void calib_read_f64(const double *p, size_t n, double *sink) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 1.2345e300) sink[0] = s;
}
__global__ void calib_write_f64(double *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}

extern "C" int pmc_calib(size_t n_doubles) {
  double *p = nullptr;
  if (hipMalloc(&p, n_doubles * 8 + 8) != hipSuccess) return -1;
  hipMemset(p, 0, n_doubles * 8 + 8);
  hipDeviceSynchronize();
  calib_read_f64<<<4096, 256>>>(p, n_doubles, p + n_doubles);
  hipDeviceSynchronize();
  calib_write_f64<<<4096, 256>>>(p, n_doubles);
  hipDeviceSynchronize();
  hipFree(p);
  return 0;
}
