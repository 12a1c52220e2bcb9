// Calibration for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters in K1's access pattern: streams N
// doubles with 8-byte-per-lane coalesced loads and stores (SoA sweep), a known byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i] * 1.0000001;
}
int main() {
  const size_t n = (size_t)64 << 20;   // 64 Mi doubles = 512 MiB read + 512 MiB written
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  for (int it = 0; it < 3; it++) calib_copy8<<<4096, 256>>>(a, b, n);
  hipDeviceSynchronize();
  printf("calib_copy8: %zu bytes read, %zu bytes written per launch\n", n * 8, n * 8);
  return 0;
}
