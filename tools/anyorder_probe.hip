// probe: does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h says: not supported on GFX9xx.)
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  int* d; hipMalloc(&d, 64);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
  for (int flags = 0; flags < 2; flags++) {
    for (int rep = 0; rep < 3; rep++) {
      hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, e[0], e[1], 0, 200000LL, d);          // ~ 2 ms at 100 MHz clock64
      hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, e[2], e[3], flags, 200000LL, d);
      hipStreamSynchronize(st);
      float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[2], e[3]); hipEventElapsedTime(&c, e[0], e[3]);
      printf("flags %d: kernel A %.3f ms, kernel B %.3f ms, start A -> end B %.3f ms (%s)\n", flags, a, b, c, c < 0.75f * (a + b) ? "overlapped" : "serial");
    }
  }
  // the same without per-kernel events (a start / stop event pair may order the dispatch by itself): total time between two recorded events
  for (int flags = 0; flags < 2; flags++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e[0], st);
      hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, 200000LL, d);
      hipExtLaunchKernelGGL(spin, dim3(8), dim3(64), 0, st, nullptr, nullptr, flags, 200000LL, d);
      hipEventRecord(e[3], st);
      hipStreamSynchronize(st);
      float c; hipEventElapsedTime(&c, e[0], e[3]);
      printf("no kernel events, flags %d: both kernels %.3f ms (%s)\n", flags, c, c < 0.13f ? "overlapped" : "serial");
    }
  }
  return 0;
}
