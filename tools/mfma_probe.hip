// probe of v_mfma_f64_16x16x4_f64 operand layout on gfx950: prints which (row, col) each lane/register holds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  // hypothesis: A[i = l%16][k = l/16], B[k = l/16][j = l%16]
  const int i = l & 15, k = l >> 4;
  const double a = 1.0 + i + 100.0 * k;        // A[i][k]
  const double b = 1.0 + 3.0 * i + 7.0 * k;    // B[k][j=i]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main() {
  double* d; hipMalloc(&d, 256 * 8);
  probe<<<1, 64>>>(d);
  double h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // reference D[i][j] = sum_k A[i][k] B[k][j]
  double D[16][16];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += (1.0 + i + 100.0 * k) * (1.0 + 3.0 * j + 7.0 * k); D[i][j] = s; }
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    if (h[l * 4 + r] != D[4 * (l / 16) + r][l % 16]) ok1 = 0;     // row = 4*(l/16)+r, col = l%16
    if (h[l * 4 + r] != D[(l / 16) + 4 * r][l % 16]) ok2 = 0;     // row = (l/16)+4r, col = l%16
  }
  printf("layout row=4*(l/16)+r: %d   layout row=(l/16)+4r: %d\n", ok1, ok2);
  printf("lane0: %g %g %g %g ; lane16: %g %g %g %g ; D[0][0]=%g D[1][0]=%g D[4][0]=%g D[0][1]=%g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67], D[0][0], D[1][0], D[4][0], D[0][1]);
  return 0;
}
