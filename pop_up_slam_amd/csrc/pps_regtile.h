// pps_regtile.h -- a <= 64-row symmetric matrix held by ONE wavefront as ten 16x16 fp64 tiles in the MFMA
// accumulator layout, and the pieces of its blocked Cholesky (shared by the band kernels and the dense-front panel).
#pragma once
#include <hip/hip_runtime.h>

namespace pps {

typedef double double4_t __attribute__((ext_vector_type(4)));

// 1/sqrt(x) to fp64 round-off: hardware estimate + two Newton steps (shorter than div + sqrt on the pivot chain).
// Written with explicit fused multiply-adds -- three dependent operations per step instead of five: this sits on the
// serial pivot chain of every panel (the library is otherwise built with -ffp-contract=off for the fp32 pop-up parity).
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const double t = x * y, hy = 0.5 * y;
#ifdef PPS_NO_FMA                                      // diagnostic build (make nofma): no contracted operation anywhere
    const double e = 0.5 - t * hy;
    y = y + y * e;
#else
    const double e = __builtin_fma(-t, hy, 0.5);       // 0.5 - 0.5 x y^2
    y = __builtin_fma(y, e, y);
#endif
  }
  return y;
}

// broadcast lane `l` (wave-uniform) of a double through two v_readlane_b32
__device__ __forceinline__ double readlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}


constexpr int kPStride = 5;                       // doubles per panel row: conflict-free operand gathers
constexpr int kRegRows = 64;                      // rows held as register tiles (one lane per row in the panel step)
constexpr int kRegRowsMax = 80;                   // ... plus up to 16 boundary rows kept as a strip of the LDS triangle (pps_k3.hip)

__device__ __forceinline__ constexpr int tile_id(int ti, int tj) { return ti * (ti + 1) / 2 + tj; }

// NT = tile rows held (5: up to 80 rows, 15 tiles; 4: up to 64 rows, 10 tiles; 3: up to 48 rows, 6 tiles -- same tile_id numbering)
template <int TJ, int NT = 4>
__device__ __forceinline__ void reg_extract_panel(const double4_t (&c)[NT * (NT + 1) / 2], double* __restrict__ P, int c0, int lane) {
  const int l16 = lane & 15, lq = lane >> 4;
  const int m = l16 - c0;
  if (m >= 0 && m < 4) {
#pragma unroll
    for (int ti = TJ; ti < NT; ti++)
#pragma unroll
      for (int r = 0; r < 4; r++) P[(16 * ti + lq + 4 * r) * kPStride + m] = c[tile_id(ti, TJ)][r];
  }
}

// Rank-nb update of every live tile.  The tiles of the NEXT panel's tile column (TJN = TJ or TJ+1) are issued first:
// the next iteration extracts its panel from them and then spends ~1000 cycles of VALU / LDS work on the panel
// factorisation, during which the remaining (independent) MFMAs drain in the matrix core instead of being waited for.
template <int TJ, int NT = 4>
__device__ __forceinline__ void reg_trailing(double4_t (&c)[NT * (NT + 1) / 2], const double* __restrict__ P, int nb, int lane, int tjn = TJ) {
  const int l16 = lane & 15, lq = lane >> 4;
  const bool kvalid = lq < nb;
  double opnd[NT];
#pragma unroll
  for (int t = TJ; t < NT; t++) { const double x = P[(16 * t + l16) * kPStride + lq]; opnd[t] = kvalid ? x : 0.0; }
  if (tjn == TJ) {
#pragma unroll
    for (int ti = TJ; ti < NT; ti++)
      c[tile_id(ti, TJ)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd[ti], opnd[TJ], c[tile_id(ti, TJ)], 0, 0, 0);
#pragma unroll
    for (int ti = TJ + 1; ti < NT; ti++)
#pragma unroll
      for (int tj = TJ + 1; tj <= ti; tj++)
        c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd[ti], opnd[tj], c[tile_id(ti, tj)], 0, 0, 0);
  } else {
    if (TJ + 1 < NT) {
#pragma unroll
      for (int ti = TJ + 1; ti < NT; ti++)
        c[tile_id(ti, TJ + 1 < NT ? TJ + 1 : NT - 1)] =
            __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd[ti], opnd[TJ + 1 < NT ? TJ + 1 : NT - 1], c[tile_id(ti, TJ + 1 < NT ? TJ + 1 : NT - 1)], 0, 0, 0);
    }
#pragma unroll
    for (int ti = TJ; ti < NT; ti++)
#pragma unroll
      for (int tj = TJ; tj <= ti; tj++)
        if (tj != TJ + 1)
          c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-opnd[ti], opnd[tj], c[tile_id(ti, tj)], 0, 0, 0);
  }
}

}  // namespace pps
