// pps_regtile.h -- a <= 64-row symmetric matrix held by ONE wavefront as ten 16x16 fp64 tiles in the MFMA
// accumulator layout, and the pieces of its blocked Cholesky (shared by the band kernels and the dense-front panel).
#pragma once
#ifndef PPS_WAVE_EMU        // (tests/cpp/wave_emu.h compiles this header for the host, one coroutine per lane)
#include <hip/hip_runtime.h>
#endif

namespace pps {

#ifndef PPS_WAVE_EMU
typedef double double4_t __attribute__((ext_vector_type(4)));
#endif

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

// The status word of a solve (result_dev[2]: 0 ok | 1 not positive definite | >= 64 an internal time-out) is RAISED, never overwritten: a not-PD
// front that reports after a hand-over time-out must not turn the internal error into an ordinary rejected step.  Non-negative doubles order
// like their bit patterns, so the maximum is one integer atomic without a return value.
#ifndef PPS_WAVE_EMU
__device__ __forceinline__ void raise_status(double* w, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(w), (unsigned long long)__double_as_longlong(v));
}
#else
inline void raise_status(double* w, double v) { if (v > *w) *w = v; }
#endif

// broadcast lane `l` (wave-uniform) of a double through two v_readlane_b32
#ifndef PPS_WAVE_EMU
__device__ __forceinline__ double readlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
#endif


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

// ------------------------------------------------------------------------------------------
// 8-column panels (round 4).  A panel step moves EIGHT columns of the current tile column through LDS at once -- one
// extraction, one read-back, one operand gather per eight pivots instead of two of each -- and factors them as two chained
// 4 x 4 pivot blocks in registers: block 1 (columns 0 .. 3), the rank-4 update of columns 4 .. 7 by it (lane = row: four
// fused multiply-adds per column, the sixteen L21 entries broadcast with v_readlane), block 2.  The trailing update of a tile is
// two back-to-back v_mfma_f64_16x16x4_f64 (k = columns 0 .. 3, then 4 .. 7).
// ------------------------------------------------------------------------------------------
constexpr int kP8Stride = 9;                      // doubles per panel row: eight columns + one pad (odd: conflict-free row reads)

template <int TJ, int NT>
__device__ __forceinline__ void reg_extract_panel8(const double4_t (&c)[NT * (NT + 1) / 2], double* P, int c0, int lane) {
  const int l16 = lane & 15, lq = lane >> 4;
  const int m = l16 - c0;                         // c0 = 0 or 8: half of the lanes hold a panel column
  if (m >= 0 && m < 8) {
#pragma unroll
    for (int ti = TJ; ti < NT; ti++)
#pragma unroll
      for (int r = 0; r < 4; r++) P[(16 * ti + lq + 4 * r) * kP8Stride + m] = c[tile_id(ti, TJ)][r];
  }
}

// Cholesky of a 4 x 4 diagonal block, evaluated redundantly by every lane from broadcast values: reciprocal pivots i_k and the
// strictly lower entries.  nb = columns of the block that exist (a narrower block, the last one of a front, zeroes the factors
// of the columns it does not have).  The reciprocal square root is evaluated unconditionally -- NaN for a pivot that is not
// positive -- and selected away: a branch per pivot would split the serial chain into basic blocks.
struct Chol4 { double i0, i1, i2, i3, l10, l20, l30, l21, l31, l32; };
__device__ __forceinline__ Chol4 chol4(double d00, double d10, double d11, double d20, double d21, double d22, double d30, double d31,
                                       double d32, double d33, int nb, bool& bad) {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // the serial pivot chain: a - b * c is one operation here
#endif
  Chol4 o;
  { const double q = rsqrt_nr(d00); const bool on = nb > 0; bad |= on && !(d00 > 0.0); o.i0 = (on && d00 > 0.0) ? q : 0.0; o.l10 = d10 * o.i0; o.l20 = d20 * o.i0; o.l30 = d30 * o.i0; }
  { const double t = d11 - o.l10 * o.l10; const double q = rsqrt_nr(t); const bool on = nb > 1; bad |= on && !(t > 0.0); o.i1 = (on && t > 0.0) ? q : 0.0; o.l21 = (d21 - o.l20 * o.l10) * o.i1; o.l31 = (d31 - o.l30 * o.l10) * o.i1; }
  { const double t = d22 - o.l20 * o.l20 - o.l21 * o.l21; const double q = rsqrt_nr(t); const bool on = nb > 2; bad |= on && !(t > 0.0); o.i2 = (on && t > 0.0) ? q : 0.0; o.l32 = (d32 - o.l30 * o.l20 - o.l31 * o.l21) * o.i2; }
  { const double t = d33 - o.l30 * o.l30 - o.l31 * o.l31 - o.l32 * o.l32; const double q = rsqrt_nr(t); const bool on = nb > 3; bad |= on && !(t > 0.0); o.i3 = (on && t > 0.0) ? q : 0.0; }
  return o;
}
// one row against the block: x = r L^-T
__device__ __forceinline__ void trsm4(const Chol4& c, double r0, double r1, double r2, double r3, double& x0, double& x1, double& x2, double& x3) {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)
#endif
  x0 = r0 * c.i0;
  x1 = (r1 - x0 * c.l10) * c.i1;
  x2 = (r2 - x0 * c.l20 - x1 * c.l21) * c.i2;
  x3 = (r3 - x0 * c.l30 - x1 * c.l31 - x2 * c.l32) * c.i3;
}
// r -= x0 a0 + x1 a1 + x2 a2 + x3 a3 (columns 4 .. 7 of a row take the rank-4 update of block 1 before block 2 is factored)
__device__ __forceinline__ double rank4(double r, double x0, double x1, double x2, double x3, double a0, double a1, double a2, double a3) {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)
#endif
  return r - x0 * a0 - x1 * a1 - x2 * a2 - x3 * a3;
}

// Rank-nb update (nb <= 8) of every live tile: per tile one MFMA for panel columns 0 .. 3 and, when the panel is wider than four
// columns, a second one for columns 4 .. 7.  The tiles of the NEXT panel's tile column (tjn = TJ or TJ + 1) are finished first:
// the next step extracts its panel from them and then spends ~1000 cycles on the pivot blocks, during which the remaining
// (independent) MFMAs drain in the matrix core instead of being waited for.
template <int TJ, int NT>
__device__ __forceinline__ void reg_trailing8(double4_t (&c)[NT * (NT + 1) / 2], const double* P, int nb, int lane, int tjn, bool short_of_end) {
  const int l16 = lane & 15, lq = lane >> 4;
  const bool v0 = lq < nb, v1 = lq + 4 < nb;
  const bool two = nb > 4;                                      // (wave-uniform)
  double a0[NT], a1[NT];
#pragma unroll
  for (int t = TJ; t < NT; t++) {
    const double x = P[(16 * t + l16) * kP8Stride + lq], z = P[(16 * t + l16) * kP8Stride + 4 + lq];
    a0[t] = v0 ? x : 0.0; a1[t] = v1 ? z : 0.0;
  }
  constexpr int TN = TJ + 1 < NT ? TJ + 1 : NT - 1;             // (clamped for the instantiation TJ = NT - 1, where it is not used)
  if (tjn == TJ) {
#pragma unroll
    for (int ti = TJ; ti < NT; ti++) c[tile_id(ti, TJ)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[ti], a0[TJ], c[tile_id(ti, TJ)], 0, 0, 0);
    if (two) {
#pragma unroll
      for (int ti = TJ; ti < NT; ti++) c[tile_id(ti, TJ)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[ti], a1[TJ], c[tile_id(ti, TJ)], 0, 0, 0);
    }
#pragma unroll
    for (int ti = TJ + 1; ti < NT; ti++)
#pragma unroll
      for (int tj = TJ + 1; tj <= ti; tj++) c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[ti], a0[tj], c[tile_id(ti, tj)], 0, 0, 0);
    if (two) {
#pragma unroll
      for (int ti = TJ + 1; ti < NT; ti++)
#pragma unroll
        for (int tj = TJ + 1; tj <= ti; tj++) c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[ti], a1[tj], c[tile_id(ti, tj)], 0, 0, 0);
    }
  } else {
    // the panel ends its tile column: column TJ is dead from here on, TJ + 1 is the next panel's
    if (TJ + 1 < NT) {
#pragma unroll
      for (int ti = TJ + 1; ti < NT; ti++) c[tile_id(ti, TN)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[ti], a0[TN], c[tile_id(ti, TN)], 0, 0, 0);
      if (two) {
#pragma unroll
        for (int ti = TJ + 1; ti < NT; ti++) c[tile_id(ti, TN)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[ti], a1[TN], c[tile_id(ti, TN)], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ti = TJ + 2; ti < NT; ti++)
#pragma unroll
      for (int tj = TJ + 2; tj <= ti; tj++) c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[ti], a0[tj], c[tile_id(ti, tj)], 0, 0, 0);
    if (two) {
#pragma unroll
      for (int ti = TJ + 2; ti < NT; ti++)
#pragma unroll
        for (int tj = TJ + 2; tj <= ti; tj++) c[tile_id(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[ti], a1[tj], c[tile_id(ti, tj)], 0, 0, 0);
    }
    if (short_of_end) {
      // the last panel of a front stops short of the tile column's end: the columns behind it are boundary columns, whose
      // entries in tile column TJ go into the update matrix
#pragma unroll
      for (int ti = TJ; ti < NT; ti++) c[tile_id(ti, TJ)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[ti], a0[TJ], c[tile_id(ti, TJ)], 0, 0, 0);
      if (two) {
#pragma unroll
        for (int ti = TJ; ti < NT; ti++) c[tile_id(ti, TJ)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[ti], a1[TJ], c[tile_id(ti, TJ)], 0, 0, 0);
      }
    }
  }
}

}  // namespace pps
