// pps_front_duo.h -- TWO wavefronts under one register-resident front (round 5).
//   reference: the numeric phase of cholmod_factorize for one supernode (isamlib/Cholesky.cpp:100-128)
//
// A front of a corridor tree is a dependent chain of ~16 k cycles on a lone wave, of which only the pivot chain (~5 k) is inherently
// serial; the rest is data movement -- extend-add, LDS triangle -> tiles, trailing MFMAs, update matrix -> HBM -- that a wave with one
// instruction in flight every ~9 cycles does slowly.  From the second level of a band group on, half of the workgroup's waves idle.  Such
// a wave becomes the HELPER of a front whose pivots fit one tile column (p <= 16: every separator front of a corridor tree):
//
//   owner                                                helper
//   clear + original entries (unless pre-assembled)  -A->
//   extend-add, target rows below r1                      extend-add, target rows from r1 on          (DevGraph::c_split; disjoint entries)
//                                                    <-B-  -C->
//   tiles (ti, 0) and the rhs row from the triangle       tiles (ti, tj >= 1) from the triangle
//                                                    <-T-                                             (the triangle is dead: panel buffers)
//   per 4-column panel k: extract, pivot block, L out -Sk-> operands of panel k, MFMAs on its tiles
//     MFMAs on (ti, 0)
//   update matrix: columns p .. 15, rhs row                update matrix: columns 16 ..
//
// Every hand-over is a flag in LDS carrying the level's epoch (never reset inside a launch), written behind a release fence over the
// LDS and read in front of an acquire fence (workgroup scope, local address space only: no wait for stores to HBM).  The helper's
// tiles are not needed by the owner again (p <= 16: no panel ever starts in tile column 1), so nothing flows back during the
// elimination.  Each panel has a buffer of its own (kDuoQ doubles, in the owner's dead triangle): the owner never overwrites what the
// helper may still be reading.
// Arithmetic: the operations of front_reg_eliminate<NT, false, false, 4> on the same operands in the same order -- per entry the
// original value, child 0, child 1, then one MFMA per panel; the pivot block is chol4 / trsm4 / rank4 -- so a front comes out
// bit-identical whether one wave or two eliminate it (tests/test_gpu_duo.py: the `make duo` build against the shipped one).
// NOT SHIPPED: measured slower than one wave per front; compiled in only with -DPPS_DUO_MODE=1..3.
#pragma once
#include "pps_front_reg.h"

namespace pps {

constexpr int kDuoQStride = 5;                       // doubles per row of a panel buffer: 4 columns + 1 pad
constexpr int kDuoQ = 64 * kDuoQStride;              // one panel buffer
constexpr int kDuoPanels = 4;                        // p <= 16, four columns per panel
constexpr int kDuoFlags = 8;                         // ints per owner wave: A, B, C, T, S0 .. S3
constexpr int kDuoMail = 80;                         // ints per owner wave: the front's record (16) and its child records (64), owner -> helper
enum { DF_A = 0, DF_B = 1, DF_C = 2, DF_T = 3, DF_S = 4 };

__device__ __forceinline__ void duo_post(int* flag, int epoch) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// bounded like flow_wait (pps_k3.hip): a partner that never arrives raises the status word instead of hanging the device
template <class G>
__device__ __forceinline__ void duo_wait(const G& d, int* flag, int epoch) {
  int spin = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != epoch) {
    if (++spin >= (1 << 22)) { if ((threadIdx.x & 63) == 0) d.result_dev[2] = kStatusInternal; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// waits for +epoch (true) or -epoch (false): the owner's verdict whether the front takes a helper at all
template <class G>
__device__ __forceinline__ bool duo_wait_either(const G& d, int* flag, int epoch) {
  int spin = 0, v;
  while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != epoch && v != -epoch) {
    if (++spin >= (1 << 22)) { if ((threadIdx.x & 63) == 0) d.result_dev[2] = kStatusInternal; v = -epoch; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  return v == epoch;
}

// index of tile (ti, tj), 1 <= tj <= ti, among the helper's tiles
__device__ __forceinline__ constexpr int duo_hid(int ti, int tj) { return (ti - 1) * ti / 2 + (tj - 1); }

// ---- the owner: tile column 0, the rhs row, the pivot chain ----
template <int NT, class G>
__device__ __forceinline__ void front_duo_owner(const G& d, int rec, double* F, int* fl, int epoch) {
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2);
  const int l16 = lane & 15, lq = lane >> 4;
  const int f = p + b, fa = f + 1;
  double4_t c[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + lq + 4 * r;
      c[ti][r] = F[tri24(row < f ? row : 0) + l16];
    }
  double y;
  { const double t = F[lane <= f ? tri24(f) + lane : 0]; y = lane < f ? t : 0.0; }
  __builtin_amdgcn_wave_barrier();
  double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  duo_wait(d, fl + DF_T, epoch);                     // the helper holds its tiles: the triangle is dead, its head becomes the panel buffers
  bool bad = false;
  int k = 0;
  for (int K = 0; K < p; K += 4, k++) {
    const int nb = p - K < 4 ? p - K : 4;
    double* Q = F + k * kDuoQ;
    const int m = l16 - K;
    if (m >= 0 && m < 4) {
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int r = 0; r < 4; r++) Q[(16 * ti + lq + 4 * r) * kDuoQStride + m] = c[ti][r];
    }
    __builtin_amdgcn_wave_barrier();
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = Q[lane * kDuoQStride + q];
#pragma unroll
    for (int q = 0; q < 4; q++) { const double t = readlane_d(y, K + q); r[q] = lane == f ? t : r[q]; }
    const Chol4 c1 = chol4(readlane_d(r[0], K), readlane_d(r[0], K + 1), readlane_d(r[1], K + 1), readlane_d(r[0], K + 2), readlane_d(r[1], K + 2),
                           readlane_d(r[2], K + 2), readlane_d(r[0], K + 3), readlane_d(r[1], K + 3), readlane_d(r[2], K + 3), readlane_d(r[3], K + 3), nb, bad);
    double x[4];
    trsm4(c1, r[0], r[1], r[2], r[3], x[0], x[1], x[2], x[3]);
#pragma unroll
    for (int q = 0; q < 4; q++) Q[lane * kDuoQStride + q] = x[q];
    duo_post(fl + DF_S + k, epoch);                  // panel k is in its buffer
    y = rank4(y, x[0], x[1], x[2], x[3], readlane_d(x[0], f), readlane_d(x[1], f), readlane_d(x[2], f), readlane_d(x[3], f));
    if (lane < fa) {
      double* __restrict__ lrow = Lp + (unsigned)(__mul24(lane, p) + K);
      lrow[0] = x[0];
      if (nb > 1) lrow[1] = x[1];
      if (nb > 2) lrow[2] = x[2];
      if (nb > 3) lrow[3] = x[3];
    }
    __builtin_amdgcn_wave_barrier();
    // rank-nb update of tile column 0
    const bool kvalid = lq < nb;
    double a[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { const double v = Q[(16 * t + l16) * kDuoQStride + lq]; a[t] = kvalid ? v : 0.0; }
#pragma unroll
    for (int ti = 0; ti < NT; ti++) c[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[ti], a[0], c[ti], 0, 0, 0);
  }
  if (bad && lane == 0) d.result_dev[2] = 1.0;       // not positive definite
  // update matrix: columns p .. 15 of tile column 0, and the rhs row
  double* __restrict__ Us = d.U + (((long long)__builtin_amdgcn_readlane(rec, 12) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 11));
  const unsigned trash_u = (unsigned)(__mul24(b + 1, b + 1) - 1);
  if (p < 16) {
#pragma unroll
    for (int ti = 0; ti < NT; ti++) {
      if (16 * ti >= f) continue;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + lq + 4 * q;
        const bool ok = row < f && l16 <= row && l16 >= p;
        Us[ok ? (unsigned)(tri24(row - p) - p + l16) : trash_u] = c[ti][q];
      }
    }
  }
  Us[(lane >= p && lane <= f) ? (unsigned)(tri24(b) + lane - p) : trash_u] = lane < f ? y : 0.0;
}

// ---- the helper: tiles (ti, tj >= 1) ----
template <int NT, class G>
__device__ __forceinline__ void front_duo_helper(const G& d, int rec, double* F, int* fl, int epoch) {
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2);
  const int l16 = lane & 15, lq = lane >> 4;
  const int f = p + b;
  constexpr int NH = NT * (NT - 1) / 2;
  double4_t c[NH];
#pragma unroll
  for (int ti = 1; ti < NT; ti++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + lq + 4 * r;
      const double* Fr = F + (tri24(row < f ? row : 0) + l16);
#pragma unroll
      for (int tj = 1; tj <= ti; tj++) c[duo_hid(ti, tj)][r] = Fr[16 * tj];
    }
  duo_post(fl + DF_T, epoch);                        // (the release fence waits for the reads above)
  int k = 0;
  for (int K = 0; K < p; K += 4, k++) {
    const int nb = p - K < 4 ? p - K : 4;
    const double* Q = F + k * kDuoQ;
    duo_wait(d, fl + DF_S + k, epoch);
    const bool kvalid = lq < nb;
    double a[NT];
#pragma unroll
    for (int t = 1; t < NT; t++) { const double v = Q[(16 * t + l16) * kDuoQStride + lq]; a[t] = kvalid ? v : 0.0; }
#pragma unroll
    for (int ti = 1; ti < NT; ti++)
#pragma unroll
      for (int tj = 1; tj <= ti; tj++) c[duo_hid(ti, tj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[ti], a[tj], c[duo_hid(ti, tj)], 0, 0, 0);
  }
  double* __restrict__ Us = d.U + (((long long)__builtin_amdgcn_readlane(rec, 12) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 11));
  const unsigned trash_u = (unsigned)(__mul24(b + 1, b + 1) - 1);
#pragma unroll
  for (int ti = 1; ti < NT; ti++) {
    if (16 * ti >= f) continue;                      // (wave-uniform)
    int rbase[4]; bool rok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int row = 16 * ti + lq + 4 * q; rok[q] = row < f; rbase[q] = tri24(row - p) - p; }
#pragma unroll
    for (int tj = 1; tj <= ti; tj++) {
      const int col = 16 * tj + l16;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + lq + 4 * q;
        const bool ok = rok[q] && col <= row;        // (col >= 16 >= p)
        Us[ok ? (unsigned)(rbase[q] + col) : trash_u] = c[duo_hid(ti, tj)][q];
      }
    }
  }
}

}  // namespace pps
