// pps_front_reg.h -- the elimination of ONE register-resident front by ONE wavefront (second half of K3's per-front work):
// the assembled packed triangle in LDS -> register tiles -> 8-column panels -> factor panel and update matrix in HBM.
//   reference: the numeric phase of cholmod_factorize for one supernode (isamlib/Cholesky.cpp:100-128)
// Shared by the band kernels, the level-per-launch kernels and the one-front harness of pps_k3.hip.  The header holds no kernel:
// tests/cpp/wave_emu.h compiles it for the host as well (PPS_WAVE_EMU: 64 coroutines in lockstep at every cross-lane operation),
// so that the panel logic is checked against numpy in the CPU suite, before a GPU sees it.
#pragma once
#include <type_traits>

#include "pps_regtile.h"

namespace pps {

// PPS_TRACE=1 instrumentation: lane 0 stamps s_memtime at phase boundaries of a front
#ifndef PPS_TR
#define PPS_TR(k) do { if (d.trace && lane == 0) d.trace[(size_t)s * 8 + (k)] = clock64(); } while (0)
#endif

__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }
// the same for 0 <= i < 4096 with the full-rate 24-bit multiplier (per-lane index arithmetic of the register-tile code)
__device__ __forceinline__ int tri24(int i) { return __mul24(i, i + 1) >> 1; }

// G: anything with L, U, result_dev, trace (DevGraph).  rec: the packed front record, one field per lane (0 front, 1 p, 2 b,
// 9/10 offset of the factor panel in L, 11/12 offset of the update matrix in U).
// NT: tile rows held (2 .. 5).  TR: in-kernel phase trace compiled in.
// STRIP: fronts of 65 .. 80 rows -- rows 0 .. 63 live in the register tiles as usual, rows 64 .. fa-1 (boundary rows: the pivots are
// among the first 64) either stay where the assembly put them, in the packed LDS triangle F, carried along panel by panel by
// lanes 0 .. 15 (NT = 4), or are a fifth tile row (NT = 5).  With STRIP the rhs row sits inside the tiles / the strip; without, it
// is a vector next to them.
// P: the panel buffer, 16 NT rows of kP8Stride doubles (at least 80 rows with STRIP).  Without STRIP, and with NT = 5, F is dead
// once the tiles are loaded, and P may be F itself.
// W: pivot columns per panel step, 8 or 4 (4 = the second pivot block never runs: the form up to round 3, kept for A/B).
template <int NT, bool TR, bool STRIP, int W = 8, class G>
__device__ __forceinline__ void front_reg_eliminate(const G& d, int rec, double* F, double* P) {      // (no __restrict__: P may be F)
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readlane(rec, 0), p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2);
  const int l16 = lane & 15, lq = lane >> 4;
  const int f = p + b, fa = f + 1;
  const bool strip = STRIP && fa > kRegRows;
  constexpr bool R5 = NT == 5;                                 // rows 64 .. 79 are a fifth tile row in registers, not an LDS strip
  // The right-hand side rides along as row f of the front.  Without a strip it is kept as a VECTOR (lane = column) next to
  // the tiles instead of inside them: a front of 48 rows + rhs then needs three tile rows, not four (6 MFMA per panel instead of
  // 10, 24 tile registers to load and store instead of 40) -- every separator front of a C2 tree.  mr = rows held in the tiles.
  const int mr = STRIP ? fa : f;
  (void)s;
  // ---- packed triangle -> register tiles ----
  // One address per (tile row, register): row base + lane column, the tile columns are immediate offsets of the LDS reads.  Entries
  // that do not exist are NOT zeroed: above the diagonal of a diagonal tile the read lands in the next rows of the triangle, a row
  // past the front reads row 0 -- finite values in entries that stay dead (an MFMA update of entry (i, j) reads row i and column j
  // only; nothing stores or extracts a dead row or a column right of the diagonal), and 3 selects + an address clamp per element
  // less in front of the first panel.
  double4_t c[NT * (NT + 1) / 2];
#pragma unroll
  for (int ti = 0; ti < NT; ti++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + lq + 4 * r;
      const double* Fr = F + (tri24(row < mr ? row : 0) + l16);
#pragma unroll
      for (int tj = 0; tj <= ti; tj++) c[tile_id(ti, tj)][r] = Fr[16 * tj];
    }
  double y = 0.0;                                              // (lane f: the rhs . rhs corner, which nothing reads)
  if (!STRIP) { const double t = F[lane <= f ? tri24(f) + lane : 0]; y = lane < f ? t : 0.0; }
  __builtin_amdgcn_wave_barrier();
  double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  long long cyc_panel = 0, cyc_trail = 0;
  bool bad = false;                                            // a pivot that is not positive: reported once, after the last panel
  auto panel_step = [&](const int K) __attribute__((always_inline)) {
    const long long tk0 = (TR && d.trace) ? clock64() : 0;
    const int nb = p - K < W ? p - K : W;
    const bool two = nb > 4;                                   // (wave-uniform) the panel has a second pivot block
    const int tjK = K >> 4, c0 = K & 15;
    switch (tjK) {
      case 0: reg_extract_panel8<0, NT>(c, P, c0, lane); break;
      case 1: reg_extract_panel8<1, NT>(c, P, c0, lane); break;
      case 2: if (NT > 2) reg_extract_panel8<(NT > 2 ? 2 : 1), NT>(c, P, c0, lane); break;
      default: if (NT > 3) reg_extract_panel8<(NT > 3 ? 3 : NT - 1), NT>(c, P, c0, lane); break;
    }
    const int row2 = kRegRows + (lane & 15);                   // the strip row of this lane (lanes 0 .. 15)
    const bool has2 = STRIP && strip && lane < 16 && row2 < fa;
    if (has2 && !R5) {                                         // (with a fifth tile row the extraction above wrote these panel rows)
#pragma unroll
      for (int m = 0; m < 8; m++) P[row2 * kP8Stride + m] = F[tri(row2) + K + m];   // (K + m < 64 <= row2: inside the row)
    }
    __builtin_amdgcn_wave_barrier();
    // ---- panel: lane = row ----
    // Without a strip the rhs row takes the idle lane f (f <= 63) through the panel like any other row: its entries of columns
    // K .. K+7 sit in lanes K .. K+7 of y.
    double r[8];
#pragma unroll
    for (int m = 0; m < 8; m++) r[m] = P[lane * kP8Stride + m];
    if (!STRIP) {
#pragma unroll
      for (int m = 0; m < 8; m++) { const double q = readlane_d(y, K + m); r[m] = lane == f ? q : r[m]; }
    }
    // pivot block 1: rows / columns K .. K+3
    const Chol4 c1 = chol4(readlane_d(r[0], K), readlane_d(r[0], K + 1), readlane_d(r[1], K + 1), readlane_d(r[0], K + 2), readlane_d(r[1], K + 2),
                           readlane_d(r[2], K + 2), readlane_d(r[0], K + 3), readlane_d(r[1], K + 3), readlane_d(r[2], K + 3), readlane_d(r[3], K + 3), nb, bad);
    double x[8];
    trsm4(c1, r[0], r[1], r[2], r[3], x[0], x[1], x[2], x[3]);
    // L21 (rows K+4 .. K+7 of the solved columns 0 .. 3) reaches every lane through v_readlane; columns 4 .. 7 of every row take the
    // rank-4 update, then rows K+4 .. K+7 of them are pivot block 2
    double l21[4][4];
    Chol4 c2 = {};
    x[4] = x[5] = x[6] = x[7] = 0.0;
    if (two) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int m = 0; m < 4; m++) l21[j][m] = readlane_d(x[m], K + 4 + j);
        r[4 + j] = rank4(r[4 + j], x[0], x[1], x[2], x[3], l21[j][0], l21[j][1], l21[j][2], l21[j][3]);
      }
      c2 = chol4(readlane_d(r[4], K + 4), readlane_d(r[4], K + 5), readlane_d(r[5], K + 5), readlane_d(r[4], K + 6), readlane_d(r[5], K + 6),
                 readlane_d(r[6], K + 6), readlane_d(r[4], K + 7), readlane_d(r[5], K + 7), readlane_d(r[6], K + 7), readlane_d(r[7], K + 7), nb - 4, bad);
      trsm4(c2, r[4], r[5], r[6], r[7], x[4], x[5], x[6], x[7]);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) P[lane * kP8Stride + m] = x[m];
    if (!STRIP) {
      // rank-nb update of the rhs row: y_j -= sum_m L[f][K+m] L[j][K+m], lane j holding row j's panel entries x[0 .. 7], the solved rhs
      // entries broadcast from lane f
      y = rank4(y, x[0], x[1], x[2], x[3], readlane_d(x[0], f), readlane_d(x[1], f), readlane_d(x[2], f), readlane_d(x[3], f));
      if (two) y = rank4(y, x[4], x[5], x[6], x[7], readlane_d(x[4], f), readlane_d(x[5], f), readlane_d(x[6], f), readlane_d(x[7], f));
    }
    if (lane < fa) {                                           // (without a strip: rows 0 .. f-1 and the rhs row in lane f; with: fa > 63 rows)
      // rows above the diagonal get whatever their lanes computed: (row, col > row) of a factor panel is never read
      // (wave_front_solve, k_front_solve), and one exec-masked block with uniform branches replaces eight masked ones
      double* __restrict__ lrow = Lp + (unsigned)(__mul24(lane, p) + K);
      lrow[0] = x[0];
      if (nb > 1) lrow[1] = x[1];
      if (nb > 2) lrow[2] = x[2];
      if (nb > 3) lrow[3] = x[3];
      if (nb > 4) lrow[4] = x[4];
      if (nb > 5) lrow[5] = x[5];
      if (nb > 6) lrow[6] = x[6];
      if (nb > 7) lrow[7] = x[7];
    }
    if (STRIP && strip) {
      // rows 64 .. fa-1 (lanes 0 .. 15) through the same two triangular solves
      double q[8], yy[8];
#pragma unroll
      for (int m = 0; m < 8; m++) q[m] = P[row2 * kP8Stride + m];
      trsm4(c1, q[0], q[1], q[2], q[3], yy[0], yy[1], yy[2], yy[3]);
      yy[4] = yy[5] = yy[6] = yy[7] = 0.0;
      if (two) {
#pragma unroll
        for (int j = 0; j < 4; j++) q[4 + j] = rank4(q[4 + j], yy[0], yy[1], yy[2], yy[3], l21[j][0], l21[j][1], l21[j][2], l21[j][3]);
        trsm4(c2, q[4], q[5], q[6], q[7], yy[4], yy[5], yy[6], yy[7]);
      }
      if (has2) {
#pragma unroll
        for (int m = 0; m < 8; m++) P[row2 * kP8Stride + m] = yy[m];
        double* __restrict__ lrow = Lp + (size_t)row2 * p + K;
        lrow[0] = yy[0];
        if (nb > 1) lrow[1] = yy[1];
        if (nb > 2) lrow[2] = yy[2];
        if (nb > 3) lrow[3] = yy[3];
        if (nb > 4) lrow[4] = yy[4];
        if (nb > 5) lrow[5] = yy[5];
        if (nb > 6) lrow[6] = yy[6];
        if (nb > 7) lrow[7] = yy[7];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (STRIP && strip && !R5) {
      // F[r][c] -= sum_k L[r][k] L[c][k] for the strip rows r and the live columns c >= K + nb: the strip is tile row 4 of the
      // front; its five 16x16 tiles are loaded from the LDS triangle, updated with one MFMA per pivot block and written back
      // (only the entries that exist: c <= r < fa).  Tile columns left of the panel are finished and skipped.
      const int cmin = K + nb;
      const bool v0 = lq < nb, v1 = lq + 4 < nb;
      const double a4r = P[(kRegRows + l16) * kP8Stride + lq], a4s = P[(kRegRows + l16) * kP8Stride + 4 + lq];
      const double a40 = v0 ? -a4r : 0.0, a41 = v1 ? -a4s : 0.0;
#pragma unroll 1                                    // one tile at a time: eight registers next to the ten resident tiles
      for (int tj = cmin >> 4; tj < 5; tj++) {
        const double br = P[(16 * tj + l16) * kP8Stride + lq], bs = P[(16 * tj + l16) * kP8Stride + 4 + lq];
        const double b0 = v0 ? br : 0.0, b1 = v1 ? bs : 0.0;
        const int col = 16 * tj + l16;
        double4_t t;
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = kRegRows + lq + 4 * q;
          ok[q] = row < fa && col <= row && col >= cmin;
          const double v = F[ok[q] ? tri(row) + col : 0];
          t[q] = ok[q] ? v : 0.0;
        }
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(a40, b0, t, 0, 0, 0);
        if (two) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a41, b1, t, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = kRegRows + lq + 4 * q;
          if (ok[q]) F[tri(row) + col] = t[q];
        }
      }
    }
    const long long tk1 = (TR && d.trace) ? clock64() : 0;
    switch (tjK) {
      case 0: reg_trailing8<0, NT>(c, P, nb, lane, (K + W) >> 4, c0 + nb < 16); break;
      case 1: reg_trailing8<1, NT>(c, P, nb, lane, (K + W) >> 4, c0 + nb < 16); break;
      case 2: if (NT > 2) reg_trailing8<(NT > 2 ? 2 : 1), NT>(c, P, nb, lane, (K + W) >> 4, c0 + nb < 16); break;
      default: if (NT > 3) reg_trailing8<(NT > 3 ? 3 : NT - 1), NT>(c, P, nb, lane, (K + W) >> 4, c0 + nb < 16); break;
    }
    __builtin_amdgcn_wave_barrier();
    if (TR && d.trace) { const long long tk2 = clock64(); cyc_panel += tk1 - tk0; cyc_trail += tk2 - tk1; }
  };
  if constexpr (NT == 5) {
    // (hipcc 7.2 miscompiled this loop with fifteen accumulator tiles once it peeled the first panel: rows 8 and up of every later
    // panel came out wrong, deterministically -- tests/test_gpu_fronts.py holds the case; without peeling the code is correct)
#pragma clang loop unroll(disable)
    for (int K = 0; K < p; K += W) panel_step(K);
  } else {
    for (int K = 0; K < p; K += W) panel_step(K);
  }
  if (bad && lane == 0) raise_status(&d.result_dev[2], 1.0);             // not positive definite
  if (TR) PPS_TR(4);
  if (TR && d.trace && lane == 0) { d.trace[(size_t)s * 8 + 6] = cyc_panel; d.trace[(size_t)s * 8 + 7] = cyc_trail; }
  // ---- update matrix: live part of the tiles -> packed global ----
  // Every store is issued by all lanes: an entry that does not exist (row >= fa, col > row, col < p) goes to the last double of
  // the front's (b+1) x (b+1) slab, which the packed triangle never reaches -- no exec-masked block per store, the sixteen row
  // bases are computed once, and whole tiles left of the pivots or below the front are skipped by wave-uniform branches.
  double* __restrict__ Us = d.U + (((long long)__builtin_amdgcn_readlane(rec, 12) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 11));
  const unsigned trash_u = (unsigned)(__mul24(b + 1, b + 1) - 1);
#pragma unroll
  for (int ti = 0; ti < NT; ti++) {
    if (16 * ti >= mr) continue;                             // (wave-uniform)
    int rbase[4]; bool rok[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int row = 16 * ti + lq + 4 * q; rok[q] = row < mr; rbase[q] = tri24(row - p) - p; }
#pragma unroll
    for (int tj = 0; tj <= ti; tj++) {
      if (16 * tj + 15 < p) continue;                        // (wave-uniform)
      const int col = 16 * tj + l16;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = 16 * ti + lq + 4 * q;
        const bool ok = rok[q] && col <= row && col >= p;
        Us[ok ? (unsigned)(rbase[q] + col) : trash_u] = c[tile_id(ti, tj)][q];
      }
    }
  }
  if (STRIP && strip && !R5) {
    for (int q = kRegRows; q < fa; q++)
      for (int col = p + lane; col <= q; col += 64) Us[tri(q - p) + col - p] = F[tri(q) + col];
  }
  if (!STRIP) Us[(lane >= p && lane <= f) ? (unsigned)(tri24(b) + lane - p) : trash_u] = lane < f ? y : 0.0;      // the rhs row of the update matrix
  if (TR) PPS_TR(5);
}

}  // namespace pps
