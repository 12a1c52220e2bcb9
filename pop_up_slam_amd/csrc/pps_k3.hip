// pps_k3.hip -- K3: multifrontal Cholesky (cholmod_factorize / cholmod_solve, isamlib/Cholesky.cpp:100-128).
//   k_band_factor / k_band_solve   one wavefront per front, a workgroup walks a sub-tree of a band of tree levels; fronts
//                                  <= 64 rows live in registers as 16x16 fp64 MFMA tiles (pps_regtile.h), 65 .. 80 rows keep a
//                                  strip in LDS, up to 128 rows in LDS tiles
//   k_front_factor / k_front_solve level-per-launch fallback (one workgroup per front) when neither the band kernels nor the
//                                  dense-front kernels (pps_dense.hip) apply
//   k_expand_ea / k_expand_el      index lists of a topology expanded in HBM
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include <hip/hip_ext.h>

#include "pps_kcommon.h"
#include "pps_regtile.h"
#include "pps_front_reg.h"

namespace pps {


// ------------------------------------------------------------------------------------------
// K3: multifrontal partial Cholesky.  One 256-thread workgroup per front; the (f+1)x(f+1) front
// (last row = right-hand side) lives in LDS (row-major, odd leading dimension), or in a global
// workspace when it does not fit.  Steps: gather original H blocks (damped diagonal,
// Cholesky.cpp:94-97) -> extend-add children update matrices -> right-looking elimination of the
// p pivot columns -> store factor panel and update matrix.
// ------------------------------------------------------------------------------------------
constexpr int kLdsLimitBytes = 160 * 1024 - 1024;

int lds_front_limit() {
  int fa = 1;
  while ((size_t)(fa + 1) * ((fa + 1) | 1) * 8 <= (size_t)kLdsLimitBytes) fa++;
  return fa - 1;   // largest f with (f+1) rows
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_front_factor(DevGraph d, int level_begin, double lambda) {
  extern __shared__ double lds[];
  const int s = d.level_fronts[level_begin + blockIdx.x];
  const int p = d.f_p[s], b = d.f_b[s];
  const int f = p + b, fa = f + 1, ld = fa | 1;
  double* F = USE_LDS ? lds : d.gwork + (size_t)blockIdx.x * d.gwork_stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < fa * ld; i += nt) F[i] = 0.0;
  __syncthreads();
  // ---- original entries: one wave per block, lane per entry ----
  {
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int a0 = d.f_asm_off[s], a1 = d.f_asm_off[s + 1];
    for (int a = a0 + wave; a < a1; a += nw) {
      const int blk = d.asm_blk[a], lrow = d.asm_lrow[a], lcol = d.asm_lcol[a];
      const int rows = d.blk_rows[blk], cols = d.blk_cols[blk], size = d.blk_size[blk];
      const double* __restrict__ h = d.H + d.blk_hoff[blk];
      const int rc = rows * cols;
      const bool diag = size > rc;
      if (lane < size) {
        double v = h[lane];   // k_hreduce has folded multi-segment blocks into their first slot
        if (lane < rc) {
          const int i = lane / cols, j = lane % cols;
          if (!diag || i >= j) {
            if (diag && i == j) v *= (1.0 + lambda);
            F[(lrow + i) * ld + lcol + j] += v;
          }
        } else {
          F[f * ld + lcol + (lane - rc)] += v;
        }
      }
    }
  }
  __syncthreads();
  // ---- extend-add of the children's update matrices ----
  for (int ci = d.f_child_off[s]; ci < d.f_child_off[s + 1]; ci++) {
    const int c = d.child[ci];
    const int bc1 = d.f_b[c] + 1;
    const double* __restrict__ Uc = d.U + d.f_Uoff[c];
    const int* __restrict__ cm = d.cmap + d.f_cmap_off[c];
    for (int idx = tid; idx < bc1 * bc1; idx += nt) {
      const int i = idx / bc1, j = idx - i * bc1;
      if (j <= i) F[cm[i] * ld + cm[j]] += Uc[idx];
    }
    __syncthreads();
  }
  // ---- eliminate the p pivot columns (right-looking) ----
  const int tx = tid & 15, ty = tid >> 4;
  for (int k = 0; k < p; k++) {
    const double dkk = F[k * ld + k];
    double dinv;
    if (!(dkk > 0.0)) {
      if (tid == 0) raise_status(&d.result_dev[2], 1.0);   // not positive definite
      dinv = 0.0;
    } else {
      dinv = 1.0 / sqrt(dkk);
    }
    for (int i = k + 1 + tid; i < fa; i += nt) F[i * ld + k] *= dinv;
    __syncthreads();
    for (int i = k + 1 + ty; i < fa; i += 16) {
      const double lik = F[i * ld + k];
      for (int j = k + 1 + tx; j <= i; j += 16) F[i * ld + j] -= lik * F[j * ld + k];
    }
    __syncthreads();   // column k+1 (diagonal included) is final before the next iteration reads it
  }
  // ---- store the factor panel ((f+1) x p, row-major) and the update matrix ((b+1) x (b+1)) ----
  double* __restrict__ Lp = d.L + d.f_Loff[s];
  for (int idx = tid; idx < fa * p; idx += nt) {
    const int i = idx / p, j = idx - i * p;
    double v = 0.0;
    if (i == j) { const double x = F[j * ld + j]; v = x > 0.0 ? sqrt(x) : 1.0; }
    else if (i > j) v = F[i * ld + j];
    Lp[idx] = v;
  }
  double* __restrict__ Us = d.U + d.f_Uoff[s];
  const int b1 = b + 1;
  for (int idx = tid; idx < b1 * b1; idx += nt) {
    const int i = idx / b1, j = idx - i * b1;
    Us[idx] = (j <= i) ? F[(p + i) * ld + p + j] : 0.0;
  }
}

static std::atomic<bool> g_attr_set[64];   // per device ordinal (idempotent set-up: a race only repeats it)

hipError_t launch_factor_level(const DevGraph& d, int level_begin, int level_count, int level_max_front, double lambda,
                               hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  const int fa = level_max_front + 1;
  const size_t bytes = (size_t)fa * (fa | 1) * 8;
  if (bytes <= (size_t)kLdsLimitBytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!g_attr_set[dev & 63]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_factor<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
      if (e != hipSuccess) return e;
      g_attr_set[dev & 63] = true;
    }
    PPS_LAUNCH(k_front_factor<true>, dim3(level_count), dim3(256), bytes, st, d, level_begin, lambda);
  } else {
    PPS_LAUNCH(k_front_factor<false>, dim3(level_count), dim3(256), 0, st, d, level_begin, lambda);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K3, wave-per-front form.  The tree levels are cut into bands; inside a band every connected
// sub-tree ("group") is walked by one workgroup: wave w takes fronts w, w+nw, ... of the current
// local level, a workgroup barrier separates the levels, update matrices travel through global
// memory (same CU, workgroup-scope visibility).  A front is a packed lower triangle in LDS
// (index(i,j) = i(i+1)/2 + j, last row = right-hand side); lane i owns row i (and i+64), so the
// elimination needs no barrier at all: column k is scaled, written back and re-read as LDS
// broadcasts by the same wave, in program order.
// ------------------------------------------------------------------------------------------
constexpr int kBandMaxRows = 128;   // rows per front including the rhs row

// One workgroup per front: row i of its (b+1)-row packed update matrix goes to row cmap[i] of the parent.
// The launch also clears what an upload needs cleared -- the zeroed block behind delta (tickets, result records, partial sums) and
// blk_dst (0xff: "no element") for k_expand_el, which follows on the same stream -- instead of one fill kernel each.
__global__ __launch_bounds__(64) void k_expand_ea(DevGraph d, double* __restrict__ zero, int n_zero, int* __restrict__ ones, int n_ones) {
  for (int i = blockIdx.x * 64 + threadIdx.x; i < n_zero; i += gridDim.x * 64) zero[i] = 0.0;
  for (int i = blockIdx.x * 64 + threadIdx.x; i < n_ones; i += gridDim.x * 64) ones[i] = -1;
  const int s = blockIdx.x;
  const int* __restrict__ m = d.cmap + d.f_cmap_off[s];
  const int len = d.f_cmap_off[s + 1] - d.f_cmap_off[s];
  int* __restrict__ out = d.ea_tgt + d.f_ea_off[s];
  const int n = len * (len + 1) / 2;
  int i = 0;                                  // row of entry e: tri(i) <= e < tri(i+1)
  for (int e = threadIdx.x; e < n; e += 64) {
    while ((i + 1) * (i + 2) / 2 <= e) i++;
    const int j = e - i * (i + 1) / 2;
    out[e] = m[i] * (m[i] + 1) / 2 + m[j];
  }
}

// One thread per assembled H block: its elements in the order the front gathers them (lower triangle of a diagonal
// block, a full off-diagonal block, then the gradient entries of a diagonal block).
__global__ __launch_bounds__(256) void k_expand_el(DevGraph d, int n_asm) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_asm) return;
  const int blk = d.asm_blk[a], rows = d.blk_rows[blk], cols = d.blk_cols[blk];
  const int lrow = d.asm_lrow[a], lcol = d.asm_lcol[a];
  const bool diag = d.blk_size[blk] != rows * cols;
  int* __restrict__ dst = d.blk_dst + d.blk_doff[blk];
  int e = d.asm_el0[a];
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) {
      if (diag && j > i) continue;
      dst[i * cols + j] = e;
      d.el_tgt[e++] = (((lrow + i) * (lrow + i + 1)) / 2 + lcol + j) | ((diag && i == j) ? (1 << 30) : 0);
    }
  if (diag) {
    const int fsz = d.asm_fsz[a];
    for (int i = 0; i < rows; i++) { dst[rows * cols + i] = e; d.el_tgt[e++] = (fsz * (fsz + 1)) / 2 + lcol + i; }
  }
}

// Both expansions and the upload's zero fill in ONE launch (frame loops: a launch less per frame).  Workgroups [0, n_fronts) do what
// k_expand_ea does, the others what k_expand_el does, 64 assembled blocks each; an assembled block defines EVERY entry of its blk_dst
// range itself (-1 for the upper triangle of a diagonal block, which no front gathers), so no fill has to run before it -- valid when
// every H block is assembled by exactly one front (the caller checks).
__global__ __launch_bounds__(64) void k_expand_lists(DevGraph d, double* __restrict__ zero, int n_zero, int n_fronts, int n_asm) {
  for (int i = blockIdx.x * 64 + threadIdx.x; i < n_zero; i += gridDim.x * 64) zero[i] = 0.0;
  if ((int)blockIdx.x < n_fronts) {
    const int s = blockIdx.x;
    const int* __restrict__ m = d.cmap + d.f_cmap_off[s];
    const int len = d.f_cmap_off[s + 1] - d.f_cmap_off[s];
    int* __restrict__ out = d.ea_tgt + d.f_ea_off[s];
    const int n = len * (len + 1) / 2;
    int i = 0;                                  // row of entry e: tri(i) <= e < tri(i+1)
    for (int e = threadIdx.x; e < n; e += 64) {
      while ((i + 1) * (i + 2) / 2 <= e) i++;
      const int j = e - i * (i + 1) / 2;
      out[e] = m[i] * (m[i] + 1) / 2 + m[j];
    }
    return;
  }
  const int a = ((int)blockIdx.x - n_fronts) * 64 + threadIdx.x;
  if (a >= n_asm) return;
  const int blk = d.asm_blk[a], rows = d.blk_rows[blk], cols = d.blk_cols[blk];
  const int lrow = d.asm_lrow[a], lcol = d.asm_lcol[a];
  const bool diag = d.blk_size[blk] != rows * cols;
  int* __restrict__ dst = d.blk_dst + d.blk_doff[blk];
  int e = d.asm_el0[a];
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) {
      if (diag && j > i) { dst[i * cols + j] = -1; continue; }
      dst[i * cols + j] = e;
      d.el_tgt[e++] = (((lrow + i) * (lrow + i + 1)) / 2 + lcol + j) | ((diag && i == j) ? (1 << 30) : 0);
    }
  if (diag) {
    const int fsz = d.asm_fsz[a];
    for (int i = 0; i < rows; i++) { dst[rows * cols + i] = e; d.el_tgt[e++] = (fsz * (fsz + 1)) / 2 + lcol + i; }
  }
}
hipError_t launch_expand_lists(const DevGraph& d, int n_fronts, int n_asm, double* zero, size_t n_zero, hipStream_t st) {
  PPS_LAUNCH(k_expand_lists, dim3(n_fronts + (n_asm + 63) / 64), dim3(64), 0, st, d, zero, (int)n_zero, n_fronts, n_asm);
  return hipGetLastError();
}

hipError_t launch_expand_el(const DevGraph& d, int n_asm, hipStream_t st) {
  if (n_asm <= 0) return hipSuccess;
  PPS_LAUNCH(k_expand_el, dim3((n_asm + 255) / 256), dim3(256), 0, st, d, n_asm);
  return hipGetLastError();
}

hipError_t launch_expand_ea(const DevGraph& d, int n_fronts, double* zero, size_t n_zero, int* ones, size_t n_ones, hipStream_t st) {
  if (n_fronts <= 0) return hipSuccess;
  PPS_LAUNCH(k_expand_ea, dim3(n_fronts), dim3(64), 0, st, d, zero, (int)n_zero, ones, (int)n_ones);
  return hipGetLastError();
}

int band_front_limit() { return kBandMaxRows - 1; }
bool band_level_solve_direct_ok(int p, int b);
int band_reg_rows() { return kRegRows; }
int band_max_rows() { return kBandMaxRows; }
// LDS of one wave in the factor kernels.  Register-only kernels (every front of the stage <= 64 rows; the level kernels): the packed
// triangle, whose first 64 x kP8Stride doubles double as the panel buffer once the tiles are loaded (the triangle is dead from then
// on), and one spare double at the end (where masked-off items of a scatter-add land).  The other kernels (LDS strip, fifth tile
// row, LDS-tile path): triangle | spare | panel buffer of 80 rows.
size_t band_lds_bytes(int max_front, bool reg_only_kernel) {
  const size_t fa = (size_t)max_front + 1, ntri = fa * (fa + 1) / 2;
  const size_t n = reg_only_kernel ? std::max<size_t>(ntri, (size_t)kRegRows * kP8Stride) + 1 : ntri + 1 + (size_t)kRegRowsMax * kP8Stride;
  return ((n + 1) & ~size_t(1)) * sizeof(double);      // (an even number of doubles: every wave's triangle starts 16-byte aligned)
}
// One row of 16x16 tiles (I, J = o, o+16, ..., I) of the trailing lower triangle gets its rank-nb
// update C -= P_I P_J^T: all LDS reads are issued unconditionally from clamped (always valid)
// addresses and masked by selects afterwards, so the NT tiles' loads overlap; then NT back-to-back
// v_mfma_f64_16x16x4_f64; then the masked stores.
template <int NT>
__device__ __forceinline__ void trailing_tile_row(double* __restrict__ F, int fa, int o, int I, int K, int nb, int lane) {
  const int l16 = lane & 15, lq = lane >> 4;
  const int kk = K + lq;                                   // this lane's k index of the MFMA operands
  const bool kvalid = lq < nb;
  const int ar = I + l16;
  const bool aok = kvalid && ar < fa;
  const double araw = F[aok ? tri(ar) + kk : 0];
  const double av = aok ? -araw : 0.0;
  int crt[4]; bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; r++) { const int cr = I + lq + 4 * r; rok[r] = cr < fa; crt[r] = rok[r] ? tri(cr) : 0; }   // D layout: row (l/16)+4r, col l%16
  double bv[NT];
  double4_t c[NT];
  bool cok[NT][4];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int br = o + 16 * t + l16;
    const bool bok = kvalid && br < fa;
    const double braw = F[bok ? tri(br) + kk : 0];
    bv[t] = bok ? braw : 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      cok[t][r] = rok[r] && br <= I + lq + 4 * r;
      const double craw = F[cok[t][r] ? crt[r] + br : 0];
      c[t][r] = cok[t][r] ? craw : 0.0;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; t++) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[t], c[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int cc = o + 16 * t + l16;
#pragma unroll
    for (int r = 0; r < 4; r++)
      if (cok[t][r]) F[crt[r] + cc] = c[t][r];
  }
}

__device__ __forceinline__ void wave_front_factor(const DevGraph& d, int s_in, double lambda, double* __restrict__ F) {
  const int lane = threadIdx.x & 63;
  const int s = uni(s_in);
  const int p = uni(d.f_p[s]), b = uni(d.f_b[s]);
  const int f = p + b, fa = f + 1;
  const int ntri = tri(fa);
  PPS_TR(0);
  for (int i = lane; i < ntri; i += 64) F[i] = 0.0;
  __builtin_amdgcn_wave_barrier();
  PPS_TR(1);
  // ---- original entries: Hf is in gather order, so value and target index are two independent
  // coalesced streams; 8 elements per lane are fetched before the first LDS update ----
  {
    const int e0 = uni(d.f_el_off[s]), e1 = uni(d.f_el_off[s + 1]);
    const double damp = 1.0 + lambda;
    const int* __restrict__ tgp = d.el_tgt;
    const double* __restrict__ hf = d.Hf;
    for (int e = e0 + lane; e < e1; e += 64 * 8) {
      int tg[8]; double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < e1 ? tgp[x] : -1; v[u] = x < e1 ? hf[x] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tg[u] >= 0) F[tg[u] & 0x3fffffff] += (tg[u] & (1 << 30)) ? v[u] * damp : v[u];   // Cholesky.cpp:94-97
    }
  }
  __builtin_amdgcn_wave_barrier();
  PPS_TR(2);
  // ---- extend-add of the children's packed update matrices (same batching) ----
  const int ci0 = uni(d.f_child_off[s]), ci1 = uni(d.f_child_off[s + 1]);
  for (int ci = ci0; ci < ci1; ci++) {
    const int c = uni(d.child[ci]);
    const int bc1 = uni(d.f_b[c]) + 1;
    const int n = tri(bc1);
    const double* __restrict__ Uc = d.U + uni64(d.f_Uoff[c]);
    const int* __restrict__ tgc = d.ea_tgt + uni64(d.f_ea_off[c]);
    for (int e = lane; e < n; e += 64 * 8) {
      int tg[8]; double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int x = e + 64 * u; tg[u] = x < n ? tgc[x] : -1; v[u] = x < n ? Uc[x] : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (tg[u] >= 0) F[tg[u]] += v[u];
    }
    __builtin_amdgcn_wave_barrier();
  }
  PPS_TR(3);
  // ---- eliminate the p pivot columns, four at a time ----
  // Panel: lane i holds rows i and i+64 of the 4 panel columns in registers; the 4x4 diagonal block is
  // broadcast with v_readlane, so the panel factorisation never waits on LDS.  Trailing update: the
  // rank-4 update C -= P_I * P_J^T of every 16x16 tile of the remaining lower triangle is ONE
  // v_mfma_f64_16x16x4_f64 (A = -P_I, B = P_J^T), operands gathered from the packed triangle in LDS.
  const int r0 = lane, r1 = lane + 64;
  const int t0 = tri(r0), t1 = tri(r1);
  long long cyc_panel = 0, cyc_trail = 0;
  for (int K = 0; K < p; K += 4) {
    const long long tk0 = d.trace ? clock64() : 0;
    double a0[4], a1[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      const bool v0 = c < p && r0 >= c && r0 < fa, v1 = c < p && r1 >= c && r1 < fa;
      const double x0 = F[v0 ? t0 + c : 0], x1 = F[v1 ? t1 + c : 0];
      a0[m] = v0 ? x0 : 0.0;
      a1[m] = v1 ? x1 : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      if (c < p) {                                         // wave-uniform
        const double dmm = (c < 64) ? readlane_d(a0[m], c) : readlane_d(a1[m], c - 64);
        double dinv = 0.0;
        if (dmm > 0.0) dinv = rsqrt_nr(dmm);
        else if (lane == 0) raise_status(&d.result_dev[2], 1.0);         // not positive definite
        if (r0 >= c) a0[m] *= dinv;                        // the diagonal becomes sqrt(dmm)
        if (r1 >= c) a1[m] *= dinv;
#pragma unroll
        for (int n = m + 1; n < 4; n++) {
          const int cn = K + n;
          if (cn < p) {
            const double lnm = (cn < 64) ? readlane_d(a0[m], cn) : readlane_d(a1[m], cn - 64);
            if (r0 >= cn) a0[n] -= a0[m] * lnm;
            if (r1 >= cn) a1[n] -= a1[m] * lnm;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int c = K + m;
      if (c < p) {
        if (r0 >= c && r0 < fa) F[t0 + c] = a0[m];
        if (r1 >= c && r1 < fa) F[t1 + c] = a1[m];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const long long tk1 = d.trace ? clock64() : 0;
    const int o = K + 4 < p ? K + 4 : p;                   // first trailing column
    const int nb = o - K;                                  // panel width (1..4)
    for (int I = o; I < fa; I += 16) {
      const int nt = ((I - o) >> 4) + 1;                   // tiles (I, J <= I) of this tile row, wave-uniform
      switch (nt) {
        case 1: trailing_tile_row<1>(F, fa, o, I, K, nb, lane); break;
        case 2: trailing_tile_row<2>(F, fa, o, I, K, nb, lane); break;
        case 3: trailing_tile_row<3>(F, fa, o, I, K, nb, lane); break;
        case 4: trailing_tile_row<4>(F, fa, o, I, K, nb, lane); break;
        case 5: trailing_tile_row<5>(F, fa, o, I, K, nb, lane); break;
        case 6: trailing_tile_row<6>(F, fa, o, I, K, nb, lane); break;
        case 7: trailing_tile_row<7>(F, fa, o, I, K, nb, lane); break;
        default: trailing_tile_row<8>(F, fa, o, I, K, nb, lane); break;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (d.trace) { const long long tk2 = clock64(); cyc_panel += tk1 - tk0; cyc_trail += tk2 - tk1; }
  }
  PPS_TR(4);
  if (d.trace && lane == 0) { d.trace[(size_t)s * 8 + 6] = cyc_panel; d.trace[(size_t)s * 8 + 7] = cyc_trail; }
  // ---- factor panel (f+1) x p row-major (diagonal = sqrt) and packed update matrix ----
  double* __restrict__ Lp = d.L + uni64(d.f_Loff[s]);
  if (lane < p) {                                          // p <= 64: lane = column
    for (int i0 = 0; i0 < fa; i0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; v[u] = F[(i < fa && lane <= i) ? tri(i) + lane : 0]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; if (i < fa && lane <= i) Lp[(size_t)i * p + lane] = v[u]; }
    }
  }
  double* __restrict__ Us = d.U + uni64(d.f_Uoff[s]);
  for (int i0 = p; i0 < fa; i0 += 8) {
    for (int j = p + lane; j < fa; j += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; v[u] = F[(i < fa && j <= i) ? tri(i) + j : 0]; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u; if (i < fa && j <= i) Us[tri(i - p) + j - p] = v[u]; }
    }
  }
  PPS_TR(5);
}

// ------------------------------------------------------------------------------------------
// Register-resident variant for fronts of <= 64 rows (all of C2, most of C3).  After assembly in LDS
// the whole front lives in the lanes' registers as the ten 16x16 tiles of the lower triangle, each in
// the MFMA accumulator layout (lane l, reg r <-> row (l/16)+4r, col l%16), so a rank-4 update of a
// tile is a single register-to-register v_mfma_f64_16x16x4_f64.  Per 4-column block only the panel
// moves through LDS: tile columns K..K+3 -> P[row][4] -> one lane per row solves its row against the
// 4x4 diagonal block (broadcast with v_readlane, Cholesky-factored redundantly by every lane) ->
// P feeds the MFMA operands.  Entries left of / above the current block are dead and may hold garbage.
// ------------------------------------------------------------------------------------------
// ---- assembly of a front's packed triangle in LDS, in pieces (one wave) ----
// Eight scatter-add items per lane in flight: (target index, value) pairs of the front-ordered H, or of a child's packed update
// matrix.  Issued as 16 independent coalesced loads, applied as LDS read-modify-writes (targets are unique inside one source).
// scatter-add items per lane in flight in the extend-add.  Measured on C2 (us per LM iteration): 8 -> 78.9, 10 -> 77.6,
// 12 -> 78.5, 16 -> 92.4 -- more loads in flight than about two dozen per lane cost more than the round trips they save
constexpr int kEaDepth = 10;
template <int N> struct ElBatch { int tg[N]; double v[N]; };
// loads are issued unconditionally from clamped (always valid) addresses and masked by selects afterwards: predicated loads
// become one exec-masked basic block each and serialise
template <int N>
__device__ __forceinline__ void el_issue(const int* __restrict__ tgp, const double* __restrict__ val, int e, int e1, int lane, ElBatch<N>& q) {
  const int last = e1 > 0 ? e1 - 1 : 0;
#pragma unroll
  for (int u = 0; u < N; u++) {
    const int x = e + lane + 64 * u, xc = x < e1 ? x : last;
    const int t = tgp[xc]; const double v = val[xc];
    q.tg[u] = x < e1 ? t : -1; q.v[u] = x < e1 ? v : 0.0;
  }
}
// The N read-modify-writes of a batch as N reads, N adds, N writes -- one LDS round trip instead of N dependent ones.  Valid
// because the targets of one source (the original entries of a front; one child's update matrix) are pairwise distinct;
// items past the end go to the spare double `tr` with a zero increment, so that nothing is predicated.
template <bool ORIG, int N>     // ORIG: entries of H (bit 30 of the target = diagonal element, damped: Cholesky.cpp:94-97)
__device__ __forceinline__ void el_apply(const ElBatch<N>& q, double damp, double* __restrict__ F, int tr) {
  if constexpr (ORIG) {
    // the original entries are the FIRST thing a cleared triangle receives and their targets are pairwise distinct, so old + inc is
    // 0 + inc = inc (H entries are sums that start at +0: never -0): a plain store -- no LDS read, no add, one LDS round trip less per batch.
#pragma unroll
    for (int u = 0; u < N; u++) {
      const int t = q.tg[u] >= 0 ? (q.tg[u] & 0x3fffffff) : tr;
      F[t] = (q.tg[u] & (1 << 30)) && q.tg[u] >= 0 ? q.v[u] * damp : q.v[u];        // (tg = -1: v = 0 into the spare double)
    }
    return;
  }
  int t[N]; double old[N];
#pragma unroll
  for (int u = 0; u < N; u++) { t[u] = q.tg[u] >= 0 ? q.tg[u] : tr; old[u] = F[t[u]]; }
#pragma unroll
  for (int u = 0; u < N; u++) F[t[u]] = old[u] + q.v[u];      // (tg = -1: v = 0)
}
// what a front needs before its children are complete: first batch of its original entries and its child records (requested),
// the cleared triangle, the original entries added.  issue -> [anything] -> finish.  NPRE items per lane are requested ahead:
// 8 when nothing else is live, 4 (256 entries: every separator front of a corridor graph) next to the register tiles of the
// front being eliminated.
template <int NPRE> struct FrontPre { ElBatch<NPRE> q; int crv; };
template <int NPRE>
__device__ __forceinline__ void front_pre_issue(const DevGraph& d, int rec, int lane, FrontPre<NPRE>& o) {
  const int e0 = __builtin_amdgcn_readlane(rec, 3), e1 = __builtin_amdgcn_readlane(rec, 4);
  const int cr0 = __builtin_amdgcn_readlane(rec, 5), nch = __builtin_amdgcn_readlane(rec, 6);
  el_issue(d.el_tgt, d.Hf, e0, e1, lane, o.q);
  // records of up to 8 children in one coalesced load
  o.crv = (lane < 8 * nch) ? d.crec[(size_t)cr0 * 8 + lane] : 0;
}
__device__ __forceinline__ void front_clear(int rec, int lane, double* __restrict__ F) {
  const int ntri = tri(__builtin_amdgcn_readlane(rec, 1) + __builtin_amdgcn_readlane(rec, 2) + 1);
  // 16 bytes per lane and store (the triangle starts 16-byte aligned; an odd tail is one more 8-byte store)
  double2* __restrict__ F2 = reinterpret_cast<double2*>(F);
  const int n2 = ntri >> 1;
#pragma unroll 4
  for (int i = lane; i < n2; i += 64) F2[i] = make_double2(0.0, 0.0);
  if (lane == 0) F[ntri - 1] = 0.0;
  __builtin_amdgcn_wave_barrier();
}
template <int NPRE>
__device__ __forceinline__ void front_pre_finish(const DevGraph& d, int rec, int lane, const FrontPre<NPRE>& o, double damp, double* __restrict__ F, int tr) {
  const int e0 = __builtin_amdgcn_readlane(rec, 3), e1 = __builtin_amdgcn_readlane(rec, 4);
  el_apply<true>(o.q, damp, F, tr);
  for (int e = e0 + 64 * NPRE; e < e1; e += 64 * 8) { ElBatch<8> q; el_issue(d.el_tgt, d.Hf, e, e1, lane, q); __builtin_amdgcn_wave_barrier(); el_apply<true>(q, damp, F, tr); }
  __builtin_amdgcn_wave_barrier();
}
// extend-add of the children's packed update matrices, child by child, one batch of 512 entries in flight (more loads in
// flight -- both children at once, two batches per child -- measured SLOWER on MI355X: DESIGN.md section 8)
// Hand-over between WORKGROUPS of one launch (round 6: the whole tree in one factor launch and one back-substitution launch, XG = true
// below).  A flag per front in global memory carries the number of the launch pair (the epoch: never reset).  Producer: the wave's stores,
// an agent-scope release (on this chip: the wave's stores acknowledged + the XCD's L2 written back), the flag.  Consumer: polls the flag,
// agent-scope acquire, then reads.  A workgroup only ever waits for workgroups with a LOWER block index -- children in the factor launch,
// parents in the back-substitution launch, whose groups are numbered the other way round -- and the hardware starts workgroups in index
// order: whatever a waiting workgroup waits for has been started and waits, if at all, for still lower indices.  No cycle, no dependence on
// how many workgroups are resident, or on what else shares the device.  The spin is bounded like flow_wait's.
struct XGroup { int* flag; int epoch; };          // flag: one int per front (of this damping value); epoch 0 / flag null: off
__device__ __forceinline__ void xg_post(const XGroup& x, int s) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(x.flag + s, x.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xg_wait(const DevGraph& d, const XGroup& x, int s) {
  int spin = 0;
  while (__hip_atomic_load(x.flag + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != x.epoch) {
    if (++spin >= (1 << 21)) { if ((threadIdx.x & 63) == 0) raise_status(&d.result_dev[2], kStatusInternal); break; }
    __builtin_amdgcn_s_sleep(2);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// XG: a child that another workgroup of this launch factors (child record slot 6) is waited for first (xg_wait)
template <bool XG = false>
__device__ __forceinline__ void front_extend_add(const DevGraph& d, int rec, int crv0, int lane, double* __restrict__ F, int tr, const XGroup* xg = nullptr) {
  const int cr0 = __builtin_amdgcn_readlane(rec, 5), nch = __builtin_amdgcn_readlane(rec, 6);
  for (int cb = 0; cb < nch; cb += 8) {
    const int crv = cb == 0 ? crv0 : ((lane < 8 * (nch - cb)) ? d.crec[(size_t)(cr0 + cb) * 8 + lane] : 0);      // (more than 8 children: rare)
    auto child = [&](int cj, int& n, const double* __restrict__& Uc, const int* __restrict__& tgc) {
      n = __builtin_amdgcn_readlane(crv, 8 * cj);
      const long long uo = ((long long)__builtin_amdgcn_readlane(crv, 8 * cj + 2) << 32) | (unsigned int)__builtin_amdgcn_readlane(crv, 8 * cj + 1);
      const long long eo = ((long long)__builtin_amdgcn_readlane(crv, 8 * cj + 4) << 32) | (unsigned int)__builtin_amdgcn_readlane(crv, 8 * cj + 3);
      Uc = d.U + uo; tgc = d.ea_tgt + eo;
    };
    int cj = 0;
    for (; cj < 8 && cb + cj < nch; cj++) {
      int n; const double* Uc; const int* tgc;
      child(cj, n, Uc, tgc);
      if (XG) { if (__builtin_amdgcn_readlane(crv, 8 * cj + 6)) xg_wait(d, *xg, __builtin_amdgcn_readlane(crv, 8 * cj + 5)); }
      // batches of kEaDepth x 64 = 640 entries: the update matrix of a separator front of a corridor tree (34 rows + rhs: 595
      // entries) is one memory round trip, not a full batch of 512 followed by a nearly empty one
      for (int e = 0; e < n; e += 64 * kEaDepth) { ElBatch<kEaDepth> q; el_issue(tgc, Uc, e, n, lane, q); __builtin_amdgcn_wave_barrier(); el_apply<false>(q, 0.0, F, tr); }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// One front from start to end.
// TR: in-kernel phase trace (PPS_TRACE=1) compiled in
// Fronts of 65 .. 80 rows (STRIP): rows 0 .. 63 live in the register tiles as usual; rows 64 .. fa-1 -- boundary rows, the pivots
// are among the first 48 -- stay where the assembly put them, in the packed LDS triangle, and are carried along panel by panel:
// triangular solve by lanes 0 .. 15, rank-4 update with lane = column.
// panel widths (pivot columns per LDS round trip) of the kernel families
constexpr int kPanelWBand = 4;    // register-only band kernels (C2: 73.1 us per LM iteration against 74.1 with 8 -- a lone wave per SIMD is bound by the
                                  // pivot chain; a whole tile column in registers, 16, measured 67.0 against 60.3 us: docs/HISTORY.md, round 5)
constexpr int kPanelWLevel = 8;   // level-per-launch kernels (several waves per SIMD)
constexpr int kPanelWWide = 8;    // the kernels that also hold fronts of 65 .. 80 rows (fifth tile row / LDS strip) and the general one
// First half of a register-resident front, the same for every tile count: the packed triangle assembled in LDS.
template <bool TR>
__device__ __forceinline__ void front_assemble(const DevGraph& d, int rec, double lambda, double* F, int tr) {
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readlane(rec, 0);
  (void)s;
  if (TR) PPS_TR(0);
  // the first gather batch and the child records are requested before the LDS triangle is cleared, so the clearing hides
  // under their latency
  FrontPre<8> pre;
  front_pre_issue(d, rec, lane, pre);
  front_clear(rec, lane, F);
  if (TR) PPS_TR(1);
  // tr: the spare double of the wave's LDS (band_lds_bytes), where the items of a scatter-add batch past its end land
  front_pre_finish(d, rec, lane, pre, 1.0 + lambda, F, tr);
  if (TR) PPS_TR(2);
  front_extend_add(d, rec, pre.crv, lane, F, tr);
  if (TR) PPS_TR(3);
}
// Second half: elimination in registers, factor panel and update matrix out.
template <int NT, bool TR, bool STRIP = false, int W = kPanelWBand>
__device__ __forceinline__ void front_eliminate_out(const DevGraph& d, int rec, double* F, double* P) {
  front_reg_eliminate<NT, TR, STRIP, W>(d, rec, F, P);
}
// One front from start to end (the level-per-launch kernels: one tile count per kernel)
template <int NT, bool TR, bool STRIP = false, int W = kPanelWBand>
__device__ __forceinline__ void wave_front_factor_reg(const DevGraph& d, int rec, double lambda, double* F, double* P, int tr) {   // (P may be F)
  front_assemble<TR>(d, rec, lambda, F, tr);
  front_eliminate_out<NT, TR, STRIP, W>(d, rec, F, P);
}

// lane K of every row of 16 lanes, broadcast to the row (DPP row_newbcast: two v_mov_b32_dpp, no SGPR hop, no LDS)
template <int K> __device__ __forceinline__ double row_bcast_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Hand-over of a front's local solution inside a band group (body_band_solve_flow): the producer's writes to X, then a release fence
// over the LDS, then the flag; the consumer polls the flag and passes an acquire fence before it reads X.  The fences are workgroup
// scope and LOCAL address space only -- a plain workgroup release would also wait for the acknowledgement of the wave's stores to delta, a
// memory round trip per level of the chain.
// A parent never waits for a child, so the wait cannot lock up; the spin is bounded all the same, and a wait that runs out raises
// kFlowTimeout in the graph's status word: the host then returns PPS_EHIP instead of numbers (read_status, pps_solve.cpp).
constexpr double kFlowTimeout = kStatusInternal; // result_dev[2]: 0 ok | 1 not positive definite | >= 64 a hand-over flag never arrived
__device__ __forceinline__ void flow_wait(const DevGraph& d, int* flow, int q) {
  int spin = 0;
  while (__hip_atomic_load(flow + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
    if (++spin >= (1 << 22)) { if ((threadIdx.x & 63) == 0) raise_status(&d.result_dev[2], kFlowTimeout); break; }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void flow_post(const DevGraph& d, int* flow, int slot, bool top) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  // (SW_DEBUG_DROP_FLAG, tests only: the top front of a group never raises its flag -- its children run into the time-out)
  if ((threadIdx.x & 63) == 0 && !((d.sw & SW_DEBUG_DROP_FLAG) && top))
    __hip_atomic_store(flow + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// The last sixteen pivots of a back-substitution (all of them for p <= 16), chain in registers: the scaling of the multipliers (does
// not depend on the right-hand side) and the chain itself.
__device__ __forceinline__ void solve_pivot_scale(double (&lk)[16], double dinv) {
    // pivots 15 .. 0 live in the first row of 16 lanes: t_k and 1 / L_kk reach the other lanes of the row as DPP row broadcasts --
    // a pivot is two v_mov_dpp + one v_fma_f64, nothing leaves the vector ALU
    // (lk is 0 on and above the diagonal since it was loaded: a select around the DPP move would be turned into a branch that
    // switches the source lane off)
#define PPS_SC(U) lk[U] *= row_bcast_d<U>(dinv);
    PPS_SC(1) PPS_SC(2) PPS_SC(3) PPS_SC(4) PPS_SC(5) PPS_SC(6) PPS_SC(7) PPS_SC(8) PPS_SC(9) PPS_SC(10) PPS_SC(11) PPS_SC(12) PPS_SC(13) PPS_SC(14) PPS_SC(15)
#undef PPS_SC
}
template <bool SCALED = false>
__device__ __forceinline__ void solve_pivot_chain(int p, double (&lk)[16], double dinv, double& tj) {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // dependent chains: a - b * c is one operation here
#endif
    if (!SCALED) solve_pivot_scale(lk, dinv);
#define PPS_BS(U) case U: tj -= lk[U] * row_bcast_d<U>(tj); [[fallthrough]];
    switch (p < 16 ? p - 1 : 15) {                            // (wave-uniform: the chain is entered at the last pivot)
      PPS_BS(15) PPS_BS(14) PPS_BS(13) PPS_BS(12) PPS_BS(11) PPS_BS(10) PPS_BS(9) PPS_BS(8)
      PPS_BS(7) PPS_BS(6) PPS_BS(5) PPS_BS(4) PPS_BS(3) PPS_BS(2)
      case 1: tj -= lk[1] * row_bcast_d<1>(tj); [[fallthrough]];
      default: break;
    }
#undef PPS_BS
    tj *= dinv;
}

// x_p = L_A^-T (y - L_B^T x_b) for one front, one wave (p <= 64).  The factor panel ((f+1) x p, contiguous) is
// copied to LDS in batches of 16 independent coalesced loads per lane -- two round trips for a C2 front instead of one
// per 8 rows -- and everything after that reads LDS; the back-substitution chain runs in registers (lane j holds
// t_j, x_k is broadcast with v_readlane).  scratch: xb[128] + the panel.
// GROUP: the front belongs to a band group (boundary values may come from the parent's local solution in LDS, the own solution
// is left there for the children); false: level-per-launch form, everything through delta
// TR (PPS_TRACE=2): phase stamps of the back-substitution in the trace slots of the front: 0 start | 1 panel in LDS | 2 boundary
// values in place | 3 y - L_B^T x_b | 4 back-substitution done | 5 end
// flow (band groups, body_band_solve_flow): every front of the group has been started at once; flow[q] != 0 <=> the local solution of the
// group's front q is in X.  The front does everything that does not need its parent's solution, waits for the parent's flag right where
// the boundary values are read, and raises its own flag behind its solution.
// XG (with flow): the launch holds every band group (k_band_solve_all) -- a group's top front waits for its parent's flag before it gathers its
// boundary values from delta, and every front with children raises its own flag behind its solution
template <bool GROUP = true, bool TR = false, bool XG = false>
__device__ __forceinline__ void wave_front_solve(const DevGraph& d, int rec, double* __restrict__ W, double* __restrict__ X, int slot, int* flow = nullptr, const XGroup* xg = nullptr) {
  const int lane = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readlane(rec, 0);
  (void)s;
  if (TR) PPS_TR(0);
  const int p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2), f = p + b;
  const double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  const int pslot = GROUP ? __builtin_amdgcn_readlane(rec, 14) : -1;
  // boundary values: from the parent's local solution vector in LDS (through cmap) when the parent was solved by this
  // workgroup, else gathered from delta.  The index load does not depend on the parent and is issued first.
  // (unpredicated loads at clamped positions -- a predicated load is a basic block of its own that waits for its data, i.e. one more
  // memory round trip per index register in front of the panel's; the root has no list: any readable address does)
  const int* __restrict__ ix = b == 0 ? d.frec : (pslot >= 0 ? d.cmap + __builtin_amdgcn_readlane(rec, 15) : d.bidx + __builtin_amdgcn_readlane(rec, 8));
  double* xb = W;
  double* PL = W + kBandMaxRows;
  const int ix0r = ix[lane < b ? lane : 0], ix1r = ix[lane + 64 < b ? lane + 64 : 0];
  const int pix = d.pidx[__builtin_amdgcn_readlane(rec, 7) + (lane < p ? lane : 0)];      // (where x_p goes: fetched with the rest)
  const int n = (f + 1) * p;
  double tj = 0.0, dinv = 0.0;
  double lk[16];
  if (GROUP && p <= 16) {                       // (wave-uniform) every separator front of a corridor tree; band form only:
    // the level-per-launch form of a large batch is issue-bound and 3 % slower with the extra address arithmetic (G = 128: 11.4 against 11.0 ms)
    // DIRECT form: no LDS copy of the panel.  Every lane requests exactly the entries it will multiply -- rows part, part + 4, ... of
    // L_B in column j (lane = 16 part + j), the sixteen rows of L_A^T in its column, the diagonal, the rhs row -- 30 independent loads
    // in one round trip, issued before the boundary values are waited for; the products then read registers.  The sums run in the
    // order of the LDS form below (same bits).
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)
#endif
    const int j = lane & 15, part = lane >> 4;
    const int jc = j < p ? j : 0, lc = lane < p ? lane : 0;
    const double* __restrict__ LB = Lp + p * p + jc;            // (b = 0: row 0 of "L_B" is the rhs row -- readable, multiplied by 0)
    const int blast = b > 0 ? b - 1 : 0;
    double lbv[12];
#pragma unroll
    for (int u = 0; u < 12; u++) { const int i = part + 4 * u; lbv[u] = LB[(i < b ? i : blast) * p]; }
#pragma unroll
    for (int u = 0; u < 16; u++) { const double l = Lp[(u < p ? u : p - 1) * p + lc]; lk[u] = lane < u ? l : 0.0; }
    const double dg = Lp[lc * p + lc], yr = Lp[f * p + jc];
    const int ix0 = lane < b ? ix0r : 0, ix1 = lane + 64 < b ? ix1r : 0;
    double g0 = 0.0, g1 = 0.0;
    const int xpar = XG ? __builtin_amdgcn_readlane(rec, 13) : -1;         // the parent's front when another workgroup of this launch solves it
    if (pslot < 0 && xpar < 0) { g0 = d.delta[ix0]; g1 = d.delta[ix1]; }
    if (flow) { dinv = 1.0 / dg; solve_pivot_scale(lk, dinv); }       // (all that can be done without the parent, before waiting for it)
    if (pslot >= 0) {
      if (flow) flow_wait(d, flow, pslot);
      const double* __restrict__ Xp = X + (size_t)pslot * kBandMaxRows; g0 = Xp[ix0]; g1 = Xp[ix1];
    } else if (XG && xpar >= 0) {
      xg_wait(d, *xg, xpar);
      g0 = d.delta[ix0]; g1 = d.delta[ix1];
    }
    if (TR) PPS_TR(1);
    if (lane < b) xb[lane] = g0;
    if (lane + 64 < b) xb[lane + 64] = g1;
    __builtin_amdgcn_wave_barrier();
    if (TR) PPS_TR(2);
    if (!flow) dinv = 1.0 / dg;
    double a[4] = {part == 0 ? yr : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 12; u++) { const int i = part + 4 * u; const double x = i < b ? xb[i] : 0.0; a[u & 3] -= lbv[u] * x; }
    for (int i0 = 48; i0 < b; i0 += 48) {                       // (boundaries of more than 48 rows: one more round trip per 48)
#pragma unroll
      for (int u = 0; u < 12; u++) { const int i = i0 + part + 4 * u; lbv[u] = LB[(i < b ? i : blast) * p]; }
#pragma unroll
      for (int u = 0; u < 12; u++) { const int i = i0 + part + 4 * u; const double x = i < b ? xb[i] : 0.0; a[u & 3] -= lbv[u] * x; }
    }
    tj = (a[0] + a[1]) + (a[2] + a[3]);
    tj += __shfl_xor(tj, 32);
    tj += __shfl_xor(tj, 16);
    tj = lane < p ? tj : 0.0;
    if (TR) PPS_TR(3);
    if (flow) solve_pivot_chain<true>(p, lk, dinv, tj); else solve_pivot_chain(p, lk, dinv, tj);
    if (TR) PPS_TR(4);
    if (lane < p) d.delta[pix] = tj;
    if (TR) PPS_TR(5);
    if (!GROUP) return;
    double* __restrict__ Xs = X + (size_t)slot * kBandMaxRows;
    if (lane < p) Xs[lane] = tj;
    if (lane < b) Xs[p + lane] = g0;
    if (lane + 64 < b) Xs[p + lane + 64] = g1;
    if (flow) flow_post(d, flow, slot, pslot < 0);                       // (behind the solution)
    if (XG) { if (__builtin_amdgcn_readlane(rec, 6) > 0) xg_post(*xg, s); }
    return;
  }
  // first batch of the panel (all of it for n <= 1024) issued right behind the index loads: the gather from delta below then waits
  // for the indices only, with the panel in flight
  double v[16];
#pragma unroll
  for (int u = 0; u < 16; u++) { const int e = 64 * u + lane; v[u] = Lp[e < n ? e : n - 1]; }
  const int ix0 = lane < b ? ix0r : 0, ix1 = lane + 64 < b ? ix1r : 0;
  double g0 = 0.0, g1 = 0.0;
  const int xpar = XG ? __builtin_amdgcn_readlane(rec, 13) : -1;
  if (pslot < 0 && xpar < 0) { g0 = d.delta[ix0]; g1 = d.delta[ix1]; }          // clamped index 0 when out of range: harmless
  // (entries past the end of the panel land in xb[127], which no front uses: b <= 126 -- unpredicated LDS writes)
#pragma unroll
  for (int u = 0; u < 16; u++) { const int e = 64 * u + lane; PL[e < n ? e : -1] = v[u]; }
  for (int e0 = 64 * 16; e0 < n; e0 += 64 * 16) {
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + 64 * u + lane; v[u] = Lp[e < n ? e : n - 1]; }
#pragma unroll
    for (int u = 0; u < 16; u++) { const int e = e0 + 64 * u + lane; PL[e < n ? e : -1] = v[u]; }
  }
  if (TR) PPS_TR(1);
  const int lc = lane < p ? lane : 0;
  int k0 = (p - 1) & ~15;
  bool pre_scaled = false;
  const bool pre32 = flow != nullptr && p <= 32;                 // (wave-uniform) L_B operands of y - L_B^T x_b in registers before the wait
  double lbq[12], yq = 0.0;
  if (flow) {
    // (data flow: the multipliers of the first sixteen pivots of the chain, the diagonal -- and for p <= 32 the entries of L_B this lane
    // multiplies, rows part, part + np, ... -- are read before the parent is waited for)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 16; u++) { const int k = k0 + u; const double l = PL[(k < p ? k : p - 1) * p + lc]; lk[u] = lane < k ? l : 0.0; }
    dinv = 1.0 / PL[lc * p + lc];
    if (k0 >= 16) {
      // the multipliers of the pivots >= 16 of the first chunk take their 1 / L_kk here (the product the chain below forms first)
#pragma unroll
      for (int u = 0; u < 16; u++) { const int k = k0 + u; if (k < p) lk[u] = lk[u] * readlane_d(dinv, k); }
      pre_scaled = true;
    }
    if (pre32) {
      const int sh = p <= 16 ? 4 : 5, np = 64 >> sh;
      const int j = lane & ((1 << sh) - 1), part = lane >> sh;
      const int jc = j < p ? j : 0;
      const int blast = b > 0 ? b - 1 : 0;
#pragma unroll
      for (int u = 0; u < 12; u++) { const int i = part + np * u; lbq[u] = PL[(p + (i < b ? i : blast)) * p + jc]; }
      yq = PL[f * p + jc];
    }
  }
  if (pslot >= 0) {
    if (flow) flow_wait(d, flow, pslot);
    const double* __restrict__ Xp = X + (size_t)pslot * kBandMaxRows;
    g0 = Xp[ix0]; g1 = Xp[ix1];
  } else if (XG && xpar >= 0) {
    xg_wait(d, *xg, xpar);
    g0 = d.delta[ix0]; g1 = d.delta[ix1];
  }
  if (lane < b) xb[lane] = g0;
  if (lane + 64 < b) xb[lane + 64] = g1;
  __builtin_amdgcn_wave_barrier();
  if (TR) PPS_TR(2);
  {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // dependent chains: a - b * c is one operation here
#endif
    // Everything the dependent chain of the back-substitution reads is prepared before the chain starts: lk[k - k0] of lane j holds
    // L_kj / L_kk for j < k (0 elsewhere), 16 pivots at a time, so that t_j -= lk * t_k is all a pivot costs -- v_readlane + fma, no
    // LDS round trip, no select -- and x = t / diag(L) is one multiplication at the end.
    if (!flow) {
#pragma unroll
      for (int u = 0; u < 16; u++) { const int k = k0 + u; const double l = PL[(k < p ? k : p - 1) * p + lc]; lk[u] = lane < k ? l : 0.0; }
      dinv = 1.0 / PL[lc * p + lc];
    }
    // y - L_B^T x_b: column j of L_B is summed by 64 / W lanes (W = 16, 32 or 64 >= p: rows i = part (mod 64 / W) each, four
    // independent partial sums per lane), then the parts are added across the wave -- 9 LDS rounds for b = 36, p = 15 instead of 36
    const int W = p <= 16 ? 16 : (p <= 32 ? 32 : 64), sh = p <= 16 ? 4 : (p <= 32 ? 5 : 6);
    const int j = lane & (W - 1), part = lane >> sh, np = 64 >> sh;
    const int jc = j < p ? j : 0;
    double a0 = (part == 0) ? (pre32 ? yq : PL[f * p + jc]) : 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* __restrict__ lb = PL + p * p + jc;
    int i_from = part;
    if (pre32) {
      // rows part + np u, u = 0 .. 11, from registers: the same sums in the same order as the loop below (u = 4 t + m: accumulator m)
#pragma unroll
      for (int u = 0; u < 12; u += 4) {
        const int i = part + np * u, i1 = i + np, i2 = i + 2 * np, i3 = i + 3 * np;
        const double x0 = i < b ? xb[i] : 0.0, x1 = i1 < b ? xb[i1] : 0.0, x2 = i2 < b ? xb[i2] : 0.0, x3 = i3 < b ? xb[i3] : 0.0;
        if (i < b) { a0 -= lbq[u] * x0; a1 -= lbq[u + 1] * x1; a2 -= lbq[u + 2] * x2; a3 -= lbq[u + 3] * x3; }
      }
      i_from = part + 12 * np;
    }
    for (int i = i_from; i < b; i += 4 * np) {            // (b is wave-uniform, `part` is not: rows past b contribute 0)
      const int i1 = i + np, i2 = i + 2 * np, i3 = i + 3 * np;
      const double l0 = lb[i * p], l1 = lb[(i1 < b ? i1 : i) * p], l2 = lb[(i2 < b ? i2 : i) * p], l3 = lb[(i3 < b ? i3 : i) * p];
      const double x0 = xb[i], x1 = i1 < b ? xb[i1] : 0.0, x2 = i2 < b ? xb[i2] : 0.0, x3 = i3 < b ? xb[i3] : 0.0;
      a0 -= l0 * x0; a1 -= l1 * x1; a2 -= l2 * x2; a3 -= l3 * x3;
    }
    tj = (a0 + a1) + (a2 + a3);
    if (W <= 32) tj += __shfl_xor(tj, 32);
    if (W == 16) tj += __shfl_xor(tj, 16);
    tj = lane < p ? tj : 0.0;
    if (TR) PPS_TR(3);
    // pivots 16 and up (leaves and roots only: p <= 16 on every other front): t_k crosses rows of 16 lanes through v_readlane
    for (; k0 >= 16; k0 -= 16) {
#pragma unroll
      for (int u = 15; u >= 0; u--) {
        const int k = k0 + u;
        if (k < p) tj -= (pre_scaled ? lk[u] : lk[u] * readlane_d(dinv, k)) * readlane_d(tj, k);        // (wave-uniform branch)
      }
      pre_scaled = false;
#pragma unroll
      for (int u = 0; u < 16; u++) { const int k = k0 - 16 + u; const double l = PL[k * p + lc]; lk[u] = lane < k ? l : 0.0; }
    }
    solve_pivot_chain(p, lk, dinv, tj);
  }
  if (TR) PPS_TR(4);
  if (lane < p) d.delta[pix] = tj;
  if (TR) PPS_TR(5);
  if (!GROUP) return;
  // own local solution [x_p | x_b] for the children inside this group
  double* __restrict__ Xs = X + (size_t)slot * kBandMaxRows;
  if (lane < p) Xs[lane] = tj;
  if (lane < b) Xs[p + lane] = g0;
  if (lane + 64 < b) Xs[p + lane + 64] = g1;
  if (flow) flow_post(d, flow, slot, pslot < 0);
  if (XG) { if (__builtin_amdgcn_readlane(rec, 6) > 0) xg_post(*xg, s); }
}

// The back-substitution of a band group as a data flow: the group's fronts, top level first, are dealt to the waves round-robin and
// every wave starts on its front at once -- record, index lists, factor panel, scaled multipliers: everything but the boundary values --
// and only then waits for its parent's solution (a flag per front in LDS).  What is left between a parent's solution and its child's is
// the gather of the boundary values, y - L_B^T x_b and the pivot chain; a level no longer costs the memory round trips of its panels,
// and no workgroup barrier holds the fast fronts of a level back.  Same arithmetic as body_band_solve, same bits.
// mg: solution slots behind the waves' scratch (fronts of the largest group of the stage); the flags sit behind them.
template <bool TR = false, bool XG = false>
__device__ __forceinline__ void body_band_solve_flow(const DevGraph& d, int g, int lds_doubles_per_wave, double* __restrict__ lds, int mg, const XGroup* xg = nullptr) {
  const int wave = uni(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* W = lds + (size_t)wave * lds_doubles_per_wave;
  double* X = lds + (size_t)nw * lds_doubles_per_wave;
  int* flow = reinterpret_cast<int*>(X + (size_t)mg * kBandMaxRows);
  const int2 span = reinterpret_cast<const int2*>(d.grp_span)[4 * (size_t)g];
  const int g0 = uni(span.x), gn = uni(span.y);
  // The group's fronts sit at consecutive positions, a parent behind its children: dealt from the last position down, every wave
  // meets a parent before any of its children -- nobody waits for a front that sits later in a queue, so the waits cannot form a cycle.
  // (the flags are cleared while the first records are on their way)
  const int i_first = g0 + gn - 1 - wave;
  int rec = i_first >= g0 ? d.frec[(size_t)i_first * 16 + (threadIdx.x & 15)] : 0;
  for (int q = threadIdx.x; q < gn; q += blockDim.x) flow[q] = 0;
  __syncthreads();
  for (int i = i_first; i >= g0; i -= nw) {
    wave_front_solve<true, TR, XG>(d, rec, W, X, i - g0, flow, xg);
    if (i - nw >= g0) rec = d.frec[(size_t)(i - nw) * 16 + (threadIdx.x & 15)];
  }
}
template <bool TR = false>
__device__ __forceinline__ void body_band_solve(const DevGraph& d, int g, int lds_doubles_per_wave, double* __restrict__ lds) {
  const int wave = uni(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* W = lds + (size_t)wave * lds_doubles_per_wave;
  double* X = lds + (size_t)nw * lds_doubles_per_wave;          // one local solution vector per front of the group
  const int l0 = uni(d.grp_lvl_off[g]), l1 = uni(d.grp_lvl_off[g + 1]);
  // The front lists of the group's levels and the record of this wave's first front on each of them are fetched before the walk
  // starts: a level then begins with the loads of its panel, not with two dependent round trips (offsets, record) behind the barrier.
  // (a group has at most band_levels <= 4 local levels; deeper ones would take the in-loop loads; the offsets come with the group's record)
  int off[5], r4[4];
  {
    const int4 ga = reinterpret_cast<const int4*>(d.grp_span)[2 * (size_t)g], gb = reinterpret_cast<const int4*>(d.grp_span)[2 * (size_t)g + 1];
    off[0] = uni(ga.x); off[1] = uni(ga.w); off[2] = uni(gb.x); off[3] = uni(gb.y); off[4] = uni(gb.z);
  }
  const int g0 = off[0];
#pragma unroll
  for (int k = 0; k < 4; k++) { const int i = off[k] + wave; r4[k] = d.frec[(size_t)(i < off[k + 1] ? i : g0) * 16 + (threadIdx.x & 15)]; }
  for (int l = l1 - 1; l >= l0; l--) {
    const int k = l - l0;
    const int i0 = k < 4 ? (k == 3 ? off[3] : k == 2 ? off[2] : k == 1 ? off[1] : off[0]) : uni(d.glvl_front_off[l]);
    const int i1 = k < 4 ? (k == 3 ? off[4] : k == 2 ? off[3] : k == 1 ? off[2] : off[1]) : uni(d.glvl_front_off[l + 1]);
    for (int i = i0 + wave; i < i1; i += nw) {
      const int rec = (k < 4 && i == i0 + wave) ? (k == 3 ? r4[3] : k == 2 ? r4[2] : k == 1 ? r4[1] : r4[0]) : d.frec[(size_t)i * 16 + (threadIdx.x & 15)];
      wave_front_solve<true, TR>(d, rec, W, X, i - g0);
    }
    // The children inside the group read their parent's solution from LDS (X), never from delta: the barrier orders LDS only.
    // (__syncthreads() is a workgroup-scope release -- it would also wait for the acknowledgement of the stores to delta, a
    // memory round trip per level)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

__global__ __launch_bounds__(512) void k_band_solve(DevGraph d, DualAlt alt, int grp_begin, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; }
  body_band_solve(d, grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
}
__global__ __launch_bounds__(768) void k_band_solve_flow(DevGraph d, DualAlt alt, int grp_begin, int lds_doubles_per_wave, int mg) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; }
  body_band_solve_flow(d, grp_begin + blockIdx.x, lds_doubles_per_wave, lds, mg);
}
__global__ __launch_bounds__(768) void k_band_solve_flow_trace(DevGraph d, int grp_begin, int lds_doubles_per_wave, int mg) {     // PPS_TRACE=2
  extern __shared__ double lds[];
  body_band_solve_flow<true>(d, grp_begin + blockIdx.x, lds_doubles_per_wave, lds, mg);
}
__global__ __launch_bounds__(512) void k_band_solve_trace(DevGraph d, int grp_begin, int lds_doubles_per_wave) {     // PPS_TRACE=2, PPS_NO_SOLVE_FLOW=1
  extern __shared__ double lds[];
  body_band_solve<true>(d, grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
}

// REG_ONLY: every front of the stage fits the register-resident path (C2: all stages) -- the LDS-tile path and the phase trace
// are compiled out, which halves the kernel's code (the instruction cache is shared by two CUs)
// REG_STRIP + R5 (with REG_ONLY): the stage also holds fronts of 65 .. 80 rows -- fifteen register tiles, k_band_factor_r5 -- and
// still nothing that needs the LDS-tile path or the trace (frame-loop trees, C3).  The general kernel (REG_ONLY = false: traces,
// pps_multi) carries such fronts as ten register tiles + a strip of the LDS triangle.
template <bool REG_ONLY, bool REG_STRIP = false, bool TR = false, bool R5 = false>
__device__ __forceinline__ void body_band_factor(const DevGraph& d, int g, double lambda, int lds_doubles_per_wave, double* __restrict__ lds) {
  const int wave = uni(threadIdx.x >> 6), nw = blockDim.x >> 6;
  double* F = lds + (size_t)wave * lds_doubles_per_wave;
  // (Fetching the level offsets and the wave's first record of every level up front, as body_band_solve does, was measured here
  // twice: held in registers the five live values push the register-tile code into spills inside the elimination -- C2 stage
  // 25.6 -> 33.4 us --; parked in LDS it gains nothing on C2 and costs C3 a third of its factor time.)
  const int l0 = d.grp_lvl_off[g], l1 = d.grp_lvl_off[g + 1];
  // (the first level's positions come with the group's record: one look-up less in front of the first front of the launch)
  const int4 ga = reinterpret_cast<const int4*>(d.grp_span)[2 * (size_t)g];
  for (int l = l0; l < l1; l++) {
    const int i1 = l == l0 ? ga.w : d.glvl_front_off[l + 1];
    for (int i = (l == l0 ? ga.x : d.glvl_front_off[l]) + wave; i < i1; i += nw) {
      const int rec = d.frec[(size_t)i * 16 + (threadIdx.x & 15)];          // packed front record, one coalesced load
      const int s = __builtin_amdgcn_readlane(rec, 0);
      const int fa = __builtin_amdgcn_readlane(rec, 1) + __builtin_amdgcn_readlane(rec, 2) + 1;
      // register-only kernels: the panel buffer is the head of the (by then dead) triangle, the spare double the last one of the
      // wave's LDS; the others keep the panel buffer (80 rows) behind triangle and spare double
      constexpr bool ALIAS = REG_ONLY && !REG_STRIP;
      double* const Pn = ALIAS ? F : F + lds_doubles_per_wave - kRegRowsMax * kP8Stride;
      const int tr = ALIAS ? lds_doubles_per_wave - 1 : lds_doubles_per_wave - kRegRowsMax * kP8Stride - 1;
      constexpr int WB = kPanelWBand, WR = kPanelWWide;
      // the assembly is the same code for every tile count (one copy in the kernel); the elimination is per tile count
      if (REG_ONLY || fa <= kRegRowsMax) {
        if (REG_ONLY) front_assemble<TR>(d, rec, lambda, F, tr);
        else if ((fa <= kRegRowsMax && !d.no_strip) || fa <= kRegRows) front_assemble<true>(d, rec, lambda, F, tr);
      }
      if (REG_ONLY && fa <= 33) front_eliminate_out<2, TR, false, REG_STRIP ? WR : WB>(d, rec, F, Pn);   // (without a strip the rhs row is a vector next to the tiles)
      else if (REG_ONLY && fa <= 49) front_eliminate_out<3, TR, false, REG_STRIP ? WR : WB>(d, rec, F, Pn);
      else if (REG_ONLY && (!REG_STRIP || fa <= kRegRows)) front_eliminate_out<4, TR, false, REG_STRIP ? WR : WB>(d, rec, F, Pn);
      else if (REG_ONLY && R5) front_eliminate_out<5, false, true, WR>(d, rec, F, Pn);             // 65 .. 80 rows: fifteen register tiles
      else if (fa <= kRegRowsMax && !d.no_strip) front_eliminate_out<4, true, true, WR>(d, rec, F, Pn);
      else if (fa <= kRegRows) front_eliminate_out<4, true, false, WR>(d, rec, F, Pn);
      else wave_front_factor(d, s, lambda, F);
    }
    __syncthreads();   // children of the next local level are complete and visible (same CU)
  }
}

// clear + original entries of a front, the first batch of NPRE x 64 entries requested before the clearing (front_pre_issue).  A
// separator front of a corridor tree has 120 - 180 original entries, a leaf up to 650: three loads per lane cover the former -- eight
// were five wasted load / read-modify-write pairs per lane (C2, same box, five runs each: 63.59 -> 63.33 us per LM iteration); eleven
// for the leaves (one round trip instead of two for the largest) measured the same as eight.
template <int NPRE>
__device__ __forceinline__ int front_orig_entries(const DevGraph& d, int rec, int lane, double damp, double* F, int tr) {
  FrontPre<NPRE> pre;
  front_pre_issue(d, rec, lane, pre);
  front_clear(rec, lane, F);
  front_pre_finish(d, rec, lane, pre, damp, F, tr);
  return pre.crv;
}
__device__ __forceinline__ int front_orig_entries_sized(const DevGraph& d, int rec, int lane, double damp, double* F, int tr) {
  const int n_el = __builtin_amdgcn_readlane(rec, 4) - __builtin_amdgcn_readlane(rec, 3);
  if (n_el <= 192) return front_orig_entries<3>(d, rec, lane, damp, F, tr);        // (wave-uniform)
  return front_orig_entries<8>(d, rec, lane, damp, F, tr);
}

// The register-only walk with the fronts of the upper local levels on waves of their own.  A band group of a dissection tree is a
// sub-tree of 8 + 4 + 2 + 1 fronts walked by 8 waves: from the second level on, half of the waves -- and the LDS triangles they own --
// idle.  Here the fronts of local level 2 (and 3) are dealt to the waves that have nothing to do on level 1: while level 1 is being
// eliminated, such a wave clears its triangle and gathers the ORIGINAL entries of its upper front into it -- the part of the assembly
// that does not depend on the children --, so that its level starts with the extend-add.  Same order of the sums (original entries,
// then the children in order): same bits.  Groups of another shape (more fronts on a level than waves, fewer than three levels, upper
// levels that do not fit the idle waves) take the plain walk.
// XG: the launch holds every band group of the tree (k_band_factor_all): children factored by other workgroups are waited for, and a
// group's top front raises its flag behind its update matrix
template <bool XG = false>
__device__ __forceinline__ void body_band_factor_pre(const DevGraph& d, int g, double lambda, int lds_doubles_per_wave, double* __restrict__ lds, const XGroup* xg = nullptr) {
  // (the host has checked the shape of every group of the stage: stage_pre in pps_upload.cpp.  Few values stay live across the levels:
  // this kernel's register allocation is at its limit, see body_band_factor)
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = uni(blockDim.x >> 6);
  const int4 ga = reinterpret_cast<const int4*>(d.grp_span)[2 * (size_t)g], gb = reinterpret_cast<const int4*>(d.grp_span)[2 * (size_t)g + 1];
  const int nl = uni(ga.z);
  const int o0 = uni(ga.x), o1 = uni(ga.w), o2 = uni(gb.x), o3 = uni(gb.y), o4 = uni(gb.z);      // first positions of the local levels
  const int n0 = o1 - o0, n1 = nl > 1 ? o2 - o1 : 0, n2 = nl > 2 ? o3 - o2 : 0, n3 = nl > 3 ? o4 - o3 : 0;
  double* F = lds + (size_t)wave * lds_doubles_per_wave;
  const int tr = lds_doubles_per_wave - 1;
  // Which wave owns which front.  Shape B -- the whole group fits the waves (4 + 2 + 1 on eight): every upper front has a wave of its
  // own, which assembles what needs no child while local level 0 is eliminated.  Shape A (8 + 4 + 2 + 1 on eight): level 1 on the waves
  // that eliminated level 0, levels 2 and 3 on waves that idle from level 1 on and pre-assemble there.
  const bool shape_b = n0 + n1 + n2 + n3 <= nw;
  const int b1 = shape_b ? n0 : 0, b2 = shape_b ? n0 + n1 : n1, b3 = b2 + n2;                    // first owner wave of local levels 1, 2, 3
  // this wave's front above level 0 that is assembled ahead of its level: its record and local level (0: none)
  int up_ll = 0, up_rec = 0;
  {
    int up_i = 0;
    if (shape_b && wave >= b1 && wave < b1 + n1) { up_ll = 1; up_i = o1 + wave - b1; }
    else if (wave >= b2 && wave < b2 + n2 && (shape_b || wave >= n1)) { up_ll = 2; up_i = o2 + wave - b2; }
    else if (wave >= b3 && wave < b3 + n3) { up_ll = 3; up_i = o3 + wave - b3; }
    if (up_ll) up_rec = d.frec[(size_t)up_i * 16 + (threadIdx.x & 15)];
  }
  const int pre_at = shape_b ? 0 : 1;                           // the level during which the upper fronts are pre-assembled
  int crv = 0;
  for (int ll = 0; ll < nl; ll++) {
    const int lb = ll == 0 ? 0 : ll == 1 ? b1 : ll == 2 ? b2 : b3;                               // owners of this level: waves [lb, lb + cnt)
    const int cnt = ll == 0 ? n0 : ll == 1 ? n1 : ll == 2 ? n2 : n3;
    const int i0 = ll == 0 ? o0 : ll == 1 ? o1 : ll == 2 ? o2 : o3;
    const bool mine = wave >= lb && wave < lb + cnt;
    const bool mine_up = mine && up_ll == ll && ll > pre_at;    // (assembled ahead: starts with the extend-add)
    if (mine) {                                                 // (wave-uniform)
      const int rec = (up_ll == ll && ll > 0) ? up_rec : d.frec[(size_t)(i0 + wave - lb) * 16 + (threadIdx.x & 15)];
      const int fa = __builtin_amdgcn_readlane(rec, 1) + __builtin_amdgcn_readlane(rec, 2) + 1;
      if (!mine_up) crv = front_orig_entries_sized(d, rec, lane, 1.0 + lambda, F, tr);
      front_extend_add<XG>(d, rec, crv, lane, F, tr, xg);
      if (fa <= 33) front_eliminate_out<2, false, false, kPanelWBand>(d, rec, F, F);
      else if (fa <= 49) front_eliminate_out<3, false, false, kPanelWBand>(d, rec, F, F);
      else front_eliminate_out<4, false, false, kPanelWBand>(d, rec, F, F);
      // (slot 13: the parent's front when another group holds it; SW_DEBUG_DROP_XFLAG, tests only: the flag is withheld and the parent's workgroup runs into its time-out)
      if (XG) { if (__builtin_amdgcn_readlane(rec, 13) >= 0 && !(d.sw & SW_DEBUG_DROP_XFLAG)) xg_post(*xg, __builtin_amdgcn_readlane(rec, 0)); }
    } else if (ll == pre_at && up_ll > pre_at) {
      // nothing to eliminate on this level: the part of the upper front's assembly that needs no child
      crv = front_orig_entries_sized(d, up_rec, lane, 1.0 + lambda, F, tr);
    }
    __syncthreads();   // children of the next local level are complete and visible (same CU)
  }
}

template <bool REG_ONLY>
__global__ __launch_bounds__(512) void k_band_factor(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  body_band_factor<REG_ONLY>(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, lds);
}

// The root stage of a tree whose top is ONE group of register-resident fronts (C2: the root front), factored and solved by one launch:
// the workgroup that has eliminated the top fronts walks back down them.  L and the forward-solved rhs rows reach the back-substitution
// through the CU's own cache behind the workgroup barrier that ends the last level of the factorisation.
// PRE: the group has the shape of the pre-assembling walk (stage_pre); FLOW: the walk back down as a data flow (a top group of several
// fronts: 4 + 2 + 1 when the tree is banded three levels per launch), mg = fronts of the group.
template <bool PRE, bool FLOW>
__global__ __launch_bounds__(512) void k_band_root(DevGraph d, DualAlt alt, int grp, double lambda, int per_wave_factor, int per_wave_solve, int mg) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  if (PRE) body_band_factor_pre(d, grp, lambda, per_wave_factor, lds); else body_band_factor<true>(d, grp, lambda, per_wave_factor, lds);
  if (FLOW) body_band_solve_flow(d, grp, per_wave_solve, lds, mg); else body_band_solve(d, grp, per_wave_solve, lds);
}

// The WHOLE tree in one launch (round 6; graphs whose every stage has the shape of the pre-assembling walk and whose groups fit the chip a
// few times over): workgroup b = band group b, leaves first -- the groups are numbered stage by stage --, hand-over of a group's top front to
// its parent's workgroup through XGroup flags.  A stage boundary no longer costs a launch boundary, and the fronts of the next stage are
// cleared and receive their original entries while their children are still being eliminated.
__global__ __launch_bounds__(512) void k_band_factor_all(DevGraph d, DualAlt alt, double lambda, int lds_doubles_per_wave, int epoch) {
  extern __shared__ double lds[];
  XGroup xg{d.k3_flag, epoch};
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; xg.flag += d.n_fronts; }
  body_band_factor_pre<true>(d, blockIdx.x, lambda, lds_doubles_per_wave, lds, &xg);
}
// ... and the whole back-substitution in one: workgroup b = group n_groups - 1 - b, the top group first
__global__ __launch_bounds__(768) void k_band_solve_all(DevGraph d, DualAlt alt, int n_groups, int lds_doubles_per_wave, int mg, int epoch) {
  extern __shared__ double lds[];
  XGroup xg{d.k3_flag + 2 * d.n_fronts, epoch};
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; xg.flag += d.n_fronts; }
  body_band_solve_flow<false, true>(d, n_groups - 1 - (int)blockIdx.x, lds_doubles_per_wave, lds, mg, &xg);
}

__global__ __launch_bounds__(512) void k_band_factor_pre(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  body_band_factor_pre(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, lds);
}

// PPS_TRACE=1 on a register-only stage: the phase trace compiled into the register-only kernel
__global__ __launch_bounds__(512) void k_band_factor_lean_trace(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  body_band_factor<true, false, true>(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, lds);
}

// Stages with fronts of 65 .. 80 rows, all of it in registers: four waves per workgroup, one per SIMD, so that a wave may hold
// fifteen accumulator tiles (120 registers) next to everything else -- the unified register file gives a lone wave 512.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_band_factor_r5(DevGraph d, DualAlt alt, int grp_begin, double lambda, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  if (blockIdx.y) { d.L = alt.L; d.U = alt.U; d.delta = alt.delta; d.result_dev = alt.result_dev; lambda = alt.lambda; }
  body_band_factor<true, true, false, true>(d, grp_begin + blockIdx.x, lambda, lds_doubles_per_wave, lds);
}

static std::atomic<bool> g_band_attr_set[64];   // per device ordinal

static hipError_t ensure_band_attrs() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!g_band_attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve_trace), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve_flow), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_pre), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve_flow_trace), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_r5), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_lean_trace), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_root<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_root<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_root<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_root<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_all), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_solve_all), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    g_band_attr_set[dev & 63] = true;
  }
  return hipSuccess;
}

// one launcher for both: ny = 1 (alt unused) or 2 (dual)
static hipError_t launch_band_factor_impl(const DevGraph& d, const DualAlt& alt, int ny, int grp_begin, int grp_count, int nwaves, int max_front, double lambda,
                                          hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool pre = false) {
  if (grp_count == 0) return hipSuccess;
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const bool reg_only = max_front + 1 <= kRegRows && !(ny == 2 && d.trace != nullptr);     // (a traced dual solve runs the general kernel)
  const int per_wave = (int)(band_lds_bytes(max_front, reg_only) / sizeof(double));
  const size_t bytes = (size_t)per_wave * nwaves * sizeof(double);
  if (reg_only && d.trace != nullptr)      // phase trace of the register-only kernel
    PPS_LAUNCH(k_band_factor_lean_trace, dim3(grp_count, ny), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave);
  else if (reg_only && pre)
    PPS_LAUNCH_EV(ev0, ev1, k_band_factor_pre, dim3(grp_count, ny), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave);
  else if (reg_only)
    PPS_LAUNCH_EV(ev0, ev1, k_band_factor<true>, dim3(grp_count, ny), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave);
  else if (max_front + 1 <= kRegRowsMax && d.trace == nullptr && !d.no_strip) {
    const int nw5 = nwaves < 4 ? nwaves : 4;                // (one wave per SIMD: see k_band_factor_r5)
    PPS_LAUNCH_EV(ev0, ev1, k_band_factor_r5, dim3(grp_count, ny), dim3(64 * nw5), (size_t)per_wave * nw5 * sizeof(double), st, d, alt, grp_begin, lambda, per_wave);
  } else
    PPS_LAUNCH_EV(ev0, ev1, k_band_factor<false>, dim3(grp_count, ny), dim3(64 * nwaves), bytes, st, d, alt, grp_begin, lambda, per_wave);
  return hipGetLastError();
}

hipError_t launch_band_factor(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_front, double lambda, hipStream_t st, hipEvent_t ev0,
                              hipEvent_t ev1, bool pre) {
  return launch_band_factor_impl(d, DualAlt{}, 1, grp_begin, grp_count, nwaves, max_front, lambda, st, ev0, ev1, pre);
}

hipError_t launch_band_factor_dual(const DevGraph& d, const DualAlt& alt, int grp_begin, int grp_count, int nwaves, int max_front, double lambda,
                                   hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, bool pre) {
  return launch_band_factor_impl(d, alt, 2, grp_begin, grp_count, nwaves, max_front, lambda, st, ev0, ev1, pre);
}

size_t band_solve_lds_bytes(int max_panel) { return (size_t)(kBandMaxRows + max_panel) * sizeof(double); }   // xb + the factor panel

hipError_t launch_band_solve(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_panel, int max_group_fronts, hipStream_t st,
                             const DualAlt* alt) {
  if (grp_count == 0) return hipSuccess;
  // per wave: xb + the largest factor panel of the stage; per workgroup: one local solution vector per front of a group
  const int per_wave = (int)(band_solve_lds_bytes(max_panel) / sizeof(double));
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const size_t bytes = ((size_t)per_wave * nwaves + (size_t)max_group_fronts * kBandMaxRows) * sizeof(double);
  const bool no_flow = (d.sw & SW_NO_SOLVE_FLOW) != 0;
  if (!no_flow && !(d.trace != nullptr && d.trace_solve && alt) && max_group_fronts > 1) {
    // data-flow form: up to twelve waves (three per SIMD), as many as the group has fronts and the LDS holds
    const size_t fixed = ((size_t)max_group_fronts * kBandMaxRows + (size_t)(max_group_fronts + 1) / 2) * sizeof(double);
    const size_t room = (size_t)kLdsLimitBytes > fixed ? ((size_t)kLdsLimitBytes - fixed) / ((size_t)per_wave * sizeof(double)) : 0;
    const int nwf = (int)std::min<size_t>(std::min<size_t>(12, (size_t)max_group_fronts), room);
    if (nwf >= 1) {
      if (d.trace != nullptr && d.trace_solve)
        PPS_LAUNCH(k_band_solve_flow_trace, dim3(grp_count), dim3(64 * nwf), (size_t)per_wave * nwf * sizeof(double) + fixed, st, d, grp_begin, per_wave, max_group_fronts);
      else
        PPS_LAUNCH(k_band_solve_flow, dim3(grp_count, alt ? 2 : 1), dim3(64 * nwf), (size_t)per_wave * nwf * sizeof(double) + fixed, st, d, alt ? *alt : DualAlt{}, grp_begin,
                   per_wave, max_group_fronts);
      return hipGetLastError();
    }
  }
  if (d.trace != nullptr && d.trace_solve && !alt) PPS_LAUNCH(k_band_solve_trace, dim3(grp_count), dim3(64 * nwaves), bytes, st, d, grp_begin, per_wave);
  else PPS_LAUNCH(k_band_solve, dim3(grp_count, alt ? 2 : 1), dim3(64 * nwaves), bytes, st, d, alt ? *alt : DualAlt{}, grp_begin, per_wave);
  return hipGetLastError();
}

// the root stage as one launch (k_band_root): one group, every front register-resident, no trace
bool band_root_fusable(const DevGraph& d, int grp_count, int max_front) {
  const bool off = (d.sw & SW_NO_ROOT_FUSE) != 0;
  return !off && grp_count == 1 && max_front + 1 <= kRegRows && d.trace == nullptr;
}
hipError_t launch_band_root(const DevGraph& d, const DualAlt* alt, int grp, int nwaves_factor, int nwaves_solve, int max_front, int max_panel,
                            int max_group_fronts, double lambda, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, bool pre) {
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const int pwf = (int)(band_lds_bytes(max_front, true) / sizeof(double)), pws = (int)(band_solve_lds_bytes(max_panel) / sizeof(double));
  // the walk back down: as a data flow when the group has more than one front and the LDS holds a wave per front (at most eight here)
  const size_t flow_fixed = ((size_t)max_group_fronts * kBandMaxRows + (size_t)(max_group_fronts + 1) / 2) * sizeof(double);
  int nw = nwaves_factor > nwaves_solve ? nwaves_factor : nwaves_solve;
  bool flow = !(d.sw & SW_NO_SOLVE_FLOW) && max_group_fronts > 1;
  if (flow) {
    const int want = nwaves_factor > (max_group_fronts < 8 ? max_group_fronts : 8) ? nwaves_factor : (max_group_fronts < 8 ? max_group_fronts : 8);
    if ((size_t)pws * want * sizeof(double) + flow_fixed <= (size_t)kLdsLimitBytes) nw = want; else flow = false;
  }
  const size_t bf = (size_t)pwf * nw * sizeof(double);
  const size_t bs = flow ? (size_t)pws * nw * sizeof(double) + flow_fixed : ((size_t)pws * nw + (size_t)max_group_fronts * kBandMaxRows) * sizeof(double);
  const size_t bytes = bf > bs ? bf : bs;
  const DualAlt a2 = alt ? *alt : DualAlt{};
  if (pre && flow) PPS_LAUNCH_EV(ev0, ev1, (k_band_root<true, true>), dim3(1, alt ? 2 : 1), dim3(64 * nw), bytes, st, d, a2, grp, lambda, pwf, pws, max_group_fronts);
  else if (pre) PPS_LAUNCH_EV(ev0, ev1, (k_band_root<true, false>), dim3(1, alt ? 2 : 1), dim3(64 * nw), bytes, st, d, a2, grp, lambda, pwf, pws, max_group_fronts);
  else if (flow) PPS_LAUNCH_EV(ev0, ev1, (k_band_root<false, true>), dim3(1, alt ? 2 : 1), dim3(64 * nw), bytes, st, d, a2, grp, lambda, pwf, pws, max_group_fronts);
  else PPS_LAUNCH_EV(ev0, ev1, (k_band_root<false, false>), dim3(1, alt ? 2 : 1), dim3(64 * nw), bytes, st, d, a2, grp, lambda, pwf, pws, max_group_fronts);
  return hipGetLastError();
}

// the whole tree: one factor launch + one back-substitution launch (k_band_factor_all / k_band_solve_all)
// nwaves_factor / max_front / max_panel / max_group_fronts: maxima over the stages; epoch: number of this launch pair (> 0, never repeated)
hipError_t launch_band_all(const DevGraph& d, const DualAlt* alt, int n_groups, int nwaves_factor, int nwaves_solve, int max_front, int max_panel,
                           int max_group_fronts, double lambda, int epoch, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  { const hipError_t e = ensure_band_attrs(); if (e != hipSuccess) return e; }
  const int pwf = (int)(band_lds_bytes(max_front, true) / sizeof(double)), pws = (int)(band_solve_lds_bytes(max_panel) / sizeof(double));
  const DualAlt a2 = alt ? *alt : DualAlt{};
  PPS_LAUNCH_EV(ev0, ev1, k_band_factor_all, dim3(n_groups, alt ? 2 : 1), dim3(64 * nwaves_factor), (size_t)pwf * nwaves_factor * sizeof(double), st, d, a2, lambda, pwf, epoch);
  const size_t fixed = ((size_t)max_group_fronts * kBandMaxRows + (size_t)(max_group_fronts + 1) / 2) * sizeof(double);
  PPS_LAUNCH(k_band_solve_all, dim3(n_groups, alt ? 2 : 1), dim3(64 * nwaves_solve), (size_t)pws * nwaves_solve * sizeof(double) + fixed, st, d, a2, n_groups, pws, max_group_fronts, epoch);
  return hipGetLastError();
}
// waves of the data-flow back-substitution of such a launch (0: the LDS does not hold a group): as launch_band_solve picks them per stage
int band_all_solve_waves(int max_panel, int max_group_fronts) {
  const size_t per_wave = band_solve_lds_bytes(max_panel);
  const size_t fixed = ((size_t)max_group_fronts * kBandMaxRows + (size_t)(max_group_fronts + 1) / 2) * sizeof(double);
  const size_t room = (size_t)kLdsLimitBytes > fixed ? ((size_t)kLdsLimitBytes - fixed) / per_wave : 0;
  return (int)std::min<size_t>(std::min<size_t>(12, (size_t)max_group_fronts), room);
}

// Back-substitution for one level (parents already solved): x_p = L_A^-T (y - L_B^T x_b).
__global__ __launch_bounds__(64) void k_front_solve(DevGraph d, int level_begin) {
  __shared__ double t[256];
  const int s = d.level_fronts[level_begin + blockIdx.x];
  const int p = d.f_p[s], b = d.f_b[s], f = p + b;
  const double* __restrict__ Lp = d.L + d.f_Loff[s];
  const int* __restrict__ bi = d.bidx + d.f_bidx_off[s];
  const int lane = threadIdx.x;
  for (int k = lane; k < p; k += 64) {
    double acc = Lp[(size_t)f * p + k];                      // y_k (forward-solved rhs row)
    for (int i = 0; i < b; i++) acc -= Lp[(size_t)(p + i) * p + k] * d.delta[bi[i]];
    t[k] = acc;
  }
  __syncthreads();
  for (int k = p - 1; k >= 0; k--) {
    const double xk = t[k] / Lp[(size_t)k * p + k];
    __syncthreads();
    for (int j = lane; j < k; j += 64) t[j] -= Lp[(size_t)k * p + j] * xk;
    if (lane == 0) t[k] = xk;
    __syncthreads();
  }
  for (int k = lane; k < p; k += 64) d.delta[d.pidx[d.f_poff[s] + k]] = t[k];
}

hipError_t launch_backsolve_level(const DevGraph& d, int level_begin, int level_count, hipStream_t st) {
  if (level_count == 0) return hipSuccess;
  PPS_LAUNCH(k_front_solve, dim3(level_count), dim3(64), 0, st, d, level_begin);
  return hipGetLastError();
}

// ---- batched forms ----
template <bool REG_ONLY>
__global__ __launch_bounds__(512) void kb_band_factor(BatchArgs a, int stage, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  DevGraph d2 = d;                                               // (one copy of the body for both damping values)
  double lam = a.lambda[b];
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta; d2.result_dev = al.result_dev;
    lam = a.lambda2[b];
  }
  body_band_factor<REG_ONLY>(d2, sg.grp_begin + blockIdx.x, lam, lds_doubles_per_wave, lds);
}

__global__ __launch_bounds__(512) void kb_band_factor_pre(BatchArgs a, int stage, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  DevGraph d2 = d;                                               // (one copy of the body for both damping values)
  double lam = a.lambda[b];
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta; d2.result_dev = al.result_dev;
    lam = a.lambda2[b];
  }
  body_band_factor_pre(d2, sg.grp_begin + blockIdx.x, lam, lds_doubles_per_wave, lds);
}

__global__ __launch_bounds__(768) void kb_band_solve_flow(BatchArgs a, int stage, int lds_doubles_per_wave, int mg) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  DevGraph d2 = d;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta; d2.result_dev = al.result_dev;
  }
  body_band_solve_flow(d2, sg.grp_begin + blockIdx.x, lds_doubles_per_wave, lds, mg);
}

__global__ __launch_bounds__(512) void kb_band_solve(BatchArgs a, int stage, int lds_doubles_per_wave) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  const BatchStage sg = a.stage_tab[(size_t)stage * a.n_total + a.b0 + b];
  if ((int)blockIdx.x >= sg.grp_count) return;
  DevGraph d2 = d;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta;
  }
  body_band_solve(d2, sg.grp_begin + blockIdx.x, lds_doubles_per_wave, lds);
}

// ---- level-per-launch form of a batch (BatchGeom::level_form) ----
// One launch per tree level and size class; a workgroup is four independent fronts (no group, no barrier), the kernel of a
// class holds exactly its tile rows: 86 / 118 VGPRs for fronts of <= 32 / <= 48 rows against the 256 of the band kernel that
// carries all three, so 5 / 3 waves share a SIMD instead of 2 (the 12 KB triangle of a 48-row front is what stops at 3).
#define PPS_LEVEL_FACTOR_KERNEL(NAME, NT, WAVES)                                                                      \
  __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void NAME(BatchArgs a, int level,   \
                                                                                                int lds_doubles_per_wave) { \
    extern __shared__ double lds[];                                                                                  \
    PPS_BATCH_PROLOGUE(BF_ACTIVE)                                                                                    \
    if (level >= d.n_levels) return;                                                                                 \
    const int wave = uni(threadIdx.x >> 6);                                                                          \
    const int k = d.cls_off[3 * level + (NT - 2)] + (int)blockIdx.x * 4 + wave;                                     \
    if (k >= d.cls_off[3 * level + (NT - 2) + 1]) return;                                                           \
    const int rec = d.frec[(size_t)d.cls_fronts[k] * 16 + (threadIdx.x & 15)];                                      \
    double* F = lds + (size_t)wave * lds_doubles_per_wave;                                                           \
    double* const Pn = F;                     /* (panel buffer = head of the dead triangle: band_lds_bytes) */          \
    const int tr = lds_doubles_per_wave - 1;                                                                         \
    DevGraph d2 = d;                            /* (one copy of the front's code for both damping values) */               \
    double lam = a.lambda[b];                                                                                        \
    if (blockIdx.z) {                                                                                                \
      const BatchAlt al = load_alt(a.alt + a.b0 + b);                                                                \
      d2.L = al.L; d2.U = al.U; d2.delta = al.delta; d2.result_dev = al.result_dev;                                  \
      lam = a.lambda2[b];                                                                                            \
    }                                                                                                                \
    wave_front_factor_reg<NT, false, false, kPanelWLevel>(d2, rec, lam, F, Pn, tr);                             \
  }
PPS_LEVEL_FACTOR_KERNEL(kb_level_factor2, 2, 5)
PPS_LEVEL_FACTOR_KERNEL(kb_level_factor3, 3, 3)
PPS_LEVEL_FACTOR_KERNEL(kb_level_factor4, 4, 2)
#undef PPS_LEVEL_FACTOR_KERNEL

// The back-substitution of one front in the level-per-launch form (large batches: throughput, every launch issue-bound).  The same sums in
// the same order as wave_front_solve<false> -- same bits -- without what a front on a latency path pays to hide its round trips: the
// panel is copied in as many batches as it has (a leaf of 46 x 18: 13, a separator of 40 x 6: 4 -- not 16), through ONE address register
// per side (the batches are immediate offsets; the last one reads up to 63 doubles past the panel -- readable: L is followed by U in the
// arena -- and writes them into the wave's own LDS slack); the pivots >= 16 of a leaf are a loop over the pivots it has instead of a
// sixteen-way unrolled chunk; no second index register (b <= 64).  Fronts outside p <= 32, b <= 64, panel <= 1024 entries take the
// general body.  G = 128: 10.9 -> ... ms of back-substitution per batch solve.
constexpr int kLevelSolveSlack = 64;       // doubles of LDS behind a wave's panel area (the last copy batch may overrun the panel)
// DIRECT (round 5): L_B never enters the LDS -- every lane loads exactly the entries of its column that it multiplies (rows part,
// part + np, ...: at most sixteen loads, issued with the copy of L_A) and the products read registers; the LDS of a wave is x_b and the
// p x p pivot block only (a 53 x 27 leaf: 7 KB instead of 12.7 KB, five waves per SIMD instead of three at the level where the waves
// spend two thirds of their cycles waiting for memory).  The sums run in the order of the LDS form: same bits.
// eligible: level_solve_direct_ok (shared with the host's LDS sizing)
__host__ __device__ __forceinline__ constexpr bool level_solve_direct_ok(int p, int b) { return p >= 1 && p <= 32 && (p <= 16 ? b <= 64 : b <= 32); }
template <int SH, bool DIRECT = false>     // 1 << SH lanes per column of L_B: 16 (p <= 16) or 32 (p <= 32)
__device__ __forceinline__ void wave_front_solve_level(const DevGraph& d, int rec, double* __restrict__ Wk) {
#ifndef PPS_NO_FMA
#pragma clang fp contract(fast)     // dependent chains: a - b * c is one operation here (as in wave_front_solve)
#endif
  const int lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readlane(rec, 1), b = __builtin_amdgcn_readlane(rec, 2), f = p + b;
  const double* __restrict__ Lp = d.L + (((long long)__builtin_amdgcn_readlane(rec, 10) << 32) | (unsigned int)__builtin_amdgcn_readlane(rec, 9));
  const int* __restrict__ ix = b == 0 ? d.frec : d.bidx + __builtin_amdgcn_readlane(rec, 8);
  double* __restrict__ xb = Wk;
  double* __restrict__ PL = Wk + kBandMaxRows;
  const int ix0 = ix[lane < b ? lane : 0];
  const int pix = d.pidx[__builtin_amdgcn_readlane(rec, 7) + (lane < p ? lane : 0)];
  const int n = DIRECT ? p * p : (f + 1) * p;                  // (DIRECT: only L_A, the first p rows of the panel, is copied)
  const double* __restrict__ src = Lp + lane;
  double* __restrict__ dst = PL + lane;
  constexpr int np = 64 >> SH;
  const int j = lane & ((1 << SH) - 1), part = lane >> SH;
  const int jc = j < p ? j : 0;
  const int blast = b > 0 ? b - 1 : 0;
  double v[16];
#pragma unroll
  for (int u = 0; u < 16; u++) if (64 * u < n) v[u] = src[64 * u];
  double lbv[16], yr = 0.0;
  if (DIRECT) {
    const double* __restrict__ LB = Lp + p * p + jc;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (4 * q * np < b) {                                    // (wave-uniform) rows part + np (4 q + m), m = 0 .. 3
#pragma unroll
        for (int m = 0; m < 4; m++) { const int i = part + np * (4 * q + m); lbv[4 * q + m] = LB[(i < blast ? i : blast) * p]; }
      }
    yr = Lp[f * p + jc];
  }
  const double g0 = d.delta[lane < b ? ix0 : 0];
#pragma unroll
  for (int u = 0; u < 16; u++) if (64 * u < n) dst[64 * u] = v[u];
  xb[lane] = lane < b ? g0 : 0.0;                        // (zero beyond b: rows past the boundary add l * 0 below)
  xb[lane + 64] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const int lc = lane < p ? lane : 0;
  const double dinv = 1.0 / PL[lc * p + lc];
  // y - L_B^T x_b: column j by 64 >> SH lanes, rows part (mod np), four partial sums per lane (rows part + np (4 t + m) in a_m)
  double a0 = part == 0 ? (DIRECT ? yr : PL[f * p + jc]) : 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const double* __restrict__ xr = xb + part;
  if (DIRECT) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (4 * q * np < b) {
        const int i0 = 4 * q * np;
        const double x0 = xr[i0], x1 = xr[i0 + np], x2 = xr[i0 + 2 * np], x3 = xr[i0 + 3 * np];
        a0 -= lbv[4 * q] * x0; a1 -= lbv[4 * q + 1] * x1; a2 -= lbv[4 * q + 2] * x2; a3 -= lbv[4 * q + 3] * x3;
      }
  } else {
    const double* __restrict__ lb = PL + p * p + jc;
    for (int i0 = 0; i0 < b; i0 += 4 * np) {               // (wave-uniform trip count)
      const int i = i0 + part, i1 = i + np, i2 = i + 2 * np, i3 = i + 3 * np;
      const double l0 = lb[(i < blast ? i : blast) * p], l1 = lb[(i1 < blast ? i1 : blast) * p], l2 = lb[(i2 < blast ? i2 : blast) * p], l3 = lb[(i3 < blast ? i3 : blast) * p];
      const double x0 = xr[i0], x1 = xr[i0 + np], x2 = xr[i0 + 2 * np], x3 = xr[i0 + 3 * np];
      a0 -= l0 * x0; a1 -= l1 * x1; a2 -= l2 * x2; a3 -= l3 * x3;
    }
  }
  double tj = (a0 + a1) + (a2 + a3);
  tj += __shfl_xor(tj, 32);
  if (SH == 4) tj += __shfl_xor(tj, 16);
  tj = lane < p ? tj : 0.0;
  for (int k = p - 1; k >= 16; k--) {                    // (leaves only)
    const double l = PL[k * p + lc];
    tj -= ((lane < k ? l : 0.0) * readlane_d(dinv, k)) * readlane_d(tj, k);
  }
  double lk[16];
#pragma unroll
  for (int u = 0; u < 16; u++) { const double l = PL[(u < p ? u : p - 1) * p + lc]; lk[u] = lane < u ? l : 0.0; }
  solve_pivot_chain(p, lk, dinv, tj);
  if (lane < p) d.delta[pix] = tj;
}

// direct: the host has checked that every front of the level is level_solve_direct_ok and sized the LDS for x_b + p x p
__global__ __launch_bounds__(256) void kb_level_solve(BatchArgs a, int level, int lds_doubles_per_wave, int direct) {
  extern __shared__ double lds[];
  PPS_BATCH_PROLOGUE(BF_ACTIVE)
  if (level >= d.n_levels) return;
  const int wave = uni(threadIdx.x >> 6);
  const int k = d.cls_off[3 * level] + (int)blockIdx.x * 4 + wave;
  if (k >= d.cls_off[3 * level + 3]) return;
  const int rec = d.frec[(size_t)d.cls_fronts[k] * 16 + (threadIdx.x & 15)];
  double* W = lds + (size_t)wave * lds_doubles_per_wave;
  DevGraph d2 = d;
  if (blockIdx.z) {
    const BatchAlt al = load_alt(a.alt + a.b0 + b);
    d2.L = al.L; d2.U = al.U; d2.delta = al.delta;
  }
  const int fp = __builtin_amdgcn_readlane(rec, 1), fb = __builtin_amdgcn_readlane(rec, 2);
  if (direct) {                                         // (kernel argument: uniform)
    if (fp <= 16) wave_front_solve_level<4, true>(d2, rec, W); else wave_front_solve_level<5, true>(d2, rec, W);
    return;
  }
  if (fp <= 32 && fb <= 64 && (fp + fb + 1) * fp <= 1024) {
    if (fp <= 16) wave_front_solve_level<4>(d2, rec, W); else wave_front_solve_level<5>(d2, rec, W);
    return;
  }
  wave_front_solve<false>(d2, rec, W, nullptr, 0);
}

// level-per-launch kernels: the packed triangle of a front of <= 16 nt rows (+ rhs), whose head doubles as the 8-column panel buffer, + the spare double
bool band_level_solve_direct_ok(int p, int b) { return level_solve_direct_ok(p, b); }
static int level_lds_doubles(int nt) {
  const size_t fa = (size_t)16 * nt + 1, n = std::max<size_t>(fa * (fa + 1) / 2, (size_t)kRegRows * kP8Stride) + 1;
  return (int)((n + 1) & ~size_t(1));
}

static std::atomic<bool> g_batch_attr_set[64];

hipError_t launch_batch_solve(const BatchArgs& a, const BatchGeom& g, hipStream_t st, hipEvent_t after_factor) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!g_batch_attr_set[dev & 63]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_factor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_factor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_solve), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_solve_flow), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_band_factor_pre), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_level_factor2), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_level_factor3), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_level_factor4), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kb_level_solve), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes);
    if (e != hipSuccess) return e;
    g_batch_attr_set[dev & 63] = true;
  }
  if (g.level_form) {
    const int nz = a.alt ? 2 : 1;
    for (int l = 0; l < g.n_levels; l++) {
      if (g.lvl_cls_blocks[l][0] > 0) { const int pw = level_lds_doubles(2); PPS_LAUNCH(kb_level_factor2, dim3(g.lvl_cls_blocks[l][0], a.n, nz), dim3(256), (size_t)pw * 4 * sizeof(double), st, a, l, pw); }
      if (g.lvl_cls_blocks[l][1] > 0) { const int pw = level_lds_doubles(3); PPS_LAUNCH(kb_level_factor3, dim3(g.lvl_cls_blocks[l][1], a.n, nz), dim3(256), (size_t)pw * 4 * sizeof(double), st, a, l, pw); }
      if (g.lvl_cls_blocks[l][2] > 0) { const int pw = level_lds_doubles(4); PPS_LAUNCH(kb_level_factor4, dim3(g.lvl_cls_blocks[l][2], a.n, nz), dim3(256), (size_t)pw * 4 * sizeof(double), st, a, l, pw); }
    }
    if (after_factor) (void)hipEventRecord(after_factor, st);
    for (int l = g.n_levels - 1; l >= 0; l--)
      if (g.lvl_blocks[l] > 0) {
        // (LDS by the level's own largest panel: the separators above the leaves hold a third of a leaf's panel -- five waves per SIMD instead of three)
        int pw = std::min(g.solve_per_wave_all, (int)(band_solve_lds_bytes(g.lvl_max_panel[l]) / sizeof(double))) + kLevelSolveSlack;
        // ... and by x_b + the pivot block alone where every front of the level takes the direct-load form (lvl_direct_pp: its largest p x p)
        const int direct = g.lvl_direct_pp[l] > 0 ? 1 : 0;
        if (direct) pw = std::min(pw, kBandMaxRows + g.lvl_direct_pp[l] + kLevelSolveSlack);
        PPS_LAUNCH(kb_level_solve, dim3(g.lvl_blocks[l], a.n, nz), dim3(256), (size_t)pw * 4 * sizeof(double), st, a, l, pw, direct);
      }
    return hipGetLastError();
  }
  for (int stg = 0; stg < g.n_stages; stg++) {
    if (g.stage_groups[stg] <= 0) continue;
    const int per_wave = g.stage_per_wave_factor[stg], nw = g.stage_nw_factor[stg];
    const size_t bytes = (size_t)per_wave * nw * sizeof(double);
    if (g.stage_reg_only[stg] && g.stage_pre[stg])
      PPS_LAUNCH(kb_band_factor_pre, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
    else if (g.stage_reg_only[stg])
      PPS_LAUNCH(kb_band_factor<true>, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
    else
      PPS_LAUNCH(kb_band_factor<false>, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
  }
  if (after_factor) (void)hipEventRecord(after_factor, st);
  for (int stg = g.n_stages - 1; stg >= 0; stg--) {
    if (g.stage_groups[stg] <= 0) continue;
    const int per_wave = g.stage_per_wave_solve[stg], nw = g.stage_nw_solve[stg];
    const size_t bytes = ((size_t)per_wave * nw + (size_t)g.stage_grp_fronts[stg] * kBandMaxRows) * sizeof(double);
    if (g.stage_nw_flow[stg] > 0) {
      const int nwf = g.stage_nw_flow[stg], mg = g.stage_grp_fronts[stg];
      const size_t fb = ((size_t)per_wave * nwf + (size_t)mg * kBandMaxRows + (size_t)(mg + 1) / 2) * sizeof(double);
      PPS_LAUNCH(kb_band_solve_flow, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nwf), fb, st, a, stg, per_wave, mg);
      continue;
    }
    PPS_LAUNCH(kb_band_solve, dim3(g.stage_groups[stg], a.n, a.alt ? 2 : 1), dim3(64 * nw), bytes, st, a, stg, per_wave);
  }
  return hipGetLastError();
}

}  // namespace pps

// ---- diagnostic: one front through the register-tile elimination, outside any graph (pps_debug_front_factor) ----
namespace pps {
template <int NT, bool STRIP>
__global__ __launch_bounds__(64) void k_debug_front(DevGraph d, int p, int b, const double* A, int per_wave) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int fa = p + b + 1;
  double* F = lds;
  // the layouts of the production kernels: without a strip the panel buffer is the head of the triangle (register-only and level
  // kernels), with a strip / a fifth tile row it sits behind triangle and spare double (general and r5 kernels)
  double* P = STRIP ? F + per_wave - kRegRowsMax * kP8Stride : F;
  for (int i = lane; i < fa * (fa + 1) / 2; i += 64) F[i] = A[i];
  __builtin_amdgcn_wave_barrier();
  int rec = 0;                                    // (L and U of the front start at offset 0)
  if (lane == 1) rec = p;
  if (lane == 2) rec = b;
  front_reg_eliminate<NT, false, STRIP>(d, rec, F, P);
}

// tiles: 2 .. 5 tile rows; 0 = the choice the band kernels make for the front (by its row count); strip: rows 64 .. 79 as a strip of
// the LDS triangle under four tile rows instead of a fifth tile row.  Returns a hipError_t (0 = ok), -1 for arguments no path takes.
int debug_front_factor(int tiles, int strip, int p, int b, const double* A_host, double* L_host, double* U_host, double* not_pd) {
  const int f = p + b, fa = f + 1;
  if (p < 1 || b < 0 || fa > kRegRowsMax || p > kRegRows) return -1;
  if (tiles == 0) tiles = fa <= 33 ? 2 : fa <= 49 ? 3 : fa <= kRegRows ? 4 : (strip ? 4 : 5);
  const bool wide = fa > kRegRows;
  if (tiles < 2 || tiles > 5 || (wide && tiles < 4) || (!wide && (tiles == 5 || strip)) || (tiles < 4 && f > 16 * tiles) || (wide && tiles == 4 && !strip)) return -1;
  const size_t ntri = (size_t)fa * (fa + 1) / 2, nL = (size_t)fa * p, nU = (size_t)(b + 1) * (b + 1);
  double *A = nullptr, *L = nullptr, *U = nullptr, *res = nullptr;
  hipError_t e = hipMalloc(&A, ntri * 8);
  if (e == hipSuccess) e = hipMalloc(&L, nL * 8);
  if (e == hipSuccess) e = hipMalloc(&U, nU * 8);
  if (e == hipSuccess) e = hipMalloc(&res, 32);
  if (e == hipSuccess) e = hipMemcpy(A, A_host, ntri * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(L, 0, nL * 8);
  if (e == hipSuccess) e = hipMemset(U, 0, nU * 8);
  if (e == hipSuccess) e = hipMemset(res, 0, 32);
  if (e == hipSuccess) {
    DevGraph d{}; d.L = L; d.U = U; d.result_dev = res;
    const int per_wave = (int)(band_lds_bytes(fa - 1, !(tiles == 5 || wide)) / 8);
    const size_t bytes = (size_t)per_wave * 8;
#define PPS_DBG_FRONT(NT_, STRIP_)                                                                                                      \
    do {                                                                                                                                \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_debug_front<NT_, STRIP_>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimitBytes); \
      if (e == hipSuccess) { hipLaunchKernelGGL((k_debug_front<NT_, STRIP_>), dim3(1), dim3(64), bytes, 0, d, p, b, A, per_wave); e = hipGetLastError(); } \
    } while (0)
    if (tiles == 5) PPS_DBG_FRONT(5, true);
    else if (tiles == 4 && wide) PPS_DBG_FRONT(4, true);
    else if (tiles == 4) PPS_DBG_FRONT(4, false);
    else if (tiles == 3) PPS_DBG_FRONT(3, false);
    else PPS_DBG_FRONT(2, false);
#undef PPS_DBG_FRONT
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(L_host, L, nL * 8, hipMemcpyDeviceToHost);
    // (the device slab is (b+1)^2 doubles -- its last one is the kernel's trash slot --, the caller's buffer the packed triangle)
    if (e == hipSuccess) e = hipMemcpy(U_host, U, (size_t)(b + 1) * (b + 2) / 2 * 8, hipMemcpyDeviceToHost);
    double r4[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(r4, res, 32, hipMemcpyDeviceToHost);
    if (not_pd) *not_pd = r4[2];
  }
  (void)hipFree(A); (void)hipFree(L); (void)hipFree(U); (void)hipFree(res);
  return (int)e;
}
}  // namespace pps
