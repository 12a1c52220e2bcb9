// pps_k2.hip -- K2: block-sparse J'J / J'b reduction (cholmod_ssmult / cholmod_sdmult, isamlib/Cholesky.cpp:87-89,120).
#include "pps_kcommon.h"

namespace pps {

// ------------------------------------------------------------------------------------------
// K2: one wavefront per H-block segment; lane = block entry, loop over <= seg_len contributions.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void body_hblocks(const DevGraph& d, int bx) {
  const int seg = uni(bx * 4 + (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (seg >= d.n_segs) return;
  // one coalesced load of the packed segment record, fields broadcast with v_readlane
  const int rec = d.srec[(size_t)seg * 8 + (lane & 7)];
  const int rows = __builtin_amdgcn_readlane(rec, 0), cols = __builtin_amdgcn_readlane(rec, 1), size = __builtin_amdgcn_readlane(rec, 2);
  const int c0 = __builtin_amdgcn_readlane(rec, 3), cnt = __builtin_amdgcn_readlane(rec, 4);
  const int hoff = __builtin_amdgcn_readlane(rec, 5), doff = __builtin_amdgcn_readlane(rec, 6), nsegb = __builtin_amdgcn_readlane(rec, 7);
  // one contribution descriptor per lane, fetched in a single coalesced load (cnt <= 64)
  int4 mine = make_int4(0, 0, 0, 0);
  if (lane < cnt) mine = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];
  const int rc = rows * cols;
  const bool act = lane < size;
  // where the finished entry goes in front-gather order: does not depend on the values, so the load is issued now
  const int dst = (act && nsegb == 1) ? d.blk_dst[doff + lane] : -1;
  const bool is_g = lane >= rc;
  const int i = is_g ? lane - rc : lane / cols;
  const int j = is_g ? 0 : lane - (lane / cols) * cols;
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  int c = 0;
  for (; c + 2 <= cnt; c += 2) {                          // two contributions' loads in flight (64 VGPRs: 8 waves per SIMD)
    double a[2][6], bb[2][6];
    int m[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int cc = c + u;
      const int jv = __builtin_amdgcn_readlane(mine.x, cc), ju = __builtin_amdgcn_readlane(mine.y, cc);
      const int ro = __builtin_amdgcn_readlane(mine.z, cc);
      m[u] = __builtin_amdgcn_readlane(mine.w, cc);
      const double* pa = J + jv + i;
      const double* pb = is_g ? J + ro : J + ju + j;
      const int sb = is_g ? 1 : cols;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const bool ok = act && k < m[u];
        a[u][k] = ok ? pa[k * rows] : 0.0;
        bb[u][k] = ok ? pb[k * sb] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[u][k], bb[u][k]);
  }
  for (; c < cnt; c++) {                                  // tail, and the single-contribution segments (most pose-plane blocks)
    const int jv = __builtin_amdgcn_readlane(mine.x, c), ju = __builtin_amdgcn_readlane(mine.y, c);
    const int ro = __builtin_amdgcn_readlane(mine.z, c), mm = __builtin_amdgcn_readlane(mine.w, c);
    const double* pa = J + jv + i;
    const double* pb = is_g ? J + ro : J + ju + j;
    const int sb = is_g ? 1 : cols;
    double a[6], bb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const bool ok = act && k < mm;
      a[k] = ok ? pa[k * rows] : 0.0;
      bb[k] = ok ? pb[k * sb] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[k], bb[k]);
  }
  if (!act) return;
  if (is_g) acc = -acc;                                   // b = -r (isam/Jacobian.h:98)
  d.H[hoff + lane] = acc;
  if (dst >= 0) d.Hf[dst] = acc;                          // final value (single-segment block): also where its front gathers it
}

__global__ __launch_bounds__(256, 2) void k_hblocks(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hblocks(d, blockIdx.x); }

// Throughput form (many graphs per launch): a wave takes S consecutive segments.  The three dependent round trips of a
// segment -- record, contribution descriptors, Jacobian slices -- are each issued for all S segments before the first
// answer is needed; the sums run in the order of body_hblocks (contribution by contribution, k ascending), bit for bit.
template <int S>
__device__ __forceinline__ void body_hblocks_t(const DevGraph& d, int bx) {
  const int slot0 = uni((bx * 4 + (threadIdx.x >> 6)) * S);        // position in the list of non-direct segments
  const int lane = threadIdx.x & 63;
  if (slot0 >= d.n_nd_segs) return;
  int rec[S];
  {
    const int sidx = (lane >> 3) < S && slot0 + (lane >> 3) < d.n_nd_segs ? d.nd_segs[slot0 + (lane >> 3)] : -1;   // lanes 8q..8q+7: segment q
#pragma unroll
    for (int q = 0; q < S; q++) {
      const int sg = __builtin_amdgcn_readlane(sidx, 8 * q);
      rec[q] = sg >= 0 ? d.srec[(size_t)sg * 8 + (lane & 7)] : 0;
    }
  }
  int rows[S], cols[S], size[S], cnt[S], hoff[S], dst[S], ii[S], jj[S];
  bool act[S], isg[S];
  int4 mine[S];
#pragma unroll
  for (int q = 0; q < S; q++) {
    rows[q] = __builtin_amdgcn_readlane(rec[q], 0); cols[q] = __builtin_amdgcn_readlane(rec[q], 1); size[q] = __builtin_amdgcn_readlane(rec[q], 2);
    const int c0 = __builtin_amdgcn_readlane(rec[q], 3);
    cnt[q] = __builtin_amdgcn_readlane(rec[q], 4); hoff[q] = __builtin_amdgcn_readlane(rec[q], 5);
    const int doff = __builtin_amdgcn_readlane(rec[q], 6), nsegb = __builtin_amdgcn_readlane(rec[q], 7);
    mine[q] = make_int4(0, 0, 0, 0);
    if (lane < cnt[q]) mine[q] = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];
    act[q] = lane < size[q];                                  // size 0 for a segment past the end
    dst[q] = (act[q] && nsegb == 1) ? d.blk_dst[doff + lane] : -1;
    const int rc = rows[q] * cols[q];
    isg[q] = lane >= rc;
    const int cq = cols[q] > 0 ? cols[q] : 1;
    ii[q] = isg[q] ? lane - rc : lane / cq;
    jj[q] = isg[q] ? 0 : lane - (lane / cq) * cq;
  }
  const double* __restrict__ J = d.J;
  double a[S][6], bb[S][6];
#pragma unroll
  for (int q = 0; q < S; q++) {                               // first contribution of every segment: all loads in flight together
    const int jv = __builtin_amdgcn_readlane(mine[q].x, 0), ju = __builtin_amdgcn_readlane(mine[q].y, 0);
    const int ro = __builtin_amdgcn_readlane(mine[q].z, 0), mm = cnt[q] > 0 ? __builtin_amdgcn_readlane(mine[q].w, 0) : 0;
    const double* pa = J + jv + ii[q];
    const double* pb = isg[q] ? J + ro : J + ju + jj[q];
    const int sb = isg[q] ? 1 : cols[q];
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const bool ok = act[q] && k < mm;
      a[q][k] = ok ? pa[k * rows[q]] : 0.0;
      bb[q][k] = ok ? pb[k * sb] : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < S; q++) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[q][k], bb[q][k]);
    for (int c = 1; c < cnt[q]; c++) {                        // further contributions (diagonal blocks)
      const int jv = __builtin_amdgcn_readlane(mine[q].x, c), ju = __builtin_amdgcn_readlane(mine[q].y, c);
      const int ro = __builtin_amdgcn_readlane(mine[q].z, c), mm = __builtin_amdgcn_readlane(mine[q].w, c);
      const double* pa = J + jv + ii[q];
      const double* pb = isg[q] ? J + ro : J + ju + jj[q];
      const int sb = isg[q] ? 1 : cols[q];
      double a2[6], b2[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const bool ok = act[q] && k < mm;
        a2[k] = ok ? pa[k * rows[q]] : 0.0;
        b2[k] = ok ? pb[k * sb] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a2[k], b2[k]);
    }
    if (act[q]) {
      if (isg[q]) acc = -acc;                                 // b = -r (isam/Jacobian.h:98)
      d.H[hoff[q] + lane] = acc;
      if (dst[q] >= 0) d.Hf[dst[q]] = acc;
    }
  }
}

// fold the partial sums of multi-segment blocks (the ground plane's diagonal) into their first slot
__device__ __forceinline__ void body_hreduce(const DevGraph& d, int bx) {
  const int blk = d.mseg_blk[bx];
  const int size = d.blk_size[blk], nseg = d.blk_nseg[blk];
  double* __restrict__ h = d.H + d.blk_hoff[blk];
  const int lane = threadIdx.x;
  if (lane >= size) return;
  double v = 0.0;
  for (int q = 0; q < nseg; q += 16) {
    double x[16];
#pragma unroll
    for (int u = 0; u < 16; u++) x[u] = (q + u < nseg) ? h[(size_t)(q + u) * size + lane] : 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++) v += x[u];
  }
  h[lane] = v;
  const int dst = d.blk_dst[d.blk_doff[blk] + lane];
  if (dst >= 0) d.Hf[dst] = v;
}

__global__ __launch_bounds__(64) void k_hreduce(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hreduce(d, blockIdx.x); }

hipError_t launch_hblocks(const DevGraph& d, hipStream_t st, const LinGuard* guard) {
  if (d.n_segs == 0) return hipSuccess;
  const LinGuard gd = guard ? *guard : LinGuard{};
  PPS_LAUNCH(k_hblocks, dim3(cdiv(d.n_segs, 4)), dim3(256), 0, st, d, gd);
  if (d.n_mseg > 0) PPS_LAUNCH(k_hreduce, dim3(d.n_mseg), dim3(64), 0, st, d, gd);
  return hipGetLastError();
}

// ---- batched forms ----
__global__ __launch_bounds__(256, 2) void kb_hblocks(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 4 >= d.n_segs) return;
  body_hblocks(d, blockIdx.x);
}

constexpr int kHblocksT = 4;      // segments per wave of the throughput form
__global__ __launch_bounds__(256) void kb_hblocks_t(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 4 * kHblocksT >= d.n_nd_segs) return;
  body_hblocks_t<kHblocksT>(d, blockIdx.x);
}

__global__ __launch_bounds__(64) void kb_hreduce(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= d.n_mseg) return;
  body_hreduce(d, blockIdx.x);
}

hipError_t launch_batch_hblocks(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.hblocks > 0) {
    if (g.k1_direct) PPS_LAUNCH(kb_hblocks_t, dim3(std::max(1, g.hblocks_nd), a.n), dim3(256), 0, st, a);   // many graphs: throughput form, direct blocks done by K1
    else PPS_LAUNCH(kb_hblocks, dim3(g.hblocks, a.n), dim3(256), 0, st, a);
  }
  if (g.hreduce > 0) PPS_LAUNCH(kb_hreduce, dim3(g.hreduce, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps
