// pps_k2.hip -- K2: block-sparse J'J / J'b reduction (cholmod_ssmult / cholmod_sdmult, isamlib/Cholesky.cpp:87-89,120).
#include "pps_kcommon.h"

namespace pps {

// ------------------------------------------------------------------------------------------
// K2, latency form (one graph): one wavefront per H-block segment; lane = block entry, loop over <= seg_len contributions,
// every lane fetching its own scalars (two contributions' loads in flight).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void body_hblocks(const DevGraph& d, int bx) {
  const int slot = uni(bx * 4 + (threadIdx.x >> 6));        // position in the list of segments K1 has not written itself
  const int lane = threadIdx.x & 63;
  if (slot >= d.n_nd_segs) return;
  const int seg = uni(d.nd_segs[slot]);
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

// ------------------------------------------------------------------------------------------
// K2, throughput form (many graphs per launch).  The Jacobian slices of a segment's contributions are staged in LDS first --
// three coalesced loads per contribution (the row node's block, the column node's block, the residual), a chunk of eight
// contributions in flight at once -- and the products read LDS.  In the latency form every J element is fetched a dozen times
// over by different lanes, which makes a large batch bound by the texture-address path (kb_hblocks_t: 21.0 -> 14.9 ms per
// G = 128 batch solve); on a single graph the extra LDS round trip costs more than it saves (C3: 87.6 -> 96.9 us), so the
// latency form stays there.  Sums run contribution by contribution, k ascending, as explicit multiply-adds: same bits.
// ------------------------------------------------------------------------------------------
constexpr int kH2Slots = 8;          // contributions staged per chunk
constexpr int kH2Stride = 80;        // doubles per slot: [row block <= 36 | column block <= 36 | residual <= 6 | pad]
constexpr int kH2WaveDoubles = kH2Slots * kH2Stride;

// what a wave knows about a segment before it touches the Jacobians: the packed record, then its contribution descriptors and
// the front-order destination of its entries -- two dependent loads, requested for ALL segments of the wave before the first
// segment is worked on
struct SegHdr { int rec; int4 mine; int dst; };
__device__ __forceinline__ void seg_fetch_record(const DevGraph& d, int seg, int lane, SegHdr& h) {
  h.rec = seg >= 0 ? d.srec[(size_t)seg * 8 + (lane & 7)] : 0;      // (size 0, cnt 0 for a slot past the end of the list)
}
__device__ __forceinline__ void seg_fetch_contrib(const DevGraph& d, int lane, SegHdr& h) {
  const int size = __builtin_amdgcn_readlane(h.rec, 2), c0 = __builtin_amdgcn_readlane(h.rec, 3), cnt = __builtin_amdgcn_readlane(h.rec, 4);
  const int doff = __builtin_amdgcn_readlane(h.rec, 6), nsegb = __builtin_amdgcn_readlane(h.rec, 7);
  // one contribution descriptor per lane, fetched in a single coalesced load (cnt <= 64)
  h.mine = make_int4(0, 0, 0, 0);
  if (lane < cnt) h.mine = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];
  // where the finished entry goes in front-gather order: does not depend on the values, so the load is issued now
  h.dst = (lane < size && nsegb == 1) ? d.blk_dst[doff + lane] : -1;
}

__device__ __forceinline__ void wave_hblock_segment(const DevGraph& d, const SegHdr& h, double* __restrict__ S) {
  const int lane = threadIdx.x & 63;
  const int rec = h.rec;
  const int4 mine = h.mine;
  const int dst = h.dst;
  const int rows = __builtin_amdgcn_readlane(rec, 0), cols = __builtin_amdgcn_readlane(rec, 1), size = __builtin_amdgcn_readlane(rec, 2);
  const int cnt = __builtin_amdgcn_readlane(rec, 4);
  const int hoff = __builtin_amdgcn_readlane(rec, 5);
  if (size == 0) return;
  const int rc = rows * cols;
  const bool act = lane < size;
  const bool is_g = lane >= rc;
  const int cdiv_ = cols > 0 ? cols : 1;
  const int i = is_g ? lane - rc : lane / cdiv_;
  const int j = is_g ? 0 : lane - (lane / cdiv_) * cdiv_;
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  for (int cb = 0; cb < cnt; cb += kH2Slots) {
    const int nc = cnt - cb < kH2Slots ? cnt - cb : kH2Slots;
    // All loads of the chunk are issued before the first LDS write (a rolled loop would wait for every contribution's data
    // before requesting the next one's).  Lanes past a slice repeat its last element and slots past the chunk repeat its last
    // contribution -- the same value to the same LDS word, or to a slot nobody reads -- so nothing is predicated (skipping the
    // unused slots by wave-uniform branches measured slower: 18.1 against 14.9 ms of K2 per G = 128 batch solve).
    if (nc == 1) {                                             // one contribution (every pose-pose block): one slot, not eight
      const int jv = __builtin_amdgcn_readlane(mine.x, cb), ju = __builtin_amdgcn_readlane(mine.y, cb);
      const int ro = __builtin_amdgcn_readlane(mine.z, cb), m = __builtin_amdgcn_readlane(mine.w, cb);
      const int nv = m * rows - 1, nu = m * cols - 1, nr = m - 1;
      const int lv1 = lane < nv ? lane : nv, lu1 = lane < nu ? lane : nu, lr1 = lane < nr ? lane : nr;
      const double xv1 = J[jv + lv1], xu1 = J[ju + lu1], xr1 = J[ro + lr1];
      S[lv1] = xv1; S[36 + lu1] = xu1; S[72 + lr1] = xr1;
    } else {
    double xv[kH2Slots], xu[kH2Slots], xr[kH2Slots];
    int lv[kH2Slots], lu[kH2Slots], lr[kH2Slots];
#pragma unroll
    for (int u = 0; u < kH2Slots; u++) {
      const int cu = cb + (u < nc ? u : nc - 1);
      const int jv = __builtin_amdgcn_readlane(mine.x, cu), ju = __builtin_amdgcn_readlane(mine.y, cu);
      const int ro = __builtin_amdgcn_readlane(mine.z, cu), m = __builtin_amdgcn_readlane(mine.w, cu);
      const int nv = m * rows - 1, nu = m * cols - 1, nr = m - 1;
      lv[u] = lane < nv ? lane : nv; lu[u] = lane < nu ? lane : nu; lr[u] = lane < nr ? lane : nr;
      xv[u] = J[jv + lv[u]]; xu[u] = J[ju + lu[u]]; xr[u] = J[ro + lr[u]];
    }
#pragma unroll
    for (int u = 0; u < kH2Slots; u++) {
      double* __restrict__ slot = S + u * kH2Stride;
      slot[lv[u]] = xv[u]; slot[36 + lu[u]] = xu[u]; slot[72 + lr[u]] = xr[u];
    }
    }
    __builtin_amdgcn_wave_barrier();
    for (int u = 0; u < nc; u++) {
      const int m = __builtin_amdgcn_readlane(mine.w, cb + u);
      const double* __restrict__ slot = S + u * kH2Stride;
      const double* __restrict__ pa = slot + i;
      const double* __restrict__ pb = is_g ? slot + 72 : slot + 36 + j;
      const int sb = is_g ? 1 : cols;
#pragma unroll
      for (int k = 0; k < 3; k++) acc = PPS_MAC(acc, pa[k * rows], pb[k * sb]);
      if (m > 3) {
#pragma unroll
        for (int k = 3; k < 6; k++) acc = PPS_MAC(acc, pa[k * rows], pb[k * sb]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!act) return;
  if (is_g) acc = -acc;                                   // b = -r (isam/Jacobian.h:98)
  d.H[hoff + lane] = acc;
  if (dst >= 0) d.Hf[dst] = acc;                          // final value (single-segment block): also where its front gathers it
}

// A wave takes S consecutive segments of the list of non-direct segments (the single-observation pose-plane blocks are written
// by K1 itself in this mode).
template <int S>
__device__ __forceinline__ void body_hblocks_t(const DevGraph& d, int bx) {
  __shared__ double h2_lds[4 * kH2WaveDoubles];
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int slot0 = uni((bx * 4 + wave) * S);        // position in the list of non-direct segments
  if (slot0 >= d.n_nd_segs) return;
  SegHdr h[S];
  {
    const int sidx = (lane >> 3) < S && slot0 + (lane >> 3) < d.n_nd_segs ? d.nd_segs[slot0 + (lane >> 3)] : -1;   // lanes 8q..8q+7: segment q
#pragma unroll
    for (int q = 0; q < S; q++) seg_fetch_record(d, __builtin_amdgcn_readlane(sidx, 8 * q), lane, h[q]);
  }
#pragma unroll
  for (int q = 0; q < S; q++) seg_fetch_contrib(d, lane, h[q]);
#pragma unroll
  for (int q = 0; q < S; q++) wave_hblock_segment(d, h[q], h2_lds + wave * kH2WaveDoubles);
}

// Fold the partial sums of a multi-segment block (the ground plane's diagonal: a hundred segments on C2) into its first slot.
// The segments are summed as four interleaved partial sums -- p_w = segments w, w + 4, w + 8, ... in order -- combined as
// (p0 + p1) + (p2 + p3): the same bits in every form.  NW = 4 (one graph): a 256-thread workgroup per block, one partial sum
// per wave, up to 32 independent loads per thread in flight -- two memory round trips for the ground plane instead of one per
// 16 segments -- and the combination through LDS.  NW = 1 (batches, where the launch is wide anyway): one wave computes the four
// partial sums one after the other.
template <int NW>
__device__ __forceinline__ void body_hreduce(const DevGraph& d, int bx) {
  __shared__ double part[4][64];
  const int blk = d.mseg_blk[bx];
  const int size = d.blk_size[blk], nseg = d.blk_nseg[blk];
  double* __restrict__ h = d.H + d.blk_hoff[blk];
  const int lane = threadIdx.x & 63, w0 = NW == 4 ? (int)(threadIdx.x >> 6) : 0;
  const int ln = lane < size ? lane : 0;                       // (idle lanes shadow entry 0: no predicated loads)
  double pv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int w = 0; w < 4; w++) {
    if (NW == 4 && w != w0) continue;
    double v = 0.0;
    if (nseg <= 4) {                                           // an ordinary landmark: one segment per partial sum
      const double t = h[(size_t)(w < nseg ? w : 0) * size + ln];
      pv[w] = w < nseg ? t : 0.0;
      continue;
    }
    for (int q = w; q < nseg; q += 4 * 16) {
      double x[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int sg = q + 4 * u; const double t = h[(size_t)(sg < nseg ? sg : 0) * size + ln]; x[u] = sg < nseg ? t : 0.0; }
#pragma unroll
      for (int u = 0; u < 16; u++) v += x[u];
    }
    pv[w] = v;
  }
  if (NW == 4) {
    part[w0][lane] = pv[0] + pv[1] + pv[2] + pv[3];            // (three of them are zero: this wave's partial sum, exactly)
    __syncthreads();
    if (w0 != 0) return;
#pragma unroll
    for (int w = 0; w < 4; w++) pv[w] = part[w][lane];
  }
  if (lane >= size) return;
  const double tot = (pv[0] + pv[1]) + (pv[2] + pv[3]);
  h[lane] = tot;
  const int dst = d.blk_dst[d.blk_doff[blk] + lane];
  if (dst >= 0) d.Hf[dst] = tot;
}

__global__ __launch_bounds__(256) void k_hreduce(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hreduce<4>(d, blockIdx.x); }

hipError_t launch_hblocks(const DevGraph& d, hipStream_t st, const LinGuard* guard) {
  if (d.n_segs == 0) return hipSuccess;
  const LinGuard gd = guard ? *guard : LinGuard{};
  if (d.n_nd_segs > 0) PPS_LAUNCH(k_hblocks, dim3(cdiv(d.n_nd_segs, 4)), dim3(256), 0, st, d, gd);
  if (d.n_mseg > 0) PPS_LAUNCH(k_hreduce, dim3(d.n_mseg), dim3(256), 0, st, d, gd);
  return hipGetLastError();
}

// ---- batched forms ----
__global__ __launch_bounds__(256, 2) void kb_hblocks(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x * 4 >= d.n_nd_segs) return;
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
  body_hreduce<1>(d, blockIdx.x);
}

hipError_t launch_batch_hblocks(const BatchArgs& a, const BatchGeom& g, hipStream_t st) {
  if (g.hblocks > 0) {
    if (g.lin_thread_form) PPS_LAUNCH(kb_hblocks_t, dim3(std::max(1, g.hblocks_nd), a.n), dim3(256), 0, st, a);   // many graphs: throughput form
    else PPS_LAUNCH(kb_hblocks, dim3(std::max(1, g.hblocks), a.n), dim3(256), 0, st, a);
  }
  if (g.hreduce > 0) PPS_LAUNCH(kb_hreduce, dim3(g.hreduce, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps
