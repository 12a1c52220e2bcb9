// pps_k2.hip -- K2: block-sparse J'J / J'b reduction (cholmod_ssmult / cholmod_sdmult, isamlib/Cholesky.cpp:87-89,120).
#include <cstdlib>

#include "pps_kcommon.h"
#include "pps_symbolic.h"

namespace pps {

// ------------------------------------------------------------------------------------------
// K2 (round 4): one wavefront per H-block segment, lane = block entry, up to 64 contributions per segment.
//  * A plain plane observation -- five of every six factors -- reaches K2 as its PRODUCT record: K1 has multiplied J' J and
//    -J' r for the two diagonal blocks it feeds (pps_k1_body.h), so a contribution is ONE coalesced load per lane and a wave keeps
//    eight of them in flight.  (Up to round 3 every contribution was twelve strided gathers from the Jacobian: the kernel was
//    bound by the texture-address path -- as long as K1 itself on C3.)
//  * The other contributions (odometry, priors, re-popping edges, off-diagonal blocks K1 does not write itself) multiply the
//    Jacobian slices as before.
//  * A block of several segments (the ground plane: a thousand observations) is one workgroup: its sixteen waves take the
//    segments w, w + 16, ... and the partial sums meet in LDS in a fixed order -- no second launch (k_hreduce is gone).
// Sums are deterministic: Jacobian-based contributions first, in list order, as explicit multiply-adds; then the product records
// in list order; partial sums of a block by wave index.  The same body serves one graph and the batches: same bits.
// ------------------------------------------------------------------------------------------
constexpr int kK2Waves = 16;    // waves per workgroup

// plist: 64 ints of LDS of this wave (the offsets of the segment's product records, compacted)
__device__ __forceinline__ double wave_segment_sum(const DevGraph& d, int rec, int4 mine, int lane, int* __restrict__ plist) {
  const int rows = __builtin_amdgcn_readlane(rec, 0), cols = __builtin_amdgcn_readlane(rec, 1), size = __builtin_amdgcn_readlane(rec, 2);
  const int cnt = __builtin_amdgcn_readlane(rec, 4);
  const int rc = rows * cols;
  const bool act = lane < size;
  const bool is_g = lane >= rc;
  const int cdiv_ = cols > 0 ? cols : 1;
  const int i = is_g ? lane - rc : lane / cdiv_;
  const int j = is_g ? 0 : lane - (lane / cdiv_) * cdiv_;
  const bool isp = lane < cnt && mine.w >= kProductFlag;
  const unsigned long long pmask = __ballot(isp);
  unsigned long long jmask = __ballot(lane < cnt && !isp);
  const double* __restrict__ J = d.J;
  const double* __restrict__ P = d.P;
  double acc = 0.0;
  // ---- product records, first batch: requested BEFORE the Jacobian-based contributions are worked on, so that a pose block's
  // odometry slices and its observations' records are one memory round trip.  One coalesced load per contribution and lane; a small
  // block (a plane's 12 entries) is summed by several SLICES of the wave at once -- slice s takes the records s, s + S, ... -- and
  // the slices' sums are added in slice order. ----
  const int np = __builtin_popcountll(pmask);
  const int S = size <= 12 ? 5 : (size <= 21 ? 3 : (size <= 32 ? 2 : 1));
  const int sl = lane / (size > 0 ? size : 1), en = lane - sl * size;
  const bool live = sl < S && size > 0;
  const int enc = live ? en : 0;
  double v0[16];
  if (np > 0) {                                                 // (wave-uniform)
    const int rank = __builtin_popcountll(pmask & ((1ull << lane) - 1ull));
    if (isp) plist[rank] = mine.y;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 16; u++) { const int k = sl + S * u; v0[u] = P[plist[k < np ? k : np - 1] + enc]; }
  }
  // ---- contributions that come as Jacobian slices (odometry, priors, re-popping edges, off-diagonal blocks): two in flight ----
  while (jmask) {                                               // (wave-uniform)
    const int c0 = __builtin_ctzll(jmask);
    jmask &= jmask - 1;
    const bool two = jmask != 0;
    const int c1 = two ? __builtin_ctzll(jmask) : c0;
    jmask &= jmask - 1;                                         // (0 stays 0)
    double a[2][6], bb[2][6];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int cc = u ? c1 : c0;
      const int jv = __builtin_amdgcn_readlane(mine.x, cc), ju = __builtin_amdgcn_readlane(mine.y, cc);
      const int ro = __builtin_amdgcn_readlane(mine.z, cc), mm = __builtin_amdgcn_readlane(mine.w, cc);
      const double* pa = J + jv + i;
      const double* pb = is_g ? J + ro : J + ju + j;
      const int sb = is_g ? 1 : cols;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const bool ok = act && k < mm && (u == 0 || two);
        a[u][k] = ok ? pa[k * rows] : 0.0;
        bb[u][k] = ok ? pb[k * sb] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, a[u][k], bb[u][k]);
  }
  if (is_g) acc = -acc;                                         // b = -r (isam/Jacobian.h:98); the product records carry the sign already
  if (!act) acc = 0.0;
  if (np > 0) {
    double pacc = 0.0;
#pragma unroll
    for (int u = 0; u < 16; u++) pacc += (live && sl + S * u < np) ? v0[u] : 0.0;
    for (int k0 = 16 * S; k0 < np; k0 += 16 * S) {              // (more than 16 records per slice: the ground plane's segments)
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int k = k0 + sl + S * u; v[u] = P[plist[k < np ? k : np - 1] + enc]; }
#pragma unroll
      for (int u = 0; u < 16; u++) pacc += (live && k0 + sl + S * u < np) ? v[u] : 0.0;
    }
    __builtin_amdgcn_wave_barrier();                            // (plist is rewritten by the wave's next segment)
    double tot = pacc;
    for (int sg = 1; sg < S; sg++) tot += __shfl(pacc, lane + sg * size, 64);      // (lanes < size: slice sg of the same entry)
    acc += act ? tot : 0.0;
  }
  return acc;
}

__device__ __forceinline__ void seg_header(const DevGraph& d, int seg, int lane, int& rec, int4& mine) {
  rec = d.srec[(size_t)seg * 8 + (lane & 7)];                   // one coalesced load of the packed segment record
  const int c0 = __builtin_amdgcn_readlane(rec, 3), cnt = __builtin_amdgcn_readlane(rec, 4);
  mine = make_int4(0, 0, 0, 0);
  if (lane < cnt) mine = reinterpret_cast<const int4*>(d.contrib)[c0 + lane];      // one contribution descriptor per lane (cnt <= 64)
  // (no product records -- the thread-per-factor K1 writes none: every contribution multiplies its Jacobian slices)
  if (!d.P && mine.w >= kProductFlag) { mine.w -= kProductFlag; mine.y = mine.x; }
}

__device__ __forceinline__ void body_hblocks2(const DevGraph& d, int bx) {
  __shared__ double part[kK2Waves][64];
  __shared__ int plist_lds[kK2Waves][64];
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  int* const plist = plist_lds[wave];
  const int nbs = (d.n_k2_single + kK2Waves - 1) / kK2Waves;
  if (bx < nbs) {
    const int slot = bx * kK2Waves + wave;
    if (slot >= d.n_k2_single) return;
    int rec; int4 mine;
    seg_header(d, uni(d.k2_single[slot]), lane, rec, mine);
    const int size = __builtin_amdgcn_readlane(rec, 2), hoff = __builtin_amdgcn_readlane(rec, 5), doff = __builtin_amdgcn_readlane(rec, 6);
    // where the finished entry goes in front-gather order: does not depend on the values, so the load is issued now
    const int dst = lane < size ? d.blk_dst[doff + lane] : -1;
    const double acc = wave_segment_sum(d, rec, mine, lane, plist);
    if (lane >= size) return;
    d.H[hoff + lane] = acc;
    if (dst >= 0) d.Hf[dst] = acc;                              // also where the block's front gathers it
    return;
  }
  // a block of several segments, sixteen at a time: the chunk's waves take one segment each, the partial sums meet in LDS in wave
  // order.  A block of at most sixteen segments (the ground plane of a thousand-pose graph) is finished here; a larger one leaves
  // one partial sum per chunk in the slot of the chunk's first segment and k_hfinish adds those up.
  const int b = bx - nbs;
  if (b >= d.n_k2_multi) return;
  const int seg0 = uni(d.k2_multi[2 * b]), info = uni(d.k2_multi[2 * b + 1]);
  const int nsc = info & 0xffff;
  const bool fin = (info >> 16) != 0;
  int rec0; int4 mine0;
  seg_header(d, seg0, lane, rec0, mine0);                       // every wave reads the chunk's first record (size, slot)
  const int size = __builtin_amdgcn_readlane(rec0, 2), hoff = __builtin_amdgcn_readlane(rec0, 5), doff = __builtin_amdgcn_readlane(rec0, 6);
  const int dst = (fin && wave == 0 && lane < size) ? d.blk_dst[doff + lane] : -1;
  double acc = 0.0;
  for (int sgi = wave; sgi < nsc; sgi += kK2Waves) {
    int rec; int4 mine;
    if (sgi == 0) { rec = rec0; mine = mine0; } else seg_header(d, seg0 + sgi, lane, rec, mine);
    acc += wave_segment_sum(d, rec, mine, lane, plist);
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0 || lane >= size) return;
  // four interleaved sums q_w = part[w] + part[w + 4] + ..., combined as (q0 + q1) + (q2 + q3): for a chunk (<= 16 segments, one per
  // wave) the order in which kb_hreduce adds the segments of a block -- the two forms of K2 give the same bits
  const int nw = nsc < kK2Waves ? nsc : kK2Waves;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int w = 0; w < 4; w++)
    for (int v = w; v < nw; v += 4) q[w] += part[v][lane];
  const double tot = (q[0] + q[1]) + (q[2] + q[3]);
  d.H[hoff + lane] = tot;
  if (dst >= 0) d.Hf[dst] = tot;
}

// the blocks of more than kK2Chunk segments: their chunks' partial sums (slot of every kK2Chunk-th segment) added in chunk order
constexpr int kK2Chunk = 16;
__device__ __forceinline__ void body_hfinish(const DevGraph& d, int bx) {
  const int lane = threadIdx.x & 63;
  const int seg0 = uni(d.k2_finish[bx]);
  const int rec = d.srec[(size_t)seg0 * 8 + (lane & 7)];
  const int size = __builtin_amdgcn_readlane(rec, 2), hoff = __builtin_amdgcn_readlane(rec, 5), doff = __builtin_amdgcn_readlane(rec, 6);
  const int nseg = __builtin_amdgcn_readlane(rec, 7);
  const int nch = (nseg + kK2Chunk - 1) / kK2Chunk;
  const int ln = lane < size ? lane : 0;
  const int dst = lane < size ? d.blk_dst[doff + lane] : -1;
  double* __restrict__ h = d.H + hoff;
  double tot = 0.0;
  for (int c0 = 0; c0 < nch; c0 += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { const int c = c0 + u; v[u] = h[(size_t)(c < nch ? c : 0) * kK2Chunk * size + ln]; }
#pragma unroll
    for (int u = 0; u < 16; u++) tot += c0 + u < nch ? v[u] : 0.0;
  }
  if (lane >= size) return;
  h[lane] = tot;
  if (dst >= 0) d.Hf[dst] = tot;
}
__global__ __launch_bounds__(64) void k_hfinish(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hfinish(d, blockIdx.x); }

__global__ __launch_bounds__(64 * kK2Waves) void k_hblocks2(DevGraph d, LinGuard gd) { if (!lin_guard(gd)) return; body_hblocks2(d, blockIdx.x); }

// K2 of the speculative linearisation (SpecLin): the launch above on the spare set's buffers
hipError_t launch_hblocks_spec(const DevGraph& d_in, const SpecLin& sl, hipStream_t st, const LinGuard* guard) {
  DevGraph d = d_in;
  d.J = sl.J; d.P = sl.P; d.H = sl.H; d.Hf = sl.Hf;
  return launch_hblocks(d, st, guard, true);
}

hipError_t launch_hblocks(const DevGraph& d_in, hipStream_t st, const LinGuard* guard, bool products) {
  const int nb = (d_in.n_k2_single + kK2Waves - 1) / kK2Waves + d_in.n_k2_multi;
  if (nb == 0) return hipSuccess;
  const LinGuard gd = guard ? *guard : LinGuard{};
  DevGraph d = d_in;
  if (!products) d.P = nullptr;                                 // (K1 ran one thread per factor: Jacobians only)
  PPS_LAUNCH(k_hblocks2, dim3(nb), dim3(64 * kK2Waves), 0, st, d, gd);
  if (d.n_k2_finish > 0) PPS_LAUNCH(k_hfinish, dim3(d.n_k2_finish), dim3(64), 0, st, d, gd);      // (only graphs with a landmark of > 1 024 observations)
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2, throughput form (batches of more than 200 000 factors, whose K1 runs one thread per factor and writes no product records).  The Jacobian slices of a segment's contributions are staged in LDS first --
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
  // (a plane observation's contribution to a diagonal block names its product record in `ju`: this form multiplies the Jacobian
  // slices of every contribution -- the thread-per-factor K1 it pairs with writes no product records)
  if (h.mine.w >= kProductFlag) { h.mine.w -= kProductFlag; h.mine.y = h.mine.x; }
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

// ------------------------------------------------------------------------------------------
// K2, throughput form by segment class.  The generic wave-per-segment body above spends ~55 wave instructions on every
// contribution whatever the block holds (three clamped loads, three LDS writes, a dozen LDS reads with computed addresses, the lane
// -> (i, j) division by a run-time column count) -- and a C2 graph has 13 000 contributions in three shapes only.  pps_multi sorts a
// graph's non-direct segments into classes once per upload (pps_device.h: k2t) and the classes get their own bodies:
//  * pose diagonal (6 x 6 + g, rows of 3 from plane observations / 6 from odometry; row slice = column slice): ONE load and ONE LDS
//    write per contribution -- lanes [0, 6 m) take the slice, the next m the residual, stored at 36 + 6 k so that the rhs lanes
//    read their second operand with the same stride (immediate offsets) as the matrix lanes;
//  * pose-pose off-diagonal (6 x 6, rows of 6): two loads, compile-time strides;
//  * plane diagonal (3 x 3 + g, rows of 3: 12 of 64 lanes in the generic body, 20-64 contributions each): FOUR segments per wave, one
//    per 16-lane group, a contribution of each per step -- one descriptor load (the same address across a group) and one load
//    of slice + residual per step.
// Everything else keeps the generic body.  Per entry the sum runs over the contributions in list order and k ascending as explicit
// multiply-adds -- the same bits as every other form of K2.
// ------------------------------------------------------------------------------------------
constexpr int kT66Slots = 8;
constexpr int kT66Stride = 72;       // doubles per slot: [slice, k-major <= 36 | r[k] at 36 + 6 k]  /  [row slice 36 | column slice 36]
constexpr int kT33U = 8;             // contributions in flight per 16-lane group
constexpr int kTcWaveDoubles = kH2WaveDoubles;          // (the generic body's area: the largest)
static_assert(kTcWaveDoubles >= kT33U * 4 * 16 && kTcWaveDoubles >= kT66Slots * kT66Stride, "one LDS area per wave serves all bodies");

__device__ __forceinline__ void wave_hblock_66_diag(const DevGraph& d, const SegHdr& h, double* __restrict__ S) {
  const int lane = threadIdx.x & 63;
  const int4 mine = h.mine;
  const int cnt = __builtin_amdgcn_readlane(h.rec, 4), hoff = __builtin_amdgcn_readlane(h.rec, 5);
  const bool is_g = lane >= 36;
  const int i = is_g ? lane - 36 : lane / 6;
  const int j = is_g ? 0 : lane - 6 * (lane / 6);
  const int pb0 = is_g ? 36 : j;
  // what a lane fetches and where it puts it, for rows of 3 and of 6: slice element / residual (lanes past both repeat the last residual)
  const int t3 = lane - 18 < 2 ? lane - 18 : 2, t6 = lane - 36 < 5 ? lane - 36 : 5;
  const int l3 = lane < 18 ? lane : t3, l6 = lane < 36 ? lane : t6;
  const int w3 = lane < 18 ? lane : 36 + 6 * t3, w6 = lane < 36 ? lane : 36 + 6 * t6;
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  for (int cb = 0; cb < cnt; cb += kT66Slots) {
    const int nc = cnt - cb < kT66Slots ? cnt - cb : kT66Slots;
    double x[kT66Slots];
    int w[kT66Slots];
#pragma unroll
    for (int u = 0; u < kT66Slots; u++) {
      const int cu = cb + (u < nc ? u : nc - 1);
      const int jv = __builtin_amdgcn_readlane(mine.x, cu), ro = __builtin_amdgcn_readlane(mine.z, cu), m = __builtin_amdgcn_readlane(mine.w, cu);
      const bool six = m > 3;
      const bool isr = six ? lane >= 36 : lane >= 18;
      w[u] = six ? w6 : w3;
      x[u] = J[(isr ? ro : jv) + (six ? l6 : l3)];
    }
#pragma unroll
    for (int u = 0; u < kT66Slots; u++) S[u * kT66Stride + w[u]] = x[u];
    __builtin_amdgcn_wave_barrier();
    for (int u = 0; u < nc; u++) {
      const int m = __builtin_amdgcn_readlane(mine.w, cb + u);
      const double* __restrict__ pa = S + u * kT66Stride + i;
      const double* __restrict__ pb = S + u * kT66Stride + pb0;
#pragma unroll
      for (int k = 0; k < 3; k++) acc = PPS_MAC(acc, pa[6 * k], pb[6 * k]);
      if (m > 3) {
#pragma unroll
        for (int k = 3; k < 6; k++) acc = PPS_MAC(acc, pa[6 * k], pb[6 * k]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane >= 42) return;
  if (is_g) acc = -acc;                                   // b = -r (isam/Jacobian.h:98)
  d.H[hoff + lane] = acc;
  if (h.dst >= 0) d.Hf[h.dst] = acc;
}

__device__ __forceinline__ void wave_hblock_66_off(const DevGraph& d, const SegHdr& h, double* __restrict__ S) {
  const int lane = threadIdx.x & 63;
  const int4 mine = h.mine;
  const int cnt = __builtin_amdgcn_readlane(h.rec, 4), hoff = __builtin_amdgcn_readlane(h.rec, 5);
  const int lc = lane < 36 ? lane : 35;
  const int i = lc / 6, j = lc - 6 * (lc / 6);
  const double* __restrict__ J = d.J;
  double acc = 0.0;
  for (int c = 0; c < cnt; c++) {                         // (one, unless two factors join the same pair of poses)
    const int jv = __builtin_amdgcn_readlane(mine.x, c), ju = __builtin_amdgcn_readlane(mine.y, c);
    const double xv = J[jv + lc], xu = J[ju + lc];
    S[lc] = xv; S[36 + lc] = xu;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 6; k++) acc = PPS_MAC(acc, S[6 * k + i], S[36 + 6 * k + j]);
    __builtin_amdgcn_wave_barrier();
  }
  if (lane >= 36) return;
  d.H[hoff + lane] = acc;
  if (h.dst >= 0) d.Hf[h.dst] = acc;
}

// four plane diagonals per wave: 16-lane group q works on list entry first + q
__device__ __forceinline__ void wave_hblock_33x4(const DevGraph& d, int first, double* __restrict__ S) {
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, gl = lane & 15;
  const int at = first + grp;
  int4 r0 = make_int4(0, 0, 0, 0), r1 = make_int4(0, 0, 0, 0);
  if (at < d.n_k2t_small) {
    const int seg = d.k2t[d.n_k2t_big + at];
    r0 = reinterpret_cast<const int4*>(d.srec)[2 * (size_t)seg]; r1 = reinterpret_cast<const int4*>(d.srec)[2 * (size_t)seg + 1];
  }
  const int size = r0.z, c0 = r0.w, cnt = r1.x, hoff = r1.y, doff = r1.z, nsegb = r1.w;
  const int dst = (gl < size && nsegb == 1) ? d.blk_dst[doff + gl] : -1;
  int cmax = __builtin_amdgcn_readlane(cnt, 0);
  { const int c1 = __builtin_amdgcn_readlane(cnt, 16), c2 = __builtin_amdgcn_readlane(cnt, 32), c3 = __builtin_amdgcn_readlane(cnt, 48);
    cmax = cmax > c1 ? cmax : c1; cmax = cmax > c2 ? cmax : c2; cmax = cmax > c3 ? cmax : c3; }
  const bool is_g = gl >= 9;
  const int t = gl - 9 < 2 ? gl - 9 : 2;                   // residual row of the lanes past the slice (lanes 12 .. 15 repeat row 2)
  const int i = is_g ? t : gl / 3;
  const int j = is_g ? 0 : gl - 3 * (gl / 3);
  const int pb0 = is_g ? 9 : j;
  const int loff = is_g ? t : gl, wpos = is_g ? 9 + 3 * t : gl;    // [slice, k-major 9 | r[k] at 9 + 3 k]
  const int last = cnt > 0 ? c0 + cnt - 1 : 0;
  const double* __restrict__ J = d.J;
  double* __restrict__ Sg = S + grp * 16;
  double acc = 0.0;
  for (int k0 = 0; k0 < cmax; k0 += kT33U) {
    int4 ds[kT33U];
#pragma unroll
    for (int u = 0; u < kT33U; u++) ds[u] = reinterpret_cast<const int4*>(d.contrib)[k0 + u < cnt ? c0 + k0 + u : last];
    double x[kT33U];
#pragma unroll
    for (int u = 0; u < kT33U; u++) x[u] = J[(is_g ? ds[u].z : ds[u].x) + loff];
#pragma unroll
    for (int u = 0; u < kT33U; u++) Sg[u * 64 + wpos] = x[u];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < kT33U; u++) {
      if (k0 + u < cnt) {
        const double* __restrict__ pa = Sg + u * 64 + i;
        const double* __restrict__ pb = Sg + u * 64 + pb0;
#pragma unroll
        for (int k = 0; k < 3; k++) acc = PPS_MAC(acc, pa[3 * k], pb[3 * k]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (gl >= size) return;
  if (is_g) acc = -acc;
  d.H[hoff + gl] = acc;
  if (dst >= 0) d.Hf[dst] = acc;
}

// GENERIC = false: the entries with a class body ([0, n_k2t_spec): pose diagonals, pose-pose blocks) and the plane diagonals;
// GENERIC = true: the entries behind them, one generic body -- a kernel of its own, so that its registers (it holds most of the 126
// the common kernel needed) do not set the occupancy of the class bodies, which wait on memory most of the time.
template <bool GENERIC>
__device__ __forceinline__ void body_hblocks_tc(const DevGraph& d, int bx) {
  __shared__ double tc_lds[4 * kTcWaveDoubles];
  const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;
  double* __restrict__ S = tc_lds + wave * kTcWaveDoubles;
  const int e_begin = GENERIC ? d.n_k2t_spec : 0, e_end = GENERIC ? d.n_k2t_big : d.n_k2t_spec;
  const int nb_big = (e_end - e_begin + 15) / 16;
  if (bx >= nb_big) {
    if (GENERIC) return;
    const int first = uni(((bx - nb_big) * 4 + wave) * 4);
    if (first < d.n_k2t_small) wave_hblock_33x4(d, first, S);
    return;
  }
  const int slot0 = uni(e_begin + (bx * 4 + wave) * 4);
  if (slot0 >= e_end) return;
  SegHdr h[4];
  int cls[4];
  {
    const int e = slot0 + (lane >> 3) < e_end && (lane >> 3) < 4 ? d.k2t[slot0 + (lane >> 3)] : -1;   // lanes 8q..8q+7: entry q
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int eq = __builtin_amdgcn_readlane(e, 8 * q);
      cls[q] = eq >= 0 ? eq >> 28 : 0;
      seg_fetch_record(d, eq >= 0 ? (eq & 0x0fffffff) : -1, lane, h[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) seg_fetch_contrib(d, lane, h[q]);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    // (a slot past the end of the list has class 0 and an empty record: the generic body returns on size 0, the class bodies are not
    // entered -- wave_hblock_66_off has no guard of its own and would zero H[0 .. 35])
    if (GENERIC) wave_hblock_segment(d, h[q], S);
    else if (cls[q] == 1) wave_hblock_66_diag(d, h[q], S);
    else if (cls[q] == 2) wave_hblock_66_off(d, h[q], S);
  }
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


// ---- batched forms ----
template <bool PRODUCTS>
__global__ __launch_bounds__(64 * kK2Waves) void kb_hblocks2(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= (d.n_k2_single + kK2Waves - 1) / kK2Waves + d.n_k2_multi) return;
  if (PRODUCTS) { body_hblocks2(d, blockIdx.x); return; }
  DevGraph dn = d;
  dn.P = nullptr;                                               // (the chunk's K1 wrote Jacobians only: closed-form mode)
  body_hblocks2(dn, blockIdx.x);
}

__global__ __launch_bounds__(64) void kb_hfinish(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= d.n_k2_finish) return;
  body_hfinish(d, blockIdx.x);
}

__global__ __launch_bounds__(256) void kb_hblocks_tc(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= (d.n_k2t_spec + 15) / 16 + (d.n_k2t_small + 15) / 16) return;
  body_hblocks_tc<false>(d, blockIdx.x);
}
__global__ __launch_bounds__(256) void kb_hblocks_tg(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= (d.n_k2t_big - d.n_k2t_spec + 15) / 16) return;
  body_hblocks_tc<true>(d, blockIdx.x);
}

__global__ __launch_bounds__(64) void kb_hreduce(BatchArgs a) {
  PPS_BATCH_PROLOGUE(BF_ACTIVE | BF_RELIN)
  if ((int)blockIdx.x >= d.n_mseg) return;
  body_hreduce<1>(d, blockIdx.x);
}

hipError_t launch_batch_hblocks(const BatchArgs& a, const BatchGeom& g, hipStream_t st, bool products) {
  if (g.lin_thread_form) {                                      // many graphs: throughput form over the Jacobians + the second pass
    // (the wave-per-segment kernel in its Jacobian-only mode, measured on the same G = 128 batch: 23.9 ms of K2 per batch solve
    // against 13.7 ms -- at this size the LDS-staged form's four segments per wave and prefetched headers win)
    if (g.k2t_blocks > 0) PPS_LAUNCH(kb_hblocks_tc, dim3(g.k2t_blocks, a.n), dim3(256), 0, st, a);
    if (g.k2tg_blocks > 0) PPS_LAUNCH(kb_hblocks_tg, dim3(g.k2tg_blocks, a.n), dim3(256), 0, st, a);
    if (g.hreduce > 0) PPS_LAUNCH(kb_hreduce, dim3(g.hreduce, a.n), dim3(64), 0, st, a);
    return hipGetLastError();
  }
  if (g.k2_blocks > 0) {
    if (products) PPS_LAUNCH(kb_hblocks2<true>, dim3(g.k2_blocks, a.n), dim3(64 * kK2Waves), 0, st, a);
    else PPS_LAUNCH(kb_hblocks2<false>, dim3(g.k2_blocks, a.n), dim3(64 * kK2Waves), 0, st, a);
  }
  if (g.k2_finish > 0) PPS_LAUNCH(kb_hfinish, dim3(g.k2_finish, a.n), dim3(64), 0, st, a);
  return hipGetLastError();
}

}  // namespace pps
