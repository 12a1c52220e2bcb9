// pps_kcommon.h -- small device helpers shared by the solver kernels (K1 .. K4) and their batched forms.
#pragma once
#include <hip/hip_runtime.h>

#include "pps_device.h"
#include "pps_geom.h"

namespace pps {

// H = J'J accumulations are written as explicit multiply-adds: three kernels (k_hblocks, the batched kb_hblocks_t and the
// direct blocks of the thread-per-factor sweep) must produce the same bits for the same block, whatever the compiler would
// have contracted on its own.  (PPS_NO_FMA: the diagnostic build without any fused operation.)
#ifdef PPS_NO_FMA
#define PPS_MAC(acc, a, b) ((acc) + (a) * (b))
#else
#define PPS_MAC(acc, a, b) __builtin_fma((a), (b), (acc))
#endif

// Values that are wave-uniform by construction (they derive from threadIdx.x >> 6) but that the
// compiler must treat as divergent: pin them into SGPRs so loops, branches and address arithmetic
// built on them are scalar instead of exec-masked "waterfall" code.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long long uni64(long long x) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}

// ------------------------------------------------------------------------------------------
// K1: one thread per factor; SoA loads (coalesced across the wave), state gathered by index.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_pose(const double* __restrict__ base, int ld, int i, double p[7]) {
#pragma unroll
  for (int k = 0; k < 7; k++) p[k] = base[(size_t)k * ld + i];
}
__device__ __forceinline__ void load_plane(const double* __restrict__ base, int ld, int i, double p[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++) p[k] = base[(size_t)k * ld + i];
}
template <int K>
__device__ __forceinline__ void load_soa(const double* __restrict__ base, int ld, int i, double* o) {
#pragma unroll
  for (int k = 0; k < K; k++) o[k] = base[(size_t)k * ld + i];
}

// The state a residual is evaluated at: the stored copy, or (APPLY) base (+) delta computed on the spot -- the fused trial kernel
// evaluates chi2 at x (+) delta without waiting for the retraction to be written (pose_exmap / plane_exmap are compiled without
// contraction, pps_geom.h: the same bits as the stored copy).
template <bool APPLY>
__device__ __forceinline__ void fetch_pose(const DevGraph& d, const double* __restrict__ pose, int idx, double o[7]) {
  if (!APPLY) { load_pose(pose, d.pose_ld, idx, o); return; }
  double p[7], dl[6];
  load_pose(pose, d.pose_ld, idx, p);
  const int off = d.pose_voff[idx];
#pragma unroll
  for (int k = 0; k < 6; k++) dl[k] = d.delta[off + k];
  pose_exmap(p, dl, o);
}
template <bool APPLY>
__device__ __forceinline__ void fetch_plane(const DevGraph& d, const double* __restrict__ plane, int idx, double o[4]) {
  if (!APPLY) { load_plane(plane, d.plane_ld, idx, o); return; }
  double p[4], dl[3];
  load_plane(plane, d.plane_ld, idx, p);
  const int off = d.plane_voff[idx];
#pragma unroll
  for (int k = 0; k < 3; k++) dl[k] = d.delta[off + k];
  plane_exmap(p, dl, o);
}

// LinGuard (pps_device.h): which state does a speculatively queued K1 linearise at?  false = neither trial was accepted
__device__ __forceinline__ bool lin_guard(const LinGuard& gd, const double* __restrict__& pose, const double* __restrict__& plane) {
  if (!gd.on) return true;
  const double c0 = *gd.chi[0], c1 = *gd.chi[1];
  if (gd.error - c0 > 0.) { pose = gd.pose[0]; plane = gd.plane[0]; return true; }
  if (!gd.single && gd.error - c1 > 0.) { pose = gd.pose[1]; plane = gd.plane[1]; return true; }
  return false;
}
__device__ __forceinline__ bool lin_guard(const LinGuard& gd) {
  if (!gd.on) return true;
  const bool a0 = gd.error - *gd.chi[0] > 0., a1 = !a0 && !gd.single && gd.error - *gd.chi[1] > 0.;
  return gd.want < 0 ? (a0 || a1) : (gd.want == 0 ? a0 : a1);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// Batched forms (pps_multi_*): the same bodies, blockIdx.y = graph.  The DevGraph record of the graph is read through
// the constant address space -- it does not change while a kernel runs -- so that its fields arrive by scalar loads
// into SGPRs exactly like the by-value kernel argument of the single-graph kernels.
// ------------------------------------------------------------------------------------------
// a pointer that came out of memory is generic to the compiler; these all point into HBM (or pinned host memory)
template <class T>
__device__ __forceinline__ T* gptr(T* p) {
  // through an integer, so that the address-space round trip is not folded away: on gfx9 a global address and its generic
  // form are the same 64 bits
  return (T*)(T __attribute__((address_space(1)))*)(unsigned long long)p;
}

__device__ __forceinline__ DevGraph load_graph(const DevGraph* gp) {
  DevGraph d = *(const DevGraph*)((const DevGraph __attribute__((address_space(4)))*)gp);   // scalar loads; unused fields drop out
#define PPS_G(f) d.f = gptr(d.f);
  PPS_G(pose_est) PPS_G(pose_lin) PPS_G(plane_est) PPS_G(plane_lin) PPS_G(pose_voff) PPS_G(plane_voff)
  PPS_G(obs_pose) PPS_G(obs_plane) PPS_G(obs_meas) PPS_G(obs_w) PPS_G(obs_ray) PPS_G(odo_a) PPS_G(odo_b) PPS_G(odo_meas) PPS_G(odo_w)
  PPS_G(pp_pose) PPS_G(pp_meas) PPS_G(pp_w) PPS_G(lp_plane) PPS_G(lp_meas) PPS_G(lp_w)
  PPS_G(J) PPS_G(P) PPS_G(H) PPS_G(L) PPS_G(U) PPS_G(delta)
  PPS_G(f_p) PPS_G(f_b) PPS_G(f_poff) PPS_G(pidx) PPS_G(f_Loff) PPS_G(f_Uoff) PPS_G(f_bidx_off) PPS_G(bidx) PPS_G(f_child_off) PPS_G(child)
  PPS_G(f_cmap_off) PPS_G(cmap) PPS_G(level_fronts) PPS_G(f_asm_off) PPS_G(asm_blk) PPS_G(asm_lrow) PPS_G(asm_lcol) PPS_G(asm_el0) PPS_G(asm_fsz)
  PPS_G(blk_rows) PPS_G(blk_cols) PPS_G(blk_size) PPS_G(blk_nseg) PPS_G(blk_hoff) PPS_G(seg_blk) PPS_G(seg_c0) PPS_G(seg_cnt) PPS_G(seg_hoff)
  PPS_G(contrib) PPS_G(mseg_blk) PPS_G(k2_single) PPS_G(k2_multi) PPS_G(k2_finish) PPS_G(f_el_off) PPS_G(el_src) PPS_G(el_tgt) PPS_G(blk_doff) PPS_G(blk_dst) PPS_G(Hf) PPS_G(f_ea_off) PPS_G(ea_tgt)
  PPS_G(grp_lvl_off) PPS_G(glvl_front_off) PPS_G(glvl_fronts) PPS_G(grp_span) PPS_G(frec) PPS_G(crec) PPS_G(srec) PPS_G(cls_off) PPS_G(cls_fronts)
  PPS_G(chi2_partials) PPS_G(dn_partials) PPS_G(ticket) PPS_G(result_dev) PPS_G(trace) PPS_G(gwork)
#undef PPS_G
  return d;
}

__device__ __forceinline__ BatchAlt load_alt(const BatchAlt* ap) {
  BatchAlt t = *(const BatchAlt*)((const BatchAlt __attribute__((address_space(4)))*)ap);
  t.L = gptr(t.L); t.U = gptr(t.U); t.delta = gptr(t.delta); t.result_dev = gptr(t.result_dev);
  t.chi2_partials = gptr(t.chi2_partials); t.dn_partials = gptr(t.dn_partials); t.ticket = gptr(t.ticket);
#pragma unroll
  for (int k = 0; k < 3; k++) { t.pose[k] = gptr(t.pose[k]); t.plane[k] = gptr(t.plane[k]); }
  return t;
}
__device__ __forceinline__ double* sel3(double* const (&p)[3], int k) { return k == 0 ? p[0] : (k == 1 ? p[1] : p[2]); }

// The linearisation point of a graph in a round is state[xsel] of its three state copies -- it takes the place of `lin` for K1
// and chi2 -- and nothing is ever written over it.
#define PPS_BATCH_PROLOGUE(NEED)                                                                             \
  const int b = blockIdx.y;                                                                                  \
  const unsigned int fl = a.flags[b];                                                                        \
  if ((fl & (NEED)) != (NEED)) return;                                                                       \
  const DevGraph d = load_graph(a.gs + a.b0 + b);                                                            \
  double* pose_lin = d.pose_lin;                                                                             \
  double* plane_lin = d.plane_lin;                                                                           \
  if (a.alt) {                                                                                               \
    const BatchAlt al_ = load_alt(a.alt + a.b0 + b);                                                         \
    pose_lin = sel3(al_.pose, a.xsel[b]); plane_lin = sel3(al_.plane, a.xsel[b]);                            \
  }                                                                                                          \
  (void)pose_lin; (void)plane_lin;

__device__ __forceinline__ int dcdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace pps
