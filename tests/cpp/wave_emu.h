// wave_emu.h -- a 64-lane wavefront on the host, for the wave-level device headers of csrc/ (test infrastructure).
//
// The register-tile code of K3 is written per lane with cross-lane operations in between (v_readlane, MFMA, LDS hand-overs behind a
// wave barrier).  Here every lane is a coroutine (ucontext) running the SAME source; a cross-lane operation publishes the
// lane's operands, yields until all 64 lanes have arrived, and reads what it needs.  Control flow around cross-lane operations is
// wave-uniform in the device code (it has to be: EXEC-masked lanes do not take part in an MFMA), so all lanes meet at the same
// sequence of yield points.  Between two yield points the lanes run one after the other, lane 0 first: code that relied on
// lockstep execution WITHOUT a wave barrier between an LDS write and another lane's read would give different results here --
// which is what a test should flag.
//
// Usage: #define PPS_WAVE_EMU, include this header, then the device header; run a body with pps_emu::run_wave(fn).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace pps_emu {

constexpr int kLanes = 64;
struct Wave {
  ucontext_t main_ctx;
  ucontext_t ctx[kLanes];
  std::vector<char> stack[kLanes];
  bool done[kLanes];
  int cur = 0;
  std::function<void()> body;
  // exchange buffers of the cross-lane operations
  double xd[kLanes], xd2[kLanes];
  int xi[kLanes];
  long long n_yields = 0, n_readlane = 0, n_mfma = 0, n_barrier = 0;
};
inline Wave& W() { static Wave w; return w; }

inline void yield() {
  Wave& w = W();
  const int me = w.cur;
  w.n_yields++;
  int nxt = me;
  do { nxt = (nxt + 1) % kLanes; } while (w.done[nxt] && nxt != me);
  if (nxt == me) return;
  w.cur = nxt;
  swapcontext(&w.ctx[me], &w.ctx[nxt]);
}
inline void lane_entry() {
  Wave& w = W();
  w.body();
  // lanes finish in the order they run: hand over to the next live lane, the last one returns to the caller of run_wave
  const int me = w.cur;
  w.done[me] = true;
  for (int k = 1; k < kLanes; k++) {
    const int nxt = (me + k) % kLanes;
    if (!w.done[nxt]) { w.cur = nxt; setcontext(&w.ctx[nxt]); }
  }
  setcontext(&w.main_ctx);
}
inline void run_wave(std::function<void()> fn) {
  Wave& w = W();
  w.body = std::move(fn);
  for (int l = 0; l < kLanes; l++) {
    w.done[l] = false;
    w.stack[l].assign(1 << 20, 0);
    getcontext(&w.ctx[l]);
    w.ctx[l].uc_stack.ss_sp = w.stack[l].data();
    w.ctx[l].uc_stack.ss_size = w.stack[l].size();
    w.ctx[l].uc_link = nullptr;
    makecontext(&w.ctx[l], lane_entry, 0);
  }
  w.cur = 0;
  swapcontext(&w.main_ctx, &w.ctx[0]);
}

struct Tid { int x; };
inline Tid tid() { return Tid{W().cur}; }
inline void wave_barrier() { W().n_barrier++; yield(); }
inline int readlane_i(int v, int l) {
  Wave& w = W();
  w.n_readlane++;
  w.xi[w.cur] = v; yield();
  const int r = w.xi[l & 63]; yield();
  return r;
}
inline double readlane_dd(double v, int l) {
  Wave& w = W();
  w.n_readlane += 2;
  w.xd[w.cur] = v; yield();
  const double r = w.xd[l & 63]; yield();
  return r;
}

}  // namespace pps_emu

// ---- the HIP vocabulary the device headers use ----
#define __device__
#define __host__
#define __forceinline__ inline
#define __global__
#define threadIdx (pps_emu::tid())
#define __builtin_amdgcn_readlane(v, l) pps_emu::readlane_i((v), (l))
#define __builtin_amdgcn_wave_barrier() pps_emu::wave_barrier()
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt((double)(x)))
#define __mul24(a, b) ((int)(a) * (int)(b))
#define clock64() 0LL

namespace pps {
typedef double double4_t __attribute__((vector_size(32)));
inline double readlane_d(double x, int l) { return pps_emu::readlane_dd(x, l); }
}  // namespace pps

// v_mfma_f64_16x16x4_f64: D = A (16 x 4) * B (4 x 16) + C.  Lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; register r of
// the accumulator is C[(l / 16) + 4 r][l % 16] (the gfx950 layout, probed on hardware: tools/mfma_probe.hip).
inline pps::double4_t pps_emu_mfma(double a, double b, pps::double4_t c) {
  pps_emu::Wave& w = pps_emu::W();
  w.n_mfma++;
  const int l = w.cur;
  w.xd[l] = a; w.xd2[l] = b;
  pps_emu::yield();
  const int col = l & 15, lq = l >> 4;
  for (int r = 0; r < 4; r++) {
    const int row = lq + 4 * r;
    double acc = c[r];
    for (int k = 0; k < 4; k++) acc = std::fma(w.xd[row + 16 * k], w.xd2[col + 16 * k], acc);
    c[r] = acc;
  }
  pps_emu::yield();
  return c;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) pps_emu_mfma((a), (b), (c))
