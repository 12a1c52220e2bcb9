// front_emu.cpp -- csrc/pps_front_reg.h (the register-tile elimination of one front, 8-column panels) run by the host wave
// emulation of wave_emu.h.  Same contract as pps_debug_front_factor (include/pps.h): one packed frontal matrix in, factor panel,
// update matrix and the not-positive-definite flag out.  Test infrastructure (tests/test_front_emu.py): the SOURCE the GPU runs,
// executed lane by lane on the CPU and compared with numpy.
#define PPS_WAVE_EMU 1
#include "wave_emu.h"

#include "pps_front_reg.h"

namespace {
struct EmuGraph { double* L; double* U; double* result_dev; long long* trace; };

template <int NT, bool STRIP, int W>
void run_front(int p, int b, const double* A, double* L, double* U, double* res, bool alias) {
  using namespace pps;
  const int fa = p + b + 1;
  const int ntri = fa * (fa + 1) / 2;
  const int prow = (STRIP ? kRegRowsMax : kRegRows) * kP8Stride;   // (every lane writes its panel row, whatever NT is)
  // alias: the panel buffer IS the start of the triangle (what the register-only kernels do); else behind it
  std::vector<double> lds((alias ? std::max(ntri, prow) : ntri + 1 + prow) + 2 + 64, std::nan(""));   // (+ 64: the tile load reads past a tiny triangle)
  double* F = lds.data();
  double* P = alias ? F : F + ntri + 1;
  for (int i = 0; i < ntri; i++) F[i] = A[i];
  EmuGraph d{L, U, res, nullptr};
  pps_emu::run_wave([&] {
    const int lane = threadIdx.x;
    int rec = 0;                                  // (L and U of the front start at offset 0)
    if (lane == 1) rec = p;
    if (lane == 2) rec = b;
    front_reg_eliminate<NT, false, STRIP, W>(d, rec, F, P);
  });
}
}  // namespace

template <int W>
static int front_factor_w(int tiles, int strip, int p, int b, const double* A, double* L, double* U, double* not_pd, long long* counts) {
  using namespace pps;
  const int f = p + b, fa = f + 1;
  if (p < 1 || b < 0 || fa > kRegRowsMax || p > kRegRows) return -1;
  if (tiles == 0) tiles = fa <= 33 ? 2 : fa <= 49 ? 3 : fa <= kRegRows ? 4 : (strip ? 4 : 5);
  const bool wide = fa > kRegRows;
  if (tiles < 2 || tiles > 5 || (wide && tiles < 4) || (!wide && (tiles == 5 || strip)) || (tiles < 4 && f > 16 * tiles) || (wide && tiles == 4 && !strip)) return -1;
  std::vector<double> Lb((size_t)fa * p, 0.0), Ub((size_t)(b + 1) * (b + 1), 0.0);
  double res[4] = {0, 0, 0, 0};
  pps_emu::Wave& w = pps_emu::W();
  w.n_yields = w.n_readlane = w.n_mfma = w.n_barrier = 0;
  if (tiles == 5) run_front<5, true, W>(p, b, A, Lb.data(), Ub.data(), res, false);
  else if (tiles == 4 && wide) run_front<4, true, W>(p, b, A, Lb.data(), Ub.data(), res, false);
  else if (tiles == 4) run_front<4, false, W>(p, b, A, Lb.data(), Ub.data(), res, true);
  else if (tiles == 3) run_front<3, false, W>(p, b, A, Lb.data(), Ub.data(), res, true);
  else run_front<2, false, W>(p, b, A, Lb.data(), Ub.data(), res, true);
  std::memcpy(L, Lb.data(), Lb.size() * 8);
  std::memcpy(U, Ub.data(), (size_t)(b + 1) * (b + 2) / 2 * 8);
  if (not_pd) *not_pd = res[2];
  if (counts) { counts[0] = w.n_readlane / pps_emu::kLanes; counts[1] = w.n_mfma / pps_emu::kLanes; counts[2] = w.n_barrier / pps_emu::kLanes; }
  return 0;
}

// W = 8: what the r5 / general / level kernels run; W = 4: the register-only band kernels (one pivot block per panel step)
extern "C" int emu_front_factor(int tiles, int strip, int p, int b, const double* A, double* L, double* U, double* not_pd, long long* counts) {
  return front_factor_w<8>(tiles, strip, p, b, A, L, U, not_pd, counts);
}
extern "C" int emu_front_factor_w4(int tiles, int strip, int p, int b, const double* A, double* L, double* U, double* not_pd, long long* counts) {
  return front_factor_w<4>(tiles, strip, p, b, A, L, U, not_pd, counts);
}
