// pps_solve.cpp -- the solve drivers.  Host control flow follows the reference line by line where it matters for parity:
//   pps_batch_optimize  == Optimizer::levenberg_marquardt  (Thirdparty/isam/isamlib/Optimizer.cpp:371-467)
//   pps_update          == Optimizer::relinearize          (Optimizer.cpp:114-185) via Slam::update, mod_batch = 1
// Everything numeric runs on the device; per LM trial one 32-byte result record (chi2, |delta|^2, not-PD flag) returns to the
// host for the accept / reject decision.
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

namespace pps_impl {

struct PhaseTimer {
  pps_graph* g; double* acc; bool on;
  PhaseTimer(pps_graph* g_, double* a) : g(g_), acc(a), on(g_->profiling >= 2) { if (on) (void)hipEventRecord(g->ev[0], g->stream); }
  ~PhaseTimer() {
    if (!on) return;
    (void)hipEventRecord(g->ev[1], g->stream);
    (void)hipEventSynchronize(g->ev[1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
    *acc += 1e-3 * ms;
  }
};

// linearise at `lin` (K1) and reduce the H blocks (K2)
// guard: the launches are speculative (dual LM loop) -- the caller counts them once it knows they ran
int do_linearize(pps_graph* g, const LinGuard* guard = nullptr) {
  if (g->profiling == 1) {
    if (g->k1_used + 2 > (int)g->k1_events.size()) {
      for (int k = 0; k < 2; k++) { hipEvent_t e; HIP_TRY(g, hipEventCreate(&e)); g->k1_events.push_back(e); }
    }
    g->k1_skip.resize(g->k1_events.size() / 2, 0);
    g->k1_skip[g->k1_used / 2] = 0;
    HIP_TRY(g, launch_linearize(g->dev, g->props.jacobian_mode, false, g->stream, guard, g->k1_events[g->k1_used], g->k1_events[g->k1_used + 1]));
    g->k1_used += 2;
  } else
  { PhaseTimer t(g, &g->stats.t_linearize); HIP_TRY(g, launch_linearize(g->dev, g->props.jacobian_mode, false, g->stream, guard)); }
  { PhaseTimer t(g, &g->stats.t_assemble); HIP_TRY(g, launch_hblocks(g->dev, g->stream, guard, k1_products(g->dev, g->props.jacobian_mode))); }
  if (!guard) g->stats.n_linearize++;
  return PPS_OK;
}

// delta = (J'J + lambda diag(J'J))^-1 J'b  (Optimizer::compute_gauss_newton_step, Optimizer.cpp:49-67)
// Factor stages leaves -> root, back-substitution stages root -> leaves, for one damping value (alt == nullptr) or two.  The root stage
// is one launch for both directions where its fronts allow it (band_root_fusable).  events: the factor launches carry their own start /
// stop events (profiling level 1; the fused root launch's pair spans its back-substitution as well).
static int enqueue_factor_solve(pps_graph* g, const DevGraph& dv, const DualAlt* alt, double lambda, hipStream_t st_, bool events) {
  const Analysis& A = g->an;
  if (g->k3_all) {                                                 // the whole tree: two launches (pps_k3.hip, XGroup)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (events) {
      if (g->fk_used + 2 > (int)g->fk_events.size()) for (int k = 0; k < 2; k++) { hipEvent_t e; HIP_TRY(g, hipEventCreate(&e)); g->fk_events.push_back(e); }
      e0 = g->fk_events[g->fk_used]; e1 = g->fk_events[g->fk_used + 1]; g->fk_used += 2;
    }
    g->k3_epoch = g->k3_epoch >= (1 << 30) ? 1 : g->k3_epoch + 1;
    HIP_TRY(g, launch_band_all(dv, alt, A.n_groups, g->k3_nw_factor, g->k3_nw_solve, g->k3_max_front, g->k3_max_panel, g->k3_max_grp, lambda, g->k3_epoch, st_, e0, e1));
    return PPS_OK;
  }
  const int top = A.n_stages - 1;
  const bool fuse = top >= 0 && band_root_fusable(dv, A.stage_grp_off[top + 1] - A.stage_grp_off[top], A.stage_max_front[top]);
  for (int st = 0; st < A.n_stages; st++) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (events) {                                                  // the launch's own start / stop: resolve_k1_events sums them into t_factor
      if (g->fk_used + 2 > (int)g->fk_events.size()) for (int k = 0; k < 2; k++) { hipEvent_t e; HIP_TRY(g, hipEventCreate(&e)); g->fk_events.push_back(e); }
      e0 = g->fk_events[g->fk_used]; e1 = g->fk_events[g->fk_used + 1]; g->fk_used += 2;
    }
    const int g0 = A.stage_grp_off[st], ng = A.stage_grp_off[st + 1] - g0;
    const bool pre = st < (int)g->stage_pre.size() && g->stage_pre[st] != 0;
    if (fuse && st == top)
      HIP_TRY(g, launch_band_root(dv, alt, g0, g->stage_nw_factor[st], g->stage_nw_solve[st], A.stage_max_front[st], g->stage_max_panel[st],
                                  g->stage_max_grp_fronts[st], lambda, st_, e0, e1, pre));
    else if (alt) HIP_TRY(g, launch_band_factor_dual(dv, *alt, g0, ng, g->stage_nw_factor[st], A.stage_max_front[st], lambda, st_, e0, e1, pre));
    else HIP_TRY(g, launch_band_factor(dv, g0, ng, g->stage_nw_factor[st], A.stage_max_front[st], lambda, st_, e0, e1, pre));
  }
  for (int st = fuse ? top - 1 : top; st >= 0; st--)
    HIP_TRY(g, launch_band_solve(dv, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_solve[st],
                                 g->stage_max_panel[st], g->stage_max_grp_fronts[st], st_, alt));
  return PPS_OK;
}

int do_solve_on(pps_graph* g, const DevGraph& dv, double lambda, hipStream_t st_) { return enqueue_factor_solve(g, dv, nullptr, lambda, st_, false); }

int do_solve(pps_graph* g, double lambda) {
  const Analysis& A = g->an;
  if (g->use_band) {
    if (g->profiling < 2) {            // no per-phase timing
      int rc = do_solve_on(g, g->dev, lambda, g->stream);
      if (rc != PPS_OK) return rc;
      g->stats.n_factorize++;
      return PPS_OK;
    }
    {
      PhaseTimer t(g, &g->stats.t_factor);
      for (int st = 0; st < A.n_stages; st++)
        HIP_TRY(g, launch_band_factor(g->dev, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_factor[st],
                                      A.stage_max_front[st], lambda, g->stream));
    }
    {
      PhaseTimer t(g, &g->stats.t_backsolve);
      for (int st = A.n_stages - 1; st >= 0; st--)
        HIP_TRY(g, launch_band_solve(g->dev, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_solve[st],
                                     g->stage_max_panel[st], g->stage_max_grp_fronts[st], g->stream));
    }
    g->stats.n_factorize++;
    return PPS_OK;
  }
  if (g->use_dense) {
    {
      PhaseTimer t(g, &g->stats.t_factor);
      HIP_TRY(g, hipMemsetAsync(g->dev.L, 0, (size_t)A.L_size * 8, g->stream));
      HIP_TRY(g, launch_dense_hpush(g->dev, g->max_el_per_front, lambda, g->stream));
      for (int l = 0; l < A.n_levels; l++) {
        const int base = A.level_off[l] + l, cnt = A.level_off[l + 1] - A.level_off[l];
        HIP_TRY(g, launch_dense_factor_level(g->dev, A.level_off[l], cnt, g->d_dw_asm + base, g->dw_asm[base + cnt], g->d_dw_pan + base,
                                             g->dw_pan[base + cnt], g->d_dw_trl + base, g->dw_trl[base + cnt], g->stream));
      }
    }
    {
      PhaseTimer t(g, &g->stats.t_backsolve);
      for (int l = A.n_levels - 1; l >= 0; l--)
        HIP_TRY(g, launch_dense_solve_level(g->dev, A.level_off[l], A.level_off[l + 1] - A.level_off[l], g->level_max_b[l], g->stream));
    }
    g->stats.n_factorize++;
    return PPS_OK;
  }
  {
    PhaseTimer t(g, &g->stats.t_factor);
    for (int l = 0; l < A.n_levels; l++)
      HIP_TRY(g, launch_factor_level(g->dev, A.level_off[l], A.level_off[l + 1] - A.level_off[l], g->level_max_front[l], lambda,
                                     g->stream));
  }
  {
    PhaseTimer t(g, &g->stats.t_backsolve);
    for (int l = A.n_levels - 1; l >= 0; l--)
      HIP_TRY(g, launch_backsolve_level(g->dev, A.level_off[l], A.level_off[l + 1] - A.level_off[l], g->stream));
  }
  g->stats.n_factorize++;
  return PPS_OK;
}

// Wait for the result record with sequence number `seq`: spin on the pinned word the chi2 kernel writes
// last (a few microseconds), falling back to a stream sync if it does not show up (launch failure).
int wait_result(pps_graph* g, volatile double* slot, double seq, hipStream_t producer = nullptr) {
  const double t0 = now_s();
  unsigned spins = 0;
  while (slot[3] != seq) {
    if ((++spins & 0x3ff) == 0 && now_s() - t0 > 0.5) {
      HIP_TRY(g, hipStreamSynchronize(producer ? producer : g->stream));
      if (slot[3] != seq) return fail(g, PPS_EHIP, "result record did not arrive");
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // status word of the solve that produced the record: 0 ok | 1 not positive definite | >= 64 an internal hand-over of the
  // back-substitution timed out (flow_wait, pps_k3.hip) -- the numbers of this record are not a solution
  if (slot[2] >= kStatusInternal) return fail(g, PPS_EHIP, "internal error: a hand-over flag between the waves or workgroups of a K3 launch never arrived (the step was discarded)");
  return PPS_OK;
}

// est <-> lin by pointer: a rejected LM trial (estimate_to_linpoint, Optimizer.cpp:454) and the final
// linpoint_to_estimate (:466) need no data movement because the other copy is dead afterwards
void swap_state(pps_graph* g) {
  std::swap(g->dev.pose_est, g->dev.pose_lin);
  std::swap(g->dev.plane_est, g->dev.plane_lin);
}

// est_to_lin (estimate_to_linpoint): every caller goes on to overwrite the whole estimate -- a Gauss-Newton step retracts into it,
// an LM solve uses it as the target of its first trial -- before anything reads it, and puts it back from lin when the step
// fails: the two copies trade places by pointer, no data moves.  lin -> est (that putting back) is a real copy.
int copy_state(pps_graph* g, bool est_to_lin) {
  const DevGraph& d = g->dev;
  if (est_to_lin) { swap_state(g); return PPS_OK; }
  HIP_TRY(g, hipMemcpyAsync(d.pose_est, d.pose_lin, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  return PPS_OK;
}

// est -> lin as data (the diagnostic entry points that linearise outside a solve and leave the estimate in place)
int linpoint_from_estimate(pps_graph* g) {
  const DevGraph& d = g->dev;
  HIP_TRY(g, hipMemcpyAsync(d.pose_lin, d.pose_est, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  return PPS_OK;
}

// chi2 (and |delta|^2, not-PD flag) -> host
int read_result(pps_graph* g, bool at_estimate, double* chi2, double* dnorm, bool* notpd) {
  if (g->n_live_factors == 0) { *chi2 = 0.0; if (dnorm) *dnorm = 0.0; if (notpd) *notpd = false; return PPS_OK; }
  {
    PhaseTimer t(g, &g->stats.t_retract_chi2);
    HIP_TRY(g, launch_chi2(g->dev, at_estimate, g->host_result, 0.0, g->stream));
  }
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  *chi2 = g->host_result[0];
  if (dnorm) *dnorm = std::sqrt(g->host_result[1]);
  if (notpd) *notpd = g->host_result[2] != 0.0;
  if (g->host_result[2] >= kStatusInternal) return fail(g, PPS_EHIP, "internal error: a hand-over flag between the waves or workgroups of a K3 launch never arrived (the step was discarded)");
  return PPS_OK;
}

// after the final stream sync of a solve: fold the K1 event pairs into stats.t_linearize
void resolve_k1_events(pps_graph* g) {
  for (int k = 0; k + 1 < g->k1_used; k += 2) {
    float ms = 0;
    if ((size_t)(k / 2) < g->k1_skip.size() && g->k1_skip[k / 2]) continue;       // a speculative K1 that left at its guard
    if (hipEventElapsedTime(&ms, g->k1_events[k], g->k1_events[k + 1]) == hipSuccess) g->stats.t_linearize += 1e-3 * ms;
  }
  g->k1_used = 0;
  for (int k = 0; k + 1 < g->fk_used; k += 2) {                // factor launches of the dual loop (two factorisations per launch)
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->fk_events[k], g->fk_events[k + 1]) == hipSuccess) g->stats.t_factor += 1e-3 * ms;
  }
  g->fk_used = 0;
}

void reset_solve_stats(pps_graph* g) {
  pps_stats& s = g->stats;
  s.t_linearize = s.t_assemble = s.t_factor = s.t_backsolve = s.t_retract_chi2 = 0;
  s.n_linearize = s.n_factorize = 0;
  s.lm_iterations = s.lm_trials_accepted = s.lm_trials_rejected = s.lm_trials_notpd = 0;
  s.t_analysis = s.t_upload = 0;
  s.n_launches = 0;
  g->launches0 = launch_count();
}

// the device copy of a handle is given up after a failed solve: streams drained, nothing on the device is trusted any more --
// the next upload sends the whole arena (up_unknown: the mirror says nothing about the device, measurements included) and the
// estimate falls back to the host's node values
void abandon_device_copy(pps_graph* g) {
  if (!g->dev_ready) return;
  (void)hipStreamSynchronize(g->stream);
  g->topo_dirty = true; g->dev_values_newer = false; g->dev_meas_newer = false;
  g->up_unknown = true; g->up_unknown_meas = true; g->pk_meas_ok = false; g->status_clean = false;
}

// A failure in the middle of a solve (a HIP error: lost device, out of memory) leaves est / lin possibly exchanged and
// speculative work in flight.  Both streams are drained and the device copy is abandoned: the next call uploads again from
// the host's node values -- the estimate falls back to the last state the host has seen -- instead of reading half-updated
// buffers.  (PPS_ENOTPD is not such a failure: the solve ran to its end.)
}  // namespace pps_impl

extern "C" {

static int update_impl(pps_graph* g);
// estimate_to_linpoint is a pointer trade (copy_state): a HIP failure behind it would leave est / lin exchanged with est holding
// dead data -- like the LM drivers, a failed update abandons the device copy, and the next call uploads from the host's values.
int pps_update(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  const int rc = update_impl(g);
  if (rc != PPS_OK && rc != PPS_ENOTPD) abandon_device_copy(g);
  return rc;
}
static int update_impl(pps_graph* g) {
  const double t0 = now_s();
  reset_solve_stats(g);
  if (g->n_live_nodes > 0 && g->n_live_factors == 0) return PPS_OK;   // no factor, no step
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  if (!g->status_clean) {                                              // (else: zero since the upload / the last chi2 kernel)
    HIP_TRY(g, launch_clear_status(g->dev, g->stream));
    // the flag stands for BOTH records: a one-step LM solve may have left a not-PD flag of a speculative factorisation that
    // was never evaluated in the second one, and this call sets status_clean again at its end
    if (g->spec_result) HIP_TRY(g, hipMemsetAsync(g->spec_result, 0, 4 * sizeof(double), g->stream));
  }
  g->status_clean = false;
  rc = copy_state(g, true); if (rc != PPS_OK) return rc;          // estimate_to_linpoint (Optimizer.cpp:116)
  rc = do_linearize(g); if (rc != PPS_OK) return rc;              // jacobian() (:119)
  rc = do_solve(g, 0.0); if (rc != PPS_OK) return rc;             // compute_gauss_newton_step, lambda = 0 (:122)
  // apply_exmap (:183) and chi2 at the new estimate in ONE launch (round 6: the trial kernel of the LM loop with one trial -- the retraction
  // blocks write est <- lin (+) delta, the chi2 blocks evaluate at lin (+) delta on the fly, same bits; it was k_retract + k_chi2)
  double chi2 = 0.0, dn = 0.0; bool notpd = false;
  if (g->n_live_factors > 0 && g->dev.n_pose + g->dev.n_plane > 0) {
    const DevGraph& d = g->dev;
    { PhaseTimer t(g, &g->stats.t_retract_chi2);
      HIP_TRY(g, launch_trial_dual(d, DualAlt{}, d.pose_lin, d.plane_lin, d.pose_est, d.plane_est, nullptr, nullptr, g->host_result, 0.0, nullptr, 0.0, g->stream, 1)); }
    rc = enqueue_state_download(g); if (rc != PPS_OK) return rc;   // (arrives with the synchronisation below)
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    chi2 = g->host_result[0]; dn = std::sqrt(g->host_result[1]); notpd = g->host_result[2] != 0.0;
    if (g->host_result[2] >= kStatusInternal) return fail(g, PPS_EHIP, "internal error: a hand-over flag between the waves or workgroups of a K3 launch never arrived (the step was discarded)");
  } else {
    { PhaseTimer t(g, &g->stats.t_retract_chi2); HIP_TRY(g, launch_retract_apply(g->dev, g->stream)); }
    rc = enqueue_state_download(g); if (rc != PPS_OK) return rc;
    rc = read_result(g, true, &chi2, &dn, &notpd); if (rc != PPS_OK) return rc;
  }
  resolve_k1_events(g);
  if (notpd) {
    // the step is garbage: put the estimate back (lin still holds it) instead of handing NaNs to the caller
    rc = copy_state(g, false); if (rc != PPS_OK) return rc;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
    return fail(g, PPS_ENOTPD, "normal equations not positive definite");
  }
  g->dev_values_newer = true;
  state_download_arrived(g);
  g->status_clean = true;                                         // the chi2 kernel took the flag with it
  g->stats.chi2_final = chi2; g->stats.last_delta_norm = dn; g->stats.lambda_final = 0;
  g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
  return PPS_OK;
}

static int lm_solve(pps_graph* g, int* iterations);

int pps_batch_optimize(pps_graph* g, int* iterations) {
  if (!g) return PPS_EINVAL;
  if (g->n_live_nodes > 0 && g->n_live_factors == 0) {          // nothing to optimise: chi2 = 0 ends LM before its first trial
    reset_solve_stats(g);
    g->tr_lambda.clear(); g->tr_chi2.clear(); g->tr_acc.clear();
    if (iterations) *iterations = 0;
    return PPS_OK;
  }
  const int rc = lm_solve(g, iterations);
  if (rc != PPS_OK && rc != PPS_ENOTPD) abandon_device_copy(g);
  return rc;
}

// Optimizer::levenberg_marquardt (Optimizer.cpp:371-467) with both candidate steps of a linearisation in the same launches.
// A rejected trial only changes lambda (same J, same H), so every solve factors H for lambda AND for lambda * factor
// (blockIdx.y of the band kernels, second L / U / delta set), applies both steps to two spare copies of the state and reduces
// both chi2 values into two pinned records.  One stream, no events: 10 launches per linearisation instead of 21 on two streams.
// The host walks the reference's lambda schedule over the records: an accepted step rotates its copy in as the new
// linearisation point, a first rejection finds the next trial's verdict already on the host.  Arithmetic, lambda schedule and
// LM trace are exactly those of the one-step-at-a-time loop below.
static int lm_solve_dual(pps_graph* g, int* iterations, double t0) {
  const pps_props& prop = g->props;
  const Analysis& A = g->an;
  if (!g->status_clean) {          // (else: both records are zero since the upload, or the last solve's chi2 kernels took the flags)
    HIP_TRY(g, launch_clear_status(g->dev, g->stream));
    HIP_TRY(g, hipMemsetAsync(g->spec_result, 0, 4 * sizeof(double), g->stream));
  }
  g->status_clean = false;
  int num_iter = 0;
  double lambda = prop.lm_lambda0;
  double* slot0 = g->host_result;                                   // chi2 at the linearisation point
  double* slot[2] = {g->host_result + 4, g->host_result + 8};       // trial for lambda / for lambda * factor
  DevGraph& d = g->dev;
  int rc = copy_state(g, true); if (rc != PPS_OK) return rc;       // estimate_to_linpoint (Optimizer.cpp:376): est is dead from here on
  // three state copies: x = the linearisation point (d.pose_lin), t[0] / t[1] = x (+) delta for the two damping values
  double *t_pose[2] = {d.pose_est, g->spec_pose}, *t_plane[2] = {d.plane_est, g->spec_plane};
  double seqs[2] = {0, 0};
  // Speculation costs what it computes: on a graph whose lower tree levels fill the GPU by themselves (C3: 5 359 fronts) the second
  // factorisation + back-substitution of a launch set are extra time, not idle lanes -- and LM accepts most steps there.  Such a
  // graph factors the second damping value only while LM zig-zags (after a rejection); the trace is the same either way.
  const bool adaptive = A.n_fronts >= 2048;
  bool use_alt = !adaptive;
  bool have_next = true;                 // trial 1 of the last launch is the step for the next lambda after a rejection
  // Round 6: the linearisation at the trial point LM is expected to accept rides in the trial launch, and K2 follows at once (SpecLin,
  // pps_device.h): when the host has seen the verdict and the prediction held, that point's J / H exist and become the current ones by
  // pointer; a wrong prediction launches the linearisation the plain way.  Graphs whose K1 is not the lane form over plain plane
  // observations keep the linearisation queued behind a device-side guard (LinGuard).
  // ... as do graphs whose sweep does not fit the chip's wave slots beside the trial blocks (k_trial_lin holds three waves per SIMD: 3 072
  // slots; C2 needs 2 700): a second round of waves costs more than the launch boundary saved (C3, same box: 384 against 365 us per iteration).
  const bool fuse_lin = !g->sw.no_spec_lin && g->spec_J != nullptr && trial_lin_ok(d, prop.jacobian_mode) && trial_lin_waves(d) <= 3072;
  int fused_pair = -1;                   // K1 event pair of the last fused launch (profiling level 1)
  int pred = 0, pred_launched = 0;       // the trial predicted to be accepted: the one that was accepted last | the one the last launch linearised
  // trials + the predicted linearisation + its H blocks: what ends every launch set in this mode
  bool error_known = false;              // `error` holds chi2 at the linearisation point (not yet while the first launch set is queued)
  double error = 0.0;
  auto enqueue_trial_lin = [&](const DualAlt& alt, double s0, double s1) -> int {
    pred_launched = use_alt ? pred : 0;
    const SpecLin sl{g->spec_J, g->spec_P, g->spec_H, g->spec_Hf, pred_launched};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    fused_pair = -1;
    if (g->profiling == 1) {
      if (g->k1_used + 2 > (int)g->k1_events.size()) for (int k = 0; k < 2; k++) { hipEvent_t e; HIP_TRY(g, hipEventCreate(&e)); g->k1_events.push_back(e); }
      g->k1_skip.resize(g->k1_events.size() / 2, 0);
      fused_pair = g->k1_used / 2;
      g->k1_skip[fused_pair] = 1;          // (counted once the predicted trial is accepted: its sweep was a linearisation the solve asked for)
      e0 = g->k1_events[g->k1_used]; e1 = g->k1_events[g->k1_used + 1]; g->k1_used += 2;
    }
    HIP_TRY(g, launch_trial_lin(d, alt, sl, d.pose_lin, d.plane_lin, t_pose[0], t_plane[0], t_pose[1], t_plane[1], slot[0], s0, slot[1], s1, g->stream, e0, e1));
    // (K2 of that linearisation leaves at once where the records say the prediction failed: 2 us instead of 7 in front of the plain sweep)
    const LinGuard gd{{d.result_dev, g->spec_result}, {nullptr, nullptr}, {nullptr, nullptr}, error, error_known ? 1 : 0, use_alt ? 0 : 1, pred_launched};
    HIP_TRY(g, launch_hblocks_spec(d, sl, g->stream, &gd));
    return PPS_OK;
  };
  auto enqueue_dual = [&](double lam) -> int {
    DualAlt alt{g->spec_L, g->spec_U, g->spec_delta, g->spec_result, g->spec_chi2_partials, g->spec_dn_partials, g->spec_ticket,
                lam * prop.lm_lambda_factor};
    have_next = use_alt;
    if (!use_alt) {
      // one damping value: the single-lambda launches, and the trial kernel with ONE trial (round 6: it used to walk both copies, the second
      // one with a stale delta that nothing read -- twice the retraction and chi2 work of a launch that C3 pays 22 us for; have_next is false,
      // the second record is never waited for)
      { const int rc2 = enqueue_factor_solve(g, d, nullptr, lam, g->stream, g->profiling == 1); if (rc2 != PPS_OK) return rc2; }
      g->stats.n_factorize += 1;
      g->seq += 1.0; seqs[0] = g->seq;
      g->seq2 += 1.0; seqs[1] = g->seq2;
      if (fuse_lin) return enqueue_trial_lin(alt, seqs[0], seqs[1]);
      HIP_TRY(g, launch_trial_dual(d, alt, d.pose_lin, d.plane_lin, t_pose[0], t_plane[0], t_pose[1], t_plane[1], slot[0], seqs[0], slot[1], seqs[1],
                                   g->stream, 1));
      return PPS_OK;
    }
    { const int rc2 = enqueue_factor_solve(g, d, &alt, lam, g->stream, g->profiling == 1); if (rc2 != PPS_OK) return rc2; }
    g->stats.n_factorize += 2;
    g->seq += 1.0; seqs[0] = g->seq;
    g->seq2 += 1.0; seqs[1] = g->seq2;
    if (fuse_lin) return enqueue_trial_lin(alt, seqs[0], seqs[1]);
    HIP_TRY(g, launch_trial_dual(d, alt, d.pose_lin, d.plane_lin, t_pose[0], t_plane[0], t_pose[1], t_plane[1], slot[0], seqs[0], slot[1], seqs[1],
                                 g->stream));
    return PPS_OK;
  };
  rc = do_linearize(g); if (rc != PPS_OK) return rc;               // jacobian() (:379)
  g->seq += 1.0;
  const double seq0 = g->seq;
  HIP_TRY(g, launch_chi2(d, false, slot0, seq0, g->stream));       // r = weighted_errors(LINPOINT); error = |r|^2 (:382-385)
  // Accept-branch speculation: the relinearisation that follows an accepted step is queued behind the trials before their
  // verdict is known; its kernels apply the accept test themselves (LinGuard) and pick the accepted copy, so the device does
  // not idle for the host round trip between chi2 and K1.
  // (not on a graph that fills the GPU by itself -- `adaptive` --: there the queued launches cost more than the host round trip they hide; C3, same
  // box, three runs each: 358-361 us per LM iteration with them, 352-354 without)
  const bool spec_lin = !g->sw.no_spec_lin && !fuse_lin && !adaptive;
  int spec_pair = -1;                    // K1 event pair of the queued speculative linearisation
  auto enqueue_spec_lin = [&](double err) -> int {
    if (!spec_lin) return PPS_OK;
    LinGuard gd{{d.result_dev, g->spec_result}, {t_pose[0], t_pose[1]}, {t_plane[0], t_plane[1]}, err, 1, have_next ? 0 : 1};
    spec_pair = g->profiling == 1 ? g->k1_used / 2 : -1;
    return do_linearize(g, &gd);
  };
  auto drop_spec_lin = [&]() { if (spec_pair >= 0 && (size_t)spec_pair < g->k1_skip.size()) g->k1_skip[spec_pair] = 1; spec_pair = -1; };
  rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot0, seq0); if (rc != PPS_OK) return rc;
  error = slot0[0]; error_known = true;
  g->stats.chi2_initial = error;
  rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot[0], seqs[0]); if (rc != PPS_OK) return rc;
  int cur = 0;                           // which of the two trials the loop is looking at
  double dnorm = std::sqrt(slot[0][1]);
  bool last_notpd = slot[0][2] != 0.0;
  int n_notpd = last_notpd ? 1 : 0;
  bool trial_taken = false;              // the loop ended on an accepted, converged step: the estimate is that trial
  while ((prop.max_iterations <= 0 || num_iter < prop.max_iterations) && dnorm > prop.epsilon2 && error > prop.epsilon_abs) {
    num_iter++;
    const double error_new = slot[cur][0];
    const double error_diff = error - error_new;
    const bool accepted = error_diff > 0.;
    g->tr_lambda.push_back(lambda); g->tr_chi2.push_back(error_new); g->tr_acc.push_back(accepted ? 1 : 0);
    if (prop.verbose) fprintf(stderr, "LM Iteration %d: (lambda=%g) %s %.12g\n", num_iter, lambda, accepted ? "residual:" : "rejected", error_new);
    if (accepted) {
      g->stats.lm_trials_accepted++;
      if (error_diff < prop.epsilon_rel * error) { error = error_new; trial_taken = true; break; }   // (:431-434)
      lambda /= prop.lm_lambda_factor;
      error = error_new;
      if (adaptive) use_alt = cur != 0;                           // (accepted at once: no speculation next time; after a rejection: keep it)
      pred = cur;
      // the accepted copy becomes the linearisation point; the old one is the spare now
      std::swap(d.pose_lin, t_pose[cur]); std::swap(d.plane_lin, t_plane[cur]);
      if (fuse_lin && cur == pred_launched) {                      // relinearise (:444): done beside the trials, at this very point
        std::swap(d.J, g->spec_J); std::swap(d.P, g->spec_P); std::swap(d.H, g->spec_H); std::swap(d.Hf, g->spec_Hf);
        g->stats.n_linearize++;
        if (fused_pair >= 0 && (size_t)fused_pair < g->k1_skip.size()) g->k1_skip[fused_pair] = 0;
      }
      else if (fuse_lin) { rc = do_linearize(g); if (rc != PPS_OK) return rc; }      // (the other trial was accepted: the plain way)
      else if (spec_lin) { g->stats.n_linearize++; spec_pair = -1; }    // ... queued already, at this very copy
      else { rc = do_linearize(g); if (rc != PPS_OK) return rc; }
      rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;      // (:458)
      rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
      cur = 0;
      rc = wait_result(g, slot[0], seqs[0]); if (rc != PPS_OK) return rc;
    } else {
      g->stats.lm_trials_rejected++;
      lambda *= prop.lm_lambda_factor;                             // estimate_to_linpoint (:454): x was never overwritten
      if (have_next) {                                             // computed alongside: nothing to launch
        cur = 1; have_next = false;
        rc = wait_result(g, slot[1], seqs[1]); if (rc != PPS_OK) return rc;
      } else {
        drop_spec_lin();                                           // both trials rejected: its kernels left J and H alone
        if (adaptive) use_alt = true;                              // LM is zig-zagging: the next rejection should be free again
        rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;    // (:458), same J and H
        rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
        cur = 0;
        rc = wait_result(g, slot[0], seqs[0]); if (rc != PPS_OK) return rc;
      }
    }
    dnorm = std::sqrt(slot[cur][1]);
    last_notpd = slot[cur][2] != 0.0;
    n_notpd += last_notpd ? 1 : 0;
  }
  // linpoint_to_estimate (:466): the estimate is the accepted trial, or the linearisation point when the pending step is dropped
  drop_spec_lin();                       // (a linearisation queued behind the last trials is not one the solve asked for)
  if (trial_taken) { std::swap(d.pose_lin, t_pose[cur]); std::swap(d.plane_lin, t_plane[cur]); }
  d.pose_est = d.pose_lin; d.plane_est = d.plane_lin;
  d.pose_lin = t_pose[0]; d.plane_lin = t_plane[0];
  g->spec_pose = t_pose[1]; g->spec_plane = t_plane[1];
  { const int rc2 = enqueue_state_download(g); if (rc2 != PPS_OK) return rc2; }
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->dev_values_newer = true;
  state_download_arrived(g);
  g->status_clean = true;                // every dual solve was followed by both chi2 kernels
  resolve_k1_events(g);
  g->stats.lm_iterations = num_iter;
  g->stats.chi2_final = error; g->stats.lambda_final = lambda; g->stats.last_delta_norm = dnorm;
  g->stats.lm_trials_notpd = n_notpd;
  g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
  if (iterations) *iterations = num_iter;
  if (last_notpd) return fail(g, PPS_ENOTPD, "normal equations not positive definite at the last LM trial");
  return PPS_OK;
}

static int lm_solve(pps_graph* g, int* iterations) {
  const double t0 = now_s();
  reset_solve_stats(g);
  g->tr_lambda.clear(); g->tr_chi2.clear(); g->tr_acc.clear();
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  if (g->use_band && g->profiling < 2 && !g->dev.trace && !g->sw.no_dual) return lm_solve_dual(g, iterations, t0);      // (PPS_NO_DUAL: the loop-forms parity test)
  const pps_props& prop = g->props;
  if (!g->status_clean) HIP_TRY(g, launch_clear_status(g->dev, g->stream));
  g->status_clean = false;
  int num_iter = 0;
  double lambda = prop.lm_lambda0;
  double* slot0 = g->host_result;       // chi2 at the linearisation point
  double* slot1 = g->host_result + 4;   // the trial: |delta|^2 of the step and chi2 after it
  // One stream, one result record per LM trial.  After every solve the trial step is applied at once (est <- lin,
  // lin <- lin (+) delta) and its chi2 is reduced, so a single record carries everything the loop condition and the accept
  // test need; a rejected trial is undone by exchanging the two copies (pointers), and if the loop ends on |delta| <= eps2
  // the pending step is undone the same way.  This is the reference's loop one step at a time: the form the profiling levels,
  // the phase trace and the graphs beyond the band kernels (dense fronts, level-per-launch fallback) run; band graphs take
  // lm_solve_dual.
  auto enqueue_trial = [&](double lam) -> int {
    int r = do_solve(g, lam); if (r != PPS_OK) return r;                       // compute_gauss_newton_step (:395,458)
    PhaseTimer t(g, &g->stats.t_retract_chi2);
    HIP_TRY(g, launch_retract_trial(g->dev, g->stream));                       // linpoint_to_estimate + self_exmap (:414-416)
    g->seq += 1.0;
    HIP_TRY(g, launch_chi2_trial(g->dev, slot1, g->seq, g->stream));           // weighted_errors(LINPOINT) (:417)
    return PPS_OK;
  };
  rc = copy_state(g, true); if (rc != PPS_OK) return rc;          // estimate_to_linpoint (Optimizer.cpp:376)
  rc = do_linearize(g); if (rc != PPS_OK) return rc;              // jacobian() (:379)
  g->seq += 1.0;
  const double seq0 = g->seq;
  HIP_TRY(g, launch_chi2(g->dev, false, slot0, seq0, g->stream)); // r = weighted_errors(LINPOINT); error = |r|^2 (:382-385)
  rc = enqueue_trial(lambda); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot0, seq0); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot1, g->seq); if (rc != PPS_OK) return rc;
  double error = slot0[0];
  g->stats.chi2_initial = error;
  double dnorm = std::sqrt(slot1[1]);
  // Not-PD is a property of ONE factorisation (one lambda): every result record carries the flag of the solve that produced
  // its step, and the chi2 kernel clears it.  CHOLMOD is silent here and LM simply rejects such a step and raises lambda
  // (Optimizer.cpp:448-455), so only a solve whose LAST trial was still not PD reports PPS_ENOTPD.
  bool last_notpd = slot1[2] != 0.0;
  int n_notpd = last_notpd ? 1 : 0;
  bool trial_pending = true;
  while ((prop.max_iterations <= 0 || num_iter < prop.max_iterations) && dnorm > prop.epsilon2 && error > prop.epsilon_abs) {
    num_iter++;
    const double error_new = slot1[0];
    const double error_diff = error - error_new;
    const bool accepted = error_diff > 0.;
    g->tr_lambda.push_back(lambda); g->tr_chi2.push_back(error_new); g->tr_acc.push_back(accepted ? 1 : 0);
    if (prop.verbose) fprintf(stderr, "LM Iteration %d: (lambda=%g) %s %.12g\n", num_iter, lambda, accepted ? "residual:" : "rejected", error_new);
    if (accepted) {
      g->stats.lm_trials_accepted++;
      if (error_diff < prop.epsilon_rel * error) { error = error_new; trial_pending = false; break; }   // (:431-434)
      lambda /= prop.lm_lambda_factor;
      error = error_new;
      rc = do_linearize(g); if (rc != PPS_OK) return rc;          // relinearise around the accepted point (:444)
    } else {
      g->stats.lm_trials_rejected++;
      lambda *= prop.lm_lambda_factor;
      swap_state(g);                                              // estimate_to_linpoint: restore (:454)
    }
    rc = enqueue_trial(lambda); if (rc != PPS_OK) return rc;      // (:458)
    rc = wait_result(g, slot1, g->seq); if (rc != PPS_OK) return rc;
    dnorm = std::sqrt(slot1[1]);
    last_notpd = slot1[2] != 0.0;
    n_notpd += last_notpd ? 1 : 0;
  }
  if (trial_pending) swap_state(g);                               // undo the pending step
  swap_state(g);                                                  // linpoint_to_estimate (:466)
  { const int rc2 = enqueue_state_download(g); if (rc2 != PPS_OK) return rc2; }
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->dev_values_newer = true;
  state_download_arrived(g);
  g->status_clean = true;                                         // every solve was followed by its chi2 kernel
  resolve_k1_events(g);
  g->stats.lm_iterations = num_iter;
  g->stats.chi2_final = error; g->stats.lambda_final = lambda; g->stats.last_delta_norm = dnorm;
  g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
  if (g->dev.trace) {
    const Analysis& A = g->an;
    std::vector<long long> tr((size_t)A.n_fronts * 8);
    (void)hipMemcpy(tr.data(), g->dev.trace, tr.size() * 8, hipMemcpyDeviceToHost);
    if (g->dev.trace_solve) {
      // PPS_TRACE=2: the slots hold the back-substitution of the last solve (parents before children)
      for (int l = A.n_levels - 1; l >= 0; l--) {
        double ph[5] = {0, 0, 0, 0, 0}, gap = 0; int n = 0, ng = 0;
        for (int s2 = 0; s2 < A.n_fronts; s2++) {
          if (A.f_level[s2] != l) continue;
          n++;
          for (int k = 0; k < 5; k++) ph[k] += (double)(tr[(size_t)s2 * 8 + k + 1] - tr[(size_t)s2 * 8 + k]);
          const int par = A.f_parent[s2];
          if (par >= 0) { gap += (double)(tr[(size_t)s2 * 8] - tr[(size_t)par * 8 + 5]); ng++; }
        }
        if (!n) continue;
        fprintf(stderr, "  solve level %d (%d fronts): panel load %.0f boundary values %.0f y - L_B^T x_b %.0f back-substitution %.0f store %.0f | start after parent's end %.0f\n",
                l, n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ng ? gap / ng : 0.0);
      }
    } else {                        // (PPS_TRACE=1; either way the common epilogue below reports iterations and the not-PD status)
    double acc[5] = {0, 0, 0, 0, 0};
    std::vector<double> lvl_tot(A.n_levels, 0.0); std::vector<int> lvl_n(A.n_levels, 0);
    for (int s2 = 0; s2 < A.n_fronts; s2++) {
      for (int k = 0; k < 5; k++) acc[k] += (double)(tr[(size_t)s2 * 8 + k + 1] - tr[(size_t)s2 * 8 + k]);
      lvl_tot[A.f_level[s2]] += (double)(tr[(size_t)s2 * 8 + 5] - tr[(size_t)s2 * 8]); lvl_n[A.f_level[s2]]++;
    }
    { double pn = 0, tr2 = 0; for (int s2 = 0; s2 < A.n_fronts; s2++) { pn += (double)tr[(size_t)s2 * 8 + 6]; tr2 += (double)tr[(size_t)s2 * 8 + 7]; }
      fprintf(stderr, "PPS_TRACE elimination split: panel %.0f trailing %.0f cycles per front\n", pn / A.n_fronts, tr2 / A.n_fronts); }
    fprintf(stderr, "PPS_TRACE mean cycles per front: zero %.0f gather %.0f extend-add %.0f eliminate %.0f store %.0f\n",
            acc[0] / A.n_fronts, acc[1] / A.n_fronts, acc[2] / A.n_fronts, acc[3] / A.n_fronts, acc[4] / A.n_fronts);
    for (int l = 0; l < A.n_levels; l++) fprintf(stderr, "  level %d: %d fronts, mean total %.0f cycles\n", l, lvl_n[l], lvl_tot[l] / std::max(1, lvl_n[l]));
    {
      // per level: the phases, and how long a front's start lies behind the end of its last child (barrier, launch boundary,
      // record load) -- the part of a tree level that no phase accounts for
      std::vector<long long> last_child_end(A.n_fronts, 0);
      for (int s2 = 0; s2 < A.n_fronts; s2++) if (A.f_parent[s2] >= 0) last_child_end[A.f_parent[s2]] = std::max(last_child_end[A.f_parent[s2]], tr[(size_t)s2 * 8 + 5]);
      for (int l = 0; l < A.n_levels; l++) {
        double ph[7] = {0, 0, 0, 0, 0, 0, 0}, gap = 0; int n = 0, ng = 0;
        for (int s2 = 0; s2 < A.n_fronts; s2++) {
          if (A.f_level[s2] != l) continue;
          n++;
          for (int k = 0; k < 5; k++) ph[k] += (double)(tr[(size_t)s2 * 8 + k + 1] - tr[(size_t)s2 * 8 + k]);
          ph[5] += (double)tr[(size_t)s2 * 8 + 6]; ph[6] += (double)tr[(size_t)s2 * 8 + 7];
          if (last_child_end[s2] > 0) { gap += (double)(tr[(size_t)s2 * 8] - last_child_end[s2]); ng++; }
        }
        if (!n) continue;
        fprintf(stderr, "  level %d: zero %.0f gather %.0f extend-add %.0f eliminate %.0f (panel %.0f trailing %.0f) store %.0f | start after last child's end %.0f\n", l,
                ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[5] / n, ph[6] / n, ph[4] / n, ng ? gap / ng : 0.0);
      }
    }
    {
      double w[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nw2 = 0;
      for (int s2 = 0; s2 < A.n_fronts; s2++) {
        if (A.f_p[s2] + A.f_b[s2] + 1 <= 64) continue;
        nw2++;
        for (int k = 0; k < 5; k++) w[k] += (double)(tr[(size_t)s2 * 8 + k + 1] - tr[(size_t)s2 * 8 + k]);
        w[5] += (double)tr[(size_t)s2 * 8 + 6]; w[6] += (double)tr[(size_t)s2 * 8 + 7];
      }
      if (nw2) fprintf(stderr, "  fronts beyond 64 rows (%d): zero %.0f gather %.0f extend-add %.0f eliminate %.0f (panel %.0f trailing %.0f) store %.0f cycles\n", nw2,
                       w[0] / nw2, w[1] / nw2, w[2] / nw2, w[3] / nw2, w[5] / nw2, w[6] / nw2, w[4] / nw2);
    }
    long long tmin = tr[0], tmax = tr[5];
    for (int s2 = 0; s2 < A.n_fronts; s2++) { tmin = std::min(tmin, tr[(size_t)s2 * 8]); tmax = std::max(tmax, tr[(size_t)s2 * 8 + 5]); }
    fprintf(stderr, "  first start -> last end: %lld cycles\n", tmax - tmin);
    }
  }
  if (iterations) *iterations = num_iter;
  g->stats.lm_trials_notpd = n_notpd;
  if (last_notpd) return fail(g, PPS_ENOTPD, "normal equations not positive definite at the last LM trial");
  return PPS_OK;
}

int pps_chi2(pps_graph* g, double* chi2) {
  if (!g || !chi2) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  double dn; bool np;
  return read_result(g, true, chi2, &dn, &np);
}

}  // extern "C"
