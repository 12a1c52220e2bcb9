// pps_device.h -- device-side data layout and kernel launchers of the graph solve.
//
// HBM layout (all fp64 unless noted; everything for one graph stays resident):
//   state        pose SoA [7][pose_ld] (tx,ty,tz,qx,qy,qz,qw), plane SoA [4][plane_ld]; two copies:
//                `est` (estimate, NodeT::_value) and `lin` (linearisation point, NodeT::_value0).
//   factors      SoA per type: int32 node indices, measurements, packed upper-triangular sqrtinf.
//   J            AoS per factor [J_a | J_b | r]: plane edge 30, odometry 78, pose prior 42, plane prior 12 doubles.
//   H            block-sparse J'J (lower triangle in elimination order) as per-segment partial blocks.
//   L / U        multifrontal factor panels and update matrices, offsets from the symbolic analysis.
//   delta        one scalar block per node, nodes in creation order (an offset never moves when the graph grows);
//                a front reaches its pivots through pidx, its boundary through bidx.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pps {

struct DevGraph {
  // ---- state ----
  int n_pose = 0, n_plane = 0, pose_ld = 0, plane_ld = 0;
  double *pose_est = nullptr, *pose_lin = nullptr;
  double *plane_est = nullptr, *plane_lin = nullptr;
  int *pose_voff = nullptr, *plane_voff = nullptr;   // scalar offset of each node in delta
  // ---- factors (SoA: value k of factor i at [k * ld + i]; the leading dimensions are capacities that grow in powers of two, so
  // a factor keeps its place when others are appended) ----
  int n_obs = 0, n_odo = 0, n_pp = 0, n_lp = 0;
  int obs_ld = 0, odo_ld = 0, pp_ld = 0, lp_ld = 0;
  int *obs_pose = nullptr, *obs_plane = nullptr; double *obs_meas = nullptr, *obs_w = nullptr;
  // plane edges [n_obs_fixed, n_obs) re-pop their measurement from two ground-edge rays at every evaluation
  // (Pose3d_Plane3d_Factor2); obs_ray is SoA [6][n_obs - n_obs_fixed]
  int n_obs_fixed = 0; double* obs_ray = nullptr;
  int *odo_a = nullptr, *odo_b = nullptr;         double *odo_meas = nullptr, *odo_w = nullptr;
  int *pp_pose = nullptr;                          double *pp_meas = nullptr, *pp_w = nullptr;
  int *lp_plane = nullptr;                         double *lp_meas = nullptr, *lp_w = nullptr;
  // ---- linear system ----
  double* J = nullptr;
  int64_t joff_obs = 0, joff_odo = 0, joff_pp = 0, joff_lp = 0;
  // product records of the plane observations (pps_symbolic.h: kPSize): slot i at P + poff_obs + 54 i = [J_p' J_p 36 | -J_p' r 6 | J_l' J_l 9 | -J_l' r 3]
  double* P = nullptr;
  int64_t poff_obs = 0;
  double* H = nullptr;       // segment slots
  double* L = nullptr;
  double* U = nullptr;
  double* delta = nullptr;   // n_scalars
  int n_scalars = 0;
  // symbolic arrays (device copies of pps::Analysis)
  int n_fronts = 0, n_levels = 0, max_front = 0, n_segs = 0, n_blocks = 0;
  int *f_p = nullptr, *f_b = nullptr, *f_poff = nullptr;
  int* pidx = nullptr;       // elimination-ordered scalar -> index in delta (pivots of front s: pidx[f_poff[s] .. + f_p[s]))
  int64_t *f_Loff = nullptr, *f_Uoff = nullptr;
  int *f_bidx_off = nullptr, *bidx = nullptr;
  int *f_child_off = nullptr, *child = nullptr;
  int *f_cmap_off = nullptr, *cmap = nullptr;
  int *level_fronts = nullptr;
  int *f_asm_off = nullptr, *asm_blk = nullptr, *asm_lrow = nullptr, *asm_lcol = nullptr;
  int *asm_el0 = nullptr, *asm_fsz = nullptr;   // per assembled block: first index in Hf, rows of its front
  int *blk_rows = nullptr, *blk_cols = nullptr, *blk_size = nullptr, *blk_nseg = nullptr;
  int64_t* blk_hoff = nullptr;
  int *seg_blk = nullptr, *seg_c0 = nullptr, *seg_cnt = nullptr;
  int64_t* seg_hoff = nullptr;
  int* contrib = nullptr;
  int n_mseg = 0; int* mseg_blk = nullptr;   // blocks with more than one segment (pre-reduced by k_hreduce)
  // wave-per-front path: flat gather / scatter lists and the band schedule
  int *f_el_off = nullptr, *el_src = nullptr, *el_tgt = nullptr;
  int *blk_doff = nullptr, *blk_dst = nullptr;
  double* Hf = nullptr;      // H in front gather order: front s reads Hf[f_el_off[s] .. f_el_off[s+1])
  int64_t* f_ea_off = nullptr; int* ea_tgt = nullptr;
  int *grp_lvl_off = nullptr, *glvl_front_off = nullptr, *glvl_fronts = nullptr;
  int* grp_span = nullptr;          // per band group, 8 ints: first position in glvl order, fronts, local levels, first position of local levels 1 .. 4, 0
  int *frec = nullptr, *crec = nullptr, *srec = nullptr;   // packed metadata records (pps_symbolic.h)
  int* obs_dir = nullptr;                                   // direct (pose, plane) blocks: 3 ints per plane-observation slot (pps_symbolic.h)
  int* k3_flag = nullptr;                                   // hand-over flags between workgroups of the whole-tree launches (XGroup, pps_k3.hip): 4 x n_fronts ints
  int* nd_segs = nullptr; int n_nd_segs = 0;                // H segments that are not direct
  // K2 work lists (round 4): the non-direct segments of single-segment blocks (one wave each) and, per block of several
  // segments, its first segment (one workgroup each: the partial sums meet in LDS)
  int* k2_single = nullptr; int n_k2_single = 0;
  int* k2_multi = nullptr; int n_k2_multi = 0;                // pairs: first segment of a chunk of <= 16 segments | segments in the chunk + (1 << 16 when it is the whole block)
  int* k2_finish = nullptr; int n_k2_finish = 0;              // first segment of every block of more than 16 segments (its chunks' partial sums are added by k_hfinish)
  // throughput form of K2 (many-graph batches; built by pps_multi on first use): the non-direct segments by class -- entries
  // [0, n_k2t_big): segment | class << 28 (0 generic, 1 pose diagonal 6 x 6 + g with rows of 3 / 6, 2 pose-pose 6 x 6 with rows of 6),
  // then n_k2t_small plane diagonals (3 x 3 + g, rows of 3: four of them per wave)
  // (round 5: the entries with a class body come first, n_k2t_spec of them; the generic ones behind run in a kernel of their own)
  int* k2t = nullptr; int n_k2t_big = 0, n_k2t_small = 0, n_k2t_spec = 0;
  int *cls_off = nullptr, *cls_fronts = nullptr;             // level-per-launch lists (pps_symbolic.h)
  // ---- reductions / status ----
  double* chi2_partials = nullptr;   // one per block of the residual sweep
  int chi2_blocks = 0;
  double* dn_partials = nullptr;     // |delta|^2 per block of the retraction kernel
  unsigned int* ticket = nullptr;    // last-block election of the chi2 reduction
  double* result_dev = nullptr;      // [0] chi2, [1] |delta|^2, [2] not-PD flag (as double), [3] reserved
  long long* trace = nullptr;        // PPS_TRACE=1: 8 timestamps (s_memtime) per front of the last factorisation
  int trace_solve = 0;               // PPS_TRACE=2: the trace slots take the phases of the back-substitution instead of the factorisation's
  int no_strip = 0;                  // PPS_NO_STRIP=1: fronts of 65 .. 80 rows take the LDS-tile path (A/B, parity tests)
  unsigned sw = 0;                   // SW_* bits of the owning handle's Switches that the launchers consult (set by upload_all)
  // step quaternions of the numerical Jacobian's rotation / plane columns: (a, 0, 0, c) = rot_exp((eps, 0, 0)) and plane_exp((eps, 0, 0)),
  // evaluated ONCE per device by the device's own functions (step_constants) -- the same bits as evaluating them per step
  double step_ac[4] = {0, 0, 0, 0};
  double* gwork = nullptr;           // global-memory front workspace for fronts that exceed LDS
  int64_t gwork_stride = 0;
};

// ---- run-time switches ----------------------------------------------------------------------------------------------------
// Every PPS_* environment variable the library understands.  They are read ONCE per handle -- pps_graph_create / pps_graph_load /
// pps_multi_create / pps_popup_create ... call read_switches() -- and never on a launch path: a handle keeps the schedule it was
// created with whatever the environment does later (A/B tools and the parity tests set the variable before they create the handle).
constexpr double kStatusInternal = 64.0;   // result_dev[2] at or above this: an internal time-out inside a kernel (PPS_EHIP), not a not-PD pivot
enum : unsigned { SW_K1_THREAD_FORM = 1u, SW_NO_SOLVE_FLOW = 2u, SW_NO_ROOT_FUSE = 4u, SW_DEBUG_DROP_FLAG = 8u, SW_DEBUG_DROP_XFLAG = 16u };
// Fifteen variables (INTEGRATION.md lists them): each one selects a path that some graph shape takes anyway (so the parity tests can put every
// graph through it) or a diagnostic.  A/B switches of experiments whose losing side was deleted do not exist.
struct Switches {
  bool k1_thread_form = false;      // PPS_K1_THREAD_FORM: thread-per-factor K1 (no product records) on graphs of any size
  // PPS_PLAIN_SCHEDULE=<bits> (no value: all): the fall-back schedules that graphs of other shapes take, forced onto every graph --
  bool no_preassemble = false;      //   1  plain walk of a band group instead of k_band_factor_pre
  bool no_solve_flow = false;       //   2  barrier form of the band back-substitution
  bool no_root_fuse = false;        //   4  root stage as two launches
  bool split_expand = false;        //   8  list expansion as two launches + a fill
  bool no_spec_lin = false;         // PPS_NO_SPEC_LIN: nothing of the next linearisation is computed before the trials' verdict is known
  bool no_dual = false;             // PPS_NO_DUAL: the one-step-at-a-time LM loop
  bool no_strip = false;            // PPS_NO_STRIP: fronts of 65 .. 80 rows through the LDS-tile path
  bool no_incremental = false;      // PPS_NO_INCREMENTAL: every analysis from scratch
  bool no_incr_compact = false;     // PPS_NO_INCR_COMPACT: compacted tables rebuilt per analysis
  bool verify_upload = false;       // PPS_DEBUG_VERIFY_UPLOAD: read the arena back after every flush, compare every skipped prefix
  bool multi_levels = false, multi_thread_form = false;   // PPS_MULTI_LEVELS / PPS_MULTI_THREAD_FORM: the throughput forms on small batches
  int debug_drop_flag = 0;          // PPS_DEBUG_DROP_FLAG (tests of the time-out path): 1 = the data-flow back-substitution withholds the hand-over flag of every
                                    // group's top front (LDS), 2 = the whole-tree factor launch withholds the flag a group's top front raises for its parent's workgroup
  int trace = 0;                    // PPS_TRACE: 1 = phase timestamps of the factorisation, 2 = of the back-substitution
  // PPS_TIMING=<bits>: host-side timing printed to stderr -- 1 analysis phases, 2 upload phases, 4 pps_multi totals, 8 pps_multi rounds
  bool analysis_timing = false, upload_timing = false;
  int multi_timing = 0;             //   (0 off, 1 totals, 2 rounds)
  int multi_split = 0;              // PPS_MULTI_SPLIT: chunks a batch is cut into (0 = by size)
  long long multi_thread_factors = 120000;   // PPS_MULTI_THREAD_FACTORS: factors per chunk above which a batch takes the throughput forms (round 6: 120 000,
                                             // measured on C2-size graphs -- G = 16 (96 k factors) 1 198 graphs/s in the latency forms against 1 136, G = 24 (144 k)
                                             // 1 229 against 1 401, G = 32 1 384 against 1 581; it was 200 000)
  unsigned dev_bits() const {
    return (k1_thread_form ? SW_K1_THREAD_FORM : 0u) | (no_solve_flow ? SW_NO_SOLVE_FLOW : 0u) | (no_root_fuse ? SW_NO_ROOT_FUSE : 0u) |
           (debug_drop_flag == 1 ? SW_DEBUG_DROP_FLAG : 0u) | (debug_drop_flag == 2 ? SW_DEBUG_DROP_XFLAG : 0u);
  }
};
Switches read_switches();           // pps_api.cpp: the only place of the library that calls getenv

// Accept-branch speculation of the dual LM loop: K1 / K2 are queued right behind the two trials, before the host has seen
// their chi2.  The kernels repeat the host's accept test (Optimizer.cpp:425-428: error - error_new > 0, trial 0 first) on the
// device records and linearise at the accepted copy of the state -- or leave J and H alone when both trials were rejected.
struct LinGuard {
  const double* chi[2];      // result_dev of trial 0 / 1 ([0] = chi2 after the step)
  const double* pose[2];
  const double* plane[2];
  double error;              // chi2 at the linearisation point the trials started from
  int on, single;            // single: only trial 0 was computed (the second record is stale: never consulted)
  int want = -1;             // 0 / 1: run only if THAT trial is the accepted one (K2 of a speculative linearisation, SpecLin); -1: if either is
};

// Round 6 -- the next linearisation off the LM loop's dependency chain.  The launch that applies the two trial steps and reduces their chi2
// (k_trial_lin) also sweeps K1 at ONE of the two trial points, x (+) delta_which -- evaluated on the spot, no launch boundary behind the
// retraction -- into a spare set of J / P / H / Hf, and a K2 launch assembles that H.  `which` is the host's prediction of the trial LM will
// accept (the one it accepted last time: LM zig-zags between "rejected at lambda, accepted at 10 lambda" and runs of first-trial accepts).
// When the host has seen the verdict and the prediction held, the spare set trades places with the current one by pointer; otherwise the
// linearisation is launched the plain way.  (Both trial points in one launch were built first and measured: 4 600 waves do not fit the
// 3 072 - 4 096 wave slots of the chip at this kernel's register count, the launch took as long as the two it replaced -- DESIGN.md section 8.)
struct SpecLin {
  double *J, *P, *H, *Hf;
  int which;                 // trial whose point is linearised (0 / 1)
};

// kernel launches issued by the calling host thread (pps_stats::n_launches is the difference over a solve call)
unsigned long long launch_count();
void count_launch();
#define PPS_LAUNCH(...) do { ::pps::count_launch(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
// the same with a start / stop event that take the dispatch's own timestamps (hipExtLaunchKernelGGL; include <hip/hip_ext.h>)
#define PPS_LAUNCH_EV(ev0, ev1, kernel, grid, block, lds, st, ...)                                                     \
  do {                                                                                                                 \
    ::pps::count_launch();                                                                                             \
    if (ev0) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__);                            \
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                                \
  } while (0)

// fills ac[4] = {a_rot, c_rot, a_plane, c_plane} for the current device (one tiny kernel + a synchronous copy the first time)
hipError_t step_constants(double ac[4]);
// All launchers enqueue on `st` and return the HIP error of the launch.
hipError_t launch_linearize(const DevGraph& d, int mode, bool at_estimate, hipStream_t st, const LinGuard* guard = nullptr, hipEvent_t ev0 = nullptr,
                            hipEvent_t ev1 = nullptr);      // ev0 / ev1: start / stop of the sweep itself (profiling level 1)
// products: K1 wrote the plane observations' product records (see k1_products)
hipError_t launch_hblocks(const DevGraph& d, hipStream_t st, const LinGuard* guard = nullptr, bool products = true);
bool k1_lane_form(const DevGraph& d, int mode);      // which form launch_linearize takes for this graph and Jacobian mode
bool k1_products(const DevGraph& d, int mode);       // whether that launch writes the product records K2 sums
hipError_t launch_factor_level(const DevGraph& d, int level_begin, int level_count, int level_max_front, double lambda,
                               hipStream_t st);
hipError_t launch_backsolve_level(const DevGraph& d, int level_begin, int level_count, hipStream_t st);
// wave-per-front band kernels: one workgroup per group of the stage, `nwaves` fronts in flight per workgroup
// Two damping values of one linearisation in the same launches (blockIdx.y = 0 / 1): the second factorisation has its own
// L / U / delta, not-PD flag and reduction scratch, everything else (J, H, the index arrays) is shared.  A rejected LM trial only
// changes lambda, so the step for lambda * factor is computed next to the step for lambda (Optimizer.cpp:448-458).
struct DualAlt {
  double *L, *U, *delta, *result_dev, *chi2_partials, *dn_partials;
  unsigned int* ticket;
  double lambda;
};
// the whole tree in one factor launch + one back-substitution launch (k_band_factor_all / k_band_solve_all: hand-over between workgroups through
// DevGraph::k3_flag); epoch: the number of this launch pair (> 0, growing)
hipError_t launch_band_all(const DevGraph& d, const DualAlt* alt, int n_groups, int nwaves_factor, int nwaves_solve, int max_front, int max_panel,
                           int max_group_fronts, double lambda, int epoch, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int band_all_solve_waves(int max_panel, int max_group_fronts);
// pre: every group of the stage has the shape the pre-assembling walk needs (k_band_factor_pre)
hipError_t launch_band_factor(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_front, double lambda, hipStream_t st,
                              hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool pre = false);
hipError_t launch_band_solve(const DevGraph& d, int grp_begin, int grp_count, int nwaves, int max_panel, int max_group_fronts, hipStream_t st,
                             const DualAlt* alt = nullptr);
hipError_t launch_band_factor_dual(const DevGraph& d, const DualAlt& alt, int grp_begin, int grp_count, int nwaves, int max_front, double lambda,
                                   hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool pre = false);
// the last factor stage and the first back-substitution stage as one launch, where band_root_fusable says so
bool band_root_fusable(const DevGraph& d, int grp_count, int max_front);
hipError_t launch_band_root(const DevGraph& d, const DualAlt* alt, int grp, int nwaves_factor, int nwaves_solve, int max_front, int max_panel,
                            int max_group_fronts, double lambda, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,      // ev0 / ev1: the dispatch's start / stop (profiling level 1)
                            bool pre = false);                                                                                          // pre: the group fits the pre-assembling walk (stage_pre)
// both trials of a dual solve: out_k <- base (+) delta_k, chi2 and |delta|^2 of each into its own result record
hipError_t launch_trial_dual(const DevGraph& d, const DualAlt& alt, const double* base_pose, const double* base_plane, double* out_pose0,
                             double* out_plane0, double* out_pose1, double* out_plane1, double* host_result0, double seq0, double* host_result1,
                             double seq1, hipStream_t st, int n_trials = 2);
// (n_trials = 1: the first trial only -- the step of pps_update, retraction + chi2 in one launch; alt and the second set of arguments are not read)
// the same + K1 (lane form) at trial point sl.which into sl's buffers, one launch (k_trial_lin); ev0 / ev1: the dispatch's start / stop
hipError_t launch_trial_lin(const DevGraph& d, const DualAlt& alt, const SpecLin& sl, const double* base_pose, const double* base_plane, double* out_pose0,
                            double* out_plane0, double* out_pose1, double* out_plane1, double* host_result0, double seq0, double* host_result1,
                            double seq1, hipStream_t st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
int trial_lin_waves(const DevGraph& d);               // waves of that launch
bool trial_lin_ok(const DevGraph& d, int mode);      // the graph's K1 is the lane form over plain plane observations: k_trial_lin applies
// K2 of the speculative linearisation: J / P of the spare set -> its H / Hf; guard (want = sl.which): the launch leaves early where the
// trials' records say that another trial -- or none -- was accepted
hipError_t launch_hblocks_spec(const DevGraph& d, const SpecLin& sl, hipStream_t st, const LinGuard* guard = nullptr);
// ea_tgt (packed update matrix of a front -> packed index in its parent) expanded from cmap / f_cmap_off / f_ea_off
hipError_t launch_expand_ea(const DevGraph& d, int n_fronts, double* zero, size_t n_zero, int* ones, size_t n_ones, hipStream_t st);   // + the upload's two fills
// el_tgt / blk_dst (H block element <-> front-ordered H <-> packed front index) expanded from the per-block records;
// blk_dst must be filled with -1 beforehand
hipError_t launch_expand_el(const DevGraph& d, int n_asm, hipStream_t st);
hipError_t launch_expand_lists(const DevGraph& d, int n_fronts, int n_asm, double* zero, size_t n_zero, hipStream_t st);   // both + the zero fill, one launch
int band_max_rows();
bool band_level_solve_direct_ok(int p, int b);   // a front whose level back-substitution may load L_B straight into registers (kb_level_solve)
size_t band_solve_lds_bytes(int max_panel);      // max_panel = largest (f+1)*p of the stage
int band_front_limit();                         // largest front (scalars, without rhs row) of the band kernels
int debug_front_factor(int tiles, int strip, int p, int b, const double* A_host, double* L_host, double* U_host, double* not_pd);   // pps_debug_front_factor
int debug_exmap(int kind, int n, const double* x_host, const double* delta_host, double* out_host);   // pps_debug_exmap (pps_k4.hip)
int band_reg_rows();                            // fronts up to this many rows (incl. rhs) take the register-resident path
size_t band_lds_bytes(int max_front, bool reg_only_kernel = false);   // LDS bytes one wave needs for the factor kernel
// est <- lin ; lin <- lin (+) delta          (LM trial: Optimizer.cpp:414-416)
hipError_t launch_retract_trial(const DevGraph& d, hipStream_t st);
// est <- lin (+) delta                        (GN step: Optimizer.cpp:183)
hipError_t launch_retract_apply(const DevGraph& d, hipStream_t st);
// chi2 at lin (at_estimate=false) or est.  The last block writes {chi2, |delta|^2 (from the preceding
// retraction), not-PD flag} straight into `host_result` (pinned host memory)
hipError_t launch_chi2(const DevGraph& d, bool at_estimate, double* host_result, double seq, hipStream_t st);
hipError_t launch_clear_status(const DevGraph& d, hipStream_t st);
// chi2 of the one-step loop's trial, after launch_retract_trial: evaluated at est (+) delta on the spot (pps_k4.hip)
hipError_t launch_chi2_trial(const DevGraph& d, double* host_result, double seq, hipStream_t st);

// dense-front form (pps_dense.hip): fronts of hundreds of rows, every step spread over many workgroups.
// L must be zeroed before launch_dense_hpush; levels run leaves -> root (factor) and root -> leaves (solve).
int dense_front_max_pivots();
hipError_t launch_dense_hpush(const DevGraph& d, int max_el_per_front, double lambda, hipStream_t st);
// off_*: device prefix sums (level_count+1 entries) of the per-front work items of the three kernels of a level:
// 32x32 assembly tiles of the (f+1)-row lower triangle, 256-row panel slabs, 64x64 tiles of the (b+1)-row update matrix
hipError_t launch_dense_factor_level(const DevGraph& d, int level_begin, int level_count, const int* off_asm, int n_asm,
                                     const int* off_pan, int n_pan, const int* off_trl, int n_trl, hipStream_t st);
hipError_t launch_dense_solve_level(const DevGraph& d, int level_begin, int level_count, int level_max_b, hipStream_t st);

// ---- batched form: G independent graphs per launch (pps_multi_*) ------------------------------------------------
// Every kernel of an LM trial is launched once for a chunk of up to kBatchMax graphs: blockIdx.y = graph, blockIdx.x =
// the block index the single-graph kernel would have (blocks past a graph's own count exit).  The graphs' DevGraph
// records sit in a device array; what changes from trial to trial -- who takes part, lambda, which of the two state
// copies is the linearisation point -- travels in the kernel arguments.
constexpr int kBatchMax = 128;
enum { BF_ACTIVE = 1,      // the graph takes part in this round
       BF_RELIN = 2 };     // ... and is re-linearised first (its last trial was accepted)
struct BatchStage { int grp_begin, grp_count; };
// Dual-lambda form of a batch (the lm_solve_dual scheme per graph): what the second factorisation of a graph writes, and the
// three copies of its state.  x = state[xsel] is the linearisation point, state[(xsel + 1) % 3] / [(xsel + 2) % 3] receive
// x (+) delta for lambda / lambda2; an accepted trial only changes xsel.
struct BatchAlt {
  double *L, *U, *delta, *result_dev, *chi2_partials, *dn_partials;
  unsigned int* ticket;
  double* pose[3];
  double* plane[3];
};
struct BatchArgs {
  const DevGraph* gs;          // [n_total]
  const BatchStage* stage_tab; // [n_stages_max][n_total]
  double* results;             // pinned host, 8 doubles per graph: [0..3] chi2 at the linearisation point, [4..7] the trial
  int n_total, b0, n;          // graphs in the batch / first graph of this chunk / graphs in this chunk
  int no_products;             // the chunk's K1 writes no product records (thread-per-factor form of a large chunk: its K2 multiplies the Jacobians)
  double seq;
  double lambda[kBatchMax];
  unsigned char flags[kBatchMax];
  // dual form (alt != nullptr): grid z = 0 / 1 of the solve and trial kernels = lambda / lambda2, 12 result doubles per graph
  // ([0..3] chi2 at x, [4..7] trial for lambda, [8..11] trial for lambda2)
  const BatchAlt* alt;
  int rstride, lin_apply;      // lin_apply: 0 (the lane-form sweep's run-time flag, see k_linearize_lanes)
  double lambda2[kBatchMax];
  unsigned char xsel[kBatchMax];
};
// grid extents (maxima over the graphs of the chunk) and LDS needs of one round
struct BatchGeom {
  int lin_blocks = 0, lin_obs_blocks = 0, lin_rest_blocks = 0, repop_blocks = 0;
  int hblocks = 0, k2t_blocks = 0, k2tg_blocks = 0, hreduce = 0, retract = 0, chi2 = 0;      // k2tg: the generic entries of the class lists
  int k2_blocks = 0, k2_finish = 0;   // workgroups of the K2 launch: maximum over the chunk's graphs of ceil(single / 16) + multi; blocks of its second pass
  long long n_factors_total = 0;
  bool lin_thread_form = false;   // numeric K1 as one thread per factor (many graphs) instead of 32 lanes per factor
  int n_stages = 0;
  int stage_groups[32] = {0}, stage_nw_factor[32] = {0}, stage_nw_solve[32] = {0};
  int stage_per_wave_factor[32] = {0}, stage_per_wave_solve[32] = {0}, stage_grp_fronts[32] = {0};
  bool stage_reg_only[32] = {false};
  bool stage_pre[32] = {false};      // every group of every graph of the chunk has the shape of the pre-assembling walk at this stage's wave count
  int stage_nw_flow[32] = {0};       // waves of the data-flow back-substitution of the stage (0: the barrier form)
  // level-per-launch form (large chunks whose fronts all take the register path): one launch per tree level and size class,
  // every front on its own wave, no groups -- more waves per SIMD on the small classes than the band kernels can hold
  bool level_form = false;
  int n_levels = 0;
  int lvl_cls_blocks[64][3] = {{0}};    // workgroups (4 fronts each) per level and class: maximum over the chunk's graphs
  int lvl_blocks[64] = {0};             // ... per level, all classes (back-substitution)
  int lvl_direct_pp[64] = {0};          // largest p x p of a level whose fronts ALL take the direct-load back-substitution (0: some do not)
  int lvl_max_panel[64] = {0};          // largest factor panel ((p + b + 1) p doubles) of a level: the LDS a wave of its back-substitution needs
  int solve_per_wave_all = 0;           // LDS doubles per wave of the level solve (largest panel of the chunk)
  int stage_max_front[32] = {0};    // largest front (scalars, without the rhs row) of the stage over the chunk's graphs
};
hipError_t launch_batch_linearize(const BatchArgs& a, const BatchGeom& g, int mode, hipStream_t st);   // K1 of the BF_RELIN graphs (mode | 2: thread-per-factor form)
hipError_t launch_batch_hblocks(const BatchArgs& a, const BatchGeom& g, hipStream_t st, bool products);   // K2 of the BF_RELIN graphs (products: their K1 ran in the lane form)
hipError_t launch_batch_chi2(const BatchArgs& a, const BatchGeom& g, int slot, hipStream_t st);        // chi2 at lin -> results[8 b + 4 slot]
hipError_t launch_batch_solve(const BatchArgs& a, const BatchGeom& g, hipStream_t st, hipEvent_t after_factor = nullptr);   // K3, lambda per graph
struct RestoreRec { double* dst; const double* src; long long n; };
hipError_t launch_batch_restore(const RestoreRec* tab, int n, hipStream_t st);                 // dst[0 .. n) <- src[0 .. n) per record
hipError_t launch_batch_begin_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st);   // not-PD flags of both factorisations cleared
hipError_t launch_batch_trial_dual(const BatchArgs& a, const BatchGeom& g, hipStream_t st);   // state[..] <- x (+) delta_z, chi2 -> slots 1 / 2

// patch upload: `table` = (dst offset in the arena, src offset in src_base, bytes, unit: 0 = multiple of 16 copied as int4, 1 = multiple of 8 copied word by
// word) int64 quadruples; table and src_base may be pinned host memory
hipError_t launch_scatter_patches(const char* table, const char* src_base, int n_patches, char* arena, hipStream_t st);

// Largest front (scalars incl. rhs row) the LDS path of the factor kernel accepts.
int lds_front_limit();

// K1 micro-benchmark over replicated edge arrays (pps_bench_sweep)
// part: 0 = plane-edge launch, 1 = odometry launch, -1 = both
hipError_t launch_sweep_bench(const DevGraph& d, int mode, int replicas, double* Jbig, int part, hipStream_t st);

}  // namespace pps
