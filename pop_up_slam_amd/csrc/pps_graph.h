// pps_graph.h -- the graph handle behind the C-ABI (include/pps.h) and what the implementation files share.
//
//   pps_api.cpp      graph bookkeeping: handles, nodes / factors, values, state snapshots, introspection, K1 micro-benchmarks
//   pps_upload.cpp   compaction + symbolic analysis, device arenas, difference upload, state / measurement transfers
//   pps_solve.cpp    pps_update (Optimizer::relinearize), pps_batch_optimize (Optimizer::levenberg_marquardt), chi2
//   pps_multi.cpp    pps_multi: G graphs per launch, lockstep LM rounds
//   pps_frames.cpp   registered frames (measurement refresh on the device), data association, point re-projection
//   pps_io.cpp       graph text format (Slam::save / Graph::write)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pps.h"
#include "pps_device.h"
#include "pps_geom.h"
#include "pps_popup_dev.h"
#include "pps_symbolic.h"

namespace pps_impl {

struct HostNode {
  int type;
  double v[7];
  bool deleted;
  int compact;   // index among live nodes (SymNode id)
  int slot;      // index in the pose / plane device array
};
struct HostFactor {
  int type;
  int a, b;
  double meas[6];
  double w[21];
  bool deleted;
  int slot;      // index in its type's device arrays
  int repop;     // plane observation that re-pops its measurement from `ray` (Pose3d_Plane3d_Factor2)
  double ray[6]; // K^-1 (u,v,1) of the two ground-edge end points
};

inline double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace pps_impl

struct pps_graph {
  pps_props props;
  std::string err;
  std::vector<pps_impl::HostNode> nodes;
  std::vector<pps_impl::HostFactor> factors;
  int n_live_nodes = 0, n_live_factors = 0, dim_nodes = 0, dim_measure = 0;
  int n_live_type[4] = {0, 0, 0, 0}, n_live_repop = 0;      // live factors per type / live re-popping plane observations
  bool topo_dirty = true;       // structure changed since the last upload
  bool analysis_stale = true;   // structure changed since the last analysis
  bool host_values_newer = true;   // host node values must be pushed before the next solve
  bool dev_values_newer = false;   // device estimate is newer than the host copy
  bool meas_dirty = false;
  bool analyzed = false;
  int n_analyses = 0;              // analyses so far; with `grown_only` it selects the frame-loop form of the analysis
  // compacted node / factor tables of the last analysis (run_analysis appends to them while the graph only grows)
  std::vector<pps::SymNode> sym_nodes; std::vector<pps::SymFactor> sym_factors;
  size_t cmp_nodes = 0, cmp_factors = 0; int64_t cmp_base[4] = {0, 0, 0, 0}; bool cmp_valid = false, cmp_has_repop = false;
  bool grown_only = true;          // nothing has been removed since the last analysis (nodes / factors were only appended)
  bool grown_only_upload = false;  // ... since the last upload (false until there has been one)
  pps::Analysis an;
  pps::AnalysisParams aprm;
  pps::Switches sw;                // the PPS_* environment switches as they were when the handle was created
  pps::AnalysisCache* acache = nullptr;   // what the last analysis left for the next one (frame loops)
  std::vector<int> pose_ids, plane_ids;   // slot -> node id
  std::vector<int> fslot_ids[4];          // per type: slot -> factor id
  std::vector<int> level_max_front;
  bool use_band = false;                  // wave-per-front band kernels (fronts <= 127 rows)
  bool use_dense = false;                 // dense-front kernels (pps_dense.hip) when the band kernels do not apply
  std::vector<int> level_max_b;           // widest boundary per level
  int max_el_per_front = 0;
  // dense-front work lists: per level a prefix sum over its fronts (count+1 entries at level_off[l] + l)
  std::vector<int> dw_asm, dw_pan, dw_trl;
  int *d_dw_asm = nullptr, *d_dw_pan = nullptr, *d_dw_trl = nullptr;
  std::vector<int> stage_max_piv, stage_nw_factor, stage_nw_solve, stage_max_grp_fronts, stage_max_panel;
  std::vector<char> stage_pre;      // every group of the stage fits the pre-assembling walk of k_band_factor_pre
  // device
  bool dev_ready = false;
  hipStream_t stream = nullptr;
  pps::DevGraph dev;
  std::vector<void*> allocs;        // fallback allocations (arena full), freed at the next full upload
  // Device memory comes from two growable arenas that are re-used across uploads (a SLAM front end changes
  // the topology every frame; hipMalloc/hipFree per array per frame would dominate): `up` holds the arrays
  // that are uploaded (mirrored in a host staging buffer and sent with ONE copy), `scr` the scratch arrays.
  struct Arena { char* base = nullptr; size_t cap = 0, off = 0, spill = 0; };
  Arena up, scr;
  char* stage = nullptr;            // pinned host mirror of `up` (one H2D copy per upload, at link rate)
  size_t stage_cap = 0;
  size_t stage_lo = 0, stage_hi = 0;   // dirty range of the mirror
  // Frame loops re-upload a topology that is the previous one plus a little: every array of the upload arena keeps its
  // place from one upload to the next (a slot with spare capacity per dev_upload call, in call order), the pinned mirror
  // knows what the device holds, and only the bytes that differ are sent -- gathered into one patch buffer, one copy, one
  // scatter kernel (dozens of small copies would cost more than they carry).
  struct UpSlot { size_t off, cap; };
  std::vector<UpSlot> up_slots;
  size_t up_cursor = 0, up_high = 0;
  bool up_unknown = true;              // the arena was (re)allocated: the mirror says nothing about the device
  // Which analysis the upload mirror holds: `an` (sequence number an_seq) was built upon the analysis an_base_seq; when that is the
  // one whose arrays were uploaded last (up_an_seq), the leading entries `an.kept` names are in the mirror already and are not
  // compared again.  PPS_DEBUG_VERIFY_UPLOAD=1 checks every such claim (hint_violation -> PPS_ESTATE).
  long an_seq = 0, an_base_seq = -1, up_an_seq = -1;
  bool hint_violation = false;
  bool verify_hints = false;           // PPS_DEBUG_VERIFY_UPLOAD, read once per upload
  bool up_unknown_meas = false;        // ... only about the measurement arrays (written behind the mirror's back)
  size_t slot_obs_meas = (size_t)-1;   // which upload slot holds obs_meas
  size_t slot_lp_meas = (size_t)-1;    // ... and lp_meas (pps_set_measurement writes both arrays behind the mirror's back)
  // packed factor arrays of the last upload: an upload that only appends fills in the new slots instead of packing every
  // factor again (pk_n = slots that are current; pk_meas_ok: the measurement rows still match the host factors)
  std::vector<int> pk_obs_a, pk_obs_b, pk_odo_a, pk_odo_b, pk_obs_ids, pk_odo_ids;
  std::vector<double> pk_obs_m, pk_obs_w, pk_odo_m, pk_odo_w;
  size_t pk_n_obs = 0, pk_n_odo = 0, pk_ld_obs = 0, pk_ld_odo = 0;
  bool pk_meas_ok = false;
  bool status_clean = false;     // result_dev / spec_result are zero: upload_all zeroed them, or the last solve's chi2 kernels consumed the flags
  bool up_inflight = false;      // upload_all left copies from the pinned buffers in flight on `stream`
  double* state_pin = nullptr; size_t state_pin_cap = 0;     // pinned staging of upload_state / download_state
  bool pin_holds_est = false;                                // state_pin holds the device estimate as it is now (enqueue_state_download)
  std::unordered_map<std::string, double> up_laps;         // PPS_UPLOAD_TIMING=1: seconds per phase of upload_all, summed; printed at destroy
  struct UpPatch { size_t off, len; bool exact8 = false; };   // exact8: 8-byte granularity, nothing around the piece may be written
  std::vector<UpPatch> up_patches;
  char* patch_host = nullptr; size_t patch_cap = 0;    // pinned: the table of k_scatter_patches (the pieces are read from the pinned mirror)
  size_t up_bytes_sent = 0, up_bytes_total = 0;        // of the last flush (stats)
  double* host_result = nullptr;   // pinned, 12 doubles: chi2 at the linearisation point | trial | speculative trial
  double seq = 0.0;                // sequence number the chi2 kernel publishes last (host polls it)
  // the second damping value of a dual solve (lambda * factor): its own L / U / delta, a third copy of the state, its own
  // reduction scratch and result record
  double *spec_pose = nullptr, *spec_plane = nullptr, *spec_chi2_partials = nullptr, *spec_dn_partials = nullptr;
  unsigned int* spec_ticket = nullptr;
  double seq2 = 0.0;
  double *spec_L = nullptr, *spec_U = nullptr, *spec_delta = nullptr;
  double* spec_result = nullptr;   // result_dev of the speculative set: its own not-PD flag
  // the spare set of J / P / H / Hf that the fused trial + linearisation launch writes (SpecLin, pps_device.h); when LM accepts the trial it was
  // made for, the set trades places with dev.J / P / H / Hf by pointer.  Null where the dual LM loop does not apply (no band schedule).
  double *spec_J = nullptr, *spec_P = nullptr, *spec_H = nullptr, *spec_Hf = nullptr;
  // the whole tree in one factor launch + one back-substitution launch (run_analysis decides; launch_band_all): maxima over the stages, and the
  // number of the last launch pair (its hand-over flags carry it: DevGraph::k3_flag)
  bool k3_all = false;
  int k3_nw_factor = 0, k3_nw_solve = 0, k3_max_front = 0, k3_max_panel = 0, k3_max_grp = 0, k3_epoch = 0;
  double *snap_pose = nullptr, *snap_plane = nullptr;   // pps_save_state
  int snap_version = -1, upload_version = 0;
  int k2t_version = -1;              // upload_version the class lists of K2's throughput form (dev.k2t) were built for
  int profiling = 0;               // 0 off, 1 = K1 event pairs without host syncs, 2 = every phase (adds syncs)
  hipEvent_t ev[2] = {nullptr, nullptr};
  unsigned long long launches0 = 0;   // launch_count() at the start of the solve call
  std::vector<char> k1_skip;          // per K1 event pair: not a linearisation that ran
  std::vector<hipEvent_t> k1_events;   // pairs (start, stop) recorded around the sweep
  std::vector<hipEvent_t> fk_events;   // profiling level 1: start / stop of every factor launch of the dual loop (dispatch timestamps)
  int fk_used = 0;
  int k1_used = 0;
  // registered frames (pps_frames_add): 2-D ground segments that re-derive edge measurements on the device
  float frames_invK[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::vector<int> fr_pose;          // frame -> pose node id
  std::vector<int> fr_seg_off{0};    // frame -> first segment
  std::vector<float> fr_seg;         // 4 floats per segment
  std::vector<int> fr_item_frame, fr_item_plane, fr_item_fid;
  bool frames_dirty = true;          // device tables must be rebuilt
  // the slot tables of the registered frames only grow while nothing is removed (n_removals): cached, and the first upload of the
  // tables after an upload_all lands on the slots of the previous layout's -- whose leading entries (fr_hint_*) are not compared again
  std::vector<int> fr_slot, fr_pslot;
  long n_removals = 0, fr_cache_removals = -1;
  int fr_hint_version = -2; size_t fr_hint_cursor = 0, fr_hint_items = 0, fr_hint_frames = 0;
  int frames_version = -1;
  int *d_item_frame = nullptr, *d_item_plane = nullptr, *d_item_slot = nullptr, *d_frame_pose_slot = nullptr, *d_frame_seg_off = nullptr;
  float* d_fr_seg = nullptr;
  bool dev_meas_newer = false;       // device edge measurements are newer than the host copies
  int n_obs_fixed = 0;               // plane observations with a stored measurement (slots below this)
  // landmark records for data association (pps_landmark_update / pps_find_closest_planes)
  struct Landmark { int plane_id, fpi, seq, deleted; float seg2d[4], seg3d[4]; };
  std::vector<Landmark> lms;
  std::unordered_map<int, int> lm_of_plane;
  bool lms_dirty = true;
  int lms_upload_version = -1;       // upload_version the slots of d_lms were resolved against
  pps::AssocLandmark* d_lms = nullptr; size_t d_lms_cap = 0;
  pps::AssocLandmark* h_lms = nullptr;                        // pinned staging of d_lms (same capacity): the copy is not waited for on its own
  pps::AssocQuery* d_queries = nullptr; pps::AssocResult* d_results = nullptr; size_t d_q_cap = 0;   // PINNED host memory: k_assoc reads the queries and writes the results over the bus
  double* d_lm_planes = nullptr; size_t d_lm_planes_cap = 0;   // [4][n] landmark planes when the solver state is not current
  char* rp_pin = nullptr; size_t rp_cap = 0;                  // pinned block of pps_reproject_points: [slots | points in | points out], read and written by the kernel
  // stats / trace
  pps_stats stats{};
  std::vector<double> tr_lambda, tr_chi2;
  std::vector<int> tr_acc;
};

namespace pps_impl {
using namespace pps;

inline int fail(pps_graph* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}
inline int hip_fail(pps_graph* g, hipError_t e, const char* what) {
  return fail(g, PPS_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(g, expr)                                   \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return hip_fail(g, _e, #expr);   \
  } while (0)


inline bool live_node(const pps_graph* g, int id, int type) {
  return id >= 0 && id < (int)g->nodes.size() && !g->nodes[id].deleted && g->nodes[id].type == type;
}

// ---- pps_upload.cpp ----
void free_device(pps_graph* g);
void release_arenas(pps_graph* g);
void up_diff(pps_graph* g, size_t o, const char* src, size_t n, bool force, size_t same_prefix = 0);
int flush_uploads(pps_graph* g);
int verify_uploads(pps_graph* g, const char* where);
int ensure_device(pps_graph* g);
int64_t j_capacity(int64_t n);
void j_bases(const pps_graph* g, int64_t base[4], int64_t* total);
void p_bases(const pps_graph* g, int64_t base[4], int64_t* total);      // product records (pps_symbolic.h: kPSize), same slab capacities
int run_analysis(pps_graph* g);
int state_pin_reserve(pps_graph* g, size_t doubles);
int linpoint_from_estimate(pps_graph* g);
int download_state(pps_graph* g);
int enqueue_state_download(pps_graph* g);      // ... before the synchronisation a solve ends with,
void state_download_arrived(pps_graph* g);     // ... and this after it
int upload_state(pps_graph* g, bool sync = true);
int download_measurements(pps_graph* g);
int upload_measurements(pps_graph* g);
int upload_all(pps_graph* g);
int prepare_solve(pps_graph* g);
// ---- pps_solve.cpp ----
int read_result(pps_graph* g, bool at_estimate, double* chi2, double* dnorm, bool* notpd);
void reset_solve_stats(pps_graph* g);
void abandon_device_copy(pps_graph* g);

template <class T>
int arena_alloc(pps_graph* g, pps_graph::Arena& a, T** out, size_t count) {
  *out = nullptr;
  if (count == 0) count = 1;
  const size_t bytes = count * sizeof(T);
  const size_t o = (a.off + 255) & ~size_t(255);
  if (a.base && o + bytes <= a.cap) { a.off = o + bytes; *out = reinterpret_cast<T*>(a.base + o); return PPS_OK; }
  a.spill += bytes + 256;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) return hip_fail(g, e, "hipMalloc");
  g->allocs.push_back(p);
  *out = static_cast<T*>(p);
  return PPS_OK;
}

template <class T>
int dev_alloc(pps_graph* g, T** out, size_t count) { return arena_alloc(g, g->scr, out, count); }


// exact_from (rows of 8-byte values only): entries [0, exact_from) of every row are newer on the device than anywhere on the
// host (measurements refreshed by k_refresh_measurements) -- exactly the entries [exact_from, used) are sent, byte for byte,
// and nothing around them (a piece rounded to the 64-byte compare stride or to the 16-byte copy unit would put the mirror's
// stale values over up to seven refreshed neighbours)
constexpr size_t kNoExact = (size_t)-1;
// same: leading ELEMENTS (of the array, or of every row) that the caller knows to be what this slot received last time
template <class T>
int dev_upload_impl(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from, size_t same);
template <class T>
int dev_upload(pps_graph* g, T** out, const std::vector<T>& v, size_t same = 0) { return dev_upload_impl(g, out, v, 0, 0, 0, false, kNoExact, same); }
template <class T>
int dev_upload_rows(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from = kNoExact,
                    size_t same = 0) {
  return dev_upload_impl(g, out, v, rows, ld, used, force, exact_from, same);
}

template <class T>
int dev_upload_impl(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from, size_t same) {
  *out = nullptr;
  pps_graph::Arena& a = g->up;
  const size_t bytes = std::max<size_t>(1, v.size()) * sizeof(T);
  const size_t k = g->up_cursor++;
  size_t o = 0, fresh_cap = 0;
  bool placed = false;
  if (a.base && g->stage) {
    if (k < g->up_slots.size() && bytes <= g->up_slots[k].cap) { o = g->up_slots[k].off; placed = true; }
    else {
      const size_t cap = (std::max<size_t>(256, bytes + bytes / 2) + 255) & ~size_t(255);
      o = (g->up_high + 255) & ~size_t(255);
      if (o + cap <= a.cap) {
        if (k < g->up_slots.size()) g->up_slots[k] = pps_graph::UpSlot{o, cap}; else g->up_slots.push_back(pps_graph::UpSlot{o, cap});
        g->up_high = o + cap;
        placed = true;
        fresh_cap = cap;
      }
    }
  }
  if (!placed) {                                     // arena exhausted (it is re-sized at the next upload): a plain allocation
    a.spill += bytes + bytes / 2 + 512;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(g, e, "hipMalloc");
    g->allocs.push_back(p);
    *out = static_cast<T*>(p);
    if (!v.empty()) HIP_TRY(g, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PPS_OK;
  }
  *out = reinterpret_cast<T*>(a.base + o);
  a.off = std::max(a.off, o + bytes);
  if (fresh_cap) {
    // A slot that has just been created (or moved behind the others because it outgrew its place) lies in a part of the arena
    // the mirror says nothing about: neither side has ever been written there, and a diff against it may find the new bytes
    // "already there" (zeros against a fresh pinned page, say) and leave the device with whatever it held.  Define the whole
    // slot -- capacity, not just what is used today: later uploads grow into it -- and send it once.
    memset(g->stage + o, 0, fresh_cap);
    if (!v.empty()) memcpy(g->stage + o, v.data(), v.size() * sizeof(T));
    g->up_bytes_total += fresh_cap;
    g->up_patches.push_back(pps_graph::UpPatch{o, fresh_cap, false});
    return PPS_OK;
  }
  if (v.empty()) return PPS_OK;
  const char* src = reinterpret_cast<const char*>(v.data());
  if (rows == 0 || g->up_unknown) { up_diff(g, o, src, v.size() * sizeof(T), force, rows == 0 ? std::min(same, v.size()) * sizeof(T) : 0); return PPS_OK; }
  const size_t keep = std::min(same, used);
  if (exact_from != kNoExact && !force && sizeof(T) == 8) {
    for (size_t r = 0; r < rows; r++) {
      const size_t ro = r * ld * sizeof(T);
      if (keep && g->verify_hints && memcmp(g->stage + o + ro, src + ro, keep * sizeof(T)) != 0) g->hint_violation = true;
      memcpy(g->stage + o + ro + keep * sizeof(T), src + ro + keep * sizeof(T), (used - keep) * sizeof(T));   // the mirror keeps the host's view of the row
      g->up_bytes_total += used * sizeof(T);
      if (used > exact_from) g->up_patches.push_back(pps_graph::UpPatch{o + ro + exact_from * sizeof(T), (used - exact_from) * sizeof(T), true});
    }
    return PPS_OK;
  }
  for (size_t r = 0; r < rows; r++) up_diff(g, o + r * ld * sizeof(T), src + r * ld * sizeof(T), used * sizeof(T), force, keep * sizeof(T));
  return PPS_OK;
}


}  // namespace pps_impl
