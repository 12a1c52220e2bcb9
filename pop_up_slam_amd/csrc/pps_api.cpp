// pps_api.cpp -- C-ABI implementation: graph container, upload, LM / GN drivers.
//
// Host control flow follows the reference line by line where it matters for parity:
//   pps_batch_optimize  == Optimizer::levenberg_marquardt  (Thirdparty/isam/isamlib/Optimizer.cpp:371-467)
//   pps_update          == Optimizer::relinearize          (Optimizer.cpp:114-185) via Slam::update, mod_batch = 1
// Everything numeric runs on the device; per LM trial one 32-byte result record (chi2, |delta|^2,
// not-PD flag) returns to the host for the accept / reject decision.
#include <hip/hip_runtime.h>

#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "../../include/pps.h"
#include "pps_device.h"
#include "pps_geom.h"
#include "pps_popup_dev.h"
#include "pps_symbolic.h"

using namespace pps;

namespace {

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

double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct pps_graph {
  pps_props props;
  std::string err;
  std::vector<HostNode> nodes;
  std::vector<HostFactor> factors;
  int n_live_nodes = 0, n_live_factors = 0, dim_nodes = 0, dim_measure = 0;
  bool topo_dirty = true;       // structure changed since the last upload
  bool analysis_stale = true;   // structure changed since the last analysis
  bool host_values_newer = true;   // host node values must be pushed before the next solve
  bool dev_values_newer = false;   // device estimate is newer than the host copy
  bool meas_dirty = false;
  bool analyzed = false;
  int n_analyses = 0;              // analyses so far; with `grown_only` it selects the frame-loop form of the analysis
  // compacted node / factor tables of the last analysis (run_analysis appends to them while the graph only grows)
  std::vector<SymNode> sym_nodes; std::vector<SymFactor> sym_factors;
  size_t cmp_nodes = 0, cmp_factors = 0; int64_t cmp_base[4] = {0, 0, 0, 0}; bool cmp_valid = false, cmp_has_repop = false;
  bool grown_only = true;          // nothing has been removed since the last analysis (nodes / factors were only appended)
  bool grown_only_upload = false;  // ... since the last upload (false until there has been one)
  Analysis an;
  AnalysisParams aprm;
  AnalysisCache* acache = nullptr;   // what the last analysis left for the next one (frame loops)
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
  // device
  bool dev_ready = false;
  hipStream_t stream = nullptr;
  DevGraph dev;
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
  bool lin_is_est = false;       // upload_state has just filled est AND lin: the estimate_to_linpoint copy of the next solve is a no-op
  bool up_inflight = false;      // upload_all left copies from the pinned buffers in flight on `stream`
  double* state_pin = nullptr; size_t state_pin_cap = 0;     // pinned staging of upload_state / download_state
  std::unordered_map<std::string, double> up_laps;         // PPS_UPLOAD_TIMING=1: seconds per phase of upload_all, summed; printed at destroy
  struct UpPatch { size_t off, len; bool exact8 = false; };   // exact8: 8-byte granularity, nothing around the piece may be written
  std::vector<UpPatch> up_patches;
  char* patch_host = nullptr; size_t patch_cap = 0;    // pinned: [table | data]
  char* patch_dev = nullptr; size_t patch_dev_cap = 0;
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
  double *snap_pose = nullptr, *snap_plane = nullptr;   // pps_save_state
  int snap_version = -1, upload_version = 0;
  int profiling = 0;               // 0 off, 1 = K1 event pairs without host syncs, 2 = every phase (adds syncs)
  hipEvent_t ev[2] = {nullptr, nullptr};
  unsigned long long launches0 = 0;   // launch_count() at the start of the solve call
  std::vector<char> k1_skip;          // per K1 event pair: not a linearisation that ran
  std::vector<hipEvent_t> k1_events;   // pairs (start, stop) recorded around the sweep
  int k1_used = 0;
  // registered frames (pps_frames_add): 2-D ground segments that re-derive edge measurements on the device
  float frames_invK[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::vector<int> fr_pose;          // frame -> pose node id
  std::vector<int> fr_seg_off{0};    // frame -> first segment
  std::vector<float> fr_seg;         // 4 floats per segment
  std::vector<int> fr_item_frame, fr_item_plane, fr_item_fid;
  bool frames_dirty = true;          // device tables must be rebuilt
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
  pps::AssocQuery* d_queries = nullptr; pps::AssocResult* d_results = nullptr; size_t d_q_cap = 0;
  double* d_lm_planes = nullptr; size_t d_lm_planes_cap = 0;   // [4][n] landmark planes when the solver state is not current
  // stats / trace
  pps_stats stats{};
  std::vector<double> tr_lambda, tr_chi2;
  std::vector<int> tr_acc;
};

namespace {

int fail(pps_graph* g, int code, const std::string& msg) {
  if (g) g->err = msg;
  return code;
}
int hip_fail(pps_graph* g, hipError_t e, const char* what) {
  return fail(g, PPS_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(g, expr)                                   \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return hip_fail(g, _e, #expr);   \
  } while (0)

void free_device(pps_graph* g) {
  for (void* p : g->allocs) (void)hipFree(p);
  g->allocs.clear();
  // arenas are kept; grow them when the last layout spilled into fallback allocations
  for (pps_graph::Arena* a : {&g->up, &g->scr}) {
    const size_t want = (a == &g->up ? std::max(a->off, g->up_high) : a->off) + a->spill;
    if (a->spill > 0 || a->base == nullptr) {
      if (a->base) (void)hipFree(a->base);
      a->cap = std::max<size_t>(size_t(1) << 20, 2 * want);
      if (hipMalloc(reinterpret_cast<void**>(&a->base), a->cap) != hipSuccess) { a->base = nullptr; a->cap = 0; }
      if (a == &g->up) { g->up_slots.clear(); g->up_high = 0; g->up_unknown = true; }
    }
    a->off = 0; a->spill = 0;
  }
  g->up_cursor = 0; g->up_patches.clear();
  if (g->stage_cap < g->up.cap) {
    if (g->stage) (void)hipHostFree(g->stage);
    g->stage = nullptr; g->stage_cap = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&g->stage), g->up.cap, hipHostMallocDefault) == hipSuccess) g->stage_cap = g->up.cap;
    g->up_unknown = true;
  }
  g->stage_lo = g->stage_hi = 0;
  g->dev = DevGraph();
}

void release_arenas(pps_graph* g) {
  for (pps_graph::Arena* a : {&g->up, &g->scr}) { if (a->base) (void)hipFree(a->base); a->base = nullptr; a->cap = a->off = a->spill = 0; }
  if (g->stage) (void)hipHostFree(g->stage);
  g->stage = nullptr; g->stage_cap = 0;
  if (g->patch_host) (void)hipHostFree(g->patch_host);
  if (g->state_pin) (void)hipHostFree(g->state_pin);
  g->state_pin = nullptr; g->state_pin_cap = 0;
  if (g->patch_dev) (void)hipFree(g->patch_dev);
  g->patch_host = g->patch_dev = nullptr; g->patch_cap = g->patch_dev_cap = 0;
  g->up_slots.clear(); g->up_high = 0; g->up_unknown = true;
}

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

// bytes [0, n) of `src` against the mirror at offset o: record (and copy into the mirror) the range that differs
void up_diff(pps_graph* g, size_t o, const char* src, size_t n, bool force) {
  if (n == 0) return;
  char* mir = g->stage + o;
  g->up_bytes_total += n;
  if (g->up_unknown || force) { memcpy(mir, src, n); g->up_patches.push_back(pps_graph::UpPatch{o, n, false}); return; }
  // first and last 64-byte chunk that differs from what the device holds (4 KB strides first: most arrays of a frame loop
  // are unchanged from end to end, or up to a short tail)
  size_t lo = 0, hi = n;
  while (lo + 4096 <= hi && memcmp(mir + lo, src + lo, 4096) == 0) lo += 4096;
  while (lo + 64 <= hi && memcmp(mir + lo, src + lo, 64) == 0) lo += 64;
  if (lo + 64 > hi && memcmp(mir + lo, src + lo, hi - lo) == 0) return;      // identical
  while (hi >= lo + 4096 && memcmp(mir + hi - 4096, src + hi - 4096, 4096) == 0) hi -= 4096;
  while (hi >= lo + 64 && memcmp(mir + hi - 64, src + hi - 64, 64) == 0) hi -= 64;
  lo &= ~size_t(15);
  memcpy(mir + lo, src + lo, hi - lo);
  g->up_patches.push_back(pps_graph::UpPatch{o + lo, hi - lo, false});
}

// The k-th upload of a layout goes where the k-th upload of the previous layout went, as long as it fits the slot.
// rows > 0: an SoA array of `rows` rows with leading dimension ld of which the first `used` entries per row are live -- the
// rows are compared one by one (appending a factor touches the end of every row, not the array from end to end).
// exact_from (rows of 8-byte values only): entries [0, exact_from) of every row are newer on the device than anywhere on the
// host (measurements refreshed by k_refresh_measurements) -- exactly the entries [exact_from, used) are sent, byte for byte,
// and nothing around them (a piece rounded to the 64-byte compare stride or to the 16-byte copy unit would put the mirror's
// stale values over up to seven refreshed neighbours)
constexpr size_t kNoExact = (size_t)-1;
template <class T>
int dev_upload_impl(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from);
template <class T>
int dev_upload(pps_graph* g, T** out, const std::vector<T>& v) { return dev_upload_impl(g, out, v, 0, 0, 0, false, kNoExact); }
template <class T>
int dev_upload_rows(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from = kNoExact) {
  return dev_upload_impl(g, out, v, rows, ld, used, force, exact_from);
}

template <class T>
int dev_upload_impl(pps_graph* g, T** out, const std::vector<T>& v, size_t rows, size_t ld, size_t used, bool force, size_t exact_from) {
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
  if (rows == 0 || g->up_unknown) { up_diff(g, o, src, v.size() * sizeof(T), force); return PPS_OK; }
  if (exact_from != kNoExact && !force && sizeof(T) == 8) {
    for (size_t r = 0; r < rows; r++) {
      const size_t ro = r * ld * sizeof(T);
      memcpy(g->stage + o + ro, src + ro, used * sizeof(T));          // the mirror keeps the host's view of the row
      g->up_bytes_total += used * sizeof(T);
      if (used > exact_from) g->up_patches.push_back(pps_graph::UpPatch{o + ro + exact_from * sizeof(T), (used - exact_from) * sizeof(T), true});
    }
    return PPS_OK;
  }
  for (size_t r = 0; r < rows; r++) up_diff(g, o + r * ld * sizeof(T), src + r * ld * sizeof(T), used * sizeof(T), force);
  return PPS_OK;
}

// send what differs: everything in one copy when the device content is unknown or most of it changed, else the patches
int flush_uploads(pps_graph* g) {
  // callers: upload_all (after its opening stream sync) and the frame tables of pps_refresh_measurements (which settles
  // up_inflight first) -- the pinned mirror and the patch buffer are never rewritten under a copy that still reads them
  size_t sent = 0;
  for (const auto& pt : g->up_patches) sent += pt.len;
  g->up_bytes_sent = sent;
  if (g->up_patches.empty()) { g->up_bytes_total = 0; return PPS_OK; }
  // One copy of the whole arena only when the device content is unknown.  Otherwise nothing but the changed pieces may be
  // written: the span between two pieces can hold what kernels have refreshed behind the mirror's back (the observation
  // measurements that stay on the device) -- a copy "from the first to the last change" would put stale values over them.
  if (g->up_unknown) {
    size_t lo = 0, hi = 0;
    for (const auto& pt : g->up_patches) hi = std::max(hi, pt.off + pt.len);
    hi = std::max(hi, g->up_high); hi = std::min(hi, g->stage_cap);
    HIP_TRY(g, hipMemcpyAsync(g->up.base + lo, g->stage + lo, hi - lo, hipMemcpyHostToDevice, g->stream));
    g->up_bytes_sent = hi - lo;
  } else if (g->up_patches.size() <= 3) {
    for (const auto& pt : g->up_patches)        // a few pieces: straight from the pinned mirror
      HIP_TRY(g, hipMemcpyAsync(g->up.base + pt.off, g->stage + pt.off, pt.len, hipMemcpyHostToDevice, g->stream));
  } else {
    // [table: 4 x int64 per patch | data, 16-byte aligned pieces] -> one copy -> scatter kernel
    const size_t np = g->up_patches.size();
    size_t need = np * 32;
    std::vector<size_t> src_off(np);
    auto plen = [&](size_t i) { const auto& pt = g->up_patches[i]; return pt.exact8 ? pt.len : ((pt.len + 15) & ~size_t(15)); };   // (exact pieces are multiples of 8)
    for (size_t i = 0; i < np; i++) { need = (need + 15) & ~size_t(15); src_off[i] = need; need += plen(i); }
    if (need > g->patch_cap) {
      if (g->patch_host) (void)hipHostFree(g->patch_host);
      g->patch_host = nullptr; g->patch_cap = 0;
      const size_t cap = std::max<size_t>(1 << 16, 2 * need);
      HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->patch_host), cap, hipHostMallocDefault));
      g->patch_cap = cap;
    }
    if (need > g->patch_dev_cap) {
      if (g->patch_dev) (void)hipFree(g->patch_dev);
      g->patch_dev = nullptr; g->patch_dev_cap = 0;
      const size_t cap = std::max<size_t>(1 << 16, 2 * need);
      HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->patch_dev), cap));
      g->patch_dev_cap = cap;
    }
    long long* tab = reinterpret_cast<long long*>(g->patch_host);
    for (size_t i = 0; i < np; i++) {
      const auto& pt = g->up_patches[i];
      tab[4 * i + 0] = (long long)pt.off; tab[4 * i + 1] = (long long)src_off[i]; tab[4 * i + 2] = (long long)plen(i); tab[4 * i + 3] = pt.exact8 ? 1 : 0;
      memcpy(g->patch_host + src_off[i], g->stage + pt.off, plen(i));    // the mirror already holds the new bytes
    }
    HIP_TRY(g, hipMemcpyAsync(g->patch_dev, g->patch_host, need, hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, launch_scatter_patches(g->patch_dev, (int)np, g->up.base, g->stream));
    // (the mirror and the patch buffer are written again by the next upload_all, which begins and ends with a stream sync)
  }
  g->up_unknown = false;
  g->up_patches.clear();
  g->stage_lo = g->stage_hi = 0;
  return PPS_OK;
}

// PPS_DEBUG_VERIFY_UPLOAD=1: after a flush, the arena on the device must equal the pinned mirror (except the observation
// measurements, which kernels refresh behind the mirror's back) -- PPS_ESTATE if not.  tests/test_gpu_pipeline.py runs a frame
// loop under it.
int verify_uploads(pps_graph* g, const char* where) {
  if (!getenv("PPS_DEBUG_VERIFY_UPLOAD") || g->up_high == 0 || g->up.spill) return PPS_OK;
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  std::vector<char> dev(g->up_high);
  HIP_TRY(g, hipMemcpy(dev.data(), g->up.base, g->up_high, hipMemcpyDeviceToHost));
  for (size_t k = 0; k < g->up_slots.size() && k < g->up_cursor; k++) {
    if (k == g->slot_obs_meas || k == g->slot_lp_meas) continue;
    const size_t o = g->up_slots[k].off, n = std::min(g->up_slots[k].cap, g->up_high - std::min(g->up_high, o));
    if (o >= g->up_high) continue;
    if (memcmp(dev.data() + o, g->stage + o, n) != 0) {
      size_t b = 0; while (b < n && dev[o + b] == g->stage[o + b]) b++;
      return fail(g, PPS_ESTATE, std::string("upload verification (") + where + "): slot " + std::to_string(k) + " (offset " + std::to_string(o) + ", capacity " +
                                     std::to_string(g->up_slots[k].cap) + ") differs from the mirror at byte " + std::to_string(b));
    }
  }
  return PPS_OK;
}

int ensure_device(pps_graph* g) {
  if (g->dev_ready) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  HIP_TRY(g, hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->host_result), 12 * sizeof(double), hipHostMallocDefault));
  HIP_TRY(g, hipEventCreate(&g->ev[0]));
  HIP_TRY(g, hipEventCreate(&g->ev[1]));
  g->dev_ready = true;
  return PPS_OK;
}

// Offsets of the four per-type slabs of the J buffer: [plane obs | odometry | pose priors | plane priors].  Every slab is
// sized for a capacity that grows in powers of two, so that a graph that gains a few factors per frame keeps all of its J
// offsets -- and with them the whole contribution list of the H-block kernel -- from one frame to the next.
int64_t j_capacity(int64_t n) { int64_t c = 16; while (c < n) c <<= 1; return c; }
void j_bases(const pps_graph* g, int64_t base[4], int64_t* total) {
  const int64_t n_pp = g->fslot_ids[F_POSE_PRIOR].size(), n_odo = g->fslot_ids[F_ODOMETRY].size(),
                n_obs = g->fslot_ids[F_PLANE_OBS].size(), n_lp = g->fslot_ids[F_PLANE_PRIOR].size();
  const int64_t joff_obs = 0, joff_odo = j_capacity(n_obs) * 30, joff_pp = joff_odo + j_capacity(n_odo) * 78,
                joff_lp = joff_pp + j_capacity(n_pp) * 42;
  base[F_POSE_PRIOR] = joff_pp; base[F_ODOMETRY] = joff_odo; base[F_PLANE_OBS] = joff_obs; base[F_PLANE_PRIOR] = joff_lp;
  if (total) *total = joff_lp + j_capacity(n_lp) * 12;
}

// ---- compaction + symbolic analysis (host only) -------------------------------------------
int run_analysis(pps_graph* g) {
  const double t0 = now_s();
  std::vector<SymNode>& sn = g->sym_nodes;
  std::vector<SymFactor>& sf = g->sym_factors;
  // A graph that only grew since the last analysis (the frame loop) appends to the compacted tables instead of walking
  // every node and factor again; re-popping edges are ordered behind the fixed ones, which moves slots: they take the full path.
  bool append = g->cmp_valid && g->grown_only && g->n_analyses > 0 && !g->cmp_has_repop && g->cmp_nodes <= g->nodes.size() &&
                g->cmp_factors <= g->factors.size() && !getenv("PPS_NO_INCR_COMPACT");
  for (size_t i = g->cmp_factors; append && i < g->factors.size(); i++)
    append = !g->factors[i].deleted && !(g->factors[i].type == F_PLANE_OBS && g->factors[i].repop);
  for (size_t i = g->cmp_nodes; append && i < g->nodes.size(); i++) append = !g->nodes[i].deleted;
  if (append) {
    for (size_t i = g->cmp_nodes; i < g->nodes.size(); i++) {
      HostNode& n = g->nodes[i];
      n.compact = (int)sn.size();
      if (n.type == NODE_POSE) { n.slot = (int)g->pose_ids.size(); g->pose_ids.push_back((int)i); sn.push_back({NODE_POSE, 6, n.slot}); }
      else { n.slot = (int)g->plane_ids.size(); g->plane_ids.push_back((int)i); sn.push_back({NODE_PLANE, 3, -1}); }
    }
    for (size_t i = g->cmp_factors; i < g->factors.size(); i++) {
      HostFactor& f = g->factors[i];
      f.slot = (int)g->fslot_ids[f.type].size();
      g->fslot_ids[f.type].push_back((int)i);
    }
    g->n_obs_fixed = (int)g->fslot_ids[F_PLANE_OBS].size();
    int64_t base[4], j_total = 0;
    j_bases(g, base, &j_total);
    if (j_total > 0x7fffffffLL) return fail(g, PPS_ENOMEM, "graph too large for int32 J offsets");
    if (memcmp(base, g->cmp_base, sizeof(base)) != 0) {          // a J slab outgrew its capacity: every offset moves
      int cnt[4] = {0, 0, 0, 0};
      for (SymFactor& q : sf) q.joff = (int)(base[q.type] + (int64_t)(cnt[q.type]++) * kJSize[q.type]);
      memcpy(g->cmp_base, base, sizeof(base));
    }
    for (size_t i = g->cmp_factors; i < g->factors.size(); i++) {
      const HostFactor& f = g->factors[i];
      SymFactor q;
      q.type = f.type;
      q.a = g->nodes[f.a].compact;
      q.b = f.b >= 0 ? g->nodes[f.b].compact : -1;
      q.joff = (int)(base[f.type] + (int64_t)f.slot * kJSize[f.type]);
      q.direct_ok = f.type == F_PLANE_OBS ? 1 : 0;
      sf.push_back(q);
    }
  } else {
  g->pose_ids.clear(); g->plane_ids.clear();
  for (int t = 0; t < 4; t++) g->fslot_ids[t].clear();
  sn.clear(); sf.clear();
  for (size_t i = 0; i < g->nodes.size(); i++) {
    HostNode& n = g->nodes[i];
    if (n.deleted) { n.compact = n.slot = -1; continue; }
    n.compact = (int)sn.size();
    if (n.type == NODE_POSE) { n.slot = (int)g->pose_ids.size(); g->pose_ids.push_back((int)i); sn.push_back({NODE_POSE, 6, n.slot}); }
    else { n.slot = (int)g->plane_ids.size(); g->plane_ids.push_back((int)i); sn.push_back({NODE_PLANE, 3, -1}); }
  }
  // plane observations with a fixed measurement first, the re-popping ones (Factor2) behind them
  g->cmp_has_repop = false;
  for (int pass = 0; pass < 2; pass++)
    for (size_t i = 0; i < g->factors.size(); i++) {
      HostFactor& f = g->factors[i];
      if (f.deleted) { f.slot = -1; continue; }
      if ((f.type == F_PLANE_OBS && f.repop) != (pass == 1)) continue;
      if (pass == 1) g->cmp_has_repop = true;
      f.slot = (int)g->fslot_ids[f.type].size();
      g->fslot_ids[f.type].push_back((int)i);
    }
  g->n_obs_fixed = 0;
  for (int id : g->fslot_ids[F_PLANE_OBS]) g->n_obs_fixed += g->factors[id].repop ? 0 : 1;
  int64_t base[4], j_total = 0;
  j_bases(g, base, &j_total);
  if (j_total > 0x7fffffffLL) return fail(g, PPS_ENOMEM, "graph too large for int32 J offsets");
  memcpy(g->cmp_base, base, sizeof(base));
  sf.reserve(g->factors.size());
  bool any_deleted = false;
  for (size_t i = 0; i < g->factors.size(); i++) {
    const HostFactor& f = g->factors[i];
    if (f.deleted) { any_deleted = true; continue; }
    SymFactor s;
    s.type = f.type;
    s.a = g->nodes[f.a].compact;
    s.b = f.b >= 0 ? g->nodes[f.b].compact : -1;
    s.joff = (int)(base[f.type] + (int64_t)f.slot * kJSize[f.type]);
    s.direct_ok = (f.type == F_PLANE_OBS && !f.repop) ? 1 : 0;
    sf.push_back(s);
  }
  // the append path relies on: table index == host index order with nothing skipped
  g->cmp_valid = !any_deleted && sn.size() == g->nodes.size();
  }
  g->cmp_nodes = g->nodes.size(); g->cmp_factors = g->factors.size();
  // band depth: 4 levels per launch when the solve is latency bound (C2: 512 fronts; 113.3 vs 115.0 us per LM iteration
  // with 3), 2 when the lower levels are throughput bound (C3: 5 360 fronts; 637 vs 710 us)
  g->aprm.band_levels = g->pose_ids.size() >= 4000 ? 2 : 4;
  // H-block segments (contributions reduced by one wave of K2): short on small graphs, where the few long segments
  // (ground plane, 32 contributions = 16 dependent load rounds) are K2's critical path; long on large ones, where the
  // number of waves is (C2: 23.3 -> 18.7 us with 8; C3: 82 -> 102 us)
  g->aprm.seg_len = g->pose_ids.size() >= 4000 ? 32 : 8;
  // a graph that is re-analysed after pure appends is a frame loop: absolute cut positions keep the left part of its tree
  g->aprm.aligned_cuts = (g->n_analyses > 0 && g->grown_only) ? 1 : 0;
  // ... and its aligned cuts leave a few fronts of 65 .. 80 rows, whose 25 KB triangles let 5 waves share a CU's LDS, not 8:
  // groups of 4 leaves (3 levels per launch) keep every front of a level on its own wave (C5: 1 580 vs 1 500 frames/s)
  if (g->aprm.aligned_cuts && g->pose_ids.size() < 4000) g->aprm.band_levels = 3;
  g->aprm.band_rows = band_front_limit();
  const char* msg = "";
  try {
  if (getenv("PPS_ANALYSIS_TIMING")) fprintf(stderr, "[analysis] %-22s %8.3f ms\n", "compaction (api)", 1e3 * (now_s() - t0));
  if (!g->acache) g->acache = analysis_cache_new();
  if (!analyze(sn, sf, g->aprm, g->an, &msg, getenv("PPS_NO_INCREMENTAL") ? nullptr : g->acache))
    return fail(g, PPS_EINVAL, std::string("analysis failed: ") + msg);
  // fronts beyond the wave-per-front kernels (loop-closure separators) run in the dense-front form, whose cost is
  // per tree level: split their supernodes into 64-pivot chunks instead of 48 (a quarter fewer levels)
  if (g->an.max_front > band_front_limit() && g->aprm.max_pivots < dense_front_max_pivots()) {
    AnalysisParams wide = g->aprm;
    wide.max_pivots = dense_front_max_pivots();
    if (!analyze(sn, sf, wide, g->an, &msg, getenv("PPS_NO_INCREMENTAL") ? nullptr : g->acache))
      return fail(g, PPS_EINVAL, std::string("analysis failed: ") + msg);
  }
  } catch (const std::bad_alloc&) {
    g->an = Analysis();
    return fail(g, PPS_ENOMEM, "symbolic analysis ran out of host memory (fronts too wide for this ordering)");
  }
  g->level_max_front.assign(g->an.n_levels, 0);
  for (int s = 0; s < g->an.n_fronts; s++) {
    int& m = g->level_max_front[g->an.f_level[s]];
    m = std::max(m, g->an.f_p[s] + g->an.f_b[s]);
  }
  {
    const Analysis& A = g->an;
    const int Bn = std::max(1, g->aprm.band_levels);
    g->stage_max_piv.assign(A.n_stages, 1);
    for (int s = 0; s < A.n_fronts; s++) { int& m = g->stage_max_piv[A.f_level[s] / Bn]; m = std::max(m, A.f_p[s]); }
    int max_piv = 0;
    for (int m : g->stage_max_piv) max_piv = std::max(max_piv, m);
    g->use_band = A.max_front <= band_front_limit() && max_piv <= 64;
    // the dense-front solve keeps a front's boundary values in LDS: 15 000 scalars is the ceiling (2-D loop-closure
    // meshes such as torus10000 reach 24 540 under this chain-based dissection and are refused, see below)
    g->use_dense = !g->use_band && max_piv <= dense_front_max_pivots() && A.max_front <= 15000;
    g->level_max_b.assign(A.n_levels, 0);
    g->max_el_per_front = 0;
    for (int s = 0; s < A.n_fronts; s++) {
      g->level_max_b[A.f_level[s]] = std::max(g->level_max_b[A.f_level[s]], A.f_b[s]);
      g->max_el_per_front = std::max(g->max_el_per_front, A.f_el_off[s + 1] - A.f_el_off[s]);
    }
    g->dw_asm.clear(); g->dw_pan.clear(); g->dw_trl.clear();
    if (g->use_dense)
      for (int l = 0; l < A.n_levels; l++) {
        int a = 0, pn = 0, t = 0;
        g->dw_asm.push_back(0); g->dw_pan.push_back(0); g->dw_trl.push_back(0);
        for (int k = A.level_off[l]; k < A.level_off[l + 1]; k++) {
          const int s = A.level_fronts[k];
          const int fa = A.f_p[s] + A.f_b[s] + 1, b1 = A.f_b[s] + 1;
          const int T32 = (fa + 31) / 32, T64 = (b1 + 63) / 64;
          a += T32 * ((A.f_p[s] + 31) / 32); pn += (fa - A.f_p[s] + 255) / 256; t += T64 * (T64 + 1) / 2;
          g->dw_asm.push_back(a); g->dw_pan.push_back(pn); g->dw_trl.push_back(t);
        }
      }
    g->stage_nw_factor.assign(A.n_stages, 1); g->stage_nw_solve.assign(A.n_stages, 1);
    g->stage_max_grp_fronts.assign(A.n_stages, 1); g->stage_max_panel.assign(A.n_stages, 1);
    for (int s = 0; s < A.n_fronts; s++) { int& m = g->stage_max_panel[A.f_level[s] / Bn]; m = std::max(m, (A.f_p[s] + A.f_b[s] + 1) * A.f_p[s]); }
    const size_t lds_budget = 150 * 1024;
    const int max_waves = 8;
    for (int st = 0; st < A.n_stages; st++) {
      const int want = std::max(1, std::min(max_waves, A.stage_max_width[st]));
      g->stage_nw_factor[st] = (int)std::max<size_t>(1, std::min<size_t>(want, lds_budget / band_lds_bytes(A.stage_max_front[st], A.stage_max_front[st] + 1 <= band_reg_rows() && !getenv("PPS_TRACE"))));
      int mg = 1;
      for (int gi = A.stage_grp_off[st]; gi < A.stage_grp_off[st + 1]; gi++)
        mg = std::max(mg, A.glvl_front_off[A.grp_lvl_off[gi + 1]] - A.glvl_front_off[A.grp_lvl_off[gi]]);
      g->stage_max_grp_fronts[st] = mg;
      // per workgroup: one local solution vector per front of a group + per wave xb and the factor panel.  A group that does
      // not fit (very wide elimination trees: hundreds of fronts in one band group) takes the graph off the band kernels.
      const size_t xbytes = (size_t)mg * band_max_rows() * sizeof(double);
      const size_t per_wave = band_solve_lds_bytes(g->stage_max_panel[st]);
      if (xbytes + per_wave > lds_budget) { g->use_band = false; g->stage_nw_solve[st] = 1; continue; }
      g->stage_nw_solve[st] = (int)std::max<size_t>(1, std::min<size_t>(want, (lds_budget - xbytes) / per_wave));
    }
    // (such a graph then runs on the level-per-launch kernels: its fronts are <= 127 rows by the use_band test above)
  }
  if (!g->use_band && !g->use_dense && g->an.max_front > 4096)
    return fail(g, PPS_ENOMEM, "fronts too wide for this ordering (max front " + std::to_string(g->an.max_front) +
                               " scalars): the pose chain is not a good dissection backbone for this graph");
  if (getenv("PPS_ANALYSIS_TIMING")) fprintf(stderr, "[analysis] %-22s %8.3f ms\n", "total incl. api", 1e3 * (now_s() - t0));
  g->analyzed = true; g->analysis_stale = false;
  g->n_analyses++; g->grown_only = true;
  g->stats.n_fronts = g->an.n_fronts; g->stats.n_levels = g->an.n_levels; g->stats.max_front = g->an.max_front;
  g->stats.nnz_L = g->an.L_size;
  g->stats.t_analysis = now_s() - t0;
  return PPS_OK;
}

// pinned staging for the estimate (pose rows, then plane rows): copies to and from pageable memory are staged by the runtime
// and cost a synchronisation each
int state_pin_reserve(pps_graph* g, size_t doubles) {
  if (doubles <= g->state_pin_cap) return PPS_OK;
  if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }
  if (g->state_pin) (void)hipHostFree(g->state_pin);
  g->state_pin = nullptr; g->state_pin_cap = 0;
  const size_t cap = std::max<size_t>(4096, 2 * doubles);
  HIP_TRY(g, hipHostMalloc(reinterpret_cast<void**>(&g->state_pin), cap * sizeof(double), hipHostMallocDefault));
  g->state_pin_cap = cap;
  return PPS_OK;
}

// pull the device estimate back into the host node table
int download_state(pps_graph* g) {
  if (!g->dev_values_newer) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  const DevGraph& d = g->dev;
  const size_t np = (size_t)7 * d.pose_ld, nl = (size_t)4 * d.plane_ld;
  int rc = state_pin_reserve(g, np + nl);
  if (rc != PPS_OK) return rc;
  double* bp = g->state_pin; double* bl = g->state_pin + np;
  HIP_TRY(g, hipMemcpyAsync(bp, d.pose_est, (np + nl) * 8, hipMemcpyDeviceToHost, g->stream));   // [poses | planes], one block
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  for (int s = 0; s < d.n_pose; s++) for (int k = 0; k < 7; k++) g->nodes[g->pose_ids[s]].v[k] = bp[(size_t)k * d.pose_ld + s];
  for (int s = 0; s < d.n_plane; s++) for (int k = 0; k < 4; k++) g->nodes[g->plane_ids[s]].v[k] = bl[(size_t)k * d.plane_ld + s];
  g->dev_values_newer = false;
  return PPS_OK;
}

int upload_state(pps_graph* g, bool sync = true) {
  DevGraph& d = g->dev;
  if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }   // the staging buffer is still being read
  const size_t np = (size_t)7 * d.pose_ld, nl = (size_t)4 * d.plane_ld;
  int rc = state_pin_reserve(g, np + nl);
  if (rc != PPS_OK) return rc;
  double* bp = g->state_pin; double* bl = g->state_pin + np;
  for (int k = 0; k < 7; k++) for (int s = d.n_pose; s < d.pose_ld; s++) bp[(size_t)k * d.pose_ld + s] = 0.0;
  for (int k = 0; k < 4; k++) for (int s = d.n_plane; s < d.plane_ld; s++) bl[(size_t)k * d.plane_ld + s] = 0.0;
  for (int s = 0; s < d.n_pose; s++) for (int k = 0; k < 7; k++) bp[(size_t)k * d.pose_ld + s] = g->nodes[g->pose_ids[s]].v[k];
  for (int s = 0; s < d.n_plane; s++) for (int k = 0; k < 4; k++) bl[(size_t)k * d.plane_ld + s] = g->nodes[g->plane_ids[s]].v[k];
  HIP_TRY(g, hipMemcpyAsync(d.pose_est, bp, (np + nl) * 8, hipMemcpyHostToDevice, g->stream));
  HIP_TRY(g, hipMemcpyAsync(d.pose_lin, d.pose_est, (np + nl) * 8, hipMemcpyDeviceToDevice, g->stream));
  g->lin_is_est = true;
  if (sync) HIP_TRY(g, hipStreamSynchronize(g->stream));
  else g->up_inflight = true;
  g->host_values_newer = false;
  return PPS_OK;
}

// SoA with leading dimension ld (>= count): value k of slot s at [k * ld + s]
template <int K>
void pack_soa(const pps_graph* g, int type, const double HostFactor::*dummy, bool weights, std::vector<double>& out, size_t ld) {
  (void)dummy;
  const std::vector<int>& ids = g->fslot_ids[type];
  const size_t n = ids.size();
  out.assign((size_t)K * ld, 0.0);
  for (size_t s = 0; s < n; s++) {
    const HostFactor& f = g->factors[ids[s]];
    const double* src = weights ? f.w : f.meas;
    for (int k = 0; k < K; k++) out[(size_t)k * ld + s] = src[k];
  }
}

// pull device-refreshed plane-observation measurements back into the host factor table
int download_measurements(pps_graph* g) {
  if (!g->dev_meas_newer) return PPS_OK;
  HIP_TRY(g, hipSetDevice(g->props.device));
  const DevGraph& d = g->dev;
  const size_t n = g->fslot_ids[F_PLANE_OBS].size(), ld = (size_t)d.obs_ld;
  std::vector<double> m((size_t)4 * ld);
  if (n) HIP_TRY(g, hipMemcpyAsync(m.data(), d.obs_meas, m.size() * 8, hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  for (size_t s2 = 0; s2 < n && s2 < (size_t)d.n_obs; s2++) {
    HostFactor& f = g->factors[g->fslot_ids[F_PLANE_OBS][s2]];
    for (int k = 0; k < 4; k++) f.meas[k] = m[(size_t)k * ld + s2];
  }
  g->dev_meas_newer = false;
  g->pk_meas_ok = false;
  return PPS_OK;
}

int upload_measurements(pps_graph* g) {
  DevGraph& d = g->dev;
  std::vector<double> m;
  pack_soa<4>(g, F_PLANE_OBS, nullptr, false, m, (size_t)d.obs_ld);
  if (d.n_obs) HIP_TRY(g, hipMemcpyAsync(d.obs_meas, m.data(), m.size() * 8, hipMemcpyHostToDevice, g->stream));
  std::vector<double> m2;
  pack_soa<4>(g, F_PLANE_PRIOR, nullptr, false, m2, (size_t)d.lp_ld);
  if (d.n_lp) HIP_TRY(g, hipMemcpyAsync(d.lp_meas, m2.data(), m2.size() * 8, hipMemcpyHostToDevice, g->stream));
  // (the upload mirror no longer describes these arrays: the next topology upload sends them whole)
  g->up_unknown_meas = true;
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->meas_dirty = false;
  return PPS_OK;
}

int upload_all(pps_graph* g) {
  const double t0 = now_s();
  const bool was_grown_only = g->grown_only_upload;
  const bool tm = getenv("PPS_UPLOAD_TIMING") != nullptr;
  double tl = t0;
  auto lap = [&](const char* what) { if (!tm) return; const double t = now_s(); g->up_laps[what] += t - tl; tl = t; };
  int rc = ensure_device(g);
  if (rc != PPS_OK) return rc;
  if (g->dev_values_newer) { rc = download_state(g); if (rc != PPS_OK) return rc; }
  // Refreshed measurements may stay on the device across an upload that only appends (see the obs_meas upload below): same
  // arena, same slot with room for the new rows, same leading dimension, no re-popping edges (their slots sit behind the
  // fixed ones and would move).
  bool keep_meas = false;
  const size_t n_obs_on_device = (size_t)g->dev.n_obs;         // (free_device below resets g->dev)
  if (g->dev_meas_newer) {
    size_t n_obs_new = 0, n_lp_new = 0; bool any_repop = false;
    for (const HostFactor& f : g->factors) if (!f.deleted) { n_obs_new += f.type == F_PLANE_OBS; n_lp_new += f.type == F_PLANE_PRIOR; any_repop = any_repop || (f.type == F_PLANE_OBS && f.repop); }
    keep_meas = g->grown_only_upload && !g->up_unknown && !g->up_unknown_meas && g->up.spill == 0 && !any_repop && g->dev.n_obs == g->dev.n_obs_fixed &&
                j_capacity((int64_t)n_obs_new) == g->dev.obs_ld && j_capacity((int64_t)n_lp_new) == g->dev.lp_ld && g->slot_obs_meas < g->up_slots.size() &&
                true;
    if (!keep_meas) { rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  }
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->up_inflight = false;
  lap("1 state/meas download + syncs");
  free_device(g);
  g->spec_L = g->spec_U = g->spec_delta = nullptr; g->spec_result = nullptr;
  g->spec_pose = g->spec_plane = g->spec_chi2_partials = g->spec_dn_partials = nullptr; g->spec_ticket = nullptr;
  g->d_item_frame = g->d_item_plane = g->d_item_slot = g->d_frame_pose_slot = g->d_frame_seg_off = nullptr; g->d_fr_seg = nullptr;
  g->frames_dirty = true;
  g->snap_pose = g->snap_plane = nullptr; g->upload_version++;
  lap("2 free_device");
  if (!g->analyzed || g->analysis_stale) { rc = run_analysis(g); if (rc != PPS_OK) return rc; }
  lap("3 analysis");
  const Analysis& A = g->an;
  DevGraph& d = g->dev;
  d.n_pose = (int)g->pose_ids.size(); d.n_plane = (int)g->plane_ids.size();
  d.no_strip = getenv("PPS_NO_STRIP") ? 1 : 0;
  d.pose_ld = std::max(1, (d.n_pose + 63) / 64 * 64); d.plane_ld = std::max(1, (d.n_plane + 63) / 64 * 64);
#define TRY(x) do { rc = (x); if (rc != PPS_OK) return rc; } while (0)
  // every copy of the state is one block [poses | planes]: one transfer moves it (copies rotate by pointer pairs, so a
  // plane array always sits behind its pose array)
  const size_t state_doubles = (size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld;
  TRY(dev_alloc(g, &d.pose_est, state_doubles)); d.plane_est = d.pose_est + (size_t)7 * d.pose_ld;
  TRY(dev_alloc(g, &d.pose_lin, state_doubles)); d.plane_lin = d.pose_lin + (size_t)7 * d.pose_ld;
  std::vector<int> pv(d.n_pose), lv(d.n_plane);
  for (int s = 0; s < d.n_pose; s++) pv[s] = A.node_voff[g->nodes[g->pose_ids[s]].compact];
  for (int s = 0; s < d.n_plane; s++) lv[s] = A.node_voff[g->nodes[g->plane_ids[s]].compact];
  TRY(dev_upload(g, &d.pose_voff, pv)); TRY(dev_upload(g, &d.plane_voff, lv));
  // factors
  d.n_obs = (int)g->fslot_ids[F_PLANE_OBS].size(); d.n_odo = (int)g->fslot_ids[F_ODOMETRY].size();
  d.n_pp = (int)g->fslot_ids[F_POSE_PRIOR].size(); d.n_lp = (int)g->fslot_ids[F_PLANE_PRIOR].size();
  { int64_t base[4]; j_bases(g, base, nullptr); d.joff_obs = base[F_PLANE_OBS]; d.joff_odo = base[F_ODOMETRY]; d.joff_pp = base[F_POSE_PRIOR]; d.joff_lp = base[F_PLANE_PRIOR]; }
  auto idx_of = [&](int type, bool second) {
    std::vector<int> v(g->fslot_ids[type].size());
    for (size_t s = 0; s < v.size(); s++) {
      const HostFactor& f = g->factors[g->fslot_ids[type][s]];
      v[s] = g->nodes[second ? f.b : f.a].slot;
    }
    return v;
  };
  std::vector<double> tmp;
  // the previous upload's packed arrays are still right for the old factors when nothing was removed since (slots only append)
  const bool incr_pack = was_grown_only;
  d.obs_ld = (int)j_capacity(d.n_obs); d.odo_ld = (int)j_capacity(d.n_odo); d.pp_ld = (int)j_capacity(d.n_pp); d.lp_ld = (int)j_capacity(d.n_lp);
  {
    // one pass over the plane observations (a HostFactor is 300 bytes: four passes were four times the memory traffic), and
    // only over the new ones when the graph has just grown
    const std::vector<int>& ids = g->fslot_ids[F_PLANE_OBS];
    const size_t n = ids.size(), ld = (size_t)d.obs_ld;
    std::vector<int>& ia = g->pk_obs_a; std::vector<int>& ib = g->pk_obs_b;
    std::vector<double>& pm = g->pk_obs_m; std::vector<double>& pw = g->pk_obs_w;
    size_t s_begin = 0;
    if (incr_pack && g->pk_ld_obs == ld && g->pk_n_obs <= n && pm.size() == 4 * ld && g->pk_obs_ids.size() == g->pk_n_obs &&
        std::equal(g->pk_obs_ids.begin(), g->pk_obs_ids.end(), ids.begin())) s_begin = g->pk_n_obs;   // (re-popping edges sit behind the fixed ones: their slots move)
    ia.resize(n); ib.resize(n); pm.resize(4 * ld); pw.resize(6 * ld);
    for (size_t s2 = s_begin; s2 < n; s2++) {
      const HostFactor& f = g->factors[ids[s2]];
      ia[s2] = g->nodes[f.a].slot; ib[s2] = g->nodes[f.b].slot;
      for (int k = 0; k < 4; k++) pm[(size_t)k * ld + s2] = f.meas[k];
      for (int k = 0; k < 6; k++) pw[(size_t)k * ld + s2] = f.w[k];
    }
    if (s_begin > 0 && !g->pk_meas_ok)                                 // the host's measurements changed: those rows again
      for (size_t s2 = 0; s2 < s_begin; s2++) { const HostFactor& f = g->factors[ids[s2]]; for (int k = 0; k < 4; k++) pm[(size_t)k * ld + s2] = f.meas[k]; }
    g->pk_n_obs = n; g->pk_ld_obs = ld; g->pk_obs_ids.resize(s_begin); g->pk_obs_ids.insert(g->pk_obs_ids.end(), ids.begin() + (std::ptrdiff_t)s_begin, ids.end());
    TRY(dev_upload(g, &d.obs_pose, ia)); TRY(dev_upload(g, &d.obs_plane, ib));
    // Measurements that a device-side refresh has rewritten (pps_refresh_measurements) stay where they are when this upload
    // only appends: the host packs its (older) copies, the mirror holds the same bytes, so nothing is sent for them and the
    // device keeps the refreshed values; only the new observations travel.  dev_meas_newer stays set.
    g->slot_obs_meas = g->up_cursor;
    TRY(dev_upload_rows(g, &d.obs_meas, pm, 4, ld, n, g->up_unknown_meas, keep_meas ? std::min(n, n_obs_on_device) : kNoExact));
    TRY(dev_upload_rows(g, &d.obs_w, pw, 6, ld, n, false));
  }
  d.n_obs_fixed = g->n_obs_fixed;
  {
    const std::vector<int>& ids = g->fslot_ids[F_PLANE_OBS];
    const size_t n2 = ids.size() - (size_t)g->n_obs_fixed;
    if (n2 > 0) {
      tmp.assign(6 * n2, 0.0);
      for (size_t k = 0; k < n2; k++)
        for (int c = 0; c < 6; c++) tmp[(size_t)c * n2 + k] = g->factors[ids[g->n_obs_fixed + k]].ray[c];
      TRY(dev_upload(g, &d.obs_ray, tmp));
    }
  }
  {
    const std::vector<int>& ids = g->fslot_ids[F_ODOMETRY];
    const size_t n = ids.size(), ld = (size_t)d.odo_ld;
    std::vector<int>& ia = g->pk_odo_a; std::vector<int>& ib = g->pk_odo_b;
    std::vector<double>& pm = g->pk_odo_m; std::vector<double>& pw = g->pk_odo_w;
    size_t s_begin = 0;
    if (incr_pack && g->pk_ld_odo == ld && g->pk_n_odo <= n && pm.size() == 6 * ld && g->pk_odo_ids.size() == g->pk_n_odo &&
        std::equal(g->pk_odo_ids.begin(), g->pk_odo_ids.end(), ids.begin())) s_begin = g->pk_n_odo;
    ia.resize(n); ib.resize(n); pm.resize(6 * ld); pw.resize(21 * ld);
    for (size_t s2 = s_begin; s2 < n; s2++) {
      const HostFactor& f = g->factors[ids[s2]];
      ia[s2] = g->nodes[f.a].slot; ib[s2] = g->nodes[f.b].slot;
      for (int k = 0; k < 6; k++) pm[(size_t)k * ld + s2] = f.meas[k];
      for (int k = 0; k < 21; k++) pw[(size_t)k * ld + s2] = f.w[k];
    }
    g->pk_n_odo = n; g->pk_ld_odo = ld; g->pk_odo_ids.resize(s_begin); g->pk_odo_ids.insert(g->pk_odo_ids.end(), ids.begin() + (std::ptrdiff_t)s_begin, ids.end());
    TRY(dev_upload(g, &d.odo_a, ia)); TRY(dev_upload(g, &d.odo_b, ib));
    TRY(dev_upload_rows(g, &d.odo_meas, pm, 6, ld, n, false));
    TRY(dev_upload_rows(g, &d.odo_w, pw, 21, ld, n, false));
  }
  TRY(dev_upload(g, &d.pp_pose, idx_of(F_POSE_PRIOR, false)));
  pack_soa<6>(g, F_POSE_PRIOR, nullptr, false, tmp, (size_t)d.pp_ld); TRY(dev_upload_rows(g, &d.pp_meas, tmp, 6, (size_t)d.pp_ld, (size_t)d.n_pp, false));
  pack_soa<21>(g, F_POSE_PRIOR, nullptr, true, tmp, (size_t)d.pp_ld); TRY(dev_upload_rows(g, &d.pp_w, tmp, 21, (size_t)d.pp_ld, (size_t)d.n_pp, false));
  TRY(dev_upload(g, &d.lp_plane, idx_of(F_PLANE_PRIOR, false)));
  pack_soa<4>(g, F_PLANE_PRIOR, nullptr, false, tmp, (size_t)d.lp_ld); g->slot_lp_meas = g->up_cursor; TRY(dev_upload_rows(g, &d.lp_meas, tmp, 4, (size_t)d.lp_ld, (size_t)d.n_lp, g->up_unknown_meas));
  pack_soa<6>(g, F_PLANE_PRIOR, nullptr, true, tmp, (size_t)d.lp_ld); TRY(dev_upload_rows(g, &d.lp_w, tmp, 6, (size_t)d.lp_ld, (size_t)d.n_lp, false));
  g->up_unknown_meas = false;
  g->pk_meas_ok = true;
  lap("4 factor packing + diff");
  // linear system storage
  TRY(dev_alloc(g, &d.J, (size_t)A.J_size)); TRY(dev_alloc(g, &d.H, (size_t)A.H_size));
  TRY(dev_alloc(g, &d.L, (size_t)A.L_size)); TRY(dev_alloc(g, &d.U, (size_t)A.U_size));
  TRY(dev_alloc(g, &g->spec_L, (size_t)A.L_size)); TRY(dev_alloc(g, &g->spec_U, (size_t)A.U_size));
  const size_t delta_doubles = (size_t)std::max(1, A.n_scalars);   // (delta and the second delta sit in the zeroed block below)
  d.n_scalars = A.n_scalars;
  d.n_fronts = A.n_fronts; d.n_levels = A.n_levels; d.max_front = A.max_front; d.n_segs = A.n_segs; d.n_blocks = A.n_blocks;
  TRY(dev_upload(g, &d.f_p, A.f_p)); TRY(dev_upload(g, &d.f_b, A.f_b)); TRY(dev_upload(g, &d.f_poff, A.f_poff)); TRY(dev_upload(g, &d.pidx, A.pidx));
  TRY(dev_upload(g, &d.f_Loff, A.f_Loff)); TRY(dev_upload(g, &d.f_Uoff, A.f_Uoff));
  TRY(dev_upload(g, &d.f_bidx_off, A.f_bidx_off)); TRY(dev_upload(g, &d.bidx, A.bidx));
  TRY(dev_upload(g, &d.f_child_off, A.f_child_off)); TRY(dev_upload(g, &d.child, A.child));
  TRY(dev_upload(g, &d.f_cmap_off, A.f_cmap_off)); TRY(dev_upload(g, &d.cmap, A.cmap));
  TRY(dev_upload(g, &d.level_fronts, A.level_fronts));
  TRY(dev_upload(g, &d.f_asm_off, A.f_asm_off)); TRY(dev_upload(g, &d.asm_blk, A.asm_blk));
  TRY(dev_upload(g, &d.asm_lrow, A.asm_lrow)); TRY(dev_upload(g, &d.asm_lcol, A.asm_lcol));
  TRY(dev_upload(g, &d.blk_rows, A.blk_rows)); TRY(dev_upload(g, &d.blk_cols, A.blk_cols));
  TRY(dev_upload(g, &d.blk_size, A.blk_size)); TRY(dev_upload(g, &d.blk_nseg, A.blk_nseg));
  TRY(dev_upload(g, &d.blk_hoff, A.blk_hoff));
  TRY(dev_upload(g, &d.seg_blk, A.seg_blk)); TRY(dev_upload(g, &d.seg_c0, A.seg_c0)); TRY(dev_upload(g, &d.seg_cnt, A.seg_cnt));
  TRY(dev_upload(g, &d.seg_hoff, A.seg_hoff));
  TRY(dev_upload(g, &d.contrib, A.contrib));
  {
    std::vector<int> mseg;
    for (int bk = 0; bk < A.n_blocks; bk++) if (A.blk_nseg[bk] > 1) mseg.push_back(bk);
    d.n_mseg = (int)mseg.size();
    TRY(dev_upload(g, &d.mseg_blk, mseg));
  }
  TRY(dev_upload(g, &d.f_el_off, A.f_el_off)); TRY(dev_alloc(g, &d.el_tgt, (size_t)std::max<int64_t>(1, A.el_total)));   // filled by k_expand_el below
  TRY(dev_upload(g, &d.asm_el0, A.asm_el0)); TRY(dev_upload(g, &d.asm_fsz, A.asm_fsz));
  TRY(dev_upload(g, &d.f_ea_off, A.f_ea_off)); TRY(dev_alloc(g, &d.ea_tgt, (size_t)std::max<int64_t>(1, A.ea_total)));   // filled by k_expand_ea below
  TRY(dev_upload(g, &d.blk_doff, A.blk_doff)); TRY(dev_alloc(g, &d.blk_dst, (size_t)std::max(1, A.blk_doff[A.n_blocks])));
  TRY(dev_alloc(g, &d.Hf, (size_t)A.el_total));
  TRY(dev_upload(g, &d.grp_lvl_off, A.grp_lvl_off)); TRY(dev_upload(g, &d.glvl_front_off, A.glvl_front_off));
  TRY(dev_upload(g, &d.glvl_fronts, A.glvl_fronts));
  TRY(dev_upload(g, &d.frec, A.frec)); TRY(dev_upload(g, &d.crec, A.crec)); TRY(dev_upload(g, &d.srec, A.srec));
  TRY(dev_upload(g, &d.obs_dir, A.obs_dir)); TRY(dev_upload(g, &d.nd_segs, A.nd_segs)); d.n_nd_segs = (int)A.nd_segs.size();
  TRY(dev_upload(g, &d.cls_off, A.cls_off)); TRY(dev_upload(g, &d.cls_fronts, A.cls_fronts));
  if (g->use_dense) { TRY(dev_upload(g, &g->d_dw_asm, g->dw_asm)); TRY(dev_upload(g, &g->d_dw_pan, g->dw_pan)); TRY(dev_upload(g, &g->d_dw_trl, g->dw_trl)); }
  d.chi2_blocks = (d.n_obs + 255) / 256 + (d.n_odo + 255) / 256 + (d.n_pp + 255) / 256 + (d.n_lp + 255) / 256;
  TRY(dev_alloc(g, &d.chi2_partials, (size_t)std::max(1, d.chi2_blocks)));

  {
    // one zeroed block: [dn_partials | ticket | spec ticket | result record | the second factorisation's result record |
    // delta | the second delta]
    const size_t n_dn = (size_t)(d.n_pose + d.n_plane + 255) / 256 + 1;
    double* zb = nullptr;
    TRY(dev_alloc(g, &zb, n_dn + 2 + 8 + 2 * delta_doubles));
    HIP_TRY(g, hipMemsetAsync(zb, 0, (n_dn + 2 + 8 + 2 * delta_doubles) * 8, g->stream));
    d.delta = zb + n_dn + 10; g->spec_delta = d.delta + delta_doubles;
    d.dn_partials = zb;
    d.ticket = reinterpret_cast<unsigned int*>(zb + n_dn);
    g->spec_ticket = reinterpret_cast<unsigned int*>(zb + n_dn + 1);
    d.result_dev = zb + n_dn + 2; g->spec_result = zb + n_dn + 6;
    g->status_clean = true;
  }
  TRY(dev_alloc(g, &g->spec_pose, state_doubles + 1)); g->spec_plane = g->spec_pose + (size_t)7 * d.pose_ld;
  TRY(dev_alloc(g, &g->spec_chi2_partials, (size_t)std::max(1, d.chi2_blocks)));
  TRY(dev_alloc(g, &g->spec_dn_partials, (size_t)(d.n_pose + d.n_plane + 255) / 256 + 1));
  if (getenv("PPS_TRACE")) { TRY(dev_alloc(g, &d.trace, (size_t)A.n_fronts * 8)); HIP_TRY(g, hipMemset(d.trace, 0, (size_t)A.n_fronts * 64)); }
  // fronts that exceed the LDS limit run from a global workspace (one slab per front of the widest level)
  if (!g->use_band && !g->use_dense && A.max_front > lds_front_limit()) {
    const int fa = A.max_front + 1;
    d.gwork_stride = (int64_t)fa * (fa | 1);
    int widest = 0;
    for (int l = 0; l < A.n_levels; l++)
      if (g->level_max_front[l] > lds_front_limit()) widest = std::max(widest, A.level_off[l + 1] - A.level_off[l]);
    TRY(dev_alloc(g, &d.gwork, (size_t)d.gwork_stride * std::max(1, widest)));
  }
#undef TRY
  lap("5 index arrays + diff");
  rc = flush_uploads(g); if (rc != PPS_OK) return rc;
  rc = verify_uploads(g, "upload_all"); if (rc != PPS_OK) return rc;
  lap("6 flush");
  if (A.ea_total > 0) HIP_TRY(g, launch_expand_ea(d, A.n_fronts, g->stream));
  HIP_TRY(g, hipMemsetAsync(d.blk_dst, 0xff, sizeof(int) * (size_t)std::max(1, A.blk_doff[A.n_blocks]), g->stream));
  HIP_TRY(g, launch_expand_el(d, (int)A.asm_blk.size(), g->stream));
  g->topo_dirty = false;
  g->meas_dirty = false;
  g->grown_only_upload = true;
  lap("7 expand kernels + sync");
  // no sync: the solve that follows runs on the same stream (and ends with one); whatever writes the pinned buffers
  // again checks up_inflight or follows upload_all's opening sync
  rc = upload_state(g, false);
  lap("8 upload_state");
  g->stats.t_upload = now_s() - t0 - g->stats.t_analysis;
  return rc;
}

int prepare_solve(pps_graph* g) {
  int rc;
  if (g->n_live_nodes == 0) return fail(g, PPS_ESTATE, "empty graph");
  if (g->dev_ready) HIP_TRY(g, hipSetDevice(g->props.device));   // handles may be driven from any host thread
  if (g->topo_dirty || !g->dev_ready || g->dev.n_scalars == 0) { rc = upload_all(g); if (rc != PPS_OK) return rc; }
  if (g->host_values_newer) { rc = upload_state(g); if (rc != PPS_OK) return rc; }
  if (g->meas_dirty) { rc = upload_measurements(g); if (rc != PPS_OK) return rc; }
  return PPS_OK;
}

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
    HIP_TRY(g, hipEventRecord(g->k1_events[g->k1_used], g->stream));
    HIP_TRY(g, launch_linearize(g->dev, g->props.jacobian_mode, false, g->stream, guard));
    HIP_TRY(g, hipEventRecord(g->k1_events[g->k1_used + 1], g->stream));
    g->k1_used += 2;
  } else
  { PhaseTimer t(g, &g->stats.t_linearize); HIP_TRY(g, launch_linearize(g->dev, g->props.jacobian_mode, false, g->stream, guard)); }
  { PhaseTimer t(g, &g->stats.t_assemble); HIP_TRY(g, launch_hblocks(g->dev, g->stream, guard)); }
  if (!guard) g->stats.n_linearize++;
  return PPS_OK;
}

// delta = (J'J + lambda diag(J'J))^-1 J'b  (Optimizer::compute_gauss_newton_step, Optimizer.cpp:49-67)
int do_solve_on(pps_graph* g, const DevGraph& dv, double lambda, hipStream_t st_) {
  const Analysis& A = g->an;
  for (int st = 0; st < A.n_stages; st++)
    HIP_TRY(g, launch_band_factor(dv, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_factor[st],
                                  A.stage_max_front[st], lambda, st_));
  for (int st = A.n_stages - 1; st >= 0; st--)
    HIP_TRY(g, launch_band_solve(dv, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_solve[st],
                                 g->stage_max_panel[st], g->stage_max_grp_fronts[st], st_));
  return PPS_OK;
}

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
  return PPS_OK;
}

// est <-> lin by pointer: a rejected LM trial (estimate_to_linpoint, Optimizer.cpp:454) and the final
// linpoint_to_estimate (:466) need no data movement because the other copy is dead afterwards
void swap_state(pps_graph* g) {
  std::swap(g->dev.pose_est, g->dev.pose_lin);
  std::swap(g->dev.plane_est, g->dev.plane_lin);
}

int copy_state(pps_graph* g, bool est_to_lin) {
  const DevGraph& d = g->dev;
  double *ps = est_to_lin ? d.pose_est : d.pose_lin, *pd = est_to_lin ? d.pose_lin : d.pose_est;
  if (est_to_lin && g->lin_is_est) { g->lin_is_est = false; return PPS_OK; }       // upload_state has just written both copies
  g->lin_is_est = false;
  HIP_TRY(g, hipMemcpyAsync(pd, ps, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
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

}  // namespace

// =========================================================================================
extern "C" {

void pps_default_props(pps_props* p) {
  if (!p) return;
  p->epsilon2 = 1e-2 * 0.1;      // Properties.h:94 x Mapping.cpp:37
  p->epsilon_abs = 1e-3 * 0.1;   // Properties.h:98 x Mapping.cpp:38
  p->epsilon_rel = 1e-5 * 0.1;   // Properties.h:99 x Mapping.cpp:39
  p->max_iterations = 500;
  p->lm_lambda0 = 1e-6;
  p->lm_lambda_factor = 10.;
  p->jacobian_mode = PPS_JAC_NUMERIC;
  p->device = 0;
  p->verbose = 0;
}

int pps_version(void) { return PPS_VERSION; }

const char* pps_last_error(const pps_graph* g) { return g ? g->err.c_str() : "null handle"; }

int pps_graph_create(const pps_props* props, pps_graph** out) {
  if (!out) return PPS_EINVAL;
  pps_graph* g = new (std::nothrow) pps_graph();
  if (!g) return PPS_ENOMEM;
  if (props) g->props = *props; else pps_default_props(&g->props);
  *out = g;
  return PPS_OK;
}

int pps_graph_destroy(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (!g->up_laps.empty()) {
    std::vector<std::pair<std::string, double>> v(g->up_laps.begin(), g->up_laps.end());
    std::sort(v.begin(), v.end());
    for (auto& kv : v) fprintf(stderr, "[upload] %-34s %9.3f ms total\n", kv.first.c_str(), 1e3 * kv.second);
  }
  if (g->acache) analysis_cache_free(g->acache);
  if (g->dev_ready) {
    (void)hipSetDevice(g->props.device);
    (void)hipStreamSynchronize(g->stream);
    free_device(g);
    release_arenas(g);
    if (g->host_result) (void)hipHostFree(g->host_result);
    if (g->ev[0]) (void)hipEventDestroy(g->ev[0]);
    if (g->ev[1]) (void)hipEventDestroy(g->ev[1]);
    for (hipEvent_t e : g->k1_events) (void)hipEventDestroy(e);
    if (g->d_lms) (void)hipFree(g->d_lms);
    if (g->d_queries) (void)hipFree(g->d_queries);
    if (g->d_results) (void)hipFree(g->d_results);
    if (g->d_lm_planes) (void)hipFree(g->d_lm_planes);
    (void)hipStreamDestroy(g->stream);
  }
  delete g;
  return PPS_OK;
}

int pps_get_props(const pps_graph* g, pps_props* out) {
  if (!g || !out) return PPS_EINVAL;
  *out = g->props;
  return PPS_OK;
}
int pps_set_props(pps_graph* g, const pps_props* p) {
  if (!g || !p) return PPS_EINVAL;
  if (g->dev_ready && p->device != g->props.device) return fail(g, PPS_ESTATE, "device cannot change after the first solve");
  g->props = *p;
  return PPS_OK;
}

static int add_node(pps_graph* g, int type, const double* v, int nv, int* id) {
  if (!g || !v) return PPS_EINVAL;
  for (int k = 0; k < nv; k++) if (!std::isfinite(v[k])) return fail(g, PPS_EINVAL, "non-finite node value");
  if (g->dev_values_newer) { int rc = download_state(g); if (rc != PPS_OK) return rc; }
  HostNode n{};
  n.type = type;
  for (int k = 0; k < nv; k++) n.v[k] = v[k];
  if (type == NODE_PLANE) normalize4(n.v);
  n.deleted = false; n.compact = n.slot = -1;
  g->nodes.push_back(n);
  g->n_live_nodes++;
  g->dim_nodes += type == NODE_POSE ? 6 : 3;
  g->topo_dirty = true; g->analysis_stale = true; g->host_values_newer = true;
  if (id) *id = (int)g->nodes.size() - 1;
  return PPS_OK;
}

int pps_add_pose(pps_graph* g, const double tq[7], int* id) { return add_node(g, NODE_POSE, tq, 7, id); }
int pps_add_plane(pps_graph* g, const double abcd[4], int* id) { return add_node(g, NODE_PLANE, abcd, 4, id); }

static bool live_node(const pps_graph* g, int id, int type) {
  return id >= 0 && id < (int)g->nodes.size() && !g->nodes[id].deleted && g->nodes[id].type == type;
}

static int add_factor(pps_graph* g, int type, int a, int b, const double* meas, int nm, const double* ut, int nw, int* fid) {
  if (!g || !meas || !ut) return PPS_EINVAL;
  for (int k = 0; k < nm; k++) if (!std::isfinite(meas[k])) return fail(g, PPS_EINVAL, "non-finite measurement");
  for (int k = 0; k < nw; k++) if (!std::isfinite(ut[k])) return fail(g, PPS_EINVAL, "non-finite sqrtinf");
  HostFactor f{};
  f.type = type; f.a = a; f.b = b; f.deleted = false; f.slot = -1;
  for (int k = 0; k < nm; k++) f.meas[k] = meas[k];
  if (nm == 4) normalize4(f.meas);
  for (int k = 0; k < nw; k++) f.w[k] = ut[k];
  g->factors.push_back(f);
  g->n_live_factors++;
  g->dim_measure += kFDim[type];
  g->topo_dirty = true; g->analysis_stale = true;
  if (fid) *fid = (int)g->factors.size() - 1;
  return PPS_OK;
}

int pps_add_pose_prior(pps_graph* g, int pose, const double meas6[6], const double ut[21], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, pose, NODE_POSE)) return fail(g, PPS_EINVAL, "pose prior: unknown pose id");
  return add_factor(g, F_POSE_PRIOR, pose, -1, meas6, 6, ut, 21, fid);
}
int pps_add_odometry(pps_graph* g, int p1, int p2, const double meas6[6], const double ut[21], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, p1, NODE_POSE) || !live_node(g, p2, NODE_POSE) || p1 == p2) return fail(g, PPS_EINVAL, "odometry: bad pose ids");
  return add_factor(g, F_ODOMETRY, p1, p2, meas6, 6, ut, 21, fid);
}
int pps_add_plane_obs(pps_graph* g, int pose, int plane, const double meas4[4], const double ut[6], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, pose, NODE_POSE) || !live_node(g, plane, NODE_PLANE)) return fail(g, PPS_EINVAL, "plane obs: bad node ids");
  return add_factor(g, F_PLANE_OBS, pose, plane, meas4, 4, ut, 6, fid);
}
// Pose3d_Plane3d_Factor2 (src/isam_plane3d.h:314-424): same nodes / noise / log-map residual, but the measured
// plane is re-derived from the edge's two ground rays and the CURRENT pose at every evaluation.
int pps_add_plane_obs2(pps_graph* g, int pose, int plane, const double meas4[4], const double ray6[6], const double ut[6],
                       int* fid) {
  if (!g || !ray6) return PPS_EINVAL;
  for (int k = 0; k < 6; k++) if (!std::isfinite(ray6[k])) return fail(g, PPS_EINVAL, "non-finite edge ray");
  int id = -1;
  int rc = pps_add_plane_obs(g, pose, plane, meas4, ut, &id);
  if (rc != PPS_OK) return rc;
  g->factors[id].repop = 1;
  memcpy(g->factors[id].ray, ray6, sizeof g->factors[id].ray);
  if (fid) *fid = id;
  return PPS_OK;
}

// precompute_edge_ray (src/isam_plane3d.h:361-373): fp32 product invK * (u,v,1) per end point, cast to fp64
int pps_edge_ray(const float invK[9], const float seg2d[4], double ray6[6]) {
  if (!invK || !seg2d || !ray6) return PPS_EINVAL;
  for (int e = 0; e < 2; e++) {
    const float u = seg2d[2 * e], v = seg2d[2 * e + 1];
    for (int i = 0; i < 3; i++) ray6[3 * e + i] = (double)(invK[i * 3 + 0] * u + invK[i * 3 + 1] * v + invK[i * 3 + 2] * 1.f);
  }
  return PPS_OK;
}

int pps_add_plane_prior(pps_graph* g, int plane, const double meas4[4], const double ut[6], int* fid) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, plane, NODE_PLANE)) return fail(g, PPS_EINVAL, "plane prior: unknown plane id");
  return add_factor(g, F_PLANE_PRIOR, plane, -1, meas4, 4, ut, 6, fid);
}

int pps_set_measurement(pps_graph* g, int fid, const double meas4[4]) { return pps_set_measurements(g, 1, &fid, meas4); }

int pps_set_measurements(pps_graph* g, int n, const int* fids, const double* meas4) {
  if (!g || !fids || !meas4 || n < 0) return PPS_EINVAL;
  if (g->dev_meas_newer) { int rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  for (int i = 0; i < n; i++) {
    const int fid = fids[i];
    if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "set_measurement: unknown factor id");
    HostFactor& f = g->factors[fid];
    if (f.type != F_PLANE_OBS && f.type != F_PLANE_PRIOR) return fail(g, PPS_EINVAL, "set_measurement: not a plane factor");
    for (int k = 0; k < 4; k++) {
      if (!std::isfinite(meas4[4 * i + k])) return fail(g, PPS_EINVAL, "non-finite measurement");
      f.meas[k] = meas4[4 * i + k];
    }
    normalize4(f.meas);
  }
  g->meas_dirty = true; g->pk_meas_ok = false;
  return PPS_OK;
}

int pps_remove_factor(pps_graph* g, int fid) {
  if (!g) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "remove_factor: unknown id");
  g->factors[fid].deleted = true;
  g->grown_only = false; g->grown_only_upload = false;
  g->n_live_factors--;
  g->dim_measure -= kFDim[g->factors[fid].type];
  g->topo_dirty = true; g->analysis_stale = true;
  return PPS_OK;
}

int pps_remove_node(pps_graph* g, int nid) {
  if (!g) return PPS_EINVAL;
  if (nid < 0 || nid >= (int)g->nodes.size() || g->nodes[nid].deleted) return fail(g, PPS_EINVAL, "remove_node: unknown id");
  if (g->dev_values_newer) { int rc = download_state(g); if (rc != PPS_OK) return rc; }
  for (size_t i = 0; i < g->factors.size(); i++) {
    HostFactor& f = g->factors[i];
    if (!f.deleted && (f.a == nid || f.b == nid)) pps_remove_factor(g, (int)i);
  }
  g->nodes[nid].deleted = true;
  g->grown_only = false; g->grown_only_upload = false;
  g->n_live_nodes--;
  g->dim_nodes -= g->nodes[nid].type == NODE_POSE ? 6 : 3;
  g->topo_dirty = true; g->analysis_stale = true; g->host_values_newer = true;
  return PPS_OK;
}

int pps_update(pps_graph* g) {
  if (!g) return PPS_EINVAL;
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
  { PhaseTimer t(g, &g->stats.t_retract_chi2); HIP_TRY(g, launch_retract_apply(g->dev, g->stream)); }   // apply_exmap (:183)
  double chi2, dn; bool notpd;
  rc = read_result(g, true, &chi2, &dn, &notpd); if (rc != PPS_OK) return rc;
  resolve_k1_events(g);
  if (notpd) {
    // the step is garbage: put the estimate back (lin still holds it) instead of handing NaNs to the caller
    rc = copy_state(g, false); if (rc != PPS_OK) return rc;
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
    return fail(g, PPS_ENOTPD, "normal equations not positive definite");
  }
  g->dev_values_newer = true; g->lin_is_est = false;
  g->status_clean = true;                                         // the chi2 kernel took the flag with it
  g->stats.chi2_final = chi2; g->stats.last_delta_norm = dn; g->stats.lambda_final = 0;
  g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
  return PPS_OK;
}

static int lm_solve(pps_graph* g, int* iterations);

// the device copy of a handle is given up after a failed solve: streams drained, nothing on the device is trusted any more --
// the next upload sends the whole arena (up_unknown: the mirror says nothing about the device, measurements included) and the
// estimate falls back to the host's node values
static void abandon_device_copy(pps_graph* g) {
  if (!g->dev_ready) return;
  (void)hipStreamSynchronize(g->stream);
  g->topo_dirty = true; g->dev_values_newer = false; g->dev_meas_newer = false;
  g->up_unknown = true; g->up_unknown_meas = true; g->pk_meas_ok = false; g->status_clean = false;
}

// A failure in the middle of a solve (a HIP error: lost device, out of memory) leaves est / lin possibly exchanged and
// speculative work in flight.  Both streams are drained and the device copy is abandoned: the next call uploads again from
// the host's node values -- the estimate falls back to the last state the host has seen -- instead of reading half-updated
// buffers.  (PPS_ENOTPD is not such a failure: the solve ran to its end.)
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
  // three state copies: x = the linearisation point (d.pose_lin), t[0] / t[1] = x (+) delta for the two damping values
  double *t_pose[2] = {d.pose_est, g->spec_pose}, *t_plane[2] = {d.plane_est, g->spec_plane};
  int rc = copy_state(g, true); if (rc != PPS_OK) return rc;       // estimate_to_linpoint (Optimizer.cpp:376): est is dead from here on
  double seqs[2] = {0, 0};
  auto enqueue_dual = [&](double lam) -> int {
    DualAlt alt{g->spec_L, g->spec_U, g->spec_delta, g->spec_result, g->spec_chi2_partials, g->spec_dn_partials, g->spec_ticket,
                lam * prop.lm_lambda_factor};
    for (int st = 0; st < A.n_stages; st++)
      HIP_TRY(g, launch_band_factor_dual(d, alt, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_factor[st],
                                         A.stage_max_front[st], lam, g->stream));
    for (int st = A.n_stages - 1; st >= 0; st--)
      HIP_TRY(g, launch_band_solve(d, A.stage_grp_off[st], A.stage_grp_off[st + 1] - A.stage_grp_off[st], g->stage_nw_solve[st],
                                   g->stage_max_panel[st], g->stage_max_grp_fronts[st], g->stream, &alt));
    g->stats.n_factorize += 2;
    g->seq += 1.0; seqs[0] = g->seq;
    g->seq2 += 1.0; seqs[1] = g->seq2;
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
  const bool spec_lin = !getenv("PPS_NO_SPEC_LIN");
  int spec_pair = -1;                    // K1 event pair of the queued speculative linearisation
  auto enqueue_spec_lin = [&](double err) -> int {
    if (!spec_lin) return PPS_OK;
    LinGuard gd{{d.result_dev, g->spec_result}, {t_pose[0], t_pose[1]}, {t_plane[0], t_plane[1]}, err, 1, 0};
    spec_pair = g->profiling == 1 ? g->k1_used / 2 : -1;
    return do_linearize(g, &gd);
  };
  auto drop_spec_lin = [&]() { if (spec_pair >= 0 && (size_t)spec_pair < g->k1_skip.size()) g->k1_skip[spec_pair] = 1; spec_pair = -1; };
  rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot0, seq0); if (rc != PPS_OK) return rc;
  double error = slot0[0];
  g->stats.chi2_initial = error;
  rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
  rc = wait_result(g, slot[0], seqs[0]); if (rc != PPS_OK) return rc;
  int cur = 0;                           // which of the two trials the loop is looking at
  bool have_next = true;                 // trial 1 of the last launch is the step for the next lambda after a rejection
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
      // the accepted copy becomes the linearisation point; the old one is the spare now
      std::swap(d.pose_lin, t_pose[cur]); std::swap(d.plane_lin, t_plane[cur]);
      if (spec_lin) { g->stats.n_linearize++; spec_pair = -1; }    // relinearise (:444): queued already, at this very copy
      else { rc = do_linearize(g); if (rc != PPS_OK) return rc; }
      rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;      // (:458)
      rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
      cur = 0; have_next = true;
      rc = wait_result(g, slot[0], seqs[0]); if (rc != PPS_OK) return rc;
    } else {
      g->stats.lm_trials_rejected++;
      lambda *= prop.lm_lambda_factor;                             // estimate_to_linpoint (:454): x was never overwritten
      if (have_next) {                                             // computed alongside: nothing to launch
        cur = 1; have_next = false;
        rc = wait_result(g, slot[1], seqs[1]); if (rc != PPS_OK) return rc;
      } else {
        drop_spec_lin();                                           // both trials rejected: its kernels left J and H alone
        rc = enqueue_dual(lambda); if (rc != PPS_OK) return rc;    // (:458), same J and H
        rc = enqueue_spec_lin(error); if (rc != PPS_OK) return rc;
        cur = 0; have_next = true;
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
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->dev_values_newer = true; g->lin_is_est = false;
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
  if (g->use_band && g->profiling < 2 && !g->dev.trace && !getenv("PPS_NO_DUAL")) return lm_solve_dual(g, iterations, t0);      // (PPS_NO_DUAL: the loop-forms parity test)
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
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  g->dev_values_newer = true; g->lin_is_est = false;
  g->status_clean = true;                                         // every solve was followed by its chi2 kernel
  resolve_k1_events(g);
  g->stats.lm_iterations = num_iter;
  g->stats.chi2_final = error; g->stats.lambda_final = lambda; g->stats.last_delta_norm = dnorm;
  g->stats.t_total = now_s() - t0; g->stats.n_launches = (int)(launch_count() - g->launches0);
  if (g->dev.trace) {
    const Analysis& A = g->an;
    std::vector<long long> tr((size_t)A.n_fronts * 8);
    (void)hipMemcpy(tr.data(), g->dev.trace, tr.size() * 8, hipMemcpyDeviceToHost);
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
  if (iterations) *iterations = num_iter;
  g->stats.lm_trials_notpd = n_notpd;
  if (last_notpd) return fail(g, PPS_ENOTPD, "normal equations not positive definite at the last LM trial");
  return PPS_OK;
}

// =========================================================================================
// pps_multi: G independent graphs solved side by side.  One C2-size solve is a dependency chain that keeps a few dozen
// of the 256 CUs busy; here every kernel of an LM trial is launched ONCE for all graphs (blockIdx.y = graph) and the
// graphs advance in lockstep rounds -- a round = [re-linearise the graphs whose last trial was accepted] + factor +
// solve + trial step + chi2 for every graph still iterating.  Per-graph lambda / accept / reject run on the host from
// one 32-byte record per graph and round, with exactly the control flow (and the arithmetic) of pps_batch_optimize.
// =========================================================================================
struct pps_multi {
  std::vector<pps_graph*> gs;
  int device = 0;
  std::string err;
  hipStream_t stream = nullptr;
  DevGraph* d_gs = nullptr; size_t cap_gs = 0;
  BatchStage* d_stage = nullptr; size_t cap_stage = 0;
  BatchAlt* d_alt = nullptr; size_t cap_alt = 0;          // dual-lambda form: second factorisation + the three state copies per graph
  double* results = nullptr; size_t cap_results = 0;      // pinned: 12 doubles per graph (8 used by the single-lambda form)
  double seq = 0.0;
  int rounds = 0; double t_total = 0;
  // profiling (pps_multi_set_profiling): HIP events at the phase boundaries of every round, resolved after the solve
  int profiling = 0;
  std::vector<hipEvent_t> evs; size_t ev_used = 0;
  double t_phase[5] = {0, 0, 0, 0, 0};     // K1 | K2 | factor | back-substitution | trial step + chi2   [seconds, device]
  long long n_relin = 0, n_solves = 0;      // graphs re-linearised / factorised, summed over the rounds
};

static int mfail(pps_multi* m, int code, const std::string& msg) { if (m) m->err = msg; return code; }
#define MHIP(m, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return mfail(m, PPS_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)

int pps_multi_create(int n, pps_graph* const* graphs, pps_multi** out) {
  if (!out || n < 1 || !graphs) return PPS_EINVAL;
  for (int i = 0; i < n; i++) {
    if (!graphs[i]) return PPS_EINVAL;
    if (graphs[i]->props.device != graphs[0]->props.device) return PPS_EINVAL;
    for (int j = 0; j < i; j++) if (graphs[j] == graphs[i]) return PPS_EINVAL;
  }
  pps_multi* m = new (std::nothrow) pps_multi();
  if (!m) return PPS_ENOMEM;
  m->gs.assign(graphs, graphs + n);
  m->device = graphs[0]->props.device;
  *out = m;
  return PPS_OK;
}

int pps_multi_destroy(pps_multi* m) {
  if (!m) return PPS_EINVAL;
  if (m->stream) {
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    (void)hipStreamDestroy(m->stream);
  }
  for (hipEvent_t e : m->evs) (void)hipEventDestroy(e);
  if (m->d_gs) (void)hipFree(m->d_gs);
  if (m->d_stage) (void)hipFree(m->d_stage);
  if (m->d_alt) (void)hipFree(m->d_alt);
  if (m->results) (void)hipHostFree(m->results);
  delete m;
  return PPS_OK;
}

const char* pps_multi_last_error(const pps_multi* m) { return m ? m->err.c_str() : "null handle"; }

static int multi_optimize(pps_multi* m, int* iterations, int* status);

int pps_multi_optimize(pps_multi* m, int* iterations, int* status) {
  if (!m) return PPS_EINVAL;
  const int rc = multi_optimize(m, iterations, status);
  if (rc != PPS_OK && rc != PPS_ENOTPD && rc != PPS_EINVAL && rc != PPS_ESTATE) {       // a HIP failure in the middle of the rounds: as a failed single solve
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (pps_graph* g : m->gs) abandon_device_copy(g);
  }
  return rc;
}

static int multi_optimize(pps_multi* m, int* iterations, int* status) {
  const double t0 = now_s();
  const int G = (int)m->gs.size();
  if (hipSetDevice(m->device) != hipSuccess) return mfail(m, PPS_EHIP, "hipSetDevice failed (no HIP device: there is no CPU fallback)");
  // ---- every graph analysed, uploaded and idle; all of them must take the wave-per-front path ----
  int mode = m->gs[0]->props.jacobian_mode, max_stages = 0;
  for (int i = 0; i < G; i++) {
    pps_graph* g = m->gs[i];
    reset_solve_stats(g);
    g->tr_lambda.clear(); g->tr_chi2.clear(); g->tr_acc.clear();
    int rc = prepare_solve(g);
    if (rc != PPS_OK) return mfail(m, rc, "graph " + std::to_string(i) + ": " + g->err);
    if (!g->use_band) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has fronts beyond the wave-per-front kernels (loop closures): solve it through its own handle");
    if (g->n_live_factors == 0) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has no factors");
    if (g->props.jacobian_mode != mode) return mfail(m, PPS_EINVAL, "all graphs of a batch share one jacobian_mode");
    if (g->an.n_stages > 32) return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + ": elimination tree too deep for the batched schedule");
    MHIP(m, hipStreamSynchronize(g->stream));
    g->status_clean = false;
    max_stages = std::max(max_stages, g->an.n_stages);
  }
  // both damping values of a linearisation in the same launches (lm_solve_dual's scheme): every uploaded handle has its second
  // factor / state set
  for (int i = 0; i < G; i++)
    if (!(m->gs[i]->spec_L && m->gs[i]->spec_U && m->gs[i]->spec_delta && m->gs[i]->spec_pose && m->gs[i]->spec_result))
      return mfail(m, PPS_ESTATE, "graph " + std::to_string(i) + " has no second factor set (not uploaded)");
  const bool dual = true;
  if (!m->stream) MHIP(m, hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  if (m->cap_results < (size_t)G) {
    if (m->results) (void)hipHostFree(m->results);
    m->results = nullptr; m->cap_results = 0;
    MHIP(m, hipHostMalloc(reinterpret_cast<void**>(&m->results), sizeof(double) * 12 * (size_t)G, hipHostMallocDefault));
    m->cap_results = G;
  }
  memset(m->results, 0, sizeof(double) * 12 * (size_t)G);
  // ---- device tables: the graphs' records and their band schedules ----
  std::vector<DevGraph> hg(G);
  std::vector<BatchStage> hs((size_t)std::max(1, max_stages) * G, BatchStage{0, 0});
  for (int i = 0; i < G; i++) {
    hg[i] = m->gs[i]->dev;
    const Analysis& A = m->gs[i]->an;
    for (int stg = 0; stg < A.n_stages; stg++) hs[(size_t)stg * G + i] = BatchStage{A.stage_grp_off[stg], A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]};
  }
  if (m->cap_gs < (size_t)G) { if (m->d_gs) (void)hipFree(m->d_gs); m->d_gs = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_gs), sizeof(DevGraph) * (size_t)G)); m->cap_gs = G; }
  if (m->cap_stage < hs.size()) { if (m->d_stage) (void)hipFree(m->d_stage); m->d_stage = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_stage), sizeof(BatchStage) * hs.size())); m->cap_stage = hs.size(); }
  MHIP(m, hipMemcpy(m->d_gs, hg.data(), sizeof(DevGraph) * (size_t)G, hipMemcpyHostToDevice));
  MHIP(m, hipMemcpy(m->d_stage, hs.data(), sizeof(BatchStage) * hs.size(), hipMemcpyHostToDevice));
  // ---- launch geometry per chunk of kBatchMax graphs ----
  int n_cu = 256;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, m->device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount; }
  const int n_chunks = (G + kBatchMax - 1) / kBatchMax;
  std::vector<BatchGeom> geom(n_chunks);
  const size_t lds_budget = 150 * 1024;
  for (int c = 0; c < n_chunks; c++) {
    BatchGeom& q = geom[c];
    q.n_stages = max_stages;
    int max_panel[32] = {0};
    for (int stg = 0; stg < 32; stg++) q.stage_reg_only[stg] = true;
    bool level_ok = true;
    for (int i = c * kBatchMax; i < std::min(G, (c + 1) * kBatchMax); i++) {
      const pps_graph* g = m->gs[i];
      const DevGraph& d = g->dev;
      const Analysis& A = g->an;
      q.lin_blocks = std::max(q.lin_blocks, (d.n_obs_fixed + 7) / 8 + (d.n_odo + 7) / 8 + (d.n_pp + 7) / 8 + (d.n_lp + 7) / 8);
      q.lin_obs_blocks = std::max(q.lin_obs_blocks, (d.n_obs_fixed + 127) / 128);
      q.lin_rest_blocks = std::max(q.lin_rest_blocks, (d.n_odo + 127) / 128 + (d.n_pp + 127) / 128 + (d.n_lp + 127) / 128);
      q.repop_blocks = std::max(q.repop_blocks, (d.n_obs - d.n_obs_fixed + 63) / 64);
      q.hblocks = std::max(q.hblocks, (d.n_segs + 3) / 4);
      q.hblocks_nd = std::max(q.hblocks_nd, (d.n_nd_segs + 15) / 16);
      q.hreduce = std::max(q.hreduce, d.n_mseg);
      q.retract = std::max(q.retract, (d.n_pose + d.n_plane + 255) / 256);
      q.chi2 = std::max(q.chi2, d.chi2_blocks);
      q.n_factors_total += (long long)d.n_obs + d.n_odo + d.n_pp + d.n_lp;
      q.n_levels = std::max(q.n_levels, A.n_levels);
      if (A.n_levels > 64 || A.max_front + 1 > band_reg_rows() || g->dev.trace) level_ok = false;
      for (int l = 0; l < A.n_levels && l < 64; l++) {
        for (int c2 = 0; c2 < 3; c2++) q.lvl_cls_blocks[l][c2] = std::max(q.lvl_cls_blocks[l][c2], (A.cls_off[3 * l + c2 + 1] - A.cls_off[3 * l + c2] + 3) / 4);
        q.lvl_blocks[l] = std::max(q.lvl_blocks[l], (A.cls_off[3 * l + 3] - A.cls_off[3 * l] + 3) / 4);
      }
      for (int stg = 0; stg < A.n_stages; stg++) {
        q.stage_groups[stg] = std::max(q.stage_groups[stg], A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]);
        q.stage_nw_factor[stg] = std::max(q.stage_nw_factor[stg], g->stage_nw_factor[stg]);
        q.stage_nw_solve[stg] = std::max(q.stage_nw_solve[stg], g->stage_nw_solve[stg]);
        q.stage_per_wave_factor[stg] = std::max(q.stage_per_wave_factor[stg], A.stage_max_front[stg]);   // (max front for now: sized below)
        q.stage_max_front[stg] = std::max(q.stage_max_front[stg], A.stage_max_front[stg]);
        max_panel[stg] = std::max(max_panel[stg], g->stage_max_panel[stg]);
        q.stage_grp_fronts[stg] = std::max(q.stage_grp_fronts[stg], g->stage_max_grp_fronts[stg]);
        if (A.stage_max_front[stg] + 1 > band_reg_rows() || g->dev.trace) q.stage_reg_only[stg] = false;
      }
    }
    // the lane-parallel central differences (32 lanes per factor, 13 of them idle) are the low-latency form; from a few
    // hundred thousand factors per launch the thread-per-factor form has the higher throughput
    q.lin_thread_form = q.n_factors_total > 200000 || getenv("PPS_MULTI_THREAD_FORM");      // (PPS_MULTI_THREAD_FORM / PPS_MULTI_LEVELS: forced onto small batches by the parity test)
    q.k1_direct = q.lin_thread_form || mode == PPS_JAC_ANALYTIC;   // the analytic sweep always runs one thread per factor
    // throughput over latency from the same size on: a launch per tree level and size class instead of a launch per band
    q.level_form = level_ok && (q.n_factors_total > 200000 || getenv("PPS_MULTI_LEVELS"));
    { int mp = 1; for (int stg = 0; stg < max_stages; stg++) mp = std::max(mp, max_panel[stg]); q.solve_per_wave_all = (int)(band_solve_lds_bytes(mp) / sizeof(double)); }
    for (int stg = 0; stg < max_stages; stg++) {
      q.stage_per_wave_factor[stg] = (int)(band_lds_bytes(q.stage_per_wave_factor[stg], q.stage_reg_only[stg]) / sizeof(double));
      q.stage_per_wave_solve[stg] = (int)(band_solve_lds_bytes(max_panel[stg]) / sizeof(double));
      const size_t fw = (size_t)q.stage_per_wave_factor[stg] * sizeof(double), sw = (size_t)q.stage_per_wave_solve[stg] * sizeof(double);
      const size_t xbytes = (size_t)q.stage_grp_fronts[stg] * band_max_rows() * sizeof(double);
      if (fw > lds_budget || xbytes + sw > lds_budget) return mfail(m, PPS_ESTATE, "a band group of this batch does not fit the LDS: solve the graphs through their own handles");
      q.stage_nw_factor[stg] = (int)std::max<size_t>(1, std::min<size_t>(q.stage_nw_factor[stg], lds_budget / fw));
      q.stage_nw_solve[stg] = (int)std::max<size_t>(1, std::min<size_t>(q.stage_nw_solve[stg], (lds_budget - xbytes) / sw));
      // Throughput, not latency, is what a batch is for.  A band group is a sub-tree (8 + 4 + 2 + 1 fronts on C2): walked
      // by 8 waves, half of the wave-slots -- and the LDS they hold -- idle on its upper levels.  When the chunk has more
      // groups than the device has wave-slots, fewer waves per group keep every slot on a front (2 waves: 94 % instead of
      // 47 %); the groups of the upper stages stay wide, there the tree depth is the cost.
      long long total_groups = 0;
      for (int i = c * kBatchMax; i < std::min(G, (c + 1) * kBatchMax); i++) {
        const Analysis& A = m->gs[i]->an;
        if (stg < A.n_stages) total_groups += (dual ? 2 : 1) * (A.stage_grp_off[stg + 1] - A.stage_grp_off[stg]);
      }
      if (total_groups > 0) {
        const long long slots_f = (long long)n_cu * std::max<size_t>(1, lds_budget / fw);
        const long long slots_s = (long long)n_cu * std::max<size_t>(1, (lds_budget - std::min(lds_budget / 2, xbytes)) / sw);
        q.stage_nw_factor[stg] = (int)std::max<long long>(1, std::min<long long>(q.stage_nw_factor[stg], (slots_f + total_groups - 1) / total_groups));
        q.stage_nw_solve[stg] = (int)std::max<long long>(1, std::min<long long>(q.stage_nw_solve[stg], (slots_s + total_groups - 1) / total_groups));
      }
    }
  }
  // ---- dual-lambda form: every graph walks lm_solve_dual's scheme, in lockstep rounds of one linearisation each ----
  {
    std::vector<BatchAlt> ha(G);
    for (int i = 0; i < G; i++) {
      pps_graph* g = m->gs[i];
      ha[i] = BatchAlt{g->spec_L, g->spec_U, g->spec_delta, g->spec_result, g->spec_chi2_partials, g->spec_dn_partials, g->spec_ticket,
                       {g->dev.pose_est, g->dev.pose_lin, g->spec_pose}, {g->dev.plane_est, g->dev.plane_lin, g->spec_plane}};
    }
    if (m->cap_alt < (size_t)G) { if (m->d_alt) (void)hipFree(m->d_alt); m->d_alt = nullptr; MHIP(m, hipMalloc(reinterpret_cast<void**>(&m->d_alt), sizeof(BatchAlt) * (size_t)G)); m->cap_alt = G; }
    MHIP(m, hipMemcpy(m->d_alt, ha.data(), sizeof(BatchAlt) * (size_t)G, hipMemcpyHostToDevice));
    struct LMD { double lambda, error, dnorm; int num_iter, cur, xsel; bool done, have_next, relin, active, last_notpd, trial_taken; int n_notpd; };
    std::vector<LMD> lm(G);
    for (int i = 0; i < G; i++) lm[i] = LMD{m->gs[i]->props.lm_lambda0, 0.0, 0.0, 0, 0, 0, false, true, true, true, false, false, 0};
    auto make_args = [&](int c) {
      BatchArgs a{};
      a.gs = m->d_gs; a.stage_tab = m->d_stage; a.results = m->results; a.n_total = G; a.b0 = c * kBatchMax;
      a.n = std::min(G, (c + 1) * kBatchMax) - a.b0; a.seq = m->seq;
      a.alt = m->d_alt; a.rstride = 12;
      for (int k = 0; k < a.n; k++) {
        const LMD& q = lm[a.b0 + k];
        a.lambda[k] = q.lambda; a.lambda2[k] = q.lambda * m->gs[a.b0 + k]->props.lm_lambda_factor;
        a.xsel[k] = (unsigned char)q.xsel;
        a.flags[k] = (unsigned char)((q.active ? BF_ACTIVE : 0) | (q.relin ? BF_RELIN : 0));
      }
      return a;
    };
    auto wait_round = [&]() -> int {
      const double tw = now_s();
      unsigned spins = 0;
      for (int i = 0; i < G; i++) {
        if (!lm[i].active) continue;
        for (int slot = 1; slot <= 2; slot++) {
          volatile double* r = m->results + 12 * (size_t)i + 4 * slot;
          while (r[3] != m->seq) {
            if ((++spins & 0x3ff) == 0 && now_s() - tw > 2.0) {
              MHIP(m, hipStreamSynchronize(m->stream));
              if (r[3] != m->seq) return mfail(m, PPS_EHIP, "result record of graph " + std::to_string(i) + " did not arrive");
            }
          }
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return PPS_OK;
    };
    m->ev_used = 0; m->n_relin = 0; m->n_solves = 0;
    for (double& t : m->t_phase) t = 0;
    auto mark = [&]() {
      if (!m->profiling) return;
      if (m->ev_used == m->evs.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; m->evs.push_back(e); }
      (void)hipEventRecord(m->evs[m->ev_used++], m->stream);
    };
    auto next_event = [&]() -> hipEvent_t {
      if (!m->profiling) return nullptr;
      if (m->ev_used == m->evs.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; m->evs.push_back(e); }
      return m->evs[m->ev_used++];
    };
    // one round of one chunk: e0 | K1 | e1 | K2 (+ chi2 at x) | e2 | factor x 2 | e3 | solve x 2 | e4 | both trials | e5
    auto run_round = [&](const BatchArgs& a, const BatchGeom& q, bool first, bool any_relin) -> int {
      if (first) MHIP(m, launch_batch_begin_dual(a, q, m->stream));
      mark();
      if (any_relin) MHIP(m, launch_batch_linearize(a, q, q.lin_thread_form ? mode | 2 : mode, m->stream));
      mark();
      if (any_relin) MHIP(m, launch_batch_hblocks(a, q, m->stream));
      if (first) MHIP(m, launch_batch_chi2(a, q, 0, m->stream));
      mark();
      hipEvent_t ef = next_event();
      MHIP(m, launch_batch_solve(a, q, m->stream, ef));
      mark();
      MHIP(m, launch_batch_trial_dual(a, q, m->stream));
      mark();
      return PPS_OK;
    };
    // the part of lm_solve_dual's loop that needs no launch: consume the verdicts that are on the host.  Returns with the
    // graph done, or active (and possibly relin) for the next round.
    auto advance = [&](int i) {
      LMD& q = lm[i];
      pps_graph* g = m->gs[i];
      const pps_props& prop = g->props;
      q.active = false; q.relin = false;
      for (;;) {
        if (!((prop.max_iterations <= 0 || q.num_iter < prop.max_iterations) && q.dnorm > prop.epsilon2 && q.error > prop.epsilon_abs)) { q.done = true; return; }
        q.num_iter++;
        const double* rec = m->results + 12 * (size_t)i + 4 * (1 + q.cur);
        const double error_new = rec[0];
        const double error_diff = q.error - error_new;
        const bool accepted = error_diff > 0.;
        g->tr_lambda.push_back(q.lambda); g->tr_chi2.push_back(error_new); g->tr_acc.push_back(accepted ? 1 : 0);
        if (accepted) {
          g->stats.lm_trials_accepted++;
          if (error_diff < prop.epsilon_rel * q.error) { q.error = error_new; q.trial_taken = true; q.done = true; return; }   // (:431-434)
          q.lambda /= prop.lm_lambda_factor;
          q.error = error_new;
          q.xsel = (q.xsel + 1 + q.cur) % 3;                           // the accepted copy is the linearisation point now
          q.relin = true; q.active = true; q.cur = 0; q.have_next = true;
          g->stats.n_linearize++; g->stats.n_factorize += 2;
          return;
        }
        g->stats.lm_trials_rejected++;
        q.lambda *= prop.lm_lambda_factor;
        if (q.have_next) {                                             // the step for this lambda was computed alongside
          q.cur = 1; q.have_next = false;
          const double* rb = m->results + 12 * (size_t)i + 8;
          q.dnorm = std::sqrt(rb[1]); q.last_notpd = rb[2] != 0.0; q.n_notpd += q.last_notpd ? 1 : 0;
          continue;
        }
        q.active = true; q.cur = 0; q.have_next = true;               // both rejected: same J and H, two more damping values
        g->stats.n_factorize += 2;
        return;
      }
    };
    m->seq += 1.0; m->rounds = 0;
    for (int c = 0; c < n_chunks; c++) {
      const BatchArgs a = make_args(c);
      int rc = run_round(a, geom[c], true, true); if (rc != PPS_OK) return rc;
    }
    m->n_relin += G; m->n_solves += 2 * (long long)G;
    { int rc = wait_round(); if (rc != PPS_OK) return rc; }
    m->rounds++;
    for (int i = 0; i < G; i++) {
      pps_graph* g = m->gs[i];
      const double* r0 = m->results + 12 * (size_t)i;
      lm[i].error = r0[0]; g->stats.chi2_initial = r0[0];
      lm[i].dnorm = std::sqrt(r0[5]); lm[i].last_notpd = r0[6] != 0.0; lm[i].n_notpd = lm[i].last_notpd ? 1 : 0;
      g->stats.n_linearize = 1; g->stats.n_factorize = 2;
    }
    for (;;) {
      int n_active = 0;
      for (int i = 0; i < G; i++) { if (!lm[i].done) advance(i); else { lm[i].active = false; lm[i].relin = false; } n_active += lm[i].active ? 1 : 0; }
      if (n_active == 0) break;
      m->seq += 1.0;
      for (int c = 0; c < n_chunks; c++) {
        const BatchArgs a = make_args(c);
        bool any = false, any_relin = false;
        for (int k = 0; k < a.n; k++) { any = any || (a.flags[k] & BF_ACTIVE); any_relin = any_relin || (a.flags[k] & BF_RELIN); }
        if (!any) continue;
        for (int k = 0; k < a.n; k++) { m->n_solves += (a.flags[k] & BF_ACTIVE) ? 2 : 0; m->n_relin += (a.flags[k] & BF_RELIN) ? 1 : 0; }
        int rc = run_round(a, geom[c], false, any_relin); if (rc != PPS_OK) return rc;
      }
      { int rc = wait_round(); if (rc != PPS_OK) return rc; }
      m->rounds++;
      for (int i = 0; i < G; i++) {
        if (!lm[i].active) continue;
        const double* r1 = m->results + 12 * (size_t)i + 4;
        lm[i].dnorm = std::sqrt(r1[1]);
        lm[i].last_notpd = r1[2] != 0.0;
        lm[i].n_notpd += lm[i].last_notpd ? 1 : 0;
      }
    }
    MHIP(m, hipStreamSynchronize(m->stream));
    for (size_t k = 0; k + 6 <= m->ev_used; k += 6) {
      const hipEvent_t* e = &m->evs[k];
      for (int ph = 0; ph < 5; ph++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e[ph], e[ph + 1]) == hipSuccess) m->t_phase[ph] += 1e-3 * ms;
      }
    }
    int first_bad = PPS_OK;
    m->t_total = now_s() - t0;
    for (int i = 0; i < G; i++) {
      pps_graph* g = m->gs[i];
      const LMD& q = lm[i];
      // linpoint_to_estimate (:466): the accepted, converged trial -- or the linearisation point when the pending step is dropped
      const int fin = q.trial_taken ? (q.xsel + 1 + q.cur) % 3 : q.xsel;
      double* const sp[3] = {ha[i].pose[0], ha[i].pose[1], ha[i].pose[2]};
      double* const sl[3] = {ha[i].plane[0], ha[i].plane[1], ha[i].plane[2]};
      g->dev.pose_est = sp[fin]; g->dev.plane_est = sl[fin];
      g->dev.pose_lin = sp[(fin + 1) % 3]; g->dev.plane_lin = sl[(fin + 1) % 3];
      g->spec_pose = sp[(fin + 2) % 3]; g->spec_plane = sl[(fin + 2) % 3];
      g->dev_values_newer = true; g->lin_is_est = false;
      g->stats.lm_iterations = q.num_iter; g->stats.chi2_final = q.error; g->stats.lambda_final = q.lambda; g->stats.last_delta_norm = q.dnorm;
      g->stats.lm_trials_notpd = q.n_notpd; g->stats.t_total = m->t_total;
      if (iterations) iterations[i] = q.num_iter;
      const int st_i = q.last_notpd ? PPS_ENOTPD : PPS_OK;
      if (st_i != PPS_OK) g->err = "normal equations not positive definite at the last LM trial";
      if (status) status[i] = st_i;
      if (st_i != PPS_OK && first_bad == PPS_OK) first_bad = st_i;
    }
    if (first_bad != PPS_OK) return mfail(m, first_bad, "at least one graph ended on a factorisation that was not positive definite (see status[])");
    return PPS_OK;
  }
}

int pps_multi_set_profiling(pps_multi* m, int level) { if (!m) return PPS_EINVAL; m->profiling = level > 0 ? 1 : 0; return PPS_OK; }

int pps_multi_phase_times(const pps_multi* m, double sec[5], long long counts[2]) {
  if (!m || !sec) return PPS_EINVAL;
  for (int k = 0; k < 5; k++) sec[k] = m->t_phase[k];
  if (counts) { counts[0] = m->n_relin; counts[1] = m->n_solves; }
  return PPS_OK;
}

int pps_multi_rounds(const pps_multi* m, int* rounds) { if (!m || !rounds) return PPS_EINVAL; *rounds = m->rounds; return PPS_OK; }

int pps_chi2(pps_graph* g, double* chi2) {
  if (!g || !chi2) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  double dn; bool np;
  return read_result(g, true, chi2, &dn, &np);
}

int pps_num_nodes(const pps_graph* g, int* n) { if (!g || !n) return PPS_EINVAL; *n = g->n_live_nodes; return PPS_OK; }
int pps_num_factors(const pps_graph* g, int* n) { if (!g || !n) return PPS_EINVAL; *n = g->n_live_factors; return PPS_OK; }

int pps_get_pose(pps_graph* g, int id, double tq[7]) {
  if (!g || !tq) return PPS_EINVAL;
  if (!live_node(g, id, NODE_POSE)) return fail(g, PPS_EINVAL, "get_pose: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(tq, g->nodes[id].v, 7 * sizeof(double));
  return PPS_OK;
}
int pps_get_plane(pps_graph* g, int id, double abcd[4]) {
  if (!g || !abcd) return PPS_EINVAL;
  if (!live_node(g, id, NODE_PLANE)) return fail(g, PPS_EINVAL, "get_plane: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(abcd, g->nodes[id].v, 4 * sizeof(double));
  return PPS_OK;
}
int pps_set_pose(pps_graph* g, int id, const double tq[7]) {
  if (!g || !tq) return PPS_EINVAL;
  if (!live_node(g, id, NODE_POSE)) return fail(g, PPS_EINVAL, "set_pose: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(g->nodes[id].v, tq, 7 * sizeof(double));
  g->host_values_newer = true;
  return PPS_OK;
}
int pps_set_plane(pps_graph* g, int id, const double abcd[4]) {
  if (!g || !abcd) return PPS_EINVAL;
  if (!live_node(g, id, NODE_PLANE)) return fail(g, PPS_EINVAL, "set_plane: unknown id");
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  memcpy(g->nodes[id].v, abcd, 4 * sizeof(double));
  normalize4(g->nodes[id].v);
  g->host_values_newer = true;
  return PPS_OK;
}

static int get_bulk(pps_graph* g, int type, int n, const int* ids, double* out, int w) {
  if (!g || !out || n < 0) return PPS_EINVAL;
  int rc = download_state(g); if (rc != PPS_OK) return rc;
  if (ids) {
    for (int i = 0; i < n; i++) {
      if (!live_node(g, ids[i], type)) return fail(g, PPS_EINVAL, "bulk get: unknown id");
      memcpy(out + (size_t)w * i, g->nodes[ids[i]].v, w * sizeof(double));
    }
  } else {
    int k = 0;
    for (size_t i = 0; i < g->nodes.size() && k < n; i++)
      if (!g->nodes[i].deleted && g->nodes[i].type == type) { memcpy(out + (size_t)w * k, g->nodes[i].v, w * sizeof(double)); k++; }
    if (k != n) return fail(g, PPS_EINVAL, "bulk get: count mismatch");
  }
  return PPS_OK;
}
int pps_get_poses(pps_graph* g, int n, const int* ids, double* out) { return get_bulk(g, NODE_POSE, n, ids, out, 7); }
int pps_get_planes(pps_graph* g, int n, const int* ids, double* out) { return get_bulk(g, NODE_PLANE, n, ids, out, 4); }

int pps_save_state(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  const DevGraph& d = g->dev;
  if (!g->snap_pose || g->snap_version != g->upload_version) {
    // (re)allocate with the current leading dimensions; owned by the allocation list of this upload
    rc = dev_alloc(g, &g->snap_pose, (size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld); if (rc != PPS_OK) return rc;
    g->snap_plane = g->snap_pose + (size_t)7 * d.pose_ld;
    g->snap_version = g->upload_version;
  }
  HIP_TRY(g, hipMemcpyAsync(g->snap_pose, d.pose_est, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  return PPS_OK;
}

int pps_restore_state(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (!g->snap_pose || g->snap_version != g->upload_version || g->topo_dirty) return fail(g, PPS_ESTATE, "no snapshot for the current topology");
  if (g->host_values_newer) return fail(g, PPS_ESTATE, "host values were modified after the snapshot");
  const DevGraph& d = g->dev;
  HIP_TRY(g, hipMemcpyAsync(d.pose_est, g->snap_pose, ((size_t)7 * d.pose_ld + (size_t)4 * d.plane_ld) * 8, hipMemcpyDeviceToDevice, g->stream));
  g->lin_is_est = false;
  g->dev_values_newer = true; g->lin_is_est = false;
  return PPS_OK;
}

int pps_get_stats(const pps_graph* g, pps_stats* out) {
  if (!g || !out) return PPS_EINVAL;
  *out = g->stats;
  out->n_poses = out->n_planes = 0;
  for (const auto& n : g->nodes) if (!n.deleted) (n.type == NODE_POSE ? out->n_poses : out->n_planes)++;
  out->n_factors = g->n_live_factors;
  out->dim_nodes = g->dim_nodes; out->dim_measure = g->dim_measure;
  return PPS_OK;
}

int pps_get_trace(const pps_graph* g, int cap, double* lambda, double* chi2, int* accepted, int* n) {
  if (!g || !n) return PPS_EINVAL;
  *n = (int)g->tr_lambda.size();
  for (int i = 0; i < *n && i < cap; i++) {
    if (lambda) lambda[i] = g->tr_lambda[i];
    if (chi2) chi2[i] = g->tr_chi2[i];
    if (accepted) accepted[i] = g->tr_acc[i];
  }
  return PPS_OK;
}

int pps_set_profiling(pps_graph* g, int level) { if (!g) return PPS_EINVAL; g->profiling = level < 0 ? 0 : level; return PPS_OK; }

int pps_factor_shape(const pps_graph* g, int fid, int* dim, int* cols) {
  if (!g) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return PPS_EINVAL;
  const HostFactor& f = g->factors[fid];
  if (dim) *dim = kFDim[f.type];
  if (cols) *cols = (g->nodes[f.a].type == NODE_POSE ? 6 : 3) + (f.b >= 0 ? (g->nodes[f.b].type == NODE_POSE ? 6 : 3) : 0);
  return PPS_OK;
}

int pps_eval_factor(pps_graph* g, int fid, int mode, double* J, double* r) {
  if (!g || !J || !r) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "eval_factor: unknown id");
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  HIP_TRY(g, launch_linearize(g->dev, mode, true, g->stream));
  const HostFactor& f = g->factors[fid];
  const int m = kFDim[f.type];
  const int da = g->nodes[f.a].type == NODE_POSE ? 6 : 3;
  const int db = f.b >= 0 ? (g->nodes[f.b].type == NODE_POSE ? 6 : 3) : 0;
  const int64_t base[4] = {g->dev.joff_pp, g->dev.joff_odo, g->dev.joff_obs, g->dev.joff_lp};
  std::vector<double> buf(kJSize[f.type]);
  HIP_TRY(g, hipMemcpyAsync(buf.data(), g->dev.J + base[f.type] + (int64_t)f.slot * kJSize[f.type], buf.size() * 8,
                            hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  const int cols = da + db;
  for (int i = 0; i < m; i++) {
    for (int j = 0; j < da; j++) J[i * cols + j] = buf[i * da + j];
    for (int j = 0; j < db; j++) J[i * cols + da + j] = buf[m * da + i * db + j];
    r[i] = buf[m * (da + db) + i];
  }
  return PPS_OK;
}

int pps_analyze(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  if (g->n_live_nodes == 0) return fail(g, PPS_ESTATE, "empty graph");
  // run_analysis re-assigns the node / factor slots; anything that is newer on the device still lives in the OLD slot layout
  // and has to come home first (a refresh or a solve followed by an edit, then this call)
  if (g->dev_ready) {
    int rc = download_state(g); if (rc != PPS_OK) return rc;
    rc = download_measurements(g); if (rc != PPS_OK) return rc;
    g->topo_dirty = true;            // the device arrays no longer match the slot tables: the next solve uploads again
  }
  return run_analysis(g);
}

int pps_analysis_reuse(const pps_graph* g, int* fronts_kept, int* fronts_total) {
  if (!g) return PPS_EINVAL;
  analysis_cache_stats(g->acache, fronts_kept, fronts_total);
  return PPS_OK;
}

int pps_analysis_dump(pps_graph* g, int64_t cap, int32_t* out, int64_t* needed) {
  if (!g || !needed) return PPS_EINVAL;
  if (!g->analyzed || g->analysis_stale) { int rc = pps_analyze(g); if (rc != PPS_OK) return rc; }
  std::vector<int32_t> v;
  dump_analysis(g->an, v);
  // append the compact-id tables the tests need: node id -> compact id, factor id -> joff
  v.push_back((int32_t)g->nodes.size());
  for (const auto& n : g->nodes) v.push_back(n.deleted ? -1 : n.compact);
  int64_t base[4]; j_bases(g, base, nullptr);
  v.push_back((int32_t)g->factors.size());
  for (const auto& f : g->factors) v.push_back(f.deleted ? -1 : (int32_t)(base[f.type] + (int64_t)f.slot * kJSize[f.type]));
  *needed = (int64_t)v.size();
  if (out && cap >= (int64_t)v.size()) memcpy(out, v.data(), v.size() * sizeof(int32_t));
  return PPS_OK;
}

// K1 alone on the solver's stream: `iters` back-to-back sweeps of the handle's graph between two HIP events
// (an event pair around ONE ~10 us launch also measures the command processor's event handling, about as long again)
int pps_time_linearize(pps_graph* g, int mode, int iters, double* sec_per_launch) {
  if (!g || iters < 1 || !sec_per_launch) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  for (int k = 0; k < 3; k++) HIP_TRY(g, launch_linearize(g->dev, mode, false, g->stream));
  HIP_TRY(g, hipEventRecord(g->ev[0], g->stream));
  for (int k = 0; k < iters; k++) HIP_TRY(g, launch_linearize(g->dev, mode, false, g->stream));
  HIP_TRY(g, hipEventRecord(g->ev[1], g->stream));
  HIP_TRY(g, hipEventSynchronize(g->ev[1]));
  float ms = 0;
  HIP_TRY(g, hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
  *sec_per_launch = 1e-3 * ms / iters;
  return PPS_OK;
}

int pps_bench_sweep(pps_graph* g, int mode, int replicas, int iters, double* sec_per_sweep, int64_t* n_plane_edges,
                    int64_t* n_odo_edges) {
  if (!g || replicas < 1 || iters < 1 || !sec_per_sweep) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  if (g->dev.n_obs_fixed != g->dev.n_obs) return fail(g, PPS_ESTATE, "bench_sweep: graph holds re-popping plane edges (Factor2)");
  DevGraph d = g->dev;   // shallow copy with replicated edge arrays
  std::vector<void*> tmp;
  auto rep = [&](auto** ptr, size_t count_per) -> int {
    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(ptr)>>;
    if (count_per == 0) return PPS_OK;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, count_per * sizeof(T) * (size_t)replicas);
    if (e != hipSuccess) return hip_fail(g, e, "hipMalloc(bench)");
    tmp.push_back(p);
    for (int r = 0; r < replicas; r++) {
      e = hipMemcpyAsync(static_cast<char*>(p) + (size_t)r * count_per * sizeof(T), *ptr, count_per * sizeof(T),
                         hipMemcpyDeviceToDevice, g->stream);
      if (e != hipSuccess) return hip_fail(g, e, "hipMemcpyAsync(bench)");
    }
    *ptr = static_cast<T*>(p);
    return PPS_OK;
  };
  auto cleanup = [&]() { (void)hipStreamSynchronize(g->stream); for (void* p : tmp) (void)hipFree(p); };
#define BT(x) do { rc = (x); if (rc != PPS_OK) { cleanup(); return rc; } } while (0)
  BT(rep(&d.obs_pose, (size_t)d.n_obs)); BT(rep(&d.obs_plane, (size_t)d.n_obs));
  BT(rep(&d.obs_meas, (size_t)4 * d.obs_ld)); BT(rep(&d.obs_w, (size_t)6 * d.obs_ld));
  BT(rep(&d.odo_a, (size_t)d.n_odo)); BT(rep(&d.odo_b, (size_t)d.n_odo));
  BT(rep(&d.odo_meas, (size_t)6 * d.odo_ld)); BT(rep(&d.odo_w, (size_t)21 * d.odo_ld));
#undef BT
  const size_t slab = (size_t)d.n_obs * 30 + (size_t)d.n_odo * 78;
  double* Jbig = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&Jbig), slab * sizeof(double) * (size_t)replicas);
  if (e != hipSuccess) { cleanup(); return hip_fail(g, e, "hipMalloc(Jbig)"); }
  tmp.push_back(Jbig);
  e = launch_sweep_bench(d, mode, replicas, Jbig, -1, g->stream);   // warm-up
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  float total_ms = 0, part_ms[2] = {0, 0};
  for (int it = 0; it < iters && e == hipSuccess; it++) {
    (void)hipEventRecord(g->ev[0], g->stream);
    e = launch_sweep_bench(d, mode, replicas, Jbig, -1, g->stream);
    (void)hipEventRecord(g->ev[1], g->stream);
    if (e == hipSuccess) e = hipEventSynchronize(g->ev[1]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
    total_ms += ms;
    for (int part = 0; part < 2 && e == hipSuccess; part++) {       // each launch on its own
      (void)hipEventRecord(g->ev[0], g->stream);
      e = launch_sweep_bench(d, mode, replicas, Jbig, part, g->stream);
      (void)hipEventRecord(g->ev[1], g->stream);
      if (e == hipSuccess) e = hipEventSynchronize(g->ev[1]);
      (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
      part_ms[part] += ms;
    }
  }
  cleanup();
  if (e != hipSuccess) return hip_fail(g, e, "sweep bench");
  sec_per_sweep[1] = 1e-3 * part_ms[0] / iters;
  sec_per_sweep[2] = 1e-3 * part_ms[1] / iters;
  *sec_per_sweep = 1e-3 * total_ms / iters;
  if (n_plane_edges) *n_plane_edges = (int64_t)d.n_obs * replicas;
  if (n_odo_edges) *n_odo_edges = (int64_t)d.n_odo * replicas;
  return PPS_OK;
}

int pps_frames_set_calibration(pps_graph* g, const float invK[9]) {
  if (!g || !invK) return PPS_EINVAL;
  memcpy(g->frames_invK, invK, sizeof g->frames_invK);
  return PPS_OK;
}

int pps_frames_add(pps_graph* g, int pose_id, int n_seg, const float* seg2d, const int* fids, int* frame_id) {
  if (!g || n_seg < 0 || !fids || (n_seg > 0 && !seg2d)) return PPS_EINVAL;
  if (!live_node(g, pose_id, NODE_POSE)) return fail(g, PPS_EINVAL, "frames_add: unknown pose id");
  for (int j = 0; j <= n_seg; j++) {
    const int fid = fids[j];
    if (fid < 0) continue;
    if (fid >= (int)g->factors.size() || g->factors[fid].deleted || g->factors[fid].type != F_PLANE_OBS || g->factors[fid].a != pose_id)
      return fail(g, PPS_EINVAL, "frames_add: fid is not a plane observation of this pose");
  }
  const int f = (int)g->fr_pose.size();
  g->fr_pose.push_back(pose_id);
  for (int k = 0; k < 4 * n_seg; k++) g->fr_seg.push_back(seg2d[k]);
  g->fr_seg_off.push_back(g->fr_seg_off.back() + n_seg);
  for (int j = 0; j <= n_seg; j++) { g->fr_item_frame.push_back(f); g->fr_item_plane.push_back(j); g->fr_item_fid.push_back(fids[j]); }
  g->frames_dirty = true;
  if (frame_id) *frame_id = f;
  return PPS_OK;
}

int pps_refresh_measurements(pps_graph* g) {
  if (!g) return PPS_EINVAL;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  if (g->fr_item_frame.empty()) return PPS_OK;
  if (g->frames_dirty) {
    std::vector<int> slot(g->fr_item_fid.size()), pslot(g->fr_pose.size());
    for (size_t i = 0; i < slot.size(); i++) {
      const int fid = g->fr_item_fid[i];
      slot[i] = (fid >= 0 && !g->factors[fid].deleted && !g->factors[fid].repop) ? g->factors[fid].slot : -1;
      if (slot[i] >= 0 && g->nodes[g->fr_pose[g->fr_item_frame[i]]].deleted) slot[i] = -1;
    }
    for (size_t f = 0; f < pslot.size(); f++) pslot[f] = g->nodes[g->fr_pose[f]].deleted ? 0 : g->nodes[g->fr_pose[f]].slot;
    // tables live in the allocation list of the current upload; older copies are simply abandoned until then.  (The topology
    // upload may still be copying out of the pinned mirror and the patch buffer this is about to write.)
    if (g->up_inflight) { HIP_TRY(g, hipStreamSynchronize(g->stream)); g->up_inflight = false; }
    rc = dev_upload(g, &g->d_item_frame, g->fr_item_frame); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_item_plane, g->fr_item_plane); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_item_slot, slot); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_frame_pose_slot, pslot); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_frame_seg_off, g->fr_seg_off); if (rc != PPS_OK) return rc;
    rc = dev_upload(g, &g->d_fr_seg, g->fr_seg); if (rc != PPS_OK) return rc;
    rc = flush_uploads(g); if (rc != PPS_OK) return rc;
    rc = verify_uploads(g, "frames"); if (rc != PPS_OK) return rc;
    g->up_inflight = true;                                  // (whoever writes the mirror next waits for this copy)
    g->frames_dirty = false;
  }
  RefreshArgs a{};
  a.n_items = (int)g->fr_item_frame.size();
  a.item_frame = g->d_item_frame; a.item_plane = g->d_item_plane; a.item_slot = g->d_item_slot;
  a.frame_pose_slot = g->d_frame_pose_slot; a.frame_seg_off = g->d_frame_seg_off; a.seg2d = g->d_fr_seg;
  memcpy(a.invK, g->frames_invK, sizeof a.invK);
  a.pose_est = g->dev.pose_est; a.pose_ld = g->dev.pose_ld;
  a.obs_meas = g->dev.obs_meas; a.n_obs = g->dev.n_obs; a.obs_ld = g->dev.obs_ld;
  HIP_TRY(g, launch_refresh_measurements(a, g->stream));
  g->dev_meas_newer = true;
  return PPS_OK;
}

int pps_get_measurement(pps_graph* g, int fid, double meas4[4]) {
  if (!g || !meas4) return PPS_EINVAL;
  if (fid < 0 || fid >= (int)g->factors.size() || g->factors[fid].deleted) return fail(g, PPS_EINVAL, "get_measurement: unknown factor id");
  const HostFactor& f = g->factors[fid];
  if (f.type != F_PLANE_OBS && f.type != F_PLANE_PRIOR) return fail(g, PPS_EINVAL, "get_measurement: not a plane factor");
  if (g->dev_meas_newer) { int rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  memcpy(meas4, f.meas, 4 * sizeof(double));
  return PPS_OK;
}


// ---- plane data association (Mapper_mono::findClosestPlane, src/Mapping.cpp:256-397) ----

void pps_assoc_default_params(pps_assoc_params* p) {
  if (!p) return;
  p->edge_asso_2ddist = 50; p->edge_asso_planedist = 4; p->edge_asso_proj = 0.5; p->edge_asso_angle = 60.0;   // Mapping.h:72-76
  p->assoc_near_frames = 5;
}

int pps_landmark_update(pps_graph* g, int plane_id, int frame_plane_indice, int frame_seq_id, const float seg2d[4],
                        const float seg3d_xy[4]) {
  if (!g) return PPS_EINVAL;
  if (!live_node(g, plane_id, NODE_PLANE)) return fail(g, PPS_EINVAL, "landmark_update: unknown plane id");
  auto it = g->lm_of_plane.find(plane_id);
  int idx;
  if (it == g->lm_of_plane.end()) {
    idx = (int)g->lms.size();
    g->lms.push_back(pps_graph::Landmark{plane_id, 0, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}});
    g->lm_of_plane[plane_id] = idx;
  } else idx = it->second;
  pps_graph::Landmark& L = g->lms[idx];
  L.fpi = frame_plane_indice; L.seq = frame_seq_id;
  for (int k = 0; k < 4; k++) { L.seg2d[k] = seg2d ? seg2d[k] : 0.f; L.seg3d[k] = seg3d_xy ? seg3d_xy[k] : 0.f; }
  g->lms_dirty = true;
  return PPS_OK;
}

int pps_landmark_set_merged(pps_graph* g, int plane_id) {
  if (!g) return PPS_EINVAL;
  auto it = g->lm_of_plane.find(plane_id);
  if (it == g->lm_of_plane.end()) return fail(g, PPS_EINVAL, "landmark_set_merged: plane id is not a landmark");
  g->lms[it->second].deleted = 1;
  g->lms_dirty = true;
  return PPS_OK;
}

int pps_find_closest_planes(pps_graph* g, const double est_pose[7], int frame_seq_id, int n, const double* planes_local,
                            const int* frame_plane_indice, const float* seg2d, const float* seg3d_xy,
                            const pps_assoc_params* prm, int* best_plane_id, double* best_err) {
  if (!g || !est_pose || n < 0 || (n > 0 && (!planes_local || !frame_plane_indice || !seg2d || !seg3d_xy)) || !best_plane_id || !best_err)
    return PPS_EINVAL;
  pps_assoc_params P;
  if (prm) P = *prm; else pps_assoc_default_params(&P);
  if (n == 0) return PPS_OK;
  const int nl = (int)g->lms.size();
  if (nl == 0) { for (int i = 0; i < n; i++) { best_plane_id[i] = -1; best_err[i] = -1.0; } return PPS_OK; }
  int rc = ensure_device(g);
  if (rc != PPS_OK) return rc;
  HIP_TRY(g, hipSetDevice(g->props.device));
  // landmark planes: straight from the solver state when it is current, else a packed copy of the host values
  const bool state_current = g->dev_ready && !g->topo_dirty && !g->host_values_newer && g->dev.n_plane > 0;
  AssocArgs a{};
  if (!state_current) {
    if (g->dev_values_newer) { rc = download_state(g); if (rc != PPS_OK) return rc; }
    if ((size_t)nl > g->d_lm_planes_cap) {
      if (g->d_lm_planes) (void)hipFree(g->d_lm_planes);
      g->d_lm_planes_cap = std::max<size_t>(256, 2 * (size_t)nl);
      HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_lm_planes), 4 * g->d_lm_planes_cap * sizeof(double)));
    }
    std::vector<double> pl(4 * (size_t)nl, 0.0);
    for (int i = 0; i < nl; i++) {
      const HostNode& nd = g->nodes[g->lms[i].plane_id];
      for (int k = 0; k < 4; k++) pl[(size_t)k * nl + i] = nd.v[k];
    }
    HIP_TRY(g, hipMemcpyAsync(g->d_lm_planes, pl.data(), pl.size() * sizeof(double), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));   // `pl` leaves scope
    a.plane_est = g->d_lm_planes; a.plane_ld = nl;
  } else {
    a.plane_est = g->dev.plane_est; a.plane_ld = g->dev.plane_ld;
  }
  if (g->lms_dirty || g->lms_upload_version != (state_current ? g->upload_version : -2)) {
    if ((size_t)nl > g->d_lms_cap) {
      if (g->d_lms) (void)hipFree(g->d_lms);
      g->d_lms_cap = std::max<size_t>(256, 2 * (size_t)nl);
      HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_lms), g->d_lms_cap * sizeof(AssocLandmark)));
    }
    std::vector<AssocLandmark> rec(nl);
    for (int i = 0; i < nl; i++) {
      const pps_graph::Landmark& L = g->lms[i];
      const HostNode& nd = g->nodes[L.plane_id];
      rec[i].plane_slot = nd.deleted ? -1 : (state_current ? nd.slot : i);
      rec[i].frame_plane_indice = L.fpi; rec[i].frame_seq_id = L.seq; rec[i].deleted = L.deleted;
      memcpy(rec[i].seg2d, L.seg2d, sizeof L.seg2d); memcpy(rec[i].seg3d, L.seg3d, sizeof L.seg3d);
    }
    HIP_TRY(g, hipMemcpyAsync(g->d_lms, rec.data(), rec.size() * sizeof(AssocLandmark), hipMemcpyHostToDevice, g->stream));
    HIP_TRY(g, hipStreamSynchronize(g->stream));
    g->lms_dirty = false;
    g->lms_upload_version = state_current ? g->upload_version : -2;
  }
  if ((size_t)n > g->d_q_cap) {
    if (g->d_queries) (void)hipFree(g->d_queries);
    if (g->d_results) (void)hipFree(g->d_results);
    g->d_q_cap = std::max<size_t>(64, 2 * (size_t)n);
    HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_queries), g->d_q_cap * sizeof(AssocQuery)));
    HIP_TRY(g, hipMalloc(reinterpret_cast<void**>(&g->d_results), g->d_q_cap * sizeof(AssocResult)));
  }
  std::vector<AssocQuery> q(n);
  for (int i = 0; i < n; i++) {
    memcpy(q[i].plane_local, planes_local + 4 * i, 4 * sizeof(double));
    memcpy(q[i].seg2d, seg2d + 4 * i, 4 * sizeof(float)); memcpy(q[i].seg3d, seg3d_xy + 4 * i, 4 * sizeof(float));
    q[i].frame_plane_indice = frame_plane_indice[i]; q[i].frame_seq_id = frame_seq_id;
    memset(q[i].pad, 0, sizeof q[i].pad);
  }
  HIP_TRY(g, hipMemcpyAsync(g->d_queries, q.data(), q.size() * sizeof(AssocQuery), hipMemcpyHostToDevice, g->stream));
  a.n_queries = n; a.n_landmarks = nl; a.queries = g->d_queries; a.landmarks = g->d_lms; a.results = g->d_results;
  memcpy(a.pose, est_pose, sizeof a.pose);
  a.edge_asso_2ddist = P.edge_asso_2ddist; a.edge_asso_planedist = P.edge_asso_planedist;
  a.edge_asso_proj = P.edge_asso_proj; a.edge_asso_angle = P.edge_asso_angle; a.assoc_near_frames = P.assoc_near_frames;
  HIP_TRY(g, launch_assoc(a, g->stream));
  std::vector<AssocResult> r(n);
  HIP_TRY(g, hipMemcpyAsync(r.data(), g->d_results, r.size() * sizeof(AssocResult), hipMemcpyDeviceToHost, g->stream));
  HIP_TRY(g, hipStreamSynchronize(g->stream));
  for (int i = 0; i < n; i++) {
    best_plane_id[i] = r[i].best >= 0 ? g->lms[r[i].best].plane_id : -1;
    best_err[i] = r[i].err;
  }
  return PPS_OK;
}


// ---- graph text format: Slam::save (Slam.cpp:84-89) -> Graph::write (Graph.h:120-131) ----
//   factor line   <name> <node ids> <measure> {sqrtinf upper triangle, row-first}   (Factor.h:148-155,208-211,169-187)
//   node line     <Type>_Node <id> <value>                                          (Node.h:148-153)
//   Pose3d (x, y, z; yaw, pitch, roll) (Pose3d.h:169-172);  Plane3d (a, b, c; d) (isam_plane3d.h:190-192)
// The plane prior prints as "Pose3d_Factor" like the pose prior (constructor name, isam_plane3d.h:438).
// precision <= 0 selects the ostream default of the reference (6 significant digits, lossy); 17 round-trips.

int pps_graph_save(pps_graph* g, const char* path, int precision) {
  if (!g || !path) return PPS_EINVAL;
  if (g->dev_values_newer) { int rc = download_state(g); if (rc != PPS_OK) return rc; }
  if (g->dev_meas_newer) { int rc = download_measurements(g); if (rc != PPS_OK) return rc; }
  FILE* f = fopen(path, "wb");
  if (!f) return fail(g, PPS_EINVAL, std::string("graph_save: cannot open ") + path);
  const int prec = precision <= 0 ? 6 : precision;
  // std::to_chars / from_chars: the format must not follow LC_NUMERIC (a host that called setlocale() with a comma decimal
  // separator would otherwise write numbers that collide with the ", " and ";" field separators)
  auto num = [&](double v) {
    char b[64];
    const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, prec);
    fwrite(b, 1, (size_t)(r.ptr - b), f);
  };
  auto pose6 = [&](const double v6[6]) {
    fputc('(', f); num(v6[0]); fputs(", ", f); num(v6[1]); fputs(", ", f); num(v6[2]); fputs("; ", f);
    num(v6[3]); fputs(", ", f); num(v6[4]); fputs(", ", f); num(v6[5]); fputc(')', f);
  };
  auto plane4 = [&](const double v[4]) {
    fputc('(', f); num(v[0]); fputs(", ", f); num(v[1]); fputs(", ", f); num(v[2]); fputs("; ", f); num(v[3]); fputc(')', f);
  };
  auto noise = [&](const double* ut, int n) {
    fputs(" {", f);
    for (int k = 0; k < n; k++) { if (k) fputc(',', f); num(ut[k]); }
    fputc('}', f);
  };
  for (size_t i = 0; i < g->factors.size(); i++) {
    const HostFactor& F = g->factors[i];
    if (F.deleted) continue;
    switch (F.type) {
      case F_POSE_PRIOR: fprintf(f, "Pose3d_Factor %d ", F.a); pose6(F.meas); noise(F.w, 21); break;
      case F_ODOMETRY: fprintf(f, "Pose3d_Pose3d_Factor %d %d ", F.a, F.b); pose6(F.meas); noise(F.w, 21); break;
      case F_PLANE_OBS: fprintf(f, "Pose3d_Plane3d_Factor %d %d ", F.a, F.b); plane4(F.meas); noise(F.w, 6); break;
      default: fprintf(f, "Pose3d_Factor %d ", F.a); plane4(F.meas); noise(F.w, 6); break;
    }
    fputc('\n', f);
  }
  for (size_t i = 0; i < g->nodes.size(); i++) {
    const HostNode& N = g->nodes[i];
    if (N.deleted) continue;
    if (N.type == NODE_POSE) {
      double ypr[3];
      quat_to_euler(N.v + 3, ypr);
      const double v6[6] = {N.v[0], N.v[1], N.v[2], ypr[0], ypr[1], ypr[2]};
      fprintf(f, "Pose3d_Node %d ", (int)i); pose6(v6);
    } else {
      fprintf(f, "Plane3d_Node %d ", (int)i); plane4(N.v);
    }
    fputc('\n', f);
  }
  const bool ok = ferror(f) == 0;
  fclose(f);
  return ok ? PPS_OK : fail(g, PPS_EINVAL, "graph_save: write error");
}

// Reads a file written by pps_graph_save (the reference has no reader for this format: checkpoint / resume).
// Node ids are re-assigned densely in file order; factor ids follow file order.
int pps_graph_load(const char* path, const pps_props* props, pps_graph** out) {
  if (!path || !out) return PPS_EINVAL;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return PPS_EINVAL;
  struct Line { std::string name; std::vector<int> ids; std::vector<double> meas, ut; };
  std::vector<Line> nodes, factors;
  std::vector<char> buf(1 << 16);
  bool bad = false;
  while (fgets(buf.data(), (int)buf.size(), f)) {
    std::string s(buf.data());
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
    if (s.empty()) continue;
    Line L;
    const size_t po = s.find('('), pc = s.find(')');
    if (po == std::string::npos || pc == std::string::npos || pc < po) { bad = true; break; }
    {
      char name[64]; int off = 0;
      if (sscanf(s.c_str(), "%63s%n", name, &off) != 1) { bad = true; break; }
      L.name = name;
      const char* p = s.c_str() + off;
      const char* end = s.c_str() + po;
      while (p < end) { char* q; long v = strtol(p, &q, 10); if (q == p) break; L.ids.push_back((int)v); p = q; }
    }
    auto numbers = [](const std::string& t, std::vector<double>& o) {
      const char* p = t.c_str();
      const char* end = p + t.size();
      while (p < end) {
        if (*p == '+') { p++; continue; }                      // from_chars takes no leading plus
        double v = 0;
        const auto r = std::from_chars(p, end, v);
        if (r.ec != std::errc() || r.ptr == p) { p++; continue; }
        o.push_back(v); p = r.ptr;
      }
    };
    numbers(s.substr(po + 1, pc - po - 1), L.meas);
    const size_t bo = s.find('{', pc), bc = s.find('}', pc);
    if (bo != std::string::npos && bc != std::string::npos) numbers(s.substr(bo + 1, bc - bo - 1), L.ut);
    if (L.name.size() > 5 && L.name.compare(L.name.size() - 5, 5, "_Node") == 0) nodes.push_back(L); else factors.push_back(L);
  }
  fclose(f);
  if (bad) return PPS_EINVAL;
  pps_graph* g = nullptr;
  int rc = pps_graph_create(props, &g);
  if (rc != PPS_OK) return rc;
  std::unordered_map<int, int> id_of;
  for (const Line& L : nodes) {
    int id = -1;
    if (L.ids.size() != 1) { rc = PPS_EINVAL; break; }
    if (L.name == "Pose3d_Node" && L.meas.size() == 6) {
      double tq[7] = {L.meas[0], L.meas[1], L.meas[2]};
      euler_to_quat(L.meas[3], L.meas[4], L.meas[5], tq + 3);
      rc = pps_add_pose(g, tq, &id);
    } else if (L.name == "Plane3d_Node" && L.meas.size() == 4) {
      rc = pps_add_plane(g, L.meas.data(), &id);
    } else rc = PPS_EINVAL;
    if (rc != PPS_OK) break;
    id_of[L.ids[0]] = id;
  }
  auto nid = [&](int file_id) { auto it = id_of.find(file_id); return it == id_of.end() ? -1 : it->second; };
  if (rc == PPS_OK)
    for (const Line& L : factors) {
      int fid;
      if (L.name == "Pose3d_Pose3d_Factor" && L.ids.size() == 2 && L.meas.size() == 6 && L.ut.size() == 21)
        rc = pps_add_odometry(g, nid(L.ids[0]), nid(L.ids[1]), L.meas.data(), L.ut.data(), &fid);
      else if (L.name == "Pose3d_Plane3d_Factor" && L.ids.size() == 2 && L.meas.size() == 4 && L.ut.size() == 6)
        rc = pps_add_plane_obs(g, nid(L.ids[0]), nid(L.ids[1]), L.meas.data(), L.ut.data(), &fid);
      else if (L.name == "Pose3d_Factor" && L.ids.size() == 1 && L.meas.size() == 6 && L.ut.size() == 21)
        rc = pps_add_pose_prior(g, nid(L.ids[0]), L.meas.data(), L.ut.data(), &fid);
      else if (L.name == "Pose3d_Factor" && L.ids.size() == 1 && L.meas.size() == 4 && L.ut.size() == 6)
        rc = pps_add_plane_prior(g, nid(L.ids[0]), L.meas.data(), L.ut.data(), &fid);
      else rc = PPS_EINVAL;
      if (rc != PPS_OK) break;
    }
  if (rc != PPS_OK) { pps_graph_destroy(g); return rc; }
  *out = g;
  return PPS_OK;
}


// Mapper_mono::reproj_to_newplane (src/Mapping.cpp:609-632): polygon vertices onto the optimised planes
int pps_reproject_points(pps_graph* g, int n, const int* plane_ids, const float* pts_xyz, float* out_xyz) {
  if (!g || n < 0 || (n > 0 && (!plane_ids || !pts_xyz || !out_xyz))) return PPS_EINVAL;
  if (n == 0) return PPS_OK;
  int rc = prepare_solve(g);
  if (rc != PPS_OK) return rc;
  std::vector<int> slot(n);
  for (int i = 0; i < n; i++) slot[i] = live_node(g, plane_ids[i], NODE_PLANE) ? g->nodes[plane_ids[i]].slot : -1;
  int* d_slot = nullptr; float *d_in = nullptr, *d_out = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_slot), (size_t)n * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_in), (size_t)3 * n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)3 * n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpyAsync(d_slot, slot.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, g->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_in, pts_xyz, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, g->stream);
  if (e == hipSuccess) e = launch_reproject(n, d_slot, d_in, g->dev.plane_est, g->dev.plane_ld, d_out, g->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_xyz, d_out, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, g->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
  (void)hipFree(d_slot); (void)hipFree(d_in); (void)hipFree(d_out);
  if (e != hipSuccess) return hip_fail(g, e, "pps_reproject_points");
  return PPS_OK;
}

}  // extern "C"
