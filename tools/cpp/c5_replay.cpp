// c5_replay.cpp -- BASELINE config 5 through the C++ facade, the host loop a maintainer of the reference would run:
// per frame pop the ground segments up at the predicted pose (popup_plane::get_plane_equation + generate_cloud),
// build the graph like Mapper_mono::processFrame (pop_planar_slam/src/Mapping.cpp:401-554), solve (batch every fifth
// frame, one step otherwise), re-derive all measurements from the new estimate (update_plane_measurement, :590-607).
// Input: the binary frame script written by tools/c5_bench_cpp.py (the same synthetic drive tools/c5_bench.py uses).
// Prints one JSON line: frames/s, final chi2, LM iterations.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "pps_isam.hpp"

using namespace isam;

namespace {
template <class T>
bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }
#define CK(expr) do { int _rc = (expr); if (_rc != PPS_OK) { fprintf(stderr, "%s failed: %d\n", #expr, _rc); return 1; } } while (0)
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s frames.bin [pixel_step]\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  const int step = argc > 2 ? atoi(argv[2]) : 2;
  int32_t hdr[3];                       // frames, width, height
  float invK[9];
  if (!rd(f, hdr, 3) || !rd(f, invK, 9)) return 2;
  const int n_frames = hdr[0], width = hdr[1], height = hdr[2];
  struct FrameIn { double odo[7]; std::vector<float> seg; std::vector<int32_t> ids; std::vector<double> dist; std::vector<float> polys; std::vector<int32_t> poly_off; };
  std::vector<FrameIn> frames(n_frames);
  for (auto& fr : frames) {
    int32_t n, nv;
    if (!rd(f, fr.odo, 7) || !rd(f, &n, 1)) return 2;
    fr.seg.resize(4 * n); fr.ids.resize(n); fr.dist.resize(n + 1); fr.poly_off.resize(n + 2);
    if ((n && (!rd(f, fr.seg.data(), 4 * n) || !rd(f, fr.ids.data(), n))) || !rd(f, fr.dist.data(), n + 1) || !rd(f, fr.poly_off.data(), n + 2)) return 2;
    nv = fr.poly_off[n + 1];
    fr.polys.resize(2 * nv);
    if (nv && !rd(f, fr.polys.data(), 2 * nv)) return 2;
  }
  fclose(f);
  try {
    Slam slam;
    Properties prop = slam.properties();             // Mapping.cpp:32-43
    prop.method = LEVENBERG_MARQUARDT; prop.mod_batch = 1; prop.quiet = true;
    prop.epsilon2 *= 0.1; prop.epsilon_abs *= 0.1; prop.epsilon_rel *= 0.1;   // :37-39
    slam.set_properties(prop);
    pps_graph* g = slam.handle();
    CK(pps_frames_set_calibration(g, invK));
    pps_popup* pp = nullptr;
    CK(pps_popup_create(0, width, height, invK, &pp));
    CK(pps_popup_set_outputs(pp, 0, 0));               // the frame loop consumes the planes and the cloud only
    std::vector<unsigned char> bgr((size_t)width * height * 3);
    for (size_t i = 0; i < bgr.size(); i++) bgr[i] = (unsigned char)(i * 2654435761u >> 24);
    CK(pps_popup_set_image(pp, bgr.data()));
    // sqrt-information 0.5 where the walls observe (lateral, forward, heading), 50 for height / tilt (pipeline.py POSE_UT)
    const double pose_var[6] = {4, 0.0004, 4, 0.0004, 4, 0.0004};
    const double ground_var[3] = {0.0025, 0.0025, 0.0025};   // sigma = 0.05, plane_3d_tum_far.yaml:23-25
    Covariance poseCov = Covariance::diagonal(pose_var, 6), groundCov = Covariance::diagonal(ground_var, 3);
    std::vector<Pose3d_Node*> all_frames;
    std::map<int, Plane3d_Node*> all_landmarks;     // key -1 = ground
    std::vector<float> planes;
    std::vector<int> fids;
    long lm_iterations = 0, lm_calls = 0, points = 0;
    // C5_TIMING=1: host wall time per call site, printed to stderr (where the frame's 0.5 ms goes)
    const bool timing = getenv("C5_TIMING") != nullptr;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](int k) { if (!timing) return; const auto t = std::chrono::steady_clock::now(); acc[k] += std::chrono::duration<double, std::micro>(t - tick).count(); tick = t; };
    const auto t0 = std::chrono::steady_clock::now();
    tick = t0;
    for (int k = 0; k < n_frames; k++) {
      const FrameIn& fr = frames[k];
      const int n = (int)fr.ids.size();
      const Pose3d temp_pose = Pose3d::from_tq(fr.odo);
      const Pose3d est = all_frames.empty() ? temp_pose : all_frames.back()->value().oplus(temp_pose);   // main_3d.cpp:366-385
      const Matrix4d T = est.wTo();
      float T32[16];
      for (int i = 0; i < 16; i++) T32[i] = (float)T[i];
      int n_valid = 0;
      lap(0);
      // (round 6) the run is not waited for: the graph construction needs the plane equations only, which the kernel's first workgroup publishes
      // at once; the previous frame's pixels are counted here, when their run has long finished
      if (k > 0) { CK(pps_popup_wait(pp, &n_valid)); points += n_valid; }
      CK(pps_popup_run_async(pp, fr.seg.data(), n, T32, fr.polys.data(), fr.poly_off.data(), n + 1, step, 10.0f, 2.5f));
      lap(1);
      planes.resize(4 * (size_t)(n + 1));
      CK(pps_popup_planes_wait(pp, planes.data()));
      lap(2);
      Pose3d_Node* poseNode = new Pose3d_Node();
      slam.add_node(poseNode);                                                   // Mapping.cpp:464-465
      poseNode->init(est);
      if (all_frames.empty()) slam.add_factor(new Pose3d_Factor(poseNode, est, poseCov));                   // :470-473
      else slam.add_factor(new Pose3d_Pose3d_Factor(all_frames.back(), poseNode, temp_pose, poseCov));      // :477-478
      all_frames.push_back(poseNode);
      fids.assign(n + 1, -1);
      for (int j = 0; j <= n; j++) {
        const int key = j == 0 ? -1 : fr.ids[j - 1];
        const Plane3d measure(Vector4d{{planes[4 * j], planes[4 * j + 1], planes[4 * j + 2], planes[4 * j + 3]}});   // normalises
        Plane3d_Node*& node = all_landmarks[key];
        if (!node) {
          node = new Plane3d_Node();
          slam.add_node(node);                                                   // :482-490
          node->init(measure.transform_from(est.oTw()));                         // :496-499
          if (key == -1) slam.add_factor(new Plane3d_Factor(node, Plane3d(Vector4d{{0, 0, -1, 0}}), groundCov));   // :500-504
        }
        double d = fr.dist[j]; d = d < 3 ? 3 : d; d = d > 8 ? 8 : d;             // :507-509
        const double s = (d - 1) * 2.0 + 5;
        const double var[3] = {s * s, s * s, s * s};
        Pose3d_Plane3d_Factor* fac = new Pose3d_Plane3d_Factor(poseNode, node, measure, Covariance::diagonal(var, 3), false);   // :513,523
        slam.add_factor(fac);
        fids[j] = fac->backend_id();
      }
      lap(3);
      if (k % 5 == 0) { lm_iterations += slam.batch_optimization(); lm_calls++; lap(4); }   // :551-554
      else { slam.update(); lap(5); }
      int frame_id = 0;
      CK(pps_frames_add(g, poseNode->backend_id(), n, fr.seg.data(), fids.data(), &frame_id));
      CK(pps_refresh_measurements(g));                                           // main_3d.cpp:504 -> Mapping.cpp:590-607
      lap(6);
    }
    { int n_valid = 0; CK(pps_popup_wait(pp, &n_valid)); points += n_valid; }
    const double chi2 = slam.chi2();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"frames\": %d, \"frames_per_sec\": %.3f, \"wall_s\": %.6f, \"pixel_step\": %d, \"final_chi2\": %.17g, \"lm_calls\": %ld, "
           "\"lm_iterations\": %ld, \"points_per_frame\": %.1f, \"nodes\": %d, \"factors\": %d, \"host\": \"C++ facade (include/pps_isam.hpp)\"}\n",
           n_frames, n_frames / wall, wall, step, chi2, lm_calls, lm_iterations, (double)points / n_frames, slam.num_nodes(), slam.num_factors());
    if (timing)
      fprintf(stderr, "[c5 timing] us per frame: pose prediction %.1f | popup_run %.1f | popup_download %.1f | graph construction %.1f | "
              "batch_optimization %.1f (per frame; 1 in 5) | update %.1f (per frame; 4 in 5) | frames_add + refresh %.1f\n",
              acc[0] / n_frames, acc[1] / n_frames, acc[2] / n_frames, acc[3] / n_frames, acc[4] / n_frames, acc[5] / n_frames, acc[6] / n_frames);
    pps_popup_destroy(pp);
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
