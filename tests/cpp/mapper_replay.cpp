// mapper_replay.cpp -- drives the C++ facade (include/pps_isam.hpp) exactly the way the reference's
// Mapper_mono::processFrame builds and solves its graph (pop_planar_slam/src/Mapping.cpp:31-43,401-554),
// from a text script of per-frame odometry and plane measurements.  Prints chi2 per frame and the
// final poses; tests/test_gpu_facade.py replays the same script through the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

#include "pps_isam.hpp"

using namespace isam;

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s script.txt [analytic]\n", argv[0]); return 2; }
  std::ifstream in(argv[1]);
  if (!in) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  try {
    Slam _slam;
    Properties prop = _slam.properties();          // Mapping.cpp:32-43
    prop.method = LEVENBERG_MARQUARDT;
    prop.mod_batch = 1;
    prop.quiet = true;
    prop.epsilon2 *= 0.1; prop.epsilon_abs *= 0.1; prop.epsilon_rel *= 0.1;
    if (argc > 2) prop.jacobian_mode = PPS_JAC_ANALYTIC;
    _slam.set_properties(prop);
    const double pose_var[6] = {4, 4, 4, 4, 4, 4};               // sigma = 2, plane_3d_tum_far.yaml:16-21
    const double plane_var[3] = {0.0025, 0.0025, 0.0025};         // sigma = 0.05, yaml:23-25
    Covariance poseCov = Covariance::diagonal(pose_var, 6);
    Covariance planeCov = Covariance::diagonal(plane_var, 3);
    const double plane_sigma_dist_mul = 2.0;                      // yaml:27

    std::vector<Pose3d_Node*> all_frames;
    std::map<int, Plane3d_Node*> all_landmarks;
    std::string tok;
    int nframes = 0;
    float inv_calib[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    in >> tok;
    if (tok == "INVK") { for (float& v : inv_calib) in >> v; in >> tok; }      // optional: K^-1 for OBS2 lines
    in >> nframes;
    for (int k = 0; k < nframes; k++) {
      double tq[7]; int fk, nobs;
      in >> tok >> fk;
      for (double& v : tq) in >> v;
      in >> nobs;
      Pose3d temp_pose = Pose3d::from_tq(tq);
      Pose3d estimate_pose;                                       // identity on the first frame (Mapping.cpp:413)
      if (!all_frames.empty()) estimate_pose = all_frames.back()->value().oplus(temp_pose);   // :414-416
      Pose3d_Node* poseNode = new Pose3d_Node();
      _slam.add_node(poseNode);                                   // :464-465
      if (all_frames.empty()) {
        _slam.add_factor(new Pose3d_Factor(poseNode, temp_pose, poseCov));   // :470-473
      } else {
        poseNode->init(estimate_pose);                            // :475
        _slam.add_factor(new Pose3d_Pose3d_Factor(all_frames.back(), poseNode, temp_pose, poseCov));   // :477-478
      }
      all_frames.push_back(poseNode);
      struct Obs { int key, ground; double dist; Vector4d m; bool repop; float seg[4]; };
      std::vector<Obs> obs(nobs);
      std::vector<int> fresh;
      for (auto& o : obs) {
        in >> tok >> o.key >> o.ground >> o.dist;
        for (double& v : o.m) in >> v;
        o.repop = tok == "OBS2";                                  // the observation re-pops its measurement from its ground edge
        if (o.repop) for (float& v : o.seg) in >> v;
        if (!all_landmarks.count(o.key)) {                        // new plane node (:482-490)
          Plane3d_Node* planeNode = new Plane3d_Node();
          _slam.add_node(planeNode);
          all_landmarks[o.key] = planeNode;
          fresh.push_back(o.key);
        }
      }
      for (auto& o : obs) {                                       // :493-530
        Plane3d_Node* planeNode = all_landmarks[o.key];
        Plane3d measure(o.m);
        bool is_new = false;
        for (int f : fresh) if (f == o.key) is_new = true;
        if (is_new && !planeNode->initialized()) {
          planeNode->init(measure.transform_from(estimate_pose.oTw()));     // :496-499
          if (o.ground) _slam.add_factor(new Plane3d_Factor(planeNode, Plane3d(Vector4d{{0, 0, -1, 0}}), planeCov));   // :500-504
        }
        double d = o.dist; d = d < 3 ? 3 : d; d = d > 8 ? 8 : d;  // :507-509
        const double s = (d - 1) * plane_sigma_dist_mul + 5;
        const double var[3] = {s * s, s * s, s * s};
        Covariance planeCov_new = Covariance::diagonal(var, 3);
        if (o.repop) {                                            // the variant of :515-521: pop the plane up from its line in every evaluation
          Pose3d_Plane3d_Factor2* fac = new Pose3d_Plane3d_Factor2(poseNode, planeNode, measure, planeCov_new, false);
          fac->precompute_edge_ray(inv_calib, o.seg);
          _slam.add_factor(fac);
        } else {
          _slam.add_factor(new Pose3d_Plane3d_Factor(poseNode, planeNode, measure, planeCov_new, false));   // :513,523
        }
      }
      int iterations = -1;
      if (k % 5 == 0) iterations = _slam.batch_optimization();    // :551-554
      else _slam.update();
      printf("FRAME %d chi2 %.17g iterations %d nodes %d factors %d\n", k, _slam.chi2(), iterations, _slam.num_nodes(), _slam.num_factors());
    }
    for (size_t k = 0; k < all_frames.size(); k++) {
      double tq[7];
      all_frames[k]->value().to_tq(tq);
      printf("POSE %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, tq[0], tq[1], tq[2], tq[3], tq[4], tq[5], tq[6]);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
