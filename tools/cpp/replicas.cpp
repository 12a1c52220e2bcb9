// replicas.cpp -- the in-process replica driver over the C-ABI (include/pps.h), C++ twin of tools/replicas.py.
//   SURVEY 8(e) / Mapping.cpp:400 (one optimisation thread per graph): ONE process, one host thread per entry of the device list,
//   each with its own handles on its device -- the graphs of the given files solved one after the other through their own
//   pps_graph handles, then all of them as one pps_multi batch -- and nothing shared between the threads.
// usage: replicas DEV[,DEV...] REPS graph0.txt [graph1.txt ...]     (graph files: pps_graph_save / Slam::save format)
// prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pps.h"

struct Result { int device; std::vector<double> chi2_single, chi2_multi; std::vector<int> it_single, it_multi; double gps_single = 0, gps_multi = 0; std::string err; };

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void worker(int dev, int reps, const std::vector<std::string>& files, Result* r) {
  r->device = dev;
  pps_props p; pps_default_props(&p); p.device = dev;
  std::vector<pps_graph*> gs;
  for (const std::string& f : files) {
    pps_graph* g = nullptr;
    if (pps_graph_load(f.c_str(), &p, &g) != PPS_OK) { r->err = "pps_graph_load " + f; return; }
    gs.push_back(g);
  }
  const int n = (int)gs.size();
  for (pps_graph* g : gs) {                                     // analysis + upload + first solve: outside the clock
    int it = 0; double c = 0;
    if (pps_save_state(g) != PPS_OK || pps_batch_optimize(g, &it) != PPS_OK || pps_chi2(g, &c) != PPS_OK) { r->err = pps_last_error(g); return; }
    r->it_single.push_back(it); r->chi2_single.push_back(c);
  }
  double t0 = now();
  for (int k = 0; k < reps; k++)
    for (pps_graph* g : gs) { int it; if (pps_restore_state(g) != PPS_OK || pps_batch_optimize(g, &it) != PPS_OK) { r->err = pps_last_error(g); return; } }
  r->gps_single = n * reps / (now() - t0);
  pps_multi* m = nullptr;
  for (pps_graph* g : gs) pps_restore_state(g);
  if (pps_multi_create(n, gs.data(), &m) != PPS_OK) { r->err = "pps_multi_create"; return; }
  r->it_multi.assign(n, 0);
  std::vector<int> st(n, 0);
  if (pps_multi_optimize(m, r->it_multi.data(), st.data()) != PPS_OK) { r->err = pps_multi_last_error(m); return; }
  for (pps_graph* g : gs) { double c = 0; pps_chi2(g, &c); r->chi2_multi.push_back(c); }
  t0 = now();
  for (int k = 0; k < reps; k++) {
    for (pps_graph* g : gs) pps_restore_state(g);
    if (pps_multi_optimize(m, nullptr, nullptr) != PPS_OK) { r->err = pps_multi_last_error(m); return; }
  }
  r->gps_multi = n * reps / (now() - t0);
  pps_multi_destroy(m);
  for (pps_graph* g : gs) pps_graph_destroy(g);
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s DEV[,DEV...] REPS graph.txt [...]\n", argv[0]); return 2; }
  std::vector<int> devs;
  for (char* tok = strtok(argv[1], ","); tok; tok = strtok(nullptr, ",")) devs.push_back(atoi(tok));
  const int reps = atoi(argv[2]);
  std::vector<std::string> files(argv + 3, argv + argc);
  std::vector<Result> res(devs.size());
  std::vector<std::thread> th;
  const double t0 = now();
  for (size_t k = 0; k < devs.size(); k++) th.emplace_back(worker, devs[k], reps, std::cref(files), &res[k]);
  for (std::thread& t : th) t.join();
  const double wall = now() - t0;
  for (const Result& r : res) if (!r.err.empty()) { fprintf(stderr, "device %d: %s\n", r.device, r.err.c_str()); return 1; }
  printf("{\"devices\": [");
  for (size_t k = 0; k < devs.size(); k++) printf("%s%d", k ? ", " : "", devs[k]);
  printf("], \"graphs_per_device\": %zu, \"reps\": %d, \"wall_s\": %.6f, \"per_device\": [", files.size(), reps, wall);
  for (size_t k = 0; k < res.size(); k++) {
    const Result& r = res[k];
    printf("%s{\"device\": %d, \"single_graphs_per_sec\": %.3f, \"multi_graphs_per_sec\": %.3f, \"chi2_single\": [", k ? ", " : "", r.device, r.gps_single, r.gps_multi);
    for (size_t i = 0; i < r.chi2_single.size(); i++) printf("%s%.17g", i ? ", " : "", r.chi2_single[i]);
    printf("], \"chi2_multi\": [");
    for (size_t i = 0; i < r.chi2_multi.size(); i++) printf("%s%.17g", i ? ", " : "", r.chi2_multi[i]);
    printf("], \"iterations\": [");
    for (size_t i = 0; i < r.it_single.size(); i++) printf("%s%d", i ? ", " : "", r.it_single[i]);
    printf("], \"iterations_multi\": [");
    for (size_t i = 0; i < r.it_multi.size(); i++) printf("%s%d", i ? ", " : "", r.it_multi[i]);
    printf("]}");
  }
  printf("]}\n");
  return 0;
}
