// pps_io.cpp -- the reference's graph text format (Slam::save / Graph::write, iSAM logs).
#include "pps_graph.h"

using namespace pps;
using namespace pps_impl;

extern "C" {

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



}  // extern "C"
