// Host half of the ground-edge selection (popup_plane::edge_get_polygons, pop_up_wall/libs/select_edge.cpp:66-409):
// contour linking and the sequential segment selection that follow the per-pixel kernels of pps_edges.hip.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/pps.h"

namespace pps_edges_host {

// one marching-squares segment (row, column of both ends); the kernels emit them in raster order of the 2x2 cells
struct CellSeg { int16_t fr, fc, tr, tc; };

struct Contour {
  std::vector<float> xy;     // every 20th point of [0, len-1) of the chosen contour as (x, y), scaled
  int n_contours = 0;        // contours found
  int n_points = 0;          // points of the chosen contour
};

// skimage.measure.find_contours linking + pop_up_fun.py:85-106 (longest first-to-last distance, sub-sampling)
Contour ground_contour(const CellSeg* segs, int n, float scale);

// pop_up_fun.py:109-204
std::vector<float> interval_tree_optimization(const std::vector<float>& lines, double overlap_thre);

struct Selection { std::vector<float> open_segs, closed_segs, open_in_closed; };

// select_edge.cpp:92-405
Selection select(const std::vector<float>& contour_xy, int width, int height, const float* lsd, int n_lsd,
                 const pps_edge_params& prm);

}  // namespace pps_edges_host
