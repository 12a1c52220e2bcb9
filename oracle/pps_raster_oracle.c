/*
 * pps_raster_oracle.c -- CPU ORACLE (test infrastructure, NOT product code): which pixels a pop-up polygon covers.
 *
 * Restates popup_plane::closed_polygons_homo_pts (/root/reference/pop_up_wall/libs/popup_plane.cpp:81-116):
 *   [polygon / 2 when downsample_poly]  -> matrix_to_points: cv::Point(float, float), i.e. C++ float -> int
 *   truncation (libs/matrix_utils.cpp:48-53) -> cv::boundingRect -> polygon - (box.x, box.y) in fp32, truncated again
 *   -> cv::fillConvexPoly into a box-sized CV_8U image -> cv::findNonZero -> (x + box.x, y + box.y) [* 2].
 *
 * cv::fillConvexPoly / cv::boundingRect / cv::clipLine / cv::LineIterator live in OpenCV (a system dependency of the
 * reference, `find_package(OpenCV)`, absent from /root/reference and from this image).  Their algorithm is restated
 * here from OpenCV's published source (modules/core|imgproc/src/drawing.cpp, 2.4.x / 3.x: FillConvexPoly, Line,
 * LineIterator::LineIterator(left_to_right = true), clipLine; XY_SHIFT = 16), statement by statement:
 *   - the outline of the polygon is drawn first with 8-connected Bresenham lines (LineIterator, connectivity 8);
 *   - the interior is filled scanline by scanline from two edge chains that start at the top-most vertex, x kept in
 *     16.16 fixed point, dx = ((xe - xs) * 2 + (ye - y)) / (2 * (ye - y)) (C division), both span ends rounded with
 *     + XY_ONE / 2 (line_type 8), spans clipped to the image.
 * PARITY UNPINNED against OpenCV itself (not installable here); pinned by hand-worked polygons and by an independent
 * closed-form numpy formulation (oracle/numpy_raster.py -> tests/golden/raster_cases.json).
 * Domain: finite vertices with |coordinate| < 2^15 (OpenCV 2.4 keeps x << 16 in a 32-bit int).
 */
#include <stdint.h>
#include <stdlib.h>

#include "pps_oracle.h"

typedef struct { int x, y; } ipt;
typedef void (*plot_fn)(void* ctx, int x, int y);

/* cv::clipLine(Size, Point&, Point&) */
static int clip_line(int width, int height, ipt* p1, ipt* p2) {
  int64_t x1, y1, x2, y2;
  int c1, c2;
  const int64_t right = width - 1, bottom = height - 1;
  if (width <= 0 || height <= 0) return 0;
  x1 = p1->x; y1 = p1->y; x2 = p2->x; y2 = p2->y;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (a - y1) * (x2 - x1) / (y2 - y1);
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (a - y2) * (x2 - x1) / (y2 - y1);
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (a - x1) * (y2 - y1) / (x2 - x1);
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (a - x2) * (y2 - y1) / (x2 - x1);
        x2 = a;
        c2 = 0;
      }
    }
    p1->x = (int)x1; p1->y = (int)y1; p2->x = (int)x2; p2->y = (int)y2;
  }
  return (c1 | c2) == 0;
}

/* Line(img, pt1, pt2, color, 8): LineIterator(img, pt1, pt2, 8, left_to_right = true), every position plotted.
 * The iterator's pointer steps (bt_pix = one column, istep = one row) are kept as coordinate steps. */
static void line8(int width, int height, ipt pt1, ipt pt2, plot_fn plot, void* ctx) {
  if ((unsigned)pt1.x >= (unsigned)width || (unsigned)pt2.x >= (unsigned)width ||
      (unsigned)pt1.y >= (unsigned)height || (unsigned)pt2.y >= (unsigned)height) {
    if (!clip_line(width, height, &pt1, &pt2)) return;
  }
  int dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
  int s = dx < 0 ? -1 : 0;
  dx = (dx ^ s) - s;
  dy = (dy ^ s) - s;
  pt1.x ^= (pt1.x ^ pt2.x) & s;
  pt1.y ^= (pt1.y ^ pt2.y) & s;
  int col_step = 1, row_step = 1;               /* bt_pix, istep */
  s = dy < 0 ? -1 : 0;
  dy = (dy ^ s) - s;
  row_step = (row_step ^ s) - s;
  /* conditional swap: the major axis becomes "dx" */
  int major_dx = col_step, major_dy = 0, minor_dx = 0, minor_dy = row_step;
  if (dy > dx) {
    int t = dx; dx = dy; dy = t;
    major_dx = 0; major_dy = row_step; minor_dx = col_step; minor_dy = 0;
  }
  int err = dx - (dy + dy);
  const int plus_delta = dx + dx, minus_delta = -(dy + dy);
  const int count = dx + 1;
  int x = pt1.x, y = pt1.y;
  for (int i = 0; i < count; i++) {
    plot(ctx, x, y);
    const int mask = err < 0 ? -1 : 0;
    err += minus_delta + (plus_delta & mask);
    x += major_dx + (minor_dx & mask);
    y += major_dy + (minor_dy & mask);
  }
}

/* FillConvexPoly(img, v, npts, color, line_type = 8, shift = 0) on a width x height image */
static void fill_convex_poly(int width, int height, const ipt* v, int npts, plot_fn plot, void* ctx) {
  enum { XY_SHIFT = 16, XY_ONE = 1 << XY_SHIFT };
  struct { int idx, di; int64_t x, dx; int ye; } edge[2];
  int i, y, imin = 0, left = 0, right = 1;
  int edges = npts;
  int xmin, xmax, ymin, ymax;
  const int64_t delta1 = XY_ONE >> 1, delta2 = XY_ONE >> 1;
  ipt p0 = v[npts - 1];
  xmin = xmax = v[0].x;
  ymin = ymax = v[0].y;
  for (i = 0; i < npts; i++) {
    ipt p = v[i];
    if (p.y < ymin) { ymin = p.y; imin = i; }
    if (p.y > ymax) ymax = p.y;
    if (p.x > xmax) xmax = p.x;
    if (p.x < xmin) xmin = p.x;
    line8(width, height, p0, p, plot, ctx);
    p0 = p;
  }
  if (npts < 3 || xmax < 0 || ymax < 0 || xmin >= width || ymin >= height) return;
  if (ymax > height - 1) ymax = height - 1;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y = ymin;
  edge[0].di = 1;
  edge[1].di = npts - 1;
  edge[0].x = edge[1].x = 0; edge[0].dx = edge[1].dx = 0;
  do {
    for (i = 0; i < 2; i++) {
      if (y >= edge[i].ye) {
        int idx = edge[i].idx, di = edge[i].di;
        int64_t xs = 0, xe;
        int ye, ty = 0;
        for (;;) {
          ty = v[idx].y;
          if (ty > y || edges == 0) break;
          xs = v[idx].x;
          idx += di;
          idx -= ((idx < npts) - 1) & npts;
          edges--;
        }
        ye = ty;
        xs <<= XY_SHIFT;
        xe = (int64_t)v[idx].x << XY_SHIFT;
        if (y >= ye) return;                      /* no more edges */
        edge[i].ye = ye;
        edge[i].dx = ((xe - xs) * 2 + (ye - y)) / (2 * (ye - y));
        edge[i].x = xs;
        edge[i].idx = idx;
      }
    }
    if (edge[left].x > edge[right].x) { left ^= 1; right ^= 1; }
    int64_t x1 = edge[left].x, x2 = edge[right].x;
    if (y >= 0) {
      int xx1 = (int)((x1 + delta1) >> XY_SHIFT);
      int xx2 = (int)((x2 + delta2) >> XY_SHIFT);
      if (xx2 >= 0 && xx1 < width) {
        if (xx1 < 0) xx1 = 0;
        if (xx2 >= width) xx2 = width - 1;
        for (int x = xx1; x <= xx2; x++) plot(ctx, x, y);     /* ICV_HLINE */
      }
    }
    x1 += edge[left].dx;
    x2 += edge[right].dx;
    edge[left].x = x1;
    edge[right].x = x2;
  } while (++y <= ymax);
}

typedef struct { int* plane_id; int width, height, box_x, box_y, box_w, box_h, scale, id; } mask_ctx;

static void plot_mask(void* c, int x, int y) {
  mask_ctx* m = (mask_ctx*)c;
  if (x < 0 || y < 0 || x >= m->box_w || y >= m->box_h) return;          /* outside the box-sized cv::Mat */
  const int X = (x + m->box_x) * m->scale, Y = (y + m->box_y) * m->scale;  /* popup_plane.cpp:104-113 */
  if (X < 0 || Y < 0 || X >= m->width || Y >= m->height) return;         /* the reference would index out of bounds */
  m->plane_id[(size_t)Y * m->width + X] = m->id;
}

/* One polygon: closed_polygons_homo_pts (popup_plane.cpp:81-116); pixels of the set get `id`. */
static void polygon_pixels(const float* xy, int npts, int step, mask_ctx* m) {
  if (npts < 1) return;
  float* P = (float*)malloc(sizeof(float) * 2 * (size_t)npts);
  ipt* q = (ipt*)malloc(sizeof(ipt) * (size_t)npts);
  for (int i = 0; i < 2 * npts; i++) P[i] = step == 2 ? xy[i] / 2 : xy[i];   /* new_polys_close / 2 (:86-87) */
  /* boundingRect of the truncated points (:90-91) */
  int xmin = (int)P[0], xmax = xmin, ymin = (int)P[1], ymax = ymin;
  for (int i = 0; i < npts; i++) {
    const int x = (int)P[2 * i], y = (int)P[2 * i + 1];
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
  m->box_x = xmin; m->box_y = ymin; m->box_w = xmax - xmin + 1; m->box_h = ymax - ymin + 1;
  /* polys_close_shift = polygon - (box.x, box.y) in fp32, then truncated (:92-96) */
  for (int i = 0; i < npts; i++) {
    const float sx = P[2 * i] - (float)xmin, sy = P[2 * i + 1] - (float)ymin;
    q[i].x = (int)sx; q[i].y = (int)sy;
  }
  m->scale = step == 2 ? 2 : 1;
  fill_convex_poly(m->box_w, m->box_h, q, npts, plot_mask, m);
  free(P); free(q);
}

/* Plane-id map of a frame: polygons in plane order, a later plane overwrites an earlier one (generate_cloud and
 * get_depth_map_good walk good_plane_indices in order, popup_plane.cpp:820-850, 893-911); -1 = no plane.
 * polys: (x, y) pairs, poly_off[nplanes + 1] vertex offsets; step 2 = downsample_poly. */
void ora_popup_mask(const float* polys, const int* poly_off, int nplanes, int width, int height, int step, int* plane_id) {
  for (size_t i = 0; i < (size_t)width * height; i++) plane_id[i] = -1;
  mask_ctx m;
  m.plane_id = plane_id; m.width = width; m.height = height;
  for (int p = 0; p < nplanes; p++) {
    m.id = p;
    polygon_pixels(polys + 2 * (size_t)poly_off[p], poly_off[p + 1] - poly_off[p], step, &m);
  }
}

/* cv::fillConvexPoly alone on integer points (for the hand-worked cases): img is width x height, 0 / 255 */
void ora_fill_convex_poly(const int* pts_xy, int npts, int width, int height, unsigned char* img) {
  int* ids = (int*)malloc(sizeof(int) * (size_t)width * height);
  for (size_t i = 0; i < (size_t)width * height; i++) ids[i] = -1;
  mask_ctx m;
  m.plane_id = ids; m.width = width; m.height = height; m.box_x = m.box_y = 0; m.box_w = width; m.box_h = height;
  m.scale = 1; m.id = 0;
  ipt* q = (ipt*)malloc(sizeof(ipt) * (size_t)(npts > 0 ? npts : 1));
  for (int i = 0; i < npts; i++) { q[i].x = pts_xy[2 * i]; q[i].y = pts_xy[2 * i + 1]; }
  if (npts > 0) fill_convex_poly(width, height, q, npts, plot_mask, &m);
  for (size_t i = 0; i < (size_t)width * height; i++) img[i] = ids[i] == 0 ? 255 : 0;
  free(q); free(ids);
}
