// pps_raster.h -- which pixels a pop-up polygon covers, on the device.
//
// The reference rasterises every plane polygon with OpenCV (popup_plane::closed_polygons_homo_pts,
// /root/reference/pop_up_wall/libs/popup_plane.cpp:81-116): vertices [halved when downsample_poly] are truncated to
// integers (cv::Point(float, float)), shifted into their bounding box, cv::fillConvexPoly draws the 8-connected
// Bresenham outline and fills scanline spans from two fixed-point edge chains, cv::findNonZero lists the pixels.
// Here the same pixel set is produced without a sequential raster pass.  For one row of the box image the set is a
// union of intervals, and every interval has a closed form:
//   - a Bresenham edge, after j steps along its major axis, has moved m(j) = floor((2 dmin j + dmaj - 1) / (2 dmaj))
//     along the minor axis (LineIterator's error test unrolled); an x-major edge covers one run of columns per row, a
//     y-major edge one column;
//   - the fill span of a row comes from replaying FillConvexPoly's edge-chain events (at most one per vertex) and
//     jumping over the rows between two events: x advances linearly there (x += dx per row).
// A workgroup derives the intervals of ITS rows into LDS (a few dozen integer operations per edge), after which a
// pixel is classified by comparing its column with a handful of intervals.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pps {

// one outline edge after cv::clipLine and LineIterator's left-to-right normalisation, in box coordinates
struct RasterLine {
  int x0, y0;      // first pixel
  int dmaj, dmin;  // steps along the major / minor axis
  int flags;       // bit 0: y is the major axis; bit 1: the row coordinate decreases along the line; bit 2: nothing to draw
};

// cv::clipLine(Size(w, h), p1, p2)
__host__ __device__ inline bool raster_clip_line(int w, int h, int& px1, int& py1, int& px2, int& py2) {
  long long x1 = px1, y1 = py1, x2 = px2, y2 = py2;
  const long long right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (a - y1) * (x2 - x1) / (y2 - y1); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
    if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (a - y2) * (x2 - x1) / (y2 - y1); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) { a = c1 == 1 ? 0 : right; y1 += (a - x1) * (y2 - y1) / (x2 - x1); x1 = a; c1 = 0; }
      if (c2) { a = c2 == 1 ? 0 : right; y2 += (a - x2) * (y2 - y1) / (x2 - x1); x2 = a; c2 = 0; }
    }
    px1 = (int)x1; py1 = (int)y1; px2 = (int)x2; py2 = (int)y2;
  }
  return (c1 | c2) == 0;
}

// Line(img, p1, p2, 8) -> LineIterator(img, p1, p2, 8, left_to_right = true) as a record
__host__ __device__ inline RasterLine raster_line(int w, int h, int x1, int y1, int x2, int y2) {
  RasterLine L{0, 0, 0, 0, 4};
  if ((unsigned)x1 >= (unsigned)w || (unsigned)x2 >= (unsigned)w || (unsigned)y1 >= (unsigned)h || (unsigned)y2 >= (unsigned)h) {
    if (!raster_clip_line(w, h, x1, y1, x2, y2)) return L;
  }
  int dx = x2 - x1, dy = y2 - y1;
  if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
  int fl = 0;
  if (dy < 0) { dy = -dy; fl |= 2; }
  if (dy > dx) { const int t = dx; dx = dy; dy = t; fl |= 1; }
  L.x0 = x1; L.y0 = y1; L.dmaj = dx; L.dmin = dy; L.flags = fl;
  return L;
}

// columns [lo, hi] of row cy the line covers (box coordinates); false: none
__host__ __device__ inline bool raster_line_row(const RasterLine& L, int cy, int& lo, int& hi) {
  if (L.flags & 4) return false;
  const int sgn = (L.flags & 2) ? -1 : 1;
  const int t = (cy - L.y0) * sgn;                 // steps down the rows from the first pixel
  if (t < 0) return false;
  const unsigned dmaj = (unsigned)L.dmaj, dmin = (unsigned)L.dmin;
  if (L.flags & 1) {                               // y-major: one pixel per row
    if (t > L.dmaj) return false;
    const unsigned m = (2u * dmin * (unsigned)t + dmaj - 1u) / (2u * dmaj);     // dmaj > dmin >= 0
    lo = hi = L.x0 + (int)m;
    return true;
  }
  if (t > L.dmin) return false;
  if (dmin == 0) { lo = L.x0; hi = L.x0 + L.dmaj; return true; }
  // m(j) == t  <=>  2 dmaj t <= 2 dmin j + dmaj - 1 < 2 dmaj (t + 1)
  const unsigned D = 2u * dmin;
  const long long n1 = 2ll * dmaj * t - dmaj + 1;
  const unsigned n2 = 2u * dmaj * (unsigned)(t + 1) - dmaj + 1u;
  const int jlo = n1 <= 0 ? 0 : (int)(((unsigned)n1 + D - 1u) / D);
  int jhi = (int)((n2 + D - 1u) / D) - 1;
  if (jhi > L.dmaj) jhi = L.dmaj;
  if (jlo > jhi) return false;
  lo = L.x0 + jlo; hi = L.x0 + jhi;
  return true;
}

// trunc(num / den) for |num| < 2^40, 0 < den < 2^20: double quotient, corrected with the exact remainder
__host__ __device__ inline long long raster_cdiv(long long num, long long den) {
  long long q = (long long)((double)num / (double)den);       // conversion truncates toward zero
  long long r = num - q * den;
  // bring the remainder to the sign of num and |r| < den
  if (num >= 0) { while (r < 0) { q--; r += den; } while (r >= den) { q++; r -= den; } }
  else { while (r > 0) { q++; r -= den; } while (r <= -den) { q--; r += den; } }
  return q;
}

// Scanline span of row cy of FillConvexPoly(img(w x h), q, npts, line_type 8, shift 0): false = the fill draws nothing there.
// q: integer vertices (x, y) in box coordinates.
__host__ __device__ inline bool raster_fill_row(const int2* __restrict__ q, int npts, int w, int h, int cy, int& xx1, int& xx2) {
  if (npts < 3) return false;
  int xmin = q[0].x, xmax = xmin, ymin = q[0].y, ymax = ymin, imin = 0;
  for (int i = 0; i < npts; i++) {
    const int2 p = q[i];
    if (p.y < ymin) { ymin = p.y; imin = i; }
    if (p.y > ymax) ymax = p.y;
    if (p.x > xmax) xmax = p.x;
    if (p.x < xmin) xmin = p.x;
  }
  if (xmax < 0 || ymax < 0 || xmin >= w || ymin >= h) return false;
  if (ymax > h - 1) ymax = h - 1;
  if (cy < ymin || cy > ymax || cy < 0) return false;
  int e_idx[2] = {imin, imin}, e_ye[2] = {ymin, ymin};
  const int e_di[2] = {1, npts - 1};
  long long e_x[2] = {0, 0}, e_dx[2] = {0, 0};
  int edges = npts, y = ymin;
  for (;;) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (y >= e_ye[i]) {
        int idx = e_idx[i];
        long long xs = 0;
        int ty = 0;
        for (;;) {
          ty = q[idx].y;
          if (ty > y || edges == 0) break;
          xs = q[idx].x;
          idx += e_di[i];
          if (idx >= npts) idx -= npts;
          edges--;
        }
        if (y >= ty) return false;                 // no more edges
        xs <<= 16;
        const long long xe = (long long)q[idx].x << 16;
        e_ye[i] = ty;
        e_dx[i] = raster_cdiv((xe - xs) * 2 + (ty - y), 2ll * (ty - y));
        e_x[i] = xs;
        e_idx[i] = idx;
      }
    }
    if (y == cy) break;
    int yn = e_ye[0] < e_ye[1] ? e_ye[0] : e_ye[1];   // next row where an edge chain changes
    if (yn > cy) yn = cy;
    e_x[0] += e_dx[0] * (yn - y);
    e_x[1] += e_dx[1] * (yn - y);
    y = yn;
  }
  const long long x1 = e_x[0] < e_x[1] ? e_x[0] : e_x[1], x2 = e_x[0] < e_x[1] ? e_x[1] : e_x[0];
  int a = (int)((x1 + 32768) >> 16), b = (int)((x2 + 32768) >> 16);
  if (!(b >= 0 && a < w)) return false;
  if (a < 0) a = 0;
  if (b >= w) b = w - 1;
  xx1 = a; xx2 = b;
  return true;
}

}  // namespace pps
