/*
 * pps_edges_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's ground-edge selection,
 * popup_plane::edge_get_polygons (pop_up_wall/libs/select_edge.cpp:66-409),
 * including the two Python helpers it calls through boost::python
 * (pop_up_python/python/pop_up_python/pop_up_fun.py:85-204) and the library
 * routines those lean on.  The libraries are NOT in /root/reference; their
 * published behaviour is restated here:
 *   - OpenCV resize(INTER_NEAREST), dilate, erode with a full rectangular element and the default anchor /
 *     border (out-of-image pixels are ignored), convertTo(alpha=-1, beta=255);
 *   - scikit-image measure.find_contours(array, 0) (marching squares, 'low' connectivity, segments linked in
 *     raster order; 2016-era _find_contours.py/_find_contours_cy.pyx);
 *   - intervaltree 2.x: half-open intervals, search(begin, end), split_overlaps(), Interval ordering
 *     (begin, end, data).
 * PARITY UNPINNED: the reference has no tests or golden vectors for this stage and none of those libraries are in
 * this image.  Cross-checks: oracle/numpy_edges.py, an independent numpy / scipy formulation of the whole stage whose
 * outputs are committed as tests/golden/edges_cases.json; scipy.ndimage for the morphology; hand-worked cases and
 * invariants (tests/test_oracle_edges.py).
 *
 * Known ambiguity: for an EVEN structuring element (the down-sampled path uses 8x8) the window is taken as
 * [-k/2, k - 1 - k/2] for both dilate and erode, OpenCV's documented formula; the default path (11x11) is symmetric
 * and unambiguous.
 */
#include "pps_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.14159265 /* select_edge.cpp:18 */

void ora_edge_default_params(ora_edge_params* p) {
  /* popup_plane.h:82,184-200, popup_plane.cpp:32-33 */
  p->downsample_contour = 0;
  p->dilation_distance = 11; p->erosion_distance = 11;
  p->pre_vertical_thre = 15; p->pre_minium_len = 15; p->pre_contour_close_thre = 50; p->interval_overlap_thre = 20;
  p->post_short_thre = 30; p->post_bind_dist_thre = 10; p->post_merge_dist_thre = 20; p->post_merge_angle_thre = 10;
  p->post_extend_thre = 15;
  p->pre_boundary_thre = 5; p->pre_merge_angle_thre = 10; p->pre_merge_dist_thre = 10; p->pre_proj_angle_thre = 20;
  p->pre_proj_cover_thre = 0.6; p->pre_proj_cover_large_thre = 0.8; p->pre_proj_dist_thre = 100;
}

/* ---- label map pre-processing: select_edge.cpp:69-78 ------------------------------------------------------- */
static void morph(const unsigned char* src, unsigned char* dst, int w, int h, int k, int is_max) {
  const int a = k / 2; /* default anchor = element centre */
  unsigned char* tmp = (unsigned char*)malloc((size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = is_max ? 0 : 255;
      for (int j = 0; j < k; j++) {
        const int xx = x + j - a;
        if (xx < 0 || xx >= w) continue;
        const int s = src[(size_t)y * w + xx];
        v = is_max ? (s > v ? s : v) : (s < v ? s : v);
      }
      tmp[(size_t)y * w + x] = (unsigned char)v;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = is_max ? 0 : 255;
      for (int i = 0; i < k; i++) {
        const int yy = y + i - a;
        if (yy < 0 || yy >= h) continue;
        const int s = tmp[(size_t)yy * w + x];
        v = is_max ? (s > v ? s : v) : (s < v ? s : v);
      }
      dst[(size_t)y * w + x] = (unsigned char)v;
    }
  free(tmp);
}

void ora_label_preprocess(const unsigned char* label, int w, int h, const ora_edge_params* prm, unsigned char* out, int* ow,
                          int* oh) {
  int W = w, H = h;
  unsigned char* a = (unsigned char*)malloc((size_t)w * h);
  unsigned char* b = (unsigned char*)malloc((size_t)w * h);
  if (prm->downsample_contour) {
    W = (int)lrint(w * 0.5); H = (int)lrint(h * 0.5); /* cv::resize dsize = round(size * 0.5), nearest: src(2x, 2y) */
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int sx = 2 * x < w - 1 ? 2 * x : w - 1, sy = 2 * y < h - 1 ? 2 * y : h - 1;
        a[(size_t)y * W + x] = label[(size_t)sy * w + sx];
      }
    morph(a, b, W, H, 8, 1);
    morph(b, a, W, H, 8, 0);
  } else {
    morph(label, b, W, H, prm->dilation_distance, 1);
    morph(b, a, W, H, prm->erosion_distance, 0);
  }
  for (size_t i = 0; i < (size_t)W * H; i++) out[i] = (unsigned char)(255 - a[i]);
  *ow = W; *oh = H;
  free(a); free(b);
}

/* ---- skimage.measure.find_contours(label, 0) ------------------------------------------------------------------ */
typedef struct { int r, c; } pt_t;
typedef struct { pt_t from, to; } seg_t;

/* marching squares at level 0: a vertex is "high" when > 0; the crossing of an edge sits on its zero end
 * (fraction (0 - from) / (to - from) = 0 or 1).  Segments in raster order of the 2x2 cells. */
static int squares(const unsigned char* a, int w, int h, seg_t** out) {
  int cap = 4096, n = 0;
  seg_t* s = (seg_t*)malloc(sizeof(seg_t) * cap);
  for (int r0 = 0; r0 + 1 < h; r0++)
    for (int c0 = 0; c0 + 1 < w; c0++) {
      const int r1 = r0 + 1, c1 = c0 + 1;
      const int ul = a[(size_t)r0 * w + c0] > 0, ur = a[(size_t)r0 * w + c1] > 0, ll = a[(size_t)r1 * w + c0] > 0,
                lr = a[(size_t)r1 * w + c1] > 0;
      const int sq = ul + 2 * ur + 4 * ll + 8 * lr;
      if (sq == 0 || sq == 15) continue;
      /* crossing points (only meaningful on edges whose ends differ) */
      const pt_t top = {r0, ul ? c1 : c0}, bottom = {r1, ll ? c1 : c0}, left = {ul ? r1 : r0, c0}, right = {ur ? r1 : r0, c1};
      pt_t f[2], t[2];
      int m = 1;
      switch (sq) {
        case 1: f[0] = top; t[0] = left; break;
        case 2: f[0] = right; t[0] = top; break;
        case 3: f[0] = right; t[0] = left; break;
        case 4: f[0] = left; t[0] = bottom; break;
        case 5: f[0] = top; t[0] = bottom; break;
        case 6: f[0] = right; t[0] = top; f[1] = left; t[1] = bottom; m = 2; break;   /* 'low' connectivity */
        case 7: f[0] = right; t[0] = bottom; break;
        case 8: f[0] = bottom; t[0] = right; break;
        case 9: f[0] = top; t[0] = left; f[1] = bottom; t[1] = right; m = 2; break;
        case 10: f[0] = bottom; t[0] = top; break;
        case 11: f[0] = bottom; t[0] = left; break;
        case 12: f[0] = left; t[0] = right; break;
        case 13: f[0] = top; t[0] = right; break;
        default: f[0] = left; t[0] = top; break; /* 14 */
      }
      for (int k = 0; k < m; k++) {
        if (f[k].r == t[k].r && f[k].c == t[k].c) continue; /* degenerate segments are ignored */
        if (n == cap) { cap *= 2; s = (seg_t*)realloc(s, sizeof(seg_t) * cap); }
        s[n].from = f[k]; s[n].to = t[k]; n++;
      }
    }
  *out = s;
  return n;
}

/* contours as doubly linked point lists so that joins are O(1); the two dictionaries of the Python code are
 * direct-mapped arrays over the pixel grid */
typedef struct node { pt_t p; int next, prev; } node_t;
typedef struct { int head, tail, alive; } contour_t;

static int assemble(const seg_t* seg, int nseg, int w, int h, node_t** nodes_out, contour_t** cont_out) {
  int* starts = (int*)malloc(sizeof(int) * (size_t)w * h);
  int* ends = (int*)malloc(sizeof(int) * (size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) starts[i] = ends[i] = -1;
  node_t* nd = (node_t*)malloc(sizeof(node_t) * (size_t)(2 * nseg + 2));
  contour_t* ct = (contour_t*)malloc(sizeof(contour_t) * (size_t)(nseg + 1));
  int nn = 0, nc = 0;
#define KEY(p) ((size_t)(p).r * w + (p).c)
  for (int i = 0; i < nseg; i++) {
    const pt_t fp = seg[i].from, tp = seg[i].to;
    const int tail = starts[KEY(tp)], head = ends[KEY(fp)];
    if (tail >= 0 && head >= 0) {
      if (tail == head) { /* close the contour */
        nd[nn].p = tp; nd[nn].next = -1; nd[nn].prev = ct[head].tail; nd[ct[head].tail].next = nn; ct[head].tail = nn; nn++;
        starts[KEY(tp)] = -1; ends[KEY(fp)] = -1;
      } else if (tail > head) { /* tail was created second: append it to head */
        nd[ct[head].tail].next = ct[tail].head; nd[ct[tail].head].prev = ct[head].tail;
        starts[KEY(tp)] = -1; ends[KEY(nd[ct[tail].tail].p)] = -1; ct[tail].alive = 0;
        ends[KEY(fp)] = -1;
        ct[head].tail = ct[tail].tail;
        ends[KEY(nd[ct[head].tail].p)] = head;
      } else { /* head was created second: prepend it to tail */
        nd[ct[head].tail].next = ct[tail].head; nd[ct[tail].head].prev = ct[head].tail;
        starts[KEY(nd[ct[head].head].p)] = -1; ends[KEY(fp)] = -1; ct[head].alive = 0;
        starts[KEY(tp)] = -1;
        ct[tail].head = ct[head].head;
        starts[KEY(nd[ct[tail].head].p)] = tail;
      }
    } else if (tail < 0 && head < 0) { /* new contour */
      nd[nn].p = fp; nd[nn].prev = -1; nd[nn].next = nn + 1;
      nd[nn + 1].p = tp; nd[nn + 1].prev = nn; nd[nn + 1].next = -1;
      ct[nc].head = nn; ct[nc].tail = nn + 1; ct[nc].alive = 1;
      starts[KEY(fp)] = nc; ends[KEY(tp)] = nc;
      nn += 2; nc++;
    } else if (tail >= 0) { /* prepend the segment to tail */
      nd[nn].p = fp; nd[nn].prev = -1; nd[nn].next = ct[tail].head; nd[ct[tail].head].prev = nn; ct[tail].head = nn; nn++;
      starts[KEY(tp)] = -1; starts[KEY(fp)] = tail;
    } else { /* append to head */
      nd[nn].p = tp; nd[nn].next = -1; nd[nn].prev = ct[head].tail; nd[ct[head].tail].next = nn; ct[head].tail = nn; nn++;
      ends[KEY(fp)] = -1; ends[KEY(tp)] = head;
    }
  }
#undef KEY
  free(starts); free(ends);
  *nodes_out = nd; *cont_out = ct;
  return nc;
}

/* pop_up_fun.py:85-106: the contour with the largest first-to-last distance, as (x, y), every 20th point of
 * [0 : len-1), times 2 when the label map was down-sampled (select_edge.cpp:86-87) */
int ora_ground_contour(const unsigned char* pre, int w, int h, int downsample, float* xy, int cap, int* n_contours,
                       int* n_points) {
  seg_t* seg = NULL;
  const int nseg = squares(pre, w, h, &seg);
  node_t* nd = NULL; contour_t* ct = NULL;
  const int nc = assemble(seg, nseg, w, h, &nd, &ct);
  int best = -1, alive = 0;
  double best_len = -1.0;
  for (int k = 0; k < nc; k++) {
    if (!ct[k].alive) continue;
    alive++;
    const double dr = (double)nd[ct[k].head].p.r - nd[ct[k].tail].p.r, dc = (double)nd[ct[k].head].p.c - nd[ct[k].tail].p.c;
    const double len = sqrt(dr * dr + dc * dc);
    if (len > best_len) { best_len = len; best = k; } /* argmax: first maximum */
  }
  if (n_contours) *n_contours = alive;
  int n = 0, total = 0;
  if (best >= 0) {
    for (int i = ct[best].head; i >= 0; i = nd[i].next) total++;
    int idx = 0;
    for (int i = ct[best].head; i >= 0; i = nd[i].next, idx++) {
      if (idx >= total - 1) break;
      if (idx % 20) continue;
      if (n < cap) { xy[2 * n] = (float)nd[i].p.c * (downsample ? 2.0f : 1.0f); xy[2 * n + 1] = (float)nd[i].p.r * (downsample ? 2.0f : 1.0f); }
      n++;
    }
  }
  if (n_points) *n_points = total;
  free(seg); free(nd); free(ct);
  return n;
}

/* ---- float helpers (matrix_utils.cpp) --------------------------------------------------------------------------- */
static float norm2f(float x, float y) { return sqrtf(x * x + y * y); }
static float normalize_to_pi(float a) { return a > 90 ? a - 180 : (a < -90 ? a + 180 : a); } /* :495-503 */
static float line_angle(const float* l) { /* select_edge.cpp:111 */
  return normalize_to_pi((float)((double)atan2f(l[3] - l[1], l[2] - l[0]) / REF_PI * 180));
}
/* matrix_utils.cpp:331-351 */
static void point_distproj_to_line(const float* bg, const float* ed, const float* q, float* dist, float* proj) {
  const float length = norm2f(ed[0] - bg[0], ed[1] - bg[1]);
  if (length < 0.001) { *dist = norm2f(q[0] - bg[0], q[1] - bg[1]); *proj = -1; return; }
  float t = ((q[0] - bg[0]) * (ed[0] - bg[0]) + (q[1] - bg[1]) * (ed[1] - bg[1])) / length / length;
  const float px = bg[0] + t * (ed[0] - bg[0]), py = bg[1] + t * (ed[1] - bg[1]);
  *dist = norm2f(q[0] - px, q[1] - py);
  if (t > 1) t = 1;
  if (t < 0) t = 0;
  *proj = t;
}
/* matrix_utils.cpp:318-329 */
static float point_dist_line(const float* bg, const float* ed, const float* q) {
  const float length = norm2f(ed[0] - bg[0], ed[1] - bg[1]);
  if (length < 0.001) return norm2f(q[0] - bg[0], q[1] - bg[1]);
  const float t = ((q[0] - bg[0]) * (ed[0] - bg[0]) + (q[1] - bg[1]) * (ed[1] - bg[1])) / length / length;
  const float px = bg[0] + t * (ed[0] - bg[0]), py = bg[1] + t * (ed[1] - bg[1]);
  return norm2f(q[0] - px, q[1] - py);
}
/* matrix_utils.cpp:229-272 (nobottom = false) */
static void direction_hit_boundary(const float* pt, const float* dir, int w, int h, float* out) {
  float lam;
  if (dir[1] < 0) {
    lam = (0.0f - pt[1]) / dir[1];
    if (lam >= 0) { const float x = pt[0] + lam * dir[0], y = pt[1] + lam * dir[1]; if (0 <= (int)x && (int)x <= w - 1) { out[0] = x; out[1] = y; return; } }
  }
  if (dir[1] > 0) {
    lam = (float)((h - 1.0 - pt[1]) / dir[1]);
    if (lam >= 0) { const float x = pt[0] + lam * dir[0], y = pt[1] + lam * dir[1]; if (0 <= (int)x && (int)x <= w - 1) { out[0] = x; out[1] = y; return; } }
  }
  if (dir[0] > 0) {
    lam = (float)((w - 1.0 - pt[0]) / dir[0]);
    if (lam >= 0) { const float x = pt[0] + lam * dir[0], y = pt[1] + lam * dir[1]; if (0 <= (int)y && (int)y <= h - 1) { out[0] = x; out[1] = y; return; } }
  }
  if (dir[0] < 0) {
    lam = (0.0f - pt[0]) / dir[0];
    if (lam >= 0) { const float x = pt[0] + lam * dir[0], y = pt[1] + lam * dir[1]; if (0 <= (int)y && (int)y <= h - 1) { out[0] = x; out[1] = y; return; } }
  }
  out[0] = -1; out[1] = -1;
}

/* max over 10 samples along the line of the distance to the nearest contour point (select_edge.cpp:120-129) */
static float contour_dist(const float* l, const float* cxy, int nc) {
  float worst = -INFINITY;
  for (int k = 0; k < 10; k++) {
    const float f = (float)(k / 10.0);
    const float sx = l[0] + f * (l[2] - l[0]), sy = l[1] + f * (l[3] - l[1]);
    float best = INFINITY;
    for (int i = 0; i < nc; i++) { const float d = norm2f(cxy[2 * i] - sx, cxy[2 * i + 1] - sy); if (d < best) best = d; }
    if (best > worst) worst = best;
  }
  return worst;
}

static void remove_row(float* m, int* n, int row) {
  memmove(m + 4 * row, m + 4 * (row + 1), sizeof(float) * 4 * (size_t)(*n - 1 - row));
  (*n)--;
}

/* ---- interval_tree_optimization (pop_up_fun.py:109-204) ----------------------------------------------------------- */
typedef struct { double b, e; double data[4]; } iv_t;
static int data_cmp(const double* a, const double* b) {
  for (int k = 0; k < 4; k++) { if (a[k] < b[k]) return -1; if (a[k] > b[k]) return 1; }
  return 0;
}
static int iv_cmp(const void* pa, const void* pb) {
  const iv_t* a = (const iv_t*)pa; const iv_t* b = (const iv_t*)pb;
  if (a->b != b->b) return a->b < b->b ? -1 : 1;
  if (a->e != b->e) return a->e < b->e ? -1 : 1;
  return data_cmp(a->data, b->data);
}
static int iv_eq(const iv_t* a, const iv_t* b) { return a->b == b->b && a->e == b->e && data_cmp(a->data, b->data) == 0; }
/* set semantics of the tree: an interval equal to a stored one is not added twice */
static void iv_add(iv_t* set, int* n, const iv_t* x) {
  for (int i = 0; i < *n; i++) if (iv_eq(&set[i], x)) return;
  set[(*n)++] = *x;
}
static void iv_remove(iv_t* set, int* n, const iv_t* x) {
  for (int i = 0; i < *n; i++) if (iv_eq(&set[i], x)) { set[i] = set[*n - 1]; (*n)--; return; }
}
static int dbl_cmp(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : (x > y ? 1 : 0); }

static int interval_tree_optimization(const float* lines, int n, double overlap_thre, float* out) {
  if (n <= 0) return 0;
  float* len = (float*)malloc(sizeof(float) * n);
  int* open = (int*)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) { len[i] = norm2f(lines[4 * i + 2] - lines[4 * i], lines[4 * i + 3] - lines[4 * i + 1]); open[i] = i; }
  int n_open = n;
  iv_t* tree = (iv_t*)malloc(sizeof(iv_t) * (size_t)(n + 2) * (size_t)(2 * n + 2));
  int nt = 0;
#define PUSH_LINE(idx) do { iv_t x; x.b = lines[4 * (idx)]; x.e = lines[4 * (idx) + 2]; for (int q = 0; q < 4; q++) x.data[q] = lines[4 * (idx) + q]; iv_add(tree, &nt, &x); } while (0)
  { /* the longest line first */
    int best = 0;
    for (int i = 1; i < n_open; i++) if (len[open[i]] > len[open[best]]) best = i;
    const int cur = open[best];
    memmove(open + best, open + best + 1, sizeof(int) * (size_t)(n_open - 1 - best)); n_open--;
    PUSH_LINE(cur);
  }
  int* pot = (int*)malloc(sizeof(int) * n);
  while (n_open > 0) {
    int np = 0;
    for (int i = 0; i < n_open; i++) {
      const double q0 = lines[4 * open[i]], q1 = lines[4 * open[i] + 2];
      double inter = 0;
      if (q0 < q1)
        for (int k = 0; k < nt; k++)
          if (tree[k].b < q1 && tree[k].e > q0) {
            double m = q1 - tree[k].b;
            const double o2 = tree[k].e - q0, ql = fabs(q1 - q0), il = tree[k].e - tree[k].b;
            if (o2 < m) m = o2;
            if (ql < m) m = ql;
            if (il < m) m = il;
            inter += m;
          }
      if (inter < overlap_thre) pot[np++] = i;
    }
    if (np == 0) break;
    int best = 0;
    for (int i = 1; i < np; i++) if (len[open[pot[i]]] > len[open[pot[best]]]) best = i;
    const int oi = pot[best], cur = open[oi];
    memmove(open + oi, open + oi + 1, sizeof(int) * (size_t)(n_open - 1 - oi)); n_open--;
    PUSH_LINE(cur);
  }
#undef PUSH_LINE
  /* split_overlaps: cut every interval at every boundary that falls inside it */
  const int raw_seg = nt;
  {
    double* bounds = (double*)malloc(sizeof(double) * 2 * (size_t)nt);
    int nb = 0;
    for (int k = 0; k < nt; k++) { bounds[nb++] = tree[k].b; bounds[nb++] = tree[k].e; }
    qsort(bounds, nb, sizeof(double), dbl_cmp);
    int u = 0;
    for (int k = 0; k < nb; k++) if (u == 0 || bounds[k] != bounds[u - 1]) bounds[u++] = bounds[k];
    nb = u;
    if (nb > 2) {
      iv_t* nw = (iv_t*)malloc(sizeof(iv_t) * (size_t)nt * (size_t)nb);
      int nn = 0;
      for (int k = 0; k + 1 < nb; k++)
        for (int i = 0; i < nt; i++)
          if (tree[i].b <= bounds[k] && bounds[k] < tree[i].e) { iv_t x = tree[i]; x.b = bounds[k]; x.e = bounds[k + 1]; iv_add(nw, &nn, &x); }
      memcpy(tree, nw, sizeof(iv_t) * (size_t)nn); nt = nn;
      free(nw);
    }
    free(bounds);
  }
  const int post = nt != raw_seg;
  if (post) {
    /* of two pieces over the same x range keep the one whose source line spans more in x */
    qsort(tree, nt, sizeof(iv_t), iv_cmp);
    iv_t* del = (iv_t*)malloc(sizeof(iv_t) * (size_t)nt * (size_t)nt + sizeof(iv_t));
    int ndel = 0;
    for (int i = 0; i < nt; i++)
      for (int j = i + 1; j < nt; j++)
        if (tree[i].b == tree[j].b && tree[i].e == tree[j].e) {
          const iv_t* x = (tree[i].data[2] - tree[i].data[0]) < (tree[j].data[2] - tree[j].data[0]) ? &tree[i] : &tree[j];
          iv_add(del, &ndel, x);
        }
    for (int k = 0; k < ndel; k++) iv_remove(tree, &nt, &del[k]);
    free(del);
    /* re-join neighbouring pieces of one source line */
    int can = 1, counter = 0;
    while (can && counter < 100) {
      can = 0; counter++;
      qsort(tree, nt, sizeof(iv_t), iv_cmp);
      for (int i = 0; i < nt && !can; i++)
        for (int j = i + 1; j < nt; j++)
          if ((tree[i].b == tree[j].e || tree[i].e == tree[j].b) && data_cmp(tree[i].data, tree[j].data) == 0) {
            iv_t m = tree[j];
            m.b = tree[i].b < tree[j].b ? tree[i].b : tree[j].b;
            m.e = tree[i].e > tree[j].e ? tree[i].e : tree[j].e;
            const iv_t a = tree[i], b = tree[j];
            iv_remove(tree, &nt, &a); iv_remove(tree, &nt, &b); iv_add(tree, &nt, &m);
            can = 1;
            break;
          }
    }
  }
  qsort(tree, nt, sizeof(iv_t), iv_cmp);
  for (int i = 0; i < nt; i++) {
    const double* raw = tree[i].data;
    if (post && (tree[i].b != raw[0] || tree[i].e != raw[2])) {
      const double f1 = (tree[i].b - raw[0]) / (raw[2] - raw[0]), f2 = (tree[i].e - raw[0]) / (raw[2] - raw[0]);
      out[4 * i] = (float)tree[i].b; out[4 * i + 1] = (float)(int)(f1 * (raw[3] - raw[1]) + raw[1]);
      out[4 * i + 2] = (float)tree[i].e; out[4 * i + 3] = (float)(int)(f2 * (raw[3] - raw[1]) + raw[1]);
    } else
      for (int q = 0; q < 4; q++) out[4 * i + q] = (float)raw[q];
  }
  free(len); free(open); free(pot); free(tree);
  return nt;
}

int ora_interval_tree_optimization(const float* lines, int n, double overlap_thre, float* out) {
  return interval_tree_optimization(lines, n, overlap_thre, out);
}

/* ---- the selection proper: select_edge.cpp:92-405 ------------------------------------------------------------- */
int ora_select_edges_from_contour(const float* cxy, int ncont, int width, int height, const float* lsd, int n_lsd,
                                  const ora_edge_params* prm, float* open_segs, int* n_open, float* closed_segs,
                                  int* n_closed, float* open_in_closed) {
  *n_open = 0; *n_closed = 0;
  float* good = (float*)malloc(sizeof(float) * 4 * (size_t)(n_lsd + 1));
  int ng = 0;
  /* Step 1 + 2: drop short, border-hugging and near-vertical lines, then lines far from the CNN boundary; order by x */
  for (int i = 0; i < n_lsd; i++) {
    const float* l = lsd + 4 * i;
    const float length = norm2f(l[2] - l[0], l[3] - l[1]);
    if (length < prm->pre_minium_len) continue;
    const double bt = prm->pre_boundary_thre;
    if ((l[0] < bt && l[2] < bt) || (l[0] > width - bt && l[2] > width - bt) || (l[1] < bt && l[3] < bt) ||
        (l[1] > height - bt && l[3] > height - bt))
      continue;
    const float ang = line_angle(l);
    if (!(fabsf(fabsf(ang) - 90) > prm->pre_vertical_thre)) continue;
    if (!(contour_dist(l, cxy, ncont) < prm->pre_contour_close_thre)) continue;
    float* g = good + 4 * ng++;
    if (l[0] > l[2]) { g[0] = l[2]; g[1] = l[3]; g[2] = l[0]; g[3] = l[1]; }
    else memcpy(g, l, sizeof(float) * 4);
  }
  /* Step 3: chain nearly collinear pieces end to start (select_edge.cpp:143-179) */
  {
    int can = 1, counter = 0;
    float* ang = (float*)malloc(sizeof(float) * (size_t)(ng + 1));
    while (can && counter < 100) {
      counter++; can = 0;
      for (int i = 0; i < ng; i++) ang[i] = line_angle(good + 4 * i);
      for (int s1 = 0; s1 < ng && !can; s1++)
        for (int s2 = s1 + 1; s2 < ng; s2++) {
          const float diff = fabsf(ang[s1] - ang[s2]);
          const float ad = diff < 180 - diff ? diff : 180 - diff;
          if (ad < prm->pre_merge_angle_thre) {
            float* a = good + 4 * s1; float* b = good + 4 * s2;
            const float d12 = norm2f(a[2] - b[0], a[3] - b[1]), d21 = norm2f(b[2] - a[0], b[3] - a[1]);
            if (d12 < prm->pre_merge_dist_thre) { a[2] = b[2]; a[3] = b[3]; remove_row(good, &ng, s2); can = 1; break; }
            if (d21 < prm->pre_merge_dist_thre) { a[0] = b[0]; a[1] = b[1]; remove_row(good, &ng, s2); can = 1; break; }
          }
        }
    }
    /* Step 4: of two near-parallel lines that cover each other, drop one (select_edge.cpp:187-253) */
    can = 1; counter = 0;
    while (can && counter < 100) {
      counter++; can = 0;
      for (int i = 0; i < ng; i++) ang[i] = line_angle(good + 4 * i);
      for (int s1 = 0; s1 < ng && !can; s1++)
        for (int s2 = s1 + 1; s2 < ng; s2++) {
          const float diff = fabsf(ang[s1] - ang[s2]);
          const float ad = diff < 180 - diff ? diff : 180 - diff;
          if (!(ad < prm->pre_proj_angle_thre)) continue;
          const float* a = good + 4 * s1; const float* b = good + 4 * s2;
          float d1b, p1b, d1e, p1e, d2b, p2b, d2e, p2e;
          point_distproj_to_line(b, b + 2, a, &d1b, &p1b);
          point_distproj_to_line(b, b + 2, a + 2, &d1e, &p1e);
          point_distproj_to_line(a, a + 2, b, &d2b, &p2b);
          point_distproj_to_line(a, a + 2, b + 2, &d2e, &p2e);
          const double dt = prm->pre_proj_dist_thre;
          if (!(d1b < dt && d1e < dt && d2b < dt && d2e < dt)) continue;
          const float c12 = fabsf(p1b - p1e), c21 = fabsf(p2b - p2e);
          if (!(c12 > prm->pre_proj_cover_thre || c21 > prm->pre_proj_cover_thre)) continue;
          int del;
          if ((c12 < c21 ? c12 : c21) < prm->pre_proj_cover_large_thre) del = c12 > c21 ? s2 : s1;
          else del = contour_dist(a, cxy, ncont) > contour_dist(b, cxy, ncont) ? s1 : s2;
          remove_row(good, &ng, del);
          can = 1;
          break;
        }
    }
    free(ang);
  }
  if (ng == 0) { free(good); return 0; }
  /* Step 5 */
  float* segs = (float*)malloc(sizeof(float) * 4 * (size_t)(ng + 2) * (size_t)(2 * ng + 2));
  int ns = interval_tree_optimization(good, ng, prm->interval_overlap_thre, segs);
  free(good);
  /* Step 6: drop short pieces, bind near end points, merge successive near-collinear pieces */
  {
    int u = 0;
    for (int i = 0; i < ns; i++)
      if (norm2f(segs[4 * i + 2] - segs[4 * i], segs[4 * i + 3] - segs[4 * i + 1]) > prm->post_short_thre) { memmove(segs + 4 * u, segs + 4 * i, sizeof(float) * 4); u++; }
    ns = u;
  }
  if (ns == 0) { free(segs); return 0; }   /* (the reference would index row 0 of an empty matrix here) */
  {
    int can = 1, counter = 0;
    while (can && counter < 100) {
      counter++; can = 0;
      for (int s = 0; s + 1 < ns; s++) {
        float* a = segs + 4 * s; float* b = a + 4;
        if (a[2] != b[0] || a[3] != b[1])
          if (norm2f(a[2] - b[0], a[3] - b[1]) < prm->post_bind_dist_thre) {
            const int mx = (int)((a[2] + b[0]) / 2), my = (int)((a[3] + b[1]) / 2);
            a[2] = (float)mx; a[3] = (float)my; b[0] = (float)mx; b[1] = (float)my;
            can = 1;
          }
      }
    }
    can = 1; counter = 0;
    while (can && counter < 100) {
      counter++; can = 0;
      for (int s = 0; s + 1 < ns; s++) {
        float* a = segs + 4 * s; float* b = a + 4;
        const float diff = fabsf(line_angle(a) - line_angle(b));
        const float ad = diff < 180 - diff ? diff : 180 - diff;
        if (ad < prm->post_merge_angle_thre) {
          const float d1 = point_dist_line(b, b + 2, a), d2 = point_dist_line(b, b + 2, a + 2);
          const float d3 = point_dist_line(a, a + 2, b), d4 = point_dist_line(a, a + 2, b + 2);
          const double mt = prm->post_merge_dist_thre;
          if (((d1 < mt) & (d2 < mt)) | ((d3 < mt) & (d4 < mt))) {
            a[2] = b[2]; a[3] = b[3];
            remove_row(segs, &ns, s + 1);
            can = 1;
            break;
          }
        }
      }
    }
  }
  /* extend the first / last piece to the image border when the extension stays near the CNN boundary */
  {
    float* first = segs; float* last = segs + 4 * (ns - 1);
    const float old_start[2] = {first[0], first[1]}, old_end[2] = {last[2], last[3]};
    float dir[2], hit[2];
    dir[0] = first[0] - first[2]; dir[1] = first[1] - first[3];
    direction_hit_boundary(first + 2, dir, width, height, hit);
    const int sx = (int)hit[0], sy = (int)hit[1];
    dir[0] = last[2] - last[0]; dir[1] = last[3] - last[1];
    direction_hit_boundary(last, dir, width, height, hit);
    const int ex = (int)hit[0], ey = (int)hit[1];
    first[0] = (float)sx; first[1] = (float)sy;
    last[2] = (float)ex; last[3] = (float)ey;
    const float b0 = contour_dist(first, cxy, ncont), b1 = contour_dist(last, cxy, ncont);
    if (b0 > prm->post_extend_thre) { first[0] = old_start[0]; first[1] = old_start[1]; }
    if (b1 > prm->post_extend_thre) { last[2] = old_end[0]; last[3] = old_end[1]; }
  }
  /* outputs: the open pieces, the closed polyline with connecting pieces inserted, and where each open piece
   * sits in the closed list */
  memcpy(open_segs, segs, sizeof(float) * 4 * (size_t)ns);
  *n_open = ns;
  int nc = 0;
  for (int s = 0; s < ns; s++) {
    if (s > 0 && (segs[4 * (s - 1) + 2] != segs[4 * s] || segs[4 * (s - 1) + 3] != segs[4 * s + 1])) {
      float* c = closed_segs + 4 * nc++;
      c[0] = segs[4 * (s - 1) + 2]; c[1] = segs[4 * (s - 1) + 3]; c[2] = segs[4 * s]; c[3] = segs[4 * s + 1];
    }
    open_in_closed[s] = (float)nc;
    memcpy(closed_segs + 4 * nc++, segs + 4 * s, sizeof(float) * 4);
  }
  *n_closed = nc;
  free(segs);
  return ns;
}

int ora_select_ground_edges(const unsigned char* label, int w, int h, const float* lsd, int n_lsd, const ora_edge_params* prm,
                            float* open_segs, int* n_open, float* closed_segs, int* n_closed, float* open_in_closed) {
  unsigned char* pre = (unsigned char*)malloc((size_t)w * h);
  int pw, ph;
  ora_label_preprocess(label, w, h, prm, pre, &pw, &ph);
  const int cap = (pw + ph) * 4 + 64;
  float* cxy = (float*)malloc(sizeof(float) * 2 * (size_t)cap);
  int ncont = ora_ground_contour(pre, pw, ph, prm->downsample_contour, cxy, cap, NULL, NULL);
  if (ncont > cap) ncont = cap;
  free(pre);
  *n_open = 0; *n_closed = 0;
  int r = 0;
  if (ncont > 0) r = ora_select_edges_from_contour(cxy, ncont, w, h, lsd, n_lsd, prm, open_segs, n_open, closed_segs, n_closed, open_in_closed);
  free(cxy);
  return r;
}
