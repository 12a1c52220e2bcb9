/*
 * pps_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C, fp64, single-threaded restatement of the reference's plane-SLAM
 * graph solve (pop_planar_slam + its vendored iSAM).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product (libpps.so) never links or calls it.
 *
 * PARITY UNPINNED for the plane factors: the reference ships no tests / golden
 * vectors for this path and cannot be compiled here (Eigen, Boost, SuiteSparse,
 * ROS absent), so the plane arithmetic is pinned only by (a) independent numpy
 * evaluations (oracle/numpy_ref.py, oracle/numpy_assoc.py -> tests/golden/) and
 * (b) analytic invariants.  The pose factors, the LM loop and the sparse solve
 * ARE pinned against external data: the public pose-graph logs bundled with the
 * reference's iSAM (tests/golden/isam_data: sphere400, sphere2500 + noise-free
 * ground truth) come out at normalised chi2 1.014 / 0.996 and 0.89 m RMSE to
 * the ground truth (tests/test_graphio.py).
 *
 * Conventions: quaternions are stored (x,y,z,w) everywhere (Eigen coeffs()
 * order); a plane (a,b,c,d) is the quaternion x=a,y=b,z=c,w=d
 * (pop_planar_slam/src/isam_plane3d.h:74-76).  A pose is 7 doubles
 * (tx,ty,tz,qx,qy,qz,qw).  Pose measurements are 6 doubles
 * (x,y,z,yaw,pitch,roll) = Pose3d::vector() (Pose3d.h:138-145).
 * sqrt-information matrices are packed upper-triangular, row-major
 * (noise_to_string order, Factor.h:169-190): 21 doubles for 6x6, 6 for 3x3.
 */
#ifndef PPS_ORACLE_H
#define PPS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_graph ora_graph;

/* Properties.h:37-110 with the app overrides of Mapping.cpp:32-43 as defaults */
typedef struct ora_props {
  double epsilon2;        /* 1e-3  stop on ||delta||               */
  double epsilon_abs;     /* 1e-4  stop on chi2                    */
  double epsilon_rel;     /* 1e-6  stop on relative improvement    */
  int    max_iterations;  /* 500                                   */
  double lm_lambda0;      /* 1e-6                                  */
  double lm_lambda_factor;/* 10                                    */
  int    analytic;        /* 0 = reference-faithful central differences (numericalDiff.cpp) */
  int    cache_ordering;  /* 0 = re-order every factorisation like Cholesky.cpp:98 */
  int    threads;         /* > 1: the closed-form Jacobian sweep (analytic = 1) runs on that many OpenMP threads; the reference has one */
} ora_props;

void ora_default_props(ora_props* p);

ora_graph* ora_create(const ora_props* p);
void ora_destroy(ora_graph* g);

/* nodes (Slam::add_node, Slam.cpp:91-94); return node id */
int ora_add_pose(ora_graph* g, const double tq[7]);
int ora_add_plane(ora_graph* g, const double abcd[4]);  /* normalised like Plane3d(Vector4d) */

/* factors (Slam::add_factor, Slam.cpp:96-105); return factor id */
int ora_add_pose_prior(ora_graph* g, int pose, const double meas6[6], const double sqrtinf_ut[21]);
int ora_add_odometry(ora_graph* g, int p1, int p2, const double meas6[6], const double sqrtinf_ut[21]);
int ora_add_plane_obs(ora_graph* g, int pose, int plane, const double meas4[4], const double sqrtinf_ut[6]);
int ora_add_plane_prior(ora_graph* g, int plane, const double meas4[4], const double sqrtinf_ut[6]);
/* Pose3d_Plane3d_Factor2 (src/isam_plane3d.h:314-424): measurement re-popped from the ground-edge rays each evaluation */
int ora_add_plane_obs2(ora_graph* g, int pose, int plane, const double meas4[4], const double ray6[6], const double sqrtinf_ut[6]);
void ora_edge_ray(const float invK[9], const float seg2d[4], double ray6[6]);                 /* isam_plane3d.h:361-373 */
void ora_repop_wall_plane(const double tq[7], const double ray6[6], double out4[4]);         /* isam_plane3d.cpp:20-55 */
void ora_set_measurement(ora_graph* g, int fid, const double meas4[4]);  /* Factor.h:206 */
void ora_remove_factor(ora_graph* g, int fid);
void ora_remove_node(ora_graph* g, int nid);

/* solve */
int    ora_batch_optimize(ora_graph* g);   /* Slam::batch_optimization -> LM; returns iterations */
void   ora_update(ora_graph* g);           /* Slam::update with mod_batch=1: one GN step     */
double ora_chi2(ora_graph* g);             /* Slam::chi2(ESTIMATE), Slam.cpp:266-268        */

/* state access */
int  ora_num_nodes(const ora_graph* g);
int  ora_num_factors(const ora_graph* g);
int  ora_node_dim(const ora_graph* g, int nid);
void ora_get_pose(const ora_graph* g, int nid, double tq[7]);
void ora_get_plane(const ora_graph* g, int nid, double abcd[4]);
void ora_set_pose(ora_graph* g, int nid, const double tq[7]);
void ora_set_plane(ora_graph* g, int nid, const double abcd[4]);

/* per-factor evaluation for fixture tests.  sel: 0 = LINPOINT, 1 = ESTIMATE */
int  ora_factor_dim(const ora_graph* g, int fid);
int  ora_factor_cols(const ora_graph* g, int fid);
void ora_factor_error(ora_graph* g, int fid, int sel, double* r_out);
/* H is (dim x cols) row-major; analytic=0 -> numericalDiff restatement */
void ora_factor_jacobian(ora_graph* g, int fid, int analytic, double* H_out, double* r_out);

/* LM trace of the last batch_optimize: per trial (lambda, chi2_new, accepted) */
int  ora_trace_len(const ora_graph* g);
void ora_trace_get(const ora_graph* g, int i, double* lambda, double* chi2_new, int* accepted);
double ora_initial_chi2(const ora_graph* g);

/* timers of the last solve, seconds: [0] linearise, [1] factor+solve, [2] retract+chi2, [3] ordering (subset of 1) */
void ora_timers(const ora_graph* g, double t[4]);
/* nnz(L) of the last factorisation */
long ora_last_nnzL(const ora_graph* g);

/* free-standing geometry helpers exposed for unit tests */
void ora_plane_transform_to(const double abcd[4], const double tq[7], double out[4]);   /* isam_plane3d.h:180-182 */
void ora_plane_transform_from(const double abcd[4], const double tq[7], double out[4]); /* isam_plane3d.h:186-188 */
void ora_plane_exmap(const double abcd[4], const double d[3], double out[4]);            /* isam_plane3d.h:101-127 */
void ora_pose_exmap(const double tq[7], const double d[6], double out[7]);               /* Pose3d.h:131-136 */
void ora_pose_vector(const double tq[7], double v6[6]);                                  /* Pose3d.h:138-145 */
void ora_pose_from_vector(const double v6[6], double tq[7]);                             /* Pose3d.h:152-155 */
void ora_pose_oplus(const double a[7], const double d[7], double out[7]);                /* Pose3d.h:222-224 */
void ora_pose_ominus(const double a[7], const double b[7], double out[7]);               /* Pose3d.h:233-235 */

/* ---- plane data association: Mapper_mono::findClosestPlane (src/Mapping.cpp:256-397) ---------- */
typedef struct ora_assoc_params {   /* Mapping.h:70-77 defaults: 50, 4, 0.5, 60, 5 */
  double edge_asso_2ddist, edge_asso_planedist, edge_asso_proj, edge_asso_angle;
  int assoc_near_frames;
} ora_assoc_params;
typedef struct ora_landmark {       /* what findClosestPlane reads of a Map_plane (Map_plane.h:28-44) */
  double plane[4];                  /* plane_vertex->value(), world frame */
  int frame_plane_indice;           /* 0 = ground, >= 1 wall */
  int frame_seq_id;
  int deleted;                      /* deteted_by_merge */
  float seg2d[4];                   /* plane_bound_close_2D_polys columns 0,1 (u0,v0,u1,v1) */
  float seg3d[4];                   /* plane_bound_close_3D_polys columns 0,1, x and y only (x0,y0,x1,y1) */
} ora_landmark;
/* best = index into lm[] or -1; best_err = bestReprojError (-1 when nothing scored / ground short-cut) */
void ora_find_closest_plane(const double est_pose[7], const double plane_local[4], int frame_plane_indice,
                            int frame_seq_id, const float seg2d[4], const float seg3d[4], const ora_landmark* lm,
                            int n_lm, const ora_assoc_params* prm, int* best, double* best_err);
/* Mapping.cpp:112-126 */
float ora_point_proj_to_lineseg(const float begin_pt[2], const float end_pt[2], const float query_pt[2]);
/* Plane3d::project_to_plane on an fp32 point (Mapping.cpp:617-618): fp64 arithmetic, cast back to fp32 */
void ora_project_to_plane(const double abcd[4], const float pt[3], float out[3]);

/* pop-up (fp32) restatement: pop_up_wall/libs/popup_plane.cpp:654-705.
 * seg2d: n x 4 (u1 v1 u2 v2) row-major, invK 3x3 row-major, T_wc 4x4 row-major.
 * planes_out: (n+1) x 4, row 0 = ground plane in the sensor frame. */
void ora_popup_planes(const float* seg2d, int n, const float invK[9], const float T_wc[16], float* planes_out);
/* same, also returning ground_seg3d_lines_world (n x 6: x0,y0,0,x1,y1,0; popup_plane.cpp:569-578) */
void ora_popup_planes_ex(const float* seg2d, int n, const float invK[9], const float T_wc[16], float* planes_out,
                         float* seg3d_world);
/* per-pixel pop-up: generate_cloud + matrixToCloud (popup_plane.cpp:807-863,925-985) given a
 * per-pixel plane-id mask (-1 = none).  xyz_out: 3 floats per pixel (world), valid_out: 1 = kept. */
void ora_popup_cloud(const int* plane_id, int width, int height, const float invK[9], const float T_wc[16],
                     const float* planes_sensor, int nplanes, float depth_thre, float ceiling_thre,
                     float* xyz_out, unsigned char* valid_out);
/* get_depth_map_good (popup_plane.cpp:866-921) without the half-res resize */
void ora_popup_depth(const int* plane_id, int width, int height, const float invK[9], const float T_wc[16],
                     const float* planes_sensor, int nplanes, const float ceiling_plane_sensor[4],
                     float ceiling_thre, float* depth_out);

/* popup_plane.cpp:616-640: all_plane_dist_to_cam (n+1) and good-plane flags (n+1; ground always good).  actual: plane indices
 * (>= 1) of the non-connecting edges, n_actual = 0: all edges count */
void ora_popup_plane_info(const float* seg2d, int n, const float invK[9], const float T_wc[16], float plane_cam_dist_thre,
                          const int* actual, int n_actual, float* dist_to_cam, int* good);
/* popup_plane.cpp:913-917: depth map known on the even pixels -> cv::resize 0.5 (INTER_AREA for a factor of exactly 2), x 4,
 * cv::resize 2 (INTER_LINEAR).  w, h even. */
void ora_depth_fill_half(const float* sparse, int w, int h, float* out);
/* popup_plane::find_2d_3d_closed_polygon_simplemode (popup_plane.cpp:409-500), walllength_threshold <= 0: closed 2-D polygon
 * of every wall (plane 0 = ground: none).  verts: up to 8 (x, y) pairs per wall; off[n + 2].  Returns the vertex count. */
int ora_popup_polygons_simple(const float* seg2d, int n, const float K[9], const float invK[9], const float T_wc[16], int width,
                              int height, float* verts, int* off);
/* pps_raster_oracle.c: popup_plane::closed_polygons_homo_pts (popup_plane.cpp:81-116) per polygon -- float -> int
 * truncation, boundingRect, cv::fillConvexPoly (restated from OpenCV's published drawing.cpp), findNonZero -- composed
 * into a plane-id map (later planes overwrite, -1 = none).  step 2 = downsample_poly. */
void ora_popup_mask(const float* polys, const int* poly_off, int nplanes, int width, int height, int step, int* plane_id);
/* cv::fillConvexPoly(img, pts, npts, 255) on a zeroed width x height CV_8U image */
void ora_fill_convex_poly(const int* pts_xy, int npts, int width, int height, unsigned char* img);

/* ---- ground-edge selection: popup_plane::edge_get_polygons (pop_up_wall/libs/select_edge.cpp:66-409) with its
 * Python helpers (pop_up_python/.../pop_up_fun.py:85-204); restated in pps_edges_oracle.c ---------------- */
typedef struct ora_edge_params {
  int downsample_contour;                    /* popup_plane.h:82 */
  int dilation_distance, erosion_distance;   /* popup_plane.cpp:32-33 */
  /* popup_plane.h:184-192 (ROS-settable) */
  double pre_vertical_thre, pre_minium_len, pre_contour_close_thre, interval_overlap_thre, post_short_thre,
         post_bind_dist_thre, post_merge_dist_thre, post_merge_angle_thre, post_extend_thre;
  /* popup_plane.h:194-200 */
  double pre_boundary_thre, pre_merge_angle_thre, pre_merge_dist_thre, pre_proj_angle_thre, pre_proj_cover_thre,
         pre_proj_cover_large_thre, pre_proj_dist_thre;
} ora_edge_params;
void ora_edge_default_params(ora_edge_params* p);
/* select_edge.cpp:69-78: [half-size nearest], dilate, erode, 255 - x.  out holds ow*oh bytes (<= w*h) */
void ora_label_preprocess(const unsigned char* label, int w, int h, const ora_edge_params* prm, unsigned char* out,
                          int* ow, int* oh);
/* pop_up_fun.py:85-106 on a pre-processed map: returns the number of sub-sampled contour points (x, y) */
int ora_ground_contour(const unsigned char* pre, int w, int h, int downsample, float* xy, int cap, int* n_contours,
                       int* n_points);
/* pop_up_fun.py:109-204; out holds up to (2n+2)(n+2) rows of 4 */
int ora_interval_tree_optimization(const float* lines, int n, double overlap_thre, float* out);
/* select_edge.cpp:92-405 given the sub-sampled contour; segment outputs hold up to 2*n_lsd+2 rows of 4 */
int ora_select_edges_from_contour(const float* cxy, int ncont, int width, int height, const float* lsd, int n_lsd,
                                  const ora_edge_params* prm, float* open_segs, int* n_open, float* closed_segs,
                                  int* n_closed, float* open_in_closed);
/* the whole of edge_get_polygons: label map (ground = 255) + LSD lines (n x 4: x1 y1 x2 y2) -> open segments, closed
 * polyline, index of each open segment in the closed list */
int ora_select_ground_edges(const unsigned char* label, int w, int h, const float* lsd, int n_lsd,
                            const ora_edge_params* prm, float* open_segs, int* n_open, float* closed_segs, int* n_closed,
                            float* open_in_closed);

#ifdef __cplusplus
}
#endif
#endif
