/*
 * pps.h -- C-ABI of the MI355X-native plane-SLAM backend (libpps.so).
 *
 * One opaque handle `pps_graph` replaces, wholesale, the three seams the
 * reference exposes for this path (paths relative to
 * /root/reference/pop_planar_slam):
 *   - isam::Slam public API            Thirdparty/isam/include/isam/Slam.h:82-215
 *   - isam::OptimizationInterface      Thirdparty/isam/include/isam/OptimizationInterface.h:224-245
 *   - isam::Cholesky (solver plugin)   Thirdparty/isam/include/isam/Cholesky.h:148-178
 * plus the popup_plane statics the mapper calls
 *   - update_plane_equation_from_seg   /root/reference/pop_up_wall/include/pop_up_wall/popup_plane.h:137-139
 *   - generate_cloud / get_depth_map_good  popup_plane.h:142-143,155-157
 *
 * Everything is plain pointers and sizes; no C++/torch types.  All functions
 * return a status (PPS_OK == 0) and never exit()/abort() (the reference's
 * require() does: Thirdparty/isam/include/isam/util.h:174-184).
 *
 * Conventions
 *   quaternion  (x,y,z,w)                      -- Eigen coeffs() order
 *   pose        (tx,ty,tz,qx,qy,qz,qw)         -- isam::Pose3d (Pose3d.h:70-274)
 *   plane       unit 4-vector (a,b,c,d)        -- isam::Plane3d (src/isam_plane3d.h:27-193)
 *   meas6       (x,y,z,yaw,pitch,roll)         -- Pose3d::vector() (Pose3d.h:138-145)
 *   sqrtinf_ut  packed upper triangle, row-major (Factor.h:169-190): 21 (6x6) / 6 (3x3)
 *   ids         handle-local integers starting at 0 (the reference uses
 *               process-global counters, Slam.cpp:47-48)
 *
 * Threading: one handle = one device + one HIP stream; distinct handles may be
 * driven from distinct host threads.  No global mutable state.
 */
#ifndef PPS_H
#define PPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPS_VERSION 304   /* round.minor: bump whenever a struct of this header changes layout or an entry point is added (pps_stats grew in 200; pps_debug_front_factor: 301;
                              pps_multi_save_state / pps_multi_restore_state: 302; pps_debug_exmap AND pps_multi_phase_times' counts[] grown from 2 to 4
                              entries -- a caller built against 302 that passes counts[2] must be rebuilt: 303; pps_popup_run_async / _planes_wait / _wait: 304) */

typedef struct pps_graph pps_graph;

enum pps_status {
  PPS_OK = 0,
  PPS_EINVAL = 1,     /* bad argument / unknown id                                   */
  PPS_ENOTPD = 2,     /* normal equations not positive definite (silent in CHOLMOD)  */
  PPS_EHIP = 3,       /* HIP runtime error (no device, OOM, launch failure)          */
  PPS_ENOMEM = 4,
  PPS_ESTATE = 5      /* operation not valid in the current state                    */
};

/* Jacobian evaluation mode of the per-edge sweep */
enum pps_jacobian_mode {
  PPS_JAC_NUMERIC = 0,   /* central differences, eps = 1e-4, through exmap: numericalDiff.cpp:32-87 (reference behaviour) */
  PPS_JAC_ANALYTIC = 1   /* closed-form derivative of the same residuals */
};

/* Mirrors isam::Properties (Properties.h:37-110); defaults = the app's values (Mapping.cpp:32-43). */
typedef struct pps_props {
  double epsilon2;          /* 1e-3 : stop when ||delta|| <= epsilon2                  */
  double epsilon_abs;       /* 1e-4 : stop when chi2 <= epsilon_abs                    */
  double epsilon_rel;       /* 1e-6 : stop when improvement < epsilon_rel * chi2       */
  int    max_iterations;    /* 500                                                      */
  double lm_lambda0;        /* 1e-6                                                     */
  double lm_lambda_factor;  /* 10                                                       */
  int    jacobian_mode;     /* enum pps_jacobian_mode, default PPS_JAC_NUMERIC          */
  int    device;            /* HIP device ordinal, default 0                            */
  int    verbose;           /* 0 = quiet (prop.quiet = true, Mapping.cpp:35)            */
} pps_props;

void pps_default_props(pps_props* p);
int  pps_version(void);
const char* pps_last_error(const pps_graph* g);      /* thread-compatible, handle-local */

/* ---- lifetime ------------------------------------------------------------------------ */
int pps_graph_create(const pps_props* props, pps_graph** out);   /* isam::Slam::Slam, Slam.cpp:69-78 */
int pps_graph_destroy(pps_graph* g);
int pps_get_props(const pps_graph* g, pps_props* out);           /* Slam::properties()     */
int pps_set_props(pps_graph* g, const pps_props* p);             /* Slam::set_properties() */

/* ---- nodes: Slam::add_node (Slam.cpp:91-94) + NodeT::init (Node.h:121-124) ------------ */
int pps_add_pose(pps_graph* g, const double tq[7], int* id);
int pps_add_plane(pps_graph* g, const double abcd[4], int* id);  /* normalised like Plane3d(Vector4d), isam_plane3d.h:59-66 */

/* ---- factors: Slam::add_factor (Slam.cpp:96-105) -------------------------------------- */
/* Pose3d_Factor            slam3d.h:58-89   */
int pps_add_pose_prior(pps_graph* g, int pose, const double meas6[6], const double sqrtinf_ut[21], int* fid);
/* Pose3d_Pose3d_Factor     slam3d.h:91-193  */
int pps_add_odometry(pps_graph* g, int pose1, int pose2, const double meas6[6], const double sqrtinf_ut[21], int* fid);
/* Pose3d_Plane3d_Factor    src/isam_plane3d.h:221-308 (relative = false, Mapping.cpp:21,513) */
int pps_add_plane_obs(pps_graph* g, int pose, int plane, const double meas4[4], const double sqrtinf_ut[6], int* fid);
/* Plane3d_Factor           src/isam_plane3d.h:428-474 */
/* Pose3d_Plane3d_Factor2 (src/isam_plane3d.h:314-424; the mapper's disabled call Mapping.cpp:515-521): a plane
 * observation whose measured plane is re-popped from the 2-D ground edge at every residual evaluation
 * (isam::get_wall_plane_equation, src/isam_plane3d.cpp:20-55, fp64) instead of being stored.  ray6 = the
 * 3x2 ground_edge_ray (column-major: ray of end point 0, ray of end point 1) from pps_edge_ray.  meas4 is
 * kept like FactorT::_measure but does not enter the error.  Its Jacobian is taken by central differences in
 * both jacobian modes (the pose perturbation moves the measurement).  Not for the ground plane. */
int pps_add_plane_obs2(pps_graph* g, int pose_id, int plane_id, const double meas4[4], const double ray6[6],
                       const double sqrtinf_ut[6], int* fid);
/* Pose3d_Plane3d_Factor2::precompute_edge_ray (src/isam_plane3d.h:361-373): fp32 invK * (u,v,1), cast to fp64 */
int pps_edge_ray(const float invK[9], const float seg2d[4], double ray6[6]);
int pps_add_plane_prior(pps_graph* g, int plane, const double meas4[4], const double sqrtinf_ut[6], int* fid);

/* FactorT::set_measurement (Factor.h:206) for plane factors; batched form for
 * Mapper_mono::update_plane_measurement (Mapping.cpp:590-607) */
int pps_set_measurement(pps_graph* g, int fid, const double meas4[4]);
int pps_set_measurements(pps_graph* g, int n, const int* fids, const double* meas4 /* n x 4 */);

/* Slam::remove_factor / remove_node (Slam.cpp:107-126); removing a node removes its factors */
int pps_remove_factor(pps_graph* g, int fid);
int pps_remove_node(pps_graph* g, int nid);

/* ---- solve ---------------------------------------------------------------------------- */
/* Slam::update with mod_batch = 1 (Slam.cpp:157-175 -> Optimizer::relinearize, Optimizer.cpp:114-185):
 * relinearise at the estimate, one Gauss-Newton step (lambda = 0). */
int pps_update(pps_graph* g);
/* Slam::batch_optimization (Slam.cpp:198-210) -> Optimizer::levenberg_marquardt (Optimizer.cpp:371-467) */
int pps_batch_optimize(pps_graph* g, int* iterations);
/* Slam::chi2(ESTIMATE) (Slam.cpp:266-268) */
int pps_chi2(pps_graph* g, double* chi2);

/* ---- many graphs side by side (BASELINE config 4 on one device; north_star reports graphs/sec) ----------------
 * One C2-size LM solve is a dependency chain that occupies a few dozen of the 256 CUs.  pps_multi runs
 * Optimizer::levenberg_marquardt (Optimizer.cpp:371-467) on n independent graphs in rounds: every kernel of an LM
 * trial is launched once for a chunk of up to 128 graphs, lambda / accept / reject stay per graph (host side, one 32-byte record per
 * graph and round).  While a chunk holds at most 120 000 factors, arithmetic, lambda schedule, iteration count and trace of every
 * graph are exactly -- bit for bit -- those of its own pps_batch_optimize.  A larger chunk takes the throughput forms (K1 as one
 * thread per factor without product records, K2 multiplying the Jacobian slices): the same sums in another rounding order, H and
 * chi2 equal to about 1e-12 relative, LM verdicts and iteration counts the same on every graph measured; which chunk a graph
 * lands in depends on the factor count of the whole batch, so below that level a graph's low-order bits can depend on what else
 * is in its batch (pps_multi_phase_times reports the forms taken).  The graphs keep belonging to the caller (same device; they must outlive the pps_multi and must
 * not be used from another thread during the call); topology edits between calls are picked up.  Graphs with
 * loop-closure fronts (dense-front kernels) are refused with PPS_ESTATE.
 *   iterations[n], status[n] (either may be NULL): LM iterations and PPS_OK / PPS_ENOTPD per graph. */
typedef struct pps_multi pps_multi;
int pps_multi_create(int n, pps_graph* const* graphs, pps_multi** out);
int pps_multi_destroy(pps_multi* m);
const char* pps_multi_last_error(const pps_multi* m);
int pps_multi_optimize(pps_multi* m, int* iterations, int* status);
int pps_multi_rounds(const pps_multi* m, int* rounds);    /* rounds of the last call (of the chunk of graphs that took most) = its longest LM run + 1 */
/* pps_save_state / pps_restore_state of every graph of the batch; the restore is ONE launch and complete on return (a benchmark
 * that solves the same graphs again from their initial estimates pays 30 us for it instead of a copy per handle) */
int pps_multi_save_state(pps_multi* m);
int pps_multi_restore_state(pps_multi* m);
/* level 1: HIP events at the phase boundaries of every round (no host syncs); after the next pps_multi_optimize
 * sec[5] = device seconds in K1 (Jacobian sweep) | K2 (H blocks) | K3 factor | K3 back-substitution | trial step + chi2,
 * counts[4] (may be NULL) = graphs re-linearised | factorised, summed over the rounds | chunks the batch of the last call was cut
 * into | forms its chunks took: bit 0 = thread-per-factor K1 + class-body K2 (throughput), bit 1 = one launch per tree level (K3);
 * the last two are valid without profiling */
int pps_multi_set_profiling(pps_multi* m, int level);
int pps_multi_phase_times(const pps_multi* m, double sec[5], long long counts[4]);

/* ---- state access (NodeT::value(), Node.h:130) ---------------------------------------- */
int pps_num_nodes(const pps_graph* g, int* n);
int pps_num_factors(const pps_graph* g, int* n);
int pps_get_pose(pps_graph* g, int id, double tq[7]);
int pps_get_plane(pps_graph* g, int id, double abcd[4]);
int pps_set_pose(pps_graph* g, int id, const double tq[7]);     /* NodeT::init on an existing node */
int pps_set_plane(pps_graph* g, int id, const double abcd[4]);
/* bulk: ids may be NULL (= all poses / all planes in insertion order); out is n x 7 / n x 4 */
int pps_get_poses(pps_graph* g, int n, const int* ids, double* out);
int pps_get_planes(pps_graph* g, int n, const int* ids, double* out);
/* Device-resident snapshot of the current estimate and its restore (bench / what-if solves):
 * the analogue of copying every NodeT::_value aside and back (Node.h:104-146). */
int pps_save_state(pps_graph* g);
int pps_restore_state(pps_graph* g);

/* ---- introspection (tests, bench, INTEGRATION) ---------------------------------------- */
typedef struct pps_stats {
  int    n_poses, n_planes, n_factors;
  int    dim_nodes, dim_measure;           /* Slam::_dim_nodes / _dim_measure               */
  int    n_fronts, n_levels, max_front;    /* multifrontal elimination tree of the last analysis */
  int64_t nnz_L;                           /* scalars stored in the factor panels           */
  int    lm_iterations;                    /* of the last batch_optimize                    */
  int    lm_trials_accepted, lm_trials_rejected;
  double chi2_initial, chi2_final, lambda_final;
  double last_delta_norm;
  /* wall-clock seconds of the last solve call, host side */
  double t_total, t_analysis, t_upload;
  /* device time (HIP events) of the last solve call, seconds, summed over launches */
  double t_linearize, t_assemble, t_factor, t_backsolve, t_retract_chi2;
  int    n_linearize, n_factorize;         /* launches of the sweep / factorizations        */
  int    lm_trials_notpd;                  /* LM trials whose factorisation hit a non-positive pivot (the step is then
                                              rejected like any other bad step; PPS_ENOTPD only if the last trial did) */
  int    n_launches;                       /* kernel launches of the last solve call (all streams)                    */
} pps_stats;
int pps_get_stats(const pps_graph* g, pps_stats* out);
/* LM trace of the last batch_optimize: per trial (lambda, chi2_new, accepted); returns count via n */
int pps_get_trace(const pps_graph* g, int cap, double* lambda, double* chi2, int* accepted, int* n);
/* HIP-event timing: 0 = off (default); 1 = event pairs around the Jacobian sweep only, no host syncs
 * (t_linearize / n_linearize = mean launch duration); 2 = every phase, adds stream syncs. */
int pps_set_profiling(pps_graph* g, int level);

/* Per-factor residual / Jacobian of the device sweep, for parity tests.
 * sel: 0 = linearisation point, 1 = estimate.  J is (dim x cols) row-major, r is the whitened residual. */
int pps_factor_shape(const pps_graph* g, int fid, int* dim, int* cols);
int pps_eval_factor(pps_graph* g, int fid, int mode /* enum pps_jacobian_mode */, double* J, double* r);

/* Host-side symbolic analysis only (no device needed): runs ordering + front construction. */
int pps_analyze(pps_graph* g);
/* Frame loops: a graph that only GROWS between two analyses (nodes / factors appended, every new factor touching a new node)
 * is analysed incrementally -- the part of the elimination tree left of the new poses, with all of its index arrays, is
 * kept (csrc/pps_symbolic.h).  fronts_kept of fronts_total of the last analysis were taken over (0 = from scratch). */
int pps_analysis_reuse(const pps_graph* g, int* fronts_kept, int* fronts_total);
/* What the last analysis left exactly as the analysis before it had it: leading entries of its index arrays (csrc/pps_symbolic.h,
 * Analysis::Kept) -- kept[6] = fronts, fronts_lists, blocks, segs, contribs, nd_segs.  The topology upload of a frame loop does not
 * compare those parts with its mirror again; tests/test_host_incremental.py checks the claim array by array. */
int pps_analysis_kept(const pps_graph* g, int kept[6]);
/* Flat dump of the analysis for host-logic tests.  Call with out == NULL to get the needed length. */
int pps_analysis_dump(pps_graph* g, int64_t cap, int32_t* out, int64_t* needed);

/* ---- K1 micro-benchmark entry: the Jacobian sweep over a batch of replicated graphs --- */
/* Replicates the handle's plane-observation and odometry edges `replicas` times in device
 * memory (state shared), runs `iters` sweeps and returns mean kernel times (HIP events, seconds):
 * sec[0] = both launches, sec[1] = plane-edge launch alone, sec[2] = odometry launch alone; plus the
 * number of plane / odometry edges per sweep.  Used for the HBM roofline figure.  mode: PPS_JAC_NUMERIC / PPS_JAC_ANALYTIC = the
 * thread-per-factor kernels, 2 = the lane-parallel numeric form (19 lanes per plane observation; what a graph below 200 000
 * factors runs). */
/* K1 of the handle's own graph, `iters` back-to-back launches on the solver's stream between two HIP events */
int pps_time_linearize(pps_graph* g, int mode, int iters, double* sec_per_launch);
int pps_bench_sweep(pps_graph* g, int mode, int replicas, int iters, double sec_per_sweep[3],
                    int64_t* n_plane_edges, int64_t* n_odo_edges);

/* ---- K3 diagnostic: one frontal matrix through the register-tile elimination, outside any graph ---- */
/* The dense partial Cholesky a front of the multifrontal factorisation goes through (the part of Cholesky.cpp's factorisation,
 * isam/Cholesky.cpp:86-130 via CHOLMOD's supernodal kernel, that happens inside one supernode), with the right-hand side as the
 * front's last row.  A: packed lower triangle, p + b + 1 rows (pivot rows, boundary rows, rhs row; row i holds i + 1 values).
 * L: (p + b + 1) x p row-major factor panel [L_A; L_B; y^T] (entries above the diagonal of L_A unspecified); U: packed lower
 * triangle of the (b + 1)-row update matrix [S; r^T].  tiles: 16-row tile rows held in registers, 2 .. 5, or 0 for what the solver
 * picks for a front of this size; strip != 0: rows 64 .. 79 as a strip of the LDS triangle under four tile rows (fronts of 65 .. 80
 * rows only).  not_pd: 1.0 when a pivot was not positive.  p <= 64, p + b + 1 <= 80.  Runs on the current device. */
int pps_debug_front_factor(int tiles, int strip, int p, int b, const double* A, double* L, double* U, double* not_pd);

/* ---- K4 diagnostic: the retraction of n nodes outside any graph ---- */
/* kind 0: Pose3d::exmap (Pose3d.h:131-136: t += d[0:3], q <- q * Exp(d[3:6])); x n x 7 (tx ty tz qx qy qz qw), delta n x 6, out n x 7.
 * kind 1: Plane3d::exmap_3dof (isam_plane3d.h:101-127: Exp(d) * q, then normalised); x n x 4, delta n x 3, out n x 4.  The device
 * functions every retraction and every numerical-difference step of the solver goes through.  Runs on the current device. */
int pps_debug_exmap(int kind, int n, const double* x, const double* delta, double* out);

/* ---- pop-up (fp32), /root/reference/pop_up_wall --------------------------------------- */
/* popup_plane::update_plane_equation_from_seg (libs/popup_plane.cpp:654-705).
 * seg2d n x 4 (u1,v1,u2,v2); invK 3x3, T_wc 4x4 row-major; planes_out (n+1) x 4, row 0 = ground.
 * Host pointers; runs on `device`. */
int pps_popup_planes(int device, const float* seg2d, int n, const float invK[9], const float T_wc[16],
                     float* planes_out);

/* One pop-up point: world xyz + packed colour, 16 bytes (pcl::PointXYZRGB carries the same payload in
 * 32 B: popup_plane.cpp:943-983).  rgba = valid<<24 | r<<16 | g<<8 | b. */
typedef struct pps_point { float x, y, z; uint32_t rgba; } pps_point;

typedef struct pps_popup pps_popup;   /* per-camera context: device buffers for one image size */
int pps_popup_create(int device, int width, int height, const float invK[9], pps_popup** out);
int pps_popup_destroy(pps_popup* p);
const char* pps_popup_last_error(const pps_popup* p);
/* upload a BGR image (u8, width*height*3); stays resident for the following runs.  NULL = no colour. */
int pps_popup_set_image(pps_popup* p, const unsigned char* bgr);
/* Fused frame kernel, one launch: K5 segments -> plane equations (get_plane_equation,
 * popup_plane.cpp:551-603) and K6 pixels -> 3-D (generate_cloud + matrixToCloud :807-863,925-985;
 * get_depth_map_good :866-921).  Results stay on the device until pps_popup_download.
 *   seg2d    n x 4 ground segments; plane i >= 1 comes from segment i-1, plane 0 = ground
 *   polys    closed 2-D polygons (x,y) of the planes to pop up (all_closed_2d_bound_polygons,
 *            popup_plane.h:70-76), poly_off[nplanes+1] vertex offsets; an empty polygon skips the plane.
 *            The pixels of a polygon are the ones popup_plane::closed_polygons_homo_pts yields (popup_plane.cpp:81-116):
 *            vertices truncated to integers like cv::Point(float, float), shifted into their bounding box, rasterised
 *            with cv::fillConvexPoly's rules (8-connected Bresenham outline + 16.16 fixed-point scanline spans) --
 *            bit for bit, for any vertex list (|coordinate| < 2^15), convex or not.  Planes are written in order, so
 *            a pixel keeps the LAST polygon that covers it (generate_cloud / get_depth_map_good, :820-850, :893-911).
 *   step     1 = every pixel; 2 = downsample_poly (:86-87,104-108): the polygon is halved BEFORE the truncation,
 *            rasterised at half size and the pixel coordinates doubled -- even pixels only
 * Filters (matrixToCloud :948-960): z_s < 0, z_s > depth_thre, z_w < -0.2 dropped; z_w clamped to
 * ceiling_thre.  Depth: z_s, with the ceiling plane substituted above ceiling_thre (:903-916). */
int pps_popup_run(pps_popup* p, const float* seg2d, int n, const float T_wc[16], const float* polys,
                  const int* poly_off, int nplanes, int step, float depth_thre, float ceiling_thre, int* n_valid);
/* The same run without waiting for it (the frame loop, Mapping.cpp:401-586 / main_3d.cpp:423-503: the graph construction needs the plane
 * equations, the pixels are nobody's input before the frame is drawn).  pps_popup_planes_wait returns the (n+1) x 4 plane equations as soon as
 * the kernel's first workgroup has written them -- a few microseconds after the launch --, pps_popup_wait the end of the run (n_valid may be
 * NULL).  Every entry point that reads results of a run (download, plane_info, fill_depth, download_segments3d, the next run) waits for a
 * run in flight by itself. */
int pps_popup_run_async(pps_popup* p, const float* seg2d, int n, const float T_wc[16], const float* polys,
                        const int* poly_off, int nplanes, int step, float depth_thre, float ceiling_thre);
int pps_popup_planes_wait(pps_popup* p, float* planes);
int pps_popup_wait(pps_popup* p, int* n_valid);
/* any output may be NULL: planes (n+1)x4 sensor-frame, cloud width*height pps_point, depth width*height,
 * plane_id width*height (-1 = none) */
int pps_popup_download(pps_popup* p, float* planes, pps_point* cloud, float* depth, int32_t* plane_id);
/* switch the optional per-pixel outputs of pps_popup_run off / on (default: both on).  The cloud alone is 16 B per pixel
 * (what generate_cloud produces); the depth map (get_depth_map_good) and the plane-id map add 4 B each. */
int pps_popup_set_outputs(pps_popup* p, int want_depth, int want_plane_id);
/* The rest of get_plane_equation's per-plane outputs for the last run (popup_plane.cpp:616-640), n+1 entries, plane 0 = ground:
 *   dist_to_cam  all_plane_dist_to_cam: camera height for the ground, distance from the camera footprint to the world
 *                ground segment for a wall (what the mapper's sigma model reads, Mapping.cpp:507-512)
 *   good         1 for the ground and for every wall whose two ground points lie in front of the camera, closer than
 *                plane_cam_dist_thre (popup_plane.h:86: 10) and -- if actual_plane_indices is given (n_actual > 0; plane
 *                indices >= 1, e.g. open_in_closed + 1 of pps_edges_select) -- not a manually connected edge: good_plane_indices
 * Either output may be NULL. */
int pps_popup_plane_info(pps_popup* p, float plane_cam_dist_thre, const int* actual_plane_indices, int n_actual,
                         float* dist_to_cam, int32_t* good);
/* Tail of get_depth_map_good for the half-resolution pop-up (popup_plane.cpp:913-917, as main_3d.cpp:454 calls it): after
 * a pps_popup_run with step = 2 the depth map is defined on the even pixels only; this spreads it over the full frame the
 * way the reference does (resize 0.5, x 4, resize 2: bilinear from the half-size map).  Even image sizes only. */
int pps_popup_fill_depth(pps_popup* p);
/* ground_seg3d_lines_world of the last run (popup_plane.cpp:569-578): n x 6 = (x0,y0,0,x1,y1,0), the
 * world-frame ground end points of every segment -- columns 0,1 of all_3d_bound_polygons_world (:494-495),
 * which data association compares (Mapping.cpp:355-360). */
int pps_popup_download_segments3d(pps_popup* p, float* seg3d_world);
/* device time of the last pps_popup_run kernel (HIP events), seconds */
int pps_popup_last_kernel_time(const pps_popup* p, double* sec);
/* popup_plane::find_2d_3d_closed_polygon_simplemode (libs/popup_plane.cpp:409-500; simple_polygon_mode, popup_plane.h): the
 * closed 2-D polygon of every wall plane of a frame, from the ground / wall boundary segments pps_edges_select returns
 * (closed_segs) and the camera pose -- what pps_popup_run takes as `polys`.  walllength_threshold <= 0 (the class default,
 * popup_plane.h:81); the wall-length cut of :502-546 relies on cv::intersectConvexConvex and is not offered.  Host code (a
 * handful of fp32 operations per segment).  K / invK: the calibration and its inverse (popup_plane::set_calibration, :72-76).
 *   verts      (x, y) pairs, cap_verts >= 8 * n;  poly_off[n + 2]: vertex offsets of planes 0 .. n; plane 0 (ground) is empty
 *   a wall whose vertical lines do not reach the image boundary gets no polygon (:480-481) */
int pps_popup_polygons_simple(const float K[9], const float invK[9], const float T_wc[16], int width, int height, const float* seg2d, int n,
                              float* verts, int cap_verts, int* poly_off, int* n_verts);
/* The polygon -> pixel-set rules of pps_popup_run (closed_polygons_homo_pts / cv::fillConvexPoly, popup_plane.cpp:81-116)
 * evaluated on the host by the same interval code the kernel runs (no device needed; used by the CPU tests):
 * plane_id width*height, -1 = none. */
int pps_popup_mask_host(const float* polys, const int* poly_off, int nplanes, int width, int height, int step, int32_t* plane_id);

/* ---- pop-up feeding the graph: Mapper_mono::update_plane_measurement (Mapping.cpp:590-607) ------
 * Frames register their 2-D ground segments once; pps_refresh_measurements then re-derives every
 * registered edge's measurement from the CURRENT pose estimate, on the device, and writes it straight
 * into the edge array the Jacobian sweep reads (fp32 pop-up math, cast to fp64 and normalised like
 * Plane3d(Vector4d)).  fids has n_seg+1 entries (plane 0 = ground); -1 skips a plane. */
int pps_frames_set_calibration(pps_graph* g, const float invK[9]);
int pps_frames_add(pps_graph* g, int pose_id, int n_seg, const float* seg2d, const int* fids, int* frame_id);
int pps_refresh_measurements(pps_graph* g);
/* read back a plane factor's current measurement (FactorT::measurement(), Factor.h:203) */
int pps_get_measurement(pps_graph* g, int fid, double meas4[4]);

/* ---- plane data association: Mapper_mono::findClosestPlane (src/Mapping.cpp:256-397) ------------
 * The handle keeps one record per landmark (what findClosestPlane reads of a Map_plane, Map_plane.h:28-44);
 * the landmark's plane itself is read from the solver state on the device.  Landmark order (= tie-break
 * order, = all_landmarks index) is the order of first registration. */
typedef struct pps_assoc_params {   /* Mapping.h:70-77; tum yaml: 10000, 2, -1, 35, 1000 */
  double edge_asso_2ddist;          /* 50   mean 2-D end-point distance gate [px] */
  double edge_asso_planedist;       /* 4    plane distance gate [m] */
  double edge_asso_proj;            /* 0.5  minimum mutual 1-D overlap of the ground segments */
  double edge_asso_angle;           /* 60   normal angle gate [deg] */
  int assoc_near_frames;            /* 5    only landmarks seen within this many frames */
} pps_assoc_params;
void pps_assoc_default_params(pps_assoc_params* p);
/* copy_plane (Map_plane.cpp:11-22) after an observation: (re)write the landmark's latest frame
 * properties; the first call for a plane id registers it (all_landmarks.push_back, Mapping.cpp:485-488).
 * seg2d = ground edge end points in the image (u0,v0,u1,v1); seg3d_xy = their world x,y (x0,y0,x1,y1);
 * both may be NULL for the ground plane. */
int pps_landmark_update(pps_graph* g, int plane_id, int frame_plane_indice, int frame_seq_id, const float seg2d[4],
                        const float seg3d_xy[4]);
int pps_landmark_set_merged(pps_graph* g, int plane_id);      /* deteted_by_merge = true (Mapping.cpp:697) */
/* n query planes of one frame against all landmarks, one launch.  planes_local n x 4 (sensor frame,
 * Map_plane::temp_value), est_pose = latest pose (+) odometry (Mapping.cpp:413-416).
 * best_plane_id[i] = plane node id of the match or -1; best_err[i] = its score (-1: none / ground). */
int pps_find_closest_planes(pps_graph* g, const double est_pose[7], int frame_seq_id, int n, const double* planes_local,
                            const int* frame_plane_indice, const float* seg2d, const float* seg3d_xy,
                            const pps_assoc_params* prm, int* best_plane_id, double* best_err);

/* ---- graph text format: Slam::save (isamlib/Slam.cpp:84-89) / Graph::write (include/isam/Graph.h:120-131) ----
 * One line per factor, then one line per node, in insertion order:
 *   Pose3d_Pose3d_Factor 3 4 (x, y, z; yaw, pitch, roll) {s11,s12,...,s66}
 *   Pose3d_Plane3d_Factor 4 17 (a, b, c; d) {s11,s12,s13,s22,s23,s33}
 *   Pose3d_Factor 0 (...) {...}            pose prior AND plane prior (name quirk, isam_plane3d.h:438)
 *   Pose3d_Node 4 (x, y, z; yaw, pitch, roll)     Plane3d_Node 17 (a, b, c; d)
 * precision <= 0: 6 significant digits like the reference's ostream default (lossy); 17 round-trips doubles.
 * pps_graph_load reads this format back into a new handle (the reference has no reader for it); node ids are
 * re-assigned densely in file order.  A Factor2 edge (pps_add_plane_obs2) prints like the reference's -- same name,
 * stored measurement (isam_plane3d.h:327) -- so its ground-edge rays are not part of the file and it reloads as a
 * plain observation. */
int pps_graph_save(pps_graph* g, const char* path, int precision);
int pps_graph_load(const char* path, const pps_props* props, pps_graph** out);

/* Mapper_mono::reproj_to_newplane (src/Mapping.cpp:609-632): stored polygon vertices (fp32 world points) projected onto
 * the CURRENT estimate of their landmark plane -- Plane3d::project_to_plane (src/isam_plane3d.h:173-178) in fp64, result
 * cast back to fp32.  plane_ids[i] is the plane node of point i; points of removed (merged) landmarks are copied through. */
int pps_reproject_points(pps_graph* g, int n, const int* plane_ids, const float* pts_xyz, float* out_xyz);

/* ---- ground-edge selection: popup_plane::edge_get_polygons (pop_up_wall/libs/select_edge.cpp:66-409) -------------
 * The step before the pop-up: the CNN label map (u8, ground = 255) and the raw LSD line segments of a frame become
 * the ground / wall boundary polyline pps_popup_run and pps_frames_add take.  Per-pixel work runs on the device:
 * [half-size nearest resize], dilate, erode and inversion of the label map in one kernel (select_edge.cpp:69-78),
 * then the marching-squares cells of skimage.measure.find_contours(label, 0) in raster order (the reference calls it
 * through boost::python, pop_up_fun.py:85-106).  The contour linking and the segment selection (steps 1-6 of
 * select_edge.cpp and interval_tree_optimization, pop_up_fun.py:109-204) are sequential work on a few hundred
 * segments and run on the host inside the same call.  LSD detection itself (line_lbd, OpenCV) is not part of this
 * library: lsd_lines are an input, as they are for edge_get_polygons. */
typedef struct pps_edge_params {
  int downsample_contour;                    /* popup_plane.h:82 (false) */
  int dilation_distance, erosion_distance;   /* popup_plane.cpp:32-33 (11, 11) */
  /* popup_plane.h:184-192: 15 15 50 20 30 10 20 10 15 */
  double pre_vertical_thre, pre_minium_len, pre_contour_close_thre, interval_overlap_thre, post_short_thre,
         post_bind_dist_thre, post_merge_dist_thre, post_merge_angle_thre, post_extend_thre;
  /* popup_plane.h:194-200: 5 10 10 20 0.6 0.8 100 */
  double pre_boundary_thre, pre_merge_angle_thre, pre_merge_dist_thre, pre_proj_angle_thre, pre_proj_cover_thre,
         pre_proj_cover_large_thre, pre_proj_dist_thre;
} pps_edge_params;
void pps_edge_default_params(pps_edge_params* p);

typedef struct pps_edges pps_edges;   /* per-camera context: device buffers for one label-map size */
int pps_edges_create(int device, int width, int height, pps_edges** out);
int pps_edges_destroy(pps_edges* e);
const char* pps_edges_last_error(const pps_edges* e);
/* label_map: width*height u8, a host pointer or (label_on_device != 0) a device pointer on the context's device (the
 * kernels run on the context's own stream: the producer of a device-resident map must have finished, e.g. by an event
 * or stream synchronisation on the caller's side).
 * lsd_lines n_lines x 4 (x1 y1 x2 y2), host.  Outputs (host, caller-allocated, 2*n_lines+2 rows each):
 *   open_segs       n_open x 4    ground_seg2d_lines_actual
 *   closed_segs     n_closed x 4  ground_seg2d_lines_connect (connecting pieces inserted)
 *   open_in_closed  n_open        row of each open segment in closed_segs (actual_walls_in_closepoly_ind)
 * No boundary in the label map / no line survives: n_open = n_closed = 0, PPS_OK (the reference prints
 * "cannot find ground edges"). */
int pps_edges_select(pps_edges* e, const unsigned char* label_map, int label_on_device, const float* lsd_lines,
                     int n_lines, const pps_edge_params* prm, float* open_segs, int* n_open, float* closed_segs,
                     int* n_closed, float* open_in_closed);
/* intermediate results of the last pps_edges_select, for tests and debugging: the pre-processed label map
 * (w x h bytes, ground = 0), the sub-sampled ground contour as (x, y) pairs, the number of contours found and the
 * length of the chosen one */
int pps_edges_download_label(pps_edges* e, unsigned char* out, int* w, int* h);
int pps_edges_contour(pps_edges* e, float* xy, int cap, int* n, int* n_contours, int* n_points);
/* device time of the kernels of the last pps_edges_select (HIP events), seconds */
int pps_edges_last_kernel_time(const pps_edges* e, double* sec);
/* The two host stages of pps_edges_select on their own (no device needed; used by the CPU tests):
 * cell segments (n x 4 int16: from row, from column, to row, to column, in raster order of the cells) -> sub-sampled
 * ground contour; contour + LSD lines -> selection. */
int pps_edges_host_contour(const int16_t* cell_segs, int n, float scale, float* xy, int cap, int* n_xy, int* n_contours,
                           int* n_points);
int pps_edges_host_select(const float* contour_xy, int n_contour, int width, int height, const float* lsd_lines, int n_lines,
                          const pps_edge_params* prm, float* open_segs, int* n_open, float* closed_segs, int* n_closed,
                          float* open_in_closed);

#ifdef __cplusplus
}
#endif
#endif /* PPS_H */
