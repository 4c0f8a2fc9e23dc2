/*
 * mml_oracle.h -- CPU restatement ("oracle") of the mm-loam scan-registration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (multi-modal-loam_amd/) may
 * include, link or call this library; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (TIERS/multi-modal-loam) ships no tests,
 * golden vectors or fixtures and cannot be built here (needs ROS melodic, PCL/FLANN, Eigen,
 * Ceres 2.1.0).  This restatement follows the in-tree arithmetic line by line (citations
 * next to every function, paths relative to /root/reference/mm-loam/) and restates the
 * published algorithms of the absent third-party pieces:
 *   - Eigen 3.3.4 (ROS melodic): fixed-size redux order, Quaternion(matrix), slerp,
 *     SelfAdjointEigenSolver<Matrix3d>::compute, ColPivHouseholderQR 5x3, LLT
 *   - FLANN (PCL 1.8 KdTreeFLANN, L2_Simple<float>): exact 5-NN, float distances
 *   - pcl::VoxelGrid<PointXYZINormal>::applyFilter (centroid per occupied voxel)
 *   - Ceres 2.1.0 trust_region_minimizer.cc / dogleg_strategy.cc / corrector.cc /
 *     loss_function.cc (TRADITIONAL_DOGLEG, Jacobi scaling, HuberLoss)
 * It is cross-checked in tests/ against independent numpy/scipy implementations
 * (brute-force kNN, numpy.linalg.eigh / lstsq, finite-difference Jacobians).
 * ONE piece IS pinned against a binary the reference links: glibc's atanf / atan2f (the float
 * overloads unionFeatureExtract.cpp:1136-1139,1159,1168 resolve to), restated in libm_f32.h and
 * compared with this image's libm on all 2^32 / 4e8 arguments (tests/test_oracle.py).
 */
#ifndef MML_ORACLE_H
#define MML_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Livox CustomPoint record, 20 bytes (livox_ros_driver/CustomPoint.msg as laid out in C++). */
typedef struct {
    uint32_t offset_time;
    float x, y, z;
    uint8_t reflectivity, tag, line, _pad;
} mmlo_livox_point;

/* ---- a3..a8: unionFeatureExtract.cpp:341-844 ------------------------------------------ */
/* pts: n x (x,y,z,intensity) float, all finite.  sharp/flat: capacity n each.
 * flags_out (optional, n ints): final CloudFeatureFlag[] (zero-initialised convention). */
void mmlo_detect_feature_points(const float* pts, int n, int* sharp, int* n_sharp, int* flat,
                                int* n_flat, int* flags_out);

/* ---- a1 + a3..a8 + a8 crop: getVeloFeature, unionFeatureExtract.cpp:1113-1317 ---------- */
/* in: n x (x,y,z,intensity), NaN rows are removed first (pcl::removeNaNFromPointCloud).
 * out_* have capacity n.  out_xyzi: cropped combined cloud (intensity zeroed, :1254-1256),
 * out_reltime = normal_x, out_ring = normal_y, out_label = normal_z (0/1/2).
 * Returns number of points in the cropped combined cloud; n_corner/n_surf = sizes of the
 * cropped corner / surf clouds (velo_corner_num / velo_surf_num, :1299-1300).
 * n_rings generalises VELO_N_SCANS (=16, :192): ring = int((deg - pitch0)/pitch_step + 0.5);
 * the reference is n_rings=16, pitch0=-15, pitch_step=2 (:1162). */
int mmlo_extract_velo(const float* in_xyzi, int n, int n_rings, float pitch0_deg, float pitch_step_deg,
                      float near_th, float far_th, float* out_xyzi, float* out_reltime,
                      int* out_ring, int* out_label, int* n_corner, int* n_surf);

/* ---- a2 + a3..a8 + crop: getHoriFeature/getHoriFeatureExtract, :891-1035 ---------------- */
int mmlo_extract_livox(const mmlo_livox_point* in, int n, int n_lines, float near_th, float far_th,
                       float* out_xyzi, float* out_reltime, int* out_line, int* out_label,
                       int* n_corner, int* n_surf);

/* ---- a9: RemoveLidarDistortion, unionPoseEstimation.cpp:402-421 -------------------------- */
/* xyz: n x 3 float in place; s: n float (normal_x); dR row-major 3x3, dt 3. */
void mmlo_undistort(float* xyz, const float* s, int n, const double* dR, const double* dt);

/* ---- a10: pcl::VoxelGrid centroid down-sample (Estimator.cpp:1013-1024) ----------------- */
/* in: n x 3 float; out capacity n x 3; returns number of voxels (output ordered by voxel idx,
 * within-voxel summation in input order: convention, std::sort there is unstable). */
int mmlo_voxel_downsample(const float* xyz, int n, float leaf, float* out_xyz);

/* ---- a13: exact 5-NN (FLANN L2_Simple<float> semantics), kd-tree ------------------------ */
typedef struct mmlo_kdtree mmlo_kdtree;
mmlo_kdtree* mmlo_kdtree_build(const float* xyz, int m);
void mmlo_kdtree_free(mmlo_kdtree*);
/* idx[5], d2[5] ascending (ties: lower index first).  Needs m >= 5. */
void mmlo_kdtree_knn5(const mmlo_kdtree*, const float* q, int* idx, float* d2);
void mmlo_bruteforce_knn5(const float* xyz, int m, const float* q, int* idx, float* d2);

/* ---- a11..a16: association + model fit --------------------------------------------------- */
typedef struct {
    double point_ori[3]; /* feature in lidar frame (float values) */
    double p1[3], p2[3]; /* line tripod, Estimator.cpp:256-264 */
    double error;        /* FeatureLine::ComputeError, Estimator.h:71-83 */
} mmlo_line_factor;

typedef struct {
    double point_ori[3];
    double point_proj[3]; /* Estimator.cpp:674 */
    double omega[3];      /* unit normal (float values), :672 */
    double error;         /* FeaturePlanVec::ComputeError, Estimator.h:118-121 */
} mmlo_plane_factor;

/* Local-map association only (laserCloud{Corner,Surf}FromLocal, Estimator.cpp:283-361 /
 * :702-767); the global cube path is mmlo_associate_*_cubes.  T_wl: row-major 4x4 (m4d).
 * Returns number of factors written (capacity n_feat). */
int mmlo_associate_lines(const float* feat_xyz, int n_feat, const float* map_xyz, int m,
                         const mmlo_kdtree* tree, const double* T_wl, double thres_dist,
                         mmlo_line_factor* out, int* out_src /*optional: feature index*/);
int mmlo_associate_planes(const float* feat_xyz, int n_feat, const float* map_xyz, int m,
                          const mmlo_kdtree* tree, const double* T_wl, double thres_dist,
                          mmlo_plane_factor* out, int* out_src);
/* a12 + two-level association: the global map of 21x11x21 cubes of 50 m (Map_Manager.h:117-146) as Estimate() copies
 * it (Estimator.cpp:1170-1184).  cube[i]: ToIndex value of map point i; cen: laserCloudCen{Width,Height,Depth}_last.
 * Features whose cube index is 5000 (outside the grid) are skipped altogether, as in the reference (:194, :623). */
typedef struct mmlo_cube_map mmlo_cube_map;
mmlo_cube_map* mmlo_cube_map_build(const float* xyz, const int* cube, int m, const int* cen);
void mmlo_cube_map_free(mmlo_cube_map*);
int mmlo_find_used_map(const float* p_xyz, const int* cen); /* Map_Manager.cpp:583-629 */
int mmlo_associate_lines2(const float* feat_xyz, int n_feat, const mmlo_cube_map* gmap, const float* map_xyz, int m,
                          const mmlo_kdtree* tree, const double* T_wl, double thres_dist, mmlo_line_factor* out,
                          int* out_src, int* from_global);
int mmlo_associate_planes2(const float* feat_xyz, int n_feat, const mmlo_cube_map* gmap, const float* map_xyz, int m,
                           const mmlo_kdtree* tree, const double* T_wl, double thres_dist, mmlo_plane_factor* out,
                           int* out_src, int* from_global);
/* Section 8(f): Estimator::MapIncrementLocal (Estimator.cpp:1585-1643) after the clear() at :1083-1085 / :1125-1127:
 * ring of `window` (50) key scans' features moved to the world frame, local map = VoxelGrid(concatenation). */
typedef struct mmlo_local_map mmlo_local_map;
mmlo_local_map* mmlo_local_map_create(int window, float leaf_corner, float leaf_surf);
void mmlo_local_map_free(mmlo_local_map*);
void mmlo_local_map_increment(mmlo_local_map*, const float* corner_xyz, int n_corner, const float* surf_xyz, int n_surf,
                              const double* T_wl);
int mmlo_local_map_size(const mmlo_local_map*, int kind);
void mmlo_local_map_get(const mmlo_local_map*, int kind, float* out_xyz);
/* Section 8(f): MAP_MANAGER::MapIncrement + MapMove (Map_Manager.cpp:125-581), corner and surf cube stores. */
typedef struct mmlo_cube_store mmlo_cube_store;
mmlo_cube_store* mmlo_cube_store_create(float leaf_corner, float leaf_surf);
void mmlo_cube_store_free(mmlo_cube_store*);
void mmlo_cube_store_increment(mmlo_cube_store*, const float* corner_world, int n_corner, const float* surf_world, int n_surf,
                               const double* T_wl);
int mmlo_cube_store_get(const mmlo_cube_store*, int which, int kind, float* xyz, int* cube, int* cen);
/* checkLocalizability, Estimator.cpp:536-565: smallest singular value of the M x 3 normal
 * matrix (= sqrt(lambda_min(N^T N))); -1 when M <= 10. */
double mmlo_check_localizability(const mmlo_plane_factor* f, int n);

/* ---- a17..a19: residuals + analytic Jacobians (ceresfunc.h:397-458, 517-570) ------------- */
/* x = [t(3), phi(3)]; T_bl row-major 4x4 (inverse extrinsic, Estimator.cpp:157-159).
 * r: 1 residual, J: 1x6.  sqrt_info = 1/lidar_m. */
void mmlo_line_residual(const mmlo_line_factor* f, const double* x, const double* T_bl,
                        double* r, double* J /*nullable*/);
/* r: 3 residuals in the {omega, tangent} basis-independent form is not possible for a
 * vector residual, so this returns the reference's rows for sqrt_info = diag(1,w,w)/lidar_m *
 * Rsvd^T with a deterministic basis; use mmlo_plane_sqnorm for basis-free checks. */
void mmlo_plane_residual(const mmlo_plane_factor* f, const double* x, const double* T_bl,
                         double plan_weight_tan, double* r /*3*/, double* J /*3x6 nullable*/);

/* ---- a20: normal equations for one frame -------------------------------------------------- */
/* H: 6x6 row-major (J^T J with Ceres' Huber correction, corrector.cc), g = J^T r, cost =
 * 0.5 * sum rho(|r|^2).  huber_delta <= 0: no loss.  Factors with |error| <= 1e-5 are skipped
 * (Estimator.cpp:1313,1325,1385,1396). */
void mmlo_linearize(const mmlo_line_factor* lf, int n_line, const mmlo_plane_factor* pf, int n_plane,
                    const double* x, const double* T_bl, double plan_weight_tan, double huber_delta,
                    double* H, double* g, double* cost);

/* ---- a20: Ceres 2.1.0 trust-region (TRADITIONAL_DOGLEG) restatement, W frames, lidar factors only */
typedef struct {
    int max_num_iterations;   /* 10, Estimator.cpp:1428 */
    int fixed_iterations;     /* !=0: disable the convergence tests (benchmark mode) */
    double huber_delta;       /* 0.1/lidar_m in 1-frame mode (:1216), <=0 none */
    double plan_weight_tan;   /* :1203,1206 */
} mmlo_solve_opts;
typedef struct {
    int iterations;           /* iterations run (successful + unsuccessful) */
    int successful;
    double initial_cost, final_cost;
    int termination;          /* 0 no-convergence(max iters) 1 gradient 2 parameter 3 function */
} mmlo_solve_summary;
/* x: 6*W in/out. lf/pf arrays are concatenated per frame with counts n_line[f], n_plane[f].
 * trace (optional): 6*W doubles per iteration (x after each iteration), capacity max_num_iterations. */
void mmlo_solve_window(const mmlo_line_factor* lf, const int* n_line, const mmlo_plane_factor* pf,
                       const int* n_plane, int W, const double* T_bl, const mmlo_solve_opts* opts,
                       double* x, mmlo_solve_summary* summary, double* trace);

/* ---- a21: Estimator::Estimate outer loop, 1-frame mode (Estimator.cpp:1143-1581) ---------- */
/* corner_feat / surf_feat: down-sampled stacks (lidar frame).  Local maps only.
 * P,Q(x,y,z,w): body pose in/out.  exTlb row-major 4x4.  Returns number of outer iterations.
 * outer_trace (optional): 7 doubles (P,Q) after each outer iteration, capacity 5. */
int mmlo_estimate_single(const float* corner_feat, int n_corner, const float* surf_feat, int n_surf,
                         const float* corner_map, int m_corner, const float* surf_map, int m_surf,
                         const double* exTlb, double* P, double* Q, int max_outer, int inner_iters,
                         int* is_degenerate, double* outer_trace);

/* ---- the per-frame GICP extrinsic refresh, icp_ext_matching (unionFeatureExtract.cpp:74-123) -------
 * pcl::GeneralizedIterativeClosestPoint as configured there (10 iterations, transformation epsilon 1e-6, PCL defaults
 * otherwise), restated from the published PCL 1.8.1 algorithm (gicp.hpp + bfgs.h); PARITY UNPINNED (PCL is not in the
 * reference tree).  src / tgt: n x 3 floats (livox surf cloud -> velodyne surf cloud).  T_inout: row-major 4 x 4 float,
 * overwritten with getFinalTransformation() only when the alignment converged (return 1); 0 = "ICP Failed". */
int mmlo_gicp_align(const float* src, int n_src, const float* tgt, int n_tgt, float* T_inout, int* outer_iterations,
                    int* bfgs_evaluations, double* last_objective);
/* pieces of it for the independent checks: objective (1/m) sum res^T M res and its gradient at x = (t, roll, pitch, yaw)
 * for given correspondences (maha9: n_src x 9, indexed by source point), and the regularised covariances of a cloud */
double mmlo_gicp_objective(const float* src, const float* tgt, const int* idx_src, const int* idx_tgt, int m, const double* maha9,
                           int n_src, const double* x, double* grad);
void mmlo_gicp_covariances(const float* pts, int n, double* out9);

/* Threading of the calling host thread (defaults 1, 1 = the bit-reference configuration of every parity test):
 * livox_line_threads: detectFeaturePoints of the Livox lines in parallel (unionFeatureExtract.cpp:1008-1015, 6 there);
 * solve_threads: residual blocks of mmlo_linearize / mmlo_solve_window in parallel (ceres num_threads, Estimator.cpp:1430). */
void mmlo_set_threading(int livox_line_threads, int solve_threads);

/* helpers exposed for tests */
/* unionLidarsAligner.cpp:1077-1153 (estimate_timeoffset, the numeric core): the Velodyne cloud goes through
 * pcl::transformPointCloud with tf (row-major 4x4, NULL = identity, :1080-1082), every Livox point gets the squared
 * distance to its nearest transformed Velodyne point (:1084-1103), and window cnt sums
 * dis_errors[i] + 0.2 * sqrt(x_i^2 + y_i^2) over i in [cnt*res, cnt*res + sliced) while cnt*res + sliced < n (:1111-1131).
 * nn_d2: n_livox floats; win_err: capacity doubles.  Returns the number of windows; *best is the window of the first
 * strict minimum below 1e6 (-1: none), *lowest its error (:1107,1141-1150). */
int mmlo_time_offset_search(const float* velo_xyz, int n_velo, const float* tf, const float* livox_xyz, int n_livox,
                            int res, int sliced, float* nn_d2, double* win_err, int capacity, int* best, double* lowest);

void mmlo_so3_exp(const double* phi, double* q_xyzw);   /* sophus/so3.hpp:585-622 */
void mmlo_so3_log(const double* q_xyzw, double* phi);   /* sophus/so3.hpp:247-287 */
void mmlo_eig3_sym(const double* A /*row-major 3x3*/, double* evals /*ascending*/, double* evecs /*columns, row-major 3x3*/);
void mmlo_plane_fit5(const double* A /*5x3 row-major*/, double* x /*3*/);

#ifdef __cplusplus
}
#endif
#endif
