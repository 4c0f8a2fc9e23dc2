/*
 * mmloam_hip.h -- C-ABI of libmmloam_hip.so: the MI355X (gfx950) implementation of the mm-loam
 * scan-registration hot path.
 *
 * The reference (TIERS/multi-modal-loam) has no plugin / FFI surface: the path sits behind two C++
 * classes.  This header defines the boundary directly beneath them; every entry point cites the
 * reference member it replaces (paths relative to /root/reference/mm-loam/).  The C++ adapter
 * (multi-modal-loam_amd/host/mmloam_adapter.hpp) re-creates the reference's method signatures on
 * top of these calls; INTEGRATION.md shows the binding a maintainer adds inside the catkin package.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes, the library owns the opaque
 *     mml_ctx (device allocations, maps, one HIP stream).  One ctx per caller thread / per GPU.
 *   - every function returns 0 on success, < 0 on error (mml_status); mml_last_error(ctx) gives
 *     a message.  No exceptions cross the boundary.  There is NO CPU fallback: without a HIP
 *     device mml_create fails with MML_ERR_NO_DEVICE.
 *   - work is batched: a ctx holds `max_scans` scan slots; the reference's one-scan-at-a-time
 *     use is slot 0, count 1.  Calls are stream-ordered on the ctx stream; functions that copy
 *     results to host memory synchronise that stream before returning.
 *   - matrices are row-major doubles; quaternions are (x, y, z, w).
 */
#ifndef MMLOAM_HIP_H
#define MMLOAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MML_ABI_VERSION 1

typedef enum {
    MML_OK = 0,
    MML_ERR_INVALID = -1,   /* bad argument (null, out of range, wrong state) */
    MML_ERR_NO_DEVICE = -2, /* no HIP device / device index out of range */
    MML_ERR_HIP = -3,       /* a HIP runtime call failed */
    MML_ERR_CAPACITY = -4,  /* input exceeds a capacity fixed in mml_config */
    MML_ERR_STATE = -5      /* call order violated (e.g. estimate before a map was set) */
} mml_status;

typedef struct mml_ctx mml_ctx;

/* livox_ros_driver/CustomPoint as laid out by roscpp (20 bytes): pass msg->points.data(). */
typedef struct {
    uint32_t offset_time;
    float x, y, z;
    uint8_t reflectivity, tag, line, _pad;
} mml_livox_point;

typedef struct {
    int max_scans;          /* scan slots (batch capacity) */
    int max_velo_points;    /* per slot */
    int max_livox_points;   /* per slot */
    int n_rings;            /* VELO_N_SCANS, unionFeatureExtract.cpp:192 (16) */
    float pitch0_deg;       /* -15: ring = int((pitch - pitch0) / pitch_step + 0.5), :1162 */
    float pitch_step_deg;   /* 2 */
    int n_livox_lines;      /* 6, unionFeatureExtract.cpp:906,959 */
    float near_th, far_th;  /* near/far_points_threshold, mm_lio_full.launch:28-29 (2.0 / 50.0) */
    float leaf_corner;      /* filter_parameter_corner, launch:43 (0.4), Estimator.cpp:78 */
    float leaf_surf;        /* filter_parameter_surf,   launch:44 (0.2), Estimator.cpp:79 */
    float cell_corner;      /* kNN grid cell edge for the corner map, metres (<=0: 5 * leaf) */
    float cell_surf;        /* kNN grid cell edge for the surf map (<=0: 5 * leaf) */
    int max_features;       /* per slot per kind after down-sampling (<=0: 8192) */
    int max_map_points;     /* per map (<=0: 1<<21) */
} mml_config;

/* Fills the reference's shipped parameters (launch/mm_lio_full.launch) for `max_scans` slots. */
void mml_config_default(mml_config* cfg, int max_scans);

int mml_abi_version(void);
int mml_create(const mml_config* cfg, int device, mml_ctx** out);
void mml_destroy(mml_ctx* ctx);
const char* mml_last_error(const mml_ctx* ctx);
/* The configuration the context was created with, defaults resolved (cell sizes, max_features, max_map_points). */
int mml_config_get(const mml_ctx* ctx, mml_config* out);
int mml_synchronize(mml_ctx* ctx);

/* ---- input ------------------------------------------------------------------------------------
 * Host -> device copy of one union_cloud message (union-cloud/msg/union_cloud.msg: velo_time_aligned
 * as packed x,y,z,intensity floats, livox_time_aligned.points) into a slot.  Either part may be
 * empty (n = 0).  Asynchronous on the ctx stream (the host buffers must stay valid until the next
 * synchronising call). */
int mml_scan_upload(mml_ctx* ctx, int slot, const float* velo_xyzi, int n_velo,
                    const mml_livox_point* livox, int n_livox);

/* Many scans per copy, for a feeder that stages a batch in two host arrays laid out like the slots themselves: scan i of
 * the call (slot first_slot + i) has its n_velo[i] x (x, y, z, intensity) floats at velo_base + i * max_velo_points * 4
 * and its n_livox[i] CustomPoint records at livox_base + i * max_livox_points (max_* as given to mml_create, which must be
 * multiples of 64 for this entry point).  Two host-to-device copies for the whole batch instead of two per scan: at
 * 0.46 MB per copy the per-scan entry point reaches ~11 GB/s, this one the PCIe rate.  Asynchronous like mml_scan_upload,
 * and on a copy stream of its own: the copies start after everything enqueued before the call, the next entry point that
 * touches one of these slots waits for them, and work enqueued on OTHER slots runs concurrently with them -- a feeder
 * that alternates two slot ranges (upload range B, then mml_step on range A, and vice versa) hides the PCIe time behind
 * the kernels. */
int mml_scan_upload_batch(mml_ctx* ctx, int first_slot, int count, const float* velo_base, const int* n_velo,
                          const mml_livox_point* livox_base, const int* n_livox);

/* Same, with the Velodyne part taken straight from a sensor_msgs/PointCloud2 payload (SURVEY section 8(f) rank 3;
 * replaces pcl::fromROSMsg at unionFeatureExtract.cpp:1119-1121): data = msg.data.data(), n_points = width * height,
 * point_step and the byte offsets of the float32 fields x, y, z, intensity as listed in msg.fields
 * (off_intensity < 0: no such field, 0 is used).  The records are decoded on the device. */
int mml_scan_upload_pointcloud2(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x,
                                int off_y, int off_z, int off_intensity, const mml_livox_point* livox, int n_livox);
/* Same again, with the Livox part in wire form as well: `livox_wire` is the serialised `points` array of a
 * livox_ros_driver/CustomMsg (the bytes after its 4-byte length prefix), 19 bytes per CustomPoint, little endian:
 * uint32 offset_time, float32 x, y, z, uint8 reflectivity, tag, line -- the fields getHoriFeatureExtract reads at
 * unionFeatureExtract.cpp:985-997.  For callers that hold the raw message (a bag reader, a non-roscpp transport); a
 * roscpp callback already holds the 20-byte structs mml_scan_upload takes.  Decoded on the device. */
int mml_scan_upload_wire(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x, int off_y,
                         int off_z, int off_intensity, const uint8_t* livox_wire, int n_livox);
/* The fused labelled cloud of a slot as the payload pcl::toROSMsg(pcl::PointCloud<PointXYZINormal>) produces for
 * velo_combine / livox_combine (unionFeatureExtract.cpp:918, :1287-1293): 48-byte records, float32 fields x 0, y 4,
 * z 8, normal_x 16 (in-sweep time), normal_y 20 (ring / line), normal_z 24 (label 0/1/2), intensity 32,
 * curvature 36; packed on the device.  out may be NULL to query *n_points only. */
int mml_scan_download_pointxyzinormal(mml_ctx* ctx, int slot, uint8_t* out, int capacity_points, int* n_points);

/* The way INTO a slot for a labelled cloud that was extracted elsewhere -- the PoseEstimation process receives
 * velo_combine / livox_combine over /union_feature_cloud (unionPoseEstimation.cpp:679-688: pcl::fromROSMsg into
 * laserCloudFullVeloRes / laserCloudFullHoriRes, merged at :746-757) and hands the cloud to RemoveLidarDistortion(cloud,
 * dR, dt) (:862) and, inside a LidarFrame (Estimator.h:33-56), to EstimateLidarPose (:872).  Inverse of
 * mml_scan_download_pointxyzinormal: n_points 48-byte PointXYZINormal records (msg.data.data(), velo_combine followed
 * by livox_combine; the first n_velo are the Velodyne part), decoded on the device: normal_x -> in-sweep time,
 * normal_y -> ring / line, normal_z -> label by abs(normal_z - k) < 1e-5 (Estimator.cpp:995-1003; anything else 0),
 * intensity kept.  Afterwards the slot is in the state mml_extract leaves (mml_scan_info_get, mml_undistort,
 * mml_downsample, mml_estimate work on it); livox_*_num count the kept points only.  Synchronous. */
int mml_cloud_upload(mml_ctx* ctx, int slot, const uint8_t* pointxyzinormal, int n_points, int n_velo);

/* ---- the aligner's time-offset search ---------------------------
 * The numeric core of LidarsParamEstimator::estimate_timeoffset (unionLidarsAligner.cpp:1077-1153): the Velodyne
 * cloud (n_velo x 3 floats) goes through pcl::transformPointCloud with tf (row-major 4x4 floats, NULL = identity;
 * _velo_hori_tf_matrix, :1080-1082); every Livox point (n_livox x 3 floats, the merged last eight messages in arrival
 * order, :1053-1068) gets the squared distance to its nearest transformed Velodyne point (:1084-1103, exact, on the
 * device grid that serves the association); window cnt sums dis_errors[i] + 0.2 * sqrt(x_i^2 + y_i^2) over
 * i in [cnt * search_resolution, cnt * search_resolution + sliced_points) while that end stays below n_livox (:1111-1131).
 * Outputs: nn_d2 (optional, n_livox floats), window_error (optional, `capacity` doubles), *n_windows, *best_window =
 * the window of the first strict minimum below 1e6 (-1: none) and *lowest_error (:1107,1141-1150).  n_velo must not
 * exceed max_map_points.  The caller turns best_window into a stamp (:1143) and applies _time_esti_error_th (:1155). */
int mml_time_offset_search(mml_ctx* ctx, const float* velo_xyz, int n_velo, const float* tf, const float* livox_xyz,
                           int n_livox, int search_resolution, int sliced_points, float* nn_d2, double* window_error,
                           int capacity, int* n_windows, int* best_window, double* lowest_error);

/* ---- SURVEY section 8(f) rank 4 (the other part): the per-frame GICP extrinsic refresh ---------------------------------
 * icp_ext_matching (unionFeatureExtract.cpp:74-123): pcl::GeneralizedIterativeClosestPoint with setMaximumIterations(10),
 * setTransformationEpsilon(1e-6), PCL defaults otherwise, identity guess -- the published PCL 1.8.1 algorithm (20-NN
 * covariances regularised to (1, 1, 1e-3), 1-NN correspondences, BFGS over (t, roll, pitch, yaw) with the float
 * transformation of applyState), all on the device.  src / tgt: n x 3 floats.  T_inout (row-major 4 x 4 float) is
 * overwritten with getFinalTransformation() only when *converged = 1 (hasConverged(), :91-96); clouds with fewer than 20
 * points or fewer than 4 correspondences do not converge ("ICP Failed", :119-122). */
typedef struct {
    int outer_iterations;        /* nr_iterations_ */
    int objective_evaluations;   /* BFGS function / gradient evaluations over all outer iterations */
    double objective;            /* f at the end of the last inner minimisation */
    int n_source, n_target;
} mml_gicp_info;
int mml_gicp_align(mml_ctx* ctx, const float* src_xyz, int n_src, const float* tgt_xyz, int n_tgt, float* T_inout, int* converged,
                   mml_gicp_info* info /* may be NULL */);
/* The call site, unionCloudHandler (:302-318), on a slot that mml_extract filled WITHOUT an extrinsic: when
 * livox_corner_num > 100 the slot's Livox surf cloud (source) is aligned to its Velodyne surf cloud (target) -- every
 * frame, _extrin_cnt never advances (:199,307) -- extrinsic_inout = extri_mtx is updated if the alignment converged
 * (*refreshed), and with apply != 0 the Livox part of the fused cloud is transformed with the (updated or kept) matrix
 * (pcl::transformPointCloud, :312).  With livox_corner_num <= 100 nothing happens, as in the reference.  The two surf
 * clouds are the reference's: label-2 points in the order of each sensor's raw cloud (:1024-1031, :1242-1252), the
 * Velodyne one cropped near + far (:1287-1293), the Livox one near only (removeNearPointCloud, :925 -- its labelled points
 * beyond far_th are part of the source although they are not part of the fused cloud).  MML_ERR_STATE when the slot holds
 * an uploaded or already undistorted cloud -- the refresh belongs between mml_extract and mml_undistort -- or when the slot's
 * raw scan was uploaded again after the extraction (the line ids of the raw records are re-derived from them).  Synchronous. */
int mml_gicp_refresh(mml_ctx* ctx, int slot, float* extrinsic_inout, int apply, int* refreshed, mml_gicp_info* info);

/* ---- a1..a8: feature extraction -----------------------------------------------------------------
 * feature_extraction::unionCloudHandler minus the PCL GICP refresh (unionFeatureExtract.cpp:266-321):
 * getVeloFeature (:1113-1317) + getHoriFeature/getHoriFeatureExtract (:891-1035) with
 * detectFeaturePoints (:341-844) per ring / line, label scatter, near/far crop.  Result per slot: the
 * fused labelled cloud [velo_combine ; livox_combine] kept on the device (what PoseEstimation merges at
 * unionPoseEstimation.cpp:746-757).  livox_extrinsic: optional row-major 4x4 float applied to the Livox
 * part when livox_corner_num > 100 (:302-318); NULL = identity. */
int mml_extract(mml_ctx* ctx, int first_slot, int count, const float* livox_extrinsic);

typedef struct {
    int n_points;       /* fused cloud size */
    int n_velo;         /* leading points that came from the Velodyne */
    int velo_corner_num, velo_surf_num;   /* union_cloud.msg:16-19, set at :1299-1300 */
    int livox_corner_num, livox_surf_num; /* set at :939-940 */
    int fused_corner_num, fused_surf_num; /* label-1 / label-2 points of the fused cloud (corner_cnt, Estimator.cpp:990-1003) */
} mml_scan_info;
int mml_scan_info_get(mml_ctx* ctx, int slot, mml_scan_info* info);
/* Copies the fused labelled cloud of a slot to host (any pointer may be NULL).  Capacity must be
 * >= n_points.  label: 0 none / 1 corner / 2 surf (normal_z); line: ring or Livox line (normal_y);
 * reltime: normal_x.  xyzi: 4 floats per point. */
int mml_scan_download(mml_ctx* ctx, int slot, float* xyzi, float* reltime, uint8_t* line, uint8_t* label,
                      int capacity);

/* Unit-level twin of feature_extraction::detectFeaturePoints (unionFeatureExtract.cpp:341-343) for one
 * scan line: pts = n x (x,y,z,intensity) on the host; sharp / flat receive indices into the line
 * (capacity n each); flags (optional, n ints) receives CloudFeatureFlag[].  Runs in the buffers of scan slot 0:
 * whatever mml_extract / mml_cloud_upload left in slot 0 (fused cloud, labels, scan info) is invalid afterwards. */
int mml_detect_line(mml_ctx* ctx, const float* pts, int n, int* sharp, int* n_sharp, int* flat, int* n_flat,
                    int* flags);

/* ---- a9: RemoveLidarDistortion (unionPoseEstimation.cpp:402-421) ---------------------------------
 * In place on the fused cloud of each slot.  dR: count x 9, dt: count x 3 (host). Sets reltime to 1. */
int mml_undistort(mml_ctx* ctx, int first_slot, int count, const double* dR, const double* dt);

/* ---- a10: label split + pcl::VoxelGrid down-sample (Estimator.cpp:992-1026) ----------------------
 * Any number of labelled points per slot (up to 8192 per kind are sorted in LDS, denser clouds take a global-sort
 * path); the output stacks hold at most max_features voxels per kind (MML_ERR_CAPACITY at the next read-back
 * otherwise).  Not reproduced: PCL's refusal to filter when the voxel grid of a cloud has more than 2^31 cells (it
 * returns the input unfiltered; needs e.g. a 100 m cloud at 5 cm leaves). */
int mml_downsample(mml_ctx* ctx, int first_slot, int count);
/* kind: 0 corner (laserCloudCornerStack), 1 surf (laserCloudSurfStack). xyz: 3 floats per feature. */
int mml_features_download(mml_ctx* ctx, int slot, int kind, float* xyz, int capacity, int* n);
/* Test hook: replace the down-sampled stack of a slot with caller-provided features. */
int mml_features_upload(mml_ctx* ctx, int slot, int kind, const float* xyz, int n);

/* ---- a13 map side: laserCloud{Corner,Surf}FromLocal (Estimator.cpp:1159-1167) --------------------
 * Uploads a map cloud (3 floats per point) and builds the radix-sorted uniform grid that replaces
 * pcl::KdTreeFLANN::setInputCloud.  kind: 0 corner, 1 surf. */
int mml_map_set_local(mml_ctx* ctx, int kind, const float* xyz, int m);
/* ---- Estimator::MapIncrementLocal (Estimator.cpp:1585-1643), on the device ----------
 * Moves the down-sampled corner / surf stacks of `slot` (what mml_downsample left there) to the world frame with
 * T_wl = transformTobeMapped (row-major 4x4; pointAssociateToMap, Map_Manager.cpp:75-89), stores them in ring slot
 * localMapID % 50, rebuilds both local maps as pcl::VoxelGrid(concatenation of the ring) -- the clear() of
 * :1083-1085 / :1125-1127 included -- and rebuilds the kNN grids, all in HBM.  The caller keeps the key-scan rule
 * (pose moved by >= sqrt(0.5) m, :1082,1124).  n_*_map (optional) receive the new map sizes. */
int mml_map_increment_local(mml_ctx* ctx, int slot, const double* T_wl, int* n_corner_map, int* n_surf_map);
/* Empties the ring (localMapID = 0); the current local maps stay until the next set / increment. */
int mml_map_local_reset(mml_ctx* ctx);
/* Copies the current local map (as set or as rebuilt by mml_map_increment_local) to the host: 3 floats per point,
 * in the order the kNN indices refer to.  xyz may be NULL to query *n only. */
int mml_map_local_download(mml_ctx* ctx, int kind, float* xyz, int capacity, int* n);

/* ---- SURVEY section 8(f) rank 2, global half: the MAP_MANAGER cube stores on the device ----------------------------
 * mml_map_global_append   = MAP_MANAGER::featureAssociateToMap (Map_Manager.cpp:91-117): the down-sampled stacks of
 *                           `slot` moved to the world frame with T_wl and appended to the pending clouds
 *                           (laserCloud*_to_map of Estimator::threadMapIncrement, Estimator.cpp:120-122);
 * mml_map_global_increment = MAP_MANAGER::MapIncrement(pending, T_wl) (:125-281) incl. MapMove (:288-581): the cube
 *                           grid follows the sensor, new points go to their cubes, cubes that received points and hold
 *                           > 300 are voxel-filtered (leaf 0.4).  As in the reference (:136-149) the map Estimate()
 *                           matches against becomes the store as it was BEFORE this call (cubes, trees and centre):
 *                           the a12 grids and laserCloudCen*_last are rebuilt from it.  n_* (optional): live store
 *                           sizes after the update.
 * mml_map_global_download: the LIVE store (xyz: 3 floats per point, cube: its index, cen: 3 ints; any may be NULL). */
int mml_map_global_append(mml_ctx* ctx, int slot, const double* T_wl);
int mml_map_global_increment(mml_ctx* ctx, const double* T_wl, int* n_corner, int* n_surf);
int mml_map_global_download(mml_ctx* ctx, int kind, float* xyz, int* cube, int capacity, int* n, int* cen);
int mml_map_global_reset(mml_ctx* ctx);

/* ---- a12: laserCloud{Corner,Surf}FromMap cube store (Estimator.cpp:1170-1184, Map_Manager.cpp:583-629)
 * The reference hands Estimate() 21*11*21 = 4851 cube clouds plus one kd-tree per cube; processPointToLine /
 * processPointToPlane look the feature's cube up (MAP_MANAGER::FindUsedMap), query THAT cube's tree when the
 * cube holds > 100 corner / > 50 surf points, and fall back to the local map otherwise or when the cube
 * neighbourhood fails the fit.  Here the cubes arrive as ONE concatenated cloud: xyz (3 floats per point) and
 * cube[i] = ToIndex(i, j, k) = i + 21 j + 441 k of point i (points of one cube in the cube cloud's order, so
 * that kd-tree tie order is preserved).  cen: laserCloudCen{Width,Height,Depth}_last (3 ints, NULL keeps
 * the current value, default 10, 5, 10).  m = 0 removes the global map (local-only association). */
int mml_map_set_global(mml_ctx* ctx, int kind, const float* xyz, const int* cube, int m, const int* cen);
/* Exact 5-NN against a local map (twin of kdtree->nearestKSearch(p, 5, idx, d2), Estimator.cpp:284,704):
 * q: nq x 3 host floats (already in the map frame); idx: nq x 5 (indices into the uploaded cloud,
 * ascending d2, ties by lower index), d2: nq x 5 (float, ((dx*dx+dy*dy)+dz*dz)). max_d2: search bound
 * (neighbours are exact for every query whose 5th distance is < max_d2; others report -1 / inf). */
int mml_knn5(mml_ctx* ctx, int kind, const float* q, int nq, float max_d2, int* idx, float* d2);

/* ---- a11..a16: association + model fit -------------------------------------------------------------
 * Estimator::processPointToLine (Estimator.cpp:148-365) and processPointToPlanVec (:573-777) on the
 * local maps for `count` slots.  T_wl: count x 16 (transformTobeMapped, :1268-1270).  Factors stay on
 * the device.  stats (count entries) may be NULL: the call then only enqueues (T_wl is staged before it returns, nothing
 * is read back, no synchronisation) -- what a caller does that goes straight on to mml_solve on the same slots. */
typedef struct {
    int n_line, n_plane;        /* factors produced (vLineFeatures / vPlanFeatures sizes) */
    int n_line_used, n_plane_used; /* with |error| > 1e-5 (Estimator.cpp:1385,1396) */
    double normal_gram[9];      /* sum of omega omega^T over plane factors */
    double min_singular;        /* checkLocalizability (:536-565): sqrt(lambda_min(gram)), -1 if n_plane <= 10 */
    int is_degenerate;          /* min_singular < 3.0 (:772-775) */
} mml_assoc_stats;
int mml_associate(mml_ctx* ctx, int first_slot, int count, const double* T_wl, double thres_dist,
                  mml_assoc_stats* stats /* count entries, may be NULL */);

/* Factor read-back for parity tests.  line: n x 10 doubles (pointOri, P1, P2, error) plus src feature
 * index; plane: n x 10 doubles (pointOri, pointProj, omega, error). Capacity in factors. */
int mml_factors_download(mml_ctx* ctx, int slot, int kind, double* out, int* src, int capacity, int* n);
/* The inverse: n factor records in the layout of mml_factors_download become the slot's vLineFeatures (kind 0) /
 * vPlanFeatures (kind 1) -- for callers that keep FeatureLine / FeaturePlanVec objects of their own (the outputs of
 * Estimator::processPointToLine / processPointToPlanVec, Estimator.h:159-186) and only want the solve, and for the
 * known-answer tests of the residual functors (a factor whose point lies exactly on its line).  Record i gets src = i and
 * the slot's feature count of that kind becomes n (the down-sampled stack no longer corresponds to the factors);
 * point / line end points / omega are stored as floats, as the reference's members are (Estimator.cpp:256-271,643-653). */
int mml_factors_upload(mml_ctx* ctx, int slot, int kind, const double* rec, int n);

/* ---- a17..a20: residual + Jacobian + normal equations ------------------------------------------------
 * Cost_NavState_IMU_Line / Cost_NavState_IMU_Plan_Vec (include/utils/ceresfunc.h:397-458, 517-570) with
 * analytic Jacobians, Ceres' Huber correction and the J^T J / J^T r reduction for one slot at pose
 * x = [t, phi].  T_bl: 16 doubles.  H: 36, g: 6, cost: 1 (host). */
int mml_linearize(mml_ctx* ctx, int slot, const double* x, const double* T_bl, double plan_weight_tan,
                  double huber_delta, double* H, double* g, double* cost);

/* The same for the `frames` (<= 8) consecutive slots of a window in ONE launch and one read-back: frame f is linearised
 * at x + f * x_stride (6 doubles [t, phi]; x_stride = 15 walks the [PR | VBias] states of the full-window solver) and
 * its record (MML_NEQ_RECORD_DOUBLES doubles, layout below) is written to records + 32 f.  This is what one trust-region
 * evaluation of Estimator::Estimate costs on the device side (Estimator.cpp:1265-1299 loops over the frames). */
int mml_linearize_window(mml_ctx* ctx, int first_slot, int frames, int x_stride, const double* x, const double* T_bl,
                         double plan_weight_tan, double huber_delta, double* records);

typedef struct {
    int max_num_iterations;  /* Estimator.cpp:1428 (10) */
    int fixed_iterations;    /* != 0: run exactly max_num_iterations, no convergence tests */
    double huber_delta;      /* 0.1 / lidar_m in 1-frame mode (:1216-1222); <= 0: no loss */
    double plan_weight_tan;  /* :1203 / :1206 */
} mml_solve_opts;
typedef struct {
    int iterations, successful;
    double initial_cost, final_cost;
    int termination;         /* 0 max iterations, 1 gradient, 2 parameter, 3 function tolerance, 4 failure (5 consecutive
                                invalid steps: the poses are handed back as they came in, as Solver::Solve does) */
} mml_solve_summary;
/* ceres::Solve replacement (Estimator.cpp:1425-1432): trust-region dogleg on the device, one problem per
 * window of `window` consecutive slots (window = 1: the reference's live 1-frame mode).  x: count x 6
 * in/out; summaries: count / window entries (may be NULL); trace (may be NULL): per problem
 * max_num_iterations x (6*window) doubles, x after every iteration (rows past summaries[p].iterations repeat the
 * final x). */
int mml_solve(mml_ctx* ctx, int first_slot, int count, int window, const double* T_bl,
              const mml_solve_opts* opts, double* x, mml_solve_summary* summaries, double* trace);

/* ---- a21: Estimator::Estimate, 1-frame mode (Estimator.cpp:1143-1581) -------------------------------
 * Outer loop (re-associate with thres_dist 25 -> 10 -> 1, solve, convergence test) for `count` slots.
 * P: count x 3, Q: count x 4 (body pose in/out), exTlb: 16.  Requires mml_downsample first. */
typedef struct {
    int outer_iterations;
    int is_degenerate;
    int n_corner_feat, n_surf_feat;
} mml_estimate_info;
int mml_estimate(mml_ctx* ctx, int first_slot, int count, const double* exTlb, double* P, double* Q,
                 int max_outer, int inner_iters, mml_estimate_info* info /* count entries or NULL */);

/* ---- the benchmarked step ---------------------------------------------------------------------------
 * One pass of the hot path over slots [first, first+count) with inputs already uploaded:
 * extract -> undistort -> downsample -> associate (1 pass, thres_dist) -> `gn_iters` trust-region
 * iterations (fixed count).  Everything stays on the device; poses: count x 6 ([t,phi], device-updated,
 * copied back).  This is what bench.py times.  The batch is spread over the context's stream lanes (mml_set_lanes), every
 * lane ordered behind whatever the context's stream holds when the call is made (no synchronisation is needed between
 * mml_scan_upload and mml_step); inside a lane the stages run in order on the lane's stream; the call returns with every
 * stream of the context drained.  The association statistics of mml_associate (counts, normal Gram matrix) are not part of
 * the step; an entry point that needs them afterwards (mml_linearize*, mml_associate with `stats`) computes them on demand. */
int mml_step(mml_ctx* ctx, int first_slot, int count, const double* dR, const double* dt,
             const double* exTlb, double thres_dist, int gn_iters, double* x_inout);

/* ---- multi-GPU window solve (one frame per GPU) ---------------------------------------
 * Record layout of one frame's normal equations for the all-gather: 32 doubles =
 * [H upper triangle row-major (21), g (6), cost (1), n_line_used, n_plane_used, 0, 0]. */
#define MML_NEQ_RECORD_DOUBLES 32
/* Writes the record of `slot` at pose x to a DEVICE buffer (32 doubles) -- e.g. a torch tensor's
 * data_ptr() that is then passed to all_gather. Stream-ordered; call mml_synchronize before the collective. */
int mml_linearize_record(mml_ctx* ctx, int slot, const double* x, const double* T_bl, double plan_weight_tan,
                         double huber_delta, double* d_record);
/* Host-side joint dogleg state machine over W gathered records (pure host code, no device needed):
 * restates the same trust-region iteration as mml_solve for a block-diagonal window. */
typedef struct mml_window_solver mml_window_solver;
mml_window_solver* mml_window_solver_create(int W, const mml_solve_opts* opts);
void mml_window_solver_destroy(mml_window_solver* s);
/* Feed the records evaluated at the point last returned in x_eval (first call: the initial x).  On return
 * x_eval holds the next point to linearise at; returns 1 when finished (x_eval = solution), 0 to continue,
 * < 0 on error. */
int mml_window_solver_step(mml_window_solver* s, const double* records /* W x 32 */, double* x_eval /* W x 6 */);
int mml_window_solver_summary(const mml_window_solver* s, mml_solve_summary* out);

/* ---- multi-GPU: RCCL inside the C-ABI ---------------------------------------------------------------
 * One process per GPU, one ctx per process.  The caller distributes the 128-byte id of rank 0 by whatever means it
 * has (a ROS parameter, a file, torch.distributed, MPI); mml_comm_init is collective (ncclCommInitRank on the ctx
 * device).  Collectives run on the ctx stream over RCCL / xGMI; no host round trip inside them. */
#define MML_COMM_ID_BYTES 128
int mml_comm_unique_id(uint8_t* id /* MML_COMM_ID_BYTES, filled by the calling rank */);
int mml_comm_init(mml_ctx* ctx, int n_ranks, int rank, const uint8_t* id);
int mml_comm_destroy(mml_ctx* ctx);   /* also done by mml_destroy */
int mml_comm_info(mml_ctx* ctx, int* n_ranks, int* rank);
/* ncclGetVersion of the RCCL copy the entry points of this section are bound to (MAJOR * 10000 + MINOR * 100 + PATCH). */
int mml_rccl_version(int* version);
typedef struct {
    int evaluations;     /* linearisations of the own frames that did work */
    int rounds;          /* kernels enqueued (max_num_iterations + 2) */
    int exchanges;       /* all-gathers enqueued */
    double device_ms;    /* HIP events on the ctx stream around the whole solve */
} mml_window_timing;
/* The joint window solve of Estimator::Estimate across ranks (Estimator.cpp:1265-1299 evaluates the frames of the
 * window one after the other, :1425-1432 solves): the window holds W = n_ranks * n_local frames (<= 8), rank r owns
 * frames [r * n_local, (r + 1) * n_local) in its slots [first_slot, first_slot + n_local), already associated
 * (mml_associate).  Per trust-region evaluation every rank linearises its own frames on the device, the
 * MML_NEQ_RECORD_DOUBLES-double records travel by ncclAllGather, and every rank advances the same device-resident
 * dogleg state machine (the iteration of mml_solve), so all ranks finish with bit-identical poses and no second
 * broadcast.  x_local: n_local x 6 in/out (own frames); x_window (may be NULL): W x 6 out; summary / timing may be
 * NULL.  Collective: every rank of the communicator must call it with the same n_local and opts. */
int mml_window_solve_allgather(mml_ctx* ctx, int first_slot, int n_local, const double* T_bl, const mml_solve_opts* opts,
                               double* x_local, double* x_window, mml_solve_summary* summary, mml_window_timing* timing);
/* Map-update exchange: the down-sampled corner / surf stacks of `slot` on rank `root` replace those of `slot` on
 * every rank (ncclBroadcast, stream-ordered), so that each replica can run the same mml_map_increment_local /
 * mml_map_global_append (the key-scan update of Estimator.cpp:1083-1085, 1125-1130).  Collective. */
int mml_comm_broadcast_features(mml_ctx* ctx, int slot, int root);
/* Replicates rank `root`'s local maps (both kinds, as set or as grown on the device) on every rank and builds the
 * kNN grids there.  Collective; synchronises the ctx. */
int mml_comm_broadcast_local_map(mml_ctx* ctx, int root);

/* ---- loopback group: the N-rank path on ONE device (test / bring-up entry points) -------------------------------------
 * RCCL refuses two ranks on one device, so the rank > 0 side of the three exchanges above -- the section offsets of the
 * gather buffers, "linearise only my frames", a broadcast whose root is another rank -- cannot run through mml_comm_init on
 * a single-GPU box.  A loopback group makes n_ranks contexts of ONE process on ONE device the ranks 0 .. n_ranks-1
 * (ctxs[r] = rank r) of a communicator whose collectives are executed by the calling thread as device-to-device copies
 * between the ranks' buffers; kernels, buffers and the device-resident state machine are those of the RCCL path.  Each
 * call drives ALL ranks (it is the collective).  Not a deployment mode: no overlap, every exchange drains the streams. */
int mml_comm_init_loopback(mml_ctx** ctxs, int n_ranks);
/* mml_window_solve_allgather for every rank of the group: rank r owns window frames [r * n_local, (r + 1) * n_local) in
 * its slots [first_slot[r], first_slot[r] + n_local).  x_window_in: W x 6 starting poses; x_window_out: n_ranks x W x 6,
 * the window as EVERY rank ends up holding it (they must be identical); summaries (may be NULL): n_ranks. */
int mml_window_solve_allgather_loopback(mml_ctx** ctxs, int n_ranks, const int* first_slot, int n_local, const double* T_bl,
                                        const mml_solve_opts* opts, const double* x_window_in, double* x_window_out,
                                        mml_solve_summary* summaries);
int mml_comm_broadcast_features_loopback(mml_ctx** ctxs, int n_ranks, int slot, int root);
int mml_comm_broadcast_local_map_loopback(mml_ctx** ctxs, int n_ranks, int root);

/* ---- verification hook: digests of the per-slot state -------------------------------------------------------------------
 * One 64-bit digest per slot for each piece of state the path leaves behind, computed on the device from what the download
 * entry points would hand out (so a digest can be recomputed on the host from mml_scan_download / mml_features_download /
 * mml_factors_download output: tests/conftest.py host_digest), for checks over batches too large to download slot by slot --
 * e.g. "every slot that was given the same scan holds the same result" at the benchmark's launch shapes.  Every word is a
 * sum over the elements e of the piece of  mix(mix(index(e)) ^ field bits ...)  (splitmix64 finaliser), i.e. independent of
 * the order the elements are stored or visited in.  out: count x MML_DIGEST_WORDS.  Pieces (what they are in the reference):
 *   0 counts       n_points, n_velo, the four union_cloud.msg *_num (union_cloud.msg:4-19), both stack sizes
 *   1 label        normal_z of every fused point, keyed by its index in [velo_combine ; livox_combine]
 *                  (unionFeatureExtract.cpp:1016-1032,1233-1252)
 *   2 line         normal_y (ring / Livox line, :1186, :997)
 *   3 xyzi         x, y, z, intensity -- after mml_undistort the output of RemoveLidarDistortion (unionPoseEstimation.cpp:402-421)
 *   4 reltime      normal_x
 *   5 / 6          corner / surf stack after pcl::VoxelGrid (Estimator.cpp:1015-1024), keyed by position in the stack
 *   7 / 8          line / plane factor records as mml_factors_download returns them (Estimator.h:59-122), keyed by src
 *   9 pose         the 6 doubles of the slot's pose after the last solve (Estimator.cpp:937-964 para_PR)
 * Synchronous. */
#define MML_DIGEST_WORDS 10
int mml_slot_digest(mml_ctx* ctx, int first_slot, int count, uint64_t* out);

/* ---- measurement hooks ----------------------------------------------------------------------------------
 * With profiling on, every kernel launch is bracketed by HIP events on the ctx stream. */
#define MML_MAX_STAGES 32
typedef struct {
    int n_stages;
    const char* name[MML_MAX_STAGES];
    double total_ms[MML_MAX_STAGES];
    long launches[MML_MAX_STAGES];
} mml_profile;
/* ---- IMU factor, marginalization prior, full-window solve (host side) --------------
 * O(W) dense work on <= 8 frames x 15 parameters; the per-frame lidar normal equations come from the device
 * (mml_linearize_record / the all-gather of section 8(e)).  No device is needed for these entry points. */
typedef struct {
    double dp[3], dv[3];
    double dq[4];             /* x, y, z, w */
    double dtime;
    double bg[3], ba[3];      /* linearized_bg / linearized_ba */
    double jacobian[225];     /* 15 x 15 row-major, order P R V BG BA (IMUIntegrator.h:86-93) */
    double covariance[225];
} mml_imu_preint;
/* IMUIntegrator::PreIntegration (IMUIntegrator.cpp:108-166).  samples: n x 7 doubles = angular_velocity xyz,
 * linear_acceleration xyz as in the message (multiplied by gnorm = 9.805 inside, :119-121), dt to the previous
 * sample (:122). */
int mml_imu_preintegrate(const double* samples, int n, const double* bg, const double* ba, mml_imu_preint* out);
/* Cost_NavState_PRV_Bias (ceresfunc.h:321-393) with sqrt_information = LLT(covariance^-1).matrixL()^T
 * (Estimator.cpp:1240-1242).  pr*: [P(3), rotation vector(3)], vb*: [V(3), bg(3), ba(3)].  residual: 15;
 * jacobian (may be NULL): 15 x 30 row-major, columns [pr_i | vb_i | pr_j | vb_j]. */
int mml_imu_factor(const mml_imu_preint* pre, const double* gravity, const double* pr_i, const double* vb_i,
                   const double* pr_j, const double* vb_j, double* residual, double* jacobian);
/* MarginalizationFactor living on (para_PR[0], para_VBias[0]) (ceresfunc.h:244-303): residual =
 * r0 + J * dx(x, x0) with the reference's dx (translation / velocity / bias differences, rotation part
 * log(exp(x)^-1 exp(x0)), :274-283); J is handed to the solver unchanged. */
typedef struct {
    double J[225];            /* linearized_jacobians, 15 x 15 row-major, columns [PR 6 | VBias 9] */
    double r0[15];            /* linearized_residuals */
    double x0[15];            /* keep_block_data */
} mml_prior;
typedef struct mml_fullwindow mml_fullwindow;
/* Estimator::Estimate in full-window mode (Estimator.cpp:1226-1254,1425-1432): W frames x [PR 6 | VBias 9]. */
mml_fullwindow* mml_fullwindow_create(int W, const mml_solve_opts* opts);
void mml_fullwindow_destroy(mml_fullwindow*);
/* IMU factor between frames f-1 and f (1 <= f < W), Estimator.cpp:1235-1248. */
int mml_fullwindow_set_imu(mml_fullwindow*, int f, const mml_imu_preint* pre, const double* gravity);
/* last_marginalization_info (NULL: none), Estimator.cpp:1249-1254. */
int mml_fullwindow_set_prior(mml_fullwindow*, const mml_prior* prior);
/* One trust-region evaluation, same protocol as mml_window_solver_step: records = W x 32 lidar records evaluated at
 * x_eval (W x 15, in/out).  Returns 1 when finished, 0 when x_eval holds the next point to evaluate, < 0 on error. */
int mml_fullwindow_step(mml_fullwindow*, const double* records, double* x_eval);
int mml_fullwindow_summary(const mml_fullwindow*, mml_solve_summary* out);
/* The dense normal equations of the whole window at x (W x 15): H (15W x 15W row-major), g, cost. */
int mml_fullwindow_normal_equations(const mml_fullwindow*, const double* records, const double* x, double* H, double* g,
                                    double* cost);
/* MarginalizationInfo::marginalize for the factor set of Estimator.cpp:1453-1546 (previous prior, IMU factor 0-1,
 * lidar factors of frame 0 as their loss-free normal equations).  x: W x 15.  out: the prior for the slid window. */
int mml_fullwindow_marginalize(const mml_fullwindow*, const double* lidar_record0, const double* x, mml_prior* out);
/* The minimisation of the mml_fullwindow_step loop with the whole trust-region iteration resident on the device: the
 * lidar factors of the W frames in slots first_slot .. first_slot + W - 1 (associated beforehand, evaluated with the
 * handle's plan_weight_tan / huber_delta), the IMU factors and the prior set on `fw` are evaluated, assembled into the
 * block-tridiagonal band of the 15 W normal equations, factored and stepped by ONE kernel launch -- no host round
 * trip per evaluation.  x: W x 15 in/out; summary / evaluations may be NULL.  Afterwards mml_fullwindow_summary
 * reports this solve and mml_fullwindow_marginalize can be called with the returned x. */
int mml_fullwindow_solve(mml_ctx* ctx, mml_fullwindow* fw, int first_slot, const double* T_bl, double* x,
                         mml_solve_summary* summary, int* evaluations);

/* Number of HIP streams mml_step pipelines its sub-batches over (1..8, default 2 or $MML_LANES).  With 1 every
 * kernel covers the whole batch and runs alone on the device, which is what per-kernel timing wants. */
int mml_set_lanes(mml_ctx* ctx, int lanes);
int mml_profile_enable(mml_ctx* ctx, int on);
int mml_profile_reset(mml_ctx* ctx);
int mml_profile_get(mml_ctx* ctx, mml_profile* out);
/* How many points of a slot's last extraction left the fast path: `redo` = points whose float pre-decision of an angle
 * predicate fell inside its guard band and were recomputed with the full decision chain (k_stencil_redo), `brk` = break-point
 * candidates finished by k_stencil_break (unionFeatureExtract.cpp:651-806).  Either may be NULL. */
int mml_extract_queue_counts(mml_ctx* ctx, int slot, int* redo, int* brk);
/* Test hook: the device's copies of the two libm routines the ring / azimuth assignment calls (csrc/libm_f32.h: glibc's atanf and
 * atan2f, the float overloads unionFeatureExtract.cpp:1136-1139,1159,1168 resolve to), evaluated on n host values on the
 * context's device.  out_atan2[i] = atan2f(y[i], x[i]), out_atan[i] = atanf(y[i]); either output may be NULL. */
int mml_libm_f32(mml_ctx* ctx, const float* y, const float* x, long n, float* out_atan2, float* out_atan);
/* How many 5-NN queries of the context's last association call (mml_associate / mml_step's association on the lane that ran
 * last) went beyond rings 0-1 of the grid into the far-query kernels (k_associate_hard): the share to watch when the map's density
 * and the configured cell edge do not fit each other. */
int mml_associate_far_count(mml_ctx* ctx, int* n);
/* Device facts for bench.py: name, CU count, total HBM bytes. */
int mml_device_info(mml_ctx* ctx, char* name, int name_cap, int* cus, size_t* hbm_bytes);
/* Device-to-device copy bandwidth probe (GB/s, read + write counted) over `bytes` bytes, `reps` repetitions: the better of a
 * grid-stride 16-byte copy and a non-temporal one with four loads in flight per lane (the practical HBM roof). */
int mml_copy_bandwidth(mml_ctx* ctx, size_t bytes, int reps, double* gbps);
/* Instruction-issue probe for bench.py's `issue` object: wave64 instructions per second the whole device sustains on eight
 * independent chains per lane at eight wavefronts per SIMD.  kind 0: v_fma_f32 (every float / VOP3 instruction class measured
 * issues at this rate, 4 cycles per wavefront: tools/issue_probe.hip, profiles/r06_issue_probe.txt); kind 1: v_add_u32 (simple
 * 32-bit integer VOP2 instructions, about twice that). */
int mml_issue_rate(mml_ctx* ctx, int kind, int reps, double* wave_instr_per_s);

#ifdef __cplusplus
}
#endif
#endif /* MMLOAM_HIP_H */
