// mmloam_adapter.hpp -- host-side mirror of the reference's C++ surface on top of the C-ABI (include/mmloam_hip.h).
//
// The reference exposes the hot path through two classes (there is no plugin / FFI layer):
//   feature_extraction          mm-loam/src/unionFeatureExtract.cpp:143     detectFeaturePoints :341, getVeloFeature :1113,
//                                                                            getHoriFeatureExtract :952
//   Estimator                   mm-loam/include/Estimator/Estimator.h:25     EstimateLidarPose :211, Estimate :216,
//                                                                            failureDetected :278
//   RemoveLidarDistortion       mm-loam/src/unionPoseEstimation.cpp:402
// This header re-creates those names, argument meanings and error behaviour (no exceptions on the data path, status
// through flags; out-of-grid / NaN points are skipped silently exactly like the reference) over plain buffers.
// PCL / Eigen / ROS are not available in the build image, so clouds are passed as mml::PointXYZINormal arrays with
// pcl::PointXYZINormal's 48-byte layout and field reuse (normal_x = in-scan time, normal_y = ring / line,
// normal_z = label, Estimator.cpp:992-1011); inside a catkin workspace the same struct aliases the PCL type, so the
// adapter works on `cloud->points.data()` without copies of the caller's cloud.
//
// Header-only; link with -lmmloam_hip.  One adapter object per caller thread (the reference's Estimator is not
// re-entrant either, Estimator.h:284-318).
#ifndef MMLOAM_ADAPTER_HPP
#define MMLOAM_ADAPTER_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "mmloam_hip.h"

namespace mml {

// pcl::PointXYZINormal memory layout (48 bytes, 16-byte aligned): x,y,z,pad | normal_x,normal_y,normal_z,pad |
// intensity, curvature, pad, pad
struct alignas(16) PointXYZINormal {
    float x, y, z, data_pad;
    float normal_x, normal_y, normal_z, normal_pad;
    float intensity, curvature, pad0, pad1;
};
static_assert(sizeof(PointXYZINormal) == 48, "must match pcl::PointXYZINormal");
using PointCloud = std::vector<PointXYZINormal>;

struct Matrix3d {  // row-major
    double m[9];
};
struct Vector3d {
    double v[3];
};
struct Matrix4d {  // row-major
    double m[16];
};
struct Quaterniond {  // x, y, z, w
    double x = 0, y = 0, z = 0, w = 1;
};

inline void check(mml_ctx* ctx, int rc, const char* what) {
    if (rc != MML_OK) throw std::runtime_error(std::string(what) + ": " + (ctx ? mml_last_error(ctx) : "no context"));
}

// RAII owner of one mml_ctx, shared by the two adapter classes of a node pair living in one process
class Context {
   public:
    explicit Context(int max_scans = 1, int device = 0, const mml_config* cfg = nullptr) {
        mml_config c;
        if (cfg)
            c = *cfg;
        else
            mml_config_default(&c, max_scans);
        int rc = mml_create(&c, device, &ctx_);
        if (rc != MML_OK) throw std::runtime_error("mml_create failed (no HIP device? there is no CPU fallback)");
        cfg_ = c;
    }
    ~Context() { mml_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    mml_ctx* get() const { return ctx_; }
    const mml_config& config() const { return cfg_; }

   private:
    mml_ctx* ctx_ = nullptr;
    mml_config cfg_;
};

// ---- feature_extraction (unionFeatureExtract.cpp:143-1320), hot-path members only -----------------------------------
class feature_extraction {
   public:
    explicit feature_extraction(Context& ctx) : ctx_(ctx) {}

    // unionFeatureExtract.cpp:341-343.  `cloud` is one scan line; indices refer to it.
    void detectFeaturePoints(const PointCloud& cloud, std::vector<int>& pointsLessSharp, std::vector<int>& pointsLessFlat) {
        const int n = (int)cloud.size();
        std::vector<float> pts(4 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            pts[4 * i] = cloud[i].x;
            pts[4 * i + 1] = cloud[i].y;
            pts[4 * i + 2] = cloud[i].z;
            pts[4 * i + 3] = cloud[i].intensity;
        }
        std::vector<int> sharp(n ? n : 1), flat(n ? n : 1);
        int ns = 0, nf = 0;
        check(ctx_.get(), mml_detect_line(ctx_.get(), pts.data(), n, sharp.data(), &ns, flat.data(), &nf, nullptr),
              "detectFeaturePoints");
        pointsLessSharp.insert(pointsLessSharp.end(), sharp.begin(), sharp.begin() + ns);
        pointsLessFlat.insert(pointsLessFlat.end(), flat.begin(), flat.begin() + nf);
    }

    // unionCloudHandler (:266-321) minus ROS (de)serialisation and the PCL GICP refresh: one union_cloud message in,
    // the labelled fused cloud [velo_combine ; livox_combine] out, plus the four counters of union_cloud.msg:16-19.
    // velo_xyzi: n_velo x (x,y,z,intensity); livox: msg->livox_time_aligned.points.data().
    void unionCloud(const float* velo_xyzi, int n_velo, const mml_livox_point* livox, int n_livox,
                    const float* livox_extrinsic /*row-major 4x4 or nullptr*/, PointCloud& fused, mml_scan_info& info,
                    int slot = 0) {
        mml_ctx* c = ctx_.get();
        check(c, mml_scan_upload(c, slot, velo_xyzi, n_velo, livox, n_livox), "scan_upload");
        check(c, mml_extract(c, slot, 1, livox_extrinsic), "extract");
        check(c, mml_scan_info_get(c, slot, &info), "scan_info");
        download(slot, info.n_points, fused);
    }

    // unionCloudHandler including its GICP refresh (:302-318): features without an extrinsic, then -- when
    // livox_corner_num > 100 -- extri_mtx re-estimated from the Livox surf cloud against the Velodyne surf cloud (every frame,
    // as the reference does) and applied to the Livox part; extri_mtx (row-major) persists across frames in the caller.
    bool unionCloudRefresh(const float* velo_xyzi, int n_velo, const mml_livox_point* livox, int n_livox, float extri_mtx[16],
                           PointCloud& fused, mml_scan_info& info, int slot = 0) {
        mml_ctx* c = ctx_.get();
        check(c, mml_scan_upload(c, slot, velo_xyzi, n_velo, livox, n_livox), "scan_upload");
        check(c, mml_extract(c, slot, 1, nullptr), "extract");
        int refreshed = 0;
        check(c, mml_gicp_refresh(c, slot, extri_mtx, 1, &refreshed, nullptr), "gicp_refresh");
        check(c, mml_scan_info_get(c, slot, &info), "scan_info");
        download(slot, info.n_points, fused);
        return refreshed != 0;
    }

    // getVeloFeature (:1113-1117) and getHoriFeatureExtract (:952-956) on their own
    void getVeloFeature(const float* velo_xyzi, int n, PointCloud& points_normal, mml_scan_info& info, int slot = 0) {
        unionCloud(velo_xyzi, n, nullptr, 0, nullptr, points_normal, info, slot);
    }
    void getHoriFeatureExtract(const mml_livox_point* pts, int n, PointCloud& laserCloud, mml_scan_info& info, int slot = 0) {
        unionCloud(nullptr, 0, pts, n, nullptr, laserCloud, info, slot);
    }

    void download(int slot, int n, PointCloud& out) {
        std::vector<float> xyzi(4 * (size_t)(n ? n : 1)), rel(n ? n : 1);
        std::vector<uint8_t> line(n ? n : 1), label(n ? n : 1);
        check(ctx_.get(), mml_scan_download(ctx_.get(), slot, xyzi.data(), rel.data(), line.data(), label.data(), n ? n : 1),
              "scan_download");
        out.assign(n, PointXYZINormal{});
        for (int i = 0; i < n; ++i) {
            out[i].x = xyzi[4 * i];
            out[i].y = xyzi[4 * i + 1];
            out[i].z = xyzi[4 * i + 2];
            out[i].data_pad = 1.f;
            out[i].intensity = xyzi[4 * i + 3];
            out[i].normal_x = rel[i];
            out[i].normal_y = (float)line[i];
            out[i].normal_z = (float)label[i];
        }
    }

   private:
    Context& ctx_;
};

// ---- icp_ext_matching (unionFeatureExtract.cpp:74-123): the GICP behind the per-frame extrinsic refresh --------------------
// bool icp_ext_matching(cloud_src, cloud_tgt, cloud_aligned, icp_mtx, en_viewer): icp_mtx (row-major here) is assigned and
// cloud_aligned filled only when the alignment converged.
inline bool icp_ext_matching(Context& ctx, const PointCloud& cloud_src, const PointCloud& cloud_tgt, PointCloud& cloud_aligned,
                             float icp_mtx[16]) {
    std::vector<float> s(3 * cloud_src.size() + 3), t(3 * cloud_tgt.size() + 3);
    for (size_t i = 0; i < cloud_src.size(); ++i) {
        s[3 * i] = cloud_src[i].x;
        s[3 * i + 1] = cloud_src[i].y;
        s[3 * i + 2] = cloud_src[i].z;
    }
    for (size_t i = 0; i < cloud_tgt.size(); ++i) {
        t[3 * i] = cloud_tgt[i].x;
        t[3 * i + 1] = cloud_tgt[i].y;
        t[3 * i + 2] = cloud_tgt[i].z;
    }
    int converged = 0;
    check(ctx.get(), mml_gicp_align(ctx.get(), s.data(), (int)cloud_src.size(), t.data(), (int)cloud_tgt.size(), icp_mtx, &converged, nullptr),
          "icp_ext_matching");
    if (!converged) return false;
    cloud_aligned = cloud_src;  // icp.align(*cloud_aligned): the source under the final transformation
    for (auto& p : cloud_aligned) {
        const float x = icp_mtx[0] * p.x + icp_mtx[1] * p.y + icp_mtx[2] * p.z + icp_mtx[3];
        const float y = icp_mtx[4] * p.x + icp_mtx[5] * p.y + icp_mtx[6] * p.z + icp_mtx[7];
        const float z = icp_mtx[8] * p.x + icp_mtx[9] * p.y + icp_mtx[10] * p.z + icp_mtx[11];
        p.x = x;
        p.y = y;
        p.z = z;
    }
    return true;
}

// ---- RemoveLidarDistortion (unionPoseEstimation.cpp:402-403): in place on the slot's device-resident fused cloud ----
inline void RemoveLidarDistortion(Context& ctx, int slot, const Matrix3d& dRlc, const Vector3d& dtlc) {
    check(ctx.get(), mml_undistort(ctx.get(), slot, 1, dRlc.m, dtlc.v), "RemoveLidarDistortion");
}

// A labelled cloud that arrived over /union_feature_cloud (velo_combine followed by livox_combine, the merge of
// unionPoseEstimation.cpp:746-757) -> scan slot: pcl::fromROSMsg on the device.  n_velo = velo_combine's size.
// n_velo = -1 (the caller does not know the split, as the reference's RemoveLidarDistortion does not): the cloud fills the
// Velodyne region first and spills the rest into the Livox region, so a merged cloud of up to max_velo_points +
// max_livox_points is accepted; the velo_* / livox_* label counts and mml_gicp_refresh are then not meaningful for the slot
// (undistortion, down-sampling and the estimator do not depend on the split).
inline void uploadCloud(Context& ctx, int slot, const PointCloud& cloud, int n_velo = -1) {
    const int n = (int)cloud.size();
    if (n_velo < 0) n_velo = n < ctx.config().max_velo_points ? n : ctx.config().max_velo_points;
    check(ctx.get(), mml_cloud_upload(ctx.get(), slot, reinterpret_cast<const uint8_t*>(cloud.data()), n, n_velo),
          "mml_cloud_upload");
}

// void RemoveLidarDistortion(pcl::PointCloud<PointType>::Ptr& cloud, const Eigen::Matrix3d& dRlc, const Eigen::Vector3d& dtlc)
// exactly as the pose node calls it (unionPoseEstimation.cpp:862): the CALLER's cloud is undistorted in place (x, y, z
// rewritten, normal_x = 1, :416-419).  The cloud stays resident in `slot` as well, so a LidarFrame built from it can
// name the slot instead of carrying the points back (LidarFrame::resident).
inline void RemoveLidarDistortion(Context& ctx, PointCloud& cloud, const Matrix3d& dRlc, const Vector3d& dtlc, int slot = 0,
                                  int n_velo = -1) {
    const int n = (int)cloud.size();
    if (n == 0) return;
    uploadCloud(ctx, slot, cloud, n_velo);
    check(ctx.get(), mml_undistort(ctx.get(), slot, 1, dRlc.m, dtlc.v), "RemoveLidarDistortion");
    std::vector<float> xyzi(4 * (size_t)n), rel(n);
    check(ctx.get(), mml_scan_download(ctx.get(), slot, xyzi.data(), rel.data(), nullptr, nullptr, n), "scan_download");
    for (int i = 0; i < n; ++i) {
        cloud[i].x = xyzi[4 * i];
        cloud[i].y = xyzi[4 * i + 1];
        cloud[i].z = xyzi[4 * i + 2];
        cloud[i].normal_x = rel[i];
    }
}

// ---- Estimator (Estimator.h:25-343), hot-path members only ------------------------------------------------------------
class Estimator {
   public:
    static const int SLIDEWINDOWSIZE = 5;  // Estimator.h:30

    struct LidarFrame {  // Estimator.h:33-56
        // laserCloud (Estimator.h:35): the caller's labelled, undistorted cloud.  When set (and not `resident`) it is
        // uploaded into scan slot `slot` at the start of EstimateLidarPose / EstimateFullWindow; a cloud that is already
        // in the slot (extracted there, or left there by RemoveLidarDistortion(ctx, cloud, ...)) is named by slot alone.
        PointCloud* laserCloud = nullptr;
        int n_velo = -1;        // velo_combine's share of laserCloud (-1: all of it)
        bool resident = false;  // the slot already holds this cloud
        int slot = 0;
        Vector3d P{{0, 0, 0}};
        Vector3d V{{0, 0, 0}};
        Quaterniond Q;
        Vector3d bg{{0, 0, 0}};
        Vector3d ba{{0, 0, 0}};
        double timeStamp = 0;
        int lidarType = 0;
    };

    // FeatureLine / FeaturePlanVec (Estimator.h:59-84, 105-122) as the association leaves them.  sqrt_info of the plane
    // factor is carried as the plane normal omega: sqrt_info^T sqrt_info = a^2 w w^T + b^2 (I - w w^T) does not depend on the
    // basis the reference's SVD happens to pick (a = 1 / lidar_m, b = plan_weight_tan / lidar_m, Estimator.cpp:675-682).
    struct FeatureLine {
        Vector3d pointOri, lineP1, lineP2;
        double error = 0;
        bool valid = false;
    };
    struct FeaturePlanVec {
        Vector3d pointOri, pointProj, omega;
        double error = 0;
        bool valid = false;
    };
    double thres_dist = 25.0;  // Estimator.h:332, set by Estimate() (25 -> 10 -> 1, Estimator.cpp:1207,1377-1381)
    // EstimateFullWindow: true = the joint ceres::Solve is one device-resident iteration (mml_fullwindow_solve, needs the
    // window in consecutive slots); false = host iteration with one device evaluation per trust-region step
    bool device_window_solve = true;

    // processPointToLine / processPointToPlanVec (Estimator.h:159-186, Estimator.cpp:148-365, 573-777) for the down-sampled
    // stacks of `slot` at lidar pose m4d = transformTobeMapped: kNN + model fit on the device (both kinds in one launch),
    // the surviving features appended to the vector with their ComputeError() value; `valid` = |error| > 1e-5, i.e.
    // the factor enters the problem (:1313,1325).  The cost functions themselves stay on the device (mml_solve).
    void processPointToLine(std::vector<FeatureLine>& vLineFeatures, int slot, const Matrix4d& m4d) {
        associate(slot, m4d);
        std::vector<double> f;
        const int n = factors(slot, 0, f);
        for (int i = 0; i < n; ++i) {
            FeatureLine l;
            for (int c = 0; c < 3; ++c) {
                l.pointOri.v[c] = f[10 * i + c];
                l.lineP1.v[c] = f[10 * i + 3 + c];
                l.lineP2.v[c] = f[10 * i + 6 + c];
            }
            l.error = f[10 * i + 9];
            l.valid = std::fabs(l.error) > 1e-5;
            vLineFeatures.push_back(l);
        }
    }
    void processPointToPlanVec(std::vector<FeaturePlanVec>& vPlanFeatures, int slot, const Matrix4d& m4d, bool& is_degenerate) {
        mml_assoc_stats st = associate(slot, m4d);
        if (st.is_degenerate) is_degenerate = true;  // only ever set, never cleared (:771-775)
        std::vector<double> f;
        const int n = factors(slot, 1, f);
        for (int i = 0; i < n; ++i) {
            FeaturePlanVec pl;
            for (int c = 0; c < 3; ++c) {
                pl.pointOri.v[c] = f[10 * i + c];
                pl.pointProj.v[c] = f[10 * i + 3 + c];
                pl.omega.v[c] = f[10 * i + 6 + c];
            }
            pl.error = f[10 * i + 9];
            pl.valid = std::fabs(pl.error) > 1e-5;
            vPlanFeatures.push_back(pl);
        }
    }

    // Estimate(lidarFrameList, exTlb, gravity, is_degenerate) (Estimator.h:216-219) in the live 1-frame mode (windowSize !=
    // SLIDEWINDOWSIZE, Estimator.cpp:1143-1581): every frame registered against the maps, poses updated in place.
    void Estimate(std::list<LidarFrame>& lidarFrameList, const Matrix4d& exTlb, const Vector3d& gravity, bool& is_degenerate) {
        (void)gravity;
        for (auto& f : lidarFrameList) {
            double Q[4] = {f.Q.x, f.Q.y, f.Q.z, f.Q.w};
            mml_estimate_info info;
            check(ctx_.get(), mml_estimate(ctx_.get(), f.slot, 1, exTlb.m, f.P.v, Q, 5, 10, &info), "Estimate");
            f.Q.x = Q[0];
            f.Q.y = Q[1];
            f.Q.z = Q[2];
            f.Q.w = Q[3];
            if (info.is_degenerate) is_degenerate = true;
        }
    }

    int keyScans() const { return key_scans_; }
    int localMapCorners() const { return n_corner_map_; }
    int localMapSurfs() const { return n_surf_map_; }

    // get_corner_map() / get_surf_map() (Estimator.h:221-226): the MAP_MANAGER cube store, concatenated
    PointCloud get_corner_map() { return global_map(0); }
    PointCloud get_surf_map() { return global_map(1); }

    // Estimator(const float& filter_corner, const float& filter_surf) (Estimator.h:147): the leaf sizes are part of
    // mml_config (leaf_corner / leaf_surf) and must match the context's.
    Estimator(Context& ctx, float filter_corner, float filter_surf) : ctx_(ctx) {
        mml_config c;
        check(ctx.get(), mml_config_get(ctx.get(), &c), "mml_config_get");
        if (c.leaf_corner != filter_corner || c.leaf_surf != filter_surf)
            throw std::invalid_argument("Estimator: filter_corner / filter_surf differ from the context's leaf_corner / leaf_surf");
    }

    // pcl::fromROSMsg + the LidarFrame hand-over: bring every frame's cloud into its slot
    void residentClouds(std::list<LidarFrame>& frames) {
        for (auto& f : frames) stage(f);
    }

    // laserCloud{Corner,Surf}FromLocal (Estimator.cpp:1159-1167): replaces kdtree->setInputCloud
    void setLocalMap(const float* corner_xyz, int n_corner, const float* surf_xyz, int n_surf) {
        check(ctx_.get(), mml_map_set_local(ctx_.get(), 0, corner_xyz, n_corner), "map corner");
        check(ctx_.get(), mml_map_set_local(ctx_.get(), 1, surf_xyz, n_surf), "map surf");
        n_corner_map_ = n_corner;
        n_surf_map_ = n_surf;
    }

    // laserCloud{Corner,Surf}FromMap[4851] + kdtree{Corner,Surf}FromMap[4851] (Estimator.cpp:1170-1184): the cube
    // clouds concatenated, cube[i] = index of point i's cube in the MAP_MANAGER arrays; cen = laserCloudCen*_last.
    void setGlobalMap(const float* corner_xyz, const int* corner_cube, int n_corner, const float* surf_xyz,
                      const int* surf_cube, int n_surf, const int cen[3]) {
        check(ctx_.get(), mml_map_set_global(ctx_.get(), 0, corner_xyz, corner_cube, n_corner, cen), "global corner");
        check(ctx_.get(), mml_map_set_global(ctx_.get(), 1, surf_xyz, surf_cube, n_surf, cen), "global surf");
    }

    // Estimator::threadMapIncrement (Estimator.cpp:92-145) with the MAP_MANAGER cube stores on the device:
    // featureAssociateToMap -> mml_map_global_append, MapIncrement (+ MapMove) -> mml_map_global_increment.
    void appendToGlobalMap(const LidarFrame& f, const Matrix4d& transformForMap) {
        check(ctx_.get(), mml_map_global_append(ctx_.get(), f.slot, transformForMap.m), "featureAssociateToMap");
    }
    void incrementGlobalMap(const Matrix4d& transform) {
        check(ctx_.get(), mml_map_global_increment(ctx_.get(), transform.m, nullptr, nullptr), "MapIncrement");
    }

    // EstimateLidarPose(std::list<LidarFrame>&, exTlb, gravity, lidarMode) (Estimator.h:211-214, Estimator.cpp:967-1140).
    // Live 1-frame mode: the list holds one frame; it is registered against the local map, the
    // pose is updated in place, and the key-scan rule (:1121-1135) grows the local map ON THE DEVICE
    // (mml_map_increment_local replaces MapIncrementLocal + the two kd-tree rebuilds of the next Estimate call).
    // failureDetected() reports the degeneracy flag (:1139).
    void EstimateLidarPose(std::list<LidarFrame>& lidarFrameList, const Matrix4d& exTlb, const Vector3d& gravity,
                           int lidarMode) {
        (void)gravity;
        if (lidarFrameList.empty()) return;
        // exRbl = exTlb.R^T, exPbl = -exRbl * exTlb.t (:973-974)
        double exRbl[9], exPbl[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) exRbl[3 * r + c] = exTlb.m[4 * c + r];
        for (int r = 0; r < 3; ++r)
            exPbl[r] = -1.0 * ((exRbl[3 * r] * exTlb.m[3] + exRbl[3 * r + 1] * exTlb.m[7]) + exRbl[3 * r + 2] * exTlb.m[11]);
        auto to_be_mapped = [&](const LidarFrame& f, double* T) {
            const double x = f.Q.x, y = f.Q.y, z = f.Q.z, w = f.Q.w;
            const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                                 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                                 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    T[4 * r + c] = (R[3 * r] * exRbl[c] + R[3 * r + 1] * exRbl[3 + c]) + R[3 * r + 2] * exRbl[6 + c];
                T[4 * r + 3] = ((R[3 * r] * exPbl[0] + R[3 * r + 1] * exPbl[1]) + R[3 * r + 2] * exPbl[2]) + f.P.v[r];
            }
            T[12] = T[13] = T[14] = 0;
            T[15] = 1;
        };
        double T[16];
        to_be_mapped(lidarFrameList.back(), T);  // :975-977
        int corner_cnt = 0;
        for (auto& f : lidarFrameList) {
            stage(f);
            mml_scan_info si;
            check(ctx_.get(), mml_scan_info_get(ctx_.get(), f.slot, &si), "scan info");
            corner_cnt += si.fused_corner_num;  // :990-996
            check(ctx_.get(), mml_downsample(ctx_.get(), f.slot, 1), "downsample");  // :1013-1024
        }
        bool is_degenerate = false;
        if (n_corner_map_ > 0 && n_surf_map_ > 100) Estimate(lidarFrameList, exTlb, gravity, is_degenerate);  // :1032-1035
        LidarFrame& front = lidarFrameList.front();
        if ((lidarMode == 1 && !is_degenerate && corner_cnt > 100) || (lidarMode == 2 && corner_cnt > 50)) {
            to_be_mapped(front, T);  // :1041-1049
        } else {                     // :1050-1066
            T[3] = front.P.v[0];
            T[7] = front.P.v[1];
        }
        if (!is_degenerate) {  // :1070-1136
            // Both modes COMPARE with last_velo_update_pose (:1082, :1119); mode 1 WRITES last_hori_update_pose (:1115), which
            // nothing ever reads: a Horizon-mode estimator that never sees a mode-2 call keeps comparing with (-1, -1, -1) and
            // grows its map on every scan farther than sqrt(0.5) m from there.  Preserved as it is.
            const double dx = last_velo_update_pose_[0] - T[3], dy = last_velo_update_pose_[1] - T[7], dz = last_velo_update_pose_[2] - T[11];
            const double d2 = lidarMode == 2 ? (double)(float)(dx * dx + dy * dy + dz * dz) : (dx * dx + dy * dy + dz * dz);
            if (d2 >= 0.5 && (lidarMode == 1 || lidarMode == 2)) {
                check(ctx_.get(), mml_map_increment_local(ctx_.get(), front.slot, T, &n_corner_map_, &n_surf_map_),
                      "MapIncrementLocal");
                ++key_scans_;
                double* last = lidarMode == 2 ? last_velo_update_pose_ : last_hori_update_pose_;
                last[0] = T[3];
                last[1] = T[7];
                last[2] = T[11];
            }
        }
        _fail_detected = is_degenerate;  // :1139
    }

    // Estimate(lidarFrameList, exTlb, gravity, is_degenerate) in full-window mode (windowSize == SLIDEWINDOWSIZE,
    // Estimator.cpp:1143-1581, IMU_Mode = 2): lidar factors of every frame associated once (thres_dist 1,
    // plan_weight_tan 3e-4, no loss) and linearised on the device, IMU factors (LidarFrame::imu) between consecutive
    // frames, the marginalization prior of the previous call; 5 outer x 10 inner iterations, then frame 0 is
    // marginalized (:1448-1567).  imu[f] (f >= 1) = the pre-integration between frames f-1 and f
    // (mml_imu_preintegrate replaces IMUIntegrator::PreIntegration).
    void EstimateFullWindow(std::vector<LidarFrame*>& frames, const std::vector<mml_imu_preint>& imu, const Matrix4d& exTlb,
                            const Vector3d& gravity) {
        const int W = (int)frames.size();
        if (W < 1 || W > 8 || (int)imu.size() < W) throw std::runtime_error("EstimateFullWindow: bad window");
        for (auto* f : frames) {
            if (f->laserCloud && !f->resident) {
                stage(*f);
                check(ctx_.get(), mml_downsample(ctx_.get(), f->slot, 1), "downsample");  // Estimator.cpp:1013-1024
            }
        }
        double T_bl[16];  // inverse of exTlb
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T_bl[4 * r + c] = exTlb.m[4 * c + r];
            T_bl[4 * r + 3] = -((exTlb.m[r] * exTlb.m[3] + exTlb.m[4 + r] * exTlb.m[7]) + exTlb.m[8 + r] * exTlb.m[11]);
        }
        T_bl[12] = T_bl[13] = T_bl[14] = 0;
        T_bl[15] = 1;
        const double w_tan = 0.0003, thres_dist = 1.0;  // :1203-1204
        std::vector<double> x(15 * (size_t)W), rec(32 * (size_t)W);
        for (int it = 0; it < 5; ++it) {
            for (int f = 0; f < W; ++f) {  // vector2double (:937-950)
                const LidarFrame& l = *frames[f];
                double* xf = &x[15 * (size_t)f];
                for (int k = 0; k < 3; ++k) {
                    xf[k] = l.P.v[k];
                    xf[6 + k] = l.V.v[k];
                    xf[9 + k] = l.bg.v[k];
                    xf[12 + k] = l.ba.v[k];
                }
                quat_log(l.Q, xf + 3);
            }
            if (it == 0)
                for (int f = 0; f < W; ++f) {
                    double T_wl[16];
                    body_to_lidar_world(*frames[f], exTlb, T_wl);
                    check(ctx_.get(), mml_associate(ctx_.get(), frames[f]->slot, 1, T_wl, thres_dist, nullptr), "associate");
                }
            const Quaterniond q_before = frames[W - 1]->Q;
            const Vector3d t_before = frames[W - 1]->P;
            mml_solve_opts so{10, 0, 0.0, w_tan};
            mml_fullwindow* fw = mml_fullwindow_create(W, &so);
            if (!fw) throw std::runtime_error("mml_fullwindow_create");
            for (int f = 1; f < W; ++f) mml_fullwindow_set_imu(fw, f, &imu[f], gravity.v);
            if (have_prior_) mml_fullwindow_set_prior(fw, &prior_);
            bool consecutive = true;
            for (int f = 1; f < W; ++f) consecutive = consecutive && frames[f]->slot == frames[0]->slot + f;
            auto lidar_records = [&]() {
                if (consecutive) {  // one launch + one read-back for the whole window
                    check(ctx_.get(), mml_linearize_window(ctx_.get(), frames[0]->slot, W, 15, x.data(), T_bl, w_tan, 0.0, rec.data()),
                          "linearize_window");
                    return;
                }
                for (int f = 0; f < W; ++f) {
                    double H[36], g[6], c;
                    check(ctx_.get(), mml_linearize(ctx_.get(), frames[f]->slot, &x[15 * (size_t)f], T_bl, w_tan, 0.0, H, g, &c),
                          "linearize");
                    double* r = &rec[32 * (size_t)f];
                    int k = 0;
                    for (int a = 0; a < 6; ++a)
                        for (int b = a; b < 6; ++b) r[k++] = H[6 * a + b];
                    for (int a = 0; a < 6; ++a) r[21 + a] = g[a];
                    r[27] = c;
                    r[28] = r[29] = r[30] = r[31] = 0;
                }
            };
            if (consecutive && device_window_solve) {  // ceres::Solve as one device-resident iteration
                const int rc = mml_fullwindow_solve(ctx_.get(), fw, frames[0]->slot, T_bl, x.data(), nullptr, nullptr);
                if (rc != MML_OK) {
                    mml_fullwindow_destroy(fw);
                    check(ctx_.get(), rc, "mml_fullwindow_solve");
                }
            } else {
                for (int guard = 0; guard < 200; ++guard) {  // host iteration, one device evaluation per step
                    lidar_records();
                    const int rc = mml_fullwindow_step(fw, rec.data(), x.data());
                    if (rc < 0) {
                        mml_fullwindow_destroy(fw);
                        throw std::runtime_error("mml_fullwindow_step");
                    }
                    if (rc == 1) break;
                }
            }
            for (int f = 0; f < W; ++f) {  // double2vector (:952-964)
                LidarFrame& l = *frames[f];
                const double* xf = &x[15 * (size_t)f];
                for (int k = 0; k < 3; ++k) {
                    l.P.v[k] = xf[k];
                    l.V.v[k] = xf[6 + k];
                    l.bg.v[k] = xf[9 + k];
                    l.ba.v[k] = xf[12 + k];
                }
                quat_exp(xf + 3, l.Q);
            }
            const Quaterniond& qa = frames[W - 1]->Q;
            const double dot = std::fabs(q_before.x * qa.x + q_before.y * qa.y + q_before.z * qa.z + q_before.w * qa.w);
            const double deltaR = 2.0 * std::acos(dot > 1.0 ? 1.0 : dot) * 180.0 / M_PI;
            double dT = 0;
            for (int k = 0; k < 3; ++k) dT += (t_before.v[k] - frames[W - 1]->P.v[k]) * (t_before.v[k] - frames[W - 1]->P.v[k]);
            const bool last = (deltaR < 0.05 && std::sqrt(dT) < 0.05) || it == 4;
            if (last && W >= 2) {  // :1453-1546
                lidar_records();
                have_prior_ = mml_fullwindow_marginalize(fw, rec.data(), x.data(), &prior_) == MML_OK;
            }
            mml_fullwindow_destroy(fw);
            if (last) break;
        }
    }

    bool failureDetected() const { return _fail_detected; }  // Estimator.h:278

   private:
    mml_assoc_stats associate(int slot, const Matrix4d& m4d) {
        mml_assoc_stats st;
        check(ctx_.get(), mml_associate(ctx_.get(), slot, 1, m4d.m, thres_dist, &st), "mml_associate");
        return st;
    }
    int factors(int slot, int kind, std::vector<double>& f) {
        int n = 0;
        check(ctx_.get(), mml_factors_download(ctx_.get(), slot, kind, nullptr, nullptr, 0, &n), "mml_factors_download");
        f.assign(10 * (size_t)(n > 0 ? n : 1), 0.0);
        check(ctx_.get(), mml_factors_download(ctx_.get(), slot, kind, f.data(), nullptr, n > 0 ? n : 1, &n), "mml_factors_download");
        return n;
    }
    PointCloud global_map(int kind) {
        int n = 0, cen[3];
        check(ctx_.get(), mml_map_global_download(ctx_.get(), kind, nullptr, nullptr, 0, &n, cen), "mml_map_global_download");
        std::vector<float> xyz(3 * (size_t)(n > 0 ? n : 1));
        check(ctx_.get(), mml_map_global_download(ctx_.get(), kind, xyz.data(), nullptr, n, &n, cen), "mml_map_global_download");
        PointCloud out(n, PointXYZINormal{});
        for (int i = 0; i < n; ++i) {
            out[i].x = xyz[3 * i];
            out[i].y = xyz[3 * i + 1];
            out[i].z = xyz[3 * i + 2];
            out[i].data_pad = 1.f;
        }
        return out;
    }
    void stage(LidarFrame& f) {
        if (f.laserCloud && !f.resident) {
            uploadCloud(ctx_, f.slot, *f.laserCloud, f.n_velo);
            f.resident = true;
        }
    }
    // Sophus::SO3d(Q).log() / SO3d::exp(phi).unit_quaternion()
    static void quat_log(const Quaterniond& q, double* w) {
        const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, n = std::sqrt(n2);
        double k;
        if (n2 < 1e-20)
            k = 2.0 / q.w - (2.0 / 3.0) * n2 / (q.w * q.w * q.w);
        else if (std::fabs(q.w) < 1e-10)
            k = (q.w > 0 ? M_PI : -M_PI) / n;
        else
            k = 2.0 * std::atan(n / q.w) / n;
        w[0] = k * q.x;
        w[1] = k * q.y;
        w[2] = k * q.z;
    }
    static void quat_exp(const double* w, Quaterniond& q) {
        const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        double im, re;
        if (th2 < 1e-20) {
            im = 0.5 - th2 / 48.0;
            re = 1.0 - th2 / 8.0;
        } else {
            const double th = std::sqrt(th2);
            im = std::sin(0.5 * th) / th;
            re = std::cos(0.5 * th);
        }
        q.x = im * w[0];
        q.y = im * w[1];
        q.z = im * w[2];
        q.w = re;
    }
    // transformTobeMapped = [Q exRbl | Q exPbl + P] (:1266-1268)
    static void body_to_lidar_world(const LidarFrame& f, const Matrix4d& exTlb, double* T) {
        const double x = f.Q.x, y = f.Q.y, z = f.Q.z, w = f.Q.w;
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
        double exRbl[9], exPbl[3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) exRbl[3 * r + c] = exTlb.m[4 * c + r];
        for (int r = 0; r < 3; ++r)
            exPbl[r] = -1.0 * ((exRbl[3 * r] * exTlb.m[3] + exRbl[3 * r + 1] * exTlb.m[7]) + exRbl[3 * r + 2] * exTlb.m[11]);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                T[4 * r + c] = (R[3 * r] * exRbl[c] + R[3 * r + 1] * exRbl[3 + c]) + R[3 * r + 2] * exRbl[6 + c];
            T[4 * r + 3] = ((R[3 * r] * exPbl[0] + R[3 * r + 1] * exPbl[1]) + R[3 * r + 2] * exPbl[2]) + f.P.v[r];
        }
        T[12] = T[13] = T[14] = 0;
        T[15] = 1;
    }
    Context& ctx_;
    mml_prior prior_;
    bool have_prior_ = false;
    int n_corner_map_ = 0, n_surf_map_ = 0;
    int key_scans_ = 0;  // scans that entered the local map (the key-scan rule of :1121-1135)
    double last_velo_update_pose_[3] = {-1.0, -1.0, -1.0};  // Estimator.h:339-340
    double last_hori_update_pose_[3] = {-1.0, -1.0, -1.0};
    bool _fail_detected = false;
};

// ---- LidarsParamEstimator (unionLidarsAligner.cpp): the numeric core of estimate_timeoffset (:1077-1153) ----------------
// velo_fov / livox: packed x, y, z floats (the FOV-cropped Velodyne cloud of :1080 and the merged Livox points of
// :1053-1068); velo_hori_tf: _velo_hori_tf_matrix, row-major.  Returns the start index of the best window in the merged
// Livox array (cnt * resolution, the argument of livox_msg_offset_vec at :1143), or -1 when no window beats the 1e6
// start value; *lowest_error is what :1155 compares with _time_esti_error_th.
struct TimeOffsetResult {
    int n_windows = 0, best_window = -1;
    double lowest_error = 1000000.0;
};
inline TimeOffsetResult EstimateTimeOffsetCore(Context& ctx, const float* velo_fov, int n_velo, const float* velo_hori_tf,
                                               const float* livox, int n_livox, int offset_search_resolution = 30,
                                               int offset_search_sliced_points = 12000) {
    TimeOffsetResult r;
    check(ctx.get(), mml_time_offset_search(ctx.get(), velo_fov, n_velo, velo_hori_tf, livox, n_livox, offset_search_resolution,
                                            offset_search_sliced_points, nullptr, nullptr, 0, &r.n_windows, &r.best_window,
                                            &r.lowest_error),
          "mml_time_offset_search");
    return r;
}

}  // namespace mml
#endif
