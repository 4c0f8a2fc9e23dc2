// adapter_gpu_probe.cpp -- drives multi-modal-loam_amd/host/mmloam_adapter.hpp the way the two reference nodes would,
// on a real device; tests/test_gpu_adapter.py compiles it, feeds it a binary scene file and compares what it prints
// with the oracle loop.  Two processes of the reference become two contexts here:
//   feature node  (unionFeatureExtract.cpp:266-321)  ctxA: unionCloud -> labelled fused cloud (host PointCloud = the
//                                                    payload of /union_feature_cloud)
//   pose node     (unionPoseEstimation.cpp:679-688,  ctxB: the cloud comes in from the "topic" (mml_cloud_upload inside
//                  :862, :872)                        RemoveLidarDistortion(cloud, ...) / LidarFrame::laserCloud),
//                                                    EstimateLidarPose / EstimateFullWindow
// Scene file (little endian): magic int 0x4d4d4c31, int mode (0 odometry, 1 full window), int n_scans, then per scan
//   int n_velo, float[4 n_velo], int n_livox, 20-byte CustomPoint[n_livox], double dR[9], dt[3], P[3], Q[4], V[3],
//   int n_imu, double[7 n_imu]; then int n_corner, float[3 n], int n_surf, float[3 n] (initial local map, may be 0);
//   mode 1: int n_windows, int W, then per window W scan indices.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mmloam_adapter.hpp"

struct Scan {
    std::vector<float> velo;
    std::vector<mml_livox_point> livox;
    double dR[9], dt[3], P[3], Q[4], V[3];
    std::vector<double> imu;
};

template <typename T>
static bool rd(FILE* f, T* p, size_t n) {
    return n == 0 || fread(p, sizeof(T), n, f) == n;
}

static unsigned long long label_hash(const mml::PointCloud& c) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < c.size(); ++i) {
        h ^= (unsigned long long)(int)c[i].normal_z + 3ull * (unsigned long long)(int)c[i].normal_y;
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int magic = 0, mode = 0, n_scans = 0;
    if (!rd(f, &magic, 1) || magic != 0x4d4d4c31 || !rd(f, &mode, 1) || !rd(f, &n_scans, 1)) return 2;
    std::vector<Scan> scans(n_scans);
    for (auto& s : scans) {
        int n = 0;
        if (!rd(f, &n, 1)) return 2;
        s.velo.resize(4 * (size_t)n);
        if (!rd(f, s.velo.data(), s.velo.size())) return 2;
        if (!rd(f, &n, 1)) return 2;
        s.livox.resize(n);
        if (!rd(f, s.livox.data(), s.livox.size())) return 2;
        if (!rd(f, s.dR, 9) || !rd(f, s.dt, 3) || !rd(f, s.P, 3) || !rd(f, s.Q, 4) || !rd(f, s.V, 3)) return 2;
        if (!rd(f, &n, 1)) return 2;
        s.imu.resize(7 * (size_t)n);
        if (!rd(f, s.imu.data(), s.imu.size())) return 2;
    }
    int nc = 0, ns = 0;
    std::vector<float> cmap, smap;
    if (!rd(f, &nc, 1)) return 2;
    cmap.resize(3 * (size_t)nc);
    if (!rd(f, cmap.data(), cmap.size()) || !rd(f, &ns, 1)) return 2;
    smap.resize(3 * (size_t)ns);
    if (!rd(f, smap.data(), smap.size())) return 2;

    try {
        mml::Context ctxA(1, 0);
        mml::feature_extraction fe(ctxA);
        bool rejected = false;
        try {
            mml::Estimator wrong(ctxA, 0.3f, 0.2f);  // leaf sizes that are not the context's must be refused
        } catch (const std::invalid_argument&) {
            rejected = true;
        }
        std::printf("ctor_rejects_mismatch %d\n", rejected ? 1 : 0);
        const mml::Matrix4d exTlb{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
        const mml::Vector3d gravity{{0.0, 0.0, -9.805}};
        if (mode == 0) {
            mml::Context ctxB(1, 0);
            mml::Estimator est(ctxB, 0.4f, 0.2f);
            for (int k = 0; k < n_scans; ++k) {
                Scan& s = scans[k];
                mml::PointCloud fused;
                mml_scan_info info;
                fe.unionCloud(s.velo.data(), (int)s.velo.size() / 4, s.livox.data(), (int)s.livox.size(), nullptr, fused, info);
                const unsigned long long h = label_hash(fused);
                // ---- the cloud crosses the topic; everything below only sees `fused` ----
                mml::Matrix3d dR;
                mml::Vector3d dt;
                for (int i = 0; i < 9; ++i) dR.m[i] = s.dR[i];
                for (int i = 0; i < 3; ++i) dt.v[i] = s.dt[i];
                mml::RemoveLidarDistortion(ctxB, fused, dR, dt, 0, info.n_velo);
                mml::Estimator::LidarFrame fr;
                fr.laserCloud = &fused;
                fr.n_velo = info.n_velo;
                fr.resident = false;  // re-upload the undistorted host cloud: what a LidarFrame carrying its points does
                fr.slot = 0;
                for (int i = 0; i < 3; ++i) fr.P.v[i] = s.P[i];
                fr.Q.x = s.Q[0];
                fr.Q.y = s.Q[1];
                fr.Q.z = s.Q[2];
                fr.Q.w = s.Q[3];
                std::list<mml::Estimator::LidarFrame> lst{fr};
                est.EstimateLidarPose(lst, exTlb, gravity, 2);
                const auto& o = lst.front();
                std::printf("scan %d n %d nv %d hash %llu fail %d P %.17g %.17g %.17g Q %.17g %.17g %.17g %.17g und0 %.9g %.9g %.9g rel0 %.9g\n", k,
                            info.n_points, info.n_velo, h, est.failureDetected() ? 1 : 0, o.P.v[0], o.P.v[1], o.P.v[2], o.Q.x, o.Q.y,
                            o.Q.z, o.Q.w, fused.empty() ? 0.f : fused[fused.size() / 2].x, fused.empty() ? 0.f : fused[fused.size() / 2].y,
                            fused.empty() ? 0.f : fused[fused.size() / 2].z, fused.empty() ? 0.f : fused[0].normal_x);
                if (k == n_scans - 1) {
                    // processPointToLine / processPointToPlanVec on the last scan at its estimated pose, and the cube store
                    // (threadMapIncrement: featureAssociateToMap + MapIncrement) behind get_corner_map / get_surf_map
                    mml::Matrix4d T{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
                    const double x = o.Q.x, y = o.Q.y, z = o.Q.z, w = o.Q.w;
                    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                                         2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
                    for (int r = 0; r < 3; ++r) {
                        for (int c = 0; c < 3; ++c) T.m[4 * r + c] = R[3 * r + c];
                        T.m[4 * r + 3] = o.P.v[r];
                    }
                    std::vector<mml::Estimator::FeatureLine> vl;
                    std::vector<mml::Estimator::FeaturePlanVec> vp;
                    bool deg = false;
                    est.thres_dist = 1.0;
                    est.processPointToLine(vl, 0, T);
                    est.processPointToPlanVec(vp, 0, T, deg);
                    int nlv = 0, npv = 0;
                    double el = 0, ep = 0;
                    for (auto& l : vl) { nlv += l.valid; el += l.error; }
                    for (auto& pl : vp) { npv += pl.valid; ep += pl.error; }
                    est.appendToGlobalMap(o, T);
                    est.incrementGlobalMap(T);
                    est.incrementGlobalMap(T);  // the map Estimate() sees lags one update behind the live store
                    std::printf("assoc lines %zu valid %d err %.12g planes %zu valid %d err %.12g deg %d gmap %zu %zu\n", vl.size(), nlv, el, vp.size(),
                                npv, ep, deg ? 1 : 0, est.get_corner_map().size(), est.get_surf_map().size());
                }
            }
        } else {
            int n_windows = 0, W = 0;
            if (!rd(f, &n_windows, 1) || !rd(f, &W, 1)) return 2;
            mml::Context ctxB(W, 0);
            mml::Estimator est(ctxB, 0.4f, 0.2f);
            est.setLocalMap(cmap.data(), nc, smap.data(), ns);
            for (int w = 0; w < n_windows; ++w) {
                std::vector<int> idx(W);
                if (!rd(f, idx.data(), W)) return 2;
                std::vector<mml::PointCloud> clouds(W);
                std::vector<mml::Estimator::LidarFrame> frames(W);
                std::vector<mml::Estimator::LidarFrame*> fp(W);
                std::vector<mml_imu_preint> imu(W);
                const double z3[3] = {0, 0, 0};
                for (int j = 0; j < W; ++j) {
                    Scan& s = scans[idx[j]];
                    mml_scan_info info;
                    fe.unionCloud(s.velo.data(), (int)s.velo.size() / 4, s.livox.data(), (int)s.livox.size(), nullptr, clouds[j], info);
                    auto& fr = frames[j];
                    fr.laserCloud = &clouds[j];
                    fr.n_velo = info.n_velo;
                    fr.slot = j;
                    for (int i = 0; i < 3; ++i) {
                        fr.P.v[i] = s.P[i];
                        fr.V.v[i] = s.V[i];
                    }
                    fr.Q.x = s.Q[0];
                    fr.Q.y = s.Q[1];
                    fr.Q.z = s.Q[2];
                    fr.Q.w = s.Q[3];
                    fp[j] = &fr;
                    if (j > 0 && mml_imu_preintegrate(s.imu.data(), (int)s.imu.size() / 7, z3, z3, &imu[j]) != MML_OK) return 3;
                }
                est.EstimateFullWindow(fp, imu, exTlb, gravity);
                for (int j = 0; j < W; ++j) {
                    const auto& o = frames[j];
                    std::printf("window %d frame %d P %.17g %.17g %.17g Q %.17g %.17g %.17g %.17g V %.17g %.17g %.17g bg %.17g %.17g %.17g ba %.17g %.17g %.17g\n",
                                w, j, o.P.v[0], o.P.v[1], o.P.v[2], o.Q.x, o.Q.y, o.Q.z, o.Q.w, o.V.v[0], o.V.v[1], o.V.v[2], o.bg.v[0],
                                o.bg.v[1], o.bg.v[2], o.ba.v[0], o.ba.v[1], o.ba.v[2]);
                }
            }
        }
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
    std::printf("ADAPTER_PROBE_DONE\n");
    return 0;
}
