// tools/live_loop.cpp -- BASELINE configs[2] from the reference's own host language: scans replayed ONE AT A TIME through the
// C++ adapter (multi-modal-loam_amd/host/mmloam_adapter.hpp) exactly as the two reference nodes would drive it, timed with
// std::chrono around every scan.  No Python in the loop: what bench.py --config 2 measures through ctypes + numpy includes ~0.4 ms
// of interpreter per scan; this is the latency a catkin node linking libmmloam_hip.so would see.
//   per scan:  mml_scan_upload (the message's buffers) -> mml_extract -> RemoveLidarDistortion(slot) -> Estimator::EstimateLidarPose
//              (down-sample, Estimate = 5 outer x 10 inner with re-association, key-scan rule, MapIncrementLocal on the device)
//              [unionFeatureExtract.cpp:266-321, unionPoseEstimation.cpp:630-933, Estimator.cpp:967-1140]
//   then, once 8 scans are in: the joint 8-scan sliding-window solve (mml_associate of the window at thres_dist 1 + mml_solve
//              window = 8: Estimator.cpp:1143-1581 with lidar factors only)
//   and, separately: mml_step on one slot (the configs[1] step at B = 1).
// Scene file: the format of tests/cpp/adapter_gpu_probe.cpp, mode 0 (bench.py writes it: scans, sweep motions, predicted poses).
// Output: one JSON object on stdout.
//   g++ -std=c++17 -O2 -I include -I multi-modal-loam_amd/host tools/live_loop.cpp -o live_loop -L multi-modal-loam_amd -lmmloam_hip
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mmloam_adapter.hpp"

struct Scan {
    std::vector<float> velo;
    std::vector<mml_livox_point> livox;
    double dR[9], dt[3], P[3], Q[4], V[3];
};

template <typename T>
static bool rd(FILE* f, T* p, size_t n) {
    return n == 0 || fread(p, sizeof(T), n, f) == n;
}
static double pct(std::vector<double> v, double q) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * (double)(v.size() - 1) + 0.5))];
}
static void pose_to_x(const double* P, const mml::Quaterniond& q, double* x) {  // [t, rotation vector]
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, n = std::sqrt(n2);
    const double k = n2 < 1e-20 ? 2.0 / q.w : 2.0 * std::atan2(n, q.w) / n;
    x[0] = P[0], x[1] = P[1], x[2] = P[2], x[3] = k * q.x, x[4] = k * q.y, x[5] = k * q.z;
}
static void x_to_T(const double* x, double* T) {
    const double th2 = x[3] * x[3] + x[4] * x[4] + x[5] * x[5], th = std::sqrt(th2);
    const double a = th2 < 1e-20 ? 1.0 : std::sin(th) / th, b = th2 < 1e-20 ? 0.5 : (1.0 - std::cos(th)) / th2;
    const double K[9] = {0, -x[5], x[4], x[5], 0, -x[3], -x[4], x[3], 0};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            const double k2 = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
            T[4 * r + c] = (r == c ? 1.0 : 0.0) + a * K[3 * r + c] + b * k2;
        }
        T[4 * r + 3] = x[r];
    }
    T[12] = T[13] = T[14] = 0, T[15] = 1;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int magic = 0, mode = 0, n_scans = 0;
    if (!rd(f, &magic, 1) || magic != 0x4d4d4c31 || !rd(f, &mode, 1) || mode != 0 || !rd(f, &n_scans, 1)) return 2;
    std::vector<Scan> scans(n_scans);
    for (auto& s : scans) {
        int n = 0;
        if (!rd(f, &n, 1)) return 2;
        s.velo.resize(4 * (size_t)n);
        if (!rd(f, s.velo.data(), s.velo.size()) || !rd(f, &n, 1)) return 2;
        s.livox.resize(n);
        if (!rd(f, s.livox.data(), s.livox.size())) return 2;
        if (!rd(f, s.dR, 9) || !rd(f, s.dt, 3) || !rd(f, s.P, 3) || !rd(f, s.Q, 4) || !rd(f, s.V, 3) || !rd(f, &n, 1)) return 2;
        std::vector<double> imu(7 * (size_t)n);
        if (!rd(f, imu.data(), imu.size())) return 2;
    }
    fclose(f);
    const int W = 8;
    try {
        const mml::Matrix4d exTlb{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
        const mml::Vector3d gravity{{0.0, 0.0, -9.805}};
        const double T_bl[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        mml::Context ctx(W, 0);
        mml_ctx* c = ctx.get();
        std::vector<double> lat, lat_win, lat_parts[4];
        double total_s = 0.0, worst = 0.0;
        int key_scans = 0, n_timed = 0;
        for (int rep = 0; rep <= reps; ++rep) {  // rep 0: warm-up (allocations of the map upkeep)
            mml::check(c, mml_map_local_reset(c), "map reset");
            mml::Estimator est(ctx, 0.4f, 0.2f);
            std::vector<double> xs(6 * (size_t)W, 0.0);
            const auto t_all = std::chrono::steady_clock::now();
            for (int i = 0; i < n_scans; ++i) {
                Scan& s = scans[i];
                const int slot = i % W;
                const auto t1 = std::chrono::steady_clock::now();
                mml::check(c, mml_scan_upload(c, slot, s.velo.data(), (int)s.velo.size() / 4, s.livox.data(), (int)s.livox.size()), "upload");
                mml::check(c, mml_extract(c, slot, 1, nullptr), "extract");
                const auto ta = std::chrono::steady_clock::now();
                mml::check(c, mml_undistort(c, slot, 1, s.dR, s.dt), "undistort");
                mml::Estimator::LidarFrame fr;
                fr.resident = true;
                fr.slot = slot;
                for (int k = 0; k < 3; ++k) fr.P.v[k] = s.P[k];
                fr.Q.x = s.Q[0], fr.Q.y = s.Q[1], fr.Q.z = s.Q[2], fr.Q.w = s.Q[3];
                std::list<mml::Estimator::LidarFrame> lst{fr};
                est.EstimateLidarPose(lst, exTlb, gravity, 2);
                const auto& o = lst.front();
                pose_to_x(o.P.v, o.Q, &xs[6 * (size_t)slot]);
                const auto t2 = std::chrono::steady_clock::now();
                if (i + 1 >= W) {  // the 8-scan sliding window, every frame re-associated at its current pose
                    std::vector<double> Tw(16 * (size_t)W);
                    for (int w = 0; w < W; ++w) x_to_T(&xs[6 * (size_t)w], &Tw[16 * (size_t)w]);
                    mml::check(c, mml_associate(c, 0, W, Tw.data(), 1.0, nullptr), "window associate");
                    mml_solve_opts so;
                    so.max_num_iterations = 10;
                    so.fixed_iterations = 0;
                    so.huber_delta = 0.0;
                    so.plan_weight_tan = 3e-4;
                    mml::check(c, mml_solve(c, 0, W, W, T_bl, &so, xs.data(), nullptr, nullptr), "window solve");
                }
                const auto t3 = std::chrono::steady_clock::now();
                if (rep > 0) {
                    lat.push_back(std::chrono::duration<double, std::milli>(t3 - t1).count());
                    lat_parts[0].push_back(std::chrono::duration<double, std::milli>(ta - t1).count());
                    lat_parts[1].push_back(std::chrono::duration<double, std::milli>(t2 - ta).count());
                    if (i + 1 >= W) lat_win.push_back(std::chrono::duration<double, std::milli>(t3 - t2).count());
                    // (P of the scene file is the ground truth perturbed by a few cm: the estimate must come back inside that)
                    for (int k = 0; k < 3; ++k) worst = std::max(worst, std::fabs(xs[6 * (size_t)slot + k] - s.P[k]));
                }
            }
            if (rep > 0) {
                total_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
                n_timed += n_scans;
            }
            key_scans = est.keyScans();
        }
        // the configs[1] step on ONE slot (extract -> undistort -> down-sample -> one association pass -> 10 fixed iterations)
        std::vector<double> lat1;
        {
            Scan& s = scans[n_scans - 1];
            double x[6];
            mml::Quaterniond q;
            q.x = s.Q[0], q.y = s.Q[1], q.z = s.Q[2], q.w = s.Q[3];
            for (int it = 0; it < 320; ++it) {
                pose_to_x(s.P, q, x);
                const auto t1 = std::chrono::steady_clock::now();
                mml::check(c, mml_step(c, (n_scans - 1) % W, 1, s.dR, s.dt, exTlb.m, 25.0, 10, x), "step");
                if (it >= 20) lat1.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            }
        }
        std::printf("{\"host\": \"C++ adapter (tools/live_loop.cpp)\", \"scans\": %d, \"reps\": %d, \"scans_per_s\": %.1f, "
                    "\"per_scan_p50_ms\": %.4f, \"per_scan_p99_ms\": %.4f, \"per_scan_max_ms\": %.4f, "
                    "\"upload_extract_p50_ms\": %.4f, \"undistort_estimate_p50_ms\": %.4f, \"window8_part_p50_ms\": %.4f, "
                    "\"configs1_step_B1_p50_ms\": %.4f, \"configs1_step_B1_p99_ms\": %.4f, \"key_scans\": %d, "
                    "\"max_abs_dP_vs_prediction_m\": %.4f}\n",
                    n_scans, reps, (double)n_timed / total_s, pct(lat, 0.5), pct(lat, 0.99), pct(lat, 1.0), pct(lat_parts[0], 0.5),
                    pct(lat_parts[1], 0.5), pct(lat_win, 0.5), pct(lat1, 0.5), pct(lat1, 0.99), key_scans, worst);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "live_loop: %s\n", e.what());
        return 1;
    }
    return 0;
}
