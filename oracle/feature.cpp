// oracle/feature.cpp -- CPU restatement of the feature-extraction half of the hot path.
// TEST INFRASTRUCTURE ONLY (see mml_oracle.h).  Citations: /root/reference/mm-loam/src/unionFeatureExtract.cpp
//
// Floating-point conventions (the reference is x86-64 gcc -O3, SSE2, no FMA):
//   * float expressions are evaluated in float, left to right, exactly as written there;
//   * unqualified sqrt()/atan()/atan2() on float arguments are the FLOAT functions -- sqrtf / atanf / atan2f of glibc: the TU
//     includes lidars_extrinsic_cali.h (:58), that <tf/tf.h> (lidars_extrinsic_cali.h:3) and tf/LinearMath/Scalar.h <math.h>,
//     which with libstdc++ >= 6 (melodic: gcc 7.5) puts std::atan2(float, float) ... into the global namespace, where overload
//     resolution prefers them (tests/test_oracle.py::test_libm_overloads_resolve_to_float).  atanf / atan2f are restated in
//     libm_f32.h (glibc's fdlibm float routines, pinned against this image's libm on all 2^32 / 4e8 arguments); sqrtf is
//     correctly rounded, std::sqrt(float) here.  The same convention as oracle/estimate.cpp (the aligner's TU);
//   * Eigen::Vector3d dot()/norm() reduce as (x0+x1)+x2 (Eigen 3.3 SSE2 linear-vectorised redux,
//     PacketSize 2), normalize() divides each coefficient by the norm and leaves a zero vector alone.
// Build with -ffp-contract=off.
#include "mml_oracle.h"
#include "threads.h"
#include "libm_f32.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct P4 {
    float x, y, z, intensity;
};

struct V3d {
    double x, y, z;
};
inline V3d v3(double x, double y, double z) { return V3d{x, y, z}; }
inline double dot(const V3d& a, const V3d& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline double norm(const V3d& a) { return std::sqrt(dot(a, a)); }
inline V3d sub(const V3d& a, const V3d& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline void normalize(V3d& a) {
    double z = dot(a, a);
    if (z > 0.0) {
        double n = std::sqrt(z);
        a.x /= n;
        a.y /= n;
        a.z /= n;
    }
}

}  // namespace

// unionFeatureExtract.cpp:341-844  feature_extraction::detectFeaturePoints
// Parity conventions for reference UB: CloudFeatureFlag[] and
// cloudAngle[] are zero-initialised; thNumCurvSize used at :492/:505 is the value left by the
// last stencil iteration (i = n-6), 2 when the stencil loop never runs; inputs are finite so the
// compaction at :369-388 is the identity.
extern "C" void mmlo_detect_feature_points(const float* pts_, int n, int* sharp, int* n_sharp, int* flat,
                                            int* n_flat, int* flags_out) {
    const P4* pt = reinterpret_cast<const P4*>(pts_);
    std::vector<int> CloudFeatureFlag(n > 0 ? n : 1, 0);
    std::vector<float> cloudCurvature(n > 0 ? n : 1, 0.f);
    std::vector<float> cloudDepth(n > 0 ? n : 1, 0.f);
    std::vector<int> cloudSortInd(n > 0 ? n : 1, 0);
    std::vector<float> cloudReflect(n > 0 ? n : 1, 0.f);
    std::vector<int> reflectSortInd(n > 0 ? n : 1, 0);
    std::vector<int> cloudAngle(n > 0 ? n : 1, 0);

    int thNumCurvSize = 2;            // :353
    float thDistanceFaraway = 50.0;   // :354
    int thNumFlat = 1;                // :355
    int thPartNum = 50;               // :356
    float thFlatThreshold = 0.02;     // :357
    float thLidarNearestDis = 1.0;    // :358
    float thBreakCornerDis = 1;       // :359

    int cloudSize = n;
    int count_num = 1;
    bool left_surf_flag = false;
    bool right_surf_flag = false;
    int scanStartInd = 5;
    int scanEndInd = cloudSize - 6;

    // ---- :407-451 stencil -------------------------------------------------------------------
    for (int i = 5; i < cloudSize - 5; i++) {
        float diffX = 0;
        float diffY = 0;
        float diffZ = 0;

        float dis = std::sqrt(pt[i].x * pt[i].x + pt[i].y * pt[i].y + pt[i].z * pt[i].z);

        V3d pt_last = v3(pt[i - 1].x, pt[i - 1].y, pt[i - 1].z);
        V3d pt_cur = v3(pt[i].x, pt[i].y, pt[i].z);
        V3d pt_next = v3(pt[i + 1].x, pt[i + 1].y, pt[i + 1].z);

        V3d dl = sub(pt_last, pt_cur);
        V3d dn = sub(pt_next, pt_cur);
        double angle_last = dot(dl, pt_cur) / (norm(dl) * norm(pt_cur));
        double angle_next = dot(dn, pt_cur) / (norm(dn) * norm(pt_cur));

        if (dis > thDistanceFaraway || (fabs(angle_last) > 0.966 && fabs(angle_next) > 0.966)) {
            thNumCurvSize = 2;
        } else {
            thNumCurvSize = 3;
        }

        if (fabs(angle_last) > 0.966 && fabs(angle_next) > 0.966) {
            cloudAngle[i] = 1;
        }

        float diffR = -2 * thNumCurvSize * pt[i].intensity;
        for (int j = 1; j <= thNumCurvSize; ++j) {
            diffX += pt[i - j].x + pt[i + j].x;
            diffY += pt[i - j].y + pt[i + j].y;
            diffZ += pt[i - j].z + pt[i + j].z;
            diffR += pt[i - j].intensity + pt[i + j].intensity;
        }
        diffX -= 2 * thNumCurvSize * pt[i].x;
        diffY -= 2 * thNumCurvSize * pt[i].y;
        diffZ -= 2 * thNumCurvSize * pt[i].z;

        cloudDepth[i] = dis;
        cloudCurvature[i] = diffX * diffX + diffY * diffY + diffZ * diffZ;
        cloudSortInd[i] = i;
        cloudReflect[i] = diffR;
        reflectSortInd[i] = i;
    }

    // ---- :453-541 partitions ---------------------------------------------------------------
    for (int j = 0; j < thPartNum; j++) {
        int sp = scanStartInd + (scanEndInd - scanStartInd) * j / thPartNum;
        int ep = scanStartInd + (scanEndInd - scanStartInd) * (j + 1) / thPartNum - 1;

        // :458-467 stable insertion sort by curvature
        for (int k = sp + 1; k <= ep; k++) {
            for (int l = k; l >= sp + 1; l--) {
                if (cloudCurvature[cloudSortInd[l]] < cloudCurvature[cloudSortInd[l - 1]]) {
                    int temp = cloudSortInd[l - 1];
                    cloudSortInd[l - 1] = cloudSortInd[l];
                    cloudSortInd[l] = temp;
                }
            }
        }
        // :470-479 stable insertion sort by reflectivity difference
        for (int k = sp + 1; k <= ep; k++) {
            for (int l = k; l >= sp + 1; l--) {
                if (cloudReflect[reflectSortInd[l]] < cloudReflect[reflectSortInd[l - 1]]) {
                    int temp = reflectSortInd[l - 1];
                    reflectSortInd[l - 1] = reflectSortInd[l];
                    reflectSortInd[l] = temp;
                }
            }
        }

        int smallestPickedNum = 1;
        int sharpestPickedNum = 1;
        // :483-519
        for (int k = sp; k <= ep; k++) {
            int ind = cloudSortInd[k];

            if (CloudFeatureFlag[ind] != 0) continue;

            if (cloudCurvature[ind] < thFlatThreshold * cloudDepth[ind] * thFlatThreshold * cloudDepth[ind]) {
                CloudFeatureFlag[ind] = 3;

                for (int l = 1; l <= thNumCurvSize; l++) {
                    float diffX = pt[ind + l].x - pt[ind + l - 1].x;
                    float diffY = pt[ind + l].y - pt[ind + l - 1].y;
                    float diffZ = pt[ind + l].z - pt[ind + l - 1].z;
                    if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.02 || cloudDepth[ind] > thDistanceFaraway) {
                        break;
                    }
                    CloudFeatureFlag[ind + l] = 1;
                }
                for (int l = -1; l >= -thNumCurvSize; l--) {
                    float diffX = pt[ind + l].x - pt[ind + l + 1].x;
                    float diffY = pt[ind + l].y - pt[ind + l + 1].y;
                    float diffZ = pt[ind + l].z - pt[ind + l + 1].z;
                    if (diffX * diffX + diffY * diffY + diffZ * diffZ > 0.02 || cloudDepth[ind] > thDistanceFaraway) {
                        break;
                    }
                    CloudFeatureFlag[ind + l] = 1;
                }
            }
        }
        // :521-539
        for (int k = sp; k <= ep; k++) {
            int ind = cloudSortInd[k];
            if (((CloudFeatureFlag[ind] == 3) && (smallestPickedNum <= thNumFlat)) ||
                ((CloudFeatureFlag[ind] == 3) && (cloudDepth[ind] > thDistanceFaraway)) || cloudAngle[ind] == 1) {
                smallestPickedNum++;
                CloudFeatureFlag[ind] = 2;
            }

            int idx = reflectSortInd[k];
            if (cloudCurvature[idx] < 0.7 * thFlatThreshold * cloudDepth[idx] * thFlatThreshold * cloudDepth[idx] &&
                sharpestPickedNum <= 3 && cloudReflect[idx] > 20.0) {
                sharpestPickedNum++;
                CloudFeatureFlag[idx] = 300;
            }
        }
    }

    // ---- :543-650 plane-intersection corners (flag 150), data-dependent stride ---------------
    for (int i = 5; i < cloudSize - 5; i += count_num) {
        float depth = std::sqrt(pt[i].x * pt[i].x + pt[i].y * pt[i].y + pt[i].z * pt[i].z);
        float ldiffX = pt[i - 4].x + pt[i - 3].x - 4 * pt[i - 2].x + pt[i - 1].x + pt[i].x;
        float ldiffY = pt[i - 4].y + pt[i - 3].y - 4 * pt[i - 2].y + pt[i - 1].y + pt[i].y;
        float ldiffZ = pt[i - 4].z + pt[i - 3].z - 4 * pt[i - 2].z + pt[i - 1].z + pt[i].z;
        float left_curvature = ldiffX * ldiffX + ldiffY * ldiffY + ldiffZ * ldiffZ;

        if (left_curvature < thFlatThreshold * depth) {
            left_surf_flag = true;
        } else {
            left_surf_flag = false;
        }

        float rdiffX = pt[i + 4].x + pt[i + 3].x - 4 * pt[i + 2].x + pt[i + 1].x + pt[i].x;
        float rdiffY = pt[i + 4].y + pt[i + 3].y - 4 * pt[i + 2].y + pt[i + 1].y + pt[i].y;
        float rdiffZ = pt[i + 4].z + pt[i + 3].z - 4 * pt[i + 2].z + pt[i + 1].z + pt[i].z;
        float right_curvature = rdiffX * rdiffX + rdiffY * rdiffY + rdiffZ * rdiffZ;

        if (right_curvature < thFlatThreshold * depth) {
            count_num = 4;
            right_surf_flag = true;
        } else {
            count_num = 1;
            right_surf_flag = false;
        }

        if (left_surf_flag && right_surf_flag) {
            V3d norm_left = v3(0, 0, 0);
            V3d norm_right = v3(0, 0, 0);
            for (int k = 1; k < 5; k++) {
                V3d tmp = v3(pt[i - k].x - pt[i].x, pt[i - k].y - pt[i].y, pt[i - k].z - pt[i].z);
                normalize(tmp);
                norm_left.x += (k / 10.0) * tmp.x;
                norm_left.y += (k / 10.0) * tmp.y;
                norm_left.z += (k / 10.0) * tmp.z;
            }
            for (int k = 1; k < 5; k++) {
                V3d tmp = v3(pt[i + k].x - pt[i].x, pt[i + k].y - pt[i].y, pt[i + k].z - pt[i].z);
                normalize(tmp);
                norm_right.x += (k / 10.0) * tmp.x;
                norm_right.y += (k / 10.0) * tmp.y;
                norm_right.z += (k / 10.0) * tmp.z;
            }
            double cc = fabs(dot(norm_left, norm_right) / (norm(norm_left) * norm(norm_right)));
            V3d last_tmp = v3(pt[i - 4].x - pt[i].x, pt[i - 4].y - pt[i].y, pt[i - 4].z - pt[i].z);
            V3d current_tmp = v3(pt[i + 4].x - pt[i].x, pt[i + 4].y - pt[i].y, pt[i + 4].z - pt[i].z);
            double last_dis = norm(last_tmp);
            double current_dis = norm(current_tmp);

            if (cc < 0.5 && last_dis > 0.05 && current_dis > 0.05) {
                CloudFeatureFlag[i] = 150;
            }
        }
    }

    // ---- :651-806 break-point corners (flag 100 / 101) ---------------------------------------
    for (int i = 5; i < cloudSize - 5; i++) {
        float diff_left[2];
        float diff_right[2];

        for (int count = 1; count < 3; count++) {
            float diffX1 = pt[i + count].x - pt[i].x;
            float diffY1 = pt[i + count].y - pt[i].y;
            float diffZ1 = pt[i + count].z - pt[i].z;
            diff_right[count - 1] = std::sqrt(diffX1 * diffX1 + diffY1 * diffY1 + diffZ1 * diffZ1);

            float diffX2 = pt[i - count].x - pt[i].x;
            float diffY2 = pt[i - count].y - pt[i].y;
            float diffZ2 = pt[i - count].z - pt[i].z;
            diff_left[count - 1] = std::sqrt(diffX2 * diffX2 + diffY2 * diffY2 + diffZ2 * diffZ2);
        }

        float depth_right = std::sqrt(pt[i + 1].x * pt[i + 1].x + pt[i + 1].y * pt[i + 1].y + pt[i + 1].z * pt[i + 1].z);
        float depth_left = std::sqrt(pt[i - 1].x * pt[i - 1].x + pt[i - 1].y * pt[i - 1].y + pt[i - 1].z * pt[i - 1].z);

        if (fabs(diff_right[0] - diff_left[0]) > thBreakCornerDis) {
            if (diff_right[0] > diff_left[0]) {
                V3d surf_vector = v3(pt[i - 1].x - pt[i].x, pt[i - 1].y - pt[i].y, pt[i - 1].z - pt[i].z);
                V3d lidar_vector = v3(pt[i].x, pt[i].y, pt[i].z);
                double cc = fabs(dot(surf_vector, lidar_vector) / (norm(surf_vector) * norm(lidar_vector)));
                if (cc < 0.95) {
                    if (depth_right > depth_left) {
                        CloudFeatureFlag[i] = 100;
                    } else {
                        if (depth_right == 0) CloudFeatureFlag[i] = 100;
                    }
                }
            } else {
                V3d surf_vector = v3(pt[i + 1].x - pt[i].x, pt[i + 1].y - pt[i].y, pt[i + 1].z - pt[i].z);
                V3d lidar_vector = v3(pt[i].x, pt[i].y, pt[i].z);
                double cc = fabs(dot(surf_vector, lidar_vector) / (norm(surf_vector) * norm(lidar_vector)));
                if (cc < 0.95) {
                    if (depth_right < depth_left) {
                        CloudFeatureFlag[i] = 100;
                    } else {
                        if (depth_left == 0) CloudFeatureFlag[i] = 100;
                    }
                }
            }
        }

        // :756-804 (the back loop reads the depth of i-k, not i+k: preserved, :782-784)
        if (CloudFeatureFlag[i] == 100) {
            V3d norm_front = v3(0, 0, 0);
            V3d norm_back = v3(0, 0, 0);
            for (int k = 1; k < 4; k++) {
                float temp_depth = std::sqrt(pt[i - k].x * pt[i - k].x + pt[i - k].y * pt[i - k].y + pt[i - k].z * pt[i - k].z);
                if (temp_depth < 1) {
                    continue;
                }
                V3d tmp = v3(pt[i - k].x - pt[i].x, pt[i - k].y - pt[i].y, pt[i - k].z - pt[i].z);
                normalize(tmp);
                norm_front.x += (k / 6.0) * tmp.x;
                norm_front.y += (k / 6.0) * tmp.y;
                norm_front.z += (k / 6.0) * tmp.z;
            }
            for (int k = 1; k < 4; k++) {
                float temp_depth = std::sqrt(pt[i - k].x * pt[i - k].x + pt[i - k].y * pt[i - k].y + pt[i - k].z * pt[i - k].z);
                if (temp_depth < 1) {
                    continue;
                }
                V3d tmp = v3(pt[i + k].x - pt[i].x, pt[i + k].y - pt[i].y, pt[i + k].z - pt[i].z);
                normalize(tmp);
                norm_back.x += (k / 6.0) * tmp.x;
                norm_back.y += (k / 6.0) * tmp.y;
                norm_back.z += (k / 6.0) * tmp.z;
            }
            double cc = fabs(dot(norm_front, norm_back) / (norm(norm_front) * norm(norm_back)));
            if (cc < 0.95) {
            } else {
                CloudFeatureFlag[i] = 101;
            }
        }
    }

    // ---- :818-842 emit ----------------------------------------------------------------------
    int ns = 0, nf = 0;
    for (int i = 5; i < cloudSize - 5; i++) {
        float dis = pt[i].x * pt[i].x + pt[i].y * pt[i].y + pt[i].z * pt[i].z;
        if (dis < thLidarNearestDis * thLidarNearestDis) continue;
        if (CloudFeatureFlag[i] == 2) {
            flat[nf++] = i;
            continue;
        }
        if (CloudFeatureFlag[i] == 100 || CloudFeatureFlag[i] == 150) {
            sharp[ns++] = i;
        }
    }
    *n_sharp = ns;
    *n_flat = nf;
    if (flags_out) {
        for (int i = 0; i < n; i++) flags_out[i] = CloudFeatureFlag[i];
    }
    (void)left_surf_flag;
    (void)right_surf_flag;
}

namespace {

// lidars_extrinsic_cali.h:451-477 removeNearFarPoints
inline bool keep_near_far(float x, float y, float z, float near_thres, float far_thres) {
    float dis = x * x + y * y + z * z;
    if (dis < near_thres * near_thres || dis > far_thres * far_thres) return false;
    return true;
}
// lidars_extrinsic_cali.h:424-449 removeNearPointCloud
inline bool keep_near(float x, float y, float z, float thres) {
    if (x * x + y * y + z * z < thres * thres) return false;
    return true;
}

struct CombPt {
    float x, y, z, intensity, normal_x, normal_y, normal_z;
};

// Shared tail of getHoriFeatureExtract (:1001-1032) / getVeloFeature (:1209-1252): bucket by line,
// detect per line, scatter labels back through normal_z.
void detect_lines_and_label(std::vector<CombPt>& cloud, int n_lines, int threads) {
    std::vector<std::vector<P4>> vlines(n_lines);
    std::vector<std::vector<int>> vgidx(n_lines);
    for (size_t i = 0; i < cloud.size(); ++i) {
        int line_idx = int(cloud[i].normal_y);
        if (line_idx >= 0 && line_idx < n_lines) {
            vlines[line_idx].push_back(P4{cloud[i].x, cloud[i].y, cloud[i].z, cloud[i].intensity});
            vgidx[line_idx].push_back((int)i);
        }
    }
    // lines write disjoint points of `cloud`: one thread per line (:1008-1015) gives the serial result bit for bit
    auto one_line = [&](int l) {
        int n = (int)vlines[l].size();
        std::vector<int> corner(n > 0 ? n : 1), surf(n > 0 ? n : 1);
        int nc = 0, nsf = 0;
        mmlo_detect_feature_points(n ? &vlines[l][0].x : nullptr, n, corner.data(), &nc, surf.data(), &nsf, nullptr);
        for (int j = 0; j < nc; ++j) cloud[vgidx[l][corner[j]]].normal_z = 1.0;
        for (int j = 0; j < nsf; ++j) cloud[vgidx[l][surf[j]]].normal_z = 2.0;
    };
    if (threads > 1)
        mmlo::pool(threads).run(n_lines, one_line);
    else
        for (int l = 0; l < n_lines; ++l) one_line(l);
}

}  // namespace

// the two restated libm routines on arrays (tests: against this image's libm, and as the checker of the device's copies)
extern "C" void mmlo_atanf(const float* x, float* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = mmlo_libm::atanf_fdlibm(x[i]);
}
extern "C" void mmlo_atan2f(const float* y, const float* x, float* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = mmlo_libm::atan2f_fdlibm(y[i], x[i]);
}

// unionFeatureExtract.cpp:1113-1317 getVeloFeature
extern "C" int mmlo_extract_velo(const float* in_xyzi, int n_in, int n_rings, float pitch0_deg,
                                 float pitch_step_deg, float near_th, float far_th, float* out_xyzi,
                                 float* out_reltime, int* out_ring, int* out_label, int* n_corner, int* n_surf) {
    // :1133 pcl::removeNaNFromPointCloud (drops points with a non-finite x, y or z)
    std::vector<P4> in;
    in.reserve(n_in);
    for (int i = 0; i < n_in; ++i) {
        const float* p = in_xyzi + 4 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        in.push_back(P4{p[0], p[1], p[2], p[3]});
    }
    int cloudSize = (int)in.size();
    *n_corner = 0;
    *n_surf = 0;
    if (cloudSize == 0) return 0;

    float startOri = -mmlo_libm::atan2f_fdlibm(in[0].y, in[0].x);                        // :1136, float overload
    float endOri = -mmlo_libm::atan2f_fdlibm(in[cloudSize - 1].y, in[cloudSize - 1].x) + 2 * M_PI;  // float + double, rounded on assignment
    if (endOri - startOri > 3 * M_PI)
        endOri -= 2 * M_PI;
    else if (endOri - startOri < M_PI)
        endOri += 2 * M_PI;

    bool halfPassed = false;
    std::vector<CombPt> laserCloud;
    laserCloud.reserve(cloudSize);
    for (int i = 0; i < cloudSize; i++) {
        CombPt point;
        point.x = in[i].x;
        point.y = in[i].y;
        point.z = in[i].z;

        // :1159 with the float overloads: sqrtf, a float division, atanf, `* 180` in float (int -> float), `/ M_PI` in double
        float angle = mmlo_libm::atanf_fdlibm(point.z / std::sqrt(point.x * point.x + point.y * point.y)) * 180 / M_PI;
        int scanID = 0;
        // :1162 scanID = int((angle + 15) / 2 + 0.5), generalised: (angle - pitch0) / step
        // A (0,0,0) record -- a no-return of some drivers -- has angle = atan(0 / 0) = NaN, and int(NaN) is undefined in C++:
        // the reference binary (x86-64, cvttsd2si) gets INT_MIN, "integer indefinite", so the point fails the range test below
        // and is dropped.  Written out here so that it does not depend on what this compiler does with the cast.
        {
            const double v = (angle - pitch0_deg) / pitch_step_deg + 0.5;
            scanID = (v != v || v >= 2147483648.0 || v <= -2147483649.0) ? (-2147483647 - 1) : int(v);
        }
        if (scanID > (n_rings - 1) || scanID < 0) {
            continue;
        }

        float ori = -mmlo_libm::atan2f_fdlibm(point.y, point.x);  // :1168
        if (!halfPassed) {
            if (ori < startOri - M_PI / 2)
                ori += 2 * M_PI;
            else if (ori > startOri + M_PI * 3 / 2)
                ori -= 2 * M_PI;

            if (ori - startOri > M_PI) halfPassed = true;
        } else {
            ori += 2 * M_PI;
            if (ori < endOri - M_PI * 3 / 2)
                ori += 2 * M_PI;
            else if (ori > endOri + M_PI / 2)
                ori -= 2 * M_PI;
        }

        float relTime = (ori - startOri) / (endOri - startOri);
        point.normal_x = relTime;
        point.normal_y = scanID;
        point.normal_z = 0;
        point.intensity = in[i].intensity;
        laserCloud.push_back(point);
    }

    detect_lines_and_label(laserCloud, n_rings, 1);  // rings serially, :1228-1230

    // :1244-1256, :1278-1300
    int nc = 0, nsf = 0, nout = 0;
    for (const auto& p : laserCloud) {
        if (std::fabs(p.normal_z - 1.0) < 1e-5 && keep_near_far(p.x, p.y, p.z, near_th, far_th)) nc++;
        if (std::fabs(p.normal_z - 2.0) < 1e-5 && keep_near_far(p.x, p.y, p.z, near_th, far_th)) nsf++;
    }
    for (const auto& p : laserCloud) {
        if (!keep_near_far(p.x, p.y, p.z, near_th, far_th)) continue;
        out_xyzi[4 * nout + 0] = p.x;
        out_xyzi[4 * nout + 1] = p.y;
        out_xyzi[4 * nout + 2] = p.z;
        out_xyzi[4 * nout + 3] = 0;  // :1254-1256
        out_reltime[nout] = p.normal_x;
        out_ring[nout] = (int)p.normal_y;
        out_label[nout] = (int)p.normal_z;
        nout++;
    }
    *n_corner = nc;
    *n_surf = nsf;
    return nout;
}

// unionFeatureExtract.cpp:891-1035 getHoriFeature + getHoriFeatureExtract
extern "C" int mmlo_extract_livox(const mmlo_livox_point* in, int n, int n_lines, float near_th, float far_th,
                                  float* out_xyzi, float* out_reltime, int* out_line, int* out_label,
                                  int* n_corner, int* n_surf) {
    *n_corner = 0;
    *n_surf = 0;
    if (n == 0) return 0;
    // :985 ros::Time().fromNSec(t).toSec(): sec = t / 1e9 (integer), nsec = t % 1e9,
    // toSec() = (double)sec + 1e-9 * (double)nsec
    auto to_sec = [](uint32_t t) -> double {
        uint32_t sec = (uint32_t)(t / 1000000000ull);
        uint32_t nsec = (uint32_t)(t % 1000000000ull);
        return (double)sec + 1e-9 * (double)nsec;
    };
    double timeSpan = to_sec(in[n - 1].offset_time);
    std::vector<CombPt> laserCloud;
    laserCloud.reserve(n);
    for (int i = 0; i < n; ++i) {
        const mmlo_livox_point& p = in[i];
        int line_num = (int)p.line;
        if (line_num > n_lines - 1) continue;
        if (p.x < 0.01) continue;
        CombPt point;
        point.x = p.x;
        point.y = p.y;
        point.z = p.z;
        point.intensity = p.reflectivity;
        point.normal_x = to_sec(p.offset_time) / timeSpan;
        point.normal_y = line_num * 1.0;
        point.normal_z = 0;
        laserCloud.push_back(point);
    }

    detect_lines_and_label(laserCloud, n_lines, mmlo::threading().livox_line_threads);

    // :916 combined: removeNearFarPoints; :925/:933 surf/corner: removeNearPointCloud
    int nc = 0, nsf = 0, nout = 0;
    for (const auto& p : laserCloud) {
        if (std::fabs(p.normal_z - 1.0) < 1e-5 && keep_near(p.x, p.y, p.z, near_th)) nc++;
        if (std::fabs(p.normal_z - 2.0) < 1e-5 && keep_near(p.x, p.y, p.z, near_th)) nsf++;
    }
    for (const auto& p : laserCloud) {
        if (!keep_near_far(p.x, p.y, p.z, near_th, far_th)) continue;
        out_xyzi[4 * nout + 0] = p.x;
        out_xyzi[4 * nout + 1] = p.y;
        out_xyzi[4 * nout + 2] = p.z;
        out_xyzi[4 * nout + 3] = p.intensity;
        out_reltime[nout] = p.normal_x;
        out_line[nout] = (int)p.normal_y;
        out_label[nout] = (int)p.normal_z;
        nout++;
    }
    *n_corner = nc;
    *n_surf = nsf;
    return nout;
}
