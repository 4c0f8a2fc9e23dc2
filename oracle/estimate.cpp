// oracle/estimate.cpp -- CPU restatement of the pose-estimation half of the hot path.
// TEST INFRASTRUCTURE ONLY (see mml_oracle.h).  Citations relative to /root/reference/mm-loam/.
// Build with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "linalg.h"
#include "mml_oracle.h"
#include "threads.h"

extern "C" void mmlo_set_threading(int livox_line_threads, int solve_threads) {
    mmlo::threading().livox_line_threads = livox_line_threads < 1 ? 1 : livox_line_threads;
    mmlo::threading().solve_threads = solve_threads < 1 ? 1 : solve_threads;
}

using namespace mmlo;

static const double kLidarM = 1.5e-3;  // include/IMUIntegrator/IMUIntegrator.h:83 lidar_m

// ------------------------------------------------------------------------------------------------
// a9  unionPoseEstimation.cpp:402-421 RemoveLidarDistortion
// Eigen pieces restated: Quaterniond(Matrix3d), normalized(), slerp (Eigen 3.3.4 Quaternion.h),
// quaternion * vector, Matrix3d^T * Vector3d.
extern "C" void mmlo_undistort(float* xyz, const float* sarr, int n, const double* dR, const double* dt) {
    for (int i = 0; i < n; i++) {
        float s = sarr[i];
        Quat qlc = qnormalized(quat_from_matrix(dR));
        // Identity().slerp(s, qlc).normalized()
        const double one = 1.0 - std::numeric_limits<double>::epsilon();
        Quat id{0, 0, 0, 1};
        double d = qdot(id, qlc);
        double absD = std::fabs(d);
        double scale0, scale1;
        double t = s;
        if (absD >= one) {
            scale0 = 1.0 - t;
            scale1 = t;
        } else {
            double theta = std::acos(absD);
            double sinTheta = std::sin(theta);
            scale0 = std::sin((1.0 - t) * theta) / sinTheta;
            scale1 = std::sin((t * theta)) / sinTheta;
        }
        if (d < 0.0) scale1 = -scale1;
        Quat q{scale0 * id.x + scale1 * qlc.x, scale0 * id.y + scale1 * qlc.y, scale0 * id.z + scale1 * qlc.z,
               scale0 * id.w + scale1 * qlc.w};
        Quat delta_qlc = qnormalized(q);
        Vec3 delta_Plc = mk(s * dt[0], s * dt[1], s * dt[2]);
        Vec3 startP = qrot(delta_qlc, mk(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])) + delta_Plc;
        Vec3 v = startP - mk(dt[0], dt[1], dt[2]);
        // dRlc.transpose() * v
        double px = (dR[0] * v.x + dR[3] * v.y) + dR[6] * v.z;
        double py = (dR[1] * v.x + dR[4] * v.y) + dR[7] * v.z;
        double pz = (dR[2] * v.x + dR[5] * v.y) + dR[8] * v.z;
        xyz[3 * i] = px;
        xyz[3 * i + 1] = py;
        xyz[3 * i + 2] = pz;
    }
}

// ------------------------------------------------------------------------------------------------
// a10  pcl::VoxelGrid<PointXYZINormal>::applyFilter (PCL 1.8.1 filters/impl/voxel_grid.hpp), as used at
// Estimator.cpp:1013-1024 with leaf sizes from Estimator.cpp:78-80.  Only x,y,z are produced (the other
// fields are not read downstream on this path).  Convention: points of one voxel are summed in input order
// (PCL uses an unstable std::sort on the voxel index, so the in-voxel order is unspecified there).
extern "C" int mmlo_voxel_downsample(const float* xyz, int n, float leaf, float* out_xyz) {
    if (n == 0) return 0;
    float inv = 1.0f / leaf;  // inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()
    float mn[3], mx[3];
    for (int c = 0; c < 3; ++c) mn[c] = mx[c] = xyz[c];
    for (int i = 1; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            float v = xyz[3 * i + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
    int min_b[3], max_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(mn[c] * inv));
        max_b[c] = static_cast<int>(floor(mx[c] * inv));
        div_b[c] = max_b[c] - min_b[c] + 1;
    }
    // "Leaf size is too small for the input dataset. Integer indices would overflow." (voxel_grid.hpp applyFilter): when the bounding
    // box holds more than INT_MAX voxels PCL warns and returns the input cloud unfiltered
    {
        const int64_t dx = static_cast<int64_t>((mx[0] - mn[0]) * inv) + 1, dy = static_cast<int64_t>((mx[1] - mn[1]) * inv) + 1,
                      dz = static_cast<int64_t>((mx[2] - mn[2]) * inv) + 1;
        if ((dx * dy * dz) > static_cast<int64_t>(2147483647)) {
            memcpy(out_xyz, xyz, sizeof(float) * 3 * (size_t)n);
            return n;
        }
    }
    int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    std::vector<std::pair<unsigned int, int>> iv(n);
    for (int i = 0; i < n; ++i) {
        int ijk0 = static_cast<int>(floor(xyz[3 * i] * inv) - static_cast<float>(min_b[0]));
        int ijk1 = static_cast<int>(floor(xyz[3 * i + 1] * inv) - static_cast<float>(min_b[1]));
        int ijk2 = static_cast<int>(floor(xyz[3 * i + 2] * inv) - static_cast<float>(min_b[2]));
        int idx = ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2];
        iv[i] = std::make_pair(static_cast<unsigned int>(idx), i);
    }
    std::stable_sort(iv.begin(), iv.end(),
                     [](const std::pair<unsigned int, int>& a, const std::pair<unsigned int, int>& b) {
                         return a.first < b.first;
                     });
    int nout = 0;
    int first = 0;
    while (first < n) {
        int last = first + 1;
        while (last < n && iv[last].first == iv[first].first) ++last;
        float sx = 0, sy = 0, sz = 0;  // AccumulatorXYZ: Vector3f xyz += p ; xyz / n
        for (int k = first; k < last; ++k) {
            int id = iv[k].second;
            sx += xyz[3 * id];
            sy += xyz[3 * id + 1];
            sz += xyz[3 * id + 2];
        }
        float cnt = static_cast<float>(last - first);
        out_xyz[3 * nout] = sx / cnt;
        out_xyz[3 * nout + 1] = sy / cnt;
        out_xyz[3 * nout + 2] = sz / cnt;
        ++nout;
        first = last;
    }
    return nout;
}

// ------------------------------------------------------------------------------------------------
// Section 8(f) rank 2: Estimator::MapIncrementLocal (Estimator.cpp:1585-1643) together with the clear() of the
// local clouds that precedes every call (:1083-1085, :1125-1127).  The local map is the voxel-filtered union of
// the last localMapWindowSize (50, Estimator.h:326) key scans' features in the world frame.
struct mmlo_local_map {
    int window;
    long id;                                   // localMapID
    float leaf[2];                             // downSizeFilterCorner / downSizeFilterSurf leaf sizes (:78-79)
    std::vector<std::vector<float>> ring[2];   // localCornerMap[i] / localSurfMap[i]
    std::vector<float> map[2];                 // laserCloudCornerFromLocal / laserCloudSurfFromLocal
};
extern "C" mmlo_local_map* mmlo_local_map_create(int window, float leaf_corner, float leaf_surf) {
    mmlo_local_map* h = new mmlo_local_map();
    h->window = window;
    h->id = 0;
    h->leaf[0] = leaf_corner;
    h->leaf[1] = leaf_surf;
    for (int k = 0; k < 2; ++k) h->ring[k].resize(window);
    return h;
}
extern "C" void mmlo_local_map_free(mmlo_local_map* h) { delete h; }
extern "C" void mmlo_local_map_increment(mmlo_local_map* h, const float* corner, int nc, const float* surf, int ns,
                                         const double* T) {
    const size_t Id = (size_t)(h->id % h->window);  // :1597
    const float* src[2] = {corner, surf};
    const int cnt[2] = {nc, ns};
    for (int k = 0; k < 2; ++k) {
        std::vector<float>& slot = h->ring[k][Id];
        slot.clear();  // :1600-1601
        for (int i = 0; i < cnt[k]; ++i) {  // :1604-1612, pointAssociateToMap (Map_Manager.cpp:75-89)
            const double x = src[k][3 * i], y = src[k][3 * i + 1], z = src[k][3 * i + 2];
            slot.push_back((float)(((T[0] * x + T[1] * y) + T[2] * z) + T[3]));
            slot.push_back((float)(((T[4] * x + T[5] * y) + T[6] * z) + T[7]));
            slot.push_back((float)(((T[8] * x + T[9] * y) + T[10] * z) + T[11]));
        }
        std::vector<float> cat;  // :1620-1623 onto the cleared cloud
        for (int i = 0; i < h->window; ++i) cat.insert(cat.end(), h->ring[k][i].begin(), h->ring[k][i].end());
        std::vector<float> out(cat.size() ? cat.size() : 3);
        const int m = mmlo_voxel_downsample(cat.data(), (int)(cat.size() / 3), h->leaf[k], out.data());  // :1630-1637
        out.resize((size_t)m * 3);
        h->map[k].swap(out);
    }
    h->id++;  // :1642
}
extern "C" int mmlo_local_map_size(const mmlo_local_map* h, int kind) { return (int)(h->map[kind].size() / 3); }
extern "C" void mmlo_local_map_get(const mmlo_local_map* h, int kind, float* out_xyz) {
    if (!h->map[kind].empty()) memcpy(out_xyz, h->map[kind].data(), h->map[kind].size() * sizeof(float));
}

// ------------------------------------------------------------------------------------------------
// Section 8(f) rank 2, global half: MAP_MANAGER::MapIncrement + MapMove (Map_Manager.cpp:125-581) for the corner and
// surf cube stores (21 x 21 x 11 cubes of 50 m, index i + 21 j + 441 k).  What Estimate() matches against is the state
// copied at the START of the last MapIncrement (laserCloud*_for_match, laserCloudCen*_last, :136-149): one update behind.
struct mmlo_cube_store {
    int cen[3] = {10, 5, 10};        // laserCloudCen{Width,Height,Depth}
    int cen_last[3] = {10, 5, 10};
    float leaf[2];
    std::vector<std::vector<float>> cube[2];        // laserCloud{Corner,Surf}Array
    std::vector<std::vector<float>> for_match[2];   // laserCloud{Corner,Surf}_for_match
};
extern "C" mmlo_cube_store* mmlo_cube_store_create(float leaf_corner, float leaf_surf) {
    mmlo_cube_store* h = new mmlo_cube_store();
    h->leaf[0] = leaf_corner;
    h->leaf[1] = leaf_surf;
    for (int k = 0; k < 2; ++k) {
        h->cube[k].resize(4851);
        h->for_match[k].resize(4851);
    }
    return h;
}
extern "C" void mmlo_cube_store_free(mmlo_cube_store* h) { delete h; }
static inline int cs_index(int i, int j, int k) { return i + 21 * j + 441 * k; }  // ToIndex (:65-67)
// MapMove (:288-581): keep the sensor's cube at least 8 cubes from every face; each step rotates one layer out
static void cs_map_move(mmlo_cube_store* h, const double* T) {
    const double t[3] = {T[3], T[7], T[11]};
    int cI = int((t[0] + 25.0) / 50.0) + h->cen[2];
    int cJ = int((t[1] + 25.0) / 50.0) + h->cen[0];
    int cK = int((t[2] + 25.0) / 50.0) + h->cen[1];
    if (t[0] + 25.0 < 0) cI--;
    if (t[1] + 25.0 < 0) cJ--;
    if (t[2] + 25.0 < 0) cK--;
    const int D = 21, Wd = 21, Hh = 11;
    auto shift = [&](int axis, int dir) {  // dir +1: contents move towards higher index, the lowest layer is emptied
        const int n[3] = {D, Wd, Hh};
        for (int kind = 0; kind < 2; ++kind) {
            auto& c = h->cube[kind];
            for (int a = 0; a < n[(axis + 1) % 3]; ++a)
                for (int b = 0; b < n[(axis + 2) % 3]; ++b) {
                    auto idx = [&](int v) {
                        int ijk[3];
                        ijk[axis] = v;
                        ijk[(axis + 1) % 3] = a;
                        ijk[(axis + 2) % 3] = b;
                        return cs_index(ijk[0], ijk[1], ijk[2]);
                    };
                    if (dir > 0) {
                        for (int v = n[axis] - 1; v >= 1; --v) c[idx(v)].swap(c[idx(v - 1)]);
                        c[idx(0)].clear();
                    } else {
                        for (int v = 0; v < n[axis] - 1; ++v) c[idx(v)].swap(c[idx(v + 1)]);
                        c[idx(n[axis] - 1)].clear();
                    }
                }
        }
    };
    while (cI < 8) {
        shift(0, +1);
        cI++;
        h->cen[2]++;
    }
    while (cI >= D - 8) {
        shift(0, -1);
        cI--;
        h->cen[2]--;
    }
    while (cJ < 8) {
        shift(1, +1);
        cJ++;
        h->cen[0]++;
    }
    while (cJ >= Wd - 8) {
        shift(1, -1);
        cJ--;
        h->cen[0]--;
    }
    while (cK < 8) {
        shift(2, +1);
        cK++;
        h->cen[1]++;
    }
    while (cK >= Hh - 8) {
        shift(2, -1);
        cK--;
        h->cen[1]--;
    }
}
// MapIncrement (:125-281): world-frame feature points (already through featureAssociateToMap), pose of the scan
extern "C" void mmlo_cube_store_increment(mmlo_cube_store* h, const float* corner, int nc, const float* surf, int ns,
                                          const double* T) {
    for (int k = 0; k < 2; ++k) h->for_match[k] = h->cube[k];  // :136-143
    for (int c = 0; c < 3; ++c) h->cen_last[c] = h->cen[c];     // :145-147
    cs_map_move(h, T);
    const float* src[2] = {corner, surf};
    const int cnt[2] = {nc, ns};
    for (int k = 0; k < 2; ++k) {
        std::vector<char> changed(4851, 0);
        for (int i = 0; i < cnt[k]; ++i) {
            const float* p = src[k] + 3 * i;
            int cI = int((p[0] + 25.0) / 50.0) + h->cen[2];
            int cJ = int((p[1] + 25.0) / 50.0) + h->cen[0];
            int cK = int((p[2] + 25.0) / 50.0) + h->cen[1];
            if (p[0] + 25.0 < 0) cI--;
            if (p[1] + 25.0 < 0) cJ--;
            if (p[2] + 25.0 < 0) cK--;
            if (cI >= 0 && cI < 21 && cJ >= 0 && cJ < 21 && cK >= 0 && cK < 11) {
                const int id = cs_index(cI, cJ, cK);
                h->cube[k][id].insert(h->cube[k][id].end(), p, p + 3);
                changed[id] = 1;
            }
        }
        for (int id = 0; id < 4851; ++id) {
            if (!changed[id]) continue;
            std::vector<float>& c = h->cube[k][id];
            if (c.size() / 3 > 300) {  // :225-233
                std::vector<float> out(c.size());
                const int m = mmlo_voxel_downsample(c.data(), (int)(c.size() / 3), h->leaf[k], out.data());
                out.resize((size_t)m * 3);
                c.swap(out);
            }
        }
    }
}
// which: 0 the live store, 1 the copy Estimate() matches against.  Returns the number of points; xyz / cube may be null.
extern "C" int mmlo_cube_store_get(const mmlo_cube_store* h, int which, int kind, float* xyz, int* cube, int* cen) {
    const auto& c = which ? h->for_match[kind] : h->cube[kind];
    int n = 0;
    for (int id = 0; id < 4851; ++id) {
        const int m = (int)(c[id].size() / 3);
        if (xyz && m) memcpy(xyz + 3 * (size_t)n, c[id].data(), sizeof(float) * 3 * (size_t)m);
        if (cube)
            for (int i = 0; i < m; ++i) cube[n + i] = id;
        n += m;
    }
    if (cen)
        for (int k = 0; k < 3; ++k) cen[k] = which ? h->cen_last[k] : h->cen[k];
    return n;
}

// ------------------------------------------------------------------------------------------------
// a13  exact 5-NN with FLANN L2_Simple<float> distance semantics: d2 = ((dx*dx + dy*dy) + dz*dz) in
// float; result ascending, ties broken by lower index (FLANN's tie order is unspecified: convention).
struct Top5 {
    float d[5];
    int id[5];
    int cnt;
    Top5() : cnt(0) {
        for (int k = 0; k < 5; ++k) {
            d[k] = INFINITY;
            id[k] = -1;
        }
    }
    inline float worst() const { return d[4]; }
    inline void insert(float dd, int ii) {
        // position: after all entries with (d < dd) or (d == dd and id < ii)
        if (cnt == 5 && !(dd < d[4] || (dd == d[4] && ii < id[4]))) return;
        int k = cnt < 5 ? cnt : 4;
        while (k > 0 && (dd < d[k - 1] || (dd == d[k - 1] && ii < id[k - 1]))) {
            d[k] = d[k - 1];
            id[k] = id[k - 1];
            --k;
        }
        d[k] = dd;
        id[k] = ii;
        if (cnt < 5) ++cnt;
    }
};
static inline float l2f(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float r = 0;
    r += dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
}

struct mmlo_kdtree {
    const float* pts;
    int m;
    std::vector<int> perm;
    struct Node {
        int lo, hi;    // range in perm
        int axis;      // -1 leaf
        float split;
        int left, right;
    };
    std::vector<Node> nodes;
    int build(int lo, int hi) {
        Node nd;
        nd.lo = lo;
        nd.hi = hi;
        nd.axis = -1;
        nd.split = 0;
        nd.left = nd.right = -1;
        int id = (int)nodes.size();
        nodes.push_back(nd);
        if (hi - lo > 8) {
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int i = lo; i < hi; ++i)
                for (int c = 0; c < 3; ++c) {
                    float v = pts[3 * perm[i] + c];
                    mn[c] = std::min(mn[c], v);
                    mx[c] = std::max(mx[c], v);
                }
            int ax = 0;
            for (int c = 1; c < 3; ++c)
                if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
            if (mx[ax] > mn[ax]) {
                int mid = (lo + hi) / 2;
                std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                                 [&](int a, int b) { return pts[3 * a + ax] < pts[3 * b + ax]; });
                float split = pts[3 * perm[mid] + ax];
                int l = build(lo, mid);
                int r = build(mid, hi);
                nodes[id].axis = ax;
                nodes[id].split = split;
                nodes[id].left = l;
                nodes[id].right = r;
            }
        }
        return id;
    }
    void search(int nid, const float* q, Top5& best) const {
        const Node& nd = nodes[nid];
        if (nd.axis < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) best.insert(l2f(q, pts + 3 * perm[i]), perm[i]);
            return;
        }
        float diff = q[nd.axis] - nd.split;
        int nearc = diff < 0 ? nd.left : nd.right;
        int farc = diff < 0 ? nd.right : nd.left;
        search(nearc, q, best);
        // left holds coords <= split, right holds coords >= split: every far-side point has
        // |q-p| >= |q-split| on this axis, and float rounding is monotone, so d2(p) >= diff*diff.
        if (best.cnt < 5 || diff * diff <= best.worst()) search(farc, q, best);
    }
};

extern "C" mmlo_kdtree* mmlo_kdtree_build(const float* xyz, int m) {
    mmlo_kdtree* t = new mmlo_kdtree();
    t->pts = xyz;
    t->m = m;
    t->perm.resize(m);
    for (int i = 0; i < m; ++i) t->perm[i] = i;
    t->nodes.reserve(m / 4 + 4);
    if (m > 0) t->build(0, m);
    return t;
}
extern "C" void mmlo_kdtree_free(mmlo_kdtree* t) { delete t; }
extern "C" void mmlo_kdtree_knn5(const mmlo_kdtree* t, const float* q, int* idx, float* d2) {
    Top5 best;
    if (t->m > 0) t->search(0, q, best);
    for (int k = 0; k < 5; ++k) {
        idx[k] = best.id[k];
        d2[k] = best.d[k];
    }
}
extern "C" void mmlo_bruteforce_knn5(const float* xyz, int m, const float* q, int* idx, float* d2) {
    Top5 best;
    for (int i = 0; i < m; ++i) best.insert(l2f(q, xyz + 3 * i), i);
    for (int k = 0; k < 5; ++k) {
        idx[k] = best.id[k];
        d2[k] = best.d[k];
    }
}

// ------------------------------------------------------------------------------------------------
// a11  Map_Manager.cpp:75-89 pointAssociateToMap: double transform, stored back to float.
static inline void point_associate_to_map(const float* pi, float* po, const double* T) {
    double x = pi[0], y = pi[1], z = pi[2];
    po[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    po[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    po[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}
static inline Vec3 transform_d(const double* T, const Vec3& p) {
    return mk(((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3], ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7],
              ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11]);
}

// Estimator.h:71-83 FeatureLine::ComputeError
static double line_error(const Vec3& po, const Vec3& a, const Vec3& b, const double* T) {
    Vec3 P = transform_d(T, po);
    double l12 = std::sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z));
    double c0 = (P.x - a.x) * (P.y - b.y) - (P.x - b.x) * (P.y - a.y);
    double c1 = (P.x - a.x) * (P.z - b.z) - (P.x - b.x) * (P.z - a.z);
    double c2 = (P.y - a.y) * (P.z - b.z) - (P.y - b.y) * (P.z - a.z);
    double a012 = std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    return a012 / l12;
}

// a14  line model of processPointToLine (Estimator.cpp:204-277 for the cube cloud, :288-358 for the local cloud:
// the two copies are arithmetically identical).  nb: coordinates of the 5 neighbours in search order.
static bool fit_line(const float* ori, const float nb[5][3], const double* T, mmlo_line_factor* f) {
    float cx = 0, cy = 0, cz = 0;
    for (int j = 0; j < 5; j++) {
        cx += nb[j][0];
        cy += nb[j][1];
        cz += nb[j][2];
    }
    cx /= 5;
    cy /= 5;
    cz /= 5;
    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
    for (int j = 0; j < 5; j++) {
        float ax = nb[j][0] - cx;
        float ay = nb[j][1] - cy;
        float az = nb[j][2] - cz;
        a11 += ax * ax;
        a12 += ax * ay;
        a13 += ax * az;
        a22 += ay * ay;
        a23 += ay * az;
        a33 += az * az;
    }
    a11 /= 5;
    a12 /= 5;
    a13 /= 5;
    a22 /= 5;
    a23 /= 5;
    a33 /= 5;
    double A[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33};
    double ev[3], V[9];
    eig3_sym(A, ev, V);
    double ud[3] = {V[2], V[5], V[8]};  // eigenvectors().col(2)
    if (!(ev[2] > 3 * ev[1])) return false;
    float x1 = cx + 0.1 * ud[0];
    float y1 = cy + 0.1 * ud[1];
    float z1 = cz + 0.1 * ud[2];
    float x2 = cx - 0.1 * ud[0];
    float y2 = cy - 0.1 * ud[1];
    float z2 = cz - 0.1 * ud[2];
    f->point_ori[0] = ori[0];
    f->point_ori[1] = ori[1];
    f->point_ori[2] = ori[2];
    f->p1[0] = x1;
    f->p1[1] = y1;
    f->p1[2] = z1;
    f->p2[0] = x2;
    f->p2[1] = y2;
    f->p2[2] = z2;
    f->error = line_error(mk(ori[0], ori[1], ori[2]), mk(x1, y1, z1), mk(x2, y2, z2), T);
    return true;
}

// a15  plane model of processPointToPlanVec (Estimator.cpp:634-693 cube cloud, :708-764 local cloud).
static bool fit_plane(const float* ori, const float* sel, const float nb[5][3], const double* T, mmlo_plane_factor* f) {
    double A[15];
    for (int j = 0; j < 5; j++) {
        A[3 * j] = nb[j][0];
        A[3 * j + 1] = nb[j][1];
        A[3 * j + 2] = nb[j][2];
    }
    double X[3];
    plane_fit5(A, X);
    float pa = X[0];
    float pb = X[1];
    float pc = X[2];
    float pd = 1;
    float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
    pa /= ps;
    pb /= ps;
    pc /= ps;
    pd /= ps;
    for (int j = 0; j < 5; j++) {
        if (std::fabs(pa * nb[j][0] + pb * nb[j][1] + pc * nb[j][2] + pd) > 0.2) return false;
    }
    double dist = pa * sel[0] + pb * sel[1] + pc * sel[2] + pd;  // float expression (:740-742)
    Vec3 omega = mk(pa, pb, pc);
    Vec3 proj = mk(sel[0], sel[1], sel[2]) - dist * omega;
    f->point_ori[0] = ori[0];
    f->point_ori[1] = ori[1];
    f->point_ori[2] = ori[2];
    f->point_proj[0] = proj.x;
    f->point_proj[1] = proj.y;
    f->point_proj[2] = proj.z;
    f->omega[0] = pa;
    f->omega[1] = pb;
    f->omega[2] = pc;
    Vec3 P = transform_d(T, mk(ori[0], ori[1], ori[2]));
    f->error = norm(P - proj);  // Estimator.h:118-121
    return true;
}

// Map_Manager.cpp:583-629 FindUsed{Corner,Surf}Map: cube index of a (float) map-frame point, 5000 outside the grid.
// cen = {laserCloudCenWidth_last, laserCloudCenHeight_last, laserCloudCenDepth_last}.
static int find_used_map(const float* p, const int* cen) {
    int cubeI = int((p[0] + 25.0) / 50.0) + cen[2];
    int cubeJ = int((p[1] + 25.0) / 50.0) + cen[0];
    int cubeK = int((p[2] + 25.0) / 50.0) + cen[1];
    if (p[0] + 25.0 < 0) cubeI--;
    if (p[1] + 25.0 < 0) cubeJ--;
    if (p[2] + 25.0 < 0) cubeK--;
    if (cubeI >= 0 && cubeI < 21 && cubeJ >= 0 && cubeJ < 21 && cubeK >= 0 && cubeK < 11)
        return cubeI + 21 * cubeJ + 21 * 21 * cubeK;  // ToIndex, Map_Manager.cpp:65-67
    return 5000;
}
extern "C" int mmlo_find_used_map(const float* p, const int* cen) { return find_used_map(p, cen); }

// The global map as Estimate() sees it (Estimator.cpp:1170-1184): per-cube clouds with their own kd-trees.
struct mmlo_cube_map {
    std::vector<std::vector<float>> xyz;   // 4851 clouds
    std::vector<mmlo_kdtree*> tree;
    int cen[3];
    ~mmlo_cube_map() {
        for (auto* t : tree)
            if (t) mmlo_kdtree_free(t);
    }
};
extern "C" mmlo_cube_map* mmlo_cube_map_build(const float* xyz, const int* cube, int m, const int* cen) {
    mmlo_cube_map* g = new mmlo_cube_map();
    g->xyz.resize(4851);
    g->tree.assign(4851, nullptr);
    for (int c = 0; c < 3; ++c) g->cen[c] = cen[c];
    for (int i = 0; i < m; ++i) {
        if (cube[i] < 0 || cube[i] >= 4851) continue;
        auto& v = g->xyz[cube[i]];
        v.push_back(xyz[3 * i]);
        v.push_back(xyz[3 * i + 1]);
        v.push_back(xyz[3 * i + 2]);
    }
    for (int c = 0; c < 4851; ++c)
        if (!g->xyz[c].empty()) g->tree[c] = mmlo_kdtree_build(g->xyz[c].data(), (int)g->xyz[c].size() / 3);
    return g;
}
extern "C" void mmlo_cube_map_free(mmlo_cube_map* g) { delete g; }

// processPointToLine (Estimator.cpp:148-365): cube cloud first (> 100 points), local cloud as fall-back.
// gmap may be null (no global map: every cube is empty).  from_global (optional): 1 where the cube cloud was used.
extern "C" int mmlo_associate_lines2(const float* feat, int n_feat, const mmlo_cube_map* gmap, const float* map, int m,
                                     const mmlo_kdtree* tree, const double* T, double thres_dist,
                                     mmlo_line_factor* out, int* out_src, int* from_global) {
    int nout = 0;
    static const int cen_default[3] = {10, 5, 10};
    for (int i = 0; i < n_feat; ++i) {
        const float* ori = feat + 3 * i;
        float sel[3];
        point_associate_to_map(ori, sel, T);
        int id = find_used_map(sel, gmap ? gmap->cen : cen_default);
        if (id == 5000) continue;                                                        // :194
        if (std::isnan(sel[0]) || std::isnan(sel[1]) || std::isnan(sel[2])) continue;  // :196
        int ind[5];
        float sq[5];
        float nb[5][3];
        if (gmap && gmap->xyz[id].size() / 3 > 100) {  // :198
            mmlo_kdtree_knn5(gmap->tree[id], sel, ind, sq);
            if (sq[4] < thres_dist) {
                for (int j = 0; j < 5; ++j)
                    for (int c = 0; c < 3; ++c) nb[j][c] = gmap->xyz[id][3 * ind[j] + c];
                if (fit_line(ori, nb, T, &out[nout])) {
                    if (out_src) out_src[nout] = i;
                    if (from_global) from_global[nout] = 1;
                    ++nout;
                    continue;  // :276
                }
            }
        }
        if (m > 20) {  // :283
            mmlo_kdtree_knn5(tree, sel, ind, sq);
            if (sq[4] < thres_dist) {
                for (int j = 0; j < 5; ++j)
                    for (int c = 0; c < 3; ++c) nb[j][c] = map[3 * ind[j] + c];
                if (fit_line(ori, nb, T, &out[nout])) {
                    if (out_src) out_src[nout] = i;
                    if (from_global) from_global[nout] = 0;
                    ++nout;
                }
            }
        }
    }
    return nout;
}

// processPointToPlanVec (Estimator.cpp:573-777): cube cloud first (> 50 points), local cloud as fall-back.
extern "C" int mmlo_associate_planes2(const float* feat, int n_feat, const mmlo_cube_map* gmap, const float* map, int m,
                                      const mmlo_kdtree* tree, const double* T, double thres_dist,
                                      mmlo_plane_factor* out, int* out_src, int* from_global) {
    int nout = 0;
    static const int cen_default[3] = {10, 5, 10};
    for (int i = 0; i < n_feat; ++i) {
        const float* ori = feat + 3 * i;
        float sel[3];
        point_associate_to_map(ori, sel, T);
        int id = find_used_map(sel, gmap ? gmap->cen : cen_default);
        if (id == 5000) continue;                                                        // :623
        if (std::isnan(sel[0]) || std::isnan(sel[1]) || std::isnan(sel[2])) continue;  // :625
        int ind[5];
        float sq[5];
        float nb[5][3];
        if (gmap && gmap->xyz[id].size() / 3 > 50) {  // :627
            mmlo_kdtree_knn5(gmap->tree[id], sel, ind, sq);
            if (sq[4] < thres_dist) {
                for (int j = 0; j < 5; ++j)
                    for (int c = 0; c < 3; ++c) nb[j][c] = gmap->xyz[id][3 * ind[j] + c];
                if (fit_plane(ori, sel, nb, T, &out[nout])) {
                    if (out_src) out_src[nout] = i;
                    if (from_global) from_global[nout] = 1;
                    ++nout;
                    continue;  // :694
                }
            }
        }
        if (m > 20) {  // :702
            mmlo_kdtree_knn5(tree, sel, ind, sq);
            if (sq[4] < thres_dist) {
                for (int j = 0; j < 5; ++j)
                    for (int c = 0; c < 3; ++c) nb[j][c] = map[3 * ind[j] + c];
                if (fit_plane(ori, sel, nb, T, &out[nout])) {
                    if (out_src) out_src[nout] = i;
                    if (from_global) from_global[nout] = 0;
                    ++nout;
                }
            }
        }
    }
    return nout;
}

// local-map-only entry points (the global cubes empty): kept as the API used throughout the tests
extern "C" int mmlo_associate_lines(const float* feat, int n_feat, const float* map, int m, const mmlo_kdtree* tree,
                                    const double* T, double thres_dist, mmlo_line_factor* out, int* out_src) {
    return mmlo_associate_lines2(feat, n_feat, nullptr, map, m, tree, T, thres_dist, out, out_src, nullptr);
}
extern "C" int mmlo_associate_planes(const float* feat, int n_feat, const float* map, int m, const mmlo_kdtree* tree,
                                     const double* T, double thres_dist, mmlo_plane_factor* out, int* out_src) {
    return mmlo_associate_planes2(feat, n_feat, nullptr, map, m, tree, T, thres_dist, out, out_src, nullptr);
}

// Estimator.cpp:536-565 checkLocalizability: JacobiSVD singular values of the M x 3 normal matrix;
// restated as sqrt(lambda_min(N^T N)).
extern "C" double mmlo_check_localizability(const mmlo_plane_factor* f, int n) {
    if (!(n > 10)) return -1;
    double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) G[3 * r + c] += f[i].omega[r] * f[i].omega[c];
    double ev[3], V[9];
    eig3_sym(G, ev, V);
    return std::sqrt(ev[0] > 0 ? ev[0] : 0.0);
}

// ------------------------------------------------------------------------------------------------
// a17..a19  ceresfunc.h:397-458 (line), :517-570 (plane-vec).  Values follow the functor (quaternion
// composition as written there); Jacobians are analytic (the reference gets them from ceres::Jet
// autodiff through Sophus::SO3<Jet>::exp): dP/dt = I, dP/dphi = -[R p_b]x J_l(phi).
struct PoseEval {
    Quat q_wl;
    Vec3 t_wl;
    Quat q_wb;
    double Jl[9];  // left Jacobian of SO(3) at phi
    Vec3 Pbl;
    Quat qbl;
};
static void left_jacobian(const Vec3& phi, double* J) {
    double th2 = dot(phi, phi);
    double a, b;  // J = I + a [phi]x + b [phi]x^2
    if (th2 < 1e-10 * 1e-10) {  // matches the Taylor branch of so3.hpp:597-604 to first order
        a = 0.5;
        b = 1.0 / 6.0;
    } else {
        double th = std::sqrt(th2);
        a = (1.0 - std::cos(th)) / th2;
        b = (th - std::sin(th)) / (th2 * th);
    }
    double K[9] = {0, -phi.z, phi.y, phi.z, 0, -phi.x, -phi.y, phi.x, 0};
    double K2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    for (int i = 0; i < 9; ++i) J[i] = a * K[i] + b * K2[i];
    J[0] += 1;
    J[4] += 1;
    J[8] += 1;
}
static PoseEval make_pose(const double* x, const double* T_bl) {
    PoseEval pe;
    double m3d[9] = {T_bl[0], T_bl[1], T_bl[2], T_bl[4], T_bl[5], T_bl[6], T_bl[8], T_bl[9], T_bl[10]};
    pe.qbl = qnormalized(quat_from_matrix(m3d));  // ceresfunc.h:404-405
    pe.Pbl = mk(T_bl[3], T_bl[7], T_bl[11]);
    Vec3 phi = mk(x[3], x[4], x[5]);
    pe.q_wb = so3_exp(phi);  // :421
    Vec3 t_wb = mk(x[0], x[1], x[2]);
    pe.q_wl = qmul(pe.q_wb, pe.qbl);             // :423
    pe.t_wl = qrot(pe.q_wb, pe.Pbl) + t_wb;      // :424
    left_jacobian(phi, pe.Jl);
    return pe;
}
// dP/dx (3x6) for P = q_wl * cp + t_wl
static void dP_dx(const PoseEval& pe, const Vec3& cp, const Vec3& P, double* D) {
    // R p_b = P - t_wb ; t_wb = t_wl - q_wb*Pbl  =>  R p_b = P - x[0:3].  Recompute directly:
    Vec3 pb = qrot(pe.qbl, cp) + pe.Pbl;
    Vec3 Rpb = qrot(pe.q_wb, pb);
    (void)P;
    double S[9] = {0, Rpb.z, -Rpb.y, -Rpb.z, 0, Rpb.x, Rpb.y, -Rpb.x, 0};  // -[Rpb]x
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            D[6 * r + c] = (r == c) ? 1.0 : 0.0;
            D[6 * r + 3 + c] = S[3 * r] * pe.Jl[c] + S[3 * r + 1] * pe.Jl[3 + c] + S[3 * r + 2] * pe.Jl[6 + c];
        }
    }
}

static void line_eval(const mmlo_line_factor* f, const PoseEval& pe, double* r, double* J) {
    Vec3 cp = mk(f->point_ori[0], f->point_ori[1], f->point_ori[2]);
    Vec3 a = mk(f->p1[0], f->p1[1], f->p1[2]);
    Vec3 b = mk(f->p2[0], f->p2[1], f->p2[2]);
    double l12 = std::sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z));
    Vec3 P = qrot(pe.q_wl, cp) + pe.t_wl;
    double c0 = (P.x - a.x) * (P.y - b.y) - (P.x - b.x) * (P.y - a.y);
    double c1 = (P.x - a.x) * (P.z - b.z) - (P.x - b.x) * (P.z - a.z);
    double c2 = (P.y - a.y) * (P.z - b.z) - (P.y - b.y) * (P.z - a.z);
    double a012 = std::sqrt(c0 * c0 + c1 * c1 + c2 * c2);
    double ld2 = a012 / l12;
    double s = P.x * P.x + P.y * P.y + P.z * P.z;
    double rs = std::sqrt(std::sqrt(s));
    double weight = 1.0 - 0.9 * std::fabs(ld2) / rs;
    double k = 1.0 / kLidarM;
    *r = k * weight * ld2;
    if (J) {
        // u = (P-a)x(P-b) = (c2, -c1, c0);  grad_P ld = ((a-b) x u_hat) / l12
        Vec3 u = mk(c2, -c1, c0);
        // a point exactly on the line: the residual is zero and the norm has no derivative there (ceres::sqrt(Jet) at 0 is NaN);
        // convention shared with the device path (lidar_eval.h sqrt_pair): a zero Jacobian row
        Vec3 uh = a012 > 0.0 ? (1.0 / a012) * u : mk(0, 0, 0);
        Vec3 gld = (1.0 / l12) * cross(a - b, uh);
        double sgn = ld2 >= 0 ? 1.0 : -1.0;
        // weight = 1 - 0.9 |ld| s^(-1/4);  d s^(-1/4)/dP = -1/2 s^(-5/4) P
        double sm14 = 1.0 / rs;
        double sm54 = sm14 / s;
        Vec3 gw = (-0.9) * ((sgn * sm14) * gld + (std::fabs(ld2) * (-0.5) * sm54) * P);
        Vec3 gr = k * (weight * gld + ld2 * gw);
        double D[18];
        dP_dx(pe, cp, P, D);
        for (int c = 0; c < 6; ++c) J[c] = gr.x * D[c] + gr.y * D[6 + c] + gr.z * D[12 + c];
    }
}

// e = weight * (P - proj) and de/dP (3x3); the 3 residual rows are S*e with S^T S = a^2 w w^T + b^2 (I - w w^T)
static void plane_core(const mmlo_plane_factor* f, const PoseEval& pe, Vec3* e, double* dedP, Vec3* cp_out, Vec3* P_out) {
    Vec3 cp = mk(f->point_ori[0], f->point_ori[1], f->point_ori[2]);
    Vec3 proj = mk(f->point_proj[0], f->point_proj[1], f->point_proj[2]);
    Vec3 P = qrot(pe.q_wl, cp) + pe.t_wl;
    Vec3 d = P - proj;
    double nd = norm(d);
    double s = P.x * P.x + P.y * P.y + P.z * P.z;
    double rs = std::sqrt(std::sqrt(s));
    double weight = 1.0 - 0.9 * nd / rs;
    *e = weight * d;
    if (dedP) {
        double sm14 = 1.0 / rs;
        double sm54 = sm14 / s;
        // (nd = 0, a point exactly on its projection: zero rows, as for the line factor)
        Vec3 gw = (-0.9) * ((nd > 0.0 ? sm14 / nd : 0.0) * d + (nd * (-0.5) * sm54) * P);
        const double dv[3] = {d.x, d.y, d.z};
        const double gv[3] = {gw.x, gw.y, gw.z};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) dedP[3 * r + c] = (r == c ? weight : 0.0) + dv[r] * gv[c];
    }
    *cp_out = cp;
    *P_out = P;
}
// Deterministic orthonormal basis {w, v2, v3}: stand-in for JacobiSVD(e1 w^T) at Estimator.cpp:675-682,
// whose U,V tangent columns are not unique.  Rows of S: a*w^T, b*v2^T, b*v3^T.
static void plane_basis(const Vec3& w, Vec3* v2, Vec3* v3) {
    Vec3 h = (std::fabs(w.x) <= std::fabs(w.y) && std::fabs(w.x) <= std::fabs(w.z)) ? mk(1, 0, 0)
             : (std::fabs(w.y) <= std::fabs(w.z))                                    ? mk(0, 1, 0)
                                                                                     : mk(0, 0, 1);
    Vec3 t = cross(w, h);
    double n = norm(t);
    *v2 = (1.0 / n) * t;
    *v3 = cross(w, *v2);
}

extern "C" void mmlo_line_residual(const mmlo_line_factor* f, const double* x, const double* T_bl, double* r,
                                   double* J) {
    PoseEval pe = make_pose(x, T_bl);
    line_eval(f, pe, r, J);
}
extern "C" void mmlo_plane_residual(const mmlo_plane_factor* f, const double* x, const double* T_bl,
                                    double plan_weight_tan, double* r, double* J) {
    PoseEval pe = make_pose(x, T_bl);
    Vec3 e, cp, P;
    double dedP[9];
    plane_core(f, pe, &e, J ? dedP : nullptr, &cp, &P);
    Vec3 w = mk(f->omega[0], f->omega[1], f->omega[2]);
    Vec3 v2, v3;
    plane_basis(w, &v2, &v3);
    double a = 1.0 / kLidarM, b = plan_weight_tan / kLidarM;
    const Vec3 rows[3] = {a * w, b * v2, b * v3};
    for (int i = 0; i < 3; ++i) r[i] = dot(rows[i], e);
    if (J) {
        double D[18];
        dP_dx(pe, cp, P, D);
        for (int i = 0; i < 3; ++i) {
            double g[3];
            for (int c = 0; c < 3; ++c) g[c] = rows[i].x * dedP[c] + rows[i].y * dedP[3 + c] + rows[i].z * dedP[6 + c];
            for (int c = 0; c < 6; ++c) J[6 * i + c] = g[0] * D[c] + g[1] * D[6 + c] + g[2] * D[12 + c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a20  normal equations with Ceres' loss correction.
// ceres/loss_function.cc HuberLoss::Evaluate and ceres/corrector.cc: for rho'' <= 0 (Huber) the corrector
// reduces to residual_scaling = sqrt(rho'), alpha = 0, i.e. r <- sqrt(rho') r, J <- sqrt(rho') J.
static inline void huber(double s, double a, double* rho0, double* rho1) {
    if (a <= 0) {
        *rho0 = s;
        *rho1 = 1.0;
        return;
    }
    double b = a * a;
    if (s > b) {
        double r = std::sqrt(s);
        *rho0 = 2.0 * a * r - b;
        *rho1 = std::max(std::numeric_limits<double>::min(), a / r);
    } else {
        *rho0 = s;
        *rho1 = 1.0;
    }
}

static void linearize_pe_serial(const mmlo_line_factor* lf, int n_line, const mmlo_plane_factor* pf, int n_plane,
                                const PoseEval& pe, double plan_weight_tan, double huber_delta, double* H, double* g,
                                double* cost, bool want_jac);
// num_threads > 1 (Estimator.cpp:1430): the residual blocks split into contiguous chunks, partial sums added in chunk order
static void linearize_pe(const mmlo_line_factor* lf, int n_line, const mmlo_plane_factor* pf, int n_plane,
                         const PoseEval& pe, double plan_weight_tan, double huber_delta, double* H, double* g,
                         double* cost, bool want_jac) {
    const int T = mmlo::threading().solve_threads;
    if (T <= 1 || n_line + n_plane < 4 * T) {
        linearize_pe_serial(lf, n_line, pf, n_plane, pe, plan_weight_tan, huber_delta, H, g, cost, want_jac);
        return;
    }
    std::vector<double> part((size_t)T * 43, 0.0);
    mmlo::pool(T).run(T, [&](int t) {
        const int l0 = (int)((long long)n_line * t / T), l1 = (int)((long long)n_line * (t + 1) / T);
        const int p0 = (int)((long long)n_plane * t / T), p1 = (int)((long long)n_plane * (t + 1) / T);
        double* o = part.data() + (size_t)t * 43;
        linearize_pe_serial(lf + l0, l1 - l0, pf + p0, p1 - p0, pe, plan_weight_tan, huber_delta, o, o + 36, o + 42, want_jac);
    });
    if (want_jac) {
        for (int i = 0; i < 36; ++i) H[i] = 0;
        for (int i = 0; i < 6; ++i) g[i] = 0;
    }
    double c = 0;
    for (int t = 0; t < T; ++t) {
        const double* o = part.data() + (size_t)t * 43;
        if (want_jac) {
            for (int i = 0; i < 36; ++i) H[i] += o[i];
            for (int i = 0; i < 6; ++i) g[i] += o[36 + i];
        }
        c += o[42];
    }
    *cost = c;
}
static void linearize_pe_serial(const mmlo_line_factor* lf, int n_line, const mmlo_plane_factor* pf, int n_plane,
                                const PoseEval& pe, double plan_weight_tan, double huber_delta, double* H, double* g,
                                double* cost, bool want_jac) {
    if (want_jac) {
        for (int i = 0; i < 36; ++i) H[i] = 0;
        for (int i = 0; i < 6; ++i) g[i] = 0;
    }
    double c = 0;
    for (int i = 0; i < n_line; ++i) {
        if (!(std::fabs(lf[i].error) > 1e-5)) continue;  // Estimator.cpp:1385
        double r, J[6];
        line_eval(&lf[i], pe, &r, want_jac ? J : nullptr);
        double rho0, rho1;
        huber(r * r, huber_delta, &rho0, &rho1);
        c += 0.5 * rho0;
        if (want_jac) {
            for (int a = 0; a < 6; ++a) {
                g[a] += rho1 * J[a] * r;
                for (int b = 0; b < 6; ++b) H[6 * a + b] += rho1 * J[a] * J[b];
            }
        }
    }
    double ka = 1.0 / kLidarM, kb = plan_weight_tan / kLidarM;
    for (int i = 0; i < n_plane; ++i) {
        if (!(std::fabs(pf[i].error) > 1e-5)) continue;  // Estimator.cpp:1396
        Vec3 e, cp, P;
        double dedP[9];
        plane_core(&pf[i], pe, &e, want_jac ? dedP : nullptr, &cp, &P);
        Vec3 w = mk(pf[i].omega[0], pf[i].omega[1], pf[i].omega[2]);
        // M = S^T S = a^2 w w^T + b^2 (I - w w^T)
        double we = dot(w, e);
        Vec3 Me = (ka * ka - kb * kb) * we * w + (kb * kb) * e;
        double s = dot(e, Me);
        double rho0, rho1;
        huber(s, huber_delta, &rho0, &rho1);
        c += 0.5 * rho0;
        if (want_jac) {
            double D[18];
            dP_dx(pe, cp, P, D);
            double E[18];  // de/dx = dedP * D  (3x6)
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 6; ++cc)
                    E[6 * r + cc] = dedP[3 * r] * D[cc] + dedP[3 * r + 1] * D[6 + cc] + dedP[3 * r + 2] * D[12 + cc];
            const double wv[3] = {w.x, w.y, w.z};
            const double Mev[3] = {Me.x, Me.y, Me.z};
            double wE[6];
            for (int cc = 0; cc < 6; ++cc) wE[cc] = wv[0] * E[cc] + wv[1] * E[6 + cc] + wv[2] * E[12 + cc];
            for (int a = 0; a < 6; ++a) {
                g[a] += rho1 * (E[a] * Mev[0] + E[6 + a] * Mev[1] + E[12 + a] * Mev[2]);
                for (int b = 0; b < 6; ++b) {
                    double ete = E[a] * E[b] + E[6 + a] * E[6 + b] + E[12 + a] * E[12 + b];
                    H[6 * a + b] += rho1 * ((ka * ka - kb * kb) * wE[a] * wE[b] + (kb * kb) * ete);
                }
            }
        }
    }
    *cost = c;
}

extern "C" void mmlo_linearize(const mmlo_line_factor* lf, int n_line, const mmlo_plane_factor* pf, int n_plane,
                               const double* x, const double* T_bl, double plan_weight_tan, double huber_delta,
                               double* H, double* g, double* cost) {
    PoseEval pe = make_pose(x, T_bl);
    linearize_pe(lf, n_line, pf, n_plane, pe, plan_weight_tan, huber_delta, H, g, cost, true);
}

// ------------------------------------------------------------------------------------------------
// a20  Ceres 2.1.0 TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) + DENSE_SCHUR, restated for a
// problem of W independent 6-D pose blocks (lidar factors only touch para_PR[f], Estimator.cpp:1314,1326).
// Options as the reference leaves them (Estimator.cpp:1425-1432 + Ceres defaults): initial radius 1e4,
// max radius 1e16, min radius 1e-32, min_relative_decrease 1e-3, function_tolerance 1e-6,
// gradient_tolerance 1e-10, parameter_tolerance 1e-8, jacobi_scaling on, monotonic steps,
// dogleg: min_diagonal 1e-6, max_diagonal 1e32, mu 1e-8..1, mu_increase_factor 10,
// increase_threshold 0.75, decrease_threshold 0.25, max_num_consecutive_invalid_steps 5.
// H-based evaluation: ||J v||^2 = v^T H v, model change = -(d^T g + 1/2 d^T H d).
namespace {
struct WindowProblem {
    const mmlo_line_factor* lf;
    const int* n_line;
    const mmlo_plane_factor* pf;
    const int* n_plane;
    int W;
    const double* T_bl;
    double plan_weight_tan, huber_delta;
    // evaluates total cost and (optionally) block-diagonal H (W x 36) and g (6W)
    double eval(const double* x, double* H, double* g) const {
        double total = 0;
        int lo = 0, po = 0;
        for (int f = 0; f < W; ++f) {
            PoseEval pe = make_pose(x + 6 * f, T_bl);
            double c;
            linearize_pe(lf + lo, n_line[f], pf + po, n_plane[f], pe, plan_weight_tan, huber_delta,
                         H ? H + 36 * f : nullptr, g ? g + 6 * f : nullptr, &c, H != nullptr);
            total += c;
            lo += n_line[f];
            po += n_plane[f];
        }
        return total;
    }
};
}  // namespace

extern "C" void mmlo_solve_window(const mmlo_line_factor* lf, const int* n_line, const mmlo_plane_factor* pf,
                                  const int* n_plane, int W, const double* T_bl, const mmlo_solve_opts* opts,
                                  double* x, mmlo_solve_summary* summary, double* trace) {
    WindowProblem prob{lf, n_line, pf, n_plane, W, T_bl, opts->plan_weight_tan, opts->huber_delta};
    const int n = 6 * W;
    std::vector<double> H(36 * W), g(n), Hc(36 * W), gc(n), scale(n), diag(n), grad(n), gn(n), step(n), delta(n),
        xc(n);
    double radius = 1e4;
    double mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
    bool reuse = false;
    double alpha = 0;
    double dogleg_step_norm = 0;
    int num_invalid = 0;
    const std::vector<double> x_entry(x, x + n);

    double cost = prob.eval(x, H.data(), g.data());
    summary->initial_cost = cost;
    summary->iterations = 0;
    summary->successful = 0;
    summary->termination = 0;
    // jacobi scaling, computed once from the initial Jacobian: 1 / (1 + sqrt(colnorm^2))
    for (int f = 0; f < W; ++f)
        for (int i = 0; i < 6; ++i) scale[6 * f + i] = 1.0 / (1.0 + std::sqrt(H[36 * f + 7 * i]));
    double x_norm = 0;
    for (int i = 0; i < n; ++i) x_norm += x[i] * x[i];
    x_norm = std::sqrt(x_norm);
    auto grad_max = [&]() {
        double m = 0;
        for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(g[i]));
        return m;
    };
    if (!opts->fixed_iterations && grad_max() <= 1e-10) {
        summary->termination = 1;
        summary->final_cost = cost;
        return;
    }

    // scaled system: Hs = S H S, gs = S g
    auto Hs = [&](int f, int a, int b) { return H[36 * f + 6 * a + b] * scale[6 * f + a] * scale[6 * f + b]; };
    auto quad = [&](const std::vector<double>& v) {  // v^T Hs v
        double q = 0;
        for (int f = 0; f < W; ++f)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) q += v[6 * f + a] * Hs(f, a, b) * v[6 * f + b];
        return q;
    };

    int iter = 0;
    while (true) {
        if (iter >= opts->max_num_iterations) break;  // NO_CONVERGENCE
        if (radius < 1e-32) break;
        ++iter;
        summary->iterations = iter;

        // ---- DoglegStrategy::ComputeStep ----
        bool solve_ok = true;
        if (!reuse) {
            reuse = true;
            for (int f = 0; f < W; ++f)
                for (int i = 0; i < 6; ++i) {
                    double d = Hs(f, i, i);
                    d = std::min(std::max(d, 1e-6), 1e32);
                    diag[6 * f + i] = std::sqrt(d);
                }
            // ComputeGradient: gradient_ = (J^T r) / diagonal
            for (int i = 0; i < n; ++i) grad[i] = g[i] * scale[i] / diag[i];
            // ComputeCauchyPoint: alpha = |grad|^2 / |J (grad / diag)|^2
            std::vector<double> sg(n);
            double gg = 0;
            for (int i = 0; i < n; ++i) {
                sg[i] = grad[i] / diag[i];
                gg += grad[i] * grad[i];
            }
            alpha = gg / quad(sg);
            // ComputeGaussNewtonStep: (Hs + mu diag^2) y = gs ; gn = -diag * y
            solve_ok = false;
            while (mu < max_mu) {
                bool ok = true;
                for (int f = 0; f < W && ok; ++f) {
                    double A[36], b[6];
                    for (int a = 0; a < 6; ++a) {
                        for (int bb = 0; bb < 6; ++bb) A[6 * a + bb] = Hs(f, a, bb);
                        A[7 * a] += mu * diag[6 * f + a] * diag[6 * f + a];
                        b[a] = g[6 * f + a] * scale[6 * f + a];
                    }
                    ok = chol_solve(A, b, 6);
                    for (int a = 0; a < 6 && ok; ++a) {
                        if (!std::isfinite(b[a])) ok = false;
                        gn[6 * f + a] = b[a];
                    }
                }
                if (!ok) {
                    mu *= mu_increase;
                    continue;
                }
                solve_ok = true;
                break;
            }
            if (solve_ok)
                for (int i = 0; i < n; ++i) gn[i] *= -diag[i];
        }
        bool step_valid = solve_ok;
        double model_cost_change = 0;
        if (solve_ok) {
            // ComputeTraditionalDoglegStep
            double gradient_norm = 0, gn_norm = 0;
            for (int i = 0; i < n; ++i) {
                gradient_norm += grad[i] * grad[i];
                gn_norm += gn[i] * gn[i];
            }
            gradient_norm = std::sqrt(gradient_norm);
            gn_norm = std::sqrt(gn_norm);
            if (gn_norm <= radius) {
                for (int i = 0; i < n; ++i) step[i] = gn[i];
                dogleg_step_norm = gn_norm;
            } else if (gradient_norm * alpha >= radius) {
                for (int i = 0; i < n; ++i) step[i] = -(radius / gradient_norm) * grad[i];
                dogleg_step_norm = radius;
            } else {
                double gdot = 0;
                for (int i = 0; i < n; ++i) gdot += grad[i] * gn[i];
                double b_dot_a = -alpha * gdot;
                double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
                double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gn_norm, 2);
                double c = b_dot_a - a_squared_norm;
                double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
                double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
                double sn = 0;
                for (int i = 0; i < n; ++i) {
                    step[i] = (-alpha * (1.0 - beta)) * grad[i] + beta * gn[i];
                    sn += step[i] * step[i];
                }
                dogleg_step_norm = std::sqrt(sn);
            }
            for (int i = 0; i < n; ++i) step[i] /= diag[i];
            // model_cost_change = -(J step)^T (r + J step / 2) = -(step^T gs + 1/2 step^T Hs step)
            double sg = 0;
            for (int i = 0; i < n; ++i) sg += step[i] * g[i] * scale[i];
            model_cost_change = -(sg + 0.5 * quad(step));
            if (!(model_cost_change > 0.0)) step_valid = false;
        }
        if (!step_valid) {
            // HandleInvalidStep (trust_region_minimizer.cc): the 5th consecutive invalid step
            // (max_num_consecutive_invalid_steps) ends the solve with FAILURE before StepIsInvalid(); Solver::Solve then
            // restores the parameters it was given (Summary::IsSolutionUsable() is false for FAILURE)
            if (++num_invalid >= 5) {
                std::memcpy(x, x_entry.data(), sizeof(double) * n);
                summary->termination = 4;
                break;
            }
            mu *= mu_increase;
            reuse = false;
            if (trace) std::memcpy(trace + (size_t)(iter - 1) * n, x, sizeof(double) * n);
            continue;
        }
        num_invalid = 0;
        double step_norm = 0;
        for (int i = 0; i < n; ++i) {
            delta[i] = step[i] * scale[i];
            xc[i] = x[i] + delta[i];
            step_norm += delta[i] * delta[i];
        }
        step_norm = std::sqrt(step_norm);
        // candidate evaluation (cost; H,g computed alongside and kept if accepted)
        double cand_cost = prob.eval(xc.data(), Hc.data(), gc.data());

        if (!opts->fixed_iterations) {
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) {  // ParameterToleranceReached
                summary->termination = 2;
                if (trace) std::memcpy(trace + (size_t)(iter - 1) * n, x, sizeof(double) * n);
                break;
            }
            if (std::fabs(cost - cand_cost) <= 1e-6 * cost) {  // FunctionToleranceReached
                summary->termination = 3;
                if (trace) std::memcpy(trace + (size_t)(iter - 1) * n, x, sizeof(double) * n);
                break;
            }
        }
        double relative_decrease = (cost - cand_cost) / model_cost_change;
        if (relative_decrease > 1e-3) {
            // HandleSuccessfulStep
            for (int i = 0; i < n; ++i) x[i] = xc[i];
            x_norm = 0;
            for (int i = 0; i < n; ++i) x_norm += x[i] * x[i];
            x_norm = std::sqrt(x_norm);
            H.swap(Hc);
            g.swap(gc);
            cost = cand_cost;
            ++summary->successful;
            if (trace) std::memcpy(trace + (size_t)(iter - 1) * n, x, sizeof(double) * n);
            if (!opts->fixed_iterations && grad_max() <= 1e-10) {
                summary->termination = 1;
                break;
            }
            // StepAccepted
            if (relative_decrease < 0.25) radius *= 0.5;
            if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
            mu = std::max(min_mu, 2.0 * mu / mu_increase);
            reuse = false;
        } else {
            // StepRejected
            radius *= 0.5;
            reuse = true;
            if (trace) std::memcpy(trace + (size_t)(iter - 1) * n, x, sizeof(double) * n);
        }
    }
    summary->final_cost = cost;
}

// ------------------------------------------------------------------------------------------------
// a21  Estimator::Estimate, windowSize != SLIDEWINDOWSIZE branch (Estimator.cpp:1143-1581) on the local maps.
extern "C" int mmlo_estimate_single(const float* corner_feat, int n_corner, const float* surf_feat, int n_surf,
                                    const float* corner_map, int m_corner, const float* surf_map, int m_surf,
                                    const double* exTlb, double* P, double* Qxyzw, int max_outer, int inner_iters,
                                    int* is_degenerate, double* outer_trace) {
    // exRbl = exTlb.R^T ; exPbl = -exRbl * exTlb.t   (:1155-1156)
    double Rbl[9], Pbl[3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rbl[3 * r + c] = exTlb[4 * c + r];
    for (int r = 0; r < 3; ++r)
        Pbl[r] = -1.0 * ((Rbl[3 * r] * exTlb[3] + Rbl[3 * r + 1] * exTlb[7]) + Rbl[3 * r + 2] * exTlb[11]);
    double T_bl[16] = {Rbl[0], Rbl[1], Rbl[2], Pbl[0], Rbl[3], Rbl[4], Rbl[5], Pbl[1],
                       Rbl[6], Rbl[7], Rbl[8], Pbl[2], 0, 0, 0, 1};
    mmlo_kdtree* tc = mmlo_kdtree_build(corner_map, m_corner);
    mmlo_kdtree* ts = mmlo_kdtree_build(surf_map, m_surf);
    std::vector<mmlo_line_factor> lf(n_corner > 0 ? n_corner : 1);
    std::vector<mmlo_plane_factor> pf(n_surf > 0 ? n_surf : 1);
    double thres_dist = 25.0;  // :1207
    *is_degenerate = 0;
    int it_done = 0;
    for (int iterOpt = 0; iterOpt < max_outer; ++iterOpt) {
        // vector2double :937-950
        Quat Q{Qxyzw[0], Qxyzw[1], Qxyzw[2], Qxyzw[3]};
        Vec3 phi = so3_log(Q);
        double x[6] = {P[0], P[1], P[2], phi.x, phi.y, phi.z};
        Quat q_before = Q;
        double t_before[3] = {P[0], P[1], P[2]};
        // transformTobeMapped :1268-1270
        double Rq[9];
        quat_to_matrix(Q, Rq);
        double T[16];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                T[4 * r + c] = (Rq[3 * r] * Rbl[c] + Rq[3 * r + 1] * Rbl[3 + c]) + Rq[3 * r + 2] * Rbl[6 + c];
        }
        Vec3 tq = qrot(Q, mk(Pbl[0], Pbl[1], Pbl[2]));
        T[3] = tq.x + P[0];
        T[7] = tq.y + P[1];
        T[11] = tq.z + P[2];
        T[12] = T[13] = T[14] = 0;
        T[15] = 1;
        int nl = mmlo_associate_lines(corner_feat, n_corner, corner_map, m_corner, tc, T, thres_dist, lf.data(), nullptr);
        int np = mmlo_associate_planes(surf_feat, n_surf, surf_map, m_surf, ts, T, thres_dist, pf.data(), nullptr);
        double min_eigen = mmlo_check_localizability(pf.data(), np);  // :771-775
        if (min_eigen < 3.0) *is_degenerate = 1;
        if (iterOpt == 0)
            thres_dist = 10.0;  // :1377-1381
        else
            thres_dist = 1.0;
        mmlo_solve_opts so;
        so.max_num_iterations = inner_iters;
        so.fixed_iterations = 0;
        so.huber_delta = 0.1 / kLidarM;  // :1221
        so.plan_weight_tan = 0.0;        // :1206
        mmlo_solve_summary sum;
        mmlo_solve_window(lf.data(), &nl, pf.data(), &np, 1, T_bl, &so, x, &sum, nullptr);
        // double2vector :952-964
        P[0] = x[0];
        P[1] = x[1];
        P[2] = x[2];
        Quat qa = so3_exp(mk(x[3], x[4], x[5]));
        Qxyzw[0] = qa.x;
        Qxyzw[1] = qa.y;
        Qxyzw[2] = qa.z;
        Qxyzw[3] = qa.w;
        it_done = iterOpt + 1;
        if (outer_trace) {
            double* o = outer_trace + 7 * iterOpt;
            o[0] = P[0];
            o[1] = P[1];
            o[2] = P[2];
            o[3] = qa.x;
            o[4] = qa.y;
            o[5] = qa.z;
            o[6] = qa.w;
        }
        // :1444-1448  angularDistance: d = a * b.conj ; 2*atan2(|d.vec|, |d.w|)
        Quat dq = qmul(q_before, qconj(qa));
        double deltaR = (2.0 * std::atan2(std::sqrt((dq.x * dq.x + dq.y * dq.y) + dq.z * dq.z), std::fabs(dq.w))) * 180.0 / M_PI;
        double dtv[3] = {t_before[0] - P[0], t_before[1] - P[1], t_before[2] - P[2]};
        double deltaT = std::sqrt((dtv[0] * dtv[0] + dtv[1] * dtv[1]) + dtv[2] * dtv[2]);
        if ((deltaR < 0.05 && deltaT < 0.05) || (iterOpt + 1) == max_outer) break;
    }
    mmlo_kdtree_free(tc);
    mmlo_kdtree_free(ts);
    return it_done;
}

// ------------------------------------------------------------------------------------------------
// unionLidarsAligner.cpp:1077-1153, the numeric core of estimate_timeoffset.
extern "C" int mmlo_time_offset_search(const float* velo_xyz, int n_velo, const float* tf, const float* livox_xyz,
                                       int n_livox, int res, int sliced, float* nn_d2, double* win_err, int capacity,
                                       int* best, double* lowest) {
    // :1080-1082 pcl::transformPointCloud(full_cloud_in, out_cloud, _velo_hori_tf_matrix): float, left to right
    // (PCL 1.8.1 common/impl/transforms.hpp)
    std::vector<float> tv((size_t)3 * (n_velo > 0 ? n_velo : 1));
    for (int i = 0; i < n_velo; ++i) {
        const float x = velo_xyz[3 * i], y = velo_xyz[3 * i + 1], z = velo_xyz[3 * i + 2];
        if (tf) {
            tv[3 * i] = tf[0] * x + tf[1] * y + tf[2] * z + tf[3];
            tv[3 * i + 1] = tf[4] * x + tf[5] * y + tf[6] * z + tf[7];
            tv[3 * i + 2] = tf[8] * x + tf[9] * y + tf[10] * z + tf[11];
        } else {
            tv[3 * i] = x;
            tv[3 * i + 1] = y;
            tv[3 * i + 2] = z;
        }
    }
    // :1084-1103 nearestKSearch(searchPoint, 1, ...): squared distance to the nearest neighbour (FLANN L2_Simple)
    mmlo_kdtree* tree = mmlo_kdtree_build(tv.data(), n_velo);
    for (int i = 0; i < n_livox; ++i) {
        int idx[5];
        float d2[5];
        mmlo_kdtree_knn5(tree, livox_xyz + 3 * i, idx, d2);
        nn_d2[i] = d2[0];
    }
    mmlo_kdtree_free(tree);
    // :1107-1150 sliding windows
    double lowest_error = 1000000.0;
    int cnt = 0, best_cnt = -1;
    while (cnt * res + sliced < n_livox) {
        double sum_error = 0;
        for (int i = cnt * res; i < cnt * res + sliced; i++) {
            const float x = livox_xyz[3 * i], y = livox_xyz[3 * i + 1];
            // `sqrt(pt.x * pt.x + pt.y * pt.y)` on floats: with <math.h> in scope (tf/LinearMath/Scalar.h) the float
            // overload is the one selected
            sum_error += nn_d2[i] + 0.2 * std::sqrt(x * x + y * y);
        }
        if (cnt < capacity && win_err) win_err[cnt] = sum_error;
        if (sum_error < lowest_error) {
            lowest_error = sum_error;
            best_cnt = cnt;
        }
        cnt++;
    }
    *best = best_cnt;
    *lowest = lowest_error;
    return cnt;
}

// ---- helpers for tests ----
extern "C" void mmlo_so3_exp(const double* phi, double* q) {
    Quat r = so3_exp(mk(phi[0], phi[1], phi[2]));
    q[0] = r.x;
    q[1] = r.y;
    q[2] = r.z;
    q[3] = r.w;
}
extern "C" void mmlo_so3_log(const double* q, double* phi) {
    Vec3 r = so3_log(Quat{q[0], q[1], q[2], q[3]});
    phi[0] = r.x;
    phi[1] = r.y;
    phi[2] = r.z;
}
extern "C" void mmlo_eig3_sym(const double* A, double* evals, double* evecs) { eig3_sym(A, evals, evecs); }
extern "C" void mmlo_plane_fit5(const double* A, double* x) { plane_fit5(A, x); }
