// undistort_voxel.hip -- rows a9 and a10 of SURVEY.md section 8.
//   a9  RemoveLidarDistortion (mm-loam/src/unionPoseEstimation.cpp:402-421): one point per lane, the
//       per-scan quaternion normalisation / angle is hoisted to one lane-uniform prologue.
//   a10 label split + pcl::VoxelGrid centroid down-sample (mm-loam/src/lio/Estimator.cpp:992-1026,
//       leaf sizes :78-80): one workgroup per (slot, kind): order-preserving compaction of the labelled
//       points, float min/max, voxel key, in-LDS bitonic sort of (key, sequence) pairs, one lane per voxel
//       sums its points in input order.
#include <math.h>

#include "mml_internal.h"

namespace {

struct Q4 {
    double x, y, z, w;
};
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 vcross(const V3& a, const V3& b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Eigen Vector4d reduction with 2-wide packets: (c0 + c2) + (c1 + c3), coefficients stored x,y,z,w
__device__ __forceinline__ double qdot(const Q4& a, const Q4& b) {
    return (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
}
__device__ __forceinline__ Q4 qnormalized(const Q4& q) {
    double n = sqrt(qdot(q, q));
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
// Eigen quaternionbase_assign_impl<Matrix3d>: rotation matrix (row-major) -> quaternion
__device__ Q4 quat_from_matrix(const double* m) {
    Q4 q;
    double t = m[0] + m[4] + m[8];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t;
        q.y = (m[2] - m[6]) * t;
        q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        double qv[3];
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[3 * k + j] - m[3 * j + k]) * t;
        qv[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        qv[k] = (m[3 * k + i] + m[3 * i + k]) * t;
        q.x = qv[0];
        q.y = qv[1];
        q.z = qv[2];
    }
    return q;
}

// params: per slot 12 doubles (dR row-major 9, dt 3); derived: per slot 8 doubles written by k_undistort_prep
// (qlc x,y,z,w | theta | sinTheta | 1/sinTheta | linear-branch flag)
__global__ void k_undistort_prep(int count, const double* params, double* derived) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= count) return;
    const double* dR = params + 12 * s;
    // Eigen::Quaterniond(dRlc).normalized()  (:410) -- identical for every point of the scan
    const Q4 qlc = qnormalized(quat_from_matrix(dR));
    const double dd = (0.0 * qlc.x + 0.0 * qlc.z) + (0.0 * qlc.y + 1.0 * qlc.w);  // Identity().dot(qlc)
    const double absD = fabs(dd);
    const double one = 1.0 - 2.220446049250313e-16;
    double theta = 0, sinTheta = 1;
    const bool linear = absD >= one;
    if (!linear) {
        theta = acos(absD);
        sinTheta = sin(theta);
    }
    double* o = derived + 8 * s;
    o[0] = qlc.x;
    o[1] = qlc.y;
    o[2] = qlc.z;
    o[3] = qlc.w;
    o[4] = theta;
    o[5] = sinTheta;
    o[6] = 1.0 / sinTheta;
    o[7] = linear ? 1.0 : 0.0;
}

__device__ __forceinline__ double rsqrt_nr(double z) {
    double y = __builtin_amdgcn_rsq(z);
    y = y * (1.5 - (0.5 * z) * (y * y));
    y = y * (1.5 - (0.5 * z) * (y * y));
    return y;
}
// Is the double v farther than tol from every float rounding boundary (the midpoints between adjacent floats)?  A float
// keeps the top 23 of the 52 mantissa bits; the midpoint of v's float cell is the low 29 bits == 2^28, so the distance
// is |low29 - 2^28| units of 2^(e-52), read straight from the bit pattern.  Values below the normal float range are
// never declared safe.
__device__ __forceinline__ bool float_round_safe(double v, double tol) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const int ef = (int)((hi >> 20) & 0x7ffu);
    const int d = abs((int)(lo & 0x1fffffffu) - 0x10000000);
    const double unit = __hiloint2double((ef - 52) << 20, 0);
    return (ef >= 1023 - 126) && ((double)d * unit > tol);
}

// The per-point arithmetic of RemoveLidarDistortion in double, written exactly as the reference (slerp with two
// divisions, normalized() with sqrt + 4 divisions).
__device__ __forceinline__ void undistort_exact(const double* dR, const double* dt, const double* dv, float s, float4& p) {
    const Q4 qlc{dv[0], dv[1], dv[2], dv[3]};
    const double dd = (0.0 * qlc.x + 0.0 * qlc.z) + (0.0 * qlc.y + 1.0 * qlc.w);
    double scale0, scale1;
    const double t = s;
    if (dv[7] != 0.0) {
        scale0 = 1.0 - t;
        scale1 = t;
    } else {
        const double theta = dv[4], sinTheta = dv[5];
        scale0 = sin((1.0 - t) * theta) / sinTheta;
        scale1 = sin((t * theta)) / sinTheta;
    }
    if (dd < 0.0) scale1 = -scale1;
    Q4 q{scale0 * 0.0 + scale1 * qlc.x, scale0 * 0.0 + scale1 * qlc.y, scale0 * 0.0 + scale1 * qlc.z,
         scale0 * 1.0 + scale1 * qlc.w};
    const Q4 dq = qnormalized(q);
    V3 qv = v3(dq.x, dq.y, dq.z);
    V3 v = v3(p.x, p.y, p.z);
    V3 uv = vcross(qv, v);
    uv = v3(uv.x + uv.x, uv.y + uv.y, uv.z + uv.z);
    V3 c2 = vcross(qv, uv);
    V3 startP = v3((v.x + dq.w * uv.x) + c2.x, (v.y + dq.w * uv.y) + c2.y, (v.z + dq.w * uv.z) + c2.z);
    startP = v3(startP.x + s * dt[0], startP.y + s * dt[1], startP.z + s * dt[2]);
    V3 w = v3(startP.x - dt[0], startP.y - dt[1], startP.z - dt[2]);
    p.x = (dR[0] * w.x + dR[3] * w.y) + dR[6] * w.z;
    p.y = (dR[1] * w.x + dR[4] * w.y) + dR[7] * w.z;
    p.z = (dR[2] * w.x + dR[5] * w.y) + dR[8] * w.z;
}

// One point per lane.  Fast form: the slerp divisions become one multiplication by the per-scan 1/sin(theta), the
// quaternion normalisation uses v_rsq_f64 + Newton; its result differs from the reference expression by ~1e-15
// relative, so the FLOAT it rounds to is the same unless the double lies within 1e-11 of a float rounding
// boundary -- in that case (a few points per million) the reference expression is evaluated.
__global__ __launch_bounds__(256) void k_undistort(int first, int NT, const int* fu_info, float4* fu_xyzi,
                                                  float* fu_rel, const double* params, const double* derived) {
    const int b = blockIdx.y + first;
    const int n = fu_info[8 * b];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* dR = params + 12 * blockIdx.y;
    const double* dt = dR + 9;
    const double* dv = derived + 8 * blockIdx.y;
    float4 p = fu_xyzi[(size_t)b * NT + i];
    const float s = fu_rel[(size_t)b * NT + i];
    const double t = s;
    const double qx = dv[0], qy = dv[1], qz = dv[2], qw = dv[3];
    double ax, ay, az, aw;
    const double theta = dv[4];
    if (dv[7] == 0.0 && theta < 0.5) {
        // slerp(Identity, q, t) of a unit quaternion q = (sin(theta) n, +-cos(theta)) is (+-sin(t theta) n, cos(t theta)),
        // already of unit length: the two sines, the divisions and the normalisation of the reference expression
        // collapse to one sine / cosine of a small angle (theta = half the rotation over one sweep), evaluated here by
        // their Taylor polynomials (truncation < 1e-18 for theta < 0.5).  Like the other fast forms the result is
        // only trusted away from float rounding boundaries (below).
        const double x = t * theta, z = x * x;
        double ps = -1.0 / 1307674368000.0;
        ps = __builtin_fma(ps, z, 1.0 / 6227020800.0);
        ps = __builtin_fma(ps, z, -1.0 / 39916800.0);
        ps = __builtin_fma(ps, z, 1.0 / 362880.0);
        ps = __builtin_fma(ps, z, -1.0 / 5040.0);
        ps = __builtin_fma(ps, z, 1.0 / 120.0);
        ps = __builtin_fma(ps, z, -1.0 / 6.0);
        const double sn = __builtin_fma(x * z, ps, x);
        double pc = -1.0 / 87178291200.0;
        pc = __builtin_fma(pc, z, 1.0 / 479001600.0);
        pc = __builtin_fma(pc, z, -1.0 / 3628800.0);
        pc = __builtin_fma(pc, z, 1.0 / 40320.0);
        pc = __builtin_fma(pc, z, -1.0 / 720.0);
        pc = __builtin_fma(pc, z, 1.0 / 24.0);
        pc = __builtin_fma(pc, z, -0.5);
        aw = __builtin_fma(z, pc, 1.0);
        const double k = (qw < 0.0 ? -dv[6] : dv[6]) * sn;
        ax = k * qx;
        ay = k * qy;
        az = k * qz;
    } else {
        double scale0, scale1;
        if (dv[7] != 0.0) {
            scale0 = 1.0 - t;
            scale1 = t;
        } else {
            const double inv = dv[6];
            scale0 = sin((1.0 - t) * theta) * inv;
            scale1 = sin((t * theta)) * inv;
        }
        if (qw < 0.0) scale1 = -scale1;
        ax = scale1 * qx;
        ay = scale1 * qy;
        az = scale1 * qz;
        aw = scale0 + scale1 * qw;
        const double inv_n = rsqrt_nr((ax * ax + az * az) + (ay * ay + aw * aw));
        ax *= inv_n;
        ay *= inv_n;
        az *= inv_n;
        aw *= inv_n;
    }
    const double vx = p.x, vy = p.y, vz = p.z;
    double ux = ay * vz - az * vy, uy = az * vx - ax * vz, uz = ax * vy - ay * vx;
    ux += ux;
    uy += uy;
    uz += uz;
    const double cx = ay * uz - az * uy, cy = az * ux - ax * uz, cz = ax * uy - ay * ux;
    const double wx = (((vx + aw * ux) + cx) + s * dt[0]) - dt[0];
    const double wy = (((vy + aw * uy) + cy) + s * dt[1]) - dt[1];
    const double wz = (((vz + aw * uz) + cz) + s * dt[2]) - dt[2];
    const double ox = (dR[0] * wx + dR[3] * wy) + dR[6] * wz;
    const double oy = (dR[1] * wx + dR[4] * wy) + dR[7] * wz;
    const double oz = (dR[2] * wx + dR[5] * wy) + dR[8] * wz;
    const double tol = 1e-11 * (((fabs(vx) + fabs(vy)) + fabs(vz)) + ((fabs(dt[0]) + fabs(dt[1])) + fabs(dt[2])) + 1e-30);
    if (float_round_safe(ox, tol) & float_round_safe(oy, tol) & float_round_safe(oz, tol)) {
        p.x = ox;
        p.y = oy;
        p.z = oz;
    } else {
        undistort_exact(dR, dt, dv, s, p);
    }
    fu_xyzi[(size_t)b * NT + i] = p;
    fu_rel[(size_t)b * NT + i] = 1.0f;  // :419
}

// ------------------------------------------------------------------------------------------------------------
constexpr int VX_THREADS = 1024;
constexpr int VX_WAVES = VX_THREADS / 64;

__device__ __forceinline__ float block_reduce_minmax(float v, bool is_min, float* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        float other = __shfl_xor(v, o);
        v = is_min ? fminf(v, other) : fmaxf(v, other);
    }
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    float r = s_red[0];
    for (int w = 1; w < VX_WAVES; ++w) r = is_min ? fminf(r, s_red[w]) : fmaxf(r, s_red[w]);
    __syncthreads();
    return r;
}

// Bitonic sort of npad = KPT * VX_THREADS keys, ascending, element e = tid + VX_THREADS * k held by thread tid in
// key[k].  The partner of element e at distance j is e ^ j: for j >= VX_THREADS that is another register of the same
// thread, for j < 64 another lane of the same wavefront (two 32-bit shuffles), and only the four distances in between
// go through LDS -- 22 barrier-separated exchanges for 8192 keys instead of 91.
template <int KPT, int DK>
__device__ __forceinline__ void bitonic_reg_step(unsigned long long (&key)[KPT], int k, int tid) {
    if constexpr (DK < KPT) {
#pragma unroll
        for (int a = 0; a < KPT; ++a) {
            constexpr int dk = DK;
            const int c = a ^ dk;
            if (c > a) {
                const bool up = ((tid + VX_THREADS * a) & k) == 0;
                const unsigned long long x = key[a], y = key[c];
                const bool sw = (x > y) == up;
                key[a] = sw ? y : x;
                key[c] = sw ? x : y;
            }
        }
    }
}
template <int KPT>
__device__ __forceinline__ void bitonic_sort_regs(unsigned long long (&key)[KPT], unsigned long long* lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NP = KPT * VX_THREADS;
#pragma unroll 1
    for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= VX_THREADS) {
                const int dk = j / VX_THREADS;  // 1, 2 or 4: register distance (compile-time in each branch)
                if (dk == 1)
                    bitonic_reg_step<KPT, 1>(key, k, tid);
                else if (dk == 2)
                    bitonic_reg_step<KPT, 2>(key, k, tid);
                else
                    bitonic_reg_step<KPT, 4>(key, k, tid);
            } else if (j < 64) {
#pragma unroll
                for (int a = 0; a < KPT; ++a) {
                    const unsigned long long x = key[a];
                    const unsigned lo = __shfl_xor((unsigned)x, j), hi = __shfl_xor((unsigned)(x >> 32), j);
                    const unsigned long long y = ((unsigned long long)hi << 32) | lo;
                    const bool up = ((tid + VX_THREADS * a) & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const unsigned long long mn = x < y ? x : y, mx = x < y ? y : x;
                    key[a] = (lower == up) ? mn : mx;
                }
            } else {
                __syncthreads();  // the previous LDS round has been read
#pragma unroll
                for (int a = 0; a < KPT; ++a) lds[tid + VX_THREADS * a] = key[a];
                __syncthreads();
#pragma unroll
                for (int a = 0; a < KPT; ++a) {
                    const int e = tid + VX_THREADS * a;
                    const unsigned long long x = key[a], y = lds[e ^ j];
                    const bool up = (e & k) == 0;
                    const bool lower = (e & j) == 0;
                    const unsigned long long mn = x < y ? x : y, mx = x < y ? y : x;
                    key[a] = (lower == up) ? mn : mx;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < KPT; ++a) lds[tid + VX_THREADS * a] = key[a];
    __syncthreads();
}

// One workgroup per (slot, kind).  LDS: keys[cap] (u64: voxel idx << 32 | sequence number).
__global__ __launch_bounds__(VX_THREADS, 8) void k_voxel(int first, int NT, int MF, int B, int cap, const int* fu_info,
                                                     const float4* fu_xyzi, const uint8_t* fu_label,
                                                     float leaf_corner, float leaf_surf, float4* ft0, float4* ft1,
                                                     int* ft_n, unsigned* seq_scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int s_wtot[VX_WAVES];
    __shared__ int s_base;
    __shared__ float s_red[VX_WAVES];
    __shared__ int s_nout;

    const int b = blockIdx.x + first;
    const int kind = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = fu_info[8 * b];
    const float4* px = fu_xyzi + (size_t)b * NT;
    const uint8_t* lab = fu_label + (size_t)b * NT;
    const int want = kind + 1;  // normal_z == 1 corner (:996), == 2 surf (:1003)
    const float leaf = kind == 0 ? leaf_corner : leaf_surf;
    float4* out = (kind == 0 ? ft0 : ft1) + (size_t)b * MF;
    // sequence -> fused index map lives in global scratch (cap entries per (slot, kind))
    unsigned* seq2idx = seq_scratch + ((size_t)b * 2 + kind) * cap;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // 1. the labelled points were listed by the crop pass (feature.hip k_crop_c) in fused-cloud order; min / max of
    //    their coordinates (getMinMax3D)
    const int nsel = fu_info[8 * b + 6 + kind];
    int cnt = nsel > cap ? cap : nsel;  // capacity overflow is reported through ft_n (negative)
    const bool overflow = nsel > cap;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int s = tid; s < cnt; s += VX_THREADS) {
        const float4 p = px[seq2idx[s]];
        mn[0] = fminf(mn[0], p.x);
        mn[1] = fminf(mn[1], p.y);
        mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x);
        mx[1] = fmaxf(mx[1], p.y);
        mx[2] = fmaxf(mx[2], p.z);
    }
    float gmn[3], gmx[3];
    for (int c = 0; c < 3; ++c) {
        gmn[c] = block_reduce_minmax(mn[c], true, s_red);
        gmx[c] = block_reduce_minmax(mx[c], false, s_red);
    }
    if (cnt == 0) {
        if (tid == 0) ft_n[kind * B + b] = 0;
        return;
    }
    // 2. voxel index (PCL 1.8.1 voxel_grid.hpp applyFilter): inverse leaf in float, floor, int, min_b offset
    const float inv = 1.0f / leaf;
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(gmn[c] * inv));
        int max_b = static_cast<int>(floor(gmx[c] * inv));
        div_b[c] = max_b - min_b[c] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    // keys (voxel idx, sequence) of my elements e = tid + VX_THREADS * k; padding sorts to the end
    auto make_key = [&](int sidx) -> unsigned long long {
        if (sidx >= cnt) return ~0ull;
        const float4 p = px[seq2idx[sidx]];
        const int ijk0 = static_cast<int>(floor(p.x * inv) - static_cast<float>(min_b[0]));
        const int ijk1 = static_cast<int>(floor(p.y * inv) - static_cast<float>(min_b[1]));
        const int ijk2 = static_cast<int>(floor(p.z * inv) - static_cast<float>(min_b[2]));
        const int idx = ijk0 + ijk1 * mul1 + ijk2 * mul2;
        return ((unsigned long long)(unsigned)idx << 32) | (unsigned)sidx;
    };
    // 3. bitonic sort ascending on (voxel idx, sequence): equivalent to a stable sort by voxel idx
    if (cnt <= VX_THREADS) {
        unsigned long long k1[1] = {make_key(tid)};
        bitonic_sort_regs<1>(k1, keys);
    } else if (cnt <= 2 * VX_THREADS) {
        unsigned long long k2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) k2[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<2>(k2, keys);
    } else if (cnt <= 4 * VX_THREADS) {
        unsigned long long k4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) k4[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<4>(k4, keys);
    } else {
        unsigned long long k8[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) k8[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<8>(k8, keys);
    }
    // 4. one lane per voxel head: centroid in input order (AccumulatorXYZ: float sum, then / n)
    if (tid == 0) {
        s_base = 0;
        s_nout = 0;
    }
    __syncthreads();
    for (int c0 = 0; c0 < cnt; c0 += VX_THREADS) {
        const int s = c0 + tid;
        bool head = false;
        unsigned vox = 0;
        if (s < cnt) {
            vox = (unsigned)(keys[s] >> 32);
            head = (s == 0) || ((unsigned)(keys[s - 1] >> 32) != vox);
        }
        unsigned long long m = __ballot(head);
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int dst = s_base;
        for (int w = 0; w < wave; ++w) dst += s_wtot[w];
        dst += __popcll(m & lt);
        if (head && dst < MF) {
            float sx = 0, sy = 0, sz = 0;
            int e = s;
            while (e < cnt && (unsigned)(keys[e] >> 32) == vox) {
                float4 p = px[seq2idx[(unsigned)(keys[e] & 0xffffffffu)]];
                sx += p.x;
                sy += p.y;
                sz += p.z;
                ++e;
            }
            float c = static_cast<float>(e - s);
            out[dst] = make_float4(sx / c, sy / c, sz / c, 0.f);
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < VX_WAVES; ++w) t += s_wtot[w];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int nout = s_base;
        if (overflow || nout > MF) nout = -1;  // MML_ERR_CAPACITY at the host
        ft_n[kind * B + b] = nout;
    }
}

}  // namespace

int mml_launch_undistort(mml_ctx* ctx, int first, int count, const double* d_params) {
    MmlStageScope t(ctx, "undistort");
    dim3 grid((ctx->NT + 255) / 256, count);
    hipLaunchKernelGGL(k_undistort_prep, dim3((count + 63) / 64), dim3(64), 0, MML_STREAM(ctx), count, d_params, ctx->d_und + 8 * (size_t)first);
    hipLaunchKernelGGL(k_undistort, grid, dim3(256), 0, MML_STREAM(ctx), first, ctx->NT, ctx->fu_info, ctx->fu_xyzi,
                       ctx->fu_rel, d_params, ctx->d_und + 8 * (size_t)first);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_launch_downsample(mml_ctx* ctx, int first, int count) {
    MmlStageScope t(ctx, "voxel_downsample");
    if (ctx->VX_CAP > 8192) return mml_downsample_big(ctx, first, count);  // labelled clouds beyond the LDS sort
    const int cap = ctx->VX_CAP;
    int npad = 1;
    while (npad < cap) npad <<= 1;
    size_t lds = (size_t)npad * sizeof(unsigned long long);
    hipLaunchKernelGGL(k_voxel, dim3(count, 2), dim3(VX_THREADS), lds, MML_STREAM(ctx), first, ctx->NT, ctx->MF, ctx->B,
                       cap, ctx->fu_info, ctx->fu_xyzi, ctx->fu_label, ctx->cfg.leaf_corner, ctx->cfg.leaf_surf,
                       ctx->ft_xyz[0], ctx->ft_xyz[1], ctx->ft_n, reinterpret_cast<unsigned*>(ctx->vx_keys));
    MML_HIP(hipGetLastError());
    return MML_OK;
}
