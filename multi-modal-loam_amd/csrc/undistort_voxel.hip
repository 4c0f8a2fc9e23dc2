// undistort_voxel.hip -- rows a9 and a10 of SURVEY.md section 8.
//   a9  RemoveLidarDistortion (mm-loam/src/unionPoseEstimation.cpp:402-421): one point per lane, the
//       per-scan quaternion normalisation / angle is hoisted to one lane-uniform prologue.
//   a10 label split + pcl::VoxelGrid centroid down-sample (mm-loam/src/lio/Estimator.cpp:992-1026,
//       leaf sizes :78-80): one workgroup per (slot, kind): order-preserving compaction of the labelled
//       points, float min/max, voxel key, in-LDS bitonic sort of (key, sequence) pairs, one lane per voxel
//       sums its points in input order.
#include <math.h>

#include "mml_internal.h"
#include "undistort_dev.h"

namespace {
using namespace mml_und;

// params: per slot 12 doubles (dR row-major 9, dt 3); derived: per slot 8 doubles written by k_undistort_prep
// (qlc x,y,z,w | theta | sinTheta | 1/sinTheta | linear-branch flag)
__global__ void k_undistort_prep(int count, const double* params, double* derived, int* flags /* 2 per slot of this call */) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= count) return;
    flags[2 * s + 1] = flags[2 * s];  // what k_undistort reads: was the slot undistorted before this call?
    flags[2 * s] |= 2;                // from now on its in-sweep time reads 1 (:419)
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

// One point per lane.  Fast form: the slerp divisions become one multiplication by the per-scan 1/sin(theta), the
// quaternion normalisation uses v_rsq_f64 + Newton; its result differs from the reference expression by ~1e-15
// relative, so the FLOAT it rounds to is the same unless the double lies within 1e-13 of a float rounding
// boundary -- in that case (a few points per million) the reference expression is evaluated.
// (in place on the line-bucketed storage: Velodyne points in [0, cb_n[0]), Livox points in [NV, NV + cb_n[1]))
// "point.normal_x = 1" (:419) is a per-slot flag, not a store per point: k_undistort_prep raises bit 1 of slot_flags[2 b]
// after saving its previous value in slot_flags[2 b + 1]; a slot that is undistorted again reads its time as 1.
__global__ __launch_bounds__(256) void k_undistort(int first, int NT, int NV, const int* cb_n, float4* ln_pts,
                                                  const int* ln_rel, const int* slot_flags, const double* params,
                                                  const double* derived) {
    const int b = blockIdx.y + first;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NT) return;
    if (i < NV ? i >= cb_n[2 * b] : i - NV >= cb_n[2 * b + 1]) return;
    const double* dR = params + 12 * blockIdx.y;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* pp = reinterpret_cast<v4f*>(ln_pts + (size_t)b * NT + i);
    const v4f raw = __builtin_nontemporal_load(pp);
    float4 p = make_float4(raw.x, raw.y, raw.z, raw.w);
    const float s = (slot_flags[2 * b + 1] & 2) ? 1.0f : __int_as_float(__builtin_nontemporal_load(&ln_rel[(size_t)b * NT + i]));
    mml_und::undistort_point(dR, dR + 9, derived + 8 * blockIdx.y, s, p);
    v4f outv = {p.x, p.y, p.z, p.w};
    __builtin_nontemporal_store(outv, pp);
}

// ------------------------------------------------------------------------------------------------------------
constexpr int VX_TAIL = 64;

// Bitonic sort of npad = KPT * VX_THREADS keys, ascending, element e = tid + VX_THREADS * k held by thread tid in
// key[k].  The partner of element e at distance j is e ^ j: for j >= VX_THREADS that is another register of the same
// thread, for j < 64 another lane of the same wavefront (two 32-bit shuffles), and only the four distances in between
// go through LDS -- 22 barrier-separated exchanges for 8192 keys instead of 91.
template <int VX_THREADS, int KPT, int DK>
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
template <int VX_THREADS, int KPT>
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
                    bitonic_reg_step<VX_THREADS, KPT, 1>(key, k, tid);
                else if (dk == 2)
                    bitonic_reg_step<VX_THREADS, KPT, 2>(key, k, tid);
                else
                    bitonic_reg_step<VX_THREADS, KPT, 4>(key, k, tid);
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

// One workgroup per (slot, kind).  LDS: keys[cap] (u64: voxel idx | fused index | where to find the point).
// Two key layouts: scans up to 65536 points carry (voxel 32 | fused index 16 | bucketed position 16); larger ones (up to 2^20
// points: 128 x 2048 rings) carry (voxel 31 | fused index 20 | place in the label list 13) and find the position through the
// list -- their labelled clouds are no larger than a small scan's, only their indices are wider.
template <int VX_THREADS>
__global__ __launch_bounds__(VX_THREADS) void k_voxel(int kind0, int first, int NT, int MF, int B, int cap_y0, int cap_y1, int list_stride, const int* fu_info,
                                                     const float4* ln_pts, const int* ln_gidx,
                                                     float leaf_corner, float leaf_surf, float4* ft0, float4* ft1,
                                                     int* ft_n, unsigned* seq_scratch) {
    constexpr int VX_WAVES = VX_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ int s_wtot[VX_WAVES];
    __shared__ float s_stage[3][VX_THREADS + VX_TAIL];
    __shared__ int s_base;
    __shared__ float s_red6[6][VX_WAVES];
    __shared__ int s_nout;

    const int b = blockIdx.x + first;
    const int kind = blockIdx.y + kind0;
    const int cap = blockIdx.y == 0 ? cap_y0 : cap_y1;  // labelled points this workgroup sorts at most
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* px = ln_pts + (size_t)b * NT;
    const int* gx = ln_gidx + (size_t)b * NT;
    const float leaf = kind == 0 ? leaf_corner : leaf_surf;
    float4* out = (kind == 0 ? ft0 : ft1) + (size_t)b * MF;
    // the labelled points of this (slot, kind): their bucketed positions, listed by the crop pass
    unsigned* seq2idx = seq_scratch + ((size_t)b * 2 + kind) * list_stride;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    static_assert(MML_VOXEL_LDS_CAP <= 8192, "the wide key layout carries the place in the label list in 13 bits");
    const bool wide = NT > 65536;
    const int vshift = wide ? 33 : 32;
    auto key_pos = [&](unsigned long long k) -> unsigned { return wide ? seq2idx[(unsigned)k & 0x1fffu] : (unsigned)k & 0xffffu; };

    // 1. the labelled points were listed by the crop pass (feature.hip k_crop_c) in fused-cloud order; min / max of
    //    their coordinates (getMinMax3D)
    const int nsel = fu_info[8 * b + 6 + kind];
    int cnt = nsel > cap ? cap : nsel;  // capacity overflow is reported through ft_n (negative)
    const bool overflow = nsel > cap;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // four points of this thread's stride per round: the index loads, then the gathers, in flight together
    for (int s0 = tid; s0 < cnt; s0 += 4 * VX_THREADS) {
        unsigned id4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id4[u] = seq2idx[min(s0 + u * VX_THREADS, cnt - 1)];
        float4 p4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p4[u] = px[id4[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (a clamped repeat of the last point changes no extremum)
            mn[0] = fminf(mn[0], p4[u].x);
            mn[1] = fminf(mn[1], p4[u].y);
            mn[2] = fminf(mn[2], p4[u].z);
            mx[0] = fmaxf(mx[0], p4[u].x);
            mx[1] = fmaxf(mx[1], p4[u].y);
            mx[2] = fmaxf(mx[2], p4[u].z);
        }
    }
    // the six extrema reduced together: one shuffle tree each, one exchange through LDS
    float gmn[3], gmx[3];
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            for (int o = 32; o > 0; o >>= 1) {
                mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
                mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
            }
            if (lane == 0) {
                s_red6[c][wave] = mn[c];
                s_red6[3 + c][wave] = mx[c];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r0 = s_red6[c][0], r1 = s_red6[3 + c][0];
            for (int w = 1; w < VX_WAVES; ++w) {
                r0 = fminf(r0, s_red6[c][w]);
                r1 = fmaxf(r1, s_red6[3 + c][w]);
            }
            gmn[c] = r0;
            gmx[c] = r1;
        }
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
    const bool unfiltered = mml_voxel_grid_overflows(gmn, gmx, inv);  // (Estimator.cpp:1015-1024 then gets its input back)
    // keys (voxel idx, fused index, position) of my elements e = tid + VX_THREADS * k; padding sorts to the end.  The
    // fused index (the point's place in [velo_combine ; livox_combine]) orders the points of a voxel as the reference
    // sums them; position and fused index both fit 16 bits on this path (NT <= 65536).
    auto make_key = [&](int sidx) -> unsigned long long {
        if (sidx >= cnt) return ~0ull;
        const unsigned pos = seq2idx[sidx];
        const float4 p = px[pos];
        const int ijk0 = static_cast<int>(floor(p.x * inv) - static_cast<float>(min_b[0]));
        const int ijk1 = static_cast<int>(floor(p.y * inv) - static_cast<float>(min_b[1]));
        const int ijk2 = static_cast<int>(floor(p.z * inv) - static_cast<float>(min_b[2]));
        // (unfiltered: every point its own voxel, in the order of the fused cloud -- the label list is in storage order)
        const int idx = unfiltered ? gx[pos] : ijk0 + ijk1 * mul1 + ijk2 * mul2;
        if (wide) return ((unsigned long long)(unsigned)idx << 33) | ((unsigned long long)(unsigned)gx[pos] << 13) | (unsigned)sidx;
        return ((unsigned long long)(unsigned)idx << 32) | ((unsigned)gx[pos] << 16) | pos;
    };
    // 3. bitonic sort ascending on (voxel idx, sequence): equivalent to a stable sort by voxel idx
    if (cnt <= VX_THREADS) {
        unsigned long long k1[1] = {make_key(tid)};
        bitonic_sort_regs<VX_THREADS, 1>(k1, keys);
    } else if (cnt <= 2 * VX_THREADS) {
        unsigned long long k2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) k2[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<VX_THREADS, 2>(k2, keys);
    } else if (cnt <= 4 * VX_THREADS) {
        unsigned long long k4[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) k4[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<VX_THREADS, 4>(k4, keys);
    } else {
        unsigned long long k8[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) k8[a] = make_key(tid + VX_THREADS * a);
        bitonic_sort_regs<VX_THREADS, 8>(k8, keys);
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
            vox = (unsigned)(keys[s] >> vshift);
            head = (s == 0) || ((unsigned)(keys[s - 1] >> vshift) != vox);
        }
        // the points of this chunk (and VX_TAIL beyond it, for runs that cross into the next chunk) are fetched by their
        // own lanes, all gathers in flight together, so that the sequential per-voxel sums below read LDS instead of
        // chasing two dependent global loads per point
        for (int t = tid; t < VX_THREADS + VX_TAIL; t += VX_THREADS) {
            const int e = c0 + t;
            if (e < cnt) {
                const float4 p = px[key_pos(keys[e])];
                s_stage[0][t] = p.x;
                s_stage[1][t] = p.y;
                s_stage[2][t] = p.z;
            }
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
            while (e < cnt && (unsigned)(keys[e] >> vshift) == vox) {
                const int t = e - c0;
                if (t < VX_THREADS + VX_TAIL) {
                    sx += s_stage[0][t];
                    sy += s_stage[1][t];
                    sz += s_stage[2][t];
                } else {  // a voxel with more than VX_TAIL points across the chunk edge
                    const float4 p = px[key_pos(keys[e])];
                    sx += p.x;
                    sy += p.y;
                    sz += p.z;
                }
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
    hipLaunchKernelGGL(k_undistort_prep, dim3((count + 63) / 64), dim3(64), 0, MML_STREAM(ctx), count, d_params, ctx->d_und + 8 * (size_t)first,
                       ctx->slot_flags + 2 * (size_t)first);
    hipLaunchKernelGGL(k_undistort, grid, dim3(256), 0, MML_STREAM(ctx), first, ctx->NT, ctx->NV, ctx->cb_n, ctx->ln_pts,
                       ctx->ln_rel, ctx->slot_flags, d_params, ctx->d_und + 8 * (size_t)first);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_launch_downsample(mml_ctx* ctx, int first, int count) {
    MmlStageScope t(ctx, "voxel_downsample");
    if (ctx->NT > (1 << 20)) return mml_downsample_big(ctx, first, count);  // indices beyond the key layouts of k_voxel
    // `cap` labelled points per (slot, kind) fit the LDS sort; a slot with more gets ft_n = -1 here and is redone through
    // the global-sort path by mml_downsample_redo_overflow (the label lists hold every labelled point: stride VX_CAP)
    // The corner lists are an order of magnitude shorter than the surf lists (hundreds against thousands of points): they get a
    // 256-thread workgroup with room for 2048 keys (16 KB), so that only the surf half of the launch is made of 1024-thread
    // workgroups holding 64 KB of keys -- two per CU, and in the pipelined step they wait for that room.  (Scans beyond 65536
    // points -- 128 rings: up to 12 800 corner candidates -- get the large workgroup for both kinds.)
    const int cap_surf = MML_VOXEL_LDS_CAP, cap_corner = mml_voxel_cap_corner(ctx);
    auto pad = [](int cap) {
        int npad = 1;
        while (npad < cap) npad <<= 1;
        return (size_t)npad * sizeof(unsigned long long);
    };
    if (cap_corner == cap_surf || count <= 16) {
        // (one launch for both kinds: a handful of scans are a chain of launches, not a question of room on the CUs)
        hipLaunchKernelGGL(k_voxel<1024>, dim3(count, 2), dim3(1024), pad(cap_surf), MML_STREAM(ctx), 0, first, ctx->NT, ctx->MF, ctx->B, cap_corner,
                           cap_surf, ctx->VX_CAP, ctx->fu_info, ctx->ln_pts, ctx->ln_gidx, ctx->cfg.leaf_corner, ctx->cfg.leaf_surf, ctx->ft_xyz[0],
                           ctx->ft_xyz[1], ctx->ft_n, reinterpret_cast<unsigned*>(ctx->vx_keys));
    } else {
        hipLaunchKernelGGL(k_voxel<256>, dim3(count, 1), dim3(256), pad(cap_corner), MML_STREAM(ctx), 0, first, ctx->NT, ctx->MF, ctx->B, cap_corner,
                           cap_corner, ctx->VX_CAP, ctx->fu_info, ctx->ln_pts, ctx->ln_gidx, ctx->cfg.leaf_corner, ctx->cfg.leaf_surf, ctx->ft_xyz[0],
                           ctx->ft_xyz[1], ctx->ft_n, reinterpret_cast<unsigned*>(ctx->vx_keys));
        hipLaunchKernelGGL(k_voxel<1024>, dim3(count, 1), dim3(1024), pad(cap_surf), MML_STREAM(ctx), 1, first, ctx->NT, ctx->MF, ctx->B, cap_surf,
                           cap_surf, ctx->VX_CAP, ctx->fu_info, ctx->ln_pts, ctx->ln_gidx, ctx->cfg.leaf_corner, ctx->cfg.leaf_surf, ctx->ft_xyz[0],
                           ctx->ft_xyz[1], ctx->ft_n, reinterpret_cast<unsigned*>(ctx->vx_keys));
    }
    MML_HIP(hipGetLastError());
    return MML_OK;
}
