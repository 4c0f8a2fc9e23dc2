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

#ifdef MML_VX_TIMING
// phase clocks of one k_voxel workgroup (kind MML_VX_TIMING of the 518th slot of the launch; tools/voxel_phases.py)
__device__ unsigned long long g_vx_dbg[16];
#define VX_MARK(id)                                    \
    do {                                               \
        if (vx_dbg) {                                  \
            const unsigned long long now_ = clock64(); \
            g_vx_dbg[id] += now_ - vx_prev;            \
            vx_prev = now_;                            \
        }                                              \
    } while (0)
extern "C" int mml_debug_vx_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_vx_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vx_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#define VX_DBG_PARAMS , bool vx_dbg, unsigned long long& vx_prev
#define VX_DBG_ARGS , vx_dbg, vx_prev
#else
#define VX_MARK(id)
#define VX_DBG_PARAMS
#define VX_DBG_ARGS
#endif
// ------------------------------------------------------------------------------------------------------------
constexpr int VX_TAIL = 64;

// inclusive scan over the 64 lanes of a wavefront
__device__ __forceinline__ int vx_wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(v, o);
        if (lane >= o) v += y;
    }
    return v;
}

// Stable LSD radix sort (8-bit digits) of the workgroup's cnt <= KPT * VX_THREADS keys on their bits [lo_bit, 64), ascending; the
// sorted keys end up in lds[0, cnt).  Element e = (wave * KPT + a) * 64 + lane is key[a] of that lane: a wavefront holds a
// contiguous run of the current order, so "stable" = (wavefront, round a, lane).  Per pass: the rank of an element among the
// elements of its digit inside its round comes from eight ballots; a per-wavefront digit histogram (LDS, 16 bits a count) gives
// the elements of the digit in earlier rounds of the wavefront; one scan over (digit, wavefront) gives the rest.  Digits on which all
// keys agree are skipped (voxel indices of a room-sized scan vary in 17-20 bits: 5 passes, 4 barriers each, ~45 vector
// instructions per key and pass -- the bitonic network this replaces ran 78 compare-exchange stages of ~10 instructions on
// 64-bit keys with 44 barriers for 4096 keys: 55 % of the kernel's instructions).
// hist: VX_WAVES x 256 u16 (LDS), wtot: VX_WAVES + 2 ints (LDS).
template <int VX_THREADS, int KPT>
__device__ __forceinline__ void radix_sort_lds(unsigned long long (&key)[KPT], int cnt, int lo_bit, unsigned long long* lds,
                                               unsigned short* hist, int* wtot, unsigned long long* s_red VX_DBG_PARAMS) {
    constexpr int VX_WAVES = VX_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // the bits on which the live keys differ
    unsigned long long o = 0ull, n = ~0ull;
#pragma unroll
    for (int a = 0; a < KPT; ++a) {
        const int e = (wave * KPT + a) * 64 + lane;
        if (e < cnt) {
            o |= key[a];
            n &= key[a];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        o |= ((unsigned long long)__shfl_xor((unsigned)(o >> 32), d) << 32) | __shfl_xor((unsigned)o, d);
        n &= ((unsigned long long)__shfl_xor((unsigned)(n >> 32), d) << 32) | __shfl_xor((unsigned)n, d);
    }
    if (lane == 0) {
        s_red[2 * wave] = o;
        s_red[2 * wave + 1] = n;
    }
    __syncthreads();
    o = 0ull;
    n = ~0ull;
#pragma unroll
    for (int w = 0; w < VX_WAVES; ++w) {
        o |= s_red[2 * w];
        n &= s_red[2 * w + 1];
    }
    const unsigned long long diff = (o ^ n) >> lo_bit << lo_bit;
    bool in_lds = false;  // do the registers hold the current order (false) or was it just scattered to lds (true)?
#pragma unroll 1
    for (int shift = lo_bit; shift < 64; shift += 8) {
        if (((diff >> shift) & 0xffull) == 0ull) continue;  // (workgroup-uniform)
        if (in_lds) {
            __syncthreads();  // the scatter of the previous pass has landed
#pragma unroll
            for (int a = 0; a < KPT; ++a) {
                const int e = (wave * KPT + a) * 64 + lane;
                if (e < cnt) key[a] = lds[e];
            }
        }
        // this wavefront's histogram row
        for (int k = lane; k < 256; k += 64) hist[wave * 256 + k] = 0;
        unsigned rd[KPT];  // rank inside the wavefront (16 bits) | digit << 16
#pragma unroll
        for (int a = 0; a < KPT; ++a) {
            const int e = (wave * KPT + a) * 64 + lane;
            const bool live = e < cnt;
            const unsigned d = (unsigned)(key[a] >> shift) & 255u;
            unsigned long long eq = __ballot(live);
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const unsigned long long m = __ballot((d >> bit) & 1u);
                eq &= ((d >> bit) & 1u) ? m : ~m;
            }
            __builtin_amdgcn_wave_barrier();
            const int before = live ? (int)hist[wave * 256 + d] : 0;  // elements of the digit in the earlier rounds of this wavefront
            __builtin_amdgcn_wave_barrier();
            if (live && (eq & lt) == 0ull) hist[wave * 256 + d] = (unsigned short)(before + __popcll(eq));
            rd[a] = (unsigned)(before + __popcll(eq & lt)) | (d << 16);
        }
        __syncthreads();
        // exclusive scan over (digit, wavefront), digit-major: thread t owns entries 4 t .. 4 t + 3 of that order
        {
            constexpr int PER = VX_WAVES * 256 / VX_THREADS;  // = 4
            static_assert(PER == 4, "four (digit, wavefront) entries per thread");
            int c[PER], run = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int idx = PER * tid + q, d = idx / VX_WAVES, w = idx % VX_WAVES;
                c[q] = hist[w * 256 + d];
                run += c[q];
            }
            const int incl = vx_wave_incl_scan(run);
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            int base = incl - run;
            for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int idx = PER * tid + q, d = idx / VX_WAVES, w = idx % VX_WAVES;
                hist[w * 256 + d] = (unsigned short)base;
                base += c[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < KPT; ++a) {
            const int e = (wave * KPT + a) * 64 + lane;
            if (e < cnt) lds[(int)hist[wave * 256 + (rd[a] >> 16)] + (int)(rd[a] & 0xffffu)] = key[a];
        }
        in_lds = true;
    }
    if (!in_lds) {  // nothing to sort on (one key, or all equal above lo_bit): the keys go to lds as they are
#pragma unroll
        for (int a = 0; a < KPT; ++a) {
            const int e = (wave * KPT + a) * 64 + lane;
            if (e < cnt) lds[e] = key[a];
        }
    }
    __syncthreads();
}

// One workgroup per (slot, kind).  LDS: keys[cap] (u64: voxel idx | fused index | where to find the point).
// Two key layouts: scans up to 65536 points carry (voxel 32 | fused index 16 | bucketed position 16); larger ones (up to 2^20
// points: 128 x 2048 rings) carry (voxel 31 | fused index 20 | place in the label list 13) and find the position through the
// list -- their labelled clouds are no larger than a small scan's, only their indices are wider.
struct VoxelArgs {
    int kind0, first, NT, MF, B, cap_y0, cap_y1, list_stride;
    int skip_above, only_above;  // > 0: leave the slots with more labelled points than this to another launch / take only those
    int zero_big;                // this launch resets the counter of the list below (the first launch of the chain)
    int* big;                    // [2 B]: big[first] = number of slots the short form left, big[B + first + k] = the k-th of them
    const int* fu_info;
    const float4* ln_pts;
    const int* ln_gidx;
    float leaf_corner, leaf_surf;
    float4* ft0;
    float4* ft1;
    int* ft_n;
    const unsigned* seq_scratch;
    const int* seq_gidx;  // the listed points' fused indices (label_append), parallel to seq_scratch
};
template <int VX_THREADS>
struct VoxelLds {
    static constexpr int VX_WAVES = VX_THREADS / 64;
    // the radix sort's histograms and the centroid pass's staging rows never live at the same time
    union {
        float stage[3][VX_THREADS + VX_TAIL];
        unsigned short hist[VX_WAVES * 256];
    } u;
    int wtot[VX_WAVES + 2];
    unsigned long long red[2 * VX_WAVES];
    float red6[6][VX_WAVES];
    int base, nout;
};

// the part of k_voxel that depends on the number of keys per thread
template <int VX_THREADS, int KPT>
__device__ __forceinline__ void voxel_sort(const VoxelArgs& A, VoxelLds<VX_THREADS>& S, unsigned long long* keys, int b, int kind, int cnt,
                                           const unsigned* seq2idx, const float4* px, const int* glist, bool wide, float leaf VX_DBG_PARAMS) {
    constexpr int VX_WAVES = VX_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1. the labelled points of this (slot, kind) were listed by the selection kernels (feature.hip label_append), in no particular
    //    order, position and fused index side by side; every lane fetches its KPT points ONCE (position, coordinates, fused index
    //    stay in registers; the fused index comes from the list -- until round 6 it was a second gather, one more 64-byte line per
    //    16-byte point), min / max of the coordinates (getMinMax3D)
    // (up to four points per lane stay in registers between the two uses; eight are fetched again for the keys -- from the L2 --:
    //  their 40 registers across the reduction would not fit the 64 the kernel is held to)
    constexpr bool KEEP = KPT <= 4;
    constexpr int KR = KEEP ? KPT : 1;
    unsigned pos[KPT];
    float x[KR], y[KR], z[KR];
    int gi[KR];
#pragma unroll
    for (int a = 0; a < KPT; ++a) pos[a] = seq2idx[min((wave * KPT + a) * 64 + lane, cnt - 1)];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int a0 = 0; a0 < KPT; a0 += 4) {
        float4 p4[4];
#pragma unroll
        for (int a = a0; a < a0 + 4 && a < KPT; ++a) p4[a - a0] = px[pos[a]];
#pragma unroll
        for (int a = a0; a < a0 + 4 && a < KPT; ++a) {  // (a clamped repeat of the last point changes no extremum)
            const float4 p = p4[a - a0];
            if constexpr (KEEP) {
                x[a] = p.x;
                y[a] = p.y;
                z[a] = p.z;
                gi[a] = glist[min((wave * KPT + a) * 64 + lane, cnt - 1)];
            }
            mn[0] = fminf(mn[0], p.x);
            mn[1] = fminf(mn[1], p.y);
            mn[2] = fminf(mn[2], p.z);
            mx[0] = fmaxf(mx[0], p.x);
            mx[1] = fmaxf(mx[1], p.y);
            mx[2] = fmaxf(mx[2], p.z);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    VX_MARK(4);
    float gmn[3], gmx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
        if (lane == 0) {
            S.red6[c][wave] = mn[c];
            S.red6[3 + c][wave] = mx[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r0 = S.red6[c][0], r1 = S.red6[3 + c][0];
        for (int w = 1; w < VX_WAVES; ++w) {
            r0 = fminf(r0, S.red6[c][w]);
            r1 = fmaxf(r1, S.red6[3 + c][w]);
        }
        gmn[c] = r0;
        gmx[c] = r1;
    }
    VX_MARK(5);
    // 2. voxel index (PCL 1.8.1 voxel_grid.hpp applyFilter): inverse leaf in float, floor, int, min_b offset
    const float inv = 1.0f / leaf;
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(gmn[c] * inv));
        const int max_b = static_cast<int>(floor(gmx[c] * inv));
        div_b[c] = max_b - min_b[c] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    const bool unfiltered = mml_voxel_grid_overflows(gmn, gmx, inv);  // (Estimator.cpp:1015-1024 then gets its input back)
    // keys (voxel idx, fused index, position).  The fused index (the point's place in [velo_combine ; livox_combine]) orders the
    // points of a voxel as the reference sums them; it is unique, so the sorted order does not depend on the order of the list.
    unsigned long long key[KPT];
#pragma unroll
    for (int a = 0; a < KPT; ++a) {
        const int sidx = (wave * KPT + a) * 64 + lane;
        float px_, py_, pz_;
        int g;
        if constexpr (KEEP) {
            px_ = x[a];
            py_ = y[a];
            pz_ = z[a];
            g = gi[a];
        } else {
            const float4 p = px[pos[a]];
            px_ = p.x;
            py_ = p.y;
            pz_ = p.z;
            g = glist[min(sidx, cnt - 1)];
        }
        const int ijk0 = static_cast<int>(floor(px_ * inv) - static_cast<float>(min_b[0]));
        const int ijk1 = static_cast<int>(floor(py_ * inv) - static_cast<float>(min_b[1]));
        const int ijk2 = static_cast<int>(floor(pz_ * inv) - static_cast<float>(min_b[2]));
        // (unfiltered: every point its own voxel, in the order of the fused cloud)
        const int idx = unfiltered ? g : ijk0 + ijk1 * mul1 + ijk2 * mul2;
        key[a] = wide ? (((unsigned long long)(unsigned)idx << 33) | ((unsigned long long)(unsigned)g << 13) | (unsigned)sidx)
                      : (((unsigned long long)(unsigned)idx << 32) | ((unsigned)g << 16) | pos[a]);
    }
    // 3. ascending on (voxel idx, fused index): a stable sort by voxel idx of the fused order
    VX_MARK(6);
    radix_sort_lds<VX_THREADS, KPT>(key, cnt, wide ? 13 : 16, keys, S.u.hist, S.wtot, S.red VX_DBG_ARGS);
    VX_MARK(7);
}

// (two 1024-thread workgroups per CU -- eight wavefronts per SIMD -- is what hides this kernel's barriers and gathers: held to 64 registers)
#ifndef MML_VOXEL_WAVES
#define MML_VOXEL_WAVES 8
#endif
// one (slot, kind) by one workgroup
template <int VX_THREADS>
__device__ __forceinline__ void voxel_slot(const VoxelArgs& A, VoxelLds<VX_THREADS>& S, unsigned long long* keys, int b, int kind, int cap) {
    constexpr int VX_WAVES = VX_THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = A.NT, MF = A.MF;
    const float4* px = A.ln_pts + (size_t)b * NT;
    const float leaf = kind == 0 ? A.leaf_corner : A.leaf_surf;
    float4* out = (kind == 0 ? A.ft0 : A.ft1) + (size_t)b * MF;
    const unsigned* seq2idx = A.seq_scratch + ((size_t)b * 2 + kind) * A.list_stride;
    const int* glist = A.seq_gidx + ((size_t)b * 2 + kind) * A.list_stride;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    static_assert(MML_VOXEL_LDS_CAP <= 8192, "the wide key layout carries the place in the label list in 13 bits");
    const bool wide = NT > 65536;
    const int vshift = wide ? 33 : 32;
    auto key_pos = [&](unsigned long long k) -> unsigned { return wide ? seq2idx[(unsigned)k & 0x1fffu] : (unsigned)k & 0xffffu; };

#ifdef MML_VX_TIMING
    const bool vx_dbg = tid == 64 && kind == MML_VX_TIMING && blockIdx.x == 517 && VX_THREADS == 512;
    unsigned long long vx_prev = clock64();
#endif
    const int nsel = A.fu_info[8 * b + 6 + kind];
    if (A.skip_above > 0 && nsel > A.skip_above) {  // (workgroup-uniform) listed for the launch of the large form
        if (tid == 0) A.big[A.B + A.first + atomicAdd(&A.big[A.first], 1)] = b;
        return;
    }
    const int cnt = nsel > cap ? cap : nsel;  // capacity overflow is reported through ft_n (negative)
    const bool overflow = nsel > cap;
    if (cnt == 0) {
        if (tid == 0) A.ft_n[kind * A.B + b] = 0;
        return;
    }
    if (cnt <= VX_THREADS)
        voxel_sort<VX_THREADS, 1>(A, S, keys, b, kind, cnt, seq2idx, px, glist, wide, leaf VX_DBG_ARGS);
    else if (cnt <= 2 * VX_THREADS)
        voxel_sort<VX_THREADS, 2>(A, S, keys, b, kind, cnt, seq2idx, px, glist, wide, leaf VX_DBG_ARGS);
    else if (cnt <= 4 * VX_THREADS)
        voxel_sort<VX_THREADS, 4>(A, S, keys, b, kind, cnt, seq2idx, px, glist, wide, leaf VX_DBG_ARGS);
    else
        voxel_sort<VX_THREADS, 8>(A, S, keys, b, kind, cnt, seq2idx, px, glist, wide, leaf VX_DBG_ARGS);
    VX_MARK(0);
    // 4. one lane per voxel head: centroid in input order (AccumulatorXYZ: float sum, then / n)
    if (tid == 0) {
        S.base = 0;
        S.nout = 0;
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
                S.u.stage[0][t] = p.x;
                S.u.stage[1][t] = p.y;
                S.u.stage[2][t] = p.z;
            }
        }
        unsigned long long m = __ballot(head);
        if (lane == 0) S.wtot[wave] = __popcll(m);
        __syncthreads();
        VX_MARK(1);
        int dst = S.base;
        for (int w = 0; w < wave; ++w) dst += S.wtot[w];
        dst += __popcll(m & lt);
        if (head && dst < MF) {
            float sx = 0, sy = 0, sz = 0;
            int e = s;
            while (e < cnt && (unsigned)(keys[e] >> vshift) == vox) {
                const int t = e - c0;
                if (t < VX_THREADS + VX_TAIL) {
                    sx += S.u.stage[0][t];
                    sy += S.u.stage[1][t];
                    sz += S.u.stage[2][t];
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
        VX_MARK(2);
        __syncthreads();
        VX_MARK(3);
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < VX_WAVES; ++w) t += S.wtot[w];
            S.base += t;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int nout = S.base;
        if (overflow || nout > MF) nout = -1;  // MML_ERR_CAPACITY at the host
        A.ft_n[kind * A.B + b] = nout;
    }
}
template <int VX_THREADS, bool LIST = false>
__global__ __launch_bounds__(VX_THREADS) __attribute__((amdgpu_waves_per_eu(MML_VOXEL_WAVES))) void k_voxel(VoxelArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ VoxelLds<VX_THREADS> S;
    const int kind = blockIdx.y + A.kind0;
    const int cap = blockIdx.y == 0 ? A.cap_y0 : A.cap_y1;  // labelled points this workgroup sorts at most
    if constexpr (LIST) {
        // the slots the short form listed, by a grid that does not grow with the batch: a launch of one 1024-thread workgroup with
        // 79 KB of LDS per slot -- nearly all of them returning at once -- waited for that room on a CU 4096 times over (0.63 ms per
        // 4096-slot launch next to the other lane's kernels, 0.085 on an empty device)
        const int nbig = A.big[A.first];
        for (int e = blockIdx.x; e < nbig; e += gridDim.x) {
            voxel_slot<VX_THREADS>(A, S, keys, __builtin_amdgcn_readfirstlane(A.big[A.B + A.first + e]), kind, cap);
            __syncthreads();  // (the keys and the staging rows are free again)
        }
    } else {
        if (A.zero_big && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) A.big[A.first] = 0;
        voxel_slot<VX_THREADS>(A, S, keys, (int)blockIdx.x + A.first, kind, cap);
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
    // (ft_n changes: the association statistics, computed lazily from ft_n as it is when somebody asks, are stale from here on)
    for (int i = 0; i < count; ++i) ctx->stats_stale[first + i] = 1;
    if (ctx->NT > (1 << 20)) return mml_downsample_big(ctx, first, count);  // indices beyond the key layouts of k_voxel
    // `cap` labelled points per (slot, kind) fit the LDS sort; a slot with more gets ft_n = -1 here and is redone through
    // the global-sort path by mml_downsample_redo_overflow (the label lists hold every labelled point: stride VX_CAP)
    // The corner lists are an order of magnitude shorter than the surf lists (hundreds against thousands of points): they get a
    // 256-thread workgroup with room for 2048 keys (16 KB), so that only the surf half of the launch is made of 1024-thread
    // workgroups holding 64 KB of keys -- two per CU, and in the pipelined step they wait for that room.  (Scans beyond 65536
    // points -- 128 rings: up to 12 800 corner candidates -- get the large workgroup for both kinds.)
    const int cap_surf = MML_VOXEL_LDS_CAP, cap_corner = mml_voxel_cap_corner(ctx);
    auto lds = [](int cap) { return (size_t)cap * sizeof(unsigned long long); };
    VoxelArgs A;
    A.first = first;
    A.NT = ctx->NT;
    A.MF = ctx->MF;
    A.B = ctx->B;
    A.list_stride = ctx->VX_CAP;
    A.fu_info = ctx->fu_info;
    A.ln_pts = ctx->ln_pts;
    A.ln_gidx = ctx->ln_gidx;
    A.leaf_corner = ctx->cfg.leaf_corner;
    A.leaf_surf = ctx->cfg.leaf_surf;
    A.ft0 = ctx->ft_xyz[0];
    A.ft1 = ctx->ft_xyz[1];
    A.ft_n = ctx->ft_n;
    A.seq_scratch = reinterpret_cast<const unsigned*>(ctx->vx_keys);
    A.seq_gidx = ctx->vx_gidx;
    A.skip_above = 0;
    A.only_above = 0;
    A.zero_big = 0;
    A.big = ctx->vx_big;
    if (cap_corner == cap_surf || count <= 16) {
        // (one launch for both kinds: a handful of scans are a chain of launches, not a question of room on the CUs)
        A.kind0 = 0;
        A.cap_y0 = cap_corner;
        A.cap_y1 = cap_surf;
        hipLaunchKernelGGL(k_voxel<1024>, dim3(count, 2), dim3(1024), lds(cap_surf), MML_STREAM(ctx), A);
    } else {
        A.kind0 = 0;
        A.cap_y0 = A.cap_y1 = cap_corner;
        A.zero_big = 1;  // (the list of this launch chain starts empty: launches of other lanes cover other slots, big[first] is theirs alone)
        hipLaunchKernelGGL(k_voxel<256>, dim3(count, 1), dim3(256), lds(cap_corner), MML_STREAM(ctx), A);
        A.zero_big = 0;
        // The surf lists of a 52.8 k-point scan hold ~3 000 points: they get 512-thread workgroups with room for 4096 keys (32 KB:
        // four per CU where the 1024-thread form with its 64 KB runs two; 0.222 -> 0.178 ms per 1024 scans); the slots with more
        // are listed by it and left to a second launch of the large form: a grid of at most one workgroup per CU that walks the list.
        constexpr int cap_mid = 4096;
        static_assert(cap_mid < MML_VOXEL_LDS_CAP, "the 512-thread form takes the short lists only");
        A.kind0 = 1;
        A.cap_y0 = A.cap_y1 = cap_mid;
        A.skip_above = cap_mid;
        hipLaunchKernelGGL(k_voxel<512>, dim3(count, 1), dim3(512), lds(cap_mid), MML_STREAM(ctx), A);
        A.cap_y0 = A.cap_y1 = cap_surf;
        A.skip_above = 0;
        A.only_above = cap_mid;
        hipLaunchKernelGGL((k_voxel<1024, true>), dim3(count < 256 ? count : 256, 1), dim3(1024), lds(cap_surf), MML_STREAM(ctx), A);
    }
    MML_HIP(hipGetLastError());
    return MML_OK;
}
