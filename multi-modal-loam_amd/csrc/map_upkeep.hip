// Device-side local map upkeep: Estimator::MapIncrementLocal (Estimator.cpp:1585-1643) together with the clear() of
// laserCloud{Corner,Surf}FromLocal that precedes every call (:1083-1085, :1125-1127).
//
// The reference keeps the last localMapWindowSize = 50 key scans' features as 50 PCL clouds in the world frame,
// concatenates them and runs pcl::VoxelGrid over the concatenation; kdtree->setInputCloud then rebuilds both trees on
// the next Estimate() (:1159-1167).  Here the ring lives in HBM next to the scan slots it is fed from:
//   k_to_world    : pointAssociateToMap (Map_Manager.cpp:75-89) of one slot's down-sampled stack into ring slot
//                   localMapID % 50
//   k_concat      : the ring in slot order = the cloud the filter sees
//   voxel filter  : bounding box -> PCL voxel index -> rocPRIM radix sort of (index, position) pairs (stable, so
//                   the points of a voxel stay in input order = the order the oracle sums them in) -> head flags ->
//                   exclusive scan -> one lane per voxel sums its run and writes the centroid
//   grid build    : the filtered cloud is already where mml_build_grid_device expects it
// Nothing but the two cloud sizes crosses PCIe.  Compiled with -ffp-contract=off.
#include <math.h>

#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "mml_internal.h"

namespace {

struct Tf12 {
    double m[12];
};

__global__ void k_to_world(const float4* feat, int n, Tf12 T, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 f = feat[i];
    const double x = f.x, y = f.y, z = f.z;
    float4 o;
    o.x = (float)(((T.m[0] * x + T.m[1] * y) + T.m[2] * z) + T.m[3]);
    o.y = (float)(((T.m[4] * x + T.m[5] * y) + T.m[6] * z) + T.m[7]);
    o.z = (float)(((T.m[8] * x + T.m[9] * y) + T.m[10] * z) + T.m[11]);
    o.w = 0.f;
    out[i] = o;
}

struct RingOffsets {
    int off[mml_ctx::LOCAL_WINDOW + 1];
};

// grid.y = ring slot
__global__ void k_concat(const float4* ring, int MF, RingOffsets R, float4* cat) {
    const int s = blockIdx.y;
    const int n = R.off[s + 1] - R.off[s];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        cat[R.off[s] + i] = ring[(size_t)s * MF + i];
}

__global__ void k_minmax(const float4* pts, int m, float* out /* 6: min xyz, max xyz */) {
    __shared__ float s[6][4];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        mn[0] = fminf(mn[0], p.x);
        mn[1] = fminf(mn[1], p.y);
        mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x);
        mx[1] = fmaxf(mx[1], p.y);
        mx[2] = fmaxf(mx[2], p.z);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
        if (lane == 0) {
            s[c][wave] = mn[c];
            s[3 + c][wave] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = s[c][0];
        for (int w = 1; w < 4; ++w) v = (c < 3) ? fminf(v, s[c][w]) : fmaxf(v, s[c][w]);
        // min / max of floats through their order-preserving integer image
        int* a = reinterpret_cast<int*>(out + c);
        const int iv = __float_as_int(v);
        const int key = iv >= 0 ? iv : (iv ^ 0x7fffffff);
        if (c < 3)
            atomicMin(a, key);
        else
            atomicMax(a, key);
    }
}
__device__ __forceinline__ float unkey(int key) { return __int_as_float(key >= 0 ? key : (key ^ 0x7fffffff)); }

// PCL 1.8.1 voxel_grid.hpp applyFilter: inverse leaf in float, floor, int, min_b offset (same arithmetic as k_voxel)
__global__ void k_vox_keys(const float4* pts, int m, const int* bbox_keys, float leaf, unsigned* keys, unsigned* vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float inv = 1.0f / leaf;
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(unkey(bbox_keys[c]) * inv));
        const int max_b = static_cast<int>(floor(unkey(bbox_keys[3 + c]) * inv));
        div_b[c] = max_b - min_b[c] + 1;
    }
    const float4 p = pts[i];
    const int ijk0 = static_cast<int>(floor(p.x * inv) - static_cast<float>(min_b[0]));
    const int ijk1 = static_cast<int>(floor(p.y * inv) - static_cast<float>(min_b[1]));
    const int ijk2 = static_cast<int>(floor(p.z * inv) - static_cast<float>(min_b[2]));
    const float bmn[3] = {unkey(bbox_keys[0]), unkey(bbox_keys[1]), unkey(bbox_keys[2])};
    const float bmx[3] = {unkey(bbox_keys[3]), unkey(bbox_keys[4]), unkey(bbox_keys[5])};
    // (more than INT_MAX voxels in the box: PCL returns the cloud unfiltered = every point its own voxel, in input order)
    keys[i] = mml_voxel_grid_overflows(bmn, bmx, inv) ? (unsigned)i : (unsigned)(ijk0 + ijk1 * div_b[0] + ijk2 * (div_b[0] * div_b[1]));
    vals[i] = (unsigned)i;
}

__global__ void k_vox_heads(const unsigned* keys, int m, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one lane per voxel: AccumulatorXYZ (float sum in input order, then / n)
__global__ void k_vox_centroid(const float4* pts, const unsigned* keys, const unsigned* vals, const int* flag,
                               const int* pos, int m, int cap, float4* out, int* n_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    if (s == m - 1) *n_out = pos[s] + flag[s];
    if (!flag[s]) return;
    const int dst = pos[s];
    if (dst >= cap) return;
    const unsigned vox = keys[s];
    float sx = 0, sy = 0, sz = 0;
    int e = s;
    while (e < m && keys[e] == vox) {
        const float4 p = pts[vals[e]];
        sx += p.x;
        sy += p.y;
        sz += p.z;
        ++e;
    }
    const float c = static_cast<float>(e - s);
    out[dst] = make_float4(sx / c, sy / c, sz / c, 0.f);
}

int ensure_tmp(mml_ctx* ctx, size_t need) {
    if (need > ctx->sort_tmp_bytes) {
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
        if (ctx->sort_tmp) MML_HIP(hipFree(ctx->sort_tmp));
        MML_HIP(hipMalloc(&ctx->sort_tmp, need));
        ctx->sort_tmp_bytes = need;
    }
    return MML_OK;
}

// rocPRIM temporary storage of the current stream lane (lanes run concurrently: no sharing)
int ensure_tmp_lane(mml_ctx* ctx, size_t need) {
    const int l = ctx->cur;
    if (need > ctx->seg_tmp_bytes[l]) {
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
        if (ctx->seg_tmp[l]) MML_HIP(hipFree(ctx->seg_tmp[l]));
        MML_HIP(hipMalloc(&ctx->seg_tmp[l], need));
        ctx->seg_tmp_bytes[l] = need;
    }
    return MML_OK;
}

// pcl::VoxelGrid over `m` device points; the filtered cloud goes to `out` (capacity cap), its size to *h_n
int voxel_filter_device(mml_ctx* ctx, const float4* pts, int m, float leaf, float4* out, int cap, int* h_n) {
    hipStream_t s = MML_STREAM(ctx);
    *h_n = 0;
    if (m == 0) return MML_OK;
    MML_REQUIRE(m <= ctx->MM && (size_t)m <= ctx->vox_cap, MML_ERR_CAPACITY, "cloud larger than max_map_points");
    int* d_bbox = ctx->d_misc;
    int* d_n = ctx->d_misc + 8;
    const int init[6] = {0x7f800000, 0x7f800000, 0x7f800000, (int)0xff800000 ^ 0x7fffffff, (int)0xff800000 ^ 0x7fffffff,
                         (int)0xff800000 ^ 0x7fffffff};  // +inf x3, key(-inf) x3
    MML_HIP(hipMemcpyAsync(d_bbox, init, sizeof(init), hipMemcpyHostToDevice, s));
    const int blocks = (m + 255) / 256;
    hipLaunchKernelGGL(k_minmax, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, s, pts, m, reinterpret_cast<float*>(d_bbox));
    hipLaunchKernelGGL(k_vox_keys, dim3(blocks), dim3(256), 0, s, pts, m, d_bbox, leaf, ctx->map_keys, ctx->map_vals);
    size_t need = 0;
    MML_HIP(rocprim::radix_sort_pairs(nullptr, need, ctx->map_keys, ctx->map_keys2, ctx->map_vals, ctx->map_vals2,
                                      (size_t)m, 0, 32, s));
    int rc = ensure_tmp(ctx, need);
    if (rc != MML_OK) return rc;
    MML_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp, need, ctx->map_keys, ctx->map_keys2, ctx->map_vals, ctx->map_vals2,
                                      (size_t)m, 0, 32, s));
    int* flag = ctx->vox_flag;
    int* pos = ctx->vox_flag + ctx->vox_cap + 1;
    hipLaunchKernelGGL(k_vox_heads, dim3(blocks), dim3(256), 0, s, ctx->map_keys2, m, flag);
    need = 0;
    MML_HIP(rocprim::exclusive_scan(nullptr, need, flag, pos, 0, (size_t)m, rocprim::plus<int>(), s));
    rc = ensure_tmp(ctx, need);
    if (rc != MML_OK) return rc;
    MML_HIP(rocprim::exclusive_scan(ctx->sort_tmp, need, flag, pos, 0, (size_t)m, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_vox_centroid, dim3(blocks), dim3(256), 0, s, pts, ctx->map_keys2, ctx->map_vals2, flag, pos, m, cap,
                       out, d_n);
    MML_HIP(hipMemcpyAsync(h_n, d_n, sizeof(int), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    MML_REQUIRE(*h_n <= cap, MML_ERR_CAPACITY, "filtered local map larger than max_map_points");
    return MML_OK;
}

}  // namespace

namespace {
__global__ void k_gather_list(const float4* pts, const unsigned* list, int n, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pts[list[i]];
}
int ensure_vox_scratch(mml_ctx* ctx, size_t pts) {
    if (pts <= ctx->vox_cap) return MML_OK;
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    if (ctx->ring_cat) MML_HIP(hipFree(ctx->ring_cat));
    if (ctx->vox_flag) MML_HIP(hipFree(ctx->vox_flag));
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->ring_cat), sizeof(float4) * pts));
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->vox_flag), sizeof(int) * 2 * (pts + 1)));
    ctx->vox_cap = pts;
    return MML_OK;
}
}  // namespace

// ---- a10 for labelled clouds beyond the LDS sort of k_voxel: the whole batch through ONE global sort --------------------
// Segment = (slot, kind).  The labelled points of every segment are gathered into one array, keyed by
// (segment << 32 | PCL voxel index in the segment's own grid), sorted by one stable rocPRIM radix sort (a voxel's points
// stay in input order = the order the oracle sums them in), and one lane per voxel writes the centroid to its slot's
// stack.  One 4-byte read-back (the number of labelled points in the batch) is the only host round trip.
namespace {
struct SegParams {
    int first, count, B, NT, MF, list_stride;
    float leaf_corner, leaf_surf;
    const int* fu_info;
    const float4* ln_pts;
    const int* ln_gidx;
    const unsigned* lists;
    const int* glists;  // the listed points' fused indices (label_append)
    int* seg_off;     // 2 * count + 1
    int* seg_bbox;    // 2 * count x 6 order-preserving int keys
    float4* cat;      // gathered points
    unsigned long long* keys;
    unsigned* vals;
    float4* ft0;
    float4* ft1;
    int* ft_n;
};

__global__ void k_seg_prefix(SegParams P, int* total) {
    __shared__ int s_run;
    const int nseg = 2 * P.count;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    // few thousand segments at most: a serial scan by one lane is a few microseconds
    if (threadIdx.x == 0) {
        int run = 0;
        for (int g = 0; g < nseg; ++g) {
            const int b = P.first + (g >> 1), kind = g & 1;
            P.seg_off[g] = run;
            run += P.fu_info[8 * b + 6 + kind];
        }
        P.seg_off[nseg] = run;
        *total = run;
    }
    for (int i = threadIdx.x; i < 6 * nseg; i += blockDim.x)
        P.seg_bbox[i] = (i % 6) < 3 ? 0x7f800000 : ((int)0xff800000 ^ 0x7fffffff);  // +inf, key(-inf)
}

// grid (blocks, segment): gather the labelled points and reduce the segment's bounding box
__global__ void k_seg_gather_bbox(SegParams P) {
    __shared__ float s[6][4];
    const int g = blockIdx.y, b = P.first + (g >> 1), kind = g & 1;
    const int n = P.fu_info[8 * b + 6 + kind], off = P.seg_off[g];
    const unsigned* list = P.lists + ((size_t)b * 2 + kind) * P.list_stride;
    const float4* px = P.ln_pts + (size_t)b * P.NT;
    const int* glist = P.glists + ((size_t)b * 2 + kind) * P.list_stride;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = px[list[i]];
        p.w = __int_as_float(glist[i]);  // the fused index rides along: it orders the points of a voxel
        P.cat[off + i] = p;
        mn[0] = fminf(mn[0], p.x);
        mn[1] = fminf(mn[1], p.y);
        mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x);
        mx[1] = fmaxf(mx[1], p.y);
        mx[2] = fmaxf(mx[2], p.z);
    }
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
        if (lane == 0) {
            s[c][wave] = mn[c];
            s[3 + c][wave] = mx[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float v = s[c][0];
        for (int w = 1; w < 4; ++w) v = (c < 3) ? fminf(v, s[c][w]) : fmaxf(v, s[c][w]);
        const int iv = __float_as_int(v);
        const int key = iv >= 0 ? iv : (iv ^ 0x7fffffff);
        if (c < 3)
            atomicMin(P.seg_bbox + 6 * g + c, key);
        else
            atomicMax(P.seg_bbox + 6 * g + c, key);
    }
}

__global__ void k_seg_keys(SegParams P) {
    const int g = blockIdx.y, kind = g & 1;
    const int off = P.seg_off[g], n = P.seg_off[g + 1] - off;
    const float leaf = kind == 0 ? P.leaf_corner : P.leaf_surf;
    const float inv = 1.0f / leaf;
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(unkey(P.seg_bbox[6 * g + c]) * inv));
        const int max_b = static_cast<int>(floor(unkey(P.seg_bbox[6 * g + 3 + c]) * inv));
        div_b[c] = max_b - min_b[c] + 1;
    }
    const float bmn[3] = {unkey(P.seg_bbox[6 * g]), unkey(P.seg_bbox[6 * g + 1]), unkey(P.seg_bbox[6 * g + 2])};
    const float bmx[3] = {unkey(P.seg_bbox[6 * g + 3]), unkey(P.seg_bbox[6 * g + 4]), unkey(P.seg_bbox[6 * g + 5])};
    const bool unfiltered = n > 0 && mml_voxel_grid_overflows(bmn, bmx, inv);  // (PCL returns the cloud as it came)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = P.cat[off + i];
        const int ijk0 = static_cast<int>(floor(p.x * inv) - static_cast<float>(min_b[0]));
        const int ijk1 = static_cast<int>(floor(p.y * inv) - static_cast<float>(min_b[1]));
        const int ijk2 = static_cast<int>(floor(p.z * inv) - static_cast<float>(min_b[2]));
        // (unfiltered: every point its own voxel, in the order of the fused cloud -- the gathered list is in storage order)
        const unsigned vox = unfiltered ? (unsigned)__float_as_int(p.w) : (unsigned)(ijk0 + ijk1 * div_b[0] + ijk2 * (div_b[0] * div_b[1]));
        // (segment 12 bits | voxel 32 bits | fused index 20 bits): one sort gives segment, voxel, reference order
        P.keys[off + i] = ((unsigned long long)g << 52) | ((unsigned long long)vox << 20) | (unsigned)__float_as_int(p.w);
        P.vals[off + i] = (unsigned)(off + i);
    }
}

__global__ void k_seg_heads(const unsigned long long* keys, int m, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flag[i] = (i == 0 || (keys[i] >> 20) != (keys[i - 1] >> 20)) ? 1 : 0;  // new (segment, voxel)
}

// one lane per voxel; the lane of a segment's first voxel also publishes the segment's voxel count
__global__ void k_seg_centroid(SegParams P, const unsigned long long* keys, const unsigned* vals, const int* flag, const int* pos, int m) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m || !flag[s]) return;
    const unsigned long long key = keys[s] >> 20;  // (segment, voxel)
    const int g = (int)(key >> 32), b = P.first + (g >> 1), kind = g & 1;
    const int off = P.seg_off[g], end = P.seg_off[g + 1];
    const int dst = pos[s] - pos[off];
    if (s == off) {
        const int nvox = (end < m ? pos[end] : pos[m - 1] + flag[m - 1]) - pos[off];
        P.ft_n[kind * P.B + b] = nvox > P.MF ? -1 : nvox;
    }
    if (dst >= P.MF) return;
    float sx = 0, sy = 0, sz = 0;
    int e = s;
    while (e < m && (keys[e] >> 20) == key) {
        const float4 p = P.cat[vals[e]];
        sx += p.x;
        sy += p.y;
        sz += p.z;
        ++e;
    }
    const float c = static_cast<float>(e - s);
    (kind == 0 ? P.ft0 : P.ft1)[(size_t)b * P.MF + dst] = make_float4(sx / c, sy / c, sz / c, 0.f);
}

__global__ void k_seg_empty(SegParams P) {  // segments without labelled points have an empty stack
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * P.count) return;
    if (P.seg_off[g + 1] == P.seg_off[g]) P.ft_n[(g & 1) * P.B + P.first + (g >> 1)] = 0;
}
}  // namespace

int mml_downsample_big(mml_ctx* ctx, int first, int count) {
    // (new stack sizes: statistics computed on demand must not pair the counts written below with an earlier association)
    for (int i = 0; i < count; ++i) ctx->stats_stale[first + i] = 1;
    hipStream_t s = MML_STREAM(ctx);
    const int nseg = 2 * count;
    // scratch for the worst case (every point of every slot labelled), allocated on first use
    const size_t cap = (size_t)ctx->B * ctx->NT;
    if (!ctx->seg_meta) {  // seg_meta is set last: it marks the scratch as complete
        const hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&ctx->seg_keys), sizeof(unsigned long long) * 2 * cap);
        const hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&ctx->seg_vals), sizeof(unsigned) * 2 * cap);
        const hipError_t e3 = hipMalloc(reinterpret_cast<void**>(&ctx->seg_cat), sizeof(float4) * cap);
        const hipError_t e4 = hipMalloc(reinterpret_cast<void**>(&ctx->seg_flag), sizeof(int) * 2 * (cap + 1));
        int* meta_buf = nullptr;
        const hipError_t e5 = hipMalloc(reinterpret_cast<void**>(&meta_buf), sizeof(int) * (8 * (size_t)ctx->B * 2 + 16) * mml_ctx::MAX_LANES);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
            // all or nothing: a later call must not find a half-allocated scratch and launch on null buffers
            (void)hipFree(ctx->seg_keys);
            (void)hipFree(ctx->seg_vals);
            (void)hipFree(ctx->seg_cat);
            (void)hipFree(ctx->seg_flag);
            (void)hipFree(meta_buf);
            ctx->seg_keys = nullptr;
            ctx->seg_vals = nullptr;
            ctx->seg_cat = nullptr;
            ctx->seg_flag = nullptr;
            (void)hipGetLastError();
            ctx->err = "mml_downsample_big: out of device memory for the global-sort scratch";
            return MML_ERR_HIP;
        }
        ctx->seg_meta = meta_buf;
    }
    // lanes work on disjoint slot ranges: each gets the part of the scratch that its slots' points can fill
    const size_t base = (size_t)first * ctx->NT;
    SegParams P;
    P.first = first;
    P.count = count;
    P.B = ctx->B;
    P.NT = ctx->NT;
    P.MF = ctx->MF;
    P.list_stride = ctx->VX_CAP;
    P.leaf_corner = ctx->cfg.leaf_corner;
    P.leaf_surf = ctx->cfg.leaf_surf;
    P.fu_info = ctx->fu_info;
    P.ln_pts = ctx->ln_pts;
    P.ln_gidx = ctx->ln_gidx;
    P.lists = reinterpret_cast<const unsigned*>(ctx->vx_keys);
    P.glists = ctx->vx_gidx;
    int* meta = ctx->seg_meta + (size_t)ctx->cur * (8 * (size_t)ctx->B * 2 + 16);
    P.seg_off = meta + 8;
    P.seg_bbox = meta + 8 + (2 * (size_t)ctx->B + 8);
    int* d_total = meta;
    P.cat = ctx->seg_cat + base;
    P.keys = ctx->seg_keys + base;
    P.vals = ctx->seg_vals + base;
    unsigned long long* keys2 = ctx->seg_keys + cap + base;
    unsigned* vals2 = ctx->seg_vals + cap + base;
    int* flag = ctx->seg_flag + base;
    int* pos = ctx->seg_flag + (cap + 1) + base;
    P.ft0 = ctx->ft_xyz[0];
    P.ft1 = ctx->ft_xyz[1];
    P.ft_n = ctx->ft_n;
    hipLaunchKernelGGL(k_seg_prefix, dim3(1), dim3(256), 0, s, P, d_total);
    int total = 0;
    MML_HIP(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, s));
    // (the gather does not depend on the read-back: it runs while the host waits for `total`)
    const int gblocks = (ctx->NT / 4 + 255) / 256 > 0 ? (ctx->NT / 4 + 255) / 256 : 1;
    hipLaunchKernelGGL(k_seg_gather_bbox, dim3(gblocks, nseg), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_seg_keys, dim3(gblocks, nseg), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_seg_empty, dim3((nseg + 255) / 256), dim3(256), 0, s, P);
    MML_HIP(hipStreamSynchronize(s));
    MML_REQUIRE(total >= 0 && (size_t)total <= (size_t)count * ctx->NT, MML_ERR_STATE, "labelled-point counts of the slots are inconsistent");
    if (total == 0) return MML_OK;
    int seg_bits = 1;
    while ((1 << seg_bits) < nseg) ++seg_bits;
    size_t need = 0;
    MML_REQUIRE(ctx->NT <= (1 << 20) && nseg <= 4096, MML_ERR_CAPACITY, "global-sort down-sampler: scans up to 2^20 points, 2048 slots per call");
    MML_HIP(rocprim::radix_sort_pairs(nullptr, need, P.keys, keys2, P.vals, vals2, (size_t)total, 0, 52 + seg_bits, s));
    int rc = ensure_tmp_lane(ctx, need);
    if (rc != MML_OK) return rc;
    MML_HIP(rocprim::radix_sort_pairs(ctx->seg_tmp[ctx->cur], need, P.keys, keys2, P.vals, vals2, (size_t)total, 0, 52 + seg_bits, s));
    const int blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_seg_heads, dim3(blocks), dim3(256), 0, s, keys2, total, flag);
    need = 0;
    MML_HIP(rocprim::exclusive_scan(nullptr, need, flag, pos, 0, (size_t)total, rocprim::plus<int>(), s));
    rc = ensure_tmp_lane(ctx, need);
    if (rc != MML_OK) return rc;
    MML_HIP(rocprim::exclusive_scan(ctx->seg_tmp[ctx->cur], need, flag, pos, 0, (size_t)total, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_seg_centroid, dim3(blocks), dim3(256), 0, s, P, keys2, vals2, flag, pos, total);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// Slots of [first, first + count) that k_voxel marked with ft_n = -1 because a labelled cloud exceeds its LDS sort:
// redone one by one through the global-sort filter.  Synchronises the stream (the marks are read back).  A slot whose
// overflow is a real capacity limit (more voxels than max_features) keeps its mark.
int mml_downsample_redo_overflow(mml_ctx* ctx, int first, int count, std::vector<int>* redone) {
    if (ctx->NT > (1 << 20)) return MML_OK;  // those contexts took the global-sort path to begin with
    hipStream_t s = MML_STREAM(ctx);
    // (pinned destinations: a device-to-host copy into pageable memory is a blocking staged copy, three of them cost more
    //  than the down-sampling kernel of a single scan)
    int* n0 = reinterpret_cast<int*>(mml_stage_alloc(ctx, 5 * (size_t)count + 2));
    int* n1 = n0 + count;
    int* info = n1 + count;
    MML_HIP(hipMemcpyAsync(n0, ctx->ft_n + first, sizeof(int) * count, hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(n1, ctx->ft_n + ctx->B + first, sizeof(int) * count, hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(info, ctx->fu_info + 8 * (size_t)first, sizeof(int) * 8 * (size_t)count, hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    for (int c = 0; c < count; ++c) {
        if (n0[c] >= 0 && n1[c] >= 0) continue;
        if (info[8 * c + 6] <= mml_voxel_cap_corner(ctx) && info[8 * c + 7] <= MML_VOXEL_LDS_CAP) continue;  // not a sort overflow
        int rc = mml_downsample_big(ctx, first + c, 1);
        if (rc != MML_OK) return rc;
        if (redone) redone->push_back(first + c);
    }
    return MML_OK;
}

int mml_map_upkeep_increment(mml_ctx* ctx, int slot, const double* T_wl, int* n_out) {
    hipStream_t s = MML_STREAM(ctx);
    constexpr int W = mml_ctx::LOCAL_WINDOW;
    const size_t ring_pts = (size_t)W * ctx->MF;
    if (!ctx->ring[0])
        for (int k = 0; k < 2; ++k) MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->ring[k]), sizeof(float4) * ring_pts));
    {
        int rc0 = ensure_vox_scratch(ctx, ring_pts);
        if (rc0 != MML_OK) return rc0;
    }
    int* n_feat = reinterpret_cast<int*>(mml_stage_alloc(ctx, 1));  // pinned
    MML_HIP(hipMemcpyAsync(&n_feat[0], ctx->ft_n + 0 * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(&n_feat[1], ctx->ft_n + 1 * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    Tf12 T;
    memcpy(T.m, T_wl, sizeof(double) * 12);
    const int Id = (int)(ctx->local_map_id % W);  // :1597
    for (int kind = 0; kind < 2; ++kind)  // both stacks are checked before any ring state changes
        MML_REQUIRE(n_feat[kind] >= 0 && n_feat[kind] <= ctx->MF, MML_ERR_STATE, "slot holds no down-sampled feature stack");
    for (int kind = 0; kind < 2; ++kind) {
        const int n = n_feat[kind];
        if (n)
            hipLaunchKernelGGL(k_to_world, dim3((n + 255) / 256), dim3(256), 0, s, ctx->ft_xyz[kind] + (size_t)slot * ctx->MF,
                               n, T, ctx->ring[kind] + (size_t)Id * ctx->MF);
        ctx->ring_n[kind][Id] = n;
        RingOffsets R;
        R.off[0] = 0;
        for (int i = 0; i < W; ++i) R.off[i + 1] = R.off[i] + ctx->ring_n[kind][i];
        const int total = R.off[W];
        if (total) hipLaunchKernelGGL(k_concat, dim3(8, W), dim3(256), 0, s, ctx->ring[kind], ctx->MF, R, ctx->ring_cat);
        int m = 0;
        int rc = voxel_filter_device(ctx, ctx->ring_cat, total, kind == 0 ? ctx->cfg.leaf_corner : ctx->cfg.leaf_surf,
                                     ctx->map_tmp + (size_t)kind * ctx->MM, ctx->MM, &m);
        if (rc != MML_OK) return rc;
        rc = mml_build_grid_device(ctx, kind, m);
        if (rc != MML_OK) return rc;
        ctx->local_map_n[kind] = m;
        if (n_out) n_out[kind] = m;
    }
    ctx->local_map_id++;  // :1642
    MML_HIP(hipGetLastError());
    return MML_OK;
}
