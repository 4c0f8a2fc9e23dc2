// map_global.hip -- SURVEY.md section 8(f) rank 2, global half: the MAP_MANAGER corner / surf cube stores on the device.
//   MAP_MANAGER::featureAssociateToMap (Map_Manager.cpp:91-117)  -> mml_cube_store_append   (features of a slot -> world)
//   MAP_MANAGER::MapIncrement (:125-281) + MapMove (:288-581)     -> mml_cube_store_increment
// The reference keeps 4851 PCL clouds per kind and rotates pointers between them; here a store is ONE array of points
// with a 16-bit cube tag each (the layout k_associate already consumes, a12):
//   * MapMove becomes a list of (axis, direction) layer shifts applied to the tags; a point whose layer leaves the
//     21 x 21 x 11 grid is dropped (the reference clears the cube that wraps around);
//   * new points are tagged with the shifted centre, per-cube counts come from a histogram, cubes that received points
//     and now hold > 300 are voxel-filtered (:225-233) -- all of them in one pass: per-cube bounding boxes through
//     atomics, 64-bit keys (cube, PCL voxel index), one stable radix sort, one lane per voxel;
//   * points of untouched cubes keep their relative order, filtered cubes are re-emitted in voxel order, new points of
//     small cubes follow the old ones: the order inside every cube equals the reference's, which is what fixes the
//     tie order of the exact 5-NN.
// What Estimate() matches against is the state at the START of the last MapIncrement (laserCloud*_for_match,
// laserCloudCen*_last, :136-149): the tagged kNN grids are therefore rebuilt from the store BEFORE it is updated.
// Compiled with -ffp-contract=off.
#include <math.h>

#include <cstring>
#include <string.h>
#include <vector>
#include <rocprim/rocprim.hpp>

#include "mml_internal.h"

namespace {

constexpr int NCUBE = 4851;
constexpr uint16_t DEAD = 0xFFFFu;

struct Tf12 {
    double m[12];
};
struct Shifts {
    int n;
    signed char axis[64], dir[64];
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

// layer shifts of MapMove applied to the cube index of every stored point
__global__ void k_apply_shifts(uint16_t* tags, int n, Shifts S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int t = tags[i];
    if (t == DEAD) return;
    int ijk[3] = {t % 21, (t / 21) % 21, t / 441};
    const int lim[3] = {21, 21, 11};
    bool dead = false;
    for (int s = 0; s < S.n && !dead; ++s) {
        const int a = S.axis[s];
        ijk[a] += S.dir[s];
        dead = ijk[a] < 0 || ijk[a] >= lim[a];
    }
    tags[i] = dead ? DEAD : (uint16_t)(ijk[0] + 21 * ijk[1] + 441 * ijk[2]);
}

// cube of a new world-frame point (:159-176), DEAD outside the grid
__global__ void k_tag_points(const float4* pts, int n, int cenW, int cenH, int cenD, uint16_t* tags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    int cubeI = int((p.x + 25.0) / 50.0) + cenD;
    int cubeJ = int((p.y + 25.0) / 50.0) + cenW;
    int cubeK = int((p.z + 25.0) / 50.0) + cenH;
    if (p.x + 25.0 < 0) cubeI--;
    if (p.y + 25.0 < 0) cubeJ--;
    if (p.z + 25.0 < 0) cubeK--;
    const bool ok = cubeI >= 0 && cubeI < 21 && cubeJ >= 0 && cubeJ < 21 && cubeK >= 0 && cubeK < 11;
    tags[i] = ok ? (uint16_t)(cubeI + 21 * cubeJ + 441 * cubeK) : DEAD;
}

__global__ void k_hist(const uint16_t* tags, int n, int* cnt, int* changed /* may be null */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = tags[i];
    if (t == DEAD) return;
    atomicAdd(&cnt[t], 1);
    if (changed) changed[t] = 1;
}

// combined index space u: [0, n0) the store, [n0, n0 + np) the pending points
__global__ void k_classify(const uint16_t* stag, int n0, const uint16_t* ptag, int np, const int* cnt, const int* changed,
                           int* keep, int* sel) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n0 + np) return;
    const int t = u < n0 ? stag[u] : ptag[u - n0];
    const bool valid = t != DEAD;
    const bool filt = valid && changed[t] && cnt[t] > 300;  // :225
    keep[u] = (valid && !filt) ? 1 : 0;
    sel[u] = filt ? 1 : 0;
}

__device__ __forceinline__ int fkey(float v) {
    const int iv = __float_as_int(v);
    return iv >= 0 ? iv : (iv ^ 0x7fffffff);
}
__device__ __forceinline__ float unkey(int key) { return __int_as_float(key >= 0 ? key : (key ^ 0x7fffffff)); }

__global__ void k_scatter(const float4* spts, const uint16_t* stag, int n0, const float4* ppts, const uint16_t* ptag, int np,
                          const int* keep, const int* keep_pos, const int* sel, const int* sel_pos, float4* out_pts,
                          uint16_t* out_tag, float4* sel_pts, uint16_t* sel_tag, int* bbox /* NCUBE x 6 keys */) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n0 + np) return;
    const float4 p = u < n0 ? spts[u] : ppts[u - n0];
    const uint16_t t = u < n0 ? stag[u] : ptag[u - n0];
    if (keep[u]) {
        out_pts[keep_pos[u]] = p;
        out_tag[keep_pos[u]] = t;
    } else if (sel[u]) {
        sel_pts[sel_pos[u]] = p;
        sel_tag[sel_pos[u]] = t;
        int* b = bbox + 6 * (int)t;  // getMinMax3D of the cube's cloud
        atomicMin(b + 0, fkey(p.x));
        atomicMin(b + 1, fkey(p.y));
        atomicMin(b + 2, fkey(p.z));
        atomicMax(b + 3, fkey(p.x));
        atomicMax(b + 4, fkey(p.y));
        atomicMax(b + 5, fkey(p.z));
    }
}

__global__ void k_init_bbox(int* bbox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NCUBE * 6) return;
    bbox[i] = (i % 6) < 3 ? 0x7f800000 : (int)(0xff800000u ^ 0x7fffffffu);
}

// PCL 1.8.1 voxel_grid.hpp applyFilter, per cube: key = cube << 40 | voxel index inside the cube's own grid
__global__ void k_cube_vox_keys(const float4* pts, const uint16_t* tags, int n, const int* bbox, float leaf,
                                unsigned long long* keys, unsigned* vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float inv = 1.0f / leaf;
    const int t = tags[i];
    const int* b = bbox + 6 * t;
    int min_b[3], div_b[3];
    for (int c = 0; c < 3; ++c) {
        min_b[c] = static_cast<int>(floor(unkey(b[c]) * inv));
        const int max_b = static_cast<int>(floor(unkey(b[3 + c]) * inv));
        div_b[c] = max_b - min_b[c] + 1;
    }
    const float4 p = pts[i];
    const int ijk0 = static_cast<int>(floor(p.x * inv) - static_cast<float>(min_b[0]));
    const int ijk1 = static_cast<int>(floor(p.y * inv) - static_cast<float>(min_b[1]));
    const int ijk2 = static_cast<int>(floor(p.z * inv) - static_cast<float>(min_b[2]));
    const unsigned idx = (unsigned)(ijk0 + ijk1 * div_b[0] + ijk2 * (div_b[0] * div_b[1]));
    keys[i] = ((unsigned long long)t << 40) | idx;
    vals[i] = (unsigned)i;
}

__global__ void k_heads64(const unsigned long long* keys, int n, int* flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void k_cube_centroid(const float4* pts, const unsigned long long* keys, const unsigned* vals, const int* flag,
                                const int* pos, int n, int base, int cap, float4* out_pts, uint16_t* out_tag, int* n_heads) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (s == n - 1) *n_heads = pos[s] + flag[s];
    if (!flag[s]) return;
    const int dst = base + pos[s];
    if (dst >= cap) return;
    const unsigned long long key = keys[s];
    float sx = 0, sy = 0, sz = 0;
    int e = s;
    while (e < n && keys[e] == key) {
        const float4 p = pts[vals[e]];
        sx += p.x;
        sy += p.y;
        sz += p.z;
        ++e;
    }
    const float c = static_cast<float>(e - s);
    out_pts[dst] = make_float4(sx / c, sy / c, sz / c, 0.f);
    out_tag[dst] = (uint16_t)(key >> 40);
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

int ensure_store(mml_ctx* ctx) {
    if (ctx->gs_pts[0]) return MML_OK;
    const size_t MM = (size_t)ctx->MM;
    ctx->gp_cap = 16 * ctx->MF;
    const size_t U = (MM + (size_t)ctx->gp_cap + 3) & ~size_t(3);  // keeps the carved-out float4 array 16-byte aligned
    for (int k = 0; k < 2; ++k) {
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_pts[k]), sizeof(float4) * MM));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_tag[k]), sizeof(uint16_t) * MM));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_pts2[k]), sizeof(float4) * MM));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_tag2[k]), sizeof(uint16_t) * MM));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gp_pts[k]), sizeof(float4) * (size_t)ctx->gp_cap));
    }
    // cnt | changed | bbox (6 per cube) | keep | keep_pos | sel | sel_pos | n_heads
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_work), sizeof(int) * (8 * (size_t)NCUBE + 4 * (U + 1) + 8)));
    // 64-bit keys (in / out), sort values (in / out), selected points + tags + pending tags
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gs_keys),
                      sizeof(unsigned long long) * 2 * U + sizeof(unsigned) * 2 * U + sizeof(float4) * U + sizeof(uint16_t) * 2 * U));
    return MML_OK;
}

}  // namespace

int mml_cube_store_reset(mml_ctx* ctx) {
    ctx->gs_n[0] = ctx->gs_n[1] = 0;
    ctx->gp_n[0] = ctx->gp_n[1] = 0;
    ctx->gs_cen[0] = 10;
    ctx->gs_cen[1] = 5;
    ctx->gs_cen[2] = 10;
    return MML_OK;
}

int mml_cube_store_append(mml_ctx* ctx, int slot, const double* T_wl) {
    hipStream_t s = MML_STREAM(ctx);
    int rc = ensure_store(ctx);
    if (rc != MML_OK) return rc;
    int n_feat[2];
    MML_HIP(hipMemcpyAsync(&n_feat[0], ctx->ft_n + 0 * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(&n_feat[1], ctx->ft_n + 1 * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    Tf12 T;
    memcpy(T.m, T_wl, sizeof(double) * 12);
    for (int kind = 0; kind < 2; ++kind) {
        const int n = n_feat[kind];
        MML_REQUIRE(n >= 0 && n <= ctx->MF, MML_ERR_STATE, "slot holds no down-sampled feature stack");
        MML_REQUIRE(ctx->gp_n[kind] + n <= ctx->gp_cap, MML_ERR_CAPACITY, "too many pending map points (16 x max_features)");
        if (n)
            hipLaunchKernelGGL(k_to_world, dim3((n + 255) / 256), dim3(256), 0, s, ctx->ft_xyz[kind] + (size_t)slot * ctx->MF, n, T,
                               ctx->gp_pts[kind] + ctx->gp_n[kind]);
        ctx->gp_n[kind] += n;
    }
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_cube_store_increment(mml_ctx* ctx, const double* T_wl, int* n_out) {
    hipStream_t s = MML_STREAM(ctx);
    int rc = ensure_store(ctx);
    if (rc != MML_OK) return rc;
    const size_t U = ((size_t)ctx->MM + (size_t)ctx->gp_cap + 3) & ~size_t(3);
    int* cnt = ctx->gs_work;
    int* changed = cnt + NCUBE;
    int* bbox = changed + NCUBE;
    int* keep = bbox + 6 * NCUBE;
    int* keep_pos = keep + (U + 1);
    int* sel = keep_pos + (U + 1);
    int* sel_pos = sel + (U + 1);
    int* d_heads = sel_pos + (U + 1);
    unsigned long long* keys = ctx->gs_keys;
    unsigned long long* keys2 = keys + U;
    unsigned* vals = reinterpret_cast<unsigned*>(keys2 + U);
    unsigned* vals2 = vals + U;
    float4* sel_pts = reinterpret_cast<float4*>(vals2 + U);
    uint16_t* sel_tag = reinterpret_cast<uint16_t*>(sel_pts + U);
    uint16_t* ptag = sel_tag + U;

    // (1) what Estimate() matches against from now on: the store as it is, with its current centre (:136-149)
    for (int kind = 0; kind < 2; ++kind) {
        const int n0 = ctx->gs_n[kind];
        MML_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * NCUBE, s));
        if (n0) hipLaunchKernelGGL(k_hist, dim3((n0 + 255) / 256), dim3(256), 0, s, ctx->gs_tag[kind], n0, cnt, (int*)nullptr);
        rc = mml_build_global_grid_device(ctx, kind, ctx->gs_pts[kind], ctx->gs_tag[kind], cnt, n0, ctx->gs_cen);
        if (rc != MML_OK) return rc;
    }
    // (2) MapMove (:288-581): the layer shifts that keep the sensor's cube 8 cubes away from every face
    Shifts S;
    S.n = 0;
    {
        int* cen = ctx->gs_cen;
        const double t[3] = {T_wl[3], T_wl[7], T_wl[11]};
        int c[3] = {int((t[0] + 25.0) / 50.0) + cen[2], int((t[1] + 25.0) / 50.0) + cen[0], int((t[2] + 25.0) / 50.0) + cen[1]};
        for (int a = 0; a < 3; ++a)
            if (t[a] + 25.0 < 0) c[a]--;
        const int lim[3] = {21, 21, 11};
        int* cen_of_axis[3] = {&cen[2], &cen[0], &cen[1]};  // I <-> Depth, J <-> Width, K <-> Height
        for (int a = 0; a < 3; ++a) {
            while (c[a] < 8) {
                MML_REQUIRE(S.n < 64, MML_ERR_INVALID, "pose far outside the cube grid");
                S.axis[S.n] = (signed char)a;
                S.dir[S.n++] = 1;
                c[a]++;
                (*cen_of_axis[a])++;
            }
            while (c[a] >= lim[a] - 8) {
                MML_REQUIRE(S.n < 64, MML_ERR_INVALID, "pose far outside the cube grid");
                S.axis[S.n] = (signed char)a;
                S.dir[S.n++] = -1;
                c[a]--;
                (*cen_of_axis[a])--;
            }
        }
    }
    // (3) distribute, filter the cubes that grew beyond 300 points, compact
    for (int kind = 0; kind < 2; ++kind) {
        const int n0 = ctx->gs_n[kind], np = ctx->gp_n[kind], nu = n0 + np;
        if (n0 && S.n) hipLaunchKernelGGL(k_apply_shifts, dim3((n0 + 255) / 256), dim3(256), 0, s, ctx->gs_tag[kind], n0, S);
        if (np)
            hipLaunchKernelGGL(k_tag_points, dim3((np + 255) / 256), dim3(256), 0, s, ctx->gp_pts[kind], np, ctx->gs_cen[0],
                               ctx->gs_cen[1], ctx->gs_cen[2], ptag);
        MML_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * 2 * NCUBE, s));  // cnt + changed
        if (n0) hipLaunchKernelGGL(k_hist, dim3((n0 + 255) / 256), dim3(256), 0, s, ctx->gs_tag[kind], n0, cnt, (int*)nullptr);
        if (np) hipLaunchKernelGGL(k_hist, dim3((np + 255) / 256), dim3(256), 0, s, ptag, np, cnt, changed);
        int nk = 0, nsel = 0, nheads = 0;
        if (nu) {
            hipLaunchKernelGGL(k_classify, dim3((nu + 255) / 256), dim3(256), 0, s, ctx->gs_tag[kind], n0, ptag, np, cnt, changed,
                               keep, sel);
            size_t need = 0;
            MML_HIP(rocprim::exclusive_scan(nullptr, need, keep, keep_pos, 0, (size_t)nu, rocprim::plus<int>(), s));
            rc = ensure_tmp(ctx, need);
            if (rc != MML_OK) return rc;
            MML_HIP(rocprim::exclusive_scan(ctx->sort_tmp, need, keep, keep_pos, 0, (size_t)nu, rocprim::plus<int>(), s));
            MML_HIP(rocprim::exclusive_scan(ctx->sort_tmp, need, sel, sel_pos, 0, (size_t)nu, rocprim::plus<int>(), s));
            int last[4];
            MML_HIP(hipMemcpyAsync(&last[0], keep + nu - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            MML_HIP(hipMemcpyAsync(&last[1], keep_pos + nu - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            MML_HIP(hipMemcpyAsync(&last[2], sel + nu - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            MML_HIP(hipMemcpyAsync(&last[3], sel_pos + nu - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            MML_HIP(hipStreamSynchronize(s));
            nk = last[0] + last[1];
            nsel = last[2] + last[3];
            MML_REQUIRE(nk + nsel <= ctx->MM, MML_ERR_CAPACITY, "cube store larger than max_map_points");
            hipLaunchKernelGGL(k_init_bbox, dim3((NCUBE * 6 + 255) / 256), dim3(256), 0, s, bbox);
            hipLaunchKernelGGL(k_scatter, dim3((nu + 255) / 256), dim3(256), 0, s, ctx->gs_pts[kind], ctx->gs_tag[kind], n0,
                               ctx->gp_pts[kind], ptag, np, keep, keep_pos, sel, sel_pos, ctx->gs_pts2[kind], ctx->gs_tag2[kind],
                               sel_pts, sel_tag, bbox);
            if (nsel) {
                const int blocks = (nsel + 255) / 256;
                const float leaf = 0.4f;  // MAP_MANAGER's own filters, Map_Manager.cpp:58-60 (corner and surf alike)
                hipLaunchKernelGGL(k_cube_vox_keys, dim3(blocks), dim3(256), 0, s, sel_pts, sel_tag, nsel, bbox, leaf, keys, vals);
                need = 0;
                MML_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, keys2, vals, vals2, (size_t)nsel, 0, 56, s));
                rc = ensure_tmp(ctx, need);
                if (rc != MML_OK) return rc;
                MML_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp, need, keys, keys2, vals, vals2, (size_t)nsel, 0, 56, s));
                hipLaunchKernelGGL(k_heads64, dim3(blocks), dim3(256), 0, s, keys2, nsel, keep);  // keep / keep_pos are free again
                need = 0;
                MML_HIP(rocprim::exclusive_scan(nullptr, need, keep, keep_pos, 0, (size_t)nsel, rocprim::plus<int>(), s));
                rc = ensure_tmp(ctx, need);
                if (rc != MML_OK) return rc;
                MML_HIP(rocprim::exclusive_scan(ctx->sort_tmp, need, keep, keep_pos, 0, (size_t)nsel, rocprim::plus<int>(), s));
                hipLaunchKernelGGL(k_cube_centroid, dim3(blocks), dim3(256), 0, s, sel_pts, keys2, vals2, keep, keep_pos, nsel, nk,
                                   ctx->MM, ctx->gs_pts2[kind], ctx->gs_tag2[kind], d_heads);
                MML_HIP(hipMemcpyAsync(&nheads, d_heads, sizeof(int), hipMemcpyDeviceToHost, s));
                MML_HIP(hipStreamSynchronize(s));
            }
        }
        MML_HIP(hipGetLastError());
        std::swap(ctx->gs_pts[kind], ctx->gs_pts2[kind]);
        std::swap(ctx->gs_tag[kind], ctx->gs_tag2[kind]);
        ctx->gs_n[kind] = nk + nheads;
        ctx->gp_n[kind] = 0;
        if (n_out) n_out[kind] = ctx->gs_n[kind];
    }
    return MML_OK;
}

// the LIVE store (tests / visualisation): points, cube index of every point, centre
int mml_cube_store_download(mml_ctx* ctx, int kind, float* xyz, int* cube, int capacity, int* n, int* cen) {
    hipStream_t s = MML_STREAM(ctx);
    const int m = ctx->gs_pts[kind] ? ctx->gs_n[kind] : 0;
    *n = m;
    if (cen)
        for (int k = 0; k < 3; ++k) cen[k] = ctx->gs_cen[k];
    if (m == 0 || (!xyz && !cube)) return MML_OK;
    MML_REQUIRE(capacity >= m, MML_ERR_CAPACITY, "capacity below the store size");
    std::vector<float4> p((size_t)m);
    std::vector<uint16_t> t((size_t)m);
    MML_HIP(hipMemcpyAsync(p.data(), ctx->gs_pts[kind], sizeof(float4) * (size_t)m, hipMemcpyDeviceToHost, s));
    MML_HIP(hipMemcpyAsync(t.data(), ctx->gs_tag[kind], sizeof(uint16_t) * (size_t)m, hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < m; ++i) {
        if (xyz) {
            xyz[3 * i] = p[i].x;
            xyz[3 * i + 1] = p[i].y;
            xyz[3 * i + 2] = p[i].z;
        }
        if (cube) cube[i] = t[i];
    }
    return MML_OK;
}
