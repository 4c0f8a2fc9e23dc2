// map_assoc.hip -- rows a11..a16 of SURVEY.md section 8.
//   map side  : radix-sorted uniform grid over laserCloud{Corner,Surf}FromLocal -- replaces the
//               pcl::KdTreeFLANN::setInputCloud rebuild at mm-loam/src/lio/Estimator.cpp:1159-1167.
//   a13       : exact 5-NN (FLANN L2_Simple<float> semantics: d2 = ((dx*dx + dy*dy) + dz*dz) in float,
//               ascending, ties by lower index), one query per lane, ring-expanding cell search with an
//               exactness bound (all points within r*cell + margin are visited before stopping).
//   a11/a14/a15/a16 : pointAssociateToMap (Map_Manager.cpp:75-89), line fit (Estimator.cpp:283-361),
//               plane fit (:702-767), FeatureLine / FeaturePlanVec::ComputeError (Estimator.h:71-83,118-121),
//               fused behind the kNN in the same lane so the 5 neighbours never leave registers.
// Compiled with -ffp-contract=off.
#include <math.h>
#include <stdlib.h>

#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "mml_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// grid build
__global__ void k_bbox(const float4* pts, int m, float* out /*6: min xyz, max xyz*/) {
    __shared__ float s[6][4];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
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
        // float atomics via int reinterpretation (monotone for the sign handled below)
        if (c < 3) {
            // atomic min on float
            int* a = reinterpret_cast<int*>(out + c);
            int old = *a;
            while (v < __int_as_float(old)) {
                int assumed = old;
                old = atomicCAS(a, assumed, __float_as_int(v));
                if (old == assumed) break;
            }
        } else {
            int* a = reinterpret_cast<int*>(out + c);
            int old = *a;
            while (v > __int_as_float(old)) {
                int assumed = old;
                old = atomicCAS(a, assumed, __float_as_int(v));
                if (old == assumed) break;
            }
        }
    }
}

struct GridDev {
    float ox, oy, oz, inv_cell, cell;
    int dx, dy, dz, ncell;
};

__device__ __forceinline__ int cell_coord(float v, float o, float inv, int dim) {
    int c = (int)floorf((v - o) * inv);
    return c < 0 ? 0 : (c >= dim ? dim - 1 : c);
}

__global__ void k_cell_keys(const float4* pts, int m, GridDev g, unsigned* keys, unsigned* vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float4 p = pts[i];
    int cx = cell_coord(p.x, g.ox, g.inv_cell, g.dx);
    int cy = cell_coord(p.y, g.oy, g.inv_cell, g.dy);
    int cz = cell_coord(p.z, g.oz, g.inv_cell, g.dz);
    keys[i] = (unsigned)(cx + g.dx * (cy + g.dy * cz));
    vals[i] = (unsigned)i;
}

__global__ void k_gather_sorted(const float4* pts, const unsigned* vals, int m, float4* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned src = vals[i];
    float4 p = pts[src];
    p.w = __uint_as_float(src);
    out[i] = p;
}

__global__ void k_gather_tags(const uint16_t* tag_orig, const unsigned* vals, int m, uint16_t* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = tag_orig[vals[i]];
}

// number of occupied cells = number of key changes in the sorted key array
__global__ void k_count_occupied(const unsigned* keys, int m, int* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool head = i < m && (i == 0 || keys[i] != keys[i - 1]);
    unsigned long long b = __ballot(head);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, __popcll(b));
}

// cell_start[c] = first sorted position whose key >= c (lower bound); cell_start[ncell] = m
__global__ void k_cell_start(const unsigned* keys, int m, int ncell, int* cell_start) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncell) return;
    int lo = 0, hi = m;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned)c)
            lo = mid + 1;
        else
            hi = mid;
    }
    cell_start[c] = lo;
}

// ---------------------------------------------------------------------------------------------------
// exact 5-NN
// The list is five 64-bit keys, (bits of d2) << 32 | index: d2 is a non-negative float, so its bit pattern orders like
// its value and one unsigned compare is the reference's order "(d, index) ascending" (ties by lower index).  A NaN d2
// has a pattern above +inf and never enters, as with the float compare.
struct Knn5 {
    unsigned long long key[5];
};
constexpr unsigned long long KNN_EMPTY = (0x7f800000ull << 32) | 0x7fffffffull;  // (INFINITY, INT_MAX)
__device__ __forceinline__ unsigned long long knn_key(float d, int id) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)id;
}
__device__ __forceinline__ float knn_d(const Knn5& k, int j) { return __uint_as_float((unsigned)(k.key[j] >> 32)); }
__device__ __forceinline__ int knn_id(const Knn5& k, int j) { return (int)(unsigned)k.key[j]; }
__device__ __forceinline__ void knn_init(Knn5& k) {
#pragma unroll
    for (int i = 0; i < 5; ++i) k.key[i] = KNN_EMPTY;
}
// Sorted insert by rank: lt_s = (x < key[s]) is monotone in s for a sorted list, so the new list is
//   key[s] = lt_s ? (lt_{s-1} ? key[s-1] : x) : key[s]
// -- five compares and two selects per word, 23 vector instructions and no dependent chain.  A wavefront runs the insert as
// soon as ONE of its lanes has a candidate below its fifth key, which is the case for nearly every candidate of the first rings
// (a lane takes ~5 (1 + ln(n / 5)) of n candidates), so the insert is most of what a candidate costs; the sinking form
// (compare, two selects, compare, two selects per stage) compiled to 38.
__device__ __forceinline__ unsigned long long sel64(bool c, unsigned long long a, unsigned long long b) {
    // (two v_cndmask; written on the halves and marked unpredictable so that no pass turns a select of a select into branches)
    const unsigned lo = __builtin_unpredictable(c) ? (unsigned)a : (unsigned)b;
    const unsigned hi = __builtin_unpredictable(c) ? (unsigned)(a >> 32) : (unsigned)(b >> 32);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void knn_insert_key(Knn5& k, unsigned long long x) {
    const bool lt0 = x < k.key[0], lt1 = x < k.key[1], lt2 = x < k.key[2], lt3 = x < k.key[3], lt4 = x < k.key[4];
    const unsigned long long c1 = sel64(lt0, k.key[0], x), c2 = sel64(lt1, k.key[1], x), c3 = sel64(lt2, k.key[2], x),
                             c4 = sel64(lt3, k.key[3], x);
    k.key[4] = sel64(lt4, c4, k.key[4]);
    k.key[3] = sel64(lt3, c3, k.key[3]);
    k.key[2] = sel64(lt2, c2, k.key[2]);
    k.key[1] = sel64(lt1, c1, k.key[1]);
    k.key[0] = sel64(lt0, x, k.key[0]);
}
__device__ __forceinline__ void knn_insert(Knn5& k, float dd, int ii) {
    const unsigned long long x = knn_key(dd, ii);
    if (!(x < k.key[4])) return;
    knn_insert_key(k, x);
}

// tags / mytag: when the grid carries cube tags (global map, a12) only points of the query's cube take part: the
// reference searches the kd-tree of ONE cube (Estimator.cpp:199,630).  Every point inside the visited radius is still
// looked at, so the exactness bound of the ring search is unchanged.
__device__ __forceinline__ void scan_range(const float4* __restrict__ pts, const uint16_t* __restrict__ tags, int mytag,
                                           int s, int e, float qx, float qy, float qz, Knn5& k) {
    // four candidates per round, their loads issued together (one load per round leaves its whole latency exposed)
    for (int i0 = s; i0 < e; i0 += 4) {
        float4 p[4];
        int tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = (unsigned)min(i0 + u, e - 1);  // (unsigned: a 32-bit offset from the scalar base, no 64-bit address arithmetic)
            p[u] = pts[i];
            tg[u] = tags ? (int)tags[i] : mytag;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
            float r = 0;
            r += dx * dx;
            r += dy * dy;
            r += dz * dz;
            // a candidate that does not take part carries a key that never enters
            const unsigned long long x = (i0 + u >= e || tg[u] != mytag) ? ~0ull : knn_key(r, (int)__float_as_uint(p[u].w));
            if (x < k.key[4]) knn_insert_key(k, x);
        }
    }
}

// Ring-expanding search.  After ring r every map point with Chebyshev cell distance <= r from the query's home
// cell has been visited, which includes every point within Euclidean distance
// rho_r = (r + inset) * cell - margin  (inset = distance from the query to the nearest face of its home cell).
// A query is finished when its 5th best squared distance is below rho_r^2 (exact 5-NN), or when rho_r^2 >= max_d2
// (the caller rejects anything with d5 >= max_d2, Estimator.cpp:285,705).
//
// Two-level schedule: every lane walks rings 0 and 1 of its own query (27 cells, the common case for a
// voxel-filtered map); queries that are still open are then finished one at a time by the WHOLE wavefront, the 64
// lanes splitting the rows of each further shell and merging their private top-5 lists with shuffles.  That bounds
// the cost of the rare far query (hundreds of mostly empty cells) by ~1/64 of a private walk.
struct KnnQuery {
    float qx, qy, qz, inset;
    float fx, fy, fz;  // the query in cell units
    int hx, hy, hz;
};

__device__ __forceinline__ KnnQuery knn_query(const MmlGrid& g, float qx, float qy, float qz) {
    KnnQuery q;
    q.qx = qx;
    q.qy = qy;
    q.qz = qz;
    const float fx = (qx - g.origin[0]) * g.inv_cell, fy = (qy - g.origin[1]) * g.inv_cell,
                fz = (qz - g.origin[2]) * g.inv_cell;
    q.fx = fx;
    q.fy = fy;
    q.fz = fz;
    q.hx = (int)floorf(fx);
    q.hy = (int)floorf(fy);
    q.hz = (int)floorf(fz);
    float inset = fminf(fminf(fminf(fx - q.hx, q.hx + 1 - fx), fminf(fy - q.hy, q.hy + 1 - fy)),
                        fminf(fz - q.hz, q.hz + 1 - fz));
    q.inset = (inset > 0.f) ? inset : 0.f;
    return q;
}

// true when the search may stop after shell r
__device__ __forceinline__ bool knn_done(const MmlGrid& g, float inset, int r, float d5, float max_d2) {
    float rho = ((float)r + inset) * g.cell;
    rho = rho - 1e-3f * g.cell;  // margin dominating the float rounding of d2 and of the cell mapping
    if (!(rho > 0.f)) return false;
    const float rho2 = rho * rho;
    return d5 < rho2 || rho2 >= max_d2;
}

// one row (fixed y,z) of shell r: the whole x-span on a face, the two end cells otherwise.
// `bound` is any upper bound of the query's final 5th squared distance (INFINITY when none is known).  Cells that
// cannot hold a point closer than that are skipped: a point stored in cell (cx, y, z) lies, per axis, within 2e-3 cells
// of the cell's slab (float rounding of the cell mapping at build and query time, the same allowance knn_done makes),
// so its distance to the query is at least cell * (|gap vector| - 4e-3); a skipped point has d2 > bound * (1 + 1e-5) in
// exact arithmetic and therefore a float d2 > bound: it could not have entered the list.
__device__ __forceinline__ void scan_shell_row(const MmlGrid& g, const KnnQuery& q, int r, int y, int z, Knn5& k,
                                               int mytag = -1, float bound = INFINITY) {
    const int DX = g.dim[0], DY = g.dim[1], DZ = g.dim[2];
    if (y < 0 || y >= DY || z < 0 || z >= DZ) return;
    int xlo = -1, xhi = DX;  // no bound yet (a far query still looking for its first five points): nothing to work out
    if (bound < INFINITY) {
        const float gy = y > q.hy ? (float)y - q.fy : (y < q.hy ? q.fy - (float)(y + 1) : 0.f);
        const float gz = z > q.hz ? (float)z - q.fz : (z < q.hz ? q.fz - (float)(z + 1) : 0.f);
        const float reach = sqrtf(bound * 1.00001f) * g.inv_cell + 4e-3f;  // cells
        const float w2 = reach * reach - (gy * gy + gz * gz);
        if (w2 < 0.f) return;
        const float w = sqrtf(w2);
        // cells of this row whose slab comes within w cells of the query along x
        xlo = (int)floorf(fmaxf(q.fx - w, -1.f));
        xhi = (int)floorf(fminf(q.fx + w, (float)DX));
    }
    const int x0 = q.hx - r, x1 = q.hx + r;
    const bool face = (z == q.hz - r || z == q.hz + r || y == q.hy - r || y == q.hy + r);
    const int rowbase = DX * (y + DY * z);
    if (face) {
        const int xa = max(max(x0, 0), xlo), xb = min(min(x1, DX - 1), xhi);
        if (xa <= xb)
            scan_range(g.pts, g.tags, mytag, g.cell_start[rowbase + xa], g.cell_start[rowbase + xb + 1], q.qx, q.qy, q.qz, k);
    } else {
        if (x0 >= 0 && x0 < DX && x0 >= xlo)
            scan_range(g.pts, g.tags, mytag, g.cell_start[rowbase + x0], g.cell_start[rowbase + x0 + 1], q.qx, q.qy, q.qz, k);
        if (x1 >= 0 && x1 < DX && x1 != x0 && x1 <= xhi)
            scan_range(g.pts, g.tags, mytag, g.cell_start[rowbase + x1], g.cell_start[rowbase + x1 + 1], q.qx, q.qy, q.qz, k);
    }
}

// ring 1 rows, nearest first, so that the bound has tightened before the edge and corner rows are reached
__device__ __constant__ const signed char kRing1Row[9][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-1, -1}, {1, -1}, {-1, 1}, {1, 1}};

// rings 0 and 1 of one query, private to the lane; true when the search may stop there
__device__ __forceinline__ bool scan_rings01(const MmlGrid& g, const KnnQuery& q, int rmax, float max_d2, Knn5& k, int mytag) {
    scan_shell_row(g, q, 0, q.hy, q.hz, k, mytag);
    if (knn_done(g, q.inset, 0, knn_d(k, 4), max_d2)) return true;
    if (rmax < 1) return false;
#pragma unroll 1
    for (int t = 0; t < 9; ++t) scan_shell_row(g, q, 1, q.hy + kRing1Row[t][0], q.hz + kRing1Row[t][1], k, mytag, knn_d(k, 4));
    return knn_done(g, q.inset, 1, knn_d(k, 4), max_d2);
}

__device__ __forceinline__ unsigned long long knn_head(const Knn5& k, int head) {
    unsigned long long key = KNN_EMPTY;
#pragma unroll
    for (int s = 0; s < 5; ++s)
        if (head == s) key = k.key[s];
    return key;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
    const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
    return ((unsigned long long)hi << 32) | lo;
}
// merge of the private sorted lists (disjoint point sets) of the lanes whose ids differ in the bits below `span` into
// their common top-5, same in all of them
template <int SPAN>
__device__ __forceinline__ void lanes_merge5(const Knn5& local, Knn5& out) {
    int head = 0;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const unsigned long long mine = knn_head(local, head);
        unsigned long long m = mine;
#pragma unroll
        for (int o = SPAN / 2; o > 0; o >>= 1) {
            const unsigned long long other = shfl_xor_u64(m, o);
            m = other < m ? other : m;
        }
        out.key[r] = m;
        if (mine == m && (unsigned)(m >> 32) < 0x7f800000u) head++;
    }
}

// Must be called by ALL 64 lanes of the wavefront (inactive queries pass valid = false and only help).
__device__ void knn5_search(const MmlGrid& g, bool valid, float qx, float qy, float qz, float max_d2, Knn5& k) {
    knn_init(k);
    const KnnQuery q = knn_query(g, qx, qy, qz);
    // no shell beyond the grid's far side holds a cell (keeps an unbounded search, max_d2 = inf, finite)
    const int rgrid = max(max(max(q.hx, g.dim[0] - 1 - q.hx), max(q.hy, g.dim[1] - 1 - q.hy)), max(q.hz, g.dim[2] - 1 - q.hz));
    const float rf = ceilf(sqrtf(max_d2) * g.inv_cell) + 1.f;
    const int rmax = rf < (float)rgrid ? (int)rf : max(rgrid, 0);
    bool pending = false;
    if (valid) {
        pending = !scan_rings01(g, q, rmax, max_d2, k, -1);
        if (rmax < 2) pending = false;
    }
    unsigned long long todo = __ballot(pending);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        KnnQuery s;
        s.qx = __shfl(q.qx, src);
        s.qy = __shfl(q.qy, src);
        s.qz = __shfl(q.qz, src);
        s.inset = __shfl(q.inset, src);
        s.fx = __shfl(q.fx, src);
        s.fy = __shfl(q.fy, src);
        s.fz = __shfl(q.fz, src);
        s.hx = __shfl(q.hx, src);
        s.hy = __shfl(q.hy, src);
        s.hz = __shfl(q.hz, src);
        Knn5 loc;
        if (lane == src)
            loc = k;
        else
            knn_init(loc);
        Knn5 best;
        knn_init(best);
        const int rmax_s = __shfl(rmax, src);  // (the bound of the query being served, not of the serving lane's own)
        // a query outside the grid: the shells before the grid's near side hold no cell
        const int rnear = max(max(max(-s.hx, s.hx - (g.dim[0] - 1)), max(-s.hy, s.hy - (g.dim[1] - 1))),
                              max(-s.hz, s.hz - (g.dim[2] - 1)));
        for (int r = max(2, rnear); r <= rmax_s; ++r) {
            const float bnd = fminf(knn_d(best, 4), __shfl(knn_d(k, 4), src));  // both bound the final 5th distance from above
            // the rows of the shell that lie inside the grid
            const int ylo = max(s.hy - r, 0), yhi = min(s.hy + r, g.dim[1] - 1), zlo = max(s.hz - r, 0), zhi = min(s.hz + r, g.dim[2] - 1);
            const int wy = yhi - ylo + 1, wz = zhi - zlo + 1;
            if (wy > 0 && wz > 0)
                for (int t = lane; t < wy * wz; t += 64)
                    scan_shell_row(g, s, r, ylo + (t % wy), zlo + (t / wy), loc, -1, fminf(bnd, knn_d(loc, 4)));
            lanes_merge5<64>(loc, best);
            if (knn_done(g, s.inset, r, knn_d(best, 4), max_d2)) break;
        }
        if (lane == src) k = best;
    }
}

__global__ __launch_bounds__(256) void k_knn5(MmlGrid g, const float* q, int nq, float max_d2, int* idx, float* d2) {
    int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < nq;
    Knn5 k;
    knn5_search(g, valid, valid ? q[3 * i] : 0.f, valid ? q[3 * i + 1] : 0.f, valid ? q[3 * i + 2] : 0.f, max_d2, k);
    if (!valid) return;
    for (int j = 0; j < 5; ++j) {
        bool ok = knn_d(k, j) < max_d2 && knn_d(k, 4) < max_d2;
        idx[5 * i + j] = ok ? knn_id(k, j) : -1;
        d2[5 * i + j] = ok ? knn_d(k, j) : INFINITY;
    }
}

#include "eig3_dev.h"

// Eigen 3.3.4 ColPivHouseholderQR<Matrix<double,5,3>>::compute + solve(-1): see oracle/linalg.h.  Register-only:
// the three columns are separate 5-vectors and the column pivoting is done with conditional swaps.
struct Col5 {
    double v[5];
};
__device__ __forceinline__ void col_swap(Col5& a, Col5& b, bool doit) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const double x = a.v[r], y = b.v[r];
        a.v[r] = doit ? y : x;
        b.v[r] = doit ? x : y;
    }
}
__device__ __forceinline__ void dswap(double& a, double& b, bool doit) {
    const double x = a, y = b;
    a = doit ? y : x;
    b = doit ? x : y;
}
__device__ __forceinline__ void iswap(int& a, int& b, bool doit) {
    const int x = a, y = b;
    a = doit ? y : x;
    b = doit ? x : y;
}
// Householder step K on column `ck` (rows K..4), applied to the `NO` remaining columns; norm downdate for them.
template <int K>
__device__ __forceinline__ void qr_step(Col5& ck, Col5& o1, Col5& o2, double& nu1, double& nd1, double& nu2, double& nd2,
                                        double& tau_out, bool has1, bool has2) {
    const double tol = 2.2250738585072014e-308;
    const double norm_downdate_threshold = 1.4901161193847656e-08;  // sqrt(eps)
    double tailSqNorm = 0;
#pragma unroll
    for (int r = K + 1; r < 5; ++r) tailSqNorm += ck.v[r] * ck.v[r];
    const double c0 = ck.v[K];
    double tau, beta;
    if (tailSqNorm <= tol) {
        tau = 0;
        beta = c0;
#pragma unroll
        for (int r = K + 1; r < 5; ++r) ck.v[r] = 0;
    } else {
        beta = sqrt(c0 * c0 + tailSqNorm);
        if (c0 >= 0.0) beta = -beta;
#pragma unroll
        for (int r = K + 1; r < 5; ++r) ck.v[r] = ck.v[r] / (c0 - beta);
        tau = (beta - c0) / beta;
    }
    tau_out = tau;
    ck.v[K] = beta;
    if (tau != 0.0) {
        if (has1) {
            double tmp = 0;
#pragma unroll
            for (int r = K + 1; r < 5; ++r) tmp += ck.v[r] * o1.v[r];
            tmp += o1.v[K];
            o1.v[K] -= tau * tmp;
#pragma unroll
            for (int r = K + 1; r < 5; ++r) o1.v[r] -= tau * ck.v[r] * tmp;
        }
        if (has2) {
            double tmp = 0;
#pragma unroll
            for (int r = K + 1; r < 5; ++r) tmp += ck.v[r] * o2.v[r];
            tmp += o2.v[K];
            o2.v[K] -= tau * tmp;
#pragma unroll
            for (int r = K + 1; r < 5; ++r) o2.v[r] -= tau * ck.v[r] * tmp;
        }
    }
    auto downdate = [&](Col5& o, double& nu, double& nd) {
        if (nu != 0.0) {
            double temp = fabs(o.v[K]) / nu;
            temp = (1.0 + temp) * (1.0 - temp);
            temp = temp < 0.0 ? 0.0 : temp;
            const double ratio = nu / nd;
            const double temp2 = temp * (ratio * ratio);
            if (temp2 <= norm_downdate_threshold) {
                double s = 0;
#pragma unroll
                for (int r = K + 1; r < 5; ++r) s += o.v[r] * o.v[r];
                nd = sqrt(s);
                nu = nd;
            } else {
                nu *= sqrt(temp);
            }
        }
    };
    if (has1) downdate(o1, nu1, nd1);
    if (has2) downdate(o2, nu2, nd2);
}

__device__ void plane_fit5(const double (*Ain)[3], double* xout) {
    Col5 c0, c1, c2;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        c0.v[r] = Ain[r][0];
        c1.v[r] = Ain[r][1];
        c2.v[r] = Ain[r][2];
    }
    auto colnorm = [](const Col5& c) {
        double s = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) s += c.v[r] * c.v[r];
        return sqrt(s);
    };
    double nd0 = colnorm(c0), nd1 = colnorm(c1), nd2 = colnorm(c2);
    double nu0 = nd0, nu1 = nd1, nu2 = nd2;
    const double eps = 2.220446049250313e-16;
    double maxn = nu0;
    if (nu1 > maxn) maxn = nu1;
    if (nu2 > maxn) maxn = nu2;
    const double threshold_helper = (maxn * eps) * (maxn * eps) / 5.0;
    int nonzero_pivots = 3;
    int p0 = 0, p1 = 1, p2 = 2;  // perm: position -> original column
    double tau0, tau1, tau2;
    // ---- k = 0: pivot among columns 0,1,2 (first maximum) ----
    {
        int big = 0;
        double bigv = nu0;
        if (nu1 > bigv) { bigv = nu1; big = 1; }
        if (nu2 > bigv) { bigv = nu2; big = 2; }
        if (nonzero_pivots == 3 && bigv * bigv < threshold_helper * 5.0) nonzero_pivots = 0;
        col_swap(c0, c1, big == 1);
        dswap(nu0, nu1, big == 1);
        dswap(nd0, nd1, big == 1);
        iswap(p0, p1, big == 1);
        col_swap(c0, c2, big == 2);
        dswap(nu0, nu2, big == 2);
        dswap(nd0, nd2, big == 2);
        iswap(p0, p2, big == 2);
        qr_step<0>(c0, c1, c2, nu1, nd1, nu2, nd2, tau0, true, true);
    }
    // ---- k = 1: pivot among columns 1,2 ----
    {
        const bool sw = nu2 > nu1;
        const double bigv = sw ? nu2 : nu1;
        if (nonzero_pivots == 3 && bigv * bigv < threshold_helper * 4.0) nonzero_pivots = 1;
        col_swap(c1, c2, sw);
        dswap(nu1, nu2, sw);
        dswap(nd1, nd2, sw);
        iswap(p1, p2, sw);
        double dum_u = 0, dum_d = 1;
        Col5 dummy = c2;
        qr_step<1>(c1, c2, dummy, nu2, nd2, dum_u, dum_d, tau1, true, false);
    }
    // ---- k = 2 ----
    {
        if (nonzero_pivots == 3 && nu2 * nu2 < threshold_helper * 3.0) nonzero_pivots = 2;
        double du1 = 0, dd1 = 1, du2 = 0, dd2 = 1;
        Col5 dmy1 = c2, dmy2 = c2;
        qr_step<2>(c2, dmy1, dmy2, du1, dd1, du2, dd2, tau2, false, false);
    }
    xout[0] = xout[1] = xout[2] = 0;
    if (nonzero_pivots == 0) return;
    double c[5] = {-1, -1, -1, -1, -1};
    // c = Q^T b : apply H_0, H_1, H_2 (only the first nonzero_pivots reflectors)
    auto apply = [&](const Col5& h, double tau, int K) {
        if (tau != 0.0) {
            double tmp = 0;
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r > K) tmp += h.v[r] * c[r];
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r == K) tmp += c[r];
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r == K) c[r] -= tau * tmp;
#pragma unroll
            for (int r = 0; r < 5; ++r)
                if (r > K) c[r] -= tau * h.v[r] * tmp;
        }
    };
    if (nonzero_pivots > 0) apply(c0, tau0, 0);
    if (nonzero_pivots > 1) apply(c1, tau1, 1);
    if (nonzero_pivots > 2) apply(c2, tau2, 2);
    // back substitution on the leading nonzero_pivots block of R (R(i,j) = column j, row i)
    double y0 = 0, y1 = 0, y2 = 0;
    if (nonzero_pivots > 2) y2 = c[2] / c2.v[2];
    if (nonzero_pivots > 1) {
        double sacc = c[1];
        if (nonzero_pivots > 2) sacc -= c2.v[1] * y2;
        y1 = sacc / c1.v[1];
    }
    {
        double sacc = c[0];
        if (nonzero_pivots > 1) sacc -= c1.v[0] * y1;
        if (nonzero_pivots > 2) sacc -= c2.v[0] * y2;
        y0 = sacc / c0.v[0];
    }
    // dst.row(perm[i]) = c.row(i)
    double x0 = 0, x1 = 0, x2 = 0;
    auto put = [&](int dst, double v) {
        x0 = dst == 0 ? v : x0;
        x1 = dst == 1 ? v : x1;
        x2 = dst == 2 ? v : x2;
    };
    put(p0, y0);
    if (nonzero_pivots > 1) put(p1, y1);
    if (nonzero_pivots > 2) put(p2, y2);
    xout[0] = x0;
    xout[1] = x1;
    xout[2] = x2;
}

// original (unsorted) coordinates of a neighbour are needed in index order: the sorted grid carries xyz next
// to the original index, so the search returns positions; keep a second lookup by original index.
struct AssocParams {
    int first, B, MF;
    MmlGrid g[2];
    const float4* map_orig[2];  // unsorted map clouds (index = original index)
    // global cube map (a12): tagged grids, unsorted clouds, per-cube point counts, grid centre
    MmlGrid gg[2];
    const float4* gmap_orig[2];
    const int* cube_cnt[2];
    int have_g[2];
    int cen[3];
    const float4* ft[2];
    const int* ft_n;
    MmlLineFactor* lf;
    MmlPlaneFactor* pf;
    const double* Twl;  // count x 16
    float thres;     // search bound (float)
    double thres_d;  // gate, compared as in the reference: (double)d2[4] < thres_dist
    int map_m[2];
    int count;
    int fresh_all;  // small batches: every feature goes to the 16-lane search of k_associate_hard from ring 0
    const int* work_off;
    int* hard_count;   // queue of features whose 5-NN search goes beyond ring 1
    int4* hard_list;
    float* hard_knn;   // 10 floats per queued feature: the top-5 (d2, index) found in rings 0-1
};

__device__ __forceinline__ void tf_point(const double* T, double x, double y, double z, double& ox, double& oy,
                                         double& oz) {
    ox = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    oy = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    oz = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

// one lane per feature: grid.x covers MF features, grid.y = slot, grid.z = kind
// model fit + factor record for one feature (a14 / a15 / a16), given its exact 5 nearest map points
__device__ __forceinline__ void store_none(const AssocParams& P, int kind, int b, int i) {
    if (kind == 0) {
        MmlLineFactor out;
        memset(&out, 0, sizeof(out));
        out.src = -1;
        P.lf[(size_t)b * P.MF + i] = out;
    } else {
        MmlPlaneFactor out;
        memset(&out, 0, sizeof(out));
        out.src = -1;
        P.pf[(size_t)b * P.MF + i] = out;
    }
}

// Map_Manager.cpp:583-629 FindUsed{Corner,Surf}Map
__device__ __forceinline__ int find_used_map(float x, float y, float z, const int* cen) {
    int cubeI = int((x + 25.0) / 50.0) + cen[2];
    int cubeJ = int((y + 25.0) / 50.0) + cen[0];
    int cubeK = int((z + 25.0) / 50.0) + cen[1];
    if (x + 25.0 < 0) cubeI--;
    if (y + 25.0 < 0) cubeJ--;
    if (z + 25.0 < 0) cubeK--;
    if (cubeI >= 0 && cubeI < 21 && cubeJ >= 0 && cubeJ < 21 && cubeK >= 0 && cubeK < 11)
        return cubeI + 21 * cubeJ + 21 * 21 * cubeK;  // ToIndex, Map_Manager.cpp:65-67
    return 5000;
}

// returns true (and stores the factor) when the neighbourhood passes the gate and yields a model
__device__ __forceinline__ bool fit_and_store(const AssocParams& P, int kind, int b, int i, const float4 f, const double* T,
                                              float sx, float sy, float sz, bool ok, const Knn5& k, const float4* mp) {
    if (kind == 0) {
        MmlLineFactor out;
        out.src = -1;
        out.error = 0;
        if (ok) {
            float nx[5], ny[5], nz[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                float4 q = mp[knn_id(k, j)];
                nx[j] = q.x;
                ny[j] = q.y;
                nz[j] = q.z;
            }
            float cx = 0, cy = 0, cz = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                cx += nx[j];
                cy += ny[j];
                cz += nz[j];
            }
            cx /= 5;
            cy /= 5;
            cz /= 5;
            float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                float ax = nx[j] - cx, ay = ny[j] - cy, az = nz[j] - cz;
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
            double ev[3], ud[3];
            eig3_sym(a11, a12, a22, a13, a23, a33, ev, ud);
            if (ev[2] > 3 * ev[1]) {
                float x1 = cx + 0.1 * ud[0];
                float y1 = cy + 0.1 * ud[1];
                float z1 = cz + 0.1 * ud[2];
                float x2 = cx - 0.1 * ud[0];
                float y2 = cy - 0.1 * ud[1];
                float z2 = cz - 0.1 * ud[2];
                out.ori[0] = f.x;
                out.ori[1] = f.y;
                out.ori[2] = f.z;
                out.p1[0] = x1;
                out.p1[1] = y1;
                out.p1[2] = z1;
                out.p2[0] = x2;
                out.p2[1] = y2;
                out.p2[2] = z2;
                out.src = i;
                // FeatureLine::ComputeError (Estimator.h:71-83)
                double Px, Py, Pz;
                tf_point(T, f.x, f.y, f.z, Px, Py, Pz);
                double ax = x1, ay = y1, az = z1, bx = x2, by = y2, bz = z2;
                double l12 = sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz));
                double c0 = (Px - ax) * (Py - by) - (Px - bx) * (Py - ay);
                double c1 = (Px - ax) * (Pz - bz) - (Px - bx) * (Pz - az);
                double c2 = (Py - ay) * (Pz - bz) - (Py - by) * (Pz - az);
                double a012 = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
                out.error = a012 / l12;
            }
        }
        if (out.src >= 0) P.lf[(size_t)b * P.MF + i] = out;
        return out.src >= 0;
    } else {
        MmlPlaneFactor out;
        out.src = -1;
        out.error = 0;
        out._pad = 0;
        if (ok) {
            double A[5][3];
            float nx[5], ny[5], nz[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                float4 q = mp[knn_id(k, j)];
                nx[j] = q.x;
                ny[j] = q.y;
                nz[j] = q.z;
                A[j][0] = q.x;
                A[j][1] = q.y;
                A[j][2] = q.z;
            }
            double X[3];
            plane_fit5(A, X);
            float pa = X[0], pb = X[1], pc = X[2], pd = 1;
            float ps = sqrtf(pa * pa + pb * pb + pc * pc);
            pa /= ps;
            pb /= ps;
            pc /= ps;
            pd /= ps;
            bool planeValid = true;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (fabs((double)(pa * nx[j] + pb * ny[j] + pc * nz[j] + pd)) > 0.2) {
                    planeValid = false;
                    break;
                }
            }
            if (planeValid) {
                double dist = pa * sx + pb * sy + pc * sz + pd;  // float expression, :740-742
                out.ori[0] = f.x;
                out.ori[1] = f.y;
                out.ori[2] = f.z;
                out.omega[0] = pa;
                out.omega[1] = pb;
                out.omega[2] = pc;
                out.proj[0] = (double)sx - dist * (double)pa;
                out.proj[1] = (double)sy - dist * (double)pb;
                out.proj[2] = (double)sz - dist * (double)pc;
                double Px, Py, Pz;
                tf_point(T, f.x, f.y, f.z, Px, Py, Pz);
                double ex = Px - out.proj[0], ey = Py - out.proj[1], ez = Pz - out.proj[2];
                out.error = sqrt((ex * ex + ey * ey) + ez * ez);  // Estimator.h:118-121
                out.src = i;
            }
        }
        if (out.src >= 0) P.pf[(size_t)b * P.MF + i] = out;
        return out.src >= 0;
    }
}

// pass 1: one feature per lane, rings 0 and 1 of the grid privately.  Features whose search is still open are
// queued for pass 2 instead of stalling their wavefront (they cluster spatially, so without the queue a few
// wavefronts full of far queries set the kernel time).
// work decomposition: item = (kind, slot) pair, ceil(nf / 128) workgroup-sized chunks each; work_off is the exclusive
// prefix over the 2 * count items.  A fixed-size grid walks the chunks, so no empty workgroups are dispatched
// (with max_features = 8192 the dense grid spent more time retiring ~30 k empty workgroups than computing).
// (256 threads, four items per thread -- as k_queue_prefix: a sixteen-wavefront workgroup waits for a CU with four free wave slots
//  on every SIMD while another lane's kernel keeps refilling them)
constexpr int AP_THREADS = 256;
__global__ __launch_bounds__(AP_THREADS) void k_assoc_prefix(int first, int count, int B, const int* ft_n, int* work_off, int* hard_count) {
    __shared__ int s_wsum[16];  // totals of the sixteen (slab, wavefront) groups of a 1024-item turn, in item order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *hard_count = 0;  // the far-query list of this call starts empty
    const int nitems = 2 * count;
    int acc = 0;  // running prefix over tiles of 1024 items
    for (int t0 = 0; t0 < nitems; t0 += 4 * AP_THREADS) {
        int v[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = t0 + q * AP_THREADS + tid;
            v[q] = 0;
            if (it < nitems) {
                const int kind = it / count, slot = it % count;
                const int nf = ft_n[kind * B + first + slot];
                v[q] = nf > 0 ? (nf + 127) / 128 : 0;
            }
        }
        // inclusive scan inside the wavefront (shuffles), then the 16 group totals
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int xx = v[q];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(xx, o);
                if (lane >= o) xx += y;
            }
            x[q] = xx;
            if (lane == 63) s_wsum[q * 4 + wave] = xx;
        }
        __syncthreads();
        int tile = 0, base[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = s_wsum[w];
            tile += t;
#pragma unroll
            for (int q = 0; q < 4; ++q) base[q] += w < q * 4 + wave ? t : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = t0 + q * AP_THREADS + tid;
            if (it < nitems) work_off[it] = acc + base[q] + x[q] - v[q];
        }
        acc += tile;
        __syncthreads();
    }
    if (tid == 0) work_off[nitems] = acc;
}

// What pass 1 leaves for a feature, parked in the first 32 bytes of the feature's own (not yet written) factor record:
// [0] = ids 0..3, [1] = (id 4, bits of d2[4], state | stage << 2, cube).
constexpr int REC_NONE = 0, REC_READY = 1, REC_QUEUED = 2;
__device__ __forceinline__ int4* assoc_rec(const AssocParams& P, int kind, int b, int i) {
    return kind == 0 ? reinterpret_cast<int4*>(P.lf + (size_t)b * P.MF + i) : reinterpret_cast<int4*>(P.pf + (size_t)b * P.MF + i);
}
static_assert(sizeof(MmlLineFactor) >= 32 && sizeof(MmlLineFactor) % 16 == 0, "record parking needs 32 aligned bytes");
static_assert(sizeof(MmlPlaneFactor) >= 32 && sizeof(MmlPlaneFactor) % 16 == 0, "record parking needs 32 aligned bytes");

// locate the (kind, slot) item that owns chunk w: last item with work_off[item] <= w
__device__ __forceinline__ int assoc_item(const AssocParams& P, int w) {
    int lo = 0, hi = 2 * P.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (P.work_off[mid] <= w)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

constexpr int HARD_REDO = 1 << 30;   // queue entry: search again in the local map (the cube-stage fit failed)
constexpr int HARD_FRESH = 1 << 29;  // queue entry: nothing searched yet, start at ring 0 without a list
// Search only: the model fit (double-precision eigen / QR code, ~150 registers) lives in k_associate_fit_all, so this
// kernel keeps a small register footprint and enough wavefronts in flight to hide its dependent gathers.
// Occupancy the three association kernels are compiled for (wavefronts per SIMD).  The search is a chain of dependent gathers
// (cell offsets, then rows of points): 77 registers gave six wavefronts per SIMD; held to 64 (36 bytes of scratch) it runs
// eight and is 10 % faster (0.289 -> 0.259 ms per 1024 scans); the fit 136 -> 128 registers (three -> four): 0.087 -> 0.080.
// The far-query search loses at 6 (0.073 -> 0.086: 96 bytes of scratch inside its shell loop) and is indifferent at 5.
#ifndef MML_AW_SEARCH
#define MML_AW_SEARCH 8
#define MML_AW_FIT 4
#define MML_AW_HARD 1
#endif
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(MML_AW_SEARCH))) void k_associate(AssocParams P) {
    const int nitems = 2 * P.count;
    const int total = P.work_off[nitems];
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int item = assoc_item(P, w);
        const int kind = item / P.count, slot = item % P.count;
        const int b = slot + P.first;
        const int i = (w - P.work_off[item]) * 128 + threadIdx.x;
        const int nf = P.ft_n[kind * P.B + b];
        if (i >= nf) continue;
        int4* rec = assoc_rec(P, kind, b, i);
        const double* T = P.Twl + 16 * slot;
        const float4 f = P.ft[kind][(size_t)b * P.MF + i];
        // Map_Manager.cpp:75-89 pointAssociateToMap: double transform stored to float
        double wx, wy, wz;
        tf_point(T, f.x, f.y, f.z, wx, wy, wz);
        const float sx = wx, sy = wy, sz = wz;
        // :192-196 / :621-625: features outside the 21 x 11 x 21 cube grid, or NaN after the transform, get no factor
        const int cube = find_used_map(sx, sy, sz, P.cen);
        // stage 0: the cube's own cloud when it holds > 100 corner / > 50 surf points (:198, :627); stage 1: local map
        const int stage = (cube != 5000 && P.have_g[kind] && P.cube_cnt[kind][cube] > (kind == 0 ? 100 : 50)) ? 0 : 1;
        if (cube == 5000 || isnan(sx) || isnan(sy) || isnan(sz) || (stage == 1 && !(P.map_m[kind] > 20))) {  // (:283 / :702)
            rec[1] = make_int4(0, 0, REC_NONE, 0);
            continue;
        }
        if (P.fresh_all) {
            // A handful of scans cannot fill the device with one lane per feature -- the ~80 candidates of rings 0-1 are then
            // one lane's serial chain (74 us for one scan).  Sixteen lanes per feature share the rows instead.
            const int hw = atomicAdd(P.hard_count, 1);
            P.hard_list[hw] = make_int4(slot, kind, i, stage | (cube << 1) | HARD_FRESH);
            rec[1] = make_int4(0, 0, REC_QUEUED, 0);
            continue;
        }
        // one call site per stage: each names ONE grid of the (wave-uniform) kind, so the grid descriptor stays in scalar
        // registers - selected per lane it is re-read from memory in front of every row
        const int ukind = __builtin_amdgcn_readfirstlane(kind);
        Knn5 k;
        knn_init(k);
        bool done = false;
        int rmax = 0;
        if (stage == 0) {
            const MmlGrid& g = P.gg[ukind];
            const KnnQuery q = knn_query(g, sx, sy, sz);
            rmax = (int)ceilf(sqrtf(P.thres) * g.inv_cell) + 1;
            done = scan_rings01(g, q, rmax, P.thres, k, cube);
        } else {
            const MmlGrid& g = P.g[ukind];
            const KnnQuery q = knn_query(g, sx, sy, sz);
            rmax = (int)ceilf(sqrtf(P.thres) * g.inv_cell) + 1;
            done = scan_rings01(g, q, rmax, P.thres, k, -1);
        }
        if (!done && rmax >= 2) {
            const int hw = atomicAdd(P.hard_count, 1);
            P.hard_list[hw] = make_int4(slot, kind, i, stage | (cube << 1));
            float* hd = P.hard_knn + 10 * (size_t)hw;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                hd[j] = knn_d(k, j);
                hd[5 + j] = __int_as_float(knn_id(k, j));
            }
            rec[1] = make_int4(0, 0, REC_QUEUED, 0);  // finished by k_associate_hard / k_associate_fit
            continue;
        }
        rec[0] = make_int4(knn_id(k, 0), knn_id(k, 1), knn_id(k, 2), knn_id(k, 3));
        rec[1] = make_int4(knn_id(k, 4), __float_as_int(knn_d(k, 4)), REC_READY | (stage << 2), cube);
    }
}

// Model fit of every feature pass 1 finished itself, one lane each.  A feature whose cube neighbourhood (stage 0) yields
// no model is appended to the far-query list with the "search again in the local map" mark (:283 / :702).
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(MML_AW_FIT))) void k_associate_fit_all(AssocParams P) {
    const int nitems = 2 * P.count;
    const int total = P.work_off[nitems];
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        const int item = assoc_item(P, w);
        const int kind = item / P.count, slot = item % P.count;
        const int b = slot + P.first;
        const int i = (w - P.work_off[item]) * 128 + threadIdx.x;
        const int nf = P.ft_n[kind * P.B + b];
        if (i >= nf) continue;
        const int4* rec = assoc_rec(P, kind, b, i);
        const int4 r0 = rec[0], r1 = rec[1];
        const int state = r1.z & 3, stage = r1.z >> 2, cube = r1.w;
        if (state == REC_QUEUED) continue;
        if (state == REC_NONE) {
            store_none(P, kind, b, i);
            continue;
        }
        const double* T = P.Twl + 16 * slot;
        const float4 f = P.ft[kind][(size_t)b * P.MF + i];
        double wx, wy, wz;
        tf_point(T, f.x, f.y, f.z, wx, wy, wz);
        const float sx = wx, sy = wy, sz = wz;
        Knn5 k;
        knn_init(k);
        const float d5 = __int_as_float(r1.y);  // (the fit reads the five indices and this gate only)
        k.key[0] = knn_key(d5, r0.x);
        k.key[1] = knn_key(d5, r0.y);
        k.key[2] = knn_key(d5, r0.z);
        k.key[3] = knn_key(d5, r0.w);
        k.key[4] = knn_key(d5, r1.x);
        bool ok = (double)d5 < P.thres_d;  // :201,285 / :631,705
#ifdef MML_EXP_NOFIT
        ok = false;
#endif
        const bool stored = fit_and_store(P, kind, b, i, f, T, sx, sy, sz, ok, k, stage == 0 ? P.gmap_orig[kind] : P.map_orig[kind]);
        if (!stored) {
            if (stage == 0 && P.map_m[kind] > 20) {
                const int hw = atomicAdd(P.hard_count, 1);
                P.hard_list[hw] = make_int4(slot, kind, i, (cube << 1) | HARD_REDO);
            } else {
                store_none(P, kind, b, i);
            }
        }
    }
}

// pass 2: queued features, one 16-lane group each (4 per wavefront).  The group continues from the top-5 list pass 1
// left after ring 1: its lanes split the rows of every further shell and merge their private lists with shuffles
// (xor 8,4,2,1 stays inside the group).  Lane 0 of the group then runs the model fit.
// Queue entry word: bit 0 stage (0 cube cloud, 1 local map), bits 1..14 cube, bit 30 HARD_REDO "search again in the
// local map".

// Far queries, search only: one 16-lane group per queued feature, the lanes split the rows of every shell and merge
// their private top-5 lists with shuffles.  The result goes back to hard_knn; the model fit runs in k_associate_fit with
// one LANE per feature -- inside this kernel it would run on one lane in sixteen.
// round 0: every queued feature, continuing after ring 1 of the stage it was queued in.
// round 1: the features whose cube-stage fit failed (HARD_REDO), local map from ring 0 (:283 / :702).
// SPAN = lanes that share one query (they split the rows of every shell): 16 for batches -- many queries, throughput --, 64 for
// the live path's handful of scans, where the kernel lasts as long as its slowest query and a far query walks up to 169 rows a shell.
template <int SPAN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MML_AW_HARD))) void k_associate_hard(AssocParams P, int round) {
    const int gl = threadIdx.x & (SPAN - 1);
    const int group = (blockIdx.x * 256 + threadIdx.x) / SPAN;
    const int ngroups = (gridDim.x * 256) / SPAN;
    const int total = *P.hard_count;
    for (int w0 = 0; w0 < total; w0 += ngroups) {
        const int w = w0 + group;
        bool live = w < total;
        int slot = 0, kind = 0, i = 0, b = P.first, stage = 1, cube = 0;
        bool fresh = false;
        float sx = 0, sy = 0, sz = 0;
        Knn5 loc, best;
        knn_init(loc);
        knn_init(best);
        if (live) {
            const int4 e = P.hard_list[w];
            if ((round == 1) != ((e.w & HARD_REDO) != 0)) live = false;  // round 0: not the entries k_associate_fit_all appended
            slot = e.x;
            kind = e.y;
            i = e.z;
            stage = round == 1 ? 1 : (e.w & 1);
            cube = (e.w >> 1) & 0x3fff;
            fresh = (e.w & HARD_FRESH) != 0;
            b = slot + P.first;
        }
        if (live) {
            const float4 f = P.ft[kind][(size_t)b * P.MF + i];
            double wx, wy, wz;
            tf_point(P.Twl + 16 * slot, f.x, f.y, f.z, wx, wy, wz);
            sx = wx;
            sy = wy;
            sz = wz;
            if (round == 0 && gl == 0 && !fresh) {  // the list pass 1 left after ring 1
                const float* hd = P.hard_knn + 10 * (size_t)w;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    loc.key[j] = knn_key(hd[j], __float_as_int(hd[5 + j]));
                }
            }
        }
        const int r0 = (round == 0 && !fresh) ? 2 : 0;
        const float start_d5 = __shfl(knn_d(loc, 4), (threadIdx.x & 63) & ~(SPAN - 1));  // the group's lane 0 (INFINITY unless round 0)
        // the entry's grid descriptor, selected field by field into registers once (a reference to one of four kernel
        // argument structs picked per lane is re-read from memory at every use)
        MmlGrid g = P.g[0];
        {
            const bool k1 = kind == 1, s0 = stage == 0;
            const MmlGrid &a = P.g[1], &c = P.gg[0], &d = P.gg[1];
#define MML_PICK(field) g.field = s0 ? (k1 ? d.field : c.field) : (k1 ? a.field : g.field)
            MML_PICK(pts);
            MML_PICK(cell_start);
            MML_PICK(tags);
            MML_PICK(origin[0]);
            MML_PICK(origin[1]);
            MML_PICK(origin[2]);
            MML_PICK(cell);
            MML_PICK(inv_cell);
            MML_PICK(dim[0]);
            MML_PICK(dim[1]);
            MML_PICK(dim[2]);
#undef MML_PICK
        }
        const int mytag = stage == 0 ? cube : -1;
        const KnnQuery q = knn_query(g, sx, sy, sz);
        const int rmax = live ? (int)ceilf(sqrtf(P.thres) * g.inv_cell) + 1 : 0;
        bool sdone = !live;
        if (!sdone && r0 > rmax) {
            sdone = true;
            lanes_merge5<SPAN>(loc, best);
        }
        for (int r = r0;; ++r) {
            if (!sdone && r > rmax) sdone = true;
            if (__all(sdone)) break;
            if (!sdone) {
                // the rows of the shell that lie inside the grid
                // (best: merged list of the shells before this one; loc: this lane's own list; lane 0's starting list
                //  of round 0 holds five real points, so its d[4] bounds the result as well)
                const int ylo = max(q.hy - r, 0), yhi = min(q.hy + r, g.dim[1] - 1), zlo = max(q.hz - r, 0), zhi = min(q.hz + r, g.dim[2] - 1);
                const int wy = yhi - ylo + 1, wz = zhi - zlo + 1;
                if (wy > 0 && wz > 0)
                    for (int t = gl; t < wy * wz; t += SPAN)
                        scan_shell_row(g, q, r, ylo + (t % wy), zlo + (t / wy), loc, mytag, fminf(fminf(knn_d(best, 4), knn_d(loc, 4)), start_d5));
            }
            lanes_merge5<SPAN>(loc, best);
            if (!sdone && knn_done(g, q.inset, r, knn_d(best, 4), P.thres)) sdone = true;
        }
        if (live && gl == 0) {
            float* hd = P.hard_knn + 10 * (size_t)w;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                hd[j] = knn_d(best, j);
                hd[5 + j] = __int_as_float(knn_id(best, j));
            }
        }
    }
}

// model fit of the far queries, one lane per queued feature (see k_associate_hard)
__global__ __launch_bounds__(128) void k_associate_fit(AssocParams P, int round) {
    const int total = *P.hard_count;
    for (int w = blockIdx.x * 128 + threadIdx.x; w < total; w += gridDim.x * 128) {
        const int4 e = P.hard_list[w];
        if ((round == 1) != ((e.w & HARD_REDO) != 0)) continue;
        const int slot = e.x, kind = e.y, i = e.z, b = slot + P.first;
        const int stage = round == 1 ? 1 : (e.w & 1);
        const float4 f = P.ft[kind][(size_t)b * P.MF + i];
        double wx, wy, wz;
        tf_point(P.Twl + 16 * slot, f.x, f.y, f.z, wx, wy, wz);
        const float sx = wx, sy = wy, sz = wz;
        Knn5 best;
        const float* hd = P.hard_knn + 10 * (size_t)w;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            best.key[j] = knn_key(hd[j], __float_as_int(hd[5 + j]));
        }
        const bool ok = (double)knn_d(best, 4) < P.thres_d;
        const bool stored = fit_and_store(P, kind, b, i, f, P.Twl + 16 * slot, sx, sy, sz, ok, best,
                                          stage == 0 ? P.gmap_orig[kind] : P.map_orig[kind]);
        if (!stored) {
            if (stage == 0 && P.map_m[kind] > 20)
                P.hard_list[w].w = e.w | HARD_REDO;  // fall back to the local map (:283 / :702)
            else
                store_none(P, kind, b, i);
        }
        if (round == 1) P.hard_list[w].w = e.w & ~HARD_REDO;
    }
}

// per-slot statistics: counts, used counts, normal Gram matrix (checkLocalizability input)
__global__ __launch_bounds__(256) void k_assoc_stats(int first, int B, int MF, const int* ft_n, const MmlLineFactor* lf,
                                                    const MmlPlaneFactor* pf, double* stats) {
    __shared__ double s_red[4][13];
    const int b = blockIdx.x + first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[13];
    for (int k = 0; k < 13; ++k) acc[k] = 0;
    const int nl = ft_n[b], np = ft_n[B + b];
    for (int i = tid; i < nl; i += 256) {
        const MmlLineFactor& f = lf[(size_t)b * MF + i];
        if (f.src >= 0) {
            acc[0] += 1;
            if (fabs(f.error) > 1e-5) acc[2] += 1;
        }
    }
    for (int i = tid; i < np; i += 256) {
        const MmlPlaneFactor& f = pf[(size_t)b * MF + i];
        if (f.src >= 0) {
            acc[1] += 1;
            if (fabs(f.error) > 1e-5) acc[3] += 1;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) acc[4 + 3 * r + c] += (double)f.omega[r] * (double)f.omega[c];
        }
    }
    for (int k = 0; k < 13; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 13) stats[16 * b + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

}  // namespace

// Builds the radix-sorted grid `g` over m points (host xyz), keeping the unsorted cloud in `orig`.
static int build_grid_into(mml_ctx* ctx, MmlGrid& g, float4* orig, const float* h_xyz, int m, float cell,
                           const uint16_t* d_tag_orig) {
    hipStream_t s = MML_STREAM(ctx);
    g.m = m;
    if (m == 0) {
        g.dim[0] = g.dim[1] = g.dim[2] = 1;
        g.ncell = 1;
        g.cell = 1.f;
        g.inv_cell = 1.f;
        g.origin[0] = g.origin[1] = g.origin[2] = 0.f;
        MML_HIP(hipMemsetAsync(g.cell_start, 0, 2 * sizeof(int), s));
        return MML_OK;
    }
    if (h_xyz) {  // host xyz (3 floats) -> device float4 (original order, w unused); null: `orig` is already filled
        std::vector<float4> tmp((size_t)m);
        for (int i = 0; i < m; ++i) tmp[i] = make_float4(h_xyz[3 * i], h_xyz[3 * i + 1], h_xyz[3 * i + 2], 0.f);
        MML_HIP(hipMemcpyAsync(orig, tmp.data(), sizeof(float4) * (size_t)m, hipMemcpyHostToDevice, s));
        MML_HIP(hipStreamSynchronize(s));  // tmp goes out of scope
    }
    MmlStageScope t(ctx, "map_build");
    float init[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float* d_bbox = reinterpret_cast<float*>(ctx->d_misc);
    MML_HIP(hipMemcpyAsync(d_bbox, init, sizeof(init), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bbox, dim3(256), dim3(256), 0, s, orig, m, d_bbox);
    float bbox[6];
    MML_HIP(hipMemcpyAsync(bbox, d_bbox, sizeof(bbox), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    const long long max_cells = (long long)4 * ctx->MM + 4096;
    const int blocks = (m + 255) / 256;
    int dim[3];
    // The configured cell edge suits a voxel-filtered map (<= 1 point per leaf).  If the cloud is denser than
    // that, shrink the cell until an occupied cell holds <= 12 points on average (bounded by the cell budget).
    for (int round = 0;; ++round) {
        for (;;) {
            long long total = 1;
            for (int c = 0; c < 3; ++c) {
                dim[c] = (int)floorf((bbox[3 + c] - bbox[c]) / cell) + 1;
                if (dim[c] < 1) dim[c] = 1;
                total *= dim[c];
            }
            if (total <= max_cells) break;
            cell *= 1.26f;
        }
        g.cell = cell;
        g.inv_cell = 1.0f / cell;
        for (int c = 0; c < 3; ++c) {
            g.origin[c] = bbox[c];
            g.dim[c] = dim[c];
        }
        g.ncell = dim[0] * dim[1] * dim[2];
        GridDev gd{g.origin[0], g.origin[1], g.origin[2], g.inv_cell, g.cell, dim[0], dim[1], dim[2], g.ncell};
        hipLaunchKernelGGL(k_cell_keys, dim3(blocks), dim3(256), 0, s, orig, m, gd, ctx->map_keys, ctx->map_vals);
        int bits = 1;
        while ((1ll << bits) < g.ncell) ++bits;
        size_t need = 0;
        MML_HIP(rocprim::radix_sort_pairs(nullptr, need, ctx->map_keys, ctx->map_keys2, ctx->map_vals, ctx->map_vals2,
                                          (size_t)m, 0, bits, s));
        if (need > ctx->sort_tmp_bytes) {
            if (ctx->sort_tmp) MML_HIP(hipFree(ctx->sort_tmp));
            MML_HIP(hipMalloc(&ctx->sort_tmp, need));
            ctx->sort_tmp_bytes = need;
        }
        MML_HIP(rocprim::radix_sort_pairs(ctx->sort_tmp, need, ctx->map_keys, ctx->map_keys2, ctx->map_vals,
                                          ctx->map_vals2, (size_t)m, 0, bits, s));
        int* d_occ = ctx->d_misc + 16;
        MML_HIP(hipMemsetAsync(d_occ, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_count_occupied, dim3(blocks), dim3(256), 0, s, ctx->map_keys2, m, d_occ);
        int occ = 1;
        MML_HIP(hipMemcpyAsync(&occ, d_occ, sizeof(int), hipMemcpyDeviceToHost, s));
        MML_HIP(hipStreamSynchronize(s));
        const double per_cell = (double)m / (occ > 0 ? occ : 1);
        const long long next_total = (long long)g.ncell * 8;
        if (per_cell <= 12.0 || round >= 4 || next_total > max_cells) break;
        cell *= 0.5f;
    }
    hipLaunchKernelGGL(k_gather_sorted, dim3(blocks), dim3(256), 0, s, orig, ctx->map_vals2, m, g.pts);
    if (d_tag_orig) hipLaunchKernelGGL(k_gather_tags, dim3(blocks), dim3(256), 0, s, d_tag_orig, ctx->map_vals2, m, g.tags);
    hipLaunchKernelGGL(k_cell_start, dim3((g.ncell + 1 + 255) / 256), dim3(256), 0, s, ctx->map_keys2, m, g.ncell,
                       g.cell_start);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_build_grid(mml_ctx* ctx, int kind, const float* h_xyz, int m) {
    MML_REQUIRE(kind == 0 || kind == 1, MML_ERR_INVALID, "map kind must be 0 (corner) or 1 (surf)");
    MML_REQUIRE(m >= 0 && m <= ctx->MM, MML_ERR_CAPACITY, "map larger than max_map_points");
    ctx->have_map[kind] = false;
    ctx->grid[kind].tags = nullptr;
    int rc = build_grid_into(ctx, ctx->grid[kind], ctx->map_tmp + (size_t)kind * ctx->MM, h_xyz, m,
                             kind == 0 ? ctx->cfg.cell_corner : ctx->cfg.cell_surf, nullptr);
    if (rc == MML_OK) ctx->have_map[kind] = true;
    return rc;
}

// Same, for a cloud that is already on the device in map_tmp + kind * MM (device-side map upkeep).
int mml_build_grid_device(mml_ctx* ctx, int kind, int m) {
    MML_REQUIRE(kind == 0 || kind == 1, MML_ERR_INVALID, "map kind must be 0 (corner) or 1 (surf)");
    MML_REQUIRE(m >= 0 && m <= ctx->MM, MML_ERR_CAPACITY, "map larger than max_map_points");
    ctx->have_map[kind] = false;
    ctx->grid[kind].tags = nullptr;
    int rc = build_grid_into(ctx, ctx->grid[kind], ctx->map_tmp + (size_t)kind * ctx->MM, nullptr, m,
                             kind == 0 ? ctx->cfg.cell_corner : ctx->cfg.cell_surf, nullptr);
    if (rc == MML_OK) ctx->have_map[kind] = true;
    return rc;
}

// a12: the global map handed to Estimate() (Estimator.cpp:1170-1184) as one concatenated cloud with the cube index
// (ToIndex) of every point.  The per-cube kd-trees become one grid whose points carry their cube as a tag.
static int ensure_global_capacity(mml_ctx* ctx, int kind, int m) {
    hipStream_t s = MML_STREAM(ctx);
    MmlGrid& g = ctx->ggrid[kind];
    if (m > ctx->gmap_cap[kind] || !ctx->cube_cnt[kind]) {
        MML_HIP(hipStreamSynchronize(s));
        if (g.pts) hipFree(g.pts);
        if (g.cell_start) hipFree(g.cell_start);
        if (g.tags) hipFree(g.tags);
        if (ctx->gmap_orig[kind]) hipFree(ctx->gmap_orig[kind]);
        if (ctx->gtag_orig[kind]) hipFree(ctx->gtag_orig[kind]);
        const size_t cap = (size_t)(m > 1024 ? m : 1024);
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&g.pts), sizeof(float4) * cap));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&g.cell_start), sizeof(int) * (4 * (size_t)ctx->MM + 4096 + 2)));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&g.tags), sizeof(uint16_t) * cap));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gmap_orig[kind]), sizeof(float4) * cap));
        MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->gtag_orig[kind]), sizeof(uint16_t) * cap));
        if (!ctx->cube_cnt[kind]) MML_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->cube_cnt[kind]), sizeof(int) * 4851));
        ctx->gmap_cap[kind] = (int)cap;
    }
    return MML_OK;
}

int mml_build_global_grid(mml_ctx* ctx, int kind, const float* h_xyz, const int* h_cube, int m, const int* cen) {
    MML_REQUIRE(kind == 0 || kind == 1, MML_ERR_INVALID, "map kind must be 0 (corner) or 1 (surf)");
    MML_REQUIRE(m >= 0 && m <= ctx->MM, MML_ERR_CAPACITY, "global map larger than max_map_points");
    hipStream_t s = MML_STREAM(ctx);
    ctx->have_gmap[kind] = false;
    if (cen) {
        ctx->cen[0] = cen[0];
        ctx->cen[1] = cen[1];
        ctx->cen[2] = cen[2];
    }
    int rc = ensure_global_capacity(ctx, kind, m);
    if (rc != MML_OK) return rc;
    std::vector<uint16_t> tags((size_t)(m ? m : 1));
    std::vector<int> cnt(4851, 0);
    for (int i = 0; i < m; ++i) {
        MML_REQUIRE(h_cube[i] >= 0 && h_cube[i] < 4851, MML_ERR_INVALID, "cube index outside [0, 4851)");
        tags[i] = (uint16_t)h_cube[i];
        cnt[h_cube[i]]++;
    }
    MML_HIP(hipMemcpyAsync(ctx->cube_cnt[kind], cnt.data(), sizeof(int) * 4851, hipMemcpyHostToDevice, s));
    if (m) MML_HIP(hipMemcpyAsync(ctx->gtag_orig[kind], tags.data(), sizeof(uint16_t) * (size_t)m, hipMemcpyHostToDevice, s));
    MML_HIP(hipStreamSynchronize(s));
    rc = build_grid_into(ctx, ctx->ggrid[kind], ctx->gmap_orig[kind], h_xyz, m,
                         kind == 0 ? ctx->cfg.cell_corner : ctx->cfg.cell_surf, ctx->gtag_orig[kind]);
    if (rc == MML_OK) ctx->have_gmap[kind] = m > 0;
    return rc;
}

// Same from device arrays (device-side cube store, map_global.hip): points, their cube tags and the 4851 cube counts.
int mml_build_global_grid_device(mml_ctx* ctx, int kind, const float4* d_pts, const uint16_t* d_tags, const int* d_cnt, int m,
                                 const int* cen) {
    MML_REQUIRE(kind == 0 || kind == 1, MML_ERR_INVALID, "map kind must be 0 (corner) or 1 (surf)");
    MML_REQUIRE(m >= 0 && m <= ctx->MM, MML_ERR_CAPACITY, "global map larger than max_map_points");
    hipStream_t s = MML_STREAM(ctx);
    ctx->have_gmap[kind] = false;
    ctx->cen[0] = cen[0];
    ctx->cen[1] = cen[1];
    ctx->cen[2] = cen[2];
    int rc = ensure_global_capacity(ctx, kind, m);
    if (rc != MML_OK) return rc;
    MML_HIP(hipMemcpyAsync(ctx->cube_cnt[kind], d_cnt, sizeof(int) * 4851, hipMemcpyDeviceToDevice, s));
    if (m) {
        MML_HIP(hipMemcpyAsync(ctx->gmap_orig[kind], d_pts, sizeof(float4) * (size_t)m, hipMemcpyDeviceToDevice, s));
        MML_HIP(hipMemcpyAsync(ctx->gtag_orig[kind], d_tags, sizeof(uint16_t) * (size_t)m, hipMemcpyDeviceToDevice, s));
    }
    rc = build_grid_into(ctx, ctx->ggrid[kind], ctx->gmap_orig[kind], nullptr, m,
                         kind == 0 ? ctx->cfg.cell_corner : ctx->cfg.cell_surf, ctx->gtag_orig[kind]);
    if (rc == MML_OK) ctx->have_gmap[kind] = m > 0;
    return rc;
}

int mml_launch_knn5(mml_ctx* ctx, int kind, const float* d_q, int nq, float max_d2, int* d_idx, float* d_d2) {
    MmlStageScope t(ctx, "knn5");
    hipLaunchKernelGGL(k_knn5, dim3((nq + 255) / 256), dim3(256), 0, MML_STREAM(ctx), ctx->grid[kind], d_q, nq, max_d2,
                       d_idx, d_d2);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// ---- the numeric core of estimate_timeoffset (unionLidarsAligner.cpp:1077-1153) ----------
namespace {
// pcl::transformPointCloud (PCL 1.8.1 common/impl/transforms.hpp), float, left to right; tf == nullptr: copy
__global__ void k_tf_cloud(const float* xyz, int n, const float* tf, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float4 o = make_float4(x, y, z, 0.f);
    if (tf) {
        o.x = tf[0] * x + tf[1] * y + tf[2] * z + tf[3];
        o.y = tf[4] * x + tf[5] * y + tf[6] * z + tf[7];
        o.z = tf[8] * x + tf[9] * y + tf[10] * z + tf[11];
    }
    out[i] = o;
}
// :1084-1103 squared distance to the nearest neighbour (the exact 5-NN search, first entry)
__global__ __launch_bounds__(256) void k_nn1(MmlGrid g, const float* q, int nq, float* d2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < nq;
    Knn5 k;
    knn5_search(g, valid, valid ? q[3 * i] : 0.f, valid ? q[3 * i + 1] : 0.f, valid ? q[3 * i + 2] : 0.f, INFINITY, k);
    if (valid) d2[i] = knn_d(k, 0);
}
// :1111-1131 one lane per window, the terms added in index order as the reference's loop does
__global__ void k_window_err(const float* q, const float* d2, int res, int sliced, int nwin, double* err) {
    const int cnt = blockIdx.x * blockDim.x + threadIdx.x;
    if (cnt >= nwin) return;
    double sum_error = 0;
    for (int i = cnt * res; i < cnt * res + sliced; ++i) {
        const float x = q[3 * i], y = q[3 * i + 1];
        sum_error += d2[i] + 0.2 * sqrtf(x * x + y * y);
    }
    err[cnt] = sum_error;
}
}  // namespace

// d_velo_xyz / d_livox_xyz: device copies of the inputs (3 floats per point); d_tf: 16 floats or nullptr; scratch and
// outputs are the caller's (capi.hip).  The grid is built into `g` (storage sized for n_velo points by the caller).
int mml_launch_time_offset(mml_ctx* ctx, MmlGrid& g, float4* d_velo4, const float* d_velo_xyz, int n_velo, const float* d_tf,
                           const float* d_livox_xyz, int n_livox, int res, int sliced, int nwin, float* d_nn, double* d_err) {
    hipStream_t s = MML_STREAM(ctx);
    if (n_velo > 0)
        hipLaunchKernelGGL(k_tf_cloud, dim3((n_velo + 255) / 256), dim3(256), 0, s, d_velo_xyz, n_velo, d_tf, d_velo4);
    // an unfiltered scan: start from a 0.5 m cell, the builder halves it where the cloud is dense
    int rc = build_grid_into(ctx, g, d_velo4, nullptr, n_velo, 0.5f, nullptr);
    if (rc != MML_OK) return rc;
    if (n_livox > 0) hipLaunchKernelGGL(k_nn1, dim3((n_livox + 255) / 256), dim3(256), 0, s, g, d_livox_xyz, n_livox, d_nn);
    if (nwin > 0)
        hipLaunchKernelGGL(k_window_err, dim3((nwin + 63) / 64), dim3(64), 0, s, d_livox_xyz, d_nn, res, sliced, nwin, d_err);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_launch_associate(mml_ctx* ctx, int first, int count, const double* d_Twl, double thres_dist, bool with_stats) {
    AssocParams P;
    P.first = first;
    P.B = ctx->B;
    P.MF = ctx->MF;
    for (int k = 0; k < 2; ++k) {
        P.g[k] = ctx->grid[k];
        P.map_orig[k] = ctx->map_tmp + (size_t)k * ctx->MM;
        P.ft[k] = ctx->ft_xyz[k];
        P.map_m[k] = ctx->grid[k].m;
    }
    for (int k = 0; k < 2; ++k) {
        P.gg[k] = ctx->ggrid[k];
        P.gmap_orig[k] = ctx->gmap_orig[k];
        P.cube_cnt[k] = ctx->cube_cnt[k];
        P.have_g[k] = ctx->have_gmap[k] ? 1 : 0;
    }
    P.cen[0] = ctx->cen[0];
    P.cen[1] = ctx->cen[1];
    P.cen[2] = ctx->cen[2];
    P.ft_n = ctx->ft_n;
    P.lf = ctx->lf;
    P.pf = ctx->pf;
    P.Twl = d_Twl;
    P.thres = (float)thres_dist;
    if ((double)P.thres < thres_dist) P.thres = nextafterf(P.thres, INFINITY);
    P.thres_d = thres_dist;
    // queue storage is sliced by slot index (2 * MF entries per slot), the counter by lane
    P.hard_count = ctx->d_misc + 32 + ctx->cur;
    P.hard_list = ctx->hard_list + (size_t)first * ctx->MF * 2;
    P.hard_knn = ctx->hard_knn + (size_t)first * ctx->MF * 2 * 10;
    P.count = count;
    static const bool no_group = getenv("MML_NO_GROUP_SEARCH") != nullptr;  // A/B switch for measurements
    P.fresh_all = (count <= 8 && ctx->assoc_group_search && !no_group) ? 1 : 0;
    int* work_off = ctx->work_off + 2 * (size_t)first + ctx->cur;
    P.work_off = work_off;
    {
        MmlStageScope t(ctx, "associate");
        hipLaunchKernelGGL(k_assoc_prefix, dim3(1), dim3(AP_THREADS), 0, MML_STREAM(ctx), first, count, ctx->B, ctx->ft_n, work_off, P.hard_count);
        hipLaunchKernelGGL(k_associate, dim3(4096), dim3(128), 0, MML_STREAM(ctx), P);
    }
    {
        // (before the far-query fit: that one overwrites the parked records of its features with their factors)
        MmlStageScope t(ctx, "associate_fit");
        hipLaunchKernelGGL(k_associate_fit_all, dim3(4096), dim3(128), 0, MML_STREAM(ctx), P);
    }
    {
        MmlStageScope t(ctx, "associate_far");
        // (a whole / half a wavefront per query while all queries of the call -- every feature of up to 4 / 8 scans -- are resident
        //  at once: 2048 workgroups hold 8192 / 16384 of them)
        // lanes per far query of a batch: FOUR.  A batch has tens of thousands of far queries -- throughput, not the slowest query,
        // is what its kernel lasts -- and the lanes of a group split the (y, z) rows of a shell: 25 rows at ring 2 leave a 16-lane
        // group's second turn half empty.  Per 1024 scans (configs[1] / configs[3]): 64 lanes 0.331, 32: 0.224, 16: 0.164 / 0.118,
        // 8: 0.153, 4: 0.147 / 0.113, 2: 0.142 / 0.153, 1: 0.185 ms ($MML_HARD_SPAN: measurement switch).
        static int span_batch = -1;
        if (span_batch < 0) span_batch = getenv("MML_HARD_SPAN") ? atoi(getenv("MML_HARD_SPAN")) : 4;
        const int span = !P.fresh_all ? span_batch : (count <= 4 ? 64 : 32);
        auto launch_hard = [&](int round) {
            if (span == 64)
                hipLaunchKernelGGL(k_associate_hard<64>, dim3(2048), dim3(256), 0, MML_STREAM(ctx), P, round);
            else if (span == 32)
                hipLaunchKernelGGL(k_associate_hard<32>, dim3(2048), dim3(256), 0, MML_STREAM(ctx), P, round);
            else if (span == 8)
                hipLaunchKernelGGL(k_associate_hard<8>, dim3(1024), dim3(256), 0, MML_STREAM(ctx), P, round);
            else if (span == 4)
                hipLaunchKernelGGL(k_associate_hard<4>, dim3(1024), dim3(256), 0, MML_STREAM(ctx), P, round);
            else if (span == 2)
                hipLaunchKernelGGL(k_associate_hard<2>, dim3(1024), dim3(256), 0, MML_STREAM(ctx), P, round);
            else if (span == 1)
                hipLaunchKernelGGL(k_associate_hard<1>, dim3(1024), dim3(256), 0, MML_STREAM(ctx), P, round);
            else
                hipLaunchKernelGGL(k_associate_hard<16>, dim3(1024), dim3(256), 0, MML_STREAM(ctx), P, round);
        };
        launch_hard(0);
        hipLaunchKernelGGL(k_associate_fit, dim3(512), dim3(128), 0, MML_STREAM(ctx), P, 0);
        if (ctx->have_gmap[0] || ctx->have_gmap[1]) {  // features whose cube neighbourhood did not yield a model
            launch_hard(1);
            hipLaunchKernelGGL(k_associate_fit, dim3(512), dim3(128), 0, MML_STREAM(ctx), P, 1);
        }
    }
    for (int i = 0; i < count; ++i) ctx->stats_stale[first + i] = 1;
    MML_HIP(hipGetLastError());
    if (with_stats) return mml_ensure_assoc_stats(ctx, first, count);
    return MML_OK;
}

// n_line / n_plane / used counts / normal Gram matrix of the slots' current factors (checkLocalizability's input, Estimator.cpp:536-565;
// the counts k_linearize copies into its records), on the context's current stream, for the slots of the range whose factors changed
// since they were last computed.  mml_step does not ask for them (0.02 ms per 1024 scans of a kernel whose output it never reads).
int mml_ensure_assoc_stats(mml_ctx* ctx, int first, int count) {
    int lo = -1, hi = -1;
    for (int i = 0; i < count; ++i)
        if (ctx->stats_stale[first + i]) {
            if (lo < 0) lo = first + i;
            hi = first + i;
        }
    if (lo < 0) return MML_OK;
    MmlStageScope t(ctx, "assoc_stats");
    hipLaunchKernelGGL(k_assoc_stats, dim3(hi - lo + 1), dim3(256), 0, MML_STREAM(ctx), lo, ctx->B, ctx->MF, ctx->ft_n, ctx->lf, ctx->pf,
                       ctx->assoc_stats);
    MML_HIP(hipGetLastError());
    for (int b = lo; b <= hi; ++b) ctx->stats_stale[b] = 0;
    return MML_OK;
}
