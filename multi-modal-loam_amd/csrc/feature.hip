// feature.hip -- rows a1..a8 of SURVEY.md section 8: per-scan edge / planar feature extraction on gfx950.
//
// Reference: mm-loam/src/unionFeatureExtract.cpp (getVeloFeature :1113-1317, getHoriFeatureExtract :952-1035,
// detectFeaturePoints :341-844) and include/lidars_extrinsic_cali.h:424-477 (crop filters).
//
// Kernel chain for a batch of scan slots (one launch each, all slots at once):
//   k_assign_onepass (+ _ends, _tables) : ring / line id, in-scan time, crop, order-preserving bucketing by line in ONE pass over the
//                                    raw scan (ring layouts up to 32 rings; storage block-major, a line = segments: SegTab);
//                                    dense layouts: k_assign_a / _b / _c_staged (three passes, lines contiguous)
//   k_stencil (+ _redo, _break)    : one wavefront per scan line: curvature / depth / reflect stencil + every per-point predicate
//                                    of the flag state machine packed in 16 bits, the stride walk for 150
//   k_select_part / k_select       : the order-dependent part of the state machine (flags 3/1/2/300 by dependency steps on order
//                                    keys -- the partition sorts are never materialised) + label scatter
//   k_crop                         : label counts, corner / surf index lists, Livox extrinsic (the crop itself is
//                                    geometric and happens in the bucketing)
// Everything that is order-independent was hoisted into per-point predicates (k_stencil).
//
// Floating point: compiled with -ffp-contract=off; float expressions are written exactly as in the reference
// (left-to-right, float), double expressions follow Eigen's (x0+x1)+x2 reduction order.  libm calls on
// float arguments follow the "double libm, round on assignment" convention documented in DESIGN.md.
#include <math.h>

#include <algorithm>
#include <stdlib.h>

#include "libm_f32.h"
#include "mml_internal.h"

namespace {

// ---- attr bits written by k_stencil ---------------------------------------------------------------------
enum : unsigned {
    A_W2 = 1u << 0,        // thNumCurvSize == 2 at this point (:424-428)
    A_ANGLE = 1u << 1,     // cloudAngle[i] = 1 (:430-432)
    A_CAND3 = 1u << 2,     // curvature < flat threshold (:488)
    A_FAR = 1u << 3,       // depth > thDistanceFaraway (:499,512,524)
    A_REFL = 1u << 4,      // reflect corner candidate (:534-535)
    A_A3_SHIFT = 5,        // 2 bits: neighbours markable to the right with window 3 (:492-504)
    A_B3_SHIFT = 7,        // 2 bits: neighbours markable to the left (:505-517)
    A_LFLAT = 1u << 9,     // left half-window flat (:565)
    A_RFLAT = 1u << 10,    // right half-window flat (:597) -> stride 4
    A_C150 = 1u << 11,     // included-angle test passes (:644)
    A_F5_SHIFT = 12,       // 2 bits: 0 none, 1 -> flag 100, 2 -> flag 101 (:677-802)
    A_NEAR = 1u << 14,     // r^2 < thLidarNearestDis^2 (:824)
    A_VIS = 1u << 15       // visited by the stride-1-or-4 walk of :543-650 (k_stencil resolves the walk per line)
};

struct CropBlk;
struct FeatParams {
    int first, NV, NL, NT, L, n_rings, n_lines;
    float pitch0, pitch_step, near_th, far_th;
    const float4* velo_in;
    const mml_livox_point* livox_in;
    const int* n_in;
    uint8_t* raw_line;
    float* raw_ori;
    int sensor_base;     // pass C launched per sensor: blockIdx.z + sensor_base
    float4* ln_pts;
    int* ln_gidx;        // fused index of the bucketed point (>= 0 kept, -1 dropped, -2 Livox beyond far_th)
    int* ln_rel;         // bits of its in-sweep time
    int* line_start;
    int* line_len;
    float* ln_curv;
    float* ln_refl;
    uint16_t* ln_attr;
    int* blk_cnt;        // [B][2][nblk_max][BLK_STRIDE] per-block histograms / exclusive offsets
    int* assign_aux;     // AssignAux per slot
    int nblk_v, nblk_l, nblk_max, ring_bits, line_bits;
    int asb_stride;      // row stride of k_assign_b's LDS tile
    CropBlk* crop_cnt;   // [B][nblk_t] per-block counts / exclusive offsets of the crop passes
    unsigned* label_idx; // [B][2][cap] storage positions of the corner / surf labelled points
    int* label_gidx;     // [B][2][cap] ... and their fused indices (k_voxel's sort key: one gather per labelled point less)
    int label_cap;       // entries per list
    int nblk_t;
    unsigned* sel_scratch;  // global-memory scratch for lines longer than sel_cap: 4 x B*NT unsigned
    int sel_cap;            // points per line k_select keeps in LDS
    int line0;              // first scan line of this k_select launch (rings and Livox lines are launched separately)
    int B;
    uint16_t* ln_final;  // optional (detect_line): final CloudFeatureFlag per line point
    int* cb_n;
    uint8_t* ln_line;    // ring / Livox line of the points of an uploaded cloud
    uint8_t* ln_label;   // label of the bucketed point (kept points only)
    int* slot_flags;     // [2 b]: bit 0 uploaded, bit 1 undistorted
    int* fu_info;
    const float* extr;  // 16 floats or nullptr
    unsigned* brk_queue;  // [B][NT] line-bucketed positions whose break-point test needs the double-precision part
    int* brk_cnt;         // [B]
    unsigned* redo_queue;  // [B][NT] positions whose float pre-decisions were not certain (k_stencil_redo)
    int* redo_cnt;         // [B]
    uint8_t* sel_done;       // [B][L] 1: the line's flags and labels were written by k_select_part, k_select skips it
    int* sel_list;           // [2][B * L][2] (slot, line) of the lines k_select_part left to k_select: rings | Livox lines
    int* sel_list_cnt;       // [B][2] list lengths of the launch that starts at slot `first`
    unsigned char* st_exit;  // [B][st_stride] k_stencil, segment mode: the stride walk's exit offsets of every tile, for the four entries
    int st_stride;
    // storage segments of the lines (mml_internal.h): line index -> storage position
    int* seg_cum;
    int* seg_pos;
    int* seg_n;
    int* seg_flat;
    int* seg_flat_n;
    unsigned long long* op_agg;
    unsigned op_epoch;
    int4* seg_rs;   // per 64 line indices: (boundary, offset below it, offset from it on, 1 = no second boundary): rounds of k_stencil
    int4* seg_rw;   //   the same for aligned windows (k_select_part)
    int seg_rstride;
    int xcd_remap;  // k_assign_c_staged: slots dealt to the XCDs (measurement switch MML_XCD_REMAP=0)
    int ab_ppt;     // points per thread of pass A on the Velodyne part (dense layouts: 4, i.e. 1024-point count blocks)
    int ends_inline;  // k_assign_onepass computes the sweep's start / end azimuth itself (a handful of scans: one launch less)
};

// ---- line index -> storage position ---------------------------------------------------------------------------------------
// ln_curv / ln_refl / ln_attr are indexed in LINE ORDER (line_start[line] + i); the points themselves (ln_pts, ln_gidx, ln_rel,
// ln_label) sit where the bucketing put them: with the three-pass bucketing a line is one contiguous run, with the one-pass
// bucketing it is a run per 4096-point block of the raw scan.  Segment s of a line holds its indices [cum[s], cum[s + 1]).
struct SegTab {
    const int* cum;
    const int* pos;
    int nseg;
};
__device__ __forceinline__ SegTab seg_tab(const FeatParams& P, int b, int line) {
    const size_t o = (size_t)b * P.L + line;
    return SegTab{P.seg_cum + o * (MML_SEG_MAX + 1), P.seg_pos + o * MML_SEG_MAX, P.seg_n[o]};
}
// storage position (inside the slot) of line index i, 0 <= i < line length; `s` is a hint and is left at i's segment
__device__ __forceinline__ int seg_xlate(const SegTab& S, int i, int& s) {
    while (s + 1 < S.nseg && S.cum[s + 1] <= i) ++s;
    return S.pos[s] + (i - S.cum[s]);
}
__device__ __forceinline__ int seg_xlate(const SegTab& S, int i) {
    int s = 0;
    return seg_xlate(S, i, s);
}
// A translation record through the SCALAR data path (s_load_dwordx4 into four scalar registers).  The compiler only uses scalar
// loads for addresses it can prove uniform and memory it can prove unwritten; a record address derived from the wavefront's number
// (threadIdx.x >> 6) and read between the kernel's own stores is neither to it, and as a vector load the five records of a tile
// held twenty vector registers.  The wait is a separate statement that names the registers, so that no use can be moved above it.
typedef int seg_v4i __attribute__((ext_vector_type(4)));
// (a pointer the compiler takes for divergent -- it depends on the wavefront's number -- as the scalar it is)
__device__ __forceinline__ const int4* seg_uniform_ptr(const int4* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const int4*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ seg_v4i seg_sload(const int4* base /* from seg_uniform_ptr */, int index) {
    seg_v4i r;
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long q = ((unsigned long long)hi << 32) | lo;
    const int off = __builtin_amdgcn_readfirstlane(index * 16);
    asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(r) : "s"(q), "s"(off) : "memory");
    return r;
}
#define SEG_SWAIT5(a, b, c, d, e) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e)::"memory")
#define SEG_SWAIT8(a, b, c, d, e, f, g, h) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h)::"memory")

struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 d3(double x, double y, double z) { return D3{x, y, z}; }
__device__ __forceinline__ double ddot(const D3& a, const D3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double dnorm(const D3& a) { return sqrt(ddot(a, a)); }
__device__ __forceinline__ void dnormalize(D3& a) {
    double z = ddot(a, a);
    if (z > 0.0) {
        double n = sqrt(z);
        a.x /= n;
        a.y /= n;
        a.z /= n;
    }
}

constexpr int MAX_LINES = 160;  // n_rings + n_livox_lines upper bound
constexpr int BLK_STRIDE = MAX_LINES + 2;  // per-block record: key histogram | valid points | points kept by the crop

// ---- a1 / a2: ring / line assignment + order-preserving bucketing, three fully parallel passes ------------------
//   pass A (one point per lane) : validity, ring / line id, raw azimuth; per-256-point-block histograms
//   pass B (one workgroup / slot / sensor): start / end azimuth, the halfPassed flip index, exclusive scans of the
//                                  block histograms -> line offsets
//   pass C (one point per lane) : in-scan time, destination = line start + block offset + rank inside the block
// The rank of a point among the lanes of its wavefront that share its key comes from one ballot per key BIT
// (lanes with equal keys = AND over bits of the matching ballot), not from a loop over distinct keys.
constexpr int AB_THREADS = 256;
constexpr int AB_WAVES = AB_THREADS / 64;

// number of set bits of a wave mask below the calling lane (v_mbcnt: no lane mask to keep in registers)
__device__ __forceinline__ int lower_count(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// (the key of an invalid lane must be 0.  Per key bit: one compare for the ballot, the lane's bit spread to a word (v_bfe_i32),
//  and mask &= ~(ballot ^ spread) on the two halves -- six vector instructions; the obvious `m &= set ? bm : ~bm` compiled to
//  eleven, a quarter of the bucketing passes' instructions at 16 rings)
__device__ __forceinline__ unsigned long long match_key(bool valid, int key, int nbits) {
    const unsigned long long vm = __ballot(valid);
    unsigned mlo = (unsigned)vm, mhi = (unsigned)(vm >> 32);
    for (int bit = 0; bit < nbits; ++bit) {
        const int e = __builtin_amdgcn_sbfe(key, bit, 1);  // -1 where the bit is set
        const unsigned long long bm = __ballot(e != 0);
        mlo &= ~((unsigned)bm ^ (unsigned)e);
        mhi &= ~((unsigned)(bm >> 32) ^ (unsigned)e);
    }
    return ((unsigned long long)mhi << 32) | mlo;
}

// streaming loads of the bucketing passes: each raw record is touched once per pass and the batch is far larger than the caches
// (the scattered stores stay plain: with 128 rings every lane of a wavefront writes to a different line, and a non-temporal
//  16-byte store then reaches HBM alone -- 0.96 -> 2.68 ms at BASELINE configs[3])
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const v4f_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4f_nt*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

struct AssignAux {  // per slot, written by passes A / B
    int first_finite, last_finite, trig, kept_velo;
    float startOri, endOri;
    int kept_livox, pad;
};
// lidars_extrinsic_cali.h:424-477: removeNearFarPoints keeps near <= |p|^2 <= far, removeNearPointCloud only tests near
__device__ __forceinline__ void crop_test(const FeatParams& P, float x, float y, float z, bool& keep, bool& near_ok) {
    const float near2 = P.near_th * P.near_th, far2 = P.far_th * P.far_th;
    const float dis = x * x + y * y + z * z;
    keep = !(dis < near2 || dis > far2);
    near_ok = !(dis < near2);
}

__global__ void k_assign_init(FeatParams P, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {  // the launch's two lists of lines left to k_select start empty (k_select_list fills them)
        P.sel_list_cnt[2 * P.first] = 0;
        P.sel_list_cnt[2 * P.first + 1] = 0;
    }
    if (t < count) {
        AssignAux* a = reinterpret_cast<AssignAux*>(P.assign_aux) + (P.first + t);
        a->first_finite = 0x7fffffff;
        a->last_finite = -1;
        a->trig = 0x7fffffff;
        P.slot_flags[2 * (P.first + t)] = 0;  // an extracted cloud, not undistorted yet
        P.brk_cnt[P.first + t] = 0;   // the stencil's two queues start empty
        P.redo_cnt[P.first + t] = 0;
        int* info = P.fu_info + 8 * (size_t)(P.first + t);  // point counts by k_assign_b, label counters by the selection kernels
#pragma unroll
        for (int q = 0; q < 8; ++q) info[q] = 0;
    }
}

// getVeloFeature per-point part (:1133, :1154-1168) -- ring id through a float estimate of the pitch, the reference
// expression (glibc's atanf, a float division, sqrtf: the float overloads the reference TU resolves to, libm_f32.h) only when the
// estimate is within 2e-4 ring widths of a rounding boundary.
// (the estimate is built from the raw v_rsq / v_rcp instructions and an eight-term odd polynomial: |error| < 4e-7 rad in all,
//  2.3e-5 degrees; the reference's own float evaluation is within 1e-5 degrees of the true pitch: the guard band below covers the
//  sum 3.6 times over for any ring spacing; the IEEE sqrtf / two divisions / atanf it replaces are 70 vector instructions per point)
__device__ __forceinline__ float atan_est(float a) {
    const float aa = fabsf(a);
    const bool inv = aa > 1.0f;
    const float t = inv ? __builtin_amdgcn_rcpf(aa) : aa;
    const float z = t * t;
    float p = -0.004054564982652664f;  // least-squares fit of atan(t) / t on [0, 1], evaluated in float: 1.35e-7 rad
    p = __builtin_fmaf(p, z, 0.021862952038645744f);
    p = __builtin_fmaf(p, z, -0.0559123232960701f);
    p = __builtin_fmaf(p, z, 0.0964219719171524f);
    p = __builtin_fmaf(p, z, -0.1390862911939621f);
    p = __builtin_fmaf(p, z, 0.19946566224098206f);
    p = __builtin_fmaf(p, z, -0.33329859375953674f);
    p = __builtin_fmaf(p, z, 0.9999993443489075f);
    float r = p * t;
    r = inv ? 1.57079637f - r : r;
    return copysignf(r, a);
}
__device__ __forceinline__ int velo_ring(const float4 p, float pitch0, float pitch_step, int R) {
    const float inv_step = __builtin_amdgcn_rcpf(pitch_step);                  // (lane-uniform)
    const float guard = __builtin_fmaf(1.2e-4f, fabsf(inv_step), 2e-4f);      // ring widths: 2e-4 + 1.2e-4 degrees
    const float rr = __builtin_fmaf(p.x, p.x, p.y * p.y);
    const float est = atan_est(p.z * __builtin_amdgcn_rsqf(rr)) * 57.29577951308232f;
    const float t = __builtin_fmaf(est - pitch0, inv_step, 0.5f);
    int scanID;
    if (fabsf(t - rintf(t)) > guard && fabsf(t) < 1e6f) {
        scanID = (int)t;
    } else {
        // :1159 `atan(point.z / sqrt(point.x * point.x + point.y * point.y)) * 180 / M_PI` with the float overloads: the product
        // with 180 in float, the division by M_PI in double, rounded to float on assignment
        const float angle = (float)((double)(mml_libm::atanf_fd(p.z / sqrtf(p.x * p.x + p.y * p.y)) * 180.0f) / M_PI);
        // int() of a NaN -- the (0,0,0) no-return records of some drivers: atan(0 / 0) -- or of a value beyond the int range is
        // "integer indefinite" = INT_MIN on the reference's x86-64 (the point then fails the range test and is dropped), where
        // v_cvt_i32_f64 returns 0 / saturates: DESIGN.md parity convention 8
        const double v = (angle - pitch0) / pitch_step + 0.5;
        scanID = (v != v || v >= 2147483648.0 || v <= -2147483649.0) ? (-2147483647 - 1) : int(v);
    }
    return (scanID > (R - 1) || scanID < 0) ? 255 : scanID;
}

// `-atan2(point.y, point.x)` of :1168 (:1136-1139 for the sweep's ends): the float overload, i.e. glibc's atan2f, whose bits
// libm_f32.h reproduces -- two IEEE float divisions and 21 float multiply / adds per point.  (Rounds 1-5 took the call as the
// DOUBLE atan2 rounded to float -- a fast double form, a guard band, a queue of undecided points and a double-double kernel
// behind it --: one azimuth in six then differs by a float ulp from what the reference binary computes.  HISTORY.md A.000.)
__device__ __forceinline__ float neg_atan2f(float y, float x) { return -mml_libm::atan2f_fd(y, x); }
__device__ __forceinline__ void sweep_azimuths(const float4 p0, const float4 p1, float& startOri, float& endOri) {
    startOri = neg_atan2f(p0.y, p0.x);                                 // :1136
    endOri = (float)((double)neg_atan2f(p1.y, p1.x) + 2 * M_PI);       // :1137-1139: float + double, rounded on assignment
}

__global__ void k_libm_f32(const float* y, const float* x, long n, float* o2, float* o1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (o2) o2[i] = mml_libm::atan2f_fd(y[i], x[i]);
    if (o1) o1[i] = mml_libm::atanf_fd(y[i]);
}

template <int PPT>
__global__ __launch_bounds__(AB_THREADS) void k_assign_a(FeatParams P) {
    __shared__ int s_bcnt[MAX_LINES];
    __shared__ int s_valid, s_keep;
    const int b = blockIdx.y + P.first;
    const int sensor = blockIdx.z;  // 0 velodyne, 1 livox
    const int tid = threadIdx.x, lane = tid & 63;
    const int nblk = sensor == 0 ? P.nblk_v : P.nblk_l;
    if ((int)blockIdx.x >= nblk) return;
    // (PPT = P.ab_ppt as a template parameter, round 5: the thread's four records are requested together -- with the run-time
    //  trip count every round waited for its own load, one kilobyte in flight per wavefront -- and before the slot's point count
    //  is read, the index clamped to the slot's buffer: the block's first round trip to memory is the records themselves)
    float4 pv[PPT];
    if (sensor == 0 && P.NV > 0) {  // (a context without a Velodyne part, max_velo_points = 0, has no buffer to clamp into)
#pragma unroll
        for (int r = 0; r < PPT; ++r)
            pv[r] = nt_load4(P.velo_in + (size_t)b * P.NV + min((int)((blockIdx.x * PPT + r) * AB_THREADS + tid), P.NV - 1));
    }
    const int n = P.n_in[2 * b + sensor];
    const int nkeys = sensor == 0 ? P.n_rings : P.n_lines;
    const int nbits = sensor == 0 ? P.ring_bits : P.line_bits;
    static_assert(MAX_LINES <= AB_THREADS, "one key per thread");
    if (tid < nkeys) s_bcnt[tid] = 0;
    if (tid == 0) {
        s_valid = 0;
        s_keep = 0;
    }
    __syncthreads();
    // (dense layouts: four points per thread -- a 130-entry histogram record per 1024 points instead of per 256: the records were a
    //  quarter of this pass's written bytes and all of pass B's work)
    const int ppt = sensor == 0 ? PPT : 1;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
    if (r >= ppt) break;
    const int i = (blockIdx.x * ppt + r) * AB_THREADS + tid;
    bool valid = false, keep = false, near_ok = false;
    int key = 0;
    if (i < n) {
        if (sensor == 0) {
            const float4 p = pv[r];
            const bool fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
            int ring = 255;  // 255: non-finite (removed at :1133), 254: finite but outside the ring table (:1163-1166)
            float ori = 0.f;
            if (fin) {
                ring = velo_ring(p, P.pitch0, P.pitch_step, P.n_rings);
                if (ring == 255) ring = 254;
                ori = neg_atan2f(p.y, p.x);
            }
            P.raw_line[(size_t)b * P.NT + i] = (uint8_t)ring;
            P.raw_ori[(size_t)b * P.NV + i] = ori;
            valid = ring < 254;
            key = valid ? ring : 0;
            if (valid) crop_test(P, p.x, p.y, p.z, keep, near_ok);
        } else {
            const mml_livox_point p = P.livox_in[(size_t)b * P.NL + i];
            const int line_num = (int)p.line;
            valid = !(line_num > nkeys - 1) && !(p.x < 0.01);
            key = valid ? line_num : 0;
            P.raw_line[(size_t)b * P.NT + P.NV + i] = (uint8_t)(valid ? line_num : 255);
            if (valid) crop_test(P, p.x, p.y, p.z, keep, near_ok);
        }
    }
    // (one LDS add per valid lane: the adds of a wavefront that meet on a key are serialised by the LDS, a few cycles, where
    //  grouping the lanes by key first -- match_key, one add per group -- was a quarter of this pass's vector instructions)
    if (valid) atomicAdd(&s_bcnt[key], 1);
    const unsigned long long vm = __ballot(valid), km = __ballot(keep);
    if (lane == 0 && vm) atomicAdd(&s_valid, __popcll(vm));
    if (lane == 0 && km) atomicAdd(&s_keep, __popcll(km));
    __builtin_amdgcn_sched_barrier(0);  // (one point at a time: the evaluations' temporaries do not overlap)
    }
    __syncthreads();
    int* cnt = P.blk_cnt + ((size_t)(b * 2 + sensor) * P.nblk_max + blockIdx.x) * BLK_STRIDE;
    if (tid < nkeys) cnt[tid] = s_bcnt[tid];
    if (tid == 0) {
        cnt[MAX_LINES] = s_valid;
        cnt[MAX_LINES + 1] = s_keep;
    }
}

#ifndef MML_ASB_THREADS
#define MML_ASB_THREADS 512
#endif
constexpr int ASB_THREADS = MML_ASB_THREADS;  // one workgroup per (slot, sensor); 512: a 1024-thread workgroup waits long for 16
                                             // free wave slots on one CU while other lanes' kernels fill the device
__global__ __launch_bounds__(ASB_THREADS) void k_assign_b(FeatParams P) {
    __shared__ int s_trig, s_first, s_last;
    __shared__ int s_tot[MAX_LINES + 2];
    const int b = blockIdx.x + P.first;
    const int sensor = blockIdx.y;
    const int tid = threadIdx.x;
    const int n = P.n_in[2 * b + sensor];
    const int ab_pts = AB_THREADS * (sensor == 0 ? P.ab_ppt : 1);
    const int nblk = (n + ab_pts - 1) / ab_pts;
    const int nkeys = sensor == 0 ? P.n_rings : P.n_lines;
    AssignAux* a = reinterpret_cast<AssignAux*>(P.assign_aux) + b;
    int* cnt0 = P.blk_cnt + ((size_t)(b * 2 + sensor) * P.nblk_max) * BLK_STRIDE;
    if (tid == 0) {
        s_trig = 0x7fffffff;
        s_first = 0x7fffffff;
        s_last = -1;
    }
    __syncthreads();
    // exclusive scan over blocks, per key (+ the valid-point and kept-point counts).  The block records are rows of
    // BLK_STRIDE ints: a tile of 64 rows is brought into LDS with coalesced loads, scanned there -- one wavefront per key,
    // one lane per row -- and written back the same way (lane-per-row accesses straight to memory touched one 64-byte sector
    // per int: 3.9 ms per 1024 dense 128-ring scans).
    {
        // (the tile is sized by the launch for the sensor's real number of columns, compact: 64 x 19 ints at 16 rings instead
        //  of 64 x 163 -- with 42 KB of LDS per workgroup this one-workgroup-per-slot kernel waited ~0.5 ms per 512-slot launch
        //  for room next to the other lanes' kernels, and its lane's scatter pass with it)
        extern __shared__ int s_tile_dyn[];
        __shared__ int s_acc[MAX_LINES + 2];
        const int lane = tid & 63;
        const int ncol = nkeys + 2;
        const int tstride = P.asb_stride;  // >= max(n_rings, n_lines) + 2, odd
        for (int k = tid; k < ncol; k += ASB_THREADS) s_acc[k] = 0;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int rows = min(64, nblk - b0);
            __syncthreads();
            for (int idx = tid; idx < rows * ncol; idx += ASB_THREADS) {
                const int r = idx / ncol, c = idx - r * ncol;
                const int col = c < nkeys ? c : MAX_LINES + (c - nkeys);
                s_tile_dyn[r * tstride + c] = cnt0[(size_t)(b0 + r) * BLK_STRIDE + col];
            }
            __syncthreads();
            for (int kk = tid >> 6; kk < ncol; kk += ASB_THREADS / 64) {
                const int base = s_acc[kk];
                const int v = lane < rows ? s_tile_dyn[lane * tstride + kk] : 0;
                int x = v;
                for (int o = 1; o < 64; o <<= 1) {
                    const int y = __shfl_up(x, o);
                    if (lane >= o) x += y;
                }
                if (lane < rows) s_tile_dyn[lane * tstride + kk] = base + x - v;
                const int tot = __shfl(x, 63);
                if (lane == 0) s_acc[kk] = base + tot;
            }
            __syncthreads();
            for (int idx = tid; idx < rows * ncol; idx += ASB_THREADS) {
                const int r = idx / ncol, c = idx - r * ncol;
                const int col = c < nkeys ? c : MAX_LINES + (c - nkeys);
                cnt0[(size_t)(b0 + r) * BLK_STRIDE + col] = s_tile_dyn[r * tstride + c];
            }
        }
        __syncthreads();
        for (int k = tid; k < ncol; k += ASB_THREADS) s_tot[k] = s_acc[k];
    }
    __syncthreads();
    if (sensor == 0) {
        const uint8_t* rl = P.raw_line + (size_t)b * P.NT;
        // first / last finite point (:1136-1139): strided search, min / max reduction
        int ff = 0x7fffffff, lf = -1;
        for (int i = tid; i < n; i += ASB_THREADS)
            if (rl[i] != 255) {
                ff = i;
                break;
            }
        for (int i = n - 1 - tid; i >= 0; i -= ASB_THREADS)
            if (rl[i] != 255) {
                lf = i;
                break;
            }
        for (int o = 32; o > 0; o >>= 1) {
            ff = min(ff, __shfl_xor(ff, o));
            lf = max(lf, __shfl_xor(lf, o));
        }
        if ((tid & 63) == 0) {
            if (ff != 0x7fffffff) atomicMin(&s_first, ff);
            if (lf >= 0) atomicMax(&s_last, lf);
        }
        __syncthreads();
        float startOri = 0.f, endOri = 0.f;
        if (s_last >= 0) {
            const float4 p0 = P.velo_in[(size_t)b * P.NV + s_first];
            const float4 p1 = P.velo_in[(size_t)b * P.NV + s_last];
            sweep_azimuths(p0, p1, startOri, endOri);
            if (endOri - startOri > 3 * M_PI)
                endOri -= 2 * M_PI;
            else if (endOri - startOri < M_PI)
                endOri += 2 * M_PI;
        }
        // index of the point that sets halfPassed (:1169-1177): a prefix-OR, found with a min-reduction
        const uint8_t* rline = P.raw_line + (size_t)b * P.NT;
        const float* rori = P.raw_ori + (size_t)b * P.NV;
        int best = 0x7fffffff;
        // eight points of this thread's stride per round, their loads in flight together (the first hit sits half-way
        // through the sweep: one point per round is a dozen dependent memory round trips)
        for (int i0 = tid; i0 < n && best == 0x7fffffff; i0 += 8 * ASB_THREADS) {
            uint8_t rl8[8];
            float ro8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * ASB_THREADS, n - 1);
                rl8[u] = rline[i];
                ro8[u] = rori[i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * ASB_THREADS;
                if (i >= n || rl8[u] >= 254 || best != 0x7fffffff) continue;
                float ori = ro8[u];
                if (ori < startOri - M_PI / 2)
                    ori += 2 * M_PI;
                else if (ori > startOri + M_PI * 3 / 2)
                    ori -= 2 * M_PI;
                if (ori - startOri > M_PI) best = i;  // indices grow along this thread's stride
            }
        }
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
        if ((tid & 63) == 0 && best != 0x7fffffff) atomicMin(&s_trig, best);
        __syncthreads();
        if (tid == 0) {
            a->startOri = startOri;
            a->endOri = endOri;
            a->trig = s_trig;
        }
    }
    if (tid == 0) {
        int acc = sensor == 0 ? 0 : P.NV;  // livox lines live behind the velodyne region
        int* ls = P.line_start + (size_t)b * P.L + (sensor == 0 ? 0 : P.n_rings);
        int* ll = P.line_len + (size_t)b * P.L + (sensor == 0 ? 0 : P.n_rings);
        const size_t lo = (size_t)b * P.L + (sensor == 0 ? 0 : P.n_rings);
        int* flat = P.seg_flat + ((size_t)b * 2 + sensor) * MML_SEG_FLAT;
        for (int r = 0; r < nkeys; ++r) {
            ls[r] = acc;
            ll[r] = s_tot[r];
            // one storage segment per line: the line is contiguous, stored where its line-order index says
            P.seg_n[lo + r] = 1;
            P.seg_cum[(lo + r) * (MML_SEG_MAX + 1)] = 0;
            P.seg_cum[(lo + r) * (MML_SEG_MAX + 1) + 1] = s_tot[r];
            P.seg_pos[(lo + r) * MML_SEG_MAX] = acc;
            if (r < MML_SEG_FLAT) flat[r] = acc;
            acc += s_tot[r];
        }
        P.seg_flat_n[((size_t)b * 2 + sensor) * 2] = nkeys < MML_SEG_FLAT ? nkeys : MML_SEG_FLAT;
        P.seg_flat_n[((size_t)b * 2 + sensor) * 2 + 1] = nkeys;
        P.cb_n[2 * b + sensor] = s_tot[nkeys];
        if (sensor == 0)
            a->kept_velo = s_tot[nkeys + 1];
        else
            a->kept_livox = s_tot[nkeys + 1];
        int* info = P.fu_info + 8 * (size_t)b;  // (zeroed by k_assign_init)
        atomicAdd(&info[0], s_tot[nkeys + 1]);
        if (sensor == 0) info[1] = s_tot[nkeys + 1];
    }
}

__device__ __forceinline__ double livox_to_sec(uint32_t t) {  // ros::Time().fromNSec(t).toSec()
    uint32_t sec = (uint32_t)(t / 1000000000ull);
    uint32_t nsec = (uint32_t)(t % 1000000000ull);
    return (double)sec + 1e-9 * (double)nsec;
}

// pass C, lane-by-lane scatter: for sensors with few lines (16 rings: four consecutive raw points share a line, the
// pieces of a line written by neighbouring lanes are merged in the L2)
__global__ __launch_bounds__(AB_THREADS) void k_assign_c_direct(FeatParams P) {
    __shared__ int s_wcnt[AB_WAVES][MAX_LINES];
    __shared__ int s_wvalid[AB_WAVES];
    const int b = blockIdx.y + P.first;
    const int sensor = blockIdx.z + P.sensor_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n_in[2 * b + sensor];
    const int i = blockIdx.x * AB_THREADS + tid;
    if ((int)(blockIdx.x * AB_THREADS) >= n) return;
    const int nkeys = sensor == 0 ? P.n_rings : P.n_lines;
    const int nbits = sensor == 0 ? P.ring_bits : P.line_bits;
    for (int k = lane; k < nkeys; k += 64) s_wcnt[wave][k] = 0;  // (a wavefront's own row)
    // the block's offsets, the line starts and the per-slot constants are requested here, next to the points themselves: a
    // lane's destination is then two LDS look-ups behind its key instead of two more dependent trips to memory (key ->
    // block offset -> line start), which is what bounded the pass -- its lanes live as long as their longest load chain
    __shared__ int s_cnt[MAX_LINES + 2];
    __shared__ int s_ls[MAX_LINES];
    const int* cnt = P.blk_cnt + ((size_t)(b * 2 + sensor) * P.nblk_max + blockIdx.x) * BLK_STRIDE;
    if (tid < nkeys) {
        s_cnt[tid] = cnt[tid];
        s_ls[tid] = P.line_start[(size_t)b * P.L + (sensor == 0 ? 0 : P.n_rings) + tid];
    }
    if (tid < 2) s_cnt[MAX_LINES + tid] = cnt[MAX_LINES + tid];
    const AssignAux aux = *(reinterpret_cast<const AssignAux*>(P.assign_aux) + b);
    // ... and so are the key and the record of the lane's own point (the record whether or not the key will call it valid:
    // the few invalid ones cost nothing, and the load no longer waits for the key)
    const int region = sensor == 0 ? 0 : P.NV;
    int key = 255;
    float4 praw = make_float4(0.f, 0.f, 0.f, 0.f);
    mml_livox_point q;
    q.x = q.y = q.z = 0.f;
    q.reflectivity = 0;
    q.offset_time = 0;
    if (i < n) {
        key = P.raw_line[(size_t)b * P.NT + region + i];
        if (sensor == 0)
            praw = nt_load4(P.velo_in + (size_t)b * P.NV + i);
        else
            q = P.livox_in[(size_t)b * P.NL + i];
    }
    __syncthreads();
    const bool valid = key < 254;
    if (!valid) key = 0;
    // the crop decision is geometric (lidars_extrinsic_cali.h:424-477), so the position of a point in the fused cloud
    // [velo kept ; livox kept] is known before any label is: the point goes straight there
    bool keep = false, near_ok = false;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t off_time = 0;
    if (valid) {
        if (sensor == 0) {
            out = make_float4(praw.x, praw.y, praw.z, 0.f);  // intensity zeroed, :1254-1256
        } else {
            out = make_float4(q.x, q.y, q.z, (float)q.reflectivity);
            praw = out;
            off_time = q.offset_time;
        }
        crop_test(P, out.x, out.y, out.z, keep, near_ok);
    }
    const unsigned long long eq = match_key(valid, key, nbits);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long km = __ballot(keep);
    if (valid && (eq & lt) == 0) s_wcnt[wave][key] = __popcll(eq);
    if (lane == 0) s_wvalid[wave] = __popcll(km);
    __syncthreads();
    if (!valid) return;
    const AssignAux* a = &aux;
    int pos = s_cnt[key] + __popcll(eq & lt);
    int fdst = (sensor == 0 ? 0 : a->kept_velo) + s_cnt[MAX_LINES + 1] + __popcll(km & lt);
    for (int w = 0; w < wave; ++w) {
        pos += s_wcnt[w][key];
        fdst += s_wvalid[w];
    }
    const int dst = s_ls[key] + pos;
    const size_t g = (size_t)b * P.NT + dst;
    P.ln_pts[g] = praw;  // (plain stores: neighbouring lanes' pieces of a line are merged in the L2 before they reach HBM)
    float rel;  // (also for the few points the crop drops: the undistortion runs over the whole region)
    if (sensor == 0) {
        const float startOri = a->startOri, endOri = a->endOri;
        float ori = P.raw_ori[(size_t)b * P.NV + i];
        if (i <= a->trig) {  // :1169-1177
            if (ori < startOri - M_PI / 2)
                ori += 2 * M_PI;
            else if (ori > startOri + M_PI * 3 / 2)
                ori -= 2 * M_PI;
        } else {  // :1178-1184
            ori += 2 * M_PI;
            if (ori < endOri - M_PI * 3 / 2)
                ori += 2 * M_PI;
            else if (ori > endOri + M_PI / 2)
                ori -= 2 * M_PI;
        }
        rel = (ori - startOri) / (endOri - startOri);  // :1186
    } else {
        const mml_livox_point* in = P.livox_in + (size_t)b * P.NL;
        const double timeSpan = livox_to_sec(in[n - 1].offset_time);  // :985
        rel = livox_to_sec(off_time) / timeSpan;                       // :995
    }
    // one 8-byte record per point: its index in the fused cloud (or -2 for a Livox point that only fails the far test --
    // its label still counts towards livox_corner_num / livox_surf_num, :925-940 -- or -1) and its in-sweep time.  The
    // label byte is written by k_select for every point of a line, the line id follows from the line table: two
    // scattered stores per point in all (the pass is bound by their number, not by their bytes).
    P.ln_gidx[g] = keep ? fdst : ((sensor == 1 && near_ok) ? -2 : -1);
    P.ln_rel[g] = __float_as_int(rel);
}

// Pass C for sensors with many lines (> 32): CB_SUB pass-A blocks per workgroup, the records staged in LDS in (line, rank)
// order before they leave:
// a dense scan has 128 rings and consecutive raw points belong to different rings, so a lane-by-lane scatter writes 16 isolated
// bytes per lane (1.6 TB/s at BASELINE configs[3]); out of the staged order consecutive lanes write consecutive records of a
// line -- runs of (points per line per workgroup) x 16 bytes.  A workgroup stages 2048 points, two per thread (round 3: 1024 --
// 8-point runs of 128 / 32 / 32 bytes per line and array; now 16-point runs), and the workgroups of one scan run on ONE XCD.
#ifndef MML_CB_TP
#define MML_CB_TP 2
#endif
constexpr int CB_TP = MML_CB_TP;                // points per thread (measurement switch: 4 = 4096-point tiles, one workgroup per CU)
constexpr int CB_THREADS = 1024;
constexpr int CB_TILE = CB_THREADS * CB_TP;     // points per workgroup
constexpr int CB_SUB = CB_TILE / AB_THREADS;    // pass-A blocks per workgroup at most (256-point blocks)
constexpr int CB_WAVES = CB_THREADS / 64;
constexpr int CB_GROUPS = CB_WAVES * CB_TP;     // (round, wavefront) groups in raw order
__global__ __launch_bounds__(CB_THREADS) void k_assign_c_staged(FeatParams P) {
    __shared__ unsigned char s_wcnt[CB_GROUPS][MAX_LINES];  // points per (group, line): at most 64
    __shared__ unsigned char s_wvalid[CB_GROUPS];
    __shared__ int s_cnt[CB_SUB][MAX_LINES + 2];
    __shared__ int s_ls[MAX_LINES];
    __shared__ int s_kstart[MAX_LINES + 1];  // where the run of each line starts in the staged order; [nkeys] = valid points
    __shared__ float4 s_pt[CB_TILE];
    __shared__ int2 s_meta[CB_TILE];
    __shared__ int s_dst[CB_TILE];
    // Workgroups are dealt to the eight XCDs round-robin in dispatch order, and every XCD has its own L2: with the plain
    // (tile, slot) = blockIdx mapping the tiles of one scan -- whose runs of a line are neighbours in memory, a run of 16 points
    // ends in the middle of a 128-byte line of the 4-byte arrays -- land on eight different L2s, none of which ever sees a whole
    // line.  Remapped: XCD x owns the slots x, x + 8, ... of the launch and walks their tiles in order, so the pieces of a line meet
    // in ONE L2 and leave it merged (configs[3]: 4.42 -> 4.01 ms per 1024 scans with the 1024-point tiles).
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd_remap && (gridDim.y & 7) == 0) {
        const int g = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = g & 7, idx = g >> 3;
        by = (idx / (int)gridDim.x) * 8 + xcd;
        bx = idx % (int)gridDim.x;
    }
    const int b = by + P.first;
    const int sensor = blockIdx.z + P.sensor_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the keys and the records of the lane's own points are requested first (the record whether or not the key will call it valid:
    // the few invalid ones cost nothing, and the load does not wait for the key) -- for the Velodyne part before the slot's point
    // count is read, the index clamped to the slot's buffer: the count decides below which lanes hold a point
    const int region = sensor == 0 ? 0 : P.NV;
    int key[CB_TP];
    float4 praw[CB_TP];
    float ori[CB_TP];
    uint32_t off_time[CB_TP];
    double timeSpan = 1.0;
    if (sensor == 0 && P.NV > 0) {
#pragma unroll
        for (int r = 0; r < CB_TP; ++r) {
            const int ic = min((int)(bx * CB_TILE + r * CB_THREADS + tid), P.NV - 1);
            key[r] = P.raw_line[(size_t)b * P.NT + ic];
            praw[r] = nt_load4(P.velo_in + (size_t)b * P.NV + ic);
            ori[r] = P.raw_ori[(size_t)b * P.NV + ic];
            off_time[r] = 0;
        }
    }
    const int n = P.n_in[2 * b + sensor];
    if ((int)(bx * CB_TILE) >= n) return;
    const int nkeys = sensor == 0 ? P.n_rings : P.n_lines;
    const int nbits = sensor == 0 ? P.ring_bits : P.line_bits;
    const int ab_ppt = sensor == 0 ? P.ab_ppt : 1;
    const int ab_pts = AB_THREADS * ab_ppt;          // points per pass-A block
    const int sub_n = CB_TILE / ab_pts;              // pass-A blocks per workgroup
    const int gpb = AB_WAVES * ab_ppt;               // (round, wavefront) groups per pass-A block
    const int nblk = (n + ab_pts - 1) / ab_pts;
    for (int k = tid; k < CB_GROUPS * MAX_LINES / 4; k += CB_THREADS) reinterpret_cast<unsigned*>(&s_wcnt[0][0])[k] = 0;
    static_assert((CB_GROUPS * MAX_LINES) % 4 == 0, "cleared as words");
    // the blocks' offsets, the line starts and the per-slot constants are requested here, next to the points themselves: a
    // lane's destination is then two LDS look-ups behind its key instead of two more dependent trips to memory (key ->
    // block offset -> line start) -- the lanes of this pass live as long as their longest load chain
    const int* cnt0 = P.blk_cnt + ((size_t)(b * 2 + sensor) * P.nblk_max + (size_t)bx * sub_n) * BLK_STRIDE;
    for (int k = tid; k < sub_n * nkeys; k += CB_THREADS) {
        const int u = k / nkeys, kk = k - u * nkeys;
        s_cnt[u][kk] = (bx * sub_n + u < nblk) ? cnt0[(size_t)u * BLK_STRIDE + kk] : 0;
    }
    for (int k = tid; k < nkeys; k += CB_THREADS) s_ls[k] = P.line_start[(size_t)b * P.L + (sensor == 0 ? 0 : P.n_rings) + k];
    if (tid < sub_n) s_cnt[tid][MAX_LINES + 1] = (bx * sub_n + tid < nblk) ? cnt0[(size_t)tid * BLK_STRIDE + MAX_LINES + 1] : 0;
    const AssignAux aux = *(reinterpret_cast<const AssignAux*>(P.assign_aux) + b);
#pragma unroll
    for (int r = 0; r < CB_TP; ++r) {
        const int i = bx * CB_TILE + r * CB_THREADS + tid;
        if (sensor == 0) {
            if (i >= n) key[r] = 255;
        } else {
            key[r] = 255;
            praw[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            ori[r] = 0.f;
            off_time[r] = 0;
            if (i < n) {
                key[r] = P.raw_line[(size_t)b * P.NT + region + i];
                const mml_livox_point q = P.livox_in[(size_t)b * P.NL + i];
                praw[r] = make_float4(q.x, q.y, q.z, (float)q.reflectivity);
                off_time[r] = q.offset_time;
            }
        }
    }
    if (sensor == 1) timeSpan = livox_to_sec(P.livox_in[(size_t)b * P.NL + n - 1].offset_time);  // :985
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    bool valid[CB_TP], keep[CB_TP], near_ok[CB_TP];
    int rank[CB_TP], krank[CB_TP];
#pragma unroll
    for (int r = 0; r < CB_TP; ++r) {
        valid[r] = key[r] < 254;
        if (!valid[r]) key[r] = 0;
        // the crop decision is geometric (lidars_extrinsic_cali.h:424-477), so the position of a point in the fused cloud
        // [velo kept ; livox kept] is known before any label is: the point goes straight there
        keep[r] = false;
        near_ok[r] = false;
        if (valid[r]) crop_test(P, praw[r].x, praw[r].y, praw[r].z, keep[r], near_ok[r]);
        const unsigned long long eq = match_key(valid[r], key[r], nbits);
        const unsigned long long km = __ballot(keep[r]);
        const int g = r * CB_WAVES + wave;
        if (valid[r] && (eq & lt) == 0) s_wcnt[g][key[r]] = (unsigned char)__popcll(eq);
        if (lane == 0) s_wvalid[g] = (unsigned char)__popcll(km);
        rank[r] = __popcll(eq & lt);
        krank[r] = __popcll(km & lt);
    }
    __syncthreads();
    // staged order: lines ascending, inside a line the raw order.  The first wavefront scans the workgroup's line counts.
    if (wave == 0) {
        int carry = 0;
        for (int k0 = 0; k0 < nkeys; k0 += 64) {
            const int k = k0 + lane;
            int c = 0;
            if (k < nkeys)
                for (int w = 0; w < CB_GROUPS; ++w) c += s_wcnt[w][k];
            int x = c;
            for (int o = 1; o < 64; o <<= 1) {
                const int y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            if (k < nkeys) s_kstart[k] = carry + x - c;
            carry += __shfl(x, 63);
        }
        if (lane == 0) s_kstart[nkeys] = carry;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < CB_TP; ++r) {
        if (!valid[r]) continue;
        const int i = bx * CB_TILE + r * CB_THREADS + tid;
        const AssignAux* a = &aux;
        const int g = r * CB_WAVES + wave;       // my (round, wavefront) group; its pass-A block holds gpb groups
        const int sub = g / gpb;
        int pos = s_cnt[sub][key[r]] + rank[r];
        int fdst = (sensor == 0 ? 0 : a->kept_velo) + s_cnt[sub][MAX_LINES + 1] + krank[r];
        for (int w = sub * gpb; w < g; ++w) {
            pos += s_wcnt[w][key[r]];
            fdst += s_wvalid[w];
        }
        float rel;  // (also for the few points the crop drops: the undistortion runs over the whole region)
        if (sensor == 0) {
            const float startOri = a->startOri, endOri = a->endOri;
            float o = ori[r];
            if (i <= a->trig) {  // :1169-1177
                if (o < startOri - M_PI / 2)
                    o += 2 * M_PI;
                else if (o > startOri + M_PI * 3 / 2)
                    o -= 2 * M_PI;
            } else {  // :1178-1184
                o += 2 * M_PI;
                if (o < endOri - M_PI * 3 / 2)
                    o += 2 * M_PI;
                else if (o > endOri + M_PI / 2)
                    o -= 2 * M_PI;
            }
            rel = (o - startOri) / (endOri - startOri);  // :1186
        } else {
            rel = livox_to_sec(off_time[r]) / timeSpan;   // :995
        }
        // the point's rank inside its line, counted from the first of this workgroup's pass-A blocks
        const int e = s_kstart[key[r]] + (pos - s_cnt[0][key[r]]);
        s_pt[e] = praw[r];
        // its index in the fused cloud (or -2 for a Livox point that only fails the far test -- its label still counts towards
        // livox_corner_num / livox_surf_num, :925-940 -- or -1) and its in-sweep time.  The label byte is written by k_select for
        // every point of a line, the line id follows from the line table.
        s_meta[e] = make_int2(keep[r] ? fdst : ((sensor == 1 && near_ok[r]) ? -2 : -1), __float_as_int(rel));
        s_dst[e] = s_ls[key[r]] + pos;
    }
    __syncthreads();
    const int nstaged = s_kstart[nkeys];
    for (int e = tid; e < nstaged; e += CB_THREADS) {
        const size_t g = (size_t)b * P.NT + s_dst[e];
        P.ln_pts[g] = s_pt[e];
        P.ln_gidx[g] = s_meta[e].x;
        P.ln_rel[g] = s_meta[e].y;
    }
}

// ---- a1 / a2 in ONE pass over the raw scan (ring layouts up to 32 rings, scans up to MML_SEG_MAX x 4096 points per sensor) -----
// The three passes above read every raw record twice and carry ring id and azimuth from pass A to pass C through memory: 66 bytes
// of traffic per point for the 40 the job needs, both passes at the HBM roof.  Here a 512-thread workgroup owns a block of
// MML_OP_BLK = 4096 consecutive raw points, eight per thread, which stay IN REGISTERS between being counted and being stored
// (sixteen per thread: 207 registers, two wavefronts per SIMD):
//   1. load; ring / line id, azimuth, crop test, the half-turn condition of :1169-1177 per point; the rank of a point among the
//      points of its line inside its wavefront round (ballots), per (round, wavefront) group counts in LDS
//   2. exclusive scan of the 64 group counts per line -> the block's line histogram and the offset of every group
//   3. the block publishes (valid points, kept points, first half-turn index) as ONE 64-bit word tagged with the launch's epoch and
//      reads the words of the blocks in front of it (decoupled look-back: a scan has at most 16 blocks, one lane per
//      predecessor, nobody waits for a successor) -> where the block's points start in the slot, where its kept points start
//      in the fused cloud, whether the half turn was passed before it
//   4. the points leave for  region + (valid points of the blocks before) + (offset of the line inside the block) + rank.
// The storage order is therefore BLOCK-major, line-bucketed inside a block: a line is a sequence of segments, one per block
// (k_assign_tables lists them: seg_cum / seg_pos), in raw order -- which is all the per-line kernels need (seg_xlate).  Nothing
// is written but the 16-byte point, the 8-byte (fused index, time) record and 2 x 32 ints per block: 42 bytes per point.
constexpr int OP_THREADS = 512, OP_PPT = MML_OP_BLK / OP_THREADS, OP_WAVES = OP_THREADS / 64, OP_GROUPS = OP_PPT * OP_WAVES;
constexpr int OP_MAXKEYS = 32, OP_CSTRIDE = OP_MAXKEYS + 1, OP_REC_POS = 80;
static_assert(OP_PPT * OP_THREADS == MML_OP_BLK, "block size");
static_assert(OP_GROUPS <= 64, "one lane per (round, wavefront) group in the scans");
static_assert(OP_REC_POS + OP_MAXKEYS <= MAX_LINES, "the block record holds histogram and positions in front of the two counts");
constexpr unsigned OP_NONE = 0x1fffu;
enum : unsigned { OPI_VALID = 1u << 8, OPI_KEEP = 1u << 9, OPI_NEAR = 1u << 10, OPI_RANK_SHIFT = 12, OPI_KRANK_SHIFT = 20 };

// start / end azimuth of the sweep (:1136-1146), by one wavefront: first and last finite record of the Velodyne part (normally
// the first and the last one: two requests), then the two arctangents on one lane.  Every block of the one-pass kernel does this
// for itself -- eight times the same few hundred instructions per scan cost less than a kernel of their own in front of it.
__device__ __forceinline__ void sweep_ends(const FeatParams& P, int b, float& startOri, float& endOri) {
    const int lane = threadIdx.x & 63;
    const int n = P.n_in[2 * b];
    const float4* in = P.velo_in + (size_t)b * P.NV;
    int ff = -1, lf = -1;
    for (int i0 = 0; i0 < n && ff < 0; i0 += 64) {
        const int i = i0 + lane;
        bool fin = false;
        if (i < n) {
            const float4 p = in[i];
            fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
        }
        const unsigned long long m = __ballot(fin);
        if (m) ff = i0 + (int)__ffsll((long long)m) - 1;
    }
    for (int i0 = n > 0 ? ((n - 1) / 64) * 64 : -1; i0 >= 0 && lf < 0; i0 -= 64) {
        const int i = i0 + lane;
        bool fin = false;
        if (i < n) {
            const float4 p = in[i];
            fin = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
        }
        const unsigned long long m = __ballot(fin);
        if (m) lf = i0 + 63 - __clzll((long long)m);
    }
    startOri = 0.f;
    endOri = 0.f;
    if (lf >= 0) {
        const float4 p0 = in[ff], p1 = in[lf];
        sweep_azimuths(p0, p1, startOri, endOri);
        if (endOri - startOri > 3 * M_PI)
            endOri -= 2 * M_PI;
        else if (endOri - startOri < M_PI)
            endOri += 2 * M_PI;
    }
}

// ... as a kernel of its own for batches, one wavefront per slot (the blocks then read the two floats instead of computing them:
// the two arctangents on one lane sit in front of every block's first barrier, +12 % on the pass at 1024 scans)
__global__ __launch_bounds__(64) void k_assign_ends(FeatParams P, int count) {
    const int t = blockIdx.x;
    if (t >= count) return;
    const int b = P.first + t;
    float so, eo;
    sweep_ends(P, b, so, eo);
    if (threadIdx.x == 0) {
        AssignAux* a = reinterpret_cast<AssignAux*>(P.assign_aux) + b;
        a->startOri = so;
        a->endOri = eo;
    }
}

// inclusive scan over the 64 lanes of a wavefront
__device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(v, o);
        if (lane >= o) v += y;
    }
    return v;
}

#ifdef MML_OP_TIMING
// phase clocks of one bucketing block (the Velodyne block MML_OP_TIMING of the 518th slot of the launch; tools/assign_phases.py):
// [0..9] its second wavefront, [10..12] the look-back section of the first one
__device__ unsigned long long g_op_dbg[16];
#define OP_MARK_(flag, prev, id)                       \
    do {                                               \
        if (flag) {                                    \
            const unsigned long long now_ = clock64(); \
            g_op_dbg[id] += now_ - prev;               \
            prev = now_;                               \
        }                                              \
    } while (0)
#define OP_MARK(id) OP_MARK_(op_dbg, op_prev, id)
#define OP_MARK0(id) OP_MARK_(op_dbg0, op_prev0, id)
extern "C" int mml_debug_op_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_op_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_op_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#else
#define OP_MARK(id)
#define OP_MARK0(id)
#endif
struct OnepassLds {
    int cnt[OP_GROUPS][OP_CSTRIDE];  // points per (group, line); after the scan: points of the line in the groups before
    int gv[OP_GROUPS], gk[OP_GROUPS];  // valid / kept points per group -> exclusive over the groups
    int hist[OP_MAXKEYS + 2], koff[OP_MAXKEYS];
    int cond[OP_WAVES];
    int base[3];
    float ori[2];
    int out[2][MML_OP_BLK];  // fused index | in-sweep time of the block's points in storage order
};
template <int SENSOR>
__device__ __forceinline__ void assign_onepass_body(const FeatParams& P, OnepassLds& S) {
    auto& s_cnt = S.cnt;
    auto& s_gv = S.gv;
    auto& s_gk = S.gk;
    auto& s_hist = S.hist;
    auto& s_koff = S.koff;
    auto& s_cond = S.cond;
    auto& s_base = S.base;
    auto& s_out = S.out;
    const int b = blockIdx.y + P.first, blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blk * MML_OP_BLK;
#ifdef MML_OP_TIMING
    const bool op_dbg = SENSOR == 0 && tid == 64 && blk == MML_OP_TIMING && b == P.first + 517;
    const bool op_dbg0 = SENSOR == 0 && tid == 0 && blk == MML_OP_TIMING && b == P.first + 517;
    unsigned long long op_prev = clock64(), op_prev0 = op_prev;
#endif
    // ---- 1. the block's records, all loads in flight together ----
    // (requested before the slot's point count is known -- the index clamped to the slot's buffer, the count decides later which
    //  lanes hold a point: the block's first memory round trip is the records themselves, not the count in front of them)
    float4 pt[OP_PPT];
    unsigned xw[OP_PPT];    // velodyne: the raw azimuth (float bits), Livox: offset_time
    unsigned info[OP_PPT];  // [0:7] line id, OPI_* flags, rank among the points of the line / among the kept points in the wavefront round
    // (a context without this sensor -- max_velo_points / max_livox_points = 0 -- has no buffer for the clamped index to land in;
    //  the combined launch of <= 16 scans still dispatches the sensor's z-plane.  The per-slot resets below are then the other
    //  sensor's: n_in is 0 for this one)
    if ((SENSOR == 0 ? P.NV : P.NL) == 0) return;
    if constexpr (SENSOR == 0) {
        const float4* in = P.velo_in + (size_t)b * P.NV;
#pragma unroll
        for (int r = 0; r < OP_PPT; ++r) pt[r] = nt_load4(in + min(i0 + r * OP_THREADS + tid, P.NV - 1));
    } else {
        const mml_livox_point* in = P.livox_in + (size_t)b * P.NL;
#pragma unroll
        for (int r = 0; r < OP_PPT; ++r) {
            // (the record as a 16-byte and a 4-byte request: left to itself the compiler fetches the two bytes it needs of the
            //  last word with a byte load each -- three requests per record)
            typedef unsigned lv_u4 __attribute__((ext_vector_type(4), aligned(4)));
            const unsigned* w = reinterpret_cast<const unsigned*>(in + min(i0 + r * OP_THREADS + tid, P.NL - 1));
            const lv_u4 head = *reinterpret_cast<const lv_u4*>(w);
            const unsigned tail = w[4];  // reflectivity | tag << 8 | line << 16 | pad << 24
            pt[r] = make_float4(__uint_as_float(head.y), __uint_as_float(head.z), __uint_as_float(head.w),
                                (float)(tail & 255u));  // (the float the reflectivity becomes at :994)
            xw[r] = head.x;                 // offset_time
            info[r] = (tail >> 16) & 255u;  // line
        }
    }
    const int n = P.n_in[2 * b + SENSOR];
    // the per-slot resets (one block of the slot: the first Velodyne block, or the first Livox block of a scan without a Velodyne part)
    if (blk == 0 && tid == 0 && (SENSOR == 0 || P.n_in[2 * b] <= 0)) {
        P.slot_flags[2 * b] = 0;  // an extracted cloud, not undistorted yet
        P.brk_cnt[b] = 0;         // the stencil's two queues start empty
        P.redo_cnt[b] = 0;
    }
    if (i0 >= n) return;
    const int nkeys = SENSOR == 0 ? P.n_rings : P.n_lines;
    const int nbits = SENSOR == 0 ? P.ring_bits : P.line_bits;
    const int region = SENSOR == 0 ? 0 : P.NV;
    for (int k = tid; k < OP_GROUPS * OP_CSTRIDE; k += OP_THREADS) (&s_cnt[0][0])[k] = 0;
    if (SENSOR == 0 && wave == 0) {
        float so, eo;
        if (P.ends_inline) {
            sweep_ends(P, b, so, eo);
        } else {
            const AssignAux* a = reinterpret_cast<const AssignAux*>(P.assign_aux) + b;
            so = a->startOri;
            eo = a->endOri;
        }
        if (lane == 0) {
            S.ori[0] = so;
            S.ori[1] = eo;
        }
    }
    struct {
        float startOri, endOri;
    } aux;
    OP_MARK(0);
    __syncthreads();  // (the counters are zero, the sweep's ends are known)
    OP_MARK(1);
    aux.startOri = S.ori[0];
    aux.endOri = S.ori[1];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int cond_min = 0x7fffffff;  // (wave-uniform) first index of this wavefront's points that sets halfPassed
#pragma unroll
    for (int r = 0; r < OP_PPT; ++r) {
        const int i = i0 + r * OP_THREADS + tid;
        bool valid = false, keep = false, near_ok = false, cond = false;
        int key = 0;
        if (i < n) {
            if constexpr (SENSOR == 0) {
                const float4 p = pt[r];
                float ori = 0.f;
                if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {  // (non-finite records are removed at :1133)
                    const int ring = velo_ring(p, P.pitch0, P.pitch_step, P.n_rings);
                    valid = ring != 255;                                  // outside the ring table: dropped at :1163-1166
                    key = valid ? ring : 0;
                    ori = neg_atan2f(p.y, p.x);
                }
                xw[r] = __float_as_uint(ori);
                if (valid) {
                    // would this point set halfPassed (:1169-1177)?  Evaluated as if the flag were still clear: only the FIRST
                    // such point matters, and for it the flag is clear
                    float o = ori;
                    if (o < aux.startOri - M_PI / 2)
                        o += 2 * M_PI;
                    else if (o > aux.startOri + M_PI * 3 / 2)
                        o -= 2 * M_PI;
                    cond = o - aux.startOri > M_PI;
                }
            } else {
                const int line_num = (int)(info[r] & 255u);
                valid = !(line_num > nkeys - 1) && !(pt[r].x < 0.01);     // :989-990
                key = valid ? line_num : 0;
            }
            if (valid) crop_test(P, pt[r].x, pt[r].y, pt[r].z, keep, near_ok);
        }
        const unsigned long long eq = match_key(valid, key, nbits);
        const unsigned long long km = __ballot(keep), vm = __ballot(valid);
        const int g = r * OP_WAVES + wave;
        if (valid && (eq & lt) == 0) s_cnt[g][key] = __popcll(eq);
        if (lane == 0) {
            s_gv[g] = __popcll(vm);
            s_gk[g] = __popcll(km);
        }
        if constexpr (SENSOR == 0) {
            const unsigned long long cm = __ballot(cond);
            if (cm && cond_min == 0x7fffffff) cond_min = i0 + r * OP_THREADS + wave * 64 + (int)__ffsll((long long)cm) - 1;
        }
        info[r] = (unsigned)key | (valid ? OPI_VALID : 0u) | (keep ? OPI_KEEP : 0u) | (near_ok ? OPI_NEAR : 0u) |
                  ((unsigned)__popcll(eq & lt) << OPI_RANK_SHIFT) | ((unsigned)__popcll(km & lt) << OPI_KRANK_SHIFT);
        // (one point at a time: interleaving the sixteen evaluations keeps all their temporaries alive at once -- 207 registers, two
        //  wavefronts per SIMD)
        __builtin_amdgcn_sched_barrier(0);
    }
    OP_MARK(2);
    if (lane == 0) s_cond[wave] = cond_min;
    __syncthreads();
    OP_MARK(3);
    // ---- 2. per line: exclusive scan over the 64 groups (lane = group) ----
    const bool gl = lane < OP_GROUPS;
    for (int k = wave; k < nkeys; k += OP_WAVES) {
        const int v = gl ? s_cnt[lane][k] : 0;
        const int x = wave_incl_scan(v);
        if (gl) s_cnt[lane][k] = x - v;
        if (lane == 63) s_hist[k] = x;
    }
    if (wave == 0) {
        const int v = gl ? s_gv[lane] : 0, x = wave_incl_scan(v);
        if (gl) s_gv[lane] = x - v;
        if (lane == 63) s_hist[nkeys] = x;
    } else if (wave == 1) {
        const int v = gl ? s_gk[lane] : 0, x = wave_incl_scan(v);
        if (gl) s_gk[lane] = x - v;
        if (lane == 63) s_hist[nkeys + 1] = x;
    }
    OP_MARK(4);
    __syncthreads();
    OP_MARK(5);
    // ---- 3. offsets of the lines inside the block; publish; look back ----
    if (wave == 0) {
        OP_MARK0(10);
        const int h = lane < nkeys ? s_hist[lane] : 0;
        const int koff = wave_incl_scan(h) - h;
        if (lane < nkeys) s_koff[lane] = koff;
        const int tv = s_hist[nkeys], tk = s_hist[nkeys + 1];
        int cmin = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < OP_WAVES; ++q) cmin = min(cmin, s_cond[q]);
        const unsigned coff = cmin == 0x7fffffff ? OP_NONE : (unsigned)(cmin - i0);
        unsigned long long* agg = P.op_agg + ((size_t)b * 2 + SENSOR) * MML_SEG_MAX;
        const unsigned epoch = P.op_epoch;
        if (lane == 0) {
            const unsigned long long word = ((unsigned long long)epoch << 39) | ((unsigned long long)coff << 26) | ((unsigned long long)tk << 13) | (unsigned)tv;
            __hip_atomic_store(agg + blk, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // lanes 0 .. blk-1: the blocks in front of this one (they were dispatched before it: they are running or done).  Livox:
        // lanes 32 ..: every Velodyne block of the slot -- its kernel has completed on this stream -- for the number of Velodyne
        // points in front of the Livox part of the fused cloud.
        // The wait is bounded: a producer is a block dispatched BEFORE this one (smaller blockIdx.x of the same (y, z); for the Livox
        // blocks also every z = 0 block -- grid order is x fastest, then y, then z) that never waits for a later block, so a word
        // arrives within microseconds.  2^22 polls (seconds) without it mean a garbled n_in / epoch or a dispatcher that does not
        // keep that order: the kernel then traps -- an error at the next synchronisation -- instead of hanging the device.
        unsigned long long w = 0;
        bool need = lane < blk;
        unsigned polls = 0;
        while (__any(need)) {
            if (need) {
                w = __hip_atomic_load(agg + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                need = (unsigned)(w >> 39) != epoch;
            }
            if (__any(need)) {
                if (++polls > (1u << 22)) __builtin_trap();
                __builtin_amdgcn_s_sleep(1);
            }
        }
        OP_MARK0(11);
        int pv = lane < blk ? (int)(w & 0x1fffu) : 0;
        int pk = lane < blk ? (int)((w >> 13) & 0x1fffu) : 0;
        int pc = 0x7fffffff;
        if (lane < blk && (unsigned)((w >> 26) & 0x1fffu) != OP_NONE) pc = lane * MML_OP_BLK + (int)((w >> 26) & 0x1fffu);
        if constexpr (SENSOR == 1) {
            // (the Velodyne blocks of the slot are part of the same launch, z = 0: dispatched before every Livox block)
            const int nbv = (P.n_in[2 * b] + MML_OP_BLK - 1) / MML_OP_BLK;
            unsigned long long wv = 0;
            bool needv = lane >= 32 && lane - 32 < nbv;
            const bool minev = needv;
            while (__any(needv)) {
                if (needv) {
                    wv = __hip_atomic_load(P.op_agg + (size_t)b * 2 * MML_SEG_MAX + (lane - 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    needv = (unsigned)(wv >> 39) != epoch;
                }
                if (__any(needv)) {
                    if (++polls > (1u << 22)) __builtin_trap();
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (minev) pk = (int)((wv >> 13) & 0x1fffu);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pv += __shfl_xor(pv, o);
            pk += __shfl_xor(pk, o);
            pc = min(pc, __shfl_xor(pc, o));
        }
        if (lane == 0) {
            s_base[0] = pv;
            s_base[1] = pk;
            s_base[2] = min(pc, cmin);
        }
        // the block's record for k_assign_tables: line histogram, where each line's segment starts, the two counts
        int* rec = P.blk_cnt + ((size_t)(b * 2 + SENSOR) * P.nblk_max + blk) * BLK_STRIDE;
        if (lane < nkeys) {
            rec[lane] = h;
            rec[OP_REC_POS + lane] = region + pv + koff;
        }
        if (lane == 0) {
            rec[MAX_LINES] = tv;
            rec[MAX_LINES + 1] = tk;
        }
    }
    OP_MARK0(12);
    __syncthreads();
    OP_MARK(6);
    // ---- 4. the points leave ----
    const int base_valid = region + s_base[0], base_keep = s_base[1], trig = s_base[2];
    double timeSpan = 1.0;
    if constexpr (SENSOR == 1) timeSpan = livox_to_sec(P.livox_in[(size_t)b * P.NL + n - 1].offset_time);  // :985
#pragma unroll
    for (int r = 0; r < OP_PPT; ++r) {
        const unsigned f = info[r];
        if (!(f & OPI_VALID)) continue;
        const int i = i0 + r * OP_THREADS + tid;
        const int key = (int)(f & 255u), g = r * OP_WAVES + wave;
        const int dst = base_valid + s_koff[key] + s_cnt[g][key] + (int)((f >> OPI_RANK_SHIFT) & 63u);
        const int fdst = base_keep + s_gk[g] + (int)((f >> OPI_KRANK_SHIFT) & 63u);
        float rel;  // (also for the few points the crop drops: the undistortion runs over the whole region)
        if constexpr (SENSOR == 0) {
            const float startOri = aux.startOri, endOri = aux.endOri;
            float ori = __uint_as_float(xw[r]);
            if (i <= trig) {  // :1169-1177
                if (ori < startOri - M_PI / 2)
                    ori += 2 * M_PI;
                else if (ori > startOri + M_PI * 3 / 2)
                    ori -= 2 * M_PI;
            } else {  // :1178-1184
                ori += 2 * M_PI;
                if (ori < endOri - M_PI * 3 / 2)
                    ori += 2 * M_PI;
                else if (ori > endOri + M_PI / 2)
                    ori -= 2 * M_PI;
            }
            rel = (ori - startOri) / (endOri - startOri);  // :1186
        } else {
            rel = livox_to_sec(xw[r]) / timeSpan;           // :995
        }
        const size_t gpos = (size_t)b * P.NT + dst;
        P.ln_pts[gpos] = pt[r];
        const bool keep = (f & OPI_KEEP) != 0;
        // the two 4-byte records of the point go through LDS, in the order the block's region is laid out, and leave it as whole
        // rows below: as scattered 4-byte stores they were what the pass waited for (0.61 -> 0.71 ms when the 8-byte record of
        // round 3 became two arrays)
        const int q = dst - base_valid;
        s_out[0][q] = keep ? fdst : ((SENSOR == 1 && (f & OPI_NEAR)) ? -2 : -1);
        s_out[1][q] = __float_as_int(rel);
        __builtin_amdgcn_sched_barrier(0);
    }
    OP_MARK(7);
    __syncthreads();
    OP_MARK(8);
    {
        const int tv = s_hist[nkeys];
        int* og = P.ln_gidx + (size_t)b * P.NT + base_valid;
        int* orl = P.ln_rel + (size_t)b * P.NT + base_valid;
        for (int k = tid; k < tv; k += OP_THREADS) {
            og[k] = s_out[0][k];
            orl[k] = s_out[1][k];
        }
    }
#ifdef MML_OP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stores have left the wavefront)
    OP_MARK(9);
#endif
}

// one launch for both sensors (a handful of scans: one launch less in the chain): z = 0 the Velodyne blocks, z = 1 the Livox blocks,
// which wait for the Velodyne blocks' words ...
__global__ __launch_bounds__(OP_THREADS) void k_assign_onepass(FeatParams P) {
    __shared__ OnepassLds S;
    if (blockIdx.z == 0)
        assign_onepass_body<0>(P, S);
    else
        assign_onepass_body<1>(P, S);
}
// ... and one per sensor for batches (the combined kernel is 14 % slower at 1024 scans: 0.64 against 0.56 ms)
// (116 / 104 registers: four wavefronts per SIMD, two workgroups per CU.  Held to 80 for three workgroups it spills 176 bytes per lane
//  and takes 0.98 ms instead of 0.59 per 1024 scans; 16-bit counts in LDS -- under 40 KB -- change nothing: registers bound it.)
template <int SENSOR>
__global__ __launch_bounds__(OP_THREADS) void k_assign_onepass_s(FeatParams P) {
    __shared__ OnepassLds S;
    assign_onepass_body<SENSOR>(P, S);
}

// the line tables of a slot from its block records, one wavefront per slot (lane = line): line lengths, line-order starts,
// storage segments (seg_cum / seg_pos / seg_flat), the valid and kept counts of both sensors
// (... and, second half, the translation records of the slot's lines from the tables still in LDS: k_seg_records without a launch of
//  its own and without reading the tables back)
constexpr int TB_THREADS = 256;
__global__ __launch_bounds__(TB_THREADS) void k_assign_tables(FeatParams P, int count) {
    __shared__ int s_cum[64][MML_SEG_MAX + 1], s_pos[64][MML_SEG_MAX];
    __shared__ int s_nseg[64], s_len[64], s_rb[64];
    const int t = blockIdx.x;
    if (t >= count) return;
    const int b = P.first + t, tid = threadIdx.x;
    if (t == 0 && tid == 64) {  // the launch's two lists of lines left to k_select start empty (k_select_list fills them)
        P.sel_list_cnt[2 * P.first] = 0;
        P.sel_list_cnt[2 * P.first + 1] = 0;
    }
    if (tid < 64) {
        const int lane = tid;
        const bool in = lane < P.L;
        const int sensor = lane < P.n_rings ? 0 : 1;
        const int key = lane - (sensor == 0 ? 0 : P.n_rings);
        const int nkeys = sensor == 0 ? P.n_rings : P.n_lines;
        const int n = P.n_in[2 * b + sensor];
        const int nblk = (n + MML_OP_BLK - 1) / MML_OP_BLK;
        const int* rec0 = P.blk_cnt + ((size_t)(b * 2 + sensor) * P.nblk_max) * BLK_STRIDE;
        const size_t lo = (size_t)b * P.L + lane;
        int acc = 0;
        if (in) {
            int* cum = P.seg_cum + lo * (MML_SEG_MAX + 1);
            int* pos = P.seg_pos + lo * MML_SEG_MAX;
            int* flat = P.seg_flat + ((size_t)b * 2 + sensor) * MML_SEG_FLAT;
            // (the blocks' records are requested together: between the stores below the compiler keeps every load where it stands,
            //  and the loop was a chain of up to sixteen round trips)
            int rp[MML_SEG_MAX], rc[MML_SEG_MAX];
#pragma unroll
            for (int k = 0; k < MML_SEG_MAX; ++k) {
                const int* rec = rec0 + (size_t)(k < nblk ? k : 0) * BLK_STRIDE;
                rp[k] = rec[OP_REC_POS + key];
                rc[k] = rec[key];
            }
#pragma unroll
            for (int k = 0; k < MML_SEG_MAX; ++k) {
                if (k >= nblk) break;
                const int p = rp[k];
                cum[k] = acc;
                pos[k] = p;
                s_cum[lane][k] = acc;
                s_pos[lane][k] = p;
                flat[k * nkeys + key] = p;
                acc += rc[k];
            }
            cum[nblk] = acc;
            s_cum[lane][nblk] = acc;
            if (nblk == 0) {
                cum[1] = 0;
                pos[0] = sensor == 0 ? 0 : P.NV;
                s_cum[lane][1] = 0;
                s_pos[lane][0] = pos[0];
            }
            P.seg_n[lo] = nblk > 0 ? nblk : 1;
            s_nseg[lane] = nblk > 0 ? nblk : 1;
            P.line_len[lo] = acc;
            s_len[lane] = acc;
        }
        // line-order starts: rings from 0, Livox lines from NV
        const int x = wave_incl_scan(in ? acc : 0);
        const int velo_total = __shfl(x, P.n_rings - 1);
        if (in) {
            const int ls = sensor == 0 ? x - acc : P.NV + (x - acc - velo_total);
            P.line_start[lo] = ls;
            s_rb[lane] = (ls >> 6) + 6 * lane;  // = seg_rec_base
        }
        if (lane < 2) {  // lane = sensor
            const int ns = P.n_in[2 * b + lane];
            const int nb = (ns + MML_OP_BLK - 1) / MML_OP_BLK;
            const int* r0 = P.blk_cnt + ((size_t)(b * 2 + lane) * P.nblk_max) * BLK_STRIDE;
            int tv = 0, tk = 0;
#pragma unroll
            for (int k = 0; k < MML_SEG_MAX; ++k) {
                const size_t o = (size_t)(k < nb ? k : 0) * BLK_STRIDE;
                const int v = r0[o + MAX_LINES], kk = r0[o + MAX_LINES + 1];
                tv += k < nb ? v : 0;
                tk += k < nb ? kk : 0;
            }
            P.cb_n[2 * b + lane] = tv;
            AssignAux* a = reinterpret_cast<AssignAux*>(P.assign_aux) + b;
            if (lane == 0)
                a->kept_velo = tk;
            else
                a->kept_livox = tk;
            // fu_info: fused points, Velodyne points; the label counters start at zero (the selection kernels add to them)
            const int tk1 = __shfl(tk, 1);
            if (lane == 0) {
                int* info = P.fu_info + 8 * (size_t)b;
                info[0] = tk + tk1;
                info[1] = tk;
#pragma unroll
                for (int q = 2; q < 8; ++q) info[q] = 0;
            }
            const int nk = lane == 0 ? P.n_rings : P.n_lines;
            P.seg_flat_n[((size_t)b * 2 + lane) * 2] = nb * nk;
            P.seg_flat_n[((size_t)b * 2 + lane) * 2 + 1] = nk;
        }
    }
    __syncthreads();
    // ---- the translation records (see k_seg_records) ----
    for (int rho = tid; rho < P.seg_rstride; rho += TB_THREADS) {
        int lo = 0, hi = P.L - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_rb[mid] <= rho)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int line = lo, r = rho - s_rb[line], n = s_len[line], nseg = s_nseg[line];
        if (r < 0 || n <= 0) continue;
        const int* cum = s_cum[line];
        const int* pos = s_pos[line];
#pragma unroll
        for (int form = 0; form < 2; ++form) {
            const int a0 = 64 * r - (form == 0 ? 5 : 0);
            const int first = min(max(a0, 0), n - 1), last = min(a0 + 63, n - 1);
            int sg = 0;
            while (sg + 1 < nseg && cum[sg + 1] <= first) ++sg;
            int4 rec;
            rec.x = cum[sg + 1];
            rec.y = pos[sg] - cum[sg];
            rec.z = sg + 1 < nseg ? pos[sg + 1] - cum[sg + 1] : 0;
            rec.w = (sg + 2 > nseg || last < cum[sg + 2]) ? 1 : 0;
            (form == 0 ? P.seg_rs : P.seg_rw)[(size_t)b * P.seg_rstride + rho] = rec;
        }
    }
}

// The translation records of a slot (see SegTab): for every run of 64 consecutive line indices of every line the ONE segment
// boundary the run may cross and the storage offset on either side of it, so that the per-line kernels translate an index with a
// compare, a select and an add on scalars they load with one scalar request -- no search, no table in LDS.  .w = 0: the run
// crosses a second boundary (a sparse line: segments shorter than 64 points), the caller then walks the segment table itself.
// `shift` = 5: the rounds of k_stencil's tiles (run r = indices 64 r - 5 .. 64 r + 58), 0: aligned windows.
__device__ __forceinline__ int seg_rec_base(const FeatParams& P, int b, int line) {
    return (P.line_start[(size_t)b * P.L + line] >> 6) + 6 * line;  // (a tile of k_stencil reads up to 5 runs past the line's last full one)
}
__global__ __launch_bounds__(256) void k_seg_records(FeatParams P, int count) {
    const int b = P.first + blockIdx.y;
    const int rho = blockIdx.x * 256 + threadIdx.x;
    if (rho >= P.seg_rstride) return;
    // the line whose records cover rho: the last line whose base is <= rho (bases grow with the line number)
    int lo = 0, hi = P.L - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_rec_base(P, b, mid) <= rho)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int line = lo, r = rho - seg_rec_base(P, b, line);
    const int n = P.line_len[(size_t)b * P.L + line];
    if (r < 0 || n <= 0) return;  // (every run the line's range holds, also those behind its end: a tile reads them, clamped)
    const SegTab S = seg_tab(P, b, line);
#pragma unroll
    for (int form = 0; form < 2; ++form) {
        const int a0 = 64 * r - (form == 0 ? 5 : 0);
        const int first = min(max(a0, 0), n - 1), last = min(a0 + 63, n - 1);
        int sg = 0;
        while (sg + 1 < S.nseg && S.cum[sg + 1] <= first) ++sg;
        int4 rec;
        rec.x = S.cum[sg + 1];
        rec.y = S.pos[sg] - S.cum[sg];
        rec.z = sg + 1 < S.nseg ? S.pos[sg + 1] - S.cum[sg + 1] : 0;
        rec.w = (sg + 2 > S.nseg || last < S.cum[sg + 2]) ? 1 : 0;
        (form == 0 ? P.seg_rs : P.seg_rw)[(size_t)b * P.seg_rstride + rho] = rec;
    }
}

// storage positions of the line indices i - H .. i + H of one line (H <= 5) for a single lane (k_stencil_redo / k_stencil_break): the
// window lies inside the two rounds i / 64 and i / 64 + 1 of the stencil records; a sparse line goes through the segment table
template <int H>
__device__ __forceinline__ void seg_point_window(const FeatParams& P, int b, int line, int i, int (&pos)[2 * H + 1]) {
    const int4* rec = P.seg_rs + (size_t)b * P.seg_rstride + seg_rec_base(P, b, line);
    const int r = i >> 6;
    const int4 r0 = rec[r], r1 = rec[r + 1];
    if (r0.w && r1.w) {
#pragma unroll
        for (int k = 0; k < 2 * H + 1; ++k) {
            const int j = i + k - H;
            const bool lo = j <= 64 * r + 58;
            const int cb = lo ? r0.x : r1.x, dl = lo ? r0.y : r1.y, dh = lo ? r0.z : r1.z;
            pos[k] = j + (j < cb ? dl : dh);
        }
    } else {
        const SegTab seg = seg_tab(P, b, line);
        int sh = 0;
#pragma unroll
        for (int k = 0; k < 2 * H + 1; ++k) pos[k] = seg_xlate(seg, i + k - H, sh);
    }
}

// ---- a3 + every order-independent predicate of a5..a7: one point per lane -----------------------------------
// The reference evaluates its angle predicates as fabs(dot / (norm * norm)) > c in double, i.e. two sqrt and a
// division per test, eight vector normalisations for the included-angle test.  Here every such predicate is first
// evaluated in the equivalent sqrt/division-free form (dot^2 vs c^2 * n1 * n2, normalisations through v_rsq_f64 +
// Newton) with a guard band several orders above the rounding error of either form; only when the value falls
// inside the band (probability ~1e-10 per point) the reference expression itself is evaluated.  Decisions are
// therefore identical to the reference arithmetic while the common path stays free of f64 sqrt / div sequences.
__device__ __forceinline__ double rsqrt_nr(double z) {  // 1/sqrt(z), relative error ~1e-16 after two Newton steps
    double y = __builtin_amdgcn_rsq(z);
    y = y * (1.5 - (0.5 * z) * (y * y));
    y = y * (1.5 - (0.5 * z) * (y * y));
    return y;
}
// decides fabs(dot / (sqrt(n1) * sqrt(n2))) > c without sqrt / div; returns false through `certain` when too close
__device__ __forceinline__ bool abs_cos_gt(double dot, double n1, double n2, double c2, bool& certain) {
    const double lhs = dot * dot, rhs = c2 * (n1 * n2);
    const double diff = lhs - rhs;
    certain = fabs(diff) > 1e-10 * rhs;  // also false for degenerate (zero-length) vectors: rhs == 0
    return diff > 0.0;
}

// The stencil of one inner point; q = its 11-point window.
// FAST: every angle predicate is decided by its float pre-decision; when one of them falls inside its guard band the
// function returns false and writes nothing -- the point is then recomputed by k_stencil_redo, which runs the full
// decision chain (float, sqrt/div-free double, reference expression).  Keeping the double-precision tails out of the
// main kernel halves its register count: they are taken by ~1 point in 100 but every wavefront paid for them in
// occupancy.
// squared length of the segment between consecutive points a and b, in the reference's operation order (:567-569)
__device__ __forceinline__ float seg_sq(const float4 a, const float4 b) {
    const float dX = b.x - a.x, dY = b.y - a.y, dZ = b.z - a.z;
    return dX * dX + dY * dY + dZ * dZ;
}
// the window of the fast kernel: points and segment lengths in the workgroup's LDS tile (a segment is shared by the six
// windows that test it, so it is computed once)
struct WinTile {
    const float4* p;  // p[0..10] = points -5..5
    const float* s;   // s[k] = |p[k+1] - p[k]|^2
    __device__ __forceinline__ float4 operator[](int k) const { return p[k]; }
    __device__ __forceinline__ float seg(int o) const { return s[5 + o]; }  // segment between points o and o + 1
};
// ... and of the redo kernel: registers
struct WinRegs {
    const float4* p;
    __device__ __forceinline__ float4 operator[](int k) const { return p[k]; }
    __device__ __forceinline__ float seg(int o) const { return seg_sq(p[5 + o], p[5 + o + 1]); }
};
// The included-angle test of :615-644 for a point whose two half-windows are flat: true when the point is a plane-intersection
// corner (flag 150 if the stride walk visits it).  `decided` = false (FAST only): the float pre-decision was not certain.
template <bool FAST, typename WIN>
__device__ __forceinline__ bool c150_point(const WIN q, bool& decided_out) {
#define PT(o) q[5 + (o)]
    // :615-644: cc = |cos| between the weighted sums of the unit vectors to the four neighbours on either
    // side; flag 150 needs cc < 0.5 and both outermost neighbours farther than 5 cm.
    // (1) float pre-decision of cc >= 0.5 (the common case on a plane: cc ~ 1), accepted only when the
    //     squared form is 1e-3 away from the threshold and neither sum nearly cancels;
    // (2) the same squared form in double (v_rsq_f64 + Newton) with a 1e-10 band;
    // (3) the reference expression.
    bool c150 = false, decided = false;
    decided_out = false;
    {
        float lx = 0.f, ly = 0.f, lz = 0.f, rx = 0.f, ry = 0.f, rz = 0.f, za4 = 0.f, zb4 = 0.f;
#pragma unroll
        for (int k = 1; k < 5; k++) {
            const float ax = PT(-k).x - PT(0).x, ay = PT(-k).y - PT(0).y, az = PT(-k).z - PT(0).z;
            const float bx = PT(k).x - PT(0).x, by = PT(k).y - PT(0).y, bz = PT(k).z - PT(0).z;
            const float za = fmaf(az, az, fmaf(ay, ay, ax * ax)), zb = fmaf(bz, bz, fmaf(by, by, bx * bx));
            const float wa = (k / 10.0f) * (za > 0.f ? __builtin_amdgcn_rsqf(za) : 1.f);
            const float wb = (k / 10.0f) * (zb > 0.f ? __builtin_amdgcn_rsqf(zb) : 1.f);
            lx = fmaf(wa, ax, lx);
            ly = fmaf(wa, ay, ly);
            lz = fmaf(wa, az, lz);
            rx = fmaf(wb, bx, rx);
            ry = fmaf(wb, by, ry);
            rz = fmaf(wb, bz, rz);
            if (k == 4) {
                za4 = za;
                zb4 = zb;
            }
        }
        const float dt = fmaf(lz, rz, fmaf(ly, ry, lx * rx));
        const float n1 = fmaf(lz, lz, fmaf(ly, ly, lx * lx)), n2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
        const float rhs = 0.25f * (n1 * n2), df = fmaf(dt, dt, -rhs);
        const bool inr = (n1 > 0.25f) & (n2 > 0.25f) & (n1 < 4.f) & (n2 < 4.f);
        // cc >= 0.5 for sure: not a 150 point.  cc < 0.5 for sure: the 5 cm test on the outermost
        // neighbours decides, in float when neither squared distance is within 1e-5 of 0.0025.
        const bool ge = (df > 1e-3f * rhs) & inr, lt = (df < -1e-3f * rhs) & inr;
        const bool k4 = (fabsf(za4 - 0.0025f) > 2.5e-8f) & (fabsf(zb4 - 0.0025f) > 2.5e-8f);
        decided = ge | (lt & k4);
        c150 = lt & (za4 > 0.0025f) & (zb4 > 0.0025f);
    }
    if (!decided) {
    if constexpr (FAST) return false;  // decided stays false
    D3 nl = d3(0, 0, 0), nr = d3(0, 0, 0);
    double last2 = 0, cur2 = 0;
#pragma unroll
    for (int k = 1; k < 5; k++) {
        const D3 tl = d3(PT(-k).x - PT(0).x, PT(-k).y - PT(0).y, PT(-k).z - PT(0).z);
        const D3 tr = d3(PT(k).x - PT(0).x, PT(k).y - PT(0).y, PT(k).z - PT(0).z);
        const double zl = ddot(tl, tl), zr = ddot(tr, tr);
        const double il = zl > 0.0 ? rsqrt_nr(zl) : 1.0, ir = zr > 0.0 ? rsqrt_nr(zr) : 1.0;
        const double wl = (k / 10.0) * il, wr = (k / 10.0) * ir;
        nl.x += wl * tl.x;
        nl.y += wl * tl.y;
        nl.z += wl * tl.z;
        nr.x += wr * tr.x;
        nr.y += wr * tr.y;
        nr.z += wr * tr.z;
        if (k == 4) {
            last2 = zl;
            cur2 = zr;
        }
    }
    bool ca;
    const bool cc_ge = abs_cos_gt(ddot(nl, nr), ddot(nl, nl), ddot(nr, nr), 0.25, ca);  // cc > 0.5
    const bool cl = fabs(last2 - 0.0025) > 1e-12, cr = fabs(cur2 - 0.0025) > 1e-12;
    c150 = !cc_ge && last2 > 0.0025 && cur2 > 0.0025;
    if (!(ca && cl && cr)) {  // reference expression (:615-644)
        D3 norm_left = d3(0, 0, 0), norm_right = d3(0, 0, 0);
#pragma unroll
        for (int k = 1; k < 5; k++) {
            D3 tmp = d3(PT(-k).x - PT(0).x, PT(-k).y - PT(0).y, PT(-k).z - PT(0).z);
            dnormalize(tmp);
            norm_left.x += (k / 10.0) * tmp.x;
            norm_left.y += (k / 10.0) * tmp.y;
            norm_left.z += (k / 10.0) * tmp.z;
        }
#pragma unroll
        for (int k = 1; k < 5; k++) {
            D3 tmp = d3(PT(k).x - PT(0).x, PT(k).y - PT(0).y, PT(k).z - PT(0).z);
            dnormalize(tmp);
            norm_right.x += (k / 10.0) * tmp.x;
            norm_right.y += (k / 10.0) * tmp.y;
            norm_right.z += (k / 10.0) * tmp.z;
        }
        double cc = fabs(ddot(norm_left, norm_right) / (dnorm(norm_left) * dnorm(norm_right)));
        D3 last_tmp = d3(PT(-4).x - PT(0).x, PT(-4).y - PT(0).y, PT(-4).z - PT(0).z);
        D3 current_tmp = d3(PT(4).x - PT(0).x, PT(4).y - PT(0).y, PT(4).z - PT(0).z);
        double last_dis = dnorm(last_tmp);
        double current_dis = dnorm(current_tmp);
        c150 = cc < 0.5 && last_dis > 0.05 && current_dis > 0.05;
    }
    }
    decided_out = true;
    return c150;
#undef PT
}

// W: the window, indexable -5..5 around its centre (an LDS pointer in the fast kernel, registers in the redo kernel).
// SECTION(): in the fast kernel the window is re-read from LDS section by section, so that no more than one section's
// points are live in registers at a time.
template <bool FAST, bool WITH150, typename WIN>
__device__ __forceinline__ bool stencil_point(const WIN q, unsigned& attr, float& curv, float& refl, bool& brk) {
#define PT(o) q[5 + (o)]
#define SECTION()                               \
    do {                                        \
        if constexpr (FAST) asm volatile("" ::: "memory"); \
    } while (0)
    const float thDistanceFaraway = 50.0;
    const float thFlatThreshold = 0.02;
    const float thLidarNearestDis = 1.0;
    const float thBreakCornerDis = 1;
    // ---- :407-451 ----
    // The ordered sums below run on natural pairs -- (x, y) and (z, w) of a point sit in adjacent registers -- so that one
    // v_pk_add_f32 / v_pk_mul_f32 does the work of two scalar operations with the SAME roundings and the same order per
    // component (a packed float operation is two IEEE operations): 13 % fewer instructions per round.  The pairs are the
    // reference's own quantities side by side, not the cross-quantity pairs the SLP vectorizer formed (and paid for in moves).
    typedef float f2 __attribute__((ext_vector_type(2)));
#define LO(o) (f2{PT(o).x, PT(o).y})
#define HI(o) (f2{PT(o).z, PT(o).w})
    const f2 c_lo = LO(0), c_hi = HI(0);
    const f2 sq0 = c_lo * c_lo;
    const float dis2 = sq0.x + sq0.y + c_hi.x * c_hi.x;
    const float dis = sqrtf(dis2);  // == (float)sqrt((double)dis2): IEEE float sqrt
    bool unsure = false;
    // :421-422 fabs(cos) > 0.966 for both neighbours.  Float pre-decision (error ~1e-6 of the cosine, accepted only
    // when the squared form is more than 1e-3 away from the threshold), then the double squared form, then the
    // reference expression.
    bool gl, gn;
    const float seg_r = q.seg(0), seg_l = q.seg(-1);  // the two segments at the point: read once, used by three sections
    {
        const float lx = PT(-1).x - PT(0).x, ly = PT(-1).y - PT(0).y, lz = PT(-1).z - PT(0).z;
        const float nx = PT(1).x - PT(0).x, ny = PT(1).y - PT(0).y, nz = PT(1).z - PT(0).z;
        // (fused multiply-adds are fine here: this is the banded pre-decision, not the reference arithmetic)
        const float dotl = fmaf(lz, PT(0).z, fmaf(ly, PT(0).y, lx * PT(0).x));
        const float dotn = fmaf(nz, PT(0).z, fmaf(ny, PT(0).y, nx * PT(0).x));
        const float rl = (0.966f * 0.966f) * (seg_l * dis2);
        const float rn = (0.966f * 0.966f) * (seg_r * dis2);
        const float el = fmaf(dotl, dotl, -rl), en = fmaf(dotn, dotn, -rn);
        gl = el > 0.f;
        gn = en > 0.f;
        const bool sure = (fabsf(el) > 1e-3f * rl) & (fabsf(en) > 1e-3f * rn) & (rl < 1e30f) & (rn < 1e30f) &
                          (rl > 1e-30f) & (rn > 1e-30f);
        if constexpr (FAST) {
            // (the fast kernel finishes the point anyway -- its flatness bits, exact float tests, feed the stride walk -- and
            //  reports it as uncertain at the end)
            unsure = !sure;
        } else if (!sure) {
            const D3 pt_cur = d3(PT(0).x, PT(0).y, PT(0).z);
            const D3 dl = d3((double)PT(-1).x - pt_cur.x, (double)PT(-1).y - pt_cur.y, (double)PT(-1).z - pt_cur.z);
            const D3 dn = d3((double)PT(1).x - pt_cur.x, (double)PT(1).y - pt_cur.y, (double)PT(1).z - pt_cur.z);
            const double n0 = ddot(pt_cur, pt_cur);
            bool c1, c2;
            gl = abs_cos_gt(ddot(dl, pt_cur), ddot(dl, dl), n0, 0.966 * 0.966, c1);
            gn = abs_cos_gt(ddot(dn, pt_cur), ddot(dn, dn), n0, 0.966 * 0.966, c2);
            if (!(c1 && c2)) {  // reference expression (:421-422)
                const double ncur = dnorm(pt_cur);
                const double angle_last = ddot(dl, pt_cur) / (dnorm(dl) * ncur);
                const double angle_next = ddot(dn, pt_cur) / (dnorm(dn) * ncur);
                gl = fabs(angle_last) > 0.966;
                gn = fabs(angle_next) > 0.966;
            }
        }
    }
    const bool grazing = gl && gn;
    int thNumCurvSize;
    if (dis > thDistanceFaraway || grazing) {
        thNumCurvSize = 2;
        attr |= A_W2;
    } else {
        thNumCurvSize = 3;
    }
    if (grazing) attr |= A_ANGLE;
    // the loop at :435-440, unrolled (j = 1, 2 always; j = 3 when the window is 3).  diffX / diffY / diffZ start at 0 and lose
    // 2 n x the centre at the end; diffR STARTS at -2 n x the centre's reflectivity (:433): the (z, w) pair starts at
    // (0, -2 n w) and ends with - (2 n z, +0) -- subtracting +0 changes nothing, not even the sign of a zero.
    const float n2 = (float)(2 * thNumCurvSize);
    f2 dxy = f2{0.f, 0.f}, dzw = f2{0.f, (float)(-2 * thNumCurvSize) * c_hi.y};
    dxy += LO(-1) + LO(1);
    dzw += HI(-1) + HI(1);
    dxy += LO(-2) + LO(2);
    dzw += HI(-2) + HI(2);
    if (thNumCurvSize == 3) {
        dxy += LO(-3) + LO(3);
        dzw += HI(-3) + HI(3);
    }
    dxy -= n2 * c_lo;
    dzw -= f2{n2 * c_hi.x, 0.f};
    const f2 dsq = dxy * dxy;
    curv = dsq.x + dsq.y + dzw.x * dzw.x;
    refl = dzw.y;
    // the half-window sums of :569 / :584 here, in the section that already holds points -3 .. 3: every point of the window is
    // read from LDS once (the rounds read 21 LDS words per point when this lived in its own section; the LDS of a CU serves
    // four SIMDs and was as loaded as their issue ports)
    const f2 lxy = LO(-4) + LO(-3) - 4.f * LO(-2) + LO(-1) + c_lo, lzw = HI(-4) + HI(-3) - 4.f * HI(-2) + HI(-1) + c_hi;
    const f2 lsq = lxy * lxy;
    const float left_curvature = lsq.x + lsq.y + lzw.x * lzw.x;
    const f2 rxy = LO(4) + LO(3) - 4.f * LO(2) + LO(1) + c_lo, rzw = HI(4) + HI(3) - 4.f * HI(2) + HI(1) + c_hi;
    const f2 rsq = rxy * rxy;
    const float right_curvature = rsq.x + rsq.y + rzw.x * rzw.x;
    // ---- predicates of :488, :499/:512, :524, :534-535 ----
    SECTION();
    if (curv < thFlatThreshold * dis * thFlatThreshold * dis) attr |= A_CAND3;
    const bool far = dis > thDistanceFaraway;
    if (far) attr |= A_FAR;
    if (curv < 0.7 * thFlatThreshold * dis * thFlatThreshold * dis && refl > 20.0) attr |= A_REFL;
    {
        // the reference compares the float sum with the double 0.02 (:569, :584).  0.02f is the float just below
        // 0.02, so for a float x: x > 0.02 <=> x > 0.02f.
        constexpr float kTh002 = 0.02f;
        static_assert((double)kTh002 < 0.02, "0.02f must round down");
        // the two loops of :492-517 stop at the first segment longer than the threshold: the six segment lengths are
        // fetched together and the counts formed without branches (a loop that breaks on loaded data is, on a wavefront, a
        // chain of dependent LDS round trips and exec-mask updates)
        const float sa0 = seg_r, sa1 = q.seg(1), sa2 = q.seg(2);     // |PT(l) - PT(l - 1)|^2, l = 1, 2, 3
        const float sb0 = seg_l, sb1 = q.seg(-2), sb2 = q.seg(-3);  // |PT(l) - PT(l + 1)|^2, l = -1, -2, -3
        const bool ga0 = !(sa0 > kTh002 || far), ga1 = ga0 && !(sa1 > kTh002), ga2 = ga1 && !(sa2 > kTh002);
        const bool gb0 = !(sb0 > kTh002 || far), gb1 = gb0 && !(sb1 > kTh002), gb2 = gb1 && !(sb2 > kTh002);
        const int a3 = (int)ga0 + (int)ga1 + (int)ga2;
        const int b3 = (int)gb0 + (int)gb1 + (int)gb2;
        attr |= (unsigned)a3 << A_A3_SHIFT;
        attr |= (unsigned)b3 << A_B3_SHIFT;
    }
    // ---- :543-650 (per visited point; which points are visited is decided in k_select) ----
    SECTION();
    {
        const float depth = dis;
        const bool lflat = left_curvature < thFlatThreshold * depth;
        const bool rflat = right_curvature < thFlatThreshold * depth;
        if (lflat) attr |= A_LFLAT;
        if (rflat) attr |= A_RFLAT;
        SECTION();
        if constexpr (WITH150) {
            if (lflat && rflat) {
                bool decided;
                const bool c150 = c150_point<FAST>(q, decided);
                if (!decided) return false;  // (FAST only)
                if (c150) attr |= A_C150;
            }
        }
    }
    // ---- :651-806 break points ----
    SECTION();
    // Only a range discontinuity of more than 1 m can make a break point; what follows that test (two more square
    // roots, a double-precision angle, six double normalisations) concerns a few dozen points per scan, but a
    // wavefront pays for it as soon as ONE of its 64 lanes qualifies.  Those points are therefore queued and
    // finished by k_stencil_break with every lane busy.
    {
        const float sq_right = seg_r, sq_left = seg_l;
        // two distances below 1 m cannot differ by more than thBreakCornerDis = 1 (sqrtf is monotone and
        // sqrtf(x) <= 1 for x < 1)
        brk = !(sq_right < 1.f && sq_left < 1.f);
    }
    if (dis2 < thLidarNearestDis * thLidarNearestDis) attr |= A_NEAR;
#undef LO
#undef HI
#undef PT
#undef SECTION
    return !unsure;
}

// Transfer table of the stride-1-or-4 walk (:543-650) over 8 positions.  row[RFLAT byte]: low word = for each offset e = 0..3 of
// the first visited position the visited byte (bits 8e..8e+7); high word = the offset into the next byte, already multiplied by 8
// (bits 8e..8e+7) -- so that the walk state IS the bit offset of both fields and a step is two bit-field extracts.  A compile-time
// constant.  The walk of a 64-point window fetches its eight rows at once and chains them in registers; the entry offset of a
// window is the exit of its predecessor.
struct alignas(16) WalkTab {
    unsigned long long row[256];
};
constexpr WalkTab make_walk_tab() {
    WalkTab t{};
    for (int m8 = 0; m8 < 256; ++m8) {
        unsigned long long r = 0;
        for (int e = 0; e < 4; ++e) {
            int pos = e;
            unsigned v = 0;
            while (pos < 8) {
                v |= 1u << pos;
                pos += ((m8 >> pos) & 1) ? 4 : 1;
            }
            r |= (unsigned long long)v << (8 * e);
            r |= (unsigned long long)(8 * (pos - 8)) << (32 + 8 * e);
        }
        t.row[m8] = r;
    }
    return t;
}
__device__ const WalkTab g_walk_tab = make_walk_tab();

#ifdef MML_ST_TIMING
// phase clocks of one k_stencil workgroup (line MML_ST_TIMING of the 8th slot of the launch): cycles spent up to each mark, summed
// over the tiles of the line
__device__ unsigned long long g_st_dbg[16];
#define ST_MARK(id)                                                       \
    do {                                                                  \
        if (st_dbg) {                                                     \
            const unsigned long long now_ = clock64();                    \
            g_st_dbg[id] += now_ - st_prev;                               \
            st_prev = now_;                                               \
        }                                                                 \
    } while (0)
extern "C" int mml_debug_st_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_st_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#else
#define ST_MARK(id)
#endif

// One WAVEFRONT per scan line (four lines per workgroup, nothing shared between them: no workgroup barrier anywhere), the line
// in tiles of ST_TILE positions -- four rounds of one point per lane through the wavefront's LDS tile of the points and of the
// squared segment lengths.  Per tile:
//   1. every order-independent predicate except the included-angle test (float pre-decisions; uncertain points -> redo queue)
//   2. the stride walk of :543-650 over the tile -- it depends only on the RFLAT bits just computed (four ballots, in scalar
//      registers): lane (w, e) walks window w entered at offset e through the transfer table, the windows are chained with
//      four lane reads, the exit offset is carried to the next tile in a scalar register
//   3. the included-angle test (eight normalisations, a third of the old kernel) only for the points the walk visits and
//      whose half-windows are both flat -- on a plane the walk strides by 4, so that is every fourth point --, compacted into
//      a list and evaluated with all lanes busy; then the attribute words are written.
// A wavefront's serial stretches (tile loads in flight, the walk's dependent chain) are covered by the other wavefronts of
// the SIMD, which are at other points of other lines: the workgroup-per-line form of this kernel lost a quarter of its time
// to three wavefronts waiting at a barrier for the one that walks.
constexpr int ST_TILE = 256;
#ifndef MML_ST_LINES
#define MML_ST_LINES 2  // (1: 0.706, 2: 0.684, 4: 0.712, 8: 0.730 ms per 1024 scans)
#endif
constexpr int ST_LINES = MML_ST_LINES;  // lines (wavefronts) per workgroup
constexpr int ST_SEGMENT_MAX_SLOTS = 16;  // batches up to this size take the segment mode (below)
// MODE 0: a wavefront walks its whole line, tile after tile (grid: lines / ST_LINES x slots) -- the form for batches, where there
//         are thousands of lines to fill the device with.
// MODE 1 + MODE 2: segment mode for a handful of scans (the live one-scan call, the single-line entry point), where a 4 000-point
//         line walked tile by tile is 16 serial round trips on an otherwise empty device: every TILE gets a wavefront (grid: tiles
//         x lines / ST_LINES x slots).  Launch 1 runs the rounds and records, per tile, where the stride walk leaves it for each of
//         the four offsets it may enter at (one byte); launch 2 composes the bytes of the tiles before its own into its entry
//         offset, reloads the tile (it is in the L2), walks it, and runs the included-angle pass.  Same results as MODE 0.
template <int MODE>
__global__ __launch_bounds__(64 * ST_LINES) void k_stencil(FeatParams P) {
#ifdef MML_ST_GLOBAL_TAB
    const unsigned long long* s_row = g_walk_tab.row;
#else
    __shared__ unsigned long long s_row[256];  // the transfer table (2 KB): eight look-ups per window walk
    for (int k = threadIdx.x; k < 256; k += 64 * ST_LINES) s_row[k] = g_walk_tab.row[k];
    __syncthreads();  // (the only workgroup barrier: before any wavefront leaves)
#endif
    // MODE 0: grid (slots, line groups), the line groups taken Livox lines first: the long lines (4 000 points against 1 800) of ALL
    // slots are dispatched before the short ones, so that the launch does not end on a few long workgroups (with the slot as the slow
    // grid index every slot's three Livox groups came behind its eight ring groups: 0.59 -> 0.54 ms per 1024-scan launch)
    const int b = (MODE == 0 ? blockIdx.x : blockIdx.z) + P.first;
    const int wave_id = threadIdx.x >> 6;
    // (the wavefront's number is a scalar: said so, everything derived from the line -- its table entries, base addresses -- stays
    //  in scalar registers)
    const int ngrp = (P.L + ST_LINES - 1) / ST_LINES, g0 = P.n_rings / ST_LINES;  // first group that holds a Livox line
    const int grp = MODE == 0 ? (int)((blockIdx.y + g0) % ngrp) : (int)blockIdx.y;
    const int line = __builtin_amdgcn_readfirstlane((int)(grp * ST_LINES + wave_id));
    if (line >= P.L) return;
    const int n = P.line_len[(size_t)b * P.L + line];
    if (n <= 0) return;
    const int start = P.line_start[(size_t)b * P.L + line];
    const size_t base = (size_t)b * P.NT + start;  // line order: ln_curv / ln_refl / ln_attr
    const float4* slot_pts = P.ln_pts + (size_t)b * P.NT;
    // where the line's points are stored: the translation records of its rounds (k_seg_records), the segment table for sparse lines
    // (its three words as the scalars they are: as values derived from the wavefront's number they sat in five vector registers
    //  across the whole tile loop, one wavefront per SIMD less)
    SegTab seg = seg_tab(P, b, line);
    seg.cum = reinterpret_cast<const int*>(seg_uniform_ptr(reinterpret_cast<const int4*>(seg.cum)));
    seg.pos = reinterpret_cast<const int*>(seg_uniform_ptr(reinterpret_cast<const int4*>(seg.pos)));
    seg.nseg = __builtin_amdgcn_readfirstlane(seg.nseg);
    const int4* seg_rec = seg_uniform_ptr(P.seg_rs + (size_t)b * P.seg_rstride + seg_rec_base(P, b, line));
    __shared__ float4 s_pt_all[ST_LINES][ST_TILE + 10];
    __shared__ float s_sq_all[ST_LINES][ST_TILE + 10];
    __shared__ unsigned short s_attr_all[ST_LINES][ST_TILE];
    float4* s_pt = s_pt_all[wave_id];
    float* s_sq = s_sq_all[wave_id];
    unsigned short* s_attr = s_attr_all[wave_id];
    unsigned short* s_list = reinterpret_cast<unsigned short*>(s_sq);  // (the segment lengths are dead once the rounds are done)
#ifdef MML_ST_TIMING
    const bool st_dbg = (threadIdx.x & 63) == 0 && line == MML_ST_TIMING && b == P.first + 517;
    unsigned long long st_prev = clock64();
#endif
    // Every phase of the tile loop derives its lane number and addresses from its own opaque copy of the thread index: the
    // compiler otherwise keeps the loop-invariant values of ALL phases in registers across the whole loop
#define PHASE_IDS()                               \
    int lane = threadIdx.x & 63;                  \
    asm volatile("" : "+v"(lane))
#define WAVE_SYNC()                                        \
    do {                                                   \
        __builtin_amdgcn_wave_barrier();                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    } while (0)
    unsigned carry = 0;  // 8 x offset of the first visited position in the tile's first byte (wave-uniform)
    unsigned char* tile_exit = P.st_exit + (size_t)b * P.st_stride + (start >> 8) + line;  // (segment mode) this line's tiles
    const int t_first = MODE == 0 ? 0 : (int)blockIdx.x * ST_TILE, t_step = MODE == 0 ? ST_TILE : (int)gridDim.x * ST_TILE;
    for (int t0 = t_first; t0 < n; t0 += t_step) {
        WAVE_SYNC();  // the previous tile has been consumed
        ST_MARK(0);
        if constexpr (MODE == 2) {  // entry offset = the exits of the tiles before this one, composed
            unsigned c = 0;
            for (int u = 0; u < t0 / ST_TILE; ++u) c = 8u * ((tile_exit[u] >> (c >> 2)) & 3u);
            carry = c;
        }
        {
            PHASE_IDS();
            // the ST_TILE + 10 points the tile's windows cover, read once, and the squared length of every segment between
            // consecutive points (each is tested by six windows).  Every lane requests its point AND the successor (the second
            // request is served from the lines the first one brings in), all requests of the phase in flight together.
            constexpr int NLD = (ST_TILE + 10 + 63) / 64;
            float4 pa[NLD];
            // Per round of 64 consecutive line indices: the one segment boundary the round may cross and the storage offset on either
            // side of it come as a precomputed record (k_seg_records) with one scalar request per round: a lane's translation is a
            // compare, a select and an add.  A round that crosses two boundaries (segments shorter than 64 points: a sparse line)
            // sends the tile through the general path below.
            static_assert(NLD == 5, "five records per tile");
            seg_v4i rr[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) rr[u] = seg_sload(seg_rec, (t0 >> 6) + u);
            SEG_SWAIT5(rr[0], rr[1], rr[2], rr[3], rr[4]);
            const bool ok = (rr[0].w & rr[1].w & rr[2].w & rr[3].w & rr[4].w) != 0;
            if (ok) {
#pragma unroll
                for (int u = 0; u < NLD; ++u) {  // every request of the phase in flight together
                    const int gp = t0 - 5 + lane + 64 * u;
                    const bool in_tile = lane + 64 * u < ST_TILE + 10;
                    const bool ha = in_tile && gp >= 0 && gp < n;
                    pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ha) pa[u] = slot_pts[(unsigned)(gp + (gp < rr[u].x ? rr[u].y : rr[u].z))];
                }
                // the successor of a point is the next lane's point (whole-wave DPP shift; lane 63 takes lane 0 of the next round):
                // one request per point, and no second set of addresses and values in registers
#pragma unroll
                for (int u = 0; u < NLD; ++u) {
                    const int k = lane + 64 * u;
                    float4 nx = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (u + 1 < NLD) {  // (bit patterns: the builtin takes an int)
                        nx.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pa[u + 1 < NLD ? u + 1 : u].x)));
                        nx.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pa[u + 1 < NLD ? u + 1 : u].y)));
                        nx.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pa[u + 1 < NLD ? u + 1 : u].z)));
                    }
                    nx.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(nx.x), __float_as_int(pa[u].x), 0x130, 0xf, 0xf, false));
                    nx.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(nx.y), __float_as_int(pa[u].y), 0x130, 0xf, 0xf, false));
                    nx.z = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(nx.z), __float_as_int(pa[u].z), 0x130, 0xf, 0xf, false));
                    if (k < ST_TILE + 10) s_pt[k] = pa[u];
                    if (k < ST_TILE + 9) s_sq[k] = seg_sq(pa[u], nx);
                }
            } else {
                // (wave-uniform, rare) a sparse line -- a round crosses two segment boundaries: point and successor through the
                // segment table, one index at a time, straight into the tile (a rolled loop: the common path must not pay registers
                // for it)
                int sh = 0;
#pragma unroll 1
                for (int k = lane; k < ST_TILE + 10; k += 64) {
                    const int gp = t0 - 5 + k;
                    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                    if (gp >= 0 && gp < n) va = slot_pts[seg_xlate(seg, gp, sh)];
                    if (gp + 1 >= 0 && gp + 1 < n) {
                        int s2 = sh;
                        vb = slot_pts[seg_xlate(seg, gp + 1, s2)];
                    }
                    s_pt[k] = va;
                    if (k < ST_TILE + 9) s_sq[k] = seg_sq(va, vb);
                }
            }
        }
        WAVE_SYNC();
        ST_MARK(1);
        unsigned redo_bits = 0;
        unsigned long long rmask[4] = {0ull, 0ull, 0ull, 0ull};
        if constexpr (MODE != 2) {
            PHASE_IDS();
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if (t0 + rep * 64 >= n) break;  // wave-uniform
                int tp = rep * 64 + lane;
                asm volatile("" : "+v"(tp));
                const int i = t0 + tp;
                const bool live = i < n;
                unsigned attr = 0;
                float curv = 0.f, refl = 0.f;
                bool brk = false, redo = false;
                if (live && i >= 5 && i < n - 5) {
                    redo = !stencil_point<true, false>(WinTile{s_pt + tp, s_sq + tp}, attr, curv, refl, brk);
                    brk = brk && !redo;
                    if (redo) attr &= A_LFLAT | A_RFLAT;  // (exact float tests, always decided: the walk needs them now)
                }
                // (addresses of the stores below from a fresh copy of the index: none of them stays live across the stencil above)
                int tq = tp;
                asm volatile("" : "+v"(tq));
                const int iq = t0 + tq;
                if (live && !redo) {
                    P.ln_curv[base + iq] = curv;
                    P.ln_refl[base + iq] = refl;
                }
                s_attr[tq] = (unsigned short)attr;
                if (redo) redo_bits |= 1u << rep;
                rmask[rep] = __ballot(attr & A_RFLAT);
                // wave-aggregated append to the slot's break-point queue
                const unsigned long long bm = __ballot(brk);
                if (bm) {
                    int first = 0;
                    if (lane == (int)__ffsll((long long)bm) - 1) first = atomicAdd(&P.brk_cnt[b], __popcll(bm));
                    first = __builtin_amdgcn_readlane(first, (int)__ffsll((long long)bm) - 1);  // (wave-uniform source lane)
                    if (brk) P.brk_queue[(size_t)b * P.NT + first + lower_count(bm)] = ((unsigned)line << 24) | (unsigned)iq;
                }
                // ... and to its redo queue
                const unsigned long long rm = __ballot(redo);
                if (rm) {
                    int first = 0;
                    if (lane == (int)__ffsll((long long)rm) - 1) first = atomicAdd(&P.redo_cnt[b], __popcll(rm));
                    first = __builtin_amdgcn_readlane(first, (int)__ffsll((long long)rm) - 1);
                    if (redo) P.redo_queue[(size_t)b * P.NT + first + lower_count(rm)] = ((unsigned)line << 24) | (unsigned)iq;
                }
            }
        } else {  // the rounds ran in launch 1: the flatness bits come back with the attribute words
            PHASE_IDS();
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if (t0 + rep * 64 >= n) break;
                const int tp = rep * 64 + lane, i = t0 + tp;
                unsigned attr = i < n ? (unsigned)P.ln_attr[base + i] : 0u;
                // (launch 1 marked the points it queued for k_stencil_redo in the -- then still unused -- walk bit: they must not
                //  get the included-angle test here and a second queue entry)
                if (attr & A_VIS) redo_bits |= 1u << rep;
                attr &= ~(unsigned)A_VIS;
                s_attr[tp] = (unsigned short)attr;
                rmask[rep] = __ballot(attr & A_RFLAT);
            }
        }
        ST_MARK(2);
        // ---- the walk over this tile: lane (w, e), w < 4, = window w entered at offset e ----
        unsigned long long vmask[4] = {0ull, 0ull, 0ull, 0ull};
#ifndef MML_ST_NOWALK
        {
            PHASE_IDS();
            const int w = (lane >> 2) & 3;
            const unsigned long long m = w == 0 ? rmask[0] : (w == 1 ? rmask[1] : (w == 2 ? rmask[2] : rmask[3]));
            unsigned long long row[8];
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) row[sb] = s_row[(unsigned)(m >> (8 * sb)) & 255u];
            // state = 8 x (offset of the first visited position in the byte) = bit offset of this entry's fields in a row
            unsigned st = 8u * (lane & 3), vlo, vhi;
            if (t0 == 0) {  // (wave-uniform) the walk starts at index 5: positions 5, 6, 7 of the line's first byte by hand
                const unsigned r5 = (unsigned)(m >> 5) & 1u, r6 = (unsigned)(m >> 6) & 1u, r7 = (unsigned)(m >> 7) & 1u;
                const unsigned v6 = r5 ^ 1u, v7 = v6 & (r6 ^ 1u);
                const unsigned first_vis = (1u << 5) | (v6 << 6) | (v7 << 7);
                const unsigned first_exit = v7 ? (r7 ? 3u : 0u) : (v6 ? 2u : 1u);
                const bool hand = w == 0;
                vlo = hand ? first_vis : __builtin_amdgcn_ubfe((unsigned)row[0], st, 8u);
                st = hand ? 8u * first_exit : __builtin_amdgcn_ubfe((unsigned)(row[0] >> 32), st, 8u);
            } else {
                vlo = __builtin_amdgcn_ubfe((unsigned)row[0], st, 8u);
                st = __builtin_amdgcn_ubfe((unsigned)(row[0] >> 32), st, 8u);
            }
#pragma unroll
            for (int sb = 1; sb < 4; ++sb) {
                vlo |= __builtin_amdgcn_ubfe((unsigned)row[sb], st, 8u) << (8 * sb);
                st = __builtin_amdgcn_ubfe((unsigned)(row[sb] >> 32), st, 8u);
            }
            vhi = __builtin_amdgcn_ubfe((unsigned)row[4], st, 8u);
            st = __builtin_amdgcn_ubfe((unsigned)(row[4] >> 32), st, 8u);
#pragma unroll
            for (int sb = 5; sb < 8; ++sb) {
                vhi |= __builtin_amdgcn_ubfe((unsigned)row[sb], st, 8u) << (8 * (sb - 4));
                st = __builtin_amdgcn_ubfe((unsigned)(row[sb] >> 32), st, 8u);
            }
            // chain the four windows: the entry offset of a window is the exit of its predecessor for ITS entry offset
            if constexpr (MODE == 1) {
                // segment mode, launch 1: where the walk leaves this tile for each of the four offsets it may enter at
                unsigned summary = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned e = 8u * c;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) e = (unsigned)__builtin_amdgcn_readlane((int)st, ww * 4 + (int)(e >> 3));
                    summary |= (e >> 3) << (2 * c);
                }
                if (lane == 0) tile_exit[t0 / ST_TILE] = (unsigned char)summary;
                (void)vlo;
                (void)vhi;
            } else {
                unsigned e = carry;  // (8 x offset)
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const int src = ww * 4 + (int)(e >> 3);
                    vmask[ww] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)vhi, src) << 32) |
                                (unsigned)__builtin_amdgcn_readlane((int)vlo, src);
                    e = (unsigned)__builtin_amdgcn_readlane((int)st, src);
                }
                carry = e;
            }
        }
#endif
        ST_MARK(3);
        if constexpr (MODE == 1) {  // the attribute words as the rounds left them (launch 2 reads the flatness bits back)
            PHASE_IDS();
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                const int tp = rep * 64 + lane, i = t0 + tp;
                if (i < n) P.ln_attr[base + i] = (unsigned short)(s_attr[tp] | (((redo_bits >> rep) & 1u) ? (unsigned)A_VIS : 0u));
            }
            continue;
        }
        // ---- list of the points that need the included-angle test, the test, the attribute words ----
        {
            PHASE_IDS();
            int n150 = 0;
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if (t0 + rep * 64 >= n) break;
                const int tp = rep * 64 + lane, i = t0 + tp;
                const unsigned attr = s_attr[tp];
                const bool vis = ((vmask[rep] >> lane) & 1ull) && i >= 5 && i < n - 5;
                if (vis) s_attr[tp] = (unsigned short)(attr | A_VIS);
                const bool want = vis && (attr & A_LFLAT) && (attr & A_RFLAT) && !((redo_bits >> rep) & 1u);
                const unsigned long long wm = __ballot(want);
                if (want) s_list[n150 + lower_count(wm)] = (unsigned short)tp;
                n150 += __popcll(wm);
            }
            WAVE_SYNC();
#ifdef MML_ST_NO150
            n150 = 0;
#endif
            for (int e = lane; e < n150; e += 64) {
                const int tp = s_list[e];
                bool decided;
                const bool c150 = c150_point<true>(WinTile{s_pt + tp, s_sq + tp}, decided);
                if (!decided)  // (about one point in a thousand) the whole point again, with the full decision chain
                    P.redo_queue[(size_t)b * P.NT + atomicAdd(&P.redo_cnt[b], 1)] = ((unsigned)line << 24) | (unsigned)(t0 + tp);
                else if (c150)
                    s_attr[tp] = (unsigned short)(s_attr[tp] | A_C150);
            }
            WAVE_SYNC();
            ST_MARK(4);
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                const int tp = rep * 64 + lane, i = t0 + tp;
                if (i < n) P.ln_attr[base + i] = ((redo_bits >> rep) & 1u) ? (unsigned short)(s_attr[tp] & A_VIS) : s_attr[tp];
            }
        }
        ST_MARK(5);
    }
#undef PHASE_IDS
#undef WAVE_SYNC
}

// ---- the stencil's two queues, worked off -----------------------------------------------------------------------------------
// Per slot the queues are short -- ~110 uncertain points, ~1000 break-point candidates of a 52.8 k-point scan -- and a launch of one
// (or four) 256-thread workgroups per slot runs mostly empty wavefronts that still wait for 118 registers each: 0.047 + 0.027 ms
// per 1024 scans, and in the two-lane timed region the redo kernel showed up as long as k_stencil itself.  Batches therefore walk
// ONE list per launch: k_queue_prefix turns the per-slot counts into offsets, a grid sized for the expected fill walks the
// concatenation with full wavefronts (entry e -> slot by a binary search over the offsets).  A handful of scans keep the per-slot grid.
// (256 threads, four items per thread: ONE workgroup of sixteen wavefronts found no CU with four free wave slots on every SIMD while
//  the other lane's kernel kept refilling the slots its four-wavefront workgroups freed -- 0.2 ms on average, up to 0.8, per call in
//  the two-lane timed region for 5 us of work; a four-wavefront workgroup is placed at once)
constexpr int QP_THREADS = 256;
__global__ __launch_bounds__(QP_THREADS) void k_queue_prefix(int first, int count, const int* cnt, int* off) {
    __shared__ int s_w[16];  // totals of the sixteen (slab, wavefront) groups of a 1024-item turn, in item order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int acc = 0;
    for (int t0 = 0; t0 < count; t0 += 4 * QP_THREADS) {
        int v[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = t0 + q * QP_THREADS + tid;
            v[q] = it < count ? cnt[first + it] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[q] = wave_incl_scan(v[q]);
            if (lane == 63) s_w[q * 4 + wave] = x[q];
        }
        __syncthreads();
        int tile = 0, base[4] = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int t = s_w[w];
            tile += t;
#pragma unroll
            for (int q = 0; q < 4; ++q) base[q] += w < q * 4 + wave ? t : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = t0 + q * QP_THREADS + tid;
            if (it < count) off[it] = acc + base[q] + x[q] - v[q];
        }
        acc += tile;
        __syncthreads();
    }
    if (tid == 0) off[count] = acc;
}
// entry e of the concatenated queues -> (slot, place in the slot's queue)
__device__ __forceinline__ int queue_slot(const int* off, int count, int e, int& k) {
    int lo = 0, hi = count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= e)
            lo = mid;
        else
            hi = mid;
    }
    k = e - off[lo];
    return lo;
}

// the points whose float pre-decisions were not certain, one per lane, with the full decision chain
__device__ __forceinline__ void stencil_redo_entry(const FeatParams& P, int b, unsigned entry) {
    {
        const int line = (int)(entry >> 24), i = (int)(entry & 0xffffffu);  // (line, index inside the line): an inner point, 5 <= i < n - 5
        const size_t pos = (size_t)b * P.NT + P.line_start[(size_t)b * P.L + line] + i;  // line order
        const float4* slot_pts = P.ln_pts + (size_t)b * P.NT;
        float4 q[11];
        int qp[11];
        seg_point_window<5>(P, b, line, i, qp);
#pragma unroll
        for (int k = 0; k < 11; ++k) q[k] = slot_pts[qp[k]];
        unsigned attr = 0;
        float curv = 0.f, refl = 0.f;
        bool brk = false;
        stencil_point<false, true>(WinRegs{q}, attr, curv, refl, brk);
        P.ln_curv[pos] = curv;
        P.ln_refl[pos] = refl;
        P.ln_attr[pos] = (uint16_t)(attr | (P.ln_attr[pos] & A_VIS));  // (the walk's bit is k_stencil's)
        if (brk) P.brk_queue[(size_t)b * P.NT + atomicAdd(&P.brk_cnt[b], 1)] = entry;
    }
}
__global__ __launch_bounds__(256) void k_stencil_redo(FeatParams P) {  // per-slot grid (blocks, slots)
    const int b = blockIdx.y + P.first;
    const int cnt = P.redo_cnt[b];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) stencil_redo_entry(P, b, P.redo_queue[(size_t)b * P.NT + e]);
}
__global__ __launch_bounds__(64) void k_stencil_redo_list(FeatParams P, int count, const int* off) {  // one list per launch
    const int total = off[count];
    for (int e = blockIdx.x * 64 + threadIdx.x; e < total; e += gridDim.x * 64) {
        int k;
        const int b = P.first + queue_slot(off, count, e, k);
        stencil_redo_entry(P, b, P.redo_queue[(size_t)b * P.NT + k]);
    }
}

// the queued break-point candidates (:651-806), one per lane
__device__ __forceinline__ void stencil_break_entry(const FeatParams& P, int b, unsigned entry) {
    const float thBreakCornerDis = 1;
    {
        const int line = (int)(entry >> 24), i = (int)(entry & 0xffffffu);
        const size_t pos = (size_t)b * P.NT + P.line_start[(size_t)b * P.L + line] + i;  // line order
        const float4* slot_pts = P.ln_pts + (size_t)b * P.NT;
        float4 q[7];
        int qp[7];
        seg_point_window<3>(P, b, line, i, qp);
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = slot_pts[qp[k]];
#define PT(o) q[3 + (o)]
        unsigned f5 = 0;
        {
            float dX1 = PT(1).x - PT(0).x, dY1 = PT(1).y - PT(0).y, dZ1 = PT(1).z - PT(0).z;
            float dX2 = PT(-1).x - PT(0).x, dY2 = PT(-1).y - PT(0).y, dZ2 = PT(-1).z - PT(0).z;
            const float sq_right = dX1 * dX1 + dY1 * dY1 + dZ1 * dZ1, sq_left = dX2 * dX2 + dY2 * dY2 + dZ2 * dZ2;
            bool f100 = false;
            // two distances below 1 m cannot differ by more than thBreakCornerDis = 1 (sqrtf is monotone and
            // sqrtf(x) <= 1 for x < 1): the square roots are only taken when a neighbour is farther away than that
            float diff_right0 = 0.f, diff_left0 = 0.f;
            const bool may_break = !(sq_right < 1.f && sq_left < 1.f);
            if (may_break) {
                diff_right0 = sqrtf(sq_right);
                diff_left0 = sqrtf(sq_left);
            }
            if (may_break && fabsf(diff_right0 - diff_left0) > thBreakCornerDis) {
                float depth_right = sqrtf(PT(1).x * PT(1).x + PT(1).y * PT(1).y + PT(1).z * PT(1).z);
                float depth_left = sqrtf(PT(-1).x * PT(-1).x + PT(-1).y * PT(-1).y + PT(-1).z * PT(-1).z);
                if (diff_right0 > diff_left0) {
                    D3 surf_vector = d3(PT(-1).x - PT(0).x, PT(-1).y - PT(0).y, PT(-1).z - PT(0).z);
                    D3 lidar_vector = d3(PT(0).x, PT(0).y, PT(0).z);
                    double cc = fabs(ddot(surf_vector, lidar_vector) / (dnorm(surf_vector) * dnorm(lidar_vector)));
                    if (cc < 0.95) {
                        if (depth_right > depth_left)
                            f100 = true;
                        else if (depth_right == 0)
                            f100 = true;
                    }
                } else {
                    D3 surf_vector = d3(PT(1).x - PT(0).x, PT(1).y - PT(0).y, PT(1).z - PT(0).z);
                    D3 lidar_vector = d3(PT(0).x, PT(0).y, PT(0).z);
                    double cc = fabs(ddot(surf_vector, lidar_vector) / (dnorm(surf_vector) * dnorm(lidar_vector)));
                    if (cc < 0.95) {
                        if (depth_right < depth_left)
                            f100 = true;
                        else if (depth_left == 0)
                            f100 = true;
                    }
                }
            }
            if (f100) {
                D3 norm_front = d3(0, 0, 0), norm_back = d3(0, 0, 0);
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    float temp_depth = sqrtf(PT(-k).x * PT(-k).x + PT(-k).y * PT(-k).y + PT(-k).z * PT(-k).z);
                    if (temp_depth < 1) continue;
                    D3 tmp = d3(PT(-k).x - PT(0).x, PT(-k).y - PT(0).y, PT(-k).z - PT(0).z);
                    dnormalize(tmp);
                    norm_front.x += (k / 6.0) * tmp.x;
                    norm_front.y += (k / 6.0) * tmp.y;
                    norm_front.z += (k / 6.0) * tmp.z;
                }
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    // the reference tests the depth of i-k here as well (:782-784)
                    float temp_depth = sqrtf(PT(-k).x * PT(-k).x + PT(-k).y * PT(-k).y + PT(-k).z * PT(-k).z);
                    if (temp_depth < 1) continue;
                    D3 tmp = d3(PT(k).x - PT(0).x, PT(k).y - PT(0).y, PT(k).z - PT(0).z);
                    dnormalize(tmp);
                    norm_back.x += (k / 6.0) * tmp.x;
                    norm_back.y += (k / 6.0) * tmp.y;
                    norm_back.z += (k / 6.0) * tmp.z;
                }
                double cc = fabs(ddot(norm_front, norm_back) / (dnorm(norm_front) * dnorm(norm_back)));
                f5 = cc < 0.95 ? 1u : 2u;
            }
        }
        if (f5) P.ln_attr[pos] = (uint16_t)(P.ln_attr[pos] | (f5 << A_F5_SHIFT));
#undef PT
    }
}
__global__ __launch_bounds__(256) void k_stencil_break(FeatParams P) {  // per-slot grid (blocks, slots)
    const int b = blockIdx.y + P.first;
    const int cnt = P.brk_cnt[b];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cnt; e += gridDim.x * 256) stencil_break_entry(P, b, P.brk_queue[(size_t)b * P.NT + e]);
}
__global__ __launch_bounds__(256) void k_stencil_break_list(FeatParams P, int count, const int* off) {  // one list per launch
    const int total = off[count];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        int k;
        const int b = P.first + queue_slot(off, count, e, k);
        stencil_break_entry(P, b, P.brk_queue[(size_t)b * P.NT + k]);
    }
}

// ---- a4 + a5 + a6 walk + a8 emit: one 256-thread workgroup per scan line ---------------------------------------
// Partition bounds of :454-455.
__device__ __forceinline__ void partition_bounds(int n, int j, int& sp, int& ep) {
    const int scanStartInd = 5, scanEndInd = n - 6;
    sp = scanStartInd + (scanEndInd - scanStartInd) * j / 50;
    ep = scanStartInd + (scanEndInd - scanStartInd) * (j + 1) / 50 - 1;
}

// The reference sorts every partition twice with a stable insertion sort (strict `<`, :453-479) and then visits the
// points of a line in the order (partition, curvature rank): flag 3 + neighbour marks (:483-519), then per partition
// the promotion to 2 / 300 (:521-539).  Nothing downstream needs the sorted arrays themselves, only ORDER
// COMPARISONS, so the sort is never materialised: the visiting order of point i is the key
// (partition(i), bits(curvature[i]), i) -- curvature is a non-negative float, so its bit pattern orders like the
// value and the index breaks ties exactly as the stable sort does; the reflect order uses the usual sign-flipped
// bit pattern of diffR.  Written as data flow:
//   * whether a candidate is picked depends only on the picked status of candidates with a LOWER visiting key whose
//     mark range covers it (|distance| <= 3): a DAG, resolved by rounds of "decide everything whose predecessors
//     are decided".  Flags written by :521-539 never feed back (they only touch already visited points).
//   * the value a point holds when :521-539 reads it, the promotion (first flag-3 in curvature order unless a grazing
//     / far pick came earlier, all far flag-3 and all grazing points; the first three reflect candidates in reflect
//     order -> 300) and later overwrites by the next partition's marks are closed-form given the keys; the only
//     place two different orders are compared (`300` written before or after the point's own visit) is resolved by
//     counting ranks on demand for the <= 3 reflect picks of a partition.
//   * the stride-1-or-4 walk of :543-650 is resolved per 64-point window for each of the 4 possible entry offsets,
//     then chained across windows.
// Bit-identical to the serial form (checked against the oracle on every test line).  Lines that do not fit the LDS
// budget run the same code on a global-memory scratch.
constexpr int SELP_THREADS = 512;
// per-point word W1 (owned and written by the point's thread only; neighbours read it):
//   [0:5] partition  [6:7] a  [8:9] b  [10] cand  [11] inpart  [12] angle  [13] far  [14] refl candidate
//   [15:20] static predecessor mask (which of the 6 neighbours can suppress me)  [21:22] state
//   [23:25] f3a | covLater<<2   [26:29] inB, eff3, G, b_first
enum : unsigned { I_PART_MASK = 63u, I_A_SHIFT = 6, I_B_SHIFT = 8, I_CAND = 1u << 10, I_INPART = 1u << 11,
                  I_ANGLE = 1u << 12, I_FAR = 1u << 13, I_REFL = 1u << 14, I_MASK_SHIFT = 15, I_ST_SHIFT = 21,
                  I_ST_MASK = 3u << 21, I_F_SHIFT = 23, I_X_SHIFT = 26 };
enum : unsigned { ST_N = 0, ST_U = 1, ST_S = 2 };

__device__ __forceinline__ bool covers(unsigned info_j, int d /* i - j */) {
    const int a = (info_j >> I_A_SHIFT) & 3, bb = (info_j >> I_B_SHIFT) & 3;
    return d > 0 ? d <= a : -d <= bb;
}
__device__ __forceinline__ unsigned refl_key(float r) {
    r = r + 0.0f;  // -0 -> +0 (they compare equal in the reference's `<`)
    const unsigned u = __float_as_uint(r);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// visiting order: (partition, curvature bits, index)
__device__ __forceinline__ bool visits_before(unsigned info_j, unsigned key_j, int j, unsigned info_i, unsigned key_i, int i) {
    const unsigned pj = info_j & I_PART_MASK, pi = info_i & I_PART_MASK;
    if (pj != pi) return pj < pi;
    if (key_j != key_i) return key_j < key_i;
    return j < i;
}
__device__ __forceinline__ unsigned st_of(unsigned w) { return (w >> I_ST_SHIFT) & 3u; }
// visits_before for neighbours: j = i + D with D a compile-time constant, so the index tie-break is the sign of D and
// the rest is one 64-bit comparison of (partition, curvature bits)
template <int D>
__device__ __forceinline__ bool nb_visits_before_me(unsigned info_j, unsigned key_j, unsigned info_i, unsigned key_i) {
    const unsigned long long kj = ((unsigned long long)(info_j & I_PART_MASK) << 32) | key_j;
    const unsigned long long ki = ((unsigned long long)(info_i & I_PART_MASK) << 32) | key_i;
    return D < 0 ? kj <= ki : kj < ki;
}
template <int D>
__device__ __forceinline__ bool me_visits_before_nb(unsigned info_i, unsigned key_i, unsigned info_j, unsigned key_j) {
    const unsigned long long kj = ((unsigned long long)(info_j & I_PART_MASK) << 32) | key_j;
    const unsigned long long ki = ((unsigned long long)(info_i & I_PART_MASK) << 32) | key_i;
    return D > 0 ? ki <= kj : ki < kj;
}
template <int Q>
constexpr int nb_dist() {  // neighbour slot q -> distance: -3, -2, -1, 1, 2, 3
    return Q < 3 ? Q - 3 : Q - 2;
}
// covers(info_j, -D): does neighbour j = i + D mark me?  (a: how far it marks towards higher indices, b: lower)
template <int D>
__device__ __forceinline__ bool nb_covers_me(unsigned info_j) {
    return D < 0 ? (unsigned)(-D) <= ((info_j >> I_A_SHIFT) & 3u) : (unsigned)D <= ((info_j >> I_B_SHIFT) & 3u);
}
// the six {key, info} records at distance -3,-2,-1,1,2,3 of point i, fetched together (one wait instead of six)
template <typename WP>
__device__ __forceinline__ void load_nb(WP W, int i, uint2 (&o)[6]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int d = q < 3 ? q - 3 : q - 2;
        o[q].x = W[2 * (i + d)];
        o[q].y = W[2 * (i + d) + 1];
    }
}
template <typename WP>
__device__ __forceinline__ void load_nb_info(WP W, int i, unsigned (&o)[6]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int d = q < 3 ? q - 3 : q - 2;
        o[q] = W[2 * (i + d) + 1];
    }
}

#ifdef MML_SEL_TIMING
__device__ unsigned long long g_sel_dbg[64];
#define SEL_MARK(id)                                                                  \
    do {                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == MML_SEL_TIMING && blockIdx.y == 517) g_sel_dbg[id] = clock64(); \
    } while (0)
extern "C" int mml_debug_sel_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sel_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#else
#define SEL_MARK(id)
#endif

// phase 1: can neighbour slot Q suppress me?   phase 2: does picked neighbour Q mark me (before my visit / from a
// later partition)?
template <int Q>
__device__ __forceinline__ unsigned p1_bit(const uint2 o, unsigned me, unsigned mk) {
    constexpr int D = nb_dist<Q>();
    const bool c = ((o.y & I_CAND) != 0) & nb_covers_me<D>(o.y) & nb_visits_before_me<D>(o.y, o.x, me, mk);
    return c ? 1u << Q : 0u;
}
template <int Q>
__device__ __forceinline__ void p2_acc(const uint2 o, unsigned me, unsigned mk, bool sel, int mypart, bool& covL, bool& covLater) {
    constexpr int D = nb_dist<Q>();
    const bool hit = (st_of(o.y) == ST_S) & nb_covers_me<D>(o.y);
    const bool later = (int)(o.y & I_PART_MASK) > mypart;
    covLater |= hit & later;
    covL |= hit & !later & (!sel | me_visits_before_nb<D>(me, mk, o.y, o.x));
}

// W: interleaved {key, W1} pairs (8 bytes per point).  R: reflect order keys (4 bytes per point).
// K > 0: the line fits the LDS budget (n <= K * SELP_THREADS).  Point i = tid + k * SELP_THREADS belongs to the same
// thread in every phase, so its attribute word and fused-cloud index are fetched ONCE, all loads in flight together,
// and live in registers afterwards: the kernel is bound by the latency of dependent phases, not by bytes, and every
// global load removed from a phase removes a full HBM round trip from the critical path of the workgroup.
// K == 0: any length, per-point state in a global scratch, attributes re-read where needed.
// ---- a8 without a pass of its own: union_cloud.msg counts and the label lists straight from the selection kernels ---------
// Until round 4 k_crop swept the label bytes of a slot again (one workgroup per slot, 0.10 ms per 1024 scans, 40 us of the one-scan
// chain) to count the labels and to list the labelled points for the voxel filter.  Both consumers of the lists (k_voxel, the
// global-sort filter) order a voxel's points by the FUSED INDEX in their sort keys, so the order of a list is free: the wavefront
// that decides a label appends it.  fu_info[b] = {fused points, velo points | velo corner, velo surf | livox corner, livox surf (kept
// or beyond far_th, :925-940) | kept corner, kept surf}: words 0 .. 1 are written and 2 .. 7 zeroed by the bucketing's table kernel;
// words 6 / 7 double as the lists' fill counters.  One 64-bit atomic per pair of counters.
// `labs`: NW labels of this lane, 4 bits each (bit 3: Livox point beyond far_th: counted, not listed); pos[u]: storage position.
// (gi: the fused indices of the lane's NW points where the caller holds them; nullptr: read again for the labelled ones)
template <int NW>
__device__ __forceinline__ void label_append(const FeatParams& P, int b, bool velo_line, unsigned labs, const int (&pos)[NW], const int* gi = nullptr) {
    static_assert(NW <= 8, "4 bits per label");
    // per lane: how many of its NW labels are corner (nibble 1), surf (2), far corner (9), far surf (10); bit 2 is never set
    const unsigned m = 0x11111111u;
    const unsigned is1 = labs & ~(labs >> 1) & ~(labs >> 3) & m, is2 = (labs >> 1) & ~labs & ~(labs >> 3) & m;
    const unsigned far = (labs >> 3) & m;
    const unsigned long long anyl = __ballot((is1 | is2 | far) != 0u);
    if (anyl == 0ull) return;  // (wave-uniform)
    const int lane = threadIdx.x & 63;
    int* info = P.fu_info + 8 * (size_t)b;
    // a lane's entries take consecutive places of the lists: one scan of the packed (corner | surf << 16) counts over the lanes
    // instead of a ballot per label kind and point
    // (the active lanes are a prefix of the wavefront -- all 64 except in the tail iteration of k_select's uncached loop, where a
    //  wavefront's lanes hold ascending indices: the scan reads lower lanes only, the total sits in the last ACTIVE lane)
    const int last = 63 - __clzll((long long)__ballot(true));
    const int mine = (int)(__popc(is1) | (__popc(is2) << 16));
    const int incl = wave_incl_scan(mine);
    const int total = __builtin_amdgcn_readlane(incl, last);
    int fc = 0, fs = 0;
    if (__ballot(far != 0u)) {  // (rare: labelled Livox points beyond far_th are counted, not listed)
        const int fm = (int)(__popc(labs & far) | (__popc((labs >> 1) & far) << 16));
        const int ft = __builtin_amdgcn_readlane(wave_incl_scan(fm), last);
        fc = ft & 0xffff;
        fs = ft >> 16;
    }
    const int nc = total & 0xffff, ns = total >> 16;
    unsigned long long base = 0;
    if (lane == 0) {
        if (total) base = atomicAdd(reinterpret_cast<unsigned long long*>(info + 6), ((unsigned long long)(unsigned)ns << 32) | (unsigned)nc);
        atomicAdd(reinterpret_cast<unsigned long long*>(info + (velo_line ? 2 : 4)),
                  ((unsigned long long)(unsigned)(ns + fs) << 32) | (unsigned)(nc + fc));
    }
    if (total == 0) return;
    const int excl = incl - mine;
    int dc = __builtin_amdgcn_readfirstlane((int)(unsigned)base) + (excl & 0xffff);
    int ds = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) + (excl >> 16);
    unsigned* list_c = P.label_idx + ((size_t)b * 2 + 0) * P.label_cap;
    unsigned* list_s = P.label_idx + ((size_t)b * 2 + 1) * P.label_cap;
    int* glist_c = P.label_gidx + ((size_t)b * 2 + 0) * P.label_cap;
    int* glist_s = P.label_gidx + ((size_t)b * 2 + 1) * P.label_cap;
    const int* gx = P.ln_gidx + (size_t)b * P.NT;
    if (mine) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if ((is1 >> (4 * u)) & 1u) {
                if (dc < P.label_cap) {
                    list_c[dc] = (unsigned)pos[u];
                    glist_c[dc] = gi ? gi[u] : gx[pos[u]];
                }
                ++dc;
            } else if ((is2 >> (4 * u)) & 1u) {
                if (ds < P.label_cap) {
                    list_s[ds] = (unsigned)pos[u];
                    glist_s[ds] = gi ? gi[u] : gx[pos[u]];
                }
                ++ds;
            }
        }
    }
}

template <int K, typename WP>
__device__ __forceinline__ void select_body(const FeatParams& P, int b, int line, int n, size_t base, WP W, WP R, int* s_sp,
                                            unsigned long long (*s_pm)[3], unsigned long long* s_minE,
                                            unsigned long long* s_minG, unsigned char* s_bfirst,
                                            unsigned char* s_list, int* s_cnt, int* s_flag, unsigned short* s_walk) {
    constexpr bool CACHED = K > 0;
    constexpr int KK = CACHED ? K : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint16_t* attr = P.ln_attr + base;
    const float* curv = P.ln_curv + base;
    const float* refl = P.ln_refl + base;
    // (fused indices and labels live at the points' STORAGE positions: translated through the line's segment table)
    const int* gidx = P.ln_gidx + (size_t)b * P.NT;
    const SegTab seg = seg_tab(P, b, line);
    const int4* seg_rec = P.seg_rw + (size_t)b * P.seg_rstride + seg_rec_base(P, b, line);  // aligned windows (k_seg_records)
    // storage position of line index i: one record per 64 indices (the lanes of a wavefront share it), the segment table for the
    // runs of a sparse line
    auto xlate = [&](int i) -> int {
        const int4 rr = seg_rec[i >> 6];
        return rr.w ? i + (i < rr.x ? rr.y : rr.z) : seg_xlate(seg, i);
    };
    unsigned r_attr[KK];
    int r_gidx[KK];
    (void)r_attr;
    (void)r_gidx;
// every point of the line, always by the same thread; `k` is a compile-time constant in the cached form
#define FOR_POINTS(...)                                          \
    if constexpr (CACHED) {                                      \
        _Pragma("unroll") for (int k = 0; k < KK; ++k) {         \
            const int i = tid + k * SELP_THREADS;                \
            if (i >= n) break;                                   \
            __VA_ARGS__                                          \
        }                                                        \
    } else {                                                     \
        for (int i = tid; i < n; i += SELP_THREADS) {            \
            constexpr int k = 0;                                 \
            (void)k;                                             \
            __VA_ARGS__                                          \
        }                                                        \
    }
#define ATTR(i, k) (CACHED ? r_attr[k] : (unsigned)attr[i])
#define RKEY(i) (CACHED ? (unsigned)R[i] : refl_key(refl[i]))

    SEL_MARK(0);
    int T = 2;  // thNumCurvSize as the last stencil iteration (i = n-6) left it (:492,505)
    unsigned at_last = 0;
    if (n >= 11) at_last = attr[n - 6];
    float r_curv[KK], r_refl[KK];
    (void)r_curv;
    (void)r_refl;
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int i = min(tid + k * SELP_THREADS, n - 1);  // clamped: unconditional, independent loads
            r_attr[k] = attr[i];
            r_curv[k] = curv[i];
            r_refl[k] = refl[i];
            r_gidx[k] = gidx[xlate(i)];
        }
    }
    if (n >= 11) T = (at_last & A_W2) ? 2 : 3;
    const int range = n - 11;
    for (int j = tid; j <= 50; j += SELP_THREADS) {
        int sp, ep;
        partition_bounds(n, j, sp, ep);
        s_sp[j] = sp;  // s_sp[50] = n - 6 = one past the last partition
    }
    for (int t = tid; t < 50 * 3; t += SELP_THREADS) (&s_pm[0][0])[t] = ~0ull;
    for (int t = tid; t < 50; t += SELP_THREADS) {
        s_minE[t] = ~0ull;
        s_minG[t] = ~0ull;
    }
    if (tid == 0) *s_cnt = 0;
    if (tid < 3) s_flag[tid] = 0;
    __syncthreads();

    SEL_MARK(1);
    // ---- phase 0: per-point record --------------------------------------------------------------------------
    const float inv_r = range >= 1 ? 1.0f / (float)range : 0.f;
    FOR_POINTS(
        unsigned inf = 0, kk = 0;
        if (range >= 1 && i >= 5 && i <= n - 7) {
            const unsigned at = ATTR(i, k);
            // partition of i: the last j with sp[j] = 5 + range * j / 50 <= i (partitions tile [5, n-7]; an empty one has
            // sp[j+1] == sp[j]), i.e. j = (50 (i - 4) - 1) / range, capped at 49.  The quotient through a float reciprocal
            // (the numerator is below 2^24, the estimate is off by at most one) and one exact correction.
            const int num = 50 * (i - 4) - 1;
            int j = (int)((float)num * inv_r);
            const int rem = num - j * range;
            j += rem < 0 ? -1 : (rem >= range ? 1 : 0);
            j = j > 49 ? 49 : j;
            const unsigned a = min((int)((at >> A_A3_SHIFT) & 3u), T), bb = min((int)((at >> A_B3_SHIFT) & 3u), T);
            inf = (unsigned)j | (a << I_A_SHIFT) | (bb << I_B_SHIFT) | I_INPART;
            if (at & A_ANGLE) inf |= I_ANGLE;
            if (at & A_FAR) inf |= I_FAR;
            if (at & A_REFL) inf |= I_REFL;
            kk = __float_as_uint(CACHED ? r_curv[k] : curv[i]);
            if (at & A_CAND3) inf |= I_CAND | (ST_U << I_ST_SHIFT);
        }
        W[2 * i] = kk;
        W[2 * i + 1] = inf;
        if constexpr (CACHED) R[i] = refl_key(r_refl[k]);
    )
    if (tid < 6) {  // three zero records on either side of the line: neighbour reads need no bounds checks
        const int j = tid < 3 ? tid - 3 : n + tid - 3;
        W[2 * j] = 0;
        W[2 * j + 1] = 0;
    }
    __syncthreads();

    SEL_MARK(2);
    constexpr bool MASKED = CACHED && KK <= 8;  // (the two mask tables of the masked form fit the 2 KB of s_walk)
    if constexpr (MASKED) {
        // The dependency rounds on bit masks.  A wavefront owns, for each k, the 64 consecutive points i = tid + 512 k: one
        // window.  The window's states are two 64-bit masks (picked S, undecided U) in scalar registers, rebuilt with a
        // ballot after every step; a lane reads the states of its six neighbours as a 7-bit field of the masks (three
        // bits of the adjacent windows appended at either end) and tests it against its static predecessor mask:
        //   suppressed = U & any(S-field & P),   picked = U & !any(S-field & P) & !any(U-field & P)
        // -- a dozen vector instructions per step instead of seven LDS reads and their unpacking.  A window iterates on
        // its own until nothing changes, publishes its masks, and a workgroup round lets the edges propagate.
        unsigned long long* s_wS = reinterpret_cast<unsigned long long*>(s_walk);
        unsigned long long* s_wU = s_wS + 8 * KK;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        auto uni64 = [](unsigned long long v) -> unsigned long long {
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return ((unsigned long long)hi << 32) | lo;
        };
        unsigned r_mk[KK];  // 7-bit fields over my neighbours -3..-1 (bits 0..2) and +1..+3 (bits 4..6):
                            // [0:6] predecessor of mine, [7:13] candidate whose mark range covers me, [14:20] in a later partition
        unsigned long long Sm[KK], Um[KK];
        // ---- phase 1: the static neighbour relations of every point; candidates nobody can suppress start as picked ----
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int i = tid + k * SELP_THREADS;
            const bool act = i < n;
            unsigned me = 0, mk = 0, pm = 0, cm = 0, lt = 0;
            if (act) {
                me = W[2 * i + 1];
                mk = W[2 * i];
                uint2 o[6];
                load_nb(W, i, o);
                const bool cand = (me & I_CAND) != 0;
                const int mypart = (me & I_INPART) ? (int)(me & I_PART_MASK) : (i < 5 ? -1 : 64);
                auto rel = [&](const uint2 nb, bool covers, bool nb_first, unsigned bit) {
                    const bool c = ((nb.y & I_CAND) != 0) & covers;
                    cm |= c ? bit : 0u;
                    pm |= (c & cand & nb_first) ? bit : 0u;
                    lt |= ((int)(nb.y & I_PART_MASK) > mypart) ? bit : 0u;
                };
                rel(o[0], nb_covers_me<-3>(o[0].y), nb_visits_before_me<-3>(o[0].y, o[0].x, me, mk), 1u << 0);
                rel(o[1], nb_covers_me<-2>(o[1].y), nb_visits_before_me<-2>(o[1].y, o[1].x, me, mk), 1u << 1);
                rel(o[2], nb_covers_me<-1>(o[2].y), nb_visits_before_me<-1>(o[2].y, o[2].x, me, mk), 1u << 2);
                rel(o[3], nb_covers_me<1>(o[3].y), nb_visits_before_me<1>(o[3].y, o[3].x, me, mk), 1u << 4);
                rel(o[4], nb_covers_me<2>(o[4].y), nb_visits_before_me<2>(o[4].y, o[4].x, me, mk), 1u << 5);
                rel(o[5], nb_covers_me<3>(o[5].y), nb_visits_before_me<3>(o[5].y, o[5].x, me, mk), 1u << 6);
            }
            r_mk[k] = pm | (cm << 7) | (lt << 14);
            const bool cand = (me & I_CAND) != 0;
            Sm[k] = __ballot(cand && pm == 0);
            Um[k] = __ballot(cand && pm != 0);
            if (lane == 0) {
                s_wS[k * 8 + wave] = Sm[k];
                s_wU[k * 8 + wave] = Um[k];
            }
        }
        __syncthreads();
        SEL_MARK(3);
        int rnd = 0;
        // the 7-bit field [i - 3, i + 3] of a window mask M extended by the adjacent windows' edge bits
        auto field = [&](unsigned long long M, unsigned long long Ml, unsigned long long Mr) -> unsigned {
            const unsigned long long e_lo = (M << 3) | (Ml >> 61);
            const unsigned e0 = (unsigned)e_lo, e1 = (unsigned)(e_lo >> 32), e2 = (unsigned)(M >> 61) | ((unsigned)(Mr & 7ull) << 3);
            const unsigned lo = lane < 32 ? e0 : e1, hi = lane < 32 ? e1 : e2;
            return __builtin_amdgcn_alignbit(hi, lo, lane & 31) & 0x7Fu;
        };
        for (;;) {
            bool pending = false;
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                if (Um[k] == 0) continue;  // wave-uniform
                const int w = k * 8 + wave;
                // A neighbour window publishes S before U; reading U before S therefore never shows a point that has
                // left U without its S bit (it may show both set: the S bit is then the truth and it suppresses).
                const unsigned long long Ul = w > 0 ? uni64(s_wU[w - 1]) : 0ull, Ur = w + 1 < 8 * KK ? uni64(s_wU[w + 1]) : 0ull;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long Sl = w > 0 ? uni64(s_wS[w - 1]) : 0ull, Sr = w + 1 < 8 * KK ? uni64(s_wS[w + 1]) : 0ull;
                unsigned long long S = Sm[k], U = Um[k];
                const unsigned pm = r_mk[k] & 0x7Fu;
                bool ch = false;
                for (;;) {
                    const unsigned fS = field(S, Sl, Sr), fU = field(U, Ul, Ur);
                    const bool isU = (fU >> 3) & 1u;
                    const bool anyS = (fS & pm) != 0, anyU = (fU & pm) != 0;
                    const unsigned long long toN = __ballot(isU && anyS), toS = __ballot(isU && !anyS && !anyU);
                    if (!(toN | toS)) break;
                    S |= toS;
                    U &= ~(toN | toS);
                    ch = true;
                }
                if (ch) {
                    Sm[k] = S;
                    Um[k] = U;
                    if (lane == 0) {
                        s_wS[w] = S;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // S lands before U (see the read side)
                        s_wU[w] = U;
                    }
                }
                pending |= U != 0;
            }
#ifdef MML_SEL_TIMING
            if (threadIdx.x == 0 && blockIdx.x == MML_SEL_TIMING && blockIdx.y == 517) g_sel_dbg[20] += 1;
#endif
            // No workgroup barrier between passes: a wavefront with undecided windows only waits for edge bits of its
            // neighbours, which they publish as soon as they have them (LDS is coherent across the workgroup and the
            // dependencies form a DAG, so some window can always advance); it simply looks again.
            if (!pending) break;
            __builtin_amdgcn_s_sleep(1);
        }
        (void)rnd;
        __syncthreads();
        SEL_MARK(4);
        // ---- phase 2: value held when :521-539 runs (f3a) + "a later partition marks me", from the final masks ---------
        // (a picked neighbour whose range covers me marks me 1 at its own, later visit -- it cannot have come first, or I
        //  would not be picked -- unless it sits in a later partition, whose marks land after :521-539 has read me)
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int i = tid + k * SELP_THREADS;
            if ((i & ~63) >= n) break;  // wave-uniform
            const int w = k * 8 + wave;
            const unsigned long long Sl = w > 0 ? uni64(s_wS[w - 1]) : 0ull, Sr = w + 1 < 8 * KK ? uni64(s_wS[w + 1]) : 0ull;
            const unsigned hit = field(Sm[k], Sl, Sr) & (r_mk[k] >> 7) & 0x7Fu;
            const unsigned later = (r_mk[k] >> 14) & 0x7Fu;
            const bool covLater = (hit & later) != 0, covL = (hit & ~later) != 0;
            const bool sel = (Sm[k] >> lane) & 1ull;
            if (i < n) {
                const unsigned me = W[2 * i + 1];
                const unsigned f3a = covL ? 1u : (sel ? 3u : 0u);
                const unsigned f = f3a | (covLater ? 4u : 0u);
                // the word is only read by its owner from here on
                W[2 * i + 1] = me | (f << I_F_SHIFT);
                // (a) first round of the reflect-candidate minimum
                if ((me & I_REFL) && (me & I_INPART))
                    atomicMin(&s_pm[me & I_PART_MASK][0], ((unsigned long long)RKEY(i) << 32) | (unsigned)i);
            }
        }
        __syncthreads();
    } else {
    // ---- phase 1: which neighbours can suppress me (static); points nobody can suppress are picked at once ------
    FOR_POINTS(
        const uint2 rec = make_uint2(W[2 * i], W[2 * i + 1]);
        const unsigned me = rec.y;
        if (!(me & I_CAND)) continue;
        const unsigned mk = rec.x;
        uint2 o[6];
        load_nb(W, i, o);
        const unsigned m = p1_bit<0>(o[0], me, mk) | p1_bit<1>(o[1], me, mk) | p1_bit<2>(o[2], me, mk) | p1_bit<3>(o[3], me, mk) |
                           p1_bit<4>(o[4], me, mk) | p1_bit<5>(o[5], me, mk);
        W[2 * i + 1] = m ? (me | (m << I_MASK_SHIFT)) : ((me & ~I_ST_MASK) | (ST_S << I_ST_SHIFT));
    )
    __syncthreads();
    SEL_MARK(3);
    unsigned pend = 0;  // cached form: which of my points are still undecided (skips the LDS read of the others)
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int i = tid + k * SELP_THREADS;
            if (i < n && st_of(W[2 * i + 1]) == ST_U) pend |= 1u << k;
        }
    }
    int rnd = 0;
    // one decision step for point i: returns true when the point got decided
    auto decide = [&](int i) -> bool {
        const unsigned me = W[2 * i + 1];
        if (st_of(me) != ST_U) return true;
        unsigned o[6];
        load_nb_info(W, i, o);
        // the six neighbour states packed 2 bits each, tested against the predecessor mask spread to the same layout
        unsigned ps = 0;
        _Pragma("unroll") for (int q = 0; q < 6; ++q) ps |= ((o[q] >> I_ST_SHIFT) & 3u) << (2 * q);
        unsigned m = (me >> I_MASK_SHIFT) & 63u;
        m = (m | (m << 4)) & 0x0F0Fu;
        m = (m | (m << 2)) & 0x3333u;
        m = (m | (m << 1)) & 0x5555u;                 // bit q -> bit 2q
        const bool anyU = (ps & m) != 0;              // ST_U = 01
        const bool anyS = (ps & (m << 1)) != 0;       // ST_S = 10
        if (anyS) {
            W[2 * i + 1] = (me & ~I_ST_MASK) | (ST_N << I_ST_SHIFT);
            return true;
        }
        if (!anyU) {
            W[2 * i + 1] = (me & ~I_ST_MASK) | (ST_S << I_ST_SHIFT);
            return true;
        }
        return false;
    };
    for (;;) {
        int undecided = 0;
        if constexpr (CACHED) {
            // The 64 lanes of a wavefront hold 64 consecutive points, and a point only waits for neighbours at most 3
            // away: most dependency chains live inside one wavefront and are followed there, a few steps per k,
            // without a workgroup barrier (LDS operations of one wavefront execute in program order).
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                const int i = tid + k * SELP_THREADS;
                bool mine = i < n && ((pend >> k) & 1u);
                for (int rep = 0; rep < 4; ++rep) {
                    if (!__any(mine)) break;
                    bool changed = false;
                    if (mine && decide(i)) {
                        mine = false;
                        changed = true;
                        pend &= ~(1u << k);
                    }
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("" ::: "memory");  // neighbours' words must be re-read in the next step
                    if (!__any(changed)) break;
                }
                undecided |= mine ? 1 : 0;
            }
        } else {
            for (int i = tid; i < n; i += SELP_THREADS)
                if (!decide(i)) undecided = 1;
        }
#ifdef MML_SEL_TIMING
        if (threadIdx.x == 0 && blockIdx.x == MML_SEL_TIMING && blockIdx.y == 517) g_sel_dbg[20] += 1;
#endif
        // one barrier per round: any wavefront with an undecided point raises the round's flag (three slots, the one
        // two rounds ahead is cleared after the barrier, when nobody reads or writes it)
        if (__any(undecided) && lane == 0) s_flag[rnd % 3] = 1;
        __syncthreads();
        const int again = s_flag[rnd % 3];
        if (tid == 0) s_flag[(rnd + 2) % 3] = 0;
        ++rnd;
        if (!again) break;
    }

    SEL_MARK(4);
    // ---- phase 2: value held when :521-539 runs (f3a) + "a later partition marks me" ------------------------------
    FOR_POINTS(
        const uint2 rec = make_uint2(W[2 * i], W[2 * i + 1]);
        const unsigned me = rec.y;
        const unsigned mk = rec.x;
        const bool sel = st_of(me) == ST_S;
        const int mypart = (me & I_INPART) ? (int)(me & I_PART_MASK) : (i < 5 ? -1 : 64);
        uint2 o[6];
        load_nb(W, i, o);
        bool covL = false, covLater = false;
        p2_acc<0>(o[0], me, mk, sel, mypart, covL, covLater);
        p2_acc<1>(o[1], me, mk, sel, mypart, covL, covLater);
        p2_acc<2>(o[2], me, mk, sel, mypart, covL, covLater);
        p2_acc<3>(o[3], me, mk, sel, mypart, covL, covLater);
        p2_acc<4>(o[4], me, mk, sel, mypart, covL, covLater);
        p2_acc<5>(o[5], me, mk, sel, mypart, covL, covLater);
        const unsigned f3a = covL ? 1u : (sel ? 3u : 0u);
        const unsigned f = f3a | (covLater ? 4u : 0u);
        // the word is only read by its owner from here on
        W[2 * i + 1] = me | (f << I_F_SHIFT);
        // (a) first round of the reflect-candidate minimum
        if ((me & I_REFL) && (me & I_INPART))
            atomicMin(&s_pm[me & I_PART_MASK][0], ((unsigned long long)RKEY(i) << 32) | (unsigned)i);
    )
    __syncthreads();
    }

    SEL_MARK(5);
    // ---- phase 3: :521-539 in closed form ---------------------------------------------------------------------------
    // (a) the three first reflect candidates in reflect order, per partition (rounds 2 and 3)
    for (int round = 1; round < 3; ++round) {
        FOR_POINTS(
            const unsigned me = W[2 * i + 1];
            if (!(me & I_REFL) || !(me & I_INPART)) continue;
            const int j = me & I_PART_MASK;
            const unsigned long long rk = ((unsigned long long)RKEY(i) << 32) | (unsigned)i;
            if (rk <= s_pm[j][round - 1]) continue;
            atomicMin(&s_pm[j][round], rk);
        )
        __syncthreads();
    }
    SEL_MARK(6);
    // (a2) + (b).  (a2): for each of the <= 3 reflect picks of a partition, does its reflect visit (B_k, k = reflect rank)
    //      come before its own curvature visit (A_k)?  Only read for a pick that holds flag 3 or is a grazing point.
    //      (b): eff3 / G bits, first point of each class in curvature order per partition.
    if constexpr (CACHED) {
        // the wavefront that owns such a pick counts both ranks over the pick's partition on the spot (two ballots per
        // 64 points), so the answer never leaves the owner's registers: no list, no extra barriers
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            const int i = tid + k * SELP_THREADS;
            if ((i & ~63) >= n) break;  // wave-uniform
            const bool act = i < n;
            const unsigned me = act ? (unsigned)W[2 * i + 1] : 0u;
            const unsigned mk = act ? (unsigned)W[2 * i] : 0u;
            const int j = me & I_PART_MASK;
            bool inB = false;
            unsigned myr = 0;
            if ((me & I_INPART) && (me & I_REFL)) {
                myr = R[i];
                const unsigned long long rk = ((unsigned long long)myr << 32) | (unsigned)i;
                inB = (rk == s_pm[j][0]) | (rk == s_pm[j][1]) | (rk == s_pm[j][2]);
            }
            const bool need = inB && ((((me >> I_F_SHIFT) & 3u) == 3u) || (me & I_ANGLE));
            bool b_first = false;
            unsigned long long bm = __ballot(need);
            while (bm) {
                const int src = (int)__ffsll((long long)bm) - 1;
                bm &= bm - 1;
                const int pj = __builtin_amdgcn_readlane(j, src), pi = __builtin_amdgcn_readlane(i, src);  // (wave-uniform source lane)
                const unsigned pk = (unsigned)__builtin_amdgcn_readlane((int)mk, src), pr = (unsigned)__builtin_amdgcn_readlane((int)myr, src);
                const int sp = s_sp[pj], ep = s_sp[pj + 1] - 1;
                int rc = 0, rr = 0;
                for (int q0 = sp; q0 <= ep; q0 += 64) {
                    const int q = q0 + lane;
                    const bool in = q <= ep;
                    const unsigned kq = in ? (unsigned)W[2 * q] : 0u, rq = in ? (unsigned)R[q] : 0u;
                    rc += __popcll(__ballot(in && ((kq < pk) || (kq == pk && q < pi))));
                    rr += __popcll(__ballot(in && ((rq < pr) || (rq == pr && q < pi))));
                }
                if (lane == src) b_first = rr < rc;
            }
            if (!(me & I_INPART)) continue;
            const bool eff3 = (((me >> I_F_SHIFT) & 3u) == 3u) && !(inB && b_first);
            const bool G = (me & I_ANGLE) || (eff3 && (me & I_FAR));
            if (inB || eff3 || G) {
                const unsigned bits = (inB ? 1u : 0u) | (eff3 ? 2u : 0u) | (G ? 4u : 0u) | (b_first ? 8u : 0u);
                W[2 * i + 1] = me | (bits << I_X_SHIFT);
                const unsigned long long ck = ((unsigned long long)mk << 32) | (unsigned)i;
                if (eff3) atomicMin(&s_minE[j], ck);
                if (G) atomicMin(&s_minG[j], ck);
            }
        }
    } else {
        // (a2) for each of the <= 3 reflect picks of a partition: does its reflect visit (B_k, k = reflect rank) come
        //      before its own curvature visit (A_k)?  One wavefront per pick counts both ranks over the partition.
        //      b_first is only read for a pick that holds flag 3 or is a grazing point (see (b) and phase 5): those few
        //      are listed first, one lane per pick.
        if (tid < 150) {
            const unsigned long long e = s_pm[tid / 3][tid % 3];
            if (e != ~0ull) {
                const unsigned wi = W[2 * (int)(unsigned)e + 1];
                if (((wi >> I_F_SHIFT) & 3u) == 3u || (wi & I_ANGLE)) s_list[atomicAdd(s_cnt, 1)] = (unsigned char)tid;
            }
        }
        __syncthreads();
        const int n_need = *s_cnt;
        for (int u = (tid >> 6); u < n_need; u += SELP_THREADS / 64) {
            const int t = s_list[u];
            const int j = t / 3, r = t % 3;
            const unsigned long long e = s_pm[j][r];
            const int i = (int)(unsigned)e;
            const int sp = s_sp[j], ep = s_sp[j + 1] - 1;
            const unsigned mk = W[2 * i];
            const unsigned mr = (unsigned)(e >> 32);
            int rc = 0, rr = 0;
            for (int q = sp + lane; q <= ep; q += 64) {
                const unsigned kq = W[2 * q], rq = RKEY(q);
                rc += (kq < mk) || (kq == mk && q < i);
                rr += (rq < mr) || (rq == mr && q < i);
            }
    #pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                rc += __shfl_xor(rc, o);
                rr += __shfl_xor(rr, o);
            }
            if (lane == 0) s_bfirst[t] = (unsigned char)(rr < rc);
        }
        __syncthreads();
        SEL_MARK(7);
        // (b) eff3 / G bits, first point of each class in curvature order per partition
        FOR_POINTS(
            const unsigned me = W[2 * i + 1];
            if (!(me & I_INPART)) continue;
            const int j = me & I_PART_MASK;
            bool inB = false, b_first = false;
            if (me & I_REFL) {
                const unsigned long long rk = ((unsigned long long)RKEY(i) << 32) | (unsigned)i;
                _Pragma("unroll") for (int r = 0; r < 3; ++r)
                    if (rk == s_pm[j][r]) {
                        inB = true;
                        b_first = ((((me >> I_F_SHIFT) & 3u) == 3u) || (me & I_ANGLE)) && s_bfirst[j * 3 + r];
                    }
            }
            const bool eff3 = (((me >> I_F_SHIFT) & 3u) == 3u) && !(inB && b_first);
            const bool G = (me & I_ANGLE) || (eff3 && (me & I_FAR));
            if (inB || eff3 || G) {
                const unsigned bits = (inB ? 1u : 0u) | (eff3 ? 2u : 0u) | (G ? 4u : 0u) | (b_first ? 8u : 0u);
                W[2 * i + 1] = me | (bits << I_X_SHIFT);
                const unsigned long long ck = ((unsigned long long)W[2 * i] << 32) | (unsigned)i;
                if (eff3) atomicMin(&s_minE[j], ck);
                if (G) atomicMin(&s_minG[j], ck);
            }
        )
    }
    __syncthreads();  // R is dead from here on: the window tables of the cached form live in its storage

    SEL_MARK(8);
    // (phase 4, the stride walk of :543-650, is resolved by k_stencil: attribute bit A_VIS)
    SEL_MARK(10);
    // ---- phase 5: final value of the serial part (:521-539 (c)), overrides (150, 100/101), emit, label scatter ---------
    uint8_t* lnlab = P.ln_label + (size_t)b * P.NT;
    // (the labels this thread decides, 4 bits each, and where they went: appended to the slot's lists below -- label_append)
    constexpr int NCH = (KK + 7) / 8;
    unsigned labs_arr[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) labs_arr[c] = 0;
    FOR_POINTS(
        const unsigned me = W[2 * i + 1];
        const unsigned at = ATTR(i, k);
        int f = (me >> I_F_SHIFT) & 3u;
        if (me & I_INPART) {
            const int j = me & I_PART_MASK;
            const unsigned bits = (me >> I_X_SHIFT) & 15u;
            if (bits) {
                const bool inB = bits & 1, eff3 = bits & 2, G = bits & 4, b_first = bits & 8;
                const unsigned long long ck = ((unsigned long long)W[2 * i] << 32) | (unsigned)i;
                const bool first = eff3 && ck == s_minE[j] && !(s_minG[j] < s_minE[j]);
                const bool picked = G || first;
                if (inB)
                    f = (picked && (me & I_ANGLE) && b_first) ? 2 : 300;
                else if (picked)
                    f = 2;
            }
        }
        if ((me >> I_F_SHIFT) & 4u) f = 1;  // marked by a point of a later partition (:503,516 of the next partitions)
        const bool inner = i >= 5 && i < n - 5;
        if (inner) {
            // visited by the stride walk and the included-angle test passes (the test implies both half-windows flat)
            if ((at & (A_VIS | A_C150)) == (A_VIS | A_C150)) f = 150;
            const unsigned f5 = (at >> A_F5_SHIFT) & 3u;
            if (f5 == 1) f = 100;
            if (f5 == 2) f = 101;
        }
        if (P.ln_final) P.ln_final[base + i] = (uint16_t)f;
        int labv = 0;
        if (inner && !(at & A_NEAR)) {
            const int lab = (f == 2) ? 2 : ((f == 100 || f == 150) ? 1 : 0);
            if (lab) {
                const int gi = CACHED ? r_gidx[k] : gidx[xlate(i)];
                if (gi >= 0)
                    labv = lab;
                else if (gi == -2)  // Livox point beyond far_th: not in the fused cloud, but counted at :925-940 and part of the
                    labv = lab | 0x80;  // surf cloud the GICP refresh aligns (:296-312); k_crop counts these, nobody lists them
            }
        }
        const int ps = xlate(i);
        lnlab[ps] = (uint8_t)labv;  // every point of the line: nobody clears the label bytes beforehand
        const unsigned code = (unsigned)((labv & 3) | ((labv & 0x80) ? 8 : 0));
        if constexpr (CACHED) {
            labs_arr[k / 8] |= code << (4 * (k % 8));
        } else {
            // (a wavefront's lanes hold ascending i: lane 0 is live whenever any lane is)
            const int p1[1] = {ps};
            label_append<1>(P, b, line < P.n_rings, code, p1);
        }
    )
    if constexpr (CACHED) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int pp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)  // (positions again from the records: kept in registers they cost k_select<4> a wavefront per SIMD)
                pp[u] = ((labs_arr[c] >> (4 * u)) & 3u) ? xlate(tid + (8 * c + u) * SELP_THREADS) : 0;
            label_append<8>(P, b, line < P.n_rings, labs_arr[c], pp);
        }
    }
    SEL_MARK(11);
#undef FOR_POINTS
#undef ATTR
#undef RKEY
}

// LDS of the cached form: W pairs with 4 pad records either side | R (reflect keys; later the window tables)
__host__ __device__ inline size_t select_lds_bytes(int cap) { return (size_t)(cap + 8) * 8 + (size_t)cap * 4; }

// list_kind < 0: one workgroup per (line, slot) of the grid.  list_kind 0 / 1: behind k_select_part, the (normally empty) list of the
// ring / Livox lines that kernel left over, walked by a small grid -- a full grid of workgroups that only find their line done
// still has to wait for 512 wave slots and 29-53 KB of LDS each, and held its stream for longer than the real work took.
#ifndef MML_SEL_ROWS_V
#define MML_SEL_ROWS_V 256
#define MML_SEL_ROWS_L 128
#endif
// (K <= 4, the ring variants of the 16-ring layout: held to the 128 registers of four wavefronts per SIMD, which they needed
//  127 of before the label lists were appended here and 130 after)
template <int K>
__global__ __launch_bounds__(SELP_THREADS) __attribute__((amdgpu_waves_per_eu(K <= 4 ? 4 : 1))) void k_select(FeatParams P, int list_kind) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long s_pm[50][3];
    __shared__ unsigned long long s_minE[50], s_minG[50];
    __shared__ int s_sp[52];
    __shared__ int s_cnt;
    __shared__ int s_flag[3];
    __shared__ __attribute__((aligned(8))) unsigned short s_walk[1024];
    // only the global-scratch form (lines beyond the LDS budget) uses these two; its dynamic LDS block is otherwise idle
    unsigned char* s_bfirst = smem;
    unsigned char* s_list = smem + 160;
    const int n_list = list_kind < 0 ? 1 : P.sel_list_cnt[2 * P.first + list_kind];
    const int* list = P.sel_list + 2 * ((size_t)(list_kind < 0 ? 0 : list_kind) * P.B + P.first) * P.L;
    for (int e = list_kind < 0 ? 0 : (int)blockIdx.x; e < n_list; e += gridDim.x) {
        const int b = list_kind < 0 ? (int)blockIdx.y + P.first : list[2 * e];
        const int line = list_kind < 0 ? (int)blockIdx.x + P.line0 : list[2 * e + 1];
        const int n = P.line_len[(size_t)b * P.L + line];
        if (n <= 0) continue;
        if (list_kind < 0 && P.sel_done && P.sel_done[(size_t)b * P.L + line]) return;  // done by k_select_part
        __syncthreads();  // (list mode: the previous line's LDS state has been read)
        const int start = P.line_start[(size_t)b * P.L + line];
        const size_t base = (size_t)b * P.NT + start;
        constexpr int cap = K * SELP_THREADS;
        if (n <= cap) {
            unsigned* W = reinterpret_cast<unsigned*>(smem) + 8;  // 4 pad records in front, 4 behind
            unsigned* R = reinterpret_cast<unsigned*>(smem) + 2 * (size_t)(cap + 8);
            select_body<K>(P, b, line, n, base, W, R, s_sp, s_pm, s_minE, s_minG, s_bfirst, s_list, &s_cnt, s_flag, s_walk);
        } else {
            // global scratch: four 4-byte slots per bucketed point (W pairs | window tables)
            // (+8 * line + 8 words: room for the pad records of every line in front of this one)
            unsigned* W = P.sel_scratch + 2 * base + 16 * ((size_t)b * (P.L + 2) + line + 1);
            select_body<0>(P, b, line, n, base, W, static_cast<unsigned*>(nullptr), s_sp, s_pm, s_minE, s_minG, s_bfirst, s_list, &s_cnt, s_flag, s_walk);
        }
        if (list_kind < 0) break;
    }
}

// the lines of a launch that k_select_part did not take, as two lists (rings, Livox lines) for k_select
__global__ void k_select_list(FeatParams P, int count) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * P.L) return;
    const int b = P.first + t / P.L, line = t % P.L;
    if (P.line_len[(size_t)b * P.L + line] <= 0 || P.sel_done[(size_t)b * P.L + line]) return;
    const int kind = line < P.n_rings ? 0 : 1;
    const int at = atomicAdd(&P.sel_list_cnt[2 * P.first + kind], 1);
    int* list = P.sel_list + 2 * ((size_t)kind * P.B + P.first) * P.L;
    list[2 * at] = b;
    list[2 * at + 1] = line;
}

// ---- a4 + a5 + a8 once more, for lines whose partitions hold 3 .. 64 points: one partition per LANE ---------------------------
// k_select above resolves the state machine with one point per lane: every relation between a point and its six neighbours is
// recomputed by the point's lane (~450 wave-instructions per 64 points, at the issue limit).  The same relations, transposed:
// the 50 partitions of a line are 50 lanes of ONE wavefront, and everything a partition knows about its points is a set of
// 64-bit masks private to its lane -- candidate, mark ranges (a >= d, b >= d), "point k - d is visited before point k" --, so
// that one 64-bit operation is one relation of all the partition's points at once:
//   neighbour k - d can suppress k        Pm_d = ((A_d << d) | edge) & G_d & C
//   neighbour k + d can suppress k        Pp_d = (B_d >> d) & ~(G_d >> d) & C        (never from a later partition)
//   a dependency step                     anyS = OR_d Pm_d & (S << d | edge)  |  Pp_d & (S >> d);   toN = U & anyS; ...
// with three edge bits travelling from a lane to the next per step (a partition only ever waits for the one before it).
// Phase A builds the bit planes of the line with one point per lane (ballots; the order bits G_d from lane shuffles of the
// curvature keys), phase B is the partition-per-lane part (dependency steps, :521-539 in closed form as in k_select: reflect
// top-3, ranks, first promotable point, with loops over the few set bits that need a key), phase C turns the result planes back
// into flags / labels with one point per lane.  Lines outside 161 <= n <= 3211 (a partition of fewer than 3 or more than 64
// points) are left to k_select (per-line flag sel_done).
#ifdef MML_SP_TIMING
__device__ unsigned long long g_sp_dbg[16];
#define SP_MARK(id)                                                       \
    do {                                                                  \
        if (sp_dbg) {                                                     \
            const unsigned long long now_ = clock64();                    \
            g_sp_dbg[id] += now_ - sp_prev;                               \
            sp_prev = now_;                                               \
        }                                                                 \
    } while (0)
extern "C" int mml_debug_sp_timing(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[16] = {};
        return hipMemcpyToSymbol(HIP_SYMBOL(g_sp_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#else
#define SP_MARK(id)
#endif
typedef unsigned long long u64m;
typedef unsigned __int128 u128m;
enum { PL_C = 0, PL_AN, PL_FR, PL_RF, PL_A0, PL_A1, PL_B0, PL_B1, PL_G1, PL_G2, PL_G3, PL_COUNT };
enum { PL_R0 = PL_C, PL_R1 = PL_AN, PL_R2 = PL_FR };  // the result planes take the place of three input planes once those are read
__device__ __forceinline__ int m_ffs(u64m v) { return v ? (int)__ffsll((long long)v) - 1 : -1; }
__device__ __forceinline__ int m_ffs(u128m v) {
    const u64m lo = (u64m)v, hi = (u64m)(v >> 64);
    return lo ? (int)__ffsll((long long)lo) - 1 : (hi ? 64 + (int)__ffsll((long long)hi) - 1 : -1);
}
template <typename M>
__device__ __forceinline__ M m_bits(int nbits) {
    return nbits >= (int)(8 * sizeof(M)) ? ~(M)0 : (((M)1 << nbits) - (M)1);
}
// M = 64-bit masks: partitions of up to 64 points (lines of 161 .. 3211 points: the rings); M = 128-bit masks: up to 128 points
// (.. 6411: the Livox lines).  WAVES = lines (wavefronts) per workgroup, MAXWIN = 64-point windows of the longest line + 2.
// (the narrow form needs 81 vector registers left alone -- one more than six wavefronts per SIMD allow: it is asked to fit them; the
//  wide form 144: asked for four wavefronts per SIMD it spills 18 registers and is still 4 % faster, 0.497 -> 0.476 ms for the stage --
//  this kernel waits for its own memory round trips, a fourth wavefront covers more of them than the spills cost)
template <typename M, int WAVES, int MAXWIN>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(sizeof(M) > 8 ? 4 : 6))) void k_select_part(FeatParams P, int n_lines_launch) {
    constexpr int MBITS = (int)(8 * sizeof(M));
    constexpr bool WIDE = MBITS > 64;
    __shared__ u64m s_plane_all[WAVES][PL_COUNT][MAXWIN];
    const int wave_id = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y + P.first;
    const int line = __builtin_amdgcn_readfirstlane((int)(P.line0 + blockIdx.x * WAVES + wave_id));  // (a scalar: see k_stencil)
    if ((int)(blockIdx.x * WAVES + wave_id) >= n_lines_launch) return;
    const int n = P.line_len[(size_t)b * P.L + line];
    const int range = n - 11;
    const bool eligible = range >= 150 && range <= 50 * MBITS;  // every partition holds 3 .. MBITS points
    // per-line flag for k_select: the narrow form (launched first) resets it, the wide form takes what the narrow one left
    if constexpr (!WIDE) {
        if (lane == 0) P.sel_done[(size_t)b * P.L + line] = eligible ? 1 : 0;
    } else {
        if (P.sel_done[(size_t)b * P.L + line]) return;
        if (eligible && lane == 0) P.sel_done[(size_t)b * P.L + line] = 1;
    }
    if (!eligible) return;
    const int start = P.line_start[(size_t)b * P.L + line];
    const size_t base = (size_t)b * P.NT + start;
    const uint16_t* attr = P.ln_attr + base;
    const float* curv = P.ln_curv + base;
    const float* refl = P.ln_refl + base;
    u64m(*pl)[MAXWIN] = s_plane_all[wave_id];
    const int nwin = (n + 63) >> 6;
#ifdef MML_SP_TIMING
    const bool sp_dbg = lane == 0 && line == MML_SP_TIMING && blockIdx.y == 517;
    unsigned long long sp_prev = clock64();
#endif
#define SP_SYNC()                                          \
    do {                                                   \
        __builtin_amdgcn_wave_barrier();                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    } while (0)
    // ---- phase A: bit planes of the line, one point per lane ------------------------------------------------------------
    const int T = (attr[n - 6] & A_W2) ? 2 : 3;  // thNumCurvSize as the last stencil iteration left it (:492,505)
    if (lane < PL_COUNT) {                        // the zero windows behind the line
        pl[lane][nwin] = 0ull;
        pl[lane][nwin + 1] = 0ull;
    }
    unsigned tail0 = 0, tail1 = 0, tail2 = 0;  // curvature keys of the three points before the window (wave-uniform)
    // (eight windows per turn, their sixteen loads in flight together: this wavefront's life is a chain of memory round trips,
    //  and the few other wavefronts of the SIMD cover only so many of them)
    for (int w4 = 0; w4 < nwin; w4 += 8) {
        unsigned at4[8], key4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = 64 * (w4 + u) + lane;
            const bool in = i < n;
            at4[u] = in ? (unsigned)attr[i] : 0u;
            key4[u] = in ? __float_as_uint(curv[i]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w4 + u;
            if (w >= nwin) break;
            const int i = 64 * w + lane;
            const unsigned key = key4[u];
            const bool inpart = i >= 5 && i <= n - 7;
            const unsigned at = inpart ? at4[u] : 0u;  // (one select instead of `inpart &&` in front of every ballot)
            // "point i - d is visited before point i" inside a partition: key (curvature bits, index) ascending, ties to the lower
            // index.  The keys of the three points before come by whole-wave shifts (DPP wave_shr:1, the lane without a source keeps
            // `old` = the carried key of the window before): three moves instead of three LDS permutes and six selects.
            const unsigned p1 = (unsigned)__builtin_amdgcn_update_dpp((int)tail2, (int)key, 0x138, 0xf, 0xf, false);
            const unsigned p2 = (unsigned)__builtin_amdgcn_update_dpp((int)tail1, (int)p1, 0x138, 0xf, 0xf, false);
            const unsigned p3 = (unsigned)__builtin_amdgcn_update_dpp((int)tail0, (int)p2, 0x138, 0xf, 0xf, false);
            // mark ranges as the raw two-bit fields of the stencil: phase B masks them with the candidate plane and applies the
            // clamp to T there (a >= 3 only exists when T == 3)
            const u64m m_c = __ballot(at & A_CAND3), m_an = __ballot(at & A_ANGLE), m_fr = __ballot(at & A_FAR), m_rf = __ballot(at & A_REFL),
                       m_a0 = __ballot(at & (1u << A_A3_SHIFT)), m_a1 = __ballot(at & (2u << A_A3_SHIFT)),
                       m_b0 = __ballot(at & (1u << A_B3_SHIFT)), m_b1 = __ballot(at & (2u << A_B3_SHIFT)), m_g1 = __ballot(p1 <= key),
                       m_g2 = __ballot(p2 <= key), m_g3 = __ballot(p3 <= key);
            if (lane == 0) {
                pl[PL_C][w] = m_c;
                pl[PL_AN][w] = m_an;
                pl[PL_FR][w] = m_fr;
                pl[PL_RF][w] = m_rf;
                pl[PL_A0][w] = m_a0;
                pl[PL_A1][w] = m_a1;
                pl[PL_B0][w] = m_b0;
                pl[PL_B1][w] = m_b1;
                pl[PL_G1][w] = m_g1;
                pl[PL_G2][w] = m_g2;
                pl[PL_G3][w] = m_g3;
            }
            tail0 = (unsigned)__builtin_amdgcn_readlane((int)key, 61);
            tail1 = (unsigned)__builtin_amdgcn_readlane((int)key, 62);
            tail2 = (unsigned)__builtin_amdgcn_readlane((int)key, 63);
        }
    }
    SP_SYNC();
    SP_MARK(0);
    // ---- phase B: one partition per lane ------------------------------------------------------------------------------------
    {
        const bool act = lane < 50;
        const int pj = act ? lane : 49;
        const int sp = 5 + range * pj / 50, spn = 5 + range * (pj + 1) / 50, L = spn - sp;  // 3 <= L <= MBITS
        const M maskL = m_bits<M>(L);
        const int w0 = sp >> 6, off = sp & 63;
        auto extract = [&](int plane) -> M {
            const u64m x0 = pl[plane][w0], x1 = pl[plane][w0 + 1];
            M v = (M)((x0 >> off) | (off ? (x1 << (64 - off)) : 0ull));
            if constexpr (WIDE) {
                const u64m x2 = pl[plane][w0 + 2];
                v |= (M)((x1 >> off) | (off ? (x2 << (64 - off)) : 0ull)) << 64;
            }
            return act ? (v & maskL) : (M)0;
        };
        const M C = extract(PL_C), AN = extract(PL_AN), FR = extract(PL_FR), RF = extract(PL_RF);
        const M a0 = extract(PL_A0) & C, a1 = extract(PL_A1) & C, b0 = extract(PL_B0) & C, b1 = extract(PL_B1) & C;
        const M t3 = T == 3 ? ~(M)0 : (M)0;  // min(mark range, thNumCurvSize): a range of 3 only under T == 3
        const M A1 = a0 | a1, A2 = a1, A3 = a1 & a0 & t3, B1 = b0 | b1, B2 = b1, B3 = b1 & b0 & t3;
        // order bits: inside the partition from the keys; a point of the partition before is always visited first
        const M G1 = extract(PL_G1) | (M)1, G2 = extract(PL_G2) | (M)3, G3 = extract(PL_G3) | (M)7;
        SP_SYNC();  // every lane has read its planes: three of them now become the (zeroed) result planes
        for (int k = lane; k < 3 * MAXWIN; k += 64) (&pl[PL_R0][0])[k] = 0ull;
        // edges: which of the LAST d points of the partition before cover my first points (a >= d), which of the FIRST d
        // points of the partition behind cover my last points (b >= d)
        const unsigned tA = (unsigned)((A1 >> (L - 1)) & (M)1) | ((unsigned)((A2 >> (L - 2)) & (M)3) << 1) | ((unsigned)((A3 >> (L - 3)) & (M)7) << 3);
        const unsigned hB = (unsigned)(B1 & (M)1) | ((unsigned)(B2 & (M)3) << 1) | ((unsigned)(B3 & (M)7) << 3);
        // (neighbour partitions by whole-wave DPP shifts: wave_shr:1 = from the lane below, wave_shl:1 = from the lane above; the lane
        //  without a source keeps `old` = 0)
        unsigned pA = (unsigned)__builtin_amdgcn_update_dpp(0, (int)tA, 0x138, 0xf, 0xf, false),
                 nB = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hB, 0x130, 0xf, 0xf, false);
        if (lane == 0 || !act) pA = 0;
        if (lane >= 49) nB = 0;
        const M inA1 = (M)(pA & 1u), inA2 = (M)((pA >> 1) & 3u), inA3 = (M)((pA >> 3) & 7u);                                 // bits 0 .. d-1
        const M inB1 = (M)(nB & 1u) << (L - 1), inB2 = (M)((nB >> 1) & 3u) << (L - 2), inB3 = (M)((nB >> 3) & 7u) << (L - 3);  // bits L-d .. L-1
        // static relations
        const M Cm1 = ((A1 << 1) | inA1) & maskL, Cm2 = ((A2 << 2) | inA2) & maskL, Cm3 = ((A3 << 3) | inA3) & maskL;
        const M Cw1 = B1 >> 1, Cw2 = B2 >> 2, Cw3 = B3 >> 3;  // covered from k + d inside the partition
        const M Pm1 = Cm1 & G1 & C, Pm2 = Cm2 & G2 & C, Pm3 = Cm3 & G3 & C;
        const M Pp1 = Cw1 & ~(G1 >> 1) & C, Pp2 = Cw2 & ~(G2 >> 2) & C, Pp3 = Cw3 & ~(G3 >> 3) & C;
        const M has_pred = Pm1 | Pm2 | Pm3 | Pp1 | Pp2 | Pp3;
        M S = C & ~has_pred, U = C & has_pred;
        SP_MARK(1);
        // dependency steps: a partition's first three points wait for the last three of the partition before
        for (int guard = 0; guard < 4096; ++guard) {
            const unsigned e = (unsigned)((S >> (L - 3)) & (M)7) | ((unsigned)((U >> (L - 3)) & (M)7) << 3);
            unsigned pe = (unsigned)__builtin_amdgcn_update_dpp(0, (int)e, 0x138, 0xf, 0xf, false);
            if (lane == 0) pe = 0;
            const M pS = (M)(pe & 7u), pU = (M)((pe >> 3) & 7u);  // bit j <-> point L' - 3 + j of the partition before
            const M S1 = (S << 1) | (pS >> 2), S2 = (S << 2) | (pS >> 1), S3 = (S << 3) | pS;
            const M U1 = (U << 1) | (pU >> 2), U2 = (U << 2) | (pU >> 1), U3 = (U << 3) | pU;
            const M anyS = (Pm1 & S1) | (Pm2 & S2) | (Pm3 & S3) | (Pp1 & (S >> 1)) | (Pp2 & (S >> 2)) | (Pp3 & (S >> 3));
            const M anyU = (Pm1 & U1) | (Pm2 & U2) | (Pm3 & U3) | (Pp1 & (U >> 1)) | (Pp2 & (U >> 2)) | (Pp3 & (U >> 3));
            const M toN = U & anyS, toS = U & ~anyS & ~anyU;
            S |= toS;
            U &= ~(toN | toS);
            if (!__any((toN | toS) != (M)0)) break;
        }
        SP_MARK(2);
        // value held when :521-539 runs: 1 when a pick of the same or an earlier partition marks me (it cannot have come before
        // my own pick, or I would not be picked), else 3 for a pick; marks from the partition behind land after :521-539
        const unsigned e2 = (unsigned)((S >> (L - 3)) & (M)7) | ((unsigned)(S & (M)7) << 3);
        unsigned pe2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)e2, 0x138, 0xf, 0xf, false),
                 ne2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)e2, 0x130, 0xf, 0xf, false);
        if (lane == 0) pe2 = 0;
        if (lane >= 49) ne2 = 0;
        const M pS = (M)(pe2 & 7u), nS = (M)((ne2 >> 3) & 7u);  // nS bit j <-> point j of the partition behind
        const M covL = (Cm1 & ((S << 1) | (pS >> 2))) | (Cm2 & ((S << 2) | (pS >> 1))) | (Cm3 & ((S << 3) | pS)) | (Cw1 & (S >> 1)) | (Cw2 & (S >> 2)) |
                       (Cw3 & (S >> 3));
        const M covLater = (inB1 & ((nS & (M)1) << (L - 1))) | (inB2 & ((nS & (M)3) << (L - 2))) | (inB3 & ((nS & (M)7) << (L - 3)));
        const M is3 = S & ~covL;
        // ---- :521-539 in closed form (as in k_select) ----
        // the three first reflect candidates in reflect order
        M inBm = (M)0, bfirst = (M)0;
        {
            u64m best[3] = {~0ull, ~0ull, ~0ull};
            M m = RF;
            while (m != (M)0) {  // four candidates per turn, their keys requested together
                int kk[4];
                float rv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    kk[u] = m_ffs(m);
                    m &= m - (M)1;
                    rv[u] = refl[sp + (kk[u] >= 0 ? kk[u] : 0)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (kk[u] < 0) continue;
                    u64m x = ((u64m)refl_key(rv[u]) << 32) | (unsigned)(sp + kk[u]);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const u64m cur = best[r];
                        const bool lt = x < cur;
                        best[r] = lt ? x : cur;
                        x = lt ? cur : x;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (best[r] != ~0ull) inBm |= (M)1 << ((int)(unsigned)best[r] - sp);
            SP_MARK(3);
            // does a pick's reflect visit come before its own curvature visit?  (only read for a pick that holds 3 or is a
            // grazing point): ranks in both orders over the whole partition.  Counted with one POINT per lane: the picks of all
            // partitions are taken four at a time -- the owning lanes broadcast partition and pick, the lanes fetch that
            // partition's keys (64 per pass), two ballots per pass give the two ranks -- instead of every owner walking its
            // partition alone, a chain of dependent loads that the whole wavefront waited for (31 % of the kernel).
            constexpr int RK = WIDE ? 4 : 8;  // picks per turn
            M need = inBm & (is3 | AN);
            for (;;) {
                const int myk = m_ffs(need);  // next pick of my own partition (if any)
                u64m owners = __ballot(myk >= 0);
                if (!owners) break;
                int src[RK], pk[RK], psp[RK], pL[RK];
                float mcv[RK], mrv[RK], cv[RK][WIDE ? 2 : 1], rv[RK][WIDE ? 2 : 1];
#pragma unroll
                for (int u = 0; u < RK; ++u) {
                    src[u] = owners ? (int)__ffsll((long long)owners) - 1 : -1;
                    owners &= owners - 1;
                    // (the owner's lane number is wave-uniform: its pick, partition start and length come by v_readlane into scalar
                    //  registers -- three LDS permutes per pick before, each a round trip in front of the loads below)
                    const int o = __builtin_amdgcn_readfirstlane(src[u] >= 0 ? src[u] : 0);
                    pk[u] = __builtin_amdgcn_readlane(myk, o);
                    psp[u] = __builtin_amdgcn_readlane(sp, o);
                    pL[u] = __builtin_amdgcn_readlane(L, o);
                    mcv[u] = curv[psp[u] + max(pk[u], 0)];
                    mrv[u] = refl[psp[u] + max(pk[u], 0)];
#pragma unroll
                    for (int h = 0; h < (WIDE ? 2 : 1); ++h) {
                        const int q = min(lane + 64 * h, pL[u] - 1);
                        cv[u][h] = curv[psp[u] + q];
                        rv[u][h] = refl[psp[u] + q];
                    }
                }
#pragma unroll
                for (int u = 0; u < RK; ++u) {
                    if (src[u] < 0) continue;  // wave-uniform
                    const unsigned mk = __float_as_uint(mcv[u]), mr = refl_key(mrv[u]);
                    int rc = 0, rr = 0;
#pragma unroll
                    for (int h = 0; h < (WIDE ? 2 : 1); ++h) {
                        const int q = lane + 64 * h;
                        const bool in = q < pL[u];
                        const unsigned kq = __float_as_uint(cv[u][h]), rq = refl_key(rv[u][h]);
                        // (key, index) pairs compared as one 64-bit number
                        const u64m mine_c = ((u64m)mk << 32) | (unsigned)pk[u], mine_r = ((u64m)mr << 32) | (unsigned)pk[u];
                        rc += __popcll(__ballot(in && ((((u64m)kq << 32) | (unsigned)q) < mine_c)));
                        rr += __popcll(__ballot(in && ((((u64m)rq << 32) | (unsigned)q) < mine_r)));
                    }
                    if (lane == src[u]) {
                        if (rr < rc) bfirst |= (M)1 << pk[u];
                        need &= need - (M)1;
                    }
                }
            }
        }
        SP_MARK(4);
        const M eff3 = is3 & ~(inBm & bfirst);
        const M Gm = AN | (eff3 & FR);
        M first = (M)0;
        {
            auto min_key = [&](M m) -> u64m {  // four keys per turn, requested together
                u64m best = ~0ull;
                while (m != (M)0) {
                    int kk[4];
                    float cv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        kk[u] = m_ffs(m);
                        m &= m - (M)1;
                        cv[u] = curv[sp + (kk[u] >= 0 ? kk[u] : 0)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u64m ck = kk[u] >= 0 ? (((u64m)__float_as_uint(cv[u]) << 32) | (unsigned)(sp + kk[u])) : ~0ull;
                        best = ck < best ? ck : best;
                    }
                }
                return best;
            };
            const u64m minE = min_key(eff3), minG = min_key(Gm);
            if (minE != ~0ull && !(minG < minE)) first = (M)1 << ((int)(unsigned)minE - sp);
        }
        SP_MARK(5);
        const M picked = Gm | first;
        const M two_after_300 = inBm & picked & AN & bfirst;
        const M F2 = ((picked & ~inBm) | two_after_300) & ~covLater;
        const M F300 = inBm & ~two_after_300 & ~covLater;
        const M F1 = covLater | (covL & ~picked & ~inBm);
        const M F3 = is3 & ~picked & ~inBm & ~covLater;
        // result planes: code 1 -> flag 1, 2 -> 2, 3 -> 3, 4 -> 300
        const M r0 = F1 | F3, r1 = F2 | F3, r2 = F300;
        // the points in front of the first and behind the last partition can only be marked (flag 1)
        unsigned head1 = 0, tail1m = 0;  // bit j <-> point sp - 1 - j / point spn + j
        if (lane == 0) {
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k + j < 3; ++k) {
                    const int d = k + j + 1;
                    const M Bd = d == 1 ? B1 : (d == 2 ? B2 : B3);
                    if ((unsigned)((S >> k) & (Bd >> k) & (M)1)) head1 |= 1u << j;
                }
        }
        if (lane == 49) {
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k + j < 3; ++k) {
                    const int d = k + j + 1;
                    const M Ad = d == 1 ? A1 : (d == 2 ? A2 : A3);
                    if ((unsigned)((S >> (L - 1 - k)) & (Ad >> (L - 1 - k)) & (M)1)) tail1m |= 1u << j;
                }
        }
        SP_SYNC();  // (the result planes have been zeroed by every lane)
        if (act) {
            auto scatter = [&](int plane, M v) {
                if (v == (M)0) return;
                const u64m lo = (u64m)v;
                if (lo << off) atomicOr(&pl[plane][w0], lo << off);
                u64m mid = off ? (lo >> (64 - off)) : 0ull;
                if constexpr (WIDE) {
                    const u64m hi = (u64m)(v >> 64);
                    mid |= hi << off;
                    const u64m top = off ? (hi >> (64 - off)) : 0ull;
                    if (top) atomicOr(&pl[plane][w0 + 2], top);
                }
                if (mid) atomicOr(&pl[plane][w0 + 1], mid);
            };
            scatter(PL_R0, r0);
            scatter(PL_R1, r1);
            scatter(PL_R2, r2);
            if (lane == 0)
                for (int j = 0; j < 3; ++j)
                    if ((head1 >> j) & 1u) atomicOr(&pl[PL_R0][(sp - 1 - j) >> 6], 1ull << ((sp - 1 - j) & 63));
            if (lane == 49)
                for (int j = 0; j < 3; ++j)
                    if ((tail1m >> j) & 1u) atomicOr(&pl[PL_R0][(spn + j) >> 6], 1ull << ((spn + j) & 63));
        }
    }
    SP_SYNC();
    SP_MARK(6);
    // ---- phase C: flags and labels, one point per lane -----------------------------------------------------------------
    // (labels and fused indices live at the points' STORAGE positions: translated through the line's segment table)
    uint8_t* lnlab = P.ln_label + (size_t)b * P.NT;
    const int* gidx = P.ln_gidx + (size_t)b * P.NT;
    const SegTab seg = seg_tab(P, b, line);
    const int4* seg_rec = seg_uniform_ptr(P.seg_rw + (size_t)b * P.seg_rstride + seg_rec_base(P, b, line));  // aligned windows (k_seg_records)
    int sh = 0;
    for (int w4 = 0; w4 < nwin; w4 += 8) {
        unsigned at4[8];
        int gi4[8], ps4[8];
        seg_v4i rr[8];  // (one scalar request per window: boundary, offsets on either side, regular?)
#pragma unroll
        for (int u = 0; u < 8; ++u) rr[u] = seg_sload(seg_rec, min(w4 + u, nwin - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) at4[u] = attr[min(64 * (w4 + u) + lane, n - 1)];
        SEG_SWAIT8(rr[0], rr[1], rr[2], rr[3], rr[4], rr[5], rr[6], rr[7]);
        bool all_ok = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(64 * (w4 + u) + lane, n - 1);
            ps4[u] = i + (i < rr[u].x ? rr[u].y : rr[u].z);
            all_ok = all_ok && rr[u].w != 0;
        }
        if (!all_ok) {  // (wave-uniform) a sparse line: through the segment table
#pragma unroll 1
            for (int u = 0; u < 8; ++u) {
                const int p = seg_xlate(seg, min(64 * (w4 + u) + lane, n - 1), sh);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q == u) ps4[q] = p;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) gi4[u] = gidx[ps4[u]];
        unsigned labs = 0;  // the labels of this lane's eight points, 4 bits each (label_append)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w4 + u, i = 64 * w + lane;
            if (w >= nwin || i >= n) continue;
            const u64m r0 = pl[PL_R0][w], r1 = pl[PL_R1][w], r2 = pl[PL_R2][w];
            const unsigned code = (unsigned)((r0 >> lane) & 1ull) | ((unsigned)((r1 >> lane) & 1ull) << 1) | ((unsigned)((r2 >> lane) & 1ull) << 2);
            int f = code == 4u ? 300 : (int)code;
            const unsigned at = at4[u];
            const bool inner = i >= 5 && i < n - 5;
            if (inner) {
                if ((at & (A_VIS | A_C150)) == (A_VIS | A_C150)) f = 150;
                const unsigned f5 = (at >> A_F5_SHIFT) & 3u;
                if (f5 == 1) f = 100;
                if (f5 == 2) f = 101;
            }
            if (P.ln_final) P.ln_final[base + i] = (uint16_t)f;
            int labv = 0;
            if (inner && !(at & A_NEAR)) {
                const int lab = (f == 2) ? 2 : ((f == 100 || f == 150) ? 1 : 0);
                if (lab) {
                    const int gi = gi4[u];
                    if (gi >= 0)
                        labv = lab;
                    else if (gi == -2)
                        labv = lab | 0x80;
                }
            }
            lnlab[ps4[u]] = (uint8_t)labv;
            labs |= (unsigned)((labv & 3) | ((labv & 0x80) ? 8 : 0)) << (4 * u);
        }
        label_append<8>(P, b, line < P.n_rings, labs, ps4, gi4);
    }
    SP_MARK(7);
#undef SP_SYNC
}
// the two forms: partitions of up to 64 points (rings), up to 128 (Livox lines)
constexpr int SP_LINES = 4, SP_LINES_WIDE = 2;
constexpr int SP_MAXWIN = 53, SP_MAXWIN_WIDE = 103;  // (3211 + 63) / 64 + 2, (6411 + 63) / 64 + 2

// K (points per thread in LDS-resident lines) variants of k_select
typedef void (*select_fn)(FeatParams, int);
static select_fn select_variant(int cap) {
    switch (cap / SELP_THREADS) {
        case 2: return k_select<2>;
        case 4: return k_select<4>;
        case 6: return k_select<6>;
        case 8: return k_select<8>;
        case 12: return k_select<12>;
        case 16: return k_select<16>;
        default: return k_select<24>;
    }
}
static int select_round_cap(int want) {
    const int ks[] = {2, 4, 6, 8, 12, 16, 24};
    for (int k : ks)
        if (k * SELP_THREADS >= want) return k * SELP_THREADS;
    return 24 * SELP_THREADS;
}

// ---- a8: removeNearFarPoints / removeNearPointCloud + compaction into the fused cloud ---------------------------
// The crop is geometric, so k_assign_c has already written every kept point to its place in the fused cloud and
// k_select has written the labels there.  What is left: the label counts of union_cloud.msg, the index lists of the
// corner- and surf-labelled points (the label split of Estimator.cpp:992-1011, consumed by the voxel down-sampler) and
// the Livox extrinsic, which the reference applies only once livox_corner_num is known (:302-318).
// One workgroup per slot: labels are 1 byte per bucketed point, so a whole scan is a few tens of KB -- counting, the scan
// of the counts and the emission of the two index lists fit one launch (two sweeps over the label bytes).  The labels
// sit at the points' line-bucketed positions: Velodyne lines in [0, cb_n[0]), Livox lines in [NV, NV + cb_n[1]) (NV is a
// multiple of 64, so both regions start on a word); the sweeps walk the concatenation of the two regions word by word.
// The lists hold bucketed positions (ascending: the Velodyne part is a prefix); the fused order the reference sums a
// voxel's points in is restored by the voxel sort, which carries the fused index in its key.
#ifndef MML_CROP_WORDS
#define MML_CROP_WORDS 16384
#endif
constexpr int CROP_THREADS = 512;  // (1024-thread workgroups wait long for wave slots next to other lanes' kernels)
__global__ __launch_bounds__(CROP_THREADS) void k_crop(FeatParams P, int cap, int lds_words) {
    __shared__ int s_w[CROP_THREADS / 64][4];
    __shared__ int s_tot[4];
    const int b = blockIdx.x + P.first;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const AssignAux* a = reinterpret_cast<const AssignAux*>(P.assign_aux) + b;
    const int nv = a->kept_velo, n = nv + a->kept_livox;
    const int cv = P.cb_n[2 * b], cl = P.cb_n[2 * b + 1];   // valid points per sensor (kept or not)
    const size_t o = (size_t)b * P.NT;
    const uint32_t* lwv = reinterpret_cast<const uint32_t*>(P.ln_label + o);
    const uint32_t* lwl = reinterpret_cast<const uint32_t*>(P.ln_label + o + P.NV);
    const int wv = (cv + 3) / 4, wl = (cl + 3) / 4, words = wv + wl;
    // The label words go through LDS in chunks of `lds_words` (coalesced loads, all of a thread's loads in flight together);
    // inside a chunk every thread owns a run of consecutive words, counts its labels, and after a workgroup scan emits the
    // positions of its corner / surf points behind the running totals of the chunks before.  (A scan whose labels fit one
    // chunk -- 52.8 k points -- makes one round; a dense 262 k-point scan used to walk its words in global memory, one
    // dependent uncoalesced load per word, twice.)
    extern __shared__ uint32_t s_lab[];
    // word w, byte q -> bucketed position (or -1 for the padding bytes behind a region)
    auto position = [&](int w, int q) -> int {
        if (w < wv) {
            const int p = 4 * w + q;
            return p < cv ? p : -1;
        }
        const int p = 4 * (w - wv) + q;
        return p < cl ? P.NV + p : -1;
    };
    int run1 = 0, run2 = 0;        // corner / surf points listed by the chunks before
    int tot[4] = {0, 0, 0, 0};     // corner, surf, velo corner, velo surf
    int far_c = 0, far_s = 0;      // this thread's labelled Livox points beyond far_th (label byte with bit 7 set)
    for (int c0 = 0; c0 < words; c0 += lds_words) {
        const int cw = min(lds_words, words - c0);
        __syncthreads();  // the previous chunk has been read
        for (int w = tid; w < cw; w += CROP_THREADS) s_lab[w] = (c0 + w) < wv ? lwv[c0 + w] : lwl[c0 + w - wv];
        __syncthreads();
        const int per = (cw + CROP_THREADS - 1) / CROP_THREADS;
        const int w0 = min(cw, tid * per), w1 = min(cw, w0 + per);
        int c[4] = {0, 0, 0, 0};
        for (int w = w0; w < w1; ++w) {
            const uint32_t v = s_lab[w];
            if (v == 0) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = position(c0 + w, q);
                const int l = (v >> (8 * q)) & 255u;
                if (p >= 0 && l) {
                    if (l & 0x80) {
                        far_c += (l & 3) == 1;
                        far_s += (l & 3) == 2;
                    } else {
                        c[l == 1 ? 0 : 1] += 1;
                        if (p < P.NV) c[l == 1 ? 2 : 3] += 1;
                    }
                }
            }
        }
        // exclusive scan over threads (corner, surf) + totals (all four)
        int inc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int x = c[k];
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            inc[k] = x;
            if (lane == 63) s_w[wave][k] = x;
        }
        __syncthreads();
        int base[2] = {0, 0};
        for (int w = 0; w < wave; ++w) {
            base[0] += s_w[w][0];
            base[1] += s_w[w][1];
        }
        int ctot[4] = {0, 0, 0, 0};
        for (int w = 0; w < CROP_THREADS / 64; ++w) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ctot[k] += s_w[w][k];
        }
        int d1 = run1 + base[0] + inc[0] - c[0], d2 = run2 + base[1] + inc[1] - c[1];
        if (c[0] | c[1]) {
            for (int w = w0; w < w1; ++w) {
                const uint32_t v = s_lab[w];
                if (v == 0) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p = position(c0 + w, q);
                    const int l = (v >> (8 * q)) & 255u;
                    if (p >= 0 && l == 1) {
                        if (d1 < cap) {
                            P.label_idx[((size_t)b * 2 + 0) * cap + d1] = (unsigned)p;
                            P.label_gidx[((size_t)b * 2 + 0) * cap + d1] = P.ln_gidx[(size_t)b * P.NT + p];
                        }
                        ++d1;
                    } else if (p >= 0 && l == 2) {
                        if (d2 < cap) {
                            P.label_idx[((size_t)b * 2 + 1) * cap + d2] = (unsigned)p;
                            P.label_gidx[((size_t)b * 2 + 1) * cap + d2] = P.ln_gidx[(size_t)b * P.NT + p];
                        }
                        ++d2;
                    }
                }
            }
        }
        run1 += ctot[0];
        run2 += ctot[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += ctot[k];
    }
    // the labelled Livox points beyond far_th: workgroup totals
    __shared__ int s_far[2];
    if (tid < 2) s_far[tid] = 0;
    if (tid < 4) s_tot[tid] = tot[tid];
    __syncthreads();
    for (int d = 32; d > 0; d >>= 1) {
        far_c += __shfl_xor(far_c, d);
        far_s += __shfl_xor(far_s, d);
    }
    if (lane == 0 && (far_c | far_s)) {
        atomicAdd(&s_far[0], far_c);
        atomicAdd(&s_far[1], far_s);
    }
    __syncthreads();
    int* info = P.fu_info + 8 * b;
    const int livox_corner = s_far[0] + s_tot[0] - s_tot[2];
    if (tid == 0) {
        info[0] = n;          // fused points
        info[1] = nv;         // ... of which velodyne
        info[2] = s_tot[2];   // velo corner / surf after near+far crop (:1287-1300)
        info[3] = s_tot[3];
        info[4] = livox_corner;                      // livox corner / surf after the near crop only (:925-940)
        info[5] = s_far[1] + s_tot[1] - s_tot[3];
        info[6] = s_tot[0];   // all kept corner- / surf-labelled points (label split, Estimator.cpp:995-1003)
        info[7] = s_tot[1];
    }
    if (P.extr != nullptr && livox_corner > 100) {  // :302-318
        // pcl::transformPointCloud (PCL 1.8.1 common/impl/transforms.hpp), float; the Livox region (points the crop
        // dropped are transformed along, nobody reads them)
        const float* e = P.extr;
        for (int p = tid; p < cl; p += CROP_THREADS) {
            float4 pt = P.ln_pts[o + P.NV + p];
            const float x = e[0] * pt.x + e[1] * pt.y + e[2] * pt.z + e[3];
            const float y = e[4] * pt.x + e[5] * pt.y + e[6] * pt.z + e[7];
            const float z = e[8] * pt.x + e[9] * pt.y + e[10] * pt.z + e[11];
            pt.x = x;
            pt.y = y;
            pt.z = z;
            P.ln_pts[o + P.NV + p] = pt;
        }
    }
}

// The Livox extrinsic of mml_extract (:302-318): pcl::transformPointCloud (PCL 1.8.1 common/impl/transforms.hpp, float) on the
// slot's Livox region once its livox_corner_num is known to exceed 100 -- i.e. after the selection kernels of the slot (points
// the crop dropped are transformed along, nobody reads them).
__global__ __launch_bounds__(256) void k_livox_extrinsic(FeatParams P) {
    const int b = blockIdx.y + P.first;
    if (!(P.fu_info[8 * (size_t)b + 4] > 100)) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P.cb_n[2 * b + 1]) return;
    const float* e = P.extr;
    const size_t o = (size_t)b * P.NT + P.NV + p;
    float4 pt = P.ln_pts[o];
    const float x = e[0] * pt.x + e[1] * pt.y + e[2] * pt.z + e[3];
    const float y = e[4] * pt.x + e[5] * pt.y + e[6] * pt.z + e[7];
    const float z = e[8] * pt.x + e[9] * pt.y + e[10] * pt.z + e[11];
    pt.x = x;
    pt.y = y;
    pt.z = z;
    P.ln_pts[o] = pt;
}

// single-line setup for mml_detect_line: slot 0 holds one line (ring 0) of n points already in ln_pts
__global__ void k_setup_single_line(FeatParams P, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < P.L) {
        P.line_start[t] = (t < P.n_rings) ? (t == 0 ? 0 : n) : P.NV;
        P.line_len[t] = (t == 0) ? n : 0;
        // one storage segment: the line is stored where its line-order index says
        P.seg_n[t] = 1;
        P.seg_cum[(size_t)t * (MML_SEG_MAX + 1)] = 0;
        P.seg_cum[(size_t)t * (MML_SEG_MAX + 1) + 1] = (t == 0) ? n : 0;
        P.seg_pos[(size_t)t * MML_SEG_MAX] = P.line_start[t];
    }
    if (t < n) {
        P.ln_gidx[t] = t;
        P.ln_rel[t] = 0;
    }
    if (t == 0) {
        P.cb_n[0] = n;
        P.cb_n[1] = 0;
        P.fu_info[0] = n;
        P.fu_info[1] = n;
#pragma unroll
        for (int q = 2; q < 8; ++q) P.fu_info[q] = 0;
    }
}

FeatParams make_params(mml_ctx* ctx, int first) {
    FeatParams P;
    P.sensor_base = 0;
    P.first = first;
    P.NV = ctx->NV;
    P.NL = ctx->NL;
    P.NT = ctx->NT;
    P.L = ctx->L;
    P.n_rings = ctx->cfg.n_rings;
    P.n_lines = ctx->cfg.n_livox_lines;
    P.pitch0 = ctx->cfg.pitch0_deg;
    P.pitch_step = ctx->cfg.pitch_step_deg;
    P.near_th = ctx->cfg.near_th;
    P.far_th = ctx->cfg.far_th;
    P.velo_in = ctx->velo_in;
    P.livox_in = ctx->livox_in;
    P.n_in = ctx->d_n_in;
    P.raw_line = ctx->raw_line;
    P.raw_ori = ctx->raw_ori;
    P.ln_pts = ctx->ln_pts;
    P.ln_gidx = ctx->ln_gidx;
    P.ln_rel = ctx->ln_rel;
    P.line_start = ctx->line_start;
    P.line_len = ctx->line_len;
    P.ln_curv = ctx->ln_curv;
    P.ln_refl = ctx->ln_refl;
    P.ln_attr = ctx->ln_attr;
    P.blk_cnt = ctx->blk_cnt;
    P.assign_aux = ctx->assign_aux;
    P.ab_ppt = ctx->cfg.n_rings > 32 ? 4 : 1;  // (dense layouts: 1024-point count blocks; CB_TILE is a multiple of either size)
    P.nblk_v = (ctx->NV + AB_THREADS * P.ab_ppt - 1) / (AB_THREADS * P.ab_ppt);
    P.nblk_l = (ctx->NL + AB_THREADS - 1) / AB_THREADS;
    P.nblk_max = P.nblk_v > P.nblk_l ? P.nblk_v : P.nblk_l;
    P.ring_bits = 1;
    while ((1 << P.ring_bits) < ctx->cfg.n_rings) ++P.ring_bits;
    P.line_bits = 1;
    while ((1 << P.line_bits) < ctx->cfg.n_livox_lines) ++P.line_bits;
    P.asb_stride = (ctx->cfg.n_rings > ctx->cfg.n_livox_lines ? ctx->cfg.n_rings : ctx->cfg.n_livox_lines) + 2;
    P.asb_stride |= 1;
    P.crop_cnt = reinterpret_cast<CropBlk*>(ctx->crop_cnt);
    P.label_idx = reinterpret_cast<unsigned*>(ctx->vx_keys);
    P.label_gidx = ctx->vx_gidx;
    P.label_cap = ctx->VX_CAP;
    P.nblk_t = (ctx->NT + 255) / 256;
    P.sel_scratch = ctx->sel_scratch;
    P.sel_cap = ctx->sel_cap;
    P.line0 = 0;
    P.B = ctx->B;
    P.ln_final = nullptr;
    P.cb_n = ctx->cb_n;
    P.ln_line = ctx->ln_line;
    P.slot_flags = ctx->slot_flags;
    P.ln_label = ctx->ln_label;
    P.fu_info = ctx->fu_info;
    P.extr = nullptr;
    P.brk_queue = ctx->brk_queue;
    P.brk_cnt = ctx->brk_cnt;
    P.redo_queue = ctx->redo_queue;
    P.redo_cnt = ctx->brk_cnt + ctx->B;
    P.sel_done = ctx->select_part ? ctx->sel_done : nullptr;
    P.sel_list = ctx->sel_list;
    P.sel_list_cnt = ctx->sel_list_cnt;
    P.st_exit = ctx->st_exit;
    P.st_stride = ctx->NT / 256 + ctx->L + 8;
    P.seg_cum = ctx->seg_cum;
    P.seg_pos = ctx->seg_pos;
    P.seg_n = ctx->seg_n;
    P.seg_flat = ctx->seg_flat;
    P.seg_flat_n = ctx->seg_flat_n;
    P.op_agg = ctx->op_agg;
    P.op_epoch = 0;
    P.ends_inline = 0;
    P.seg_rs = ctx->seg_rs;
    P.seg_rw = ctx->seg_rw;
    P.seg_rstride = ctx->seg_rstride;
    {
        static int remap = -1;
        if (remap < 0) remap = getenv("MML_XCD_REMAP") ? atoi(getenv("MML_XCD_REMAP")) : 1;
        P.xcd_remap = remap;
    }
    return P;
}

}  // namespace

extern "C" int mml_libm_f32(mml_ctx* ctx, const float* y, const float* x, long n, float* out_atan2, float* out_atan) {
    if (!ctx || !y || !x || n < 0) return MML_ERR_INVALID;
    if (n == 0) return MML_OK;
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    float *dy = nullptr, *dx = nullptr, *d2 = nullptr, *d1 = nullptr;
    auto release = [&]() {
        if (dy) (void)hipFree(dy);
        if (dx) (void)hipFree(dx);
        if (d2) (void)hipFree(d2);
        if (d1) (void)hipFree(d1);
    };
    const size_t bytes = sizeof(float) * (size_t)n;
    if (hipMalloc(&dy, bytes) != hipSuccess || hipMalloc(&dx, bytes) != hipSuccess ||
        (out_atan2 && hipMalloc(&d2, bytes) != hipSuccess) || (out_atan && hipMalloc(&d1, bytes) != hipSuccess)) {
        release();
        return MML_ERR_HIP;
    }
    bool ok = hipMemcpy(dy, y, bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_libm_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, MML_STREAM(ctx), dy, dx, n, d2, d1);
        ok = hipStreamSynchronize(MML_STREAM(ctx)) == hipSuccess;
    }
    if (ok && out_atan2) ok = hipMemcpy(out_atan2, d2, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && out_atan) ok = hipMemcpy(out_atan, d1, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    release();
    return ok ? MML_OK : MML_ERR_HIP;
}

int mml_launch_extract(mml_ctx* ctx, int first, int count, bool have_extrinsic) {
    for (int i = 0; i < count; ++i) ctx->raw_extracted[first + i] = 1;  // (mml_gicp_refresh re-derives line ids from the raw buffers)
    FeatParams P = make_params(ctx, first);
    P.extr = have_extrinsic ? ctx->d_extr : nullptr;
    hipStream_t s = MML_STREAM(ctx);
    if (ctx->onepass) {
        // one pass over the raw scan: 4096-point blocks that keep their points in registers between counting and storing
        ctx->op_epoch = (ctx->op_epoch + 1) & 0x1ffffffu;
        if (ctx->op_epoch == 0) ctx->op_epoch = 1;  // (0 is what freshly allocated aggregate words hold)
        P.op_epoch = ctx->op_epoch;
        const int nbv = (ctx->NV + MML_OP_BLK - 1) / MML_OP_BLK, nbl = (ctx->NL + MML_OP_BLK - 1) / MML_OP_BLK;
        P.ends_inline = count <= ST_SEGMENT_MAX_SLOTS ? 1 : 0;
        if (!P.ends_inline) {
            MmlStageScope t(ctx, "assign_ends");
            hipLaunchKernelGGL(k_assign_ends, dim3(count), dim3(64), 0, s, P, count);
        }
        {
            MmlStageScope t(ctx, "assign_onepass");
            const int nb = nbv > nbl ? nbv : nbl;
            if (P.ends_inline) {
                if (nb > 0) hipLaunchKernelGGL(k_assign_onepass, dim3(nb, count, nbl > 0 ? 2 : 1), dim3(OP_THREADS), 0, s, P);
            } else {
                if (nbv > 0) hipLaunchKernelGGL(k_assign_onepass_s<0>, dim3(nbv, count), dim3(OP_THREADS), 0, s, P);
                if (nbl > 0) hipLaunchKernelGGL(k_assign_onepass_s<1>, dim3(nbl, count), dim3(OP_THREADS), 0, s, P);
            }
        }
        {
            MmlStageScope t(ctx, "assign_tables");
            hipLaunchKernelGGL(k_assign_tables, dim3(count), dim3(TB_THREADS), 0, s, P, count);
        }
    } else {
    {
        MmlStageScope t(ctx, "assign_count");
        hipLaunchKernelGGL(k_assign_init, dim3((count + 255) / 256), dim3(256), 0, s, P, count);
        if (P.ab_ppt == 4)
            hipLaunchKernelGGL((k_assign_a<4>), dim3(P.nblk_max, count, 2), dim3(AB_THREADS), 0, s, P);
        else
            hipLaunchKernelGGL((k_assign_a<1>), dim3(P.nblk_max, count, 2), dim3(AB_THREADS), 0, s, P);
    }
    {
        MmlStageScope t(ctx, "assign_scan");
        hipLaunchKernelGGL(k_assign_b, dim3(count, 2), dim3(ASB_THREADS), sizeof(int) * 64 * (size_t)P.asb_stride, s, P);
        hipLaunchKernelGGL(k_seg_records, dim3((ctx->seg_rstride + 255) / 256, count), dim3(256), 0, s, P, count);
    }
    {
        MmlStageScope t(ctx, "assign_scatter");
        if (P.n_rings > 32) {  // dense scans: consecutive raw points belong to different rings
            P.sensor_base = 0;
            hipLaunchKernelGGL(k_assign_c_staged, dim3((ctx->NV + CB_TILE - 1) / CB_TILE, count, 1), dim3(CB_THREADS), 0, s, P);
            if (P.nblk_l > 0) {  // (a context without a Livox region, max_livox_points = 0: a zero-sized grid is not a launch)
                P.sensor_base = 1;
                hipLaunchKernelGGL(k_assign_c_direct, dim3(P.nblk_l, count, 1), dim3(AB_THREADS), 0, s, P);
                P.sensor_base = 0;
            }
        } else {
            hipLaunchKernelGGL(k_assign_c_direct, dim3(P.nblk_max, count, 2), dim3(AB_THREADS), 0, s, P);
        }
    }
    }
    {
        MmlStageScope t(ctx, "stencil");
        if (count > ST_SEGMENT_MAX_SLOTS) {  // one wavefront per scan line
            hipLaunchKernelGGL(k_stencil<0>, dim3(count, (ctx->L + ST_LINES - 1) / ST_LINES), dim3(64 * ST_LINES), 0, s, P);
        } else {  // a handful of scans: one wavefront per tile, two launches
            const int nominal = 2 * (ctx->NT / (ctx->L > 0 ? ctx->L : 1)) + ST_TILE;
            const dim3 grid((nominal + ST_TILE - 1) / ST_TILE, (ctx->L + ST_LINES - 1) / ST_LINES, count);
            hipLaunchKernelGGL(k_stencil<1>, grid, dim3(64 * ST_LINES), 0, s, P);
            hipLaunchKernelGGL(k_stencil<2>, grid, dim3(64 * ST_LINES), 0, s, P);
        }
        if (count > ST_SEGMENT_MAX_SLOTS) {
            // one list per launch and queue (offsets sliced like work_off: by first slot and lane); the grids are sized for twice the
            // usual fill (0.2 % / 2 % of the points) and stride over whatever more there is
            int* off_r = ctx->queue_off + (size_t)first + ctx->cur;
            int* off_b = off_r + (ctx->B + mml_ctx::MAX_LANES + 1);
            const long pts = (long)count * ctx->NT;
            const int g_r = (int)std::min<long>(std::max<long>(pts / 250 / 64, 64), 1 << 16);
            const int g_b = (int)std::min<long>(std::max<long>(pts / 25 / 256, 64), 1 << 16);
            hipLaunchKernelGGL(k_queue_prefix, dim3(1), dim3(QP_THREADS), 0, s, first, count, P.redo_cnt, off_r);
            hipLaunchKernelGGL(k_stencil_redo_list, dim3(g_r), dim3(64), 0, s, P, count, off_r);
            hipLaunchKernelGGL(k_queue_prefix, dim3(1), dim3(QP_THREADS), 0, s, first, count, P.brk_cnt, off_b);  // (the redo pass appends)
            hipLaunchKernelGGL(k_stencil_break_list, dim3(g_b), dim3(256), 0, s, P, count, off_b);
        } else {
            hipLaunchKernelGGL(k_stencil_redo, dim3(4, count), dim3(256), 0, s, P);
            hipLaunchKernelGGL(k_stencil_break, dim3(2, count), dim3(256), 0, s, P);
        }
    }
    {
        MmlStageScope t(ctx, "select");
        // rings and Livox lines have different nominal lengths: each group runs the variant whose LDS block fits it, so the
        // short rings do not pay (in occupancy) for the long Livox lines
        FeatParams Pv = P;
        Pv.sel_cap = ctx->sel_cap_velo;
        FeatParams Pl = P;
        Pl.line0 = ctx->cfg.n_rings;
        // (a handful of scans -- the live one-scan call -- cannot fill the device with one wavefront per line: there the
        //  workgroup-per-line kernel is the shorter chain, 0.054 against 0.132 ms for one scan)
        const bool part = ctx->select_part && count > ST_SEGMENT_MAX_SLOTS;
        if (!part) {
            Pv.sel_done = nullptr;
            Pl.sel_done = nullptr;
        }
        if (part) {
            // lines whose partitions hold 3 .. 64 / .. 128 points: one partition per lane; what is left over (short, very long or
            // ragged lines) is listed and goes through k_select, a small grid walking the two lists
            // (the two list counters of the launch were zeroed by the bucketing: k_assign_tables / k_assign_init)
            hipLaunchKernelGGL((k_select_part<u64m, SP_LINES, SP_MAXWIN>), dim3((ctx->L + SP_LINES - 1) / SP_LINES, count), dim3(64 * SP_LINES), 0, s,
                               P, ctx->L);
            hipLaunchKernelGGL((k_select_part<u128m, SP_LINES_WIDE, SP_MAXWIN_WIDE>), dim3((ctx->L + SP_LINES_WIDE - 1) / SP_LINES_WIDE, count),
                               dim3(64 * SP_LINES_WIDE), 0, s, P, ctx->L);
            hipLaunchKernelGGL(k_select_list, dim3((count * ctx->L + 255) / 256), dim3(256), 0, s, P, count);
            const int rows_v = std::min(ctx->cfg.n_rings * count, MML_SEL_ROWS_V), rows_l = std::min((ctx->L - ctx->cfg.n_rings) * count, MML_SEL_ROWS_L);
            hipLaunchKernelGGL(select_variant(ctx->sel_cap_velo), dim3(rows_v), dim3(SELP_THREADS), select_lds_bytes(ctx->sel_cap_velo), s, Pv, 0);
            if (rows_l > 0)
                hipLaunchKernelGGL(select_variant(ctx->sel_cap), dim3(rows_l), dim3(SELP_THREADS), select_lds_bytes(ctx->sel_cap), s, Pl, 1);
        } else if (count <= ST_SEGMENT_MAX_SLOTS && ctx->L > ctx->cfg.n_rings && ctx->sel_cap >= ctx->sel_cap_velo) {
            // a handful of scans: rings and Livox lines in ONE launch of the long-line variant -- its larger LDS block costs a
            // device this empty nothing, and the rings' 25 us run under the Livox lines' 42 instead of in front of them (the
            // variant only sets how many points a thread holds: the same flags come out, tests/test_gpu_shapes.py (3))
            FeatParams Pa = P;
            Pa.sel_done = nullptr;
            hipLaunchKernelGGL(select_variant(ctx->sel_cap), dim3(ctx->L, count), dim3(SELP_THREADS), select_lds_bytes(ctx->sel_cap), s, Pa, -1);
        } else {
            hipLaunchKernelGGL(select_variant(ctx->sel_cap_velo), dim3(ctx->cfg.n_rings, count), dim3(SELP_THREADS),
                               select_lds_bytes(ctx->sel_cap_velo), s, Pv, -1);
            if (ctx->L > ctx->cfg.n_rings)
                hipLaunchKernelGGL(select_variant(ctx->sel_cap), dim3(ctx->L - ctx->cfg.n_rings, count), dim3(SELP_THREADS),
                                   select_lds_bytes(ctx->sel_cap), s, Pl, -1);
        }
    }
    // (no crop pass: the label counts of union_cloud.msg and the label lists were written by the selection kernels, label_append)
    if (P.extr != nullptr && ctx->NL > 0) {
        MmlStageScope t(ctx, "livox_extrinsic");
        hipLaunchKernelGGL(k_livox_extrinsic, dim3((ctx->NL + 255) / 256, count), dim3(256), 0, s, P);
    }
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// ---- mml_cloud_upload back end: a labelled fused cloud that arrived over /union_feature_cloud ------------------------
// pcl::fromROSMsg<PointXYZINormal> of velo_combine / livox_combine (unionPoseEstimation.cpp:679-688) on the device:
// 48-byte records (x 0, y 4, z 8, normal_x 16 in-sweep time, normal_y 20 ring / line, normal_z 24 label, intensity 32)
// into the slot's fused cloud, labels by the tests of Estimator.cpp:995-1003 (abs(normal_z - k) < 1e-5), then the
// bookkeeping extract would have left behind (point counts for k_crop, which rebuilds the label counts and lists).
__global__ void k_decode_xyzinormal(const float* raw, int n, int n_velo, int slot, FeatParams P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        AssignAux* a = reinterpret_cast<AssignAux*>(P.assign_aux) + slot;
        a->kept_velo = n_velo;
        a->kept_livox = n - n_velo;
        P.slot_flags[2 * slot] = 1;       // uploaded: intensities and line ids are the caller's; normal_x as given
        P.cb_n[2 * slot] = n_velo;        // every uploaded point is a valid, kept point of its sensor's region
        P.cb_n[2 * slot + 1] = n - n_velo;
        int* info = P.fu_info + 8 * slot;
        info[4] = 0;
        info[5] = 0;
    }
    if (i >= n) return;
    const float* r = raw + 12 * (size_t)i;
    // fused index i -> storage position: the Velodyne part from 0, the Livox part from NV
    const size_t o = (size_t)slot * P.NT + (i < n_velo ? i : P.NV + (i - n_velo));
    P.ln_pts[o] = make_float4(r[0], r[1], r[2], r[8]);
    P.ln_gidx[o] = i;
    P.ln_rel[o] = __float_as_int(r[4]);
    const float ln = r[5], nz = r[6];
    P.ln_line[o] = (uint8_t)(ln >= 0.f && ln < 255.f ? (int)ln : 255);
    // std::abs(normal_z - 1.0) < 1e-5 etc. are evaluated in double on the float field
    uint8_t lab = 0;
    if (fabs((double)nz - 1.0) < 1e-5) lab = 1;
    else if (fabs((double)nz - 2.0) < 1e-5) lab = 2;
    P.ln_label[o] = lab;
}

int mml_launch_cloud_decode(mml_ctx* ctx, int slot, const float* d_raw, int n, int n_velo) {
    FeatParams P = make_params(ctx, slot);
    hipStream_t s = MML_STREAM(ctx);
    hipLaunchKernelGGL(k_decode_xyzinormal, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, s, d_raw, n, n_velo, slot, P);
    const int lab_words = (ctx->NT + 3) / 4;
    const int lds_words = lab_words <= MML_CROP_WORDS ? lab_words : MML_CROP_WORDS;
    hipLaunchKernelGGL(k_crop, dim3(1), dim3(CROP_THREADS), sizeof(uint32_t) * (size_t)lds_words, s, P, ctx->VX_CAP, lds_words);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// ring / line id of every raw point of one slot into raw_line[] (pass A of the three-pass bucketing on that slot; its block
// histograms land in blk_cnt, which nobody reads afterwards)
int mml_launch_raw_lines(mml_ctx* ctx, int slot) {
    FeatParams P = make_params(ctx, slot);
    if (P.ab_ppt == 4)
        hipLaunchKernelGGL((k_assign_a<4>), dim3(P.nblk_max, 1, 2), dim3(AB_THREADS), 0, MML_STREAM(ctx), P);
    else
        hipLaunchKernelGGL((k_assign_a<1>), dim3(P.nblk_max, 1, 2), dim3(AB_THREADS), 0, MML_STREAM(ctx), P);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

// mml_detect_line back end: pts already copied to ln_pts[0..n) of slot 0; results in ln_label[0..n) and ln_final.
int mml_launch_detect_line(mml_ctx* ctx, int n, uint16_t* d_final) {
    FeatParams P = make_params(ctx, 0);
    P.ln_final = d_final;
    hipStream_t s = MML_STREAM(ctx);
    const int t = (n > ctx->L ? n : ctx->L);
    hipLaunchKernelGGL(k_setup_single_line, dim3((t + 255) / 256), dim3(256), 0, s, P, n);
    hipLaunchKernelGGL(k_seg_records, dim3((ctx->seg_rstride + 255) / 256, 1), dim3(256), 0, s, P, 1);
    MML_HIP(hipMemsetAsync(ctx->brk_cnt, 0, sizeof(int), s));
    MML_HIP(hipMemsetAsync(ctx->brk_cnt + ctx->B, 0, sizeof(int), s));
    if (n > 0) {
        // line 0 holds the whole input: one wavefront per tile (up to 256 of them, longer lines loop)
        const int tiles = (n + ST_TILE - 1) / ST_TILE;
        const dim3 grid(tiles < 256 ? tiles : 256, 1, 1);
        hipLaunchKernelGGL(k_stencil<1>, grid, dim3(64 * ST_LINES), 0, s, P);
        hipLaunchKernelGGL(k_stencil<2>, grid, dim3(64 * ST_LINES), 0, s, P);
        hipLaunchKernelGGL(k_stencil_redo, dim3(4, 1), dim3(256), 0, s, P);
        hipLaunchKernelGGL(k_stencil_break, dim3(2, 1), dim3(256), 0, s, P);
    }
    if (ctx->select_part) {
        hipLaunchKernelGGL((k_select_part<u64m, SP_LINES, SP_MAXWIN>), dim3(1, 1), dim3(64 * SP_LINES), 0, s, P, 1);
        hipLaunchKernelGGL((k_select_part<u128m, SP_LINES_WIDE, SP_MAXWIN_WIDE>), dim3(1, 1), dim3(64 * SP_LINES_WIDE), 0, s, P, 1);
    }
    // (direct mode: the workgroup finds its line done by k_select_part, or does it)
    hipLaunchKernelGGL(select_variant(ctx->sel_cap), dim3(1, 1), dim3(SELP_THREADS), select_lds_bytes(ctx->sel_cap), s, P, -1);
    MML_HIP(hipGetLastError());
    return MML_OK;
}

int mml_feature_init(mml_ctx* ctx) {
    {
        const char* e = getenv("MML_SELECT_PART");  // measurement switch: 0 = every line through k_select
        ctx->select_part = !(e && atoi(e) == 0);
    }
    {
        // one-pass bucketing: ring layouts whose block tables fit (lines as lanes of one wavefront, <= 16 blocks per sensor);
        // dense layouts (128 rings, 262 k points) keep the three staged passes.  MML_ASSIGN_ONEPASS=0: measurement switch
        const char* e = getenv("MML_ASSIGN_ONEPASS");
        const int nbv = (ctx->NV + MML_OP_BLK - 1) / MML_OP_BLK, nbl = (ctx->NL + MML_OP_BLK - 1) / MML_OP_BLK;
        ctx->onepass = !(e && atoi(e) == 0) && ctx->cfg.n_rings >= 1 && ctx->cfg.n_rings <= OP_MAXKEYS && ctx->cfg.n_livox_lines <= OP_MAXKEYS &&
                       ctx->L <= 64 && nbv <= MML_SEG_MAX && nbl <= MML_SEG_MAX && nbv * ctx->cfg.n_rings <= MML_SEG_FLAT &&
                       nbl * ctx->cfg.n_livox_lines <= MML_SEG_FLAT;
    }
    // LDS budget of k_select: twice the nominal ring length / the nominal Livox line length + 2 %, rounded up to a
    // whole number of points per thread (the default 16 x 1800 + 6 x 4000 layout gets 4096 points = 51.6 KB, three
    // workgroups per CU); longer lines take the global-scratch form of the same code
    const int ring_nominal = ctx->NV / (ctx->cfg.n_rings > 0 ? ctx->cfg.n_rings : 1);
    int capl = (51 * (ctx->NL / (ctx->cfg.n_livox_lines > 0 ? ctx->cfg.n_livox_lines : 1))) / 50;
    int cap = capl > 2 * ring_nominal ? capl : 2 * ring_nominal;  // Livox lines, and the single-line entry point
    cap = select_round_cap(cap);
    ctx->sel_cap = cap;
    // rings: nominal length + 12 % (a 16 x 1800 scan gets 2048 points = 28.9 KB, five workgroups per CU)
    ctx->sel_cap_velo = select_round_cap((ring_nominal * 9) / 8);
    if (ctx->sel_cap_velo > cap) ctx->sel_cap_velo = cap;
    MML_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(select_variant(cap)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_lds_bytes(cap)));
    MML_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(select_variant(ctx->sel_cap_velo)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)select_lds_bytes(ctx->sel_cap_velo)));
    return MML_OK;
}
