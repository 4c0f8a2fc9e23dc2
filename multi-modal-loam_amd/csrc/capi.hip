// capi.hip -- the C-ABI of libmmloam_hip.so (include/mmloam_hip.h): context, buffers, call sequencing.
// No compute lives here; every entry point validates, stages small parameter blocks through a pinned ring and
// enqueues the kernels of feature.hip / undistort_voxel.hip / map_assoc.hip / solve.hip on the ctx stream.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "mml_internal.h"

int mml_launch_detect_line(mml_ctx* ctx, int n, uint16_t* d_final);

namespace {

template <typename T>
hipError_t dalloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (n ? n : 1));
}

constexpr int kPackMaxSlots = 16;  // calls of up to this many slots move their parameter blocks / results in one copy each way
// pinned staging ring for small host<->device parameter blocks
constexpr size_t kStageDoubles = 1u << 22;  // 32 MiB: a 1024-scan step stages ~45 k doubles; a wrap synchronises the device

double* stage_alloc(mml_ctx* ctx, size_t doubles) {
    if (ctx->stage_cursor + doubles > ctx->h_stage_doubles) {
        mml_sync_all(ctx);
        ctx->stage_cursor = 0;
    }
    double* p = ctx->h_stage + ctx->stage_cursor;
    ctx->stage_cursor += (doubles + 7) & ~size_t(7);
    return p;
}

}  // namespace

double* mml_stage_alloc(mml_ctx* ctx, size_t doubles) { return stage_alloc(ctx, doubles); }

extern "C" {

void mml_config_default(mml_config* cfg, int max_scans) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->max_scans = max_scans;
    cfg->max_velo_points = 16 * 1800;
    cfg->max_livox_points = 24000;
    cfg->n_rings = 16;
    cfg->pitch0_deg = -15.0f;
    cfg->pitch_step_deg = 2.0f;
    cfg->n_livox_lines = 6;
    cfg->near_th = 2.0f;
    cfg->far_th = 50.0f;
    cfg->leaf_corner = 0.4f;
    cfg->leaf_surf = 0.2f;
    cfg->cell_corner = 0.f;
    cfg->cell_surf = 0.f;
    cfg->max_features = 0;
    cfg->max_map_points = 0;
}

int mml_abi_version(void) { return MML_ABI_VERSION; }

const char* mml_last_error(const mml_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int mml_config_get(const mml_ctx* ctx, mml_config* out) {
    if (!ctx || !out) return MML_ERR_INVALID;
    *out = ctx->cfg;
    return MML_OK;
}

void mml_destroy(mml_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l)
        if (ctx->streams[l]) hipStreamSynchronize(ctx->streams[l]);
    if (ctx->copy_stream) hipStreamSynchronize(ctx->copy_stream);
    mml_comm_destroy(ctx);
    mml_fullwindow_dev_release(ctx);
    void* ptrs[] = {ctx->wstate, ctx->wrec, ctx->waux, ctx->hard_knn, ctx->d_und, ctx->crop_cnt, ctx->velo_in,  ctx->livox_in, ctx->d_n_in,   ctx->raw_line, ctx->raw_ori,  ctx->ln_pts,
                    ctx->ln_gidx, ctx->ln_rel, ctx->line_start, ctx->line_len, ctx->seg_cum, ctx->seg_pos, ctx->seg_n, ctx->seg_flat, ctx->seg_flat_n, ctx->op_agg, ctx->seg_rs, ctx->seg_rw, ctx->ln_curv, ctx->ln_refl,  ctx->ln_attr,
                    ctx->sel_scratch, ctx->blk_cnt, ctx->assign_aux, ctx->brk_queue, ctx->brk_cnt, ctx->redo_queue, ctx->st_exit, ctx->vx_big, ctx->sel_done, ctx->sel_list, ctx->sel_list_cnt,
                    ctx->cb_n,     ctx->queue_off, ctx->slot_flags, ctx->ln_line,  ctx->ln_label,
                    ctx->fu_info,  ctx->ft_xyz[0], ctx->ft_xyz[1], ctx->ft_n,   ctx->vx_keys,  ctx->vx_gidx, ctx->lf,
                    ctx->pf,       ctx->assoc_stats, ctx->hard_list, ctx->work_off, ctx->grid[0].pts, ctx->grid[1].pts, ctx->grid[0].cell_start,
                    ctx->grid[1].cell_start, ctx->map_tmp, ctx->map_keys, ctx->map_keys2, ctx->map_vals,
                    ctx->map_vals2, ctx->sort_tmp, ctx->d_x, ctx->d_pose_in, ctx->d_result, ctx->d_summ, ctx->d_trace, ctx->d_rec,
                    ctx->d_extr,   ctx->d_misc,   ctx->ggrid[0].pts, ctx->ggrid[1].pts, ctx->ggrid[0].cell_start,
                    ctx->ggrid[1].cell_start, ctx->ggrid[0].tags, ctx->ggrid[1].tags, ctx->gmap_orig[0],
                    ctx->gmap_orig[1], ctx->gtag_orig[0], ctx->gtag_orig[1], ctx->cube_cnt[0], ctx->cube_cnt[1],
                    ctx->ring[0],  ctx->ring[1],  ctx->ring_cat, ctx->vox_flag, ctx->wire_stage, ctx->gs_pts[0], ctx->gs_pts[1], ctx->gs_tag[0],
                    ctx->gs_tag[1], ctx->gs_pts2[0], ctx->gs_pts2[1], ctx->gs_tag2[0], ctx->gs_tag2[1], ctx->gp_pts[0],
                    ctx->gp_pts[1], ctx->gs_work, ctx->gs_keys, ctx->seg_keys, ctx->seg_vals, ctx->seg_cat, ctx->seg_flag, ctx->seg_meta,
                    ctx->seg_tmp[0], ctx->seg_tmp[1], ctx->seg_tmp[2], ctx->seg_tmp[3], ctx->seg_tmp[4], ctx->seg_tmp[5],
                    ctx->seg_tmp[6], ctx->seg_tmp[7]};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    for (auto& pe : ctx->pending) {
        hipEventDestroy(pe.a);
        hipEventDestroy(pe.b);
    }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    for (auto& g : ctx->win_graphs) hipGraphExecDestroy(g.exec);
    for (auto& u : ctx->uploads) hipEventDestroy(u.done);
    for (auto e : ctx->upload_event_pool) hipEventDestroy(e);
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l) {
        if (ctx->lane_mark[l]) hipEventDestroy(ctx->lane_mark[l]);
        if (ctx->streams[l]) hipStreamDestroy(ctx->streams[l]);
    }
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

int mml_create(const mml_config* cfg, int device, mml_ctx** out) {
    if (!cfg || !out) return MML_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MML_ERR_NO_DEVICE;  // no CPU fallback
    if (device < 0 || device >= ndev) return MML_ERR_NO_DEVICE;
    if (cfg->max_scans <= 0 || cfg->max_velo_points < 0 || cfg->max_livox_points < 0 || cfg->n_rings <= 0 ||
        cfg->n_rings > 128 || cfg->n_livox_lines <= 0 || cfg->n_livox_lines > 32)
        return MML_ERR_INVALID;
    mml_ctx* ctx = new mml_ctx();
    ctx->cfg = *cfg;
    ctx->device = device;
    if (ctx->cfg.cell_corner <= 0) ctx->cfg.cell_corner = 5.0f * ctx->cfg.leaf_corner;
    if (ctx->cfg.cell_surf <= 0) ctx->cfg.cell_surf = 5.0f * ctx->cfg.leaf_surf;
    if (ctx->cfg.max_features <= 0) ctx->cfg.max_features = 8192;
    if (ctx->cfg.max_map_points <= 0) ctx->cfg.max_map_points = 1 << 21;
    ctx->B = cfg->max_scans;
    ctx->NV = (cfg->max_velo_points + 63) & ~63;
    ctx->NL = (cfg->max_livox_points + 63) & ~63;
    ctx->NT = ctx->NV + ctx->NL;
    ctx->L = cfg->n_rings + cfg->n_livox_lines;
    ctx->MF = ctx->cfg.max_features;
    ctx->MM = ctx->cfg.max_map_points;
    // label lists per (slot, kind) hold every labelled point (8 B x NT per slot): up to MML_VOXEL_LDS_CAP of them feed
    // the LDS sort of k_voxel, denser labelled clouds and scans beyond 64k points (e.g. 128 x 2048 rings) take the
    // global-sort filter (mml_downsample_big)
    ctx->VX_CAP = (int)ctx->NT;
    ctx->h_n_in.assign((size_t)ctx->B * 2, 0);
    ctx->raw_extracted.assign((size_t)ctx->B, 0);
    ctx->stats_stale.assign((size_t)ctx->B, 1);
    auto fail = [&](hipError_t e, const char* what) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        // keep ctx alive so the caller can read the message? No: report through the return code only.
        mml_destroy(ctx);
        return MML_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    // (two lanes: with calls of ~2000 scans two to four lanes give the same rate, with 8192 two give 360 k scans/s against 351 k for
    //  four, and a call of a few hundred scans splits into launches that still fill the device)
    ctx->n_lanes = 2;
    if (const char* e_l = getenv("MML_LANES")) {  // tuning knob: number of stream lanes mml_step pipelines over
        int v = atoi(e_l);
        if (v >= 1 && v <= mml_ctx::MAX_LANES) ctx->n_lanes = v;
    }
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l) {
        if ((e = hipStreamCreateWithFlags(&ctx->streams[l], hipStreamNonBlocking)) != hipSuccess)
            return fail(e, "hipStreamCreate");
        if ((e = hipEventCreateWithFlags(&ctx->lane_mark[l], hipEventDisableTiming)) != hipSuccess)
            return fail(e, "hipEventCreate");
    }
    if ((e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    const size_t B = ctx->B, NV = ctx->NV, NL = ctx->NL, NT = ctx->NT, L = ctx->L, MF = ctx->MF, MM = ctx->MM;
#define ALLOC(ptr, n)                                                  \
    if ((e = dalloc(&(ptr), (n))) != hipSuccess) return fail(e, #ptr); \
    if ((e = hipMemsetAsync((ptr), 0, sizeof(*(ptr)) * ((n) ? (n) : 1), MML_STREAM(ctx))) != hipSuccess) return fail(e, #ptr)
    ALLOC(ctx->velo_in, B * NV);
    ALLOC(ctx->livox_in, B * NL);
    ALLOC(ctx->d_n_in, B * 2);
    ALLOC(ctx->raw_line, B * NT);
    ALLOC(ctx->raw_ori, B * NV);
    ALLOC(ctx->ln_pts, B * NT);
    ALLOC(ctx->ln_gidx, B * NT);
    ALLOC(ctx->ln_rel, B * NT);
    ALLOC(ctx->slot_flags, B * 2);
    ALLOC(ctx->line_start, B * L);
    ALLOC(ctx->line_len, B * L);
    ALLOC(ctx->seg_cum, B * L * (MML_SEG_MAX + 1));
    ALLOC(ctx->seg_pos, B * L * MML_SEG_MAX);
    ALLOC(ctx->seg_n, B * L);
    ALLOC(ctx->seg_flat, B * 2 * MML_SEG_FLAT);
    ALLOC(ctx->seg_flat_n, B * 4);
    ALLOC(ctx->op_agg, B * 2 * MML_SEG_MAX);
    ctx->seg_rstride = (int)((NT >> 6) + 6 * L + 8);
    ALLOC(ctx->seg_rs, B * (size_t)ctx->seg_rstride);
    ALLOC(ctx->seg_rw, B * (size_t)ctx->seg_rstride);
    ALLOC(ctx->ln_curv, B * NT);
    ALLOC(ctx->ln_refl, B * NT);
    ALLOC(ctx->ln_attr, B * NT);
    ALLOC(ctx->sel_scratch, 4 * B * NT + 16 * B * (L + 2) + 64);
    {
        const size_t nblk = ((NV > NL ? NV : NL) + 255) / 256;
        ALLOC(ctx->blk_cnt, B * 2 * nblk * 162);
    }
    ALLOC(ctx->assign_aux, B * 8);
    ALLOC(ctx->brk_queue, B * NT);
    ALLOC(ctx->brk_cnt, 2 * B);
    ALLOC(ctx->queue_off, 2 * ((size_t)B + mml_ctx::MAX_LANES + 1));
    ALLOC(ctx->redo_queue, B * NT);
    ALLOC(ctx->st_exit, B * (NT / 256 + L + 8));
    ALLOC(ctx->vx_big, 2 * B);
    ALLOC(ctx->sel_done, B * L + 8);
    ALLOC(ctx->sel_list, 4 * B * L + 8);
    ALLOC(ctx->sel_list_cnt, 2 * B + 8);
    ALLOC(ctx->crop_cnt, B * ((NT + 255) / 256) * 8);
    ALLOC(ctx->cb_n, B * 2);
    ALLOC(ctx->ln_line, B * NT);
    ALLOC(ctx->ln_label, B * NT);
    ALLOC(ctx->fu_info, B * 8);
    ALLOC(ctx->ft_xyz[0], B * MF);
    ALLOC(ctx->ft_xyz[1], B * MF);
    ALLOC(ctx->ft_n, B * 2);
    ALLOC(ctx->vx_keys, B * (size_t)ctx->VX_CAP);  // B*2*VX_CAP unsigned
    ALLOC(ctx->vx_gidx, 2 * B * (size_t)ctx->VX_CAP);
    ALLOC(ctx->lf, B * MF);
    ALLOC(ctx->pf, B * MF);
    ALLOC(ctx->assoc_stats, B * 16);
    ALLOC(ctx->hard_list, B * MF * 2);
    ALLOC(ctx->hard_knn, B * MF * 2 * 10);
    ALLOC(ctx->work_off, 2 * B + 8);
    for (int k = 0; k < 2; ++k) {
        ALLOC(ctx->grid[k].pts, MM);
        ALLOC(ctx->grid[k].cell_start, 4 * MM + 4096 + 2);
        ctx->grid[k].m = 0;
    }
    ALLOC(ctx->map_tmp, 2 * MM);
    ALLOC(ctx->map_keys, MM);
    ALLOC(ctx->map_keys2, MM);
    ALLOC(ctx->map_vals, MM);
    ALLOC(ctx->map_vals2, MM);
    ALLOC(ctx->d_x, B * 6);
    ALLOC(ctx->d_pose_in, B * 64);
    ALLOC(ctx->d_result, B * MML_SOLVE_RESULT);
    ALLOC(ctx->d_summ, B * 8);
    ALLOC(ctx->d_trace, B * 6 * 64);
    ALLOC(ctx->d_rec, B * 32);
    ALLOC(ctx->d_und, B * 8);
    ALLOC(ctx->d_extr, 16);
    ALLOC(ctx->d_misc, 64);
#undef ALLOC
    ctx->h_stage_doubles = kStageDoubles;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), sizeof(double) * (ctx->h_stage_doubles + 8),
                           hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc");
    ctx->stage_cursor = 0;
    if (mml_feature_init(ctx) != MML_OK) return fail(hipErrorUnknown, "mml_feature_init");
    if ((e = hipStreamSynchronize(MML_STREAM(ctx))) != hipSuccess) return fail(e, "hipStreamSynchronize");
    *out = ctx;
    return MML_OK;
}

int mml_synchronize(mml_ctx* ctx) {
    if (!ctx) return MML_ERR_INVALID;
    return mml_sync_all(ctx);
}

static int check_slots(mml_ctx* ctx, int first, int count) {
    MML_REQUIRE(ctx != nullptr, MML_ERR_INVALID, "null ctx");
    MML_REQUIRE(first >= 0 && count > 0 && first + count <= ctx->B, MML_ERR_INVALID, "slot range out of bounds");
    return MML_OK;
}
#define CHECK_SLOTS(first, count)                        \
    do {                                                 \
        if (!ctx) return MML_ERR_INVALID;                \
        int rc_ = check_slots(ctx, (first), (count));    \
        if (rc_ != MML_OK) return rc_;                   \
        if (hipSetDevice(ctx->device) != hipSuccess) {   \
            ctx->err = "hipSetDevice failed";            \
            return MML_ERR_HIP;                          \
        }                                                \
        if (!ctx->uploads.empty()) {                     \
            rc_ = mml_uploads_wait(ctx, (first), (count)); \
            if (rc_ != MML_OK) return rc_;               \
        }                                                \
    } while (0)

// copy a block of doubles to the device through the pinned ring (asynchronous, safe against reuse)
static int upload_doubles(mml_ctx* ctx, double* d_dst, const double* h_src, size_t n) {
    double* st = stage_alloc(ctx, n);
    memcpy(st, h_src, sizeof(double) * n);
    MML_HIP(hipMemcpyAsync(d_dst, st, sizeof(double) * n, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    return MML_OK;
}

int mml_scan_upload(mml_ctx* ctx, int slot, const float* velo_xyzi, int n_velo, const mml_livox_point* livox,
                    int n_livox) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(n_velo >= 0 && n_livox >= 0, MML_ERR_INVALID, "negative point count");
    MML_REQUIRE(n_velo <= ctx->cfg.max_velo_points && n_livox <= ctx->cfg.max_livox_points, MML_ERR_CAPACITY,
                "scan exceeds max_velo_points / max_livox_points");
    MML_REQUIRE((n_velo == 0 || velo_xyzi) && (n_livox == 0 || livox), MML_ERR_INVALID, "null point buffer");
    if (n_velo)
        MML_HIP(hipMemcpyAsync(ctx->velo_in + (size_t)slot * ctx->NV, velo_xyzi, sizeof(float) * 4 * (size_t)n_velo,
                               hipMemcpyHostToDevice, MML_STREAM(ctx)));
    if (n_livox)
        MML_HIP(hipMemcpyAsync(ctx->livox_in + (size_t)slot * ctx->NL, livox, sizeof(mml_livox_point) * (size_t)n_livox,
                               hipMemcpyHostToDevice, MML_STREAM(ctx)));
    ctx->h_n_in[2 * slot] = n_velo;
    ctx->raw_extracted[slot] = 0;
    ctx->h_n_in[2 * slot + 1] = n_livox;
    // counts travel through the pinned ring as raw bytes
    double* st = stage_alloc(ctx, 1);
    int* sti = reinterpret_cast<int*>(st);
    sti[0] = n_velo;
    sti[1] = n_livox;
    MML_HIP(hipMemcpyAsync(ctx->d_n_in + 2 * slot, sti, sizeof(int) * 2, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    return MML_OK;
}

int mml_scan_upload_batch(mml_ctx* ctx, int first_slot, int count, const float* velo_base, const int* n_velo,
                          const mml_livox_point* livox_base, const int* n_livox) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(n_velo && n_livox, MML_ERR_INVALID, "null count arrays");
    MML_REQUIRE(ctx->NV == ctx->cfg.max_velo_points && ctx->NL == ctx->cfg.max_livox_points, MML_ERR_INVALID,
                "mml_scan_upload_batch needs max_velo_points / max_livox_points that are multiples of 64 (the slot stride)");
    bool any_v = false, any_l = false;
    for (int i = 0; i < count; ++i) {
        MML_REQUIRE(n_velo[i] >= 0 && n_livox[i] >= 0, MML_ERR_INVALID, "negative point count");
        MML_REQUIRE(n_velo[i] <= ctx->cfg.max_velo_points && n_livox[i] <= ctx->cfg.max_livox_points, MML_ERR_CAPACITY,
                    "scan exceeds max_velo_points / max_livox_points");
        any_v = any_v || n_velo[i] > 0;
        any_l = any_l || n_livox[i] > 0;
    }
    MML_REQUIRE((!any_v || velo_base) && (!any_l || livox_base), MML_ERR_INVALID, "null point buffer");
    // the copies go behind everything already enqueued on the lanes (a kernel may still be reading these slots) ...
    hipStream_t cs = ctx->copy_stream;
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l) {
        MML_HIP(hipEventRecord(ctx->lane_mark[l], ctx->streams[l]));
        MML_HIP(hipStreamWaitEvent(cs, ctx->lane_mark[l], 0));
    }
    if (any_v)
        MML_HIP(hipMemcpyAsync(ctx->velo_in + (size_t)first_slot * ctx->NV, velo_base, sizeof(float4) * (size_t)count * ctx->NV,
                               hipMemcpyHostToDevice, cs));
    if (any_l)
        MML_HIP(hipMemcpyAsync(ctx->livox_in + (size_t)first_slot * ctx->NL, livox_base, sizeof(mml_livox_point) * (size_t)count * ctx->NL,
                               hipMemcpyHostToDevice, cs));
    double* st = stage_alloc(ctx, (size_t)count);
    int* sti = reinterpret_cast<int*>(st);
    for (int i = 0; i < count; ++i) {
        ctx->h_n_in[2 * (first_slot + i)] = sti[2 * i] = n_velo[i];
        ctx->raw_extracted[first_slot + i] = 0;
        ctx->h_n_in[2 * (first_slot + i) + 1] = sti[2 * i + 1] = n_livox[i];
    }
    MML_HIP(hipMemcpyAsync(ctx->d_n_in + 2 * (size_t)first_slot, sti, sizeof(int) * 2 * (size_t)count, hipMemcpyHostToDevice, cs));
    // ... and whatever touches these slots next waits for them (CHECK_SLOTS -> mml_uploads_wait); work on other slots
    // enqueued from now on runs concurrently with the copy
    mml_ctx::Upload u;
    u.first = first_slot;
    u.count = count;
    if (!ctx->upload_event_pool.empty()) {
        u.done = ctx->upload_event_pool.back();
        ctx->upload_event_pool.pop_back();
    } else {
        MML_HIP(hipEventCreateWithFlags(&u.done, hipEventDisableTiming));
    }
    MML_HIP(hipEventRecord(u.done, cs));
    ctx->uploads.push_back(u);
    return MML_OK;
}

// ---- wire formats ---------------------------------------------------------------------
namespace {
__device__ __forceinline__ float load_f32_unaligned(const uint8_t* p) {
    const unsigned u = (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
    return __uint_as_float(u);
}
// pcl::fromROSMsg<PointXYZI> on the device: fields located by byte offset inside point_step-sized records
__global__ void k_decode_pointcloud2(const uint8_t* raw, int n, int step, int ox, int oy, int oz, int oi, float4* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = raw + (size_t)i * step;
    float4 v;
    v.x = load_f32_unaligned(p + ox);
    v.y = load_f32_unaligned(p + oy);
    v.z = load_f32_unaligned(p + oz);
    v.w = oi >= 0 ? load_f32_unaligned(p + oi) : 0.f;
    out[i] = v;
}
// serialised livox_ros_driver/CustomPoint records (19 bytes, unaligned) -> the 20-byte structs of livox_in
__global__ void k_decode_custompoints(const uint8_t* raw, int n, mml_livox_point* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* p = raw + (size_t)i * 19;
    mml_livox_point q;
    q.offset_time = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    q.x = load_f32_unaligned(p + 4);
    q.y = load_f32_unaligned(p + 8);
    q.z = load_f32_unaligned(p + 12);
    q.reflectivity = p[16];
    q.tag = p[17];
    q.line = p[18];
    q._pad = 0;
    out[i] = q;
}
// The fused cloud [velo_combine ; livox_combine] in its own order, gathered from the line-bucketed storage: position pos
// (Velodyne region [0, cv), Livox region [NV, NV + cl)) holds fused point ln_gidx[pos] when that is >= 0.  The Velodyne
// part of an extracted cloud carries intensity 0 (unionFeatureExtract.cpp:1254-1256); an uploaded cloud keeps its own.
struct FusedView {
    const float4* pts;
    const int* gidx;
    const int* rel;
    const uint8_t* line;      // uploaded clouds
    const uint8_t* label;
    const int* cb_n;          // this slot's two valid counts
    const int* seg_flat;      // extracted clouds: per sensor the starts of the storage segments in storage order (block-major,
    const int* seg_flat_n;    //   line inside a block) -- 2 x MML_SEG_FLAT ints -- and (entries, lines per block) per sensor
    int NV, NT, L, n_rings, flags;
};
__device__ __forceinline__ bool fused_at(const FusedView& V, int pos, int& g, float4& p, float& rel, int& line) {
    if (pos >= V.NT) return false;
    if (pos < V.NV ? pos >= V.cb_n[0] : pos - V.NV >= V.cb_n[1]) return false;
    g = V.gidx[pos];
    if (g < 0) return false;
    p = V.pts[pos];
    rel = (V.flags & 2) ? 1.0f : __int_as_float(V.rel[pos]);  // RemoveLidarDistortion leaves normal_x = 1 (unionPoseEstimation.cpp:419)
    if (V.flags & 1) {
        line = V.line[pos];
    } else {
        if (pos < V.NV) p.w = 0.f;  // intensity of the Velodyne part is zeroed (unionFeatureExtract.cpp:1254-1256)
        // last storage segment of the region whose start is <= pos (starts are non-decreasing; an empty segment shares its start
        // with the next one); segment f belongs to line f mod (lines per block)
        const int sensor = pos < V.NV ? 0 : 1;
        const int* flat = V.seg_flat + sensor * MML_SEG_FLAT;
        int lo = 0, hi = V.seg_flat_n[2 * sensor] - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (flat[mid] <= pos)
                lo = mid;
            else
                hi = mid - 1;
        }
        line = lo % V.seg_flat_n[2 * sensor + 1];
    }
    return true;
}
// pcl::toROSMsg<PointXYZINormal> payload: 48-byte records, x y z 1 | normal_x normal_y normal_z 0 | intensity curvature 0 0
__global__ void k_encode_xyzinormal(FusedView V, float* out) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    int g, line;
    float4 p;
    float rel;
    if (!fused_at(V, pos, g, p, rel, line)) return;
    float* o = out + 12 * (size_t)g;
    o[0] = p.x;
    o[1] = p.y;
    o[2] = p.z;
    o[3] = 1.0f;
    o[4] = rel;                    // normal_x: in-sweep time (unionFeatureExtract.cpp:1186)
    o[5] = (float)line;            // normal_y: ring / Livox line
    o[6] = (float)V.label[pos];    // normal_z: 0 none, 1 corner, 2 surf (:1018-1021)
    o[7] = 0.f;
    o[8] = p.w;                    // intensity
    o[9] = 0.f;                    // curvature
    o[10] = 0.f;
    o[11] = 0.f;
}
// the same cloud as four arrays (mml_scan_download): xyzi | reltime | line | label, back to back in one staging buffer
__global__ void k_fused_arrays(FusedView V, int n, float4* xyzi, float* rel, uint8_t* line, uint8_t* label) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    int g, ln;
    float4 p;
    float r;
    if (!fused_at(V, pos, g, p, r, ln)) return;
    xyzi[g] = p;
    rel[g] = r;
    line[g] = (uint8_t)ln;
    label[g] = V.label[pos];
}
// ---- mml_slot_digest: order-independent 64-bit digests of what the download entry points would hand out -----------------------
__host__ __device__ __forceinline__ unsigned long long dg_mix(unsigned long long z) {  // splitmix64 finaliser
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
__host__ __device__ __forceinline__ unsigned long long dg_key(unsigned long long i, unsigned tag) {
    return dg_mix(i * 0x9E3779B97F4A7C15ULL + tag);
}
__device__ __forceinline__ unsigned long long dg_pair(float a, float b) {
    return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
__device__ __forceinline__ unsigned long long dg_dbits(double d) { return (unsigned long long)__double_as_longlong(d); }
// sum over the workgroup, one atomic per wavefront
__device__ __forceinline__ void dg_commit(unsigned long long v, unsigned long long* dst) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = (unsigned long long)__shfl_down(lo, o) | ((unsigned long long)__shfl_down(hi, o) << 32);
        v += t;
        lo = (unsigned)v;
        hi = (unsigned)(v >> 32);
    }
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}
struct DigestArgs {
    const float4* pts;
    const int* gidx;
    const int* rel;
    const uint8_t* line;
    const uint8_t* label;
    const int* cb_n;
    const int* seg_flat;
    const int* seg_flat_n;
    const int* slot_flags;
    const int* fu_info;
    const int* ft_n;
    const float4* ft[2];
    const MmlLineFactor* lf;
    const MmlPlaneFactor* pf;
    const double* x;
    int B, NV, NT, L, MF, n_rings, first;
};
// pieces 1-4: the fused cloud, one storage position per thread (blockIdx.y = slot of the call)
__global__ void __launch_bounds__(256) k_digest_cloud(DigestArgs A, unsigned long long* out) {
    const int slot = A.first + blockIdx.y;
    FusedView V;
    const size_t off = (size_t)slot * A.NT;
    V.pts = A.pts + off;
    V.gidx = A.gidx + off;
    V.rel = A.rel + off;
    V.line = A.line + off;
    V.label = A.label + off;
    V.cb_n = A.cb_n + 2 * (size_t)slot;
    V.seg_flat = A.seg_flat + (size_t)slot * 2 * MML_SEG_FLAT;
    V.seg_flat_n = A.seg_flat_n + (size_t)slot * 4;
    V.NV = A.NV;
    V.NT = A.NT;
    V.L = A.L;
    V.n_rings = A.n_rings;
    V.flags = A.slot_flags[2 * (size_t)slot];
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    int g = 0, ln = 0;
    float4 p;
    float r;
    unsigned long long h1 = 0, h2 = 0, h3 = 0, h4 = 0;
    if (fused_at(V, pos, g, p, r, ln)) {
        h1 = dg_mix(dg_key(g, 1) ^ (unsigned long long)V.label[pos]);
        h2 = dg_mix(dg_key(g, 2) ^ (unsigned long long)(ln & 255));
        h3 = dg_mix(dg_key(g, 3) ^ dg_pair(p.x, p.y)) + dg_mix(dg_key(g, 4) ^ dg_pair(p.z, p.w));
        h4 = dg_mix(dg_key(g, 5) ^ (unsigned long long)__float_as_uint(r));
    }
    unsigned long long* o = out + (size_t)blockIdx.y * MML_DIGEST_WORDS;
    dg_commit(h1, o + 1);
    dg_commit(h2, o + 2);
    dg_commit(h3, o + 3);
    dg_commit(h4, o + 4);
}
// pieces 0, 5-9: counts, stacks, factor records, pose (one stack entry per thread)
__global__ void __launch_bounds__(256) k_digest_stacks(DigestArgs A, unsigned long long* out) {
    const int slot = A.first + blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = A.ft_n[slot], n1 = A.ft_n[A.B + slot];
    unsigned long long h5 = 0, h6 = 0, h7 = 0, h8 = 0;
    if (i < n0) {
        const float4 q = A.ft[0][(size_t)slot * A.MF + i];
        h5 = dg_mix(dg_key(i, 6) ^ dg_pair(q.x, q.y)) + dg_mix(dg_key(i, 8) ^ (unsigned long long)__float_as_uint(q.z));
        const MmlLineFactor f = A.lf[(size_t)slot * A.MF + i];
        if (f.src >= 0) {
            unsigned long long h = dg_key((unsigned)f.src, 10);
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits((double)f.ori[c]));
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits((double)f.p1[c]));
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits((double)f.p2[c]));
            h7 = dg_mix(h ^ dg_dbits(f.error));
        }
    }
    if (i < n1) {
        const float4 q = A.ft[1][(size_t)slot * A.MF + i];
        h6 = dg_mix(dg_key(i, 7) ^ dg_pair(q.x, q.y)) + dg_mix(dg_key(i, 9) ^ (unsigned long long)__float_as_uint(q.z));
        const MmlPlaneFactor f = A.pf[(size_t)slot * A.MF + i];
        if (f.src >= 0) {
            unsigned long long h = dg_key((unsigned)f.src, 11);
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits((double)f.ori[c]));
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits(f.proj[c]));
            for (int c = 0; c < 3; ++c) h = dg_mix(h ^ dg_dbits((double)f.omega[c]));
            h8 = dg_mix(h ^ dg_dbits(f.error));
        }
    }
    unsigned long long* o = out + (size_t)blockIdx.y * MML_DIGEST_WORDS;
    dg_commit(h5, o + 5);
    dg_commit(h6, o + 6);
    dg_commit(h7, o + 7);
    dg_commit(h8, o + 8);
    if (i == 0) {
        unsigned long long h = dg_key(0, 13);
        for (int c = 0; c < 6; ++c) h = dg_mix(h ^ (unsigned long long)(unsigned)A.fu_info[8 * (size_t)slot + c]);
        h = dg_mix(h ^ (unsigned long long)(unsigned)n0);
        h = dg_mix(h ^ (unsigned long long)(unsigned)n1);
        o[0] = h;
        unsigned long long hp = dg_key(0, 12);
        for (int c = 0; c < 6; ++c) hp = dg_mix(hp ^ dg_dbits(A.x[6 * (size_t)slot + c]));
        o[9] = hp;
    }
}
int ensure_wire_stage(mml_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->wire_stage_bytes) return MML_OK;
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    if (ctx->wire_stage) MML_HIP(hipFree(ctx->wire_stage));
    MML_HIP(hipMalloc(&ctx->wire_stage, bytes));
    ctx->wire_stage_bytes = bytes;
    return MML_OK;
}
}  // namespace

namespace {
int upload_wire_impl(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x, int off_y,
                     int off_z, int off_intensity, const mml_livox_point* livox, const uint8_t* livox_wire, int n_livox);
}
int mml_scan_upload_pointcloud2(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x,
                                int off_y, int off_z, int off_intensity, const mml_livox_point* livox, int n_livox) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(n_livox <= 0 || livox, MML_ERR_INVALID, "null point buffer");
    return upload_wire_impl(ctx, slot, data, n_points, point_step, off_x, off_y, off_z, off_intensity, livox, nullptr, n_livox);
}
int mml_scan_upload_wire(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x, int off_y,
                         int off_z, int off_intensity, const uint8_t* livox_wire, int n_livox) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(n_livox <= 0 || livox_wire, MML_ERR_INVALID, "null point buffer");
    return upload_wire_impl(ctx, slot, data, n_points, point_step, off_x, off_y, off_z, off_intensity, nullptr, livox_wire, n_livox);
}
namespace {
int upload_wire_impl(mml_ctx* ctx, int slot, const uint8_t* data, int n_points, int point_step, int off_x, int off_y,
                     int off_z, int off_intensity, const mml_livox_point* livox, const uint8_t* livox_wire, int n_livox) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(n_points >= 0 && n_livox >= 0, MML_ERR_INVALID, "negative point count");
    MML_REQUIRE(n_points <= ctx->cfg.max_velo_points && n_livox <= ctx->cfg.max_livox_points, MML_ERR_CAPACITY,
                "scan exceeds max_velo_points / max_livox_points");
    MML_REQUIRE(n_points == 0 || data, MML_ERR_INVALID, "null point buffer");
    MML_REQUIRE(point_step >= 12 && point_step <= 256, MML_ERR_INVALID, "point_step out of range");
    const int offs[4] = {off_x, off_y, off_z, off_intensity};
    for (int k = 0; k < 4; ++k)
        MML_REQUIRE((k == 3 && offs[k] < 0) || (offs[k] >= 0 && offs[k] + 4 <= point_step), MML_ERR_INVALID,
                    "field offset outside the point record");
    const size_t bytes = (size_t)n_points * point_step;
    const size_t lbytes = livox_wire ? (size_t)n_livox * 19 : 0;
    const size_t loff = (bytes + 15) & ~size_t(15);  // the Livox records follow the Velodyne payload in the staging buffer
    if (n_points || lbytes) {
        int rc = ensure_wire_stage(ctx, loff + lbytes);
        if (rc != MML_OK) return rc;
    }
    if (n_points) {
        MML_HIP(hipMemcpyAsync(ctx->wire_stage, data, bytes, hipMemcpyHostToDevice, MML_STREAM(ctx)));
        hipLaunchKernelGGL(k_decode_pointcloud2, dim3((n_points + 255) / 256), dim3(256), 0, MML_STREAM(ctx),
                           reinterpret_cast<const uint8_t*>(ctx->wire_stage), n_points, point_step, off_x, off_y, off_z,
                           off_intensity, ctx->velo_in + (size_t)slot * ctx->NV);
        MML_HIP(hipGetLastError());
    }
    // the Livox part: decoded from its wire form, or as in mml_scan_upload; the two counts travel as there
    if (n_livox && livox_wire) {
        uint8_t* dst = reinterpret_cast<uint8_t*>(ctx->wire_stage) + loff;
        MML_HIP(hipMemcpyAsync(dst, livox_wire, lbytes, hipMemcpyHostToDevice, MML_STREAM(ctx)));
        hipLaunchKernelGGL(k_decode_custompoints, dim3((n_livox + 255) / 256), dim3(256), 0, MML_STREAM(ctx), dst, n_livox,
                           ctx->livox_in + (size_t)slot * ctx->NL);
        MML_HIP(hipGetLastError());
    } else if (n_livox) {
        MML_HIP(hipMemcpyAsync(ctx->livox_in + (size_t)slot * ctx->NL, livox, sizeof(mml_livox_point) * (size_t)n_livox,
                               hipMemcpyHostToDevice, MML_STREAM(ctx)));
    }
    ctx->h_n_in[2 * slot] = n_points;
    ctx->raw_extracted[slot] = 0;
    ctx->h_n_in[2 * slot + 1] = n_livox;
    double* st = stage_alloc(ctx, 1);
    int* sti = reinterpret_cast<int*>(st);
    sti[0] = n_points;
    sti[1] = n_livox;
    MML_HIP(hipMemcpyAsync(ctx->d_n_in + 2 * slot, sti, sizeof(int) * 2, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    return MML_OK;
}
}  // namespace

namespace {
int fused_view(mml_ctx* ctx, int slot, FusedView& V) {
    int fl = 0;
    MML_HIP(hipMemcpyAsync(&fl, ctx->slot_flags + 2 * (size_t)slot, sizeof(int), hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    const size_t off = (size_t)slot * ctx->NT;
    V.pts = ctx->ln_pts + off;
    V.gidx = ctx->ln_gidx + off;
    V.rel = ctx->ln_rel + off;
    V.line = ctx->ln_line + off;
    V.label = ctx->ln_label + off;
    V.cb_n = ctx->cb_n + 2 * (size_t)slot;
    V.seg_flat = ctx->seg_flat + (size_t)slot * 2 * MML_SEG_FLAT;
    V.seg_flat_n = ctx->seg_flat_n + (size_t)slot * 4;
    V.NV = ctx->NV;
    V.NT = ctx->NT;
    V.L = ctx->L;
    V.n_rings = ctx->cfg.n_rings;
    V.flags = fl;
    return MML_OK;
}
}  // namespace

int mml_scan_download_pointxyzinormal(mml_ctx* ctx, int slot, uint8_t* out, int capacity_points, int* n_points) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(n_points != nullptr, MML_ERR_INVALID, "null n_points");
    mml_scan_info info;
    int rc = mml_scan_info_get(ctx, slot, &info);
    if (rc != MML_OK) return rc;
    *n_points = info.n_points;
    if (!out || info.n_points == 0) return MML_OK;
    MML_REQUIRE(capacity_points >= info.n_points, MML_ERR_CAPACITY, "download capacity too small");
    const size_t bytes = (size_t)info.n_points * 48;
    rc = ensure_wire_stage(ctx, bytes);
    if (rc != MML_OK) return rc;
    FusedView V;
    rc = fused_view(ctx, slot, V);
    if (rc != MML_OK) return rc;
    hipLaunchKernelGGL(k_encode_xyzinormal, dim3((ctx->NT + 255) / 256), dim3(256), 0, MML_STREAM(ctx), V,
                       reinterpret_cast<float*>(ctx->wire_stage));
    MML_HIP(hipGetLastError());
    MML_HIP(hipMemcpyAsync(out, ctx->wire_stage, bytes, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    return MML_OK;
}

int mml_cloud_upload(mml_ctx* ctx, int slot, const uint8_t* pointxyzinormal, int n_points, int n_velo) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(n_points >= 0 && n_velo >= 0 && n_velo <= n_points, MML_ERR_INVALID, "bad point counts");
    MML_REQUIRE(n_points == 0 || pointxyzinormal, MML_ERR_INVALID, "null point buffer");
    MML_REQUIRE(n_velo <= ctx->NV && n_points - n_velo <= ctx->NL, MML_ERR_CAPACITY,
                "cloud exceeds max_velo_points / max_livox_points (the Velodyne and Livox parts have their own regions)");
    const size_t bytes = (size_t)n_points * 48;
    int rc = ensure_wire_stage(ctx, bytes ? bytes : 48);
    if (rc != MML_OK) return rc;
    if (bytes) MML_HIP(hipMemcpyAsync(ctx->wire_stage, pointxyzinormal, bytes, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    rc = mml_launch_cloud_decode(ctx, slot, reinterpret_cast<const float*>(ctx->wire_stage), n_points, n_velo);
    if (rc != MML_OK) return rc;
    // the staging buffer is reused by the next wire-format call and the host buffer belongs to the caller
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    return MML_OK;
}

int mml_extract(mml_ctx* ctx, int first_slot, int count, const float* livox_extrinsic) {
    CHECK_SLOTS(first_slot, count);
    bool have = false;
    if (livox_extrinsic) {
        static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        have = memcmp(I, livox_extrinsic, sizeof(I)) != 0;
        if (have) {
            double* st = stage_alloc(ctx, 8);
            memcpy(st, livox_extrinsic, sizeof(float) * 16);
            MML_HIP(hipMemcpyAsync(ctx->d_extr, st, sizeof(float) * 16, hipMemcpyHostToDevice, MML_STREAM(ctx)));
        }
    }
    return mml_launch_extract(ctx, first_slot, count, have);
}

int mml_scan_info_get(mml_ctx* ctx, int slot, mml_scan_info* info) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(info != nullptr, MML_ERR_INVALID, "null info");
    int* h = reinterpret_cast<int*>(stage_alloc(ctx, 4));  // pinned: a pageable destination makes the copy a blocking staged one
    MML_HIP(hipMemcpyAsync(h, ctx->fu_info + 8 * slot, sizeof(int) * 8, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    info->n_points = h[0];
    info->n_velo = h[1];
    info->velo_corner_num = h[2];
    info->velo_surf_num = h[3];
    info->livox_corner_num = h[4];
    info->livox_surf_num = h[5];
    info->fused_corner_num = h[6];
    info->fused_surf_num = h[7];
    return MML_OK;
}

int mml_scan_download(mml_ctx* ctx, int slot, float* xyzi, float* reltime, uint8_t* line, uint8_t* label,
                      int capacity) {
    CHECK_SLOTS(slot, 1);
    mml_scan_info info;
    int rc = mml_scan_info_get(ctx, slot, &info);
    if (rc != MML_OK) return rc;
    MML_REQUIRE(capacity >= info.n_points, MML_ERR_CAPACITY, "download capacity too small");
    const size_t n = info.n_points;
    if (n == 0) return MML_OK;
    // the fused order exists only here: gathered into a staging buffer (16 + 4 + 1 + 1 bytes per point), then copied out
    const size_t n16 = (n + 15) & ~size_t(15);
    rc = ensure_wire_stage(ctx, n16 * 22);
    if (rc != MML_OK) return rc;
    uint8_t* st = reinterpret_cast<uint8_t*>(ctx->wire_stage);
    float4* s_xyzi = reinterpret_cast<float4*>(st);
    float* s_rel = reinterpret_cast<float*>(st + n16 * 16);
    uint8_t* s_line = st + n16 * 20;
    uint8_t* s_label = st + n16 * 21;
    FusedView V;
    rc = fused_view(ctx, slot, V);
    if (rc != MML_OK) return rc;
    hipLaunchKernelGGL(k_fused_arrays, dim3((ctx->NT + 255) / 256), dim3(256), 0, MML_STREAM(ctx), V, (int)n, s_xyzi, s_rel, s_line, s_label);
    MML_HIP(hipGetLastError());
    if (xyzi) MML_HIP(hipMemcpyAsync(xyzi, s_xyzi, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    if (reltime) MML_HIP(hipMemcpyAsync(reltime, s_rel, sizeof(float) * n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    if (line) MML_HIP(hipMemcpyAsync(line, s_line, n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    if (label) MML_HIP(hipMemcpyAsync(label, s_label, n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    return MML_OK;
}

int mml_slot_digest(mml_ctx* ctx, int first_slot, int count, uint64_t* out) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(out != nullptr, MML_ERR_INVALID, "null out");
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    unsigned long long* d = nullptr;
    const size_t bytes = sizeof(unsigned long long) * MML_DIGEST_WORDS * (size_t)count;
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&d), bytes));
    hipError_t e = hipMemsetAsync(d, 0, bytes, MML_STREAM(ctx));
    DigestArgs A;
    A.pts = ctx->ln_pts;
    A.gidx = ctx->ln_gidx;
    A.rel = ctx->ln_rel;
    A.line = ctx->ln_line;
    A.label = ctx->ln_label;
    A.cb_n = ctx->cb_n;
    A.seg_flat = ctx->seg_flat;
    A.seg_flat_n = ctx->seg_flat_n;
    A.slot_flags = ctx->slot_flags;
    A.fu_info = ctx->fu_info;
    A.ft_n = ctx->ft_n;
    A.ft[0] = ctx->ft_xyz[0];
    A.ft[1] = ctx->ft_xyz[1];
    A.lf = ctx->lf;
    A.pf = ctx->pf;
    A.x = ctx->d_x;
    A.B = ctx->B;
    A.NV = ctx->NV;
    A.NT = ctx->NT;
    A.L = ctx->L;
    A.MF = ctx->MF;
    A.n_rings = ctx->cfg.n_rings;
    // (gridDim.y is limited to 65535: calls of more slots go in pieces)
    for (int done = 0; done < count && e == hipSuccess; done += 32768) {
        const int c = std::min(32768, count - done);
        A.first = first_slot + done;
        unsigned long long* o = d + (size_t)done * MML_DIGEST_WORDS;
        hipLaunchKernelGGL(k_digest_cloud, dim3((ctx->NT + 255) / 256, c), dim3(256), 0, MML_STREAM(ctx), A, o);
        hipLaunchKernelGGL(k_digest_stacks, dim3((ctx->MF + 255) / 256, c), dim3(256), 0, MML_STREAM(ctx), A, o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, MML_STREAM(ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(MML_STREAM(ctx));
    hipFree(d);
    if (e != hipSuccess) {
        ctx->err = std::string("mml_slot_digest: ") + hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    return MML_OK;
}

int mml_detect_line(mml_ctx* ctx, const float* pts, int n, int* sharp, int* n_sharp, int* flat, int* n_flat,
                    int* flags) {
    CHECK_SLOTS(0, 1);
    MML_REQUIRE(n >= 0 && n_sharp && n_flat && (n == 0 || (pts && sharp && flat)), MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(n <= ctx->NV, MML_ERR_CAPACITY, "line longer than max_velo_points");
    *n_sharp = 0;
    *n_flat = 0;
    if (n == 0) return MML_OK;
    for (int i = 0; i < n; ++i)
        MML_REQUIRE(std::isfinite(pts[4 * i]) && std::isfinite(pts[4 * i + 1]) && std::isfinite(pts[4 * i + 2]),
                    MML_ERR_INVALID, "detectFeaturePoints: non-finite input (reference indexes pre-compaction)");
    MML_HIP(hipMemcpyAsync(ctx->ln_pts, pts, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    // ln_final scratch: reuse ln_ord_c of slot 0's livox region?  Use a dedicated view of raw_ori (NV floats >= n*2 bytes).
    uint16_t* d_final = reinterpret_cast<uint16_t*>(ctx->raw_ori);
    int rc = mml_launch_detect_line(ctx, n, d_final);
    if (rc != MML_OK) return rc;
    std::vector<uint8_t> lab(n);
    std::vector<uint16_t> fin(n);
    MML_HIP(hipMemcpyAsync(lab.data(), ctx->ln_label, n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipMemcpyAsync(fin.data(), d_final, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    int ns = 0, nf = 0;
    for (int i = 0; i < n; ++i) {  // ascending i, as the emit loop at unionFeatureExtract.cpp:818-842
        if (lab[i] == 2) flat[nf++] = i;
        if (lab[i] == 1) sharp[ns++] = i;
        if (flags) flags[i] = fin[i];
    }
    *n_sharp = ns;
    *n_flat = nf;
    return MML_OK;
}

int mml_undistort(mml_ctx* ctx, int first_slot, int count, const double* dR, const double* dt) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(dR && dt, MML_ERR_INVALID, "null dR/dt");
    double* st = stage_alloc(ctx, 12 * (size_t)count);
    for (int i = 0; i < count; ++i) {
        memcpy(st + 12 * i, dR + 9 * i, sizeof(double) * 9);
        memcpy(st + 12 * i + 9, dt + 3 * i, sizeof(double) * 3);
    }
    double* d_par = ctx->d_pose_in + 64 * (size_t)first_slot;  // 64 doubles per slot: [0, 12*count) of this call's slice
    MML_HIP(hipMemcpyAsync(d_par, st, sizeof(double) * 12 * count, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    return mml_launch_undistort(ctx, first_slot, count, d_par);
}

int mml_downsample(mml_ctx* ctx, int first_slot, int count) {
    CHECK_SLOTS(first_slot, count);
    int rc = mml_launch_downsample(ctx, first_slot, count);
    if (rc != MML_OK) return rc;
    // pcl::VoxelGrid takes a cloud of any size: slots whose labelled cloud is beyond the LDS sort are redone
    return mml_downsample_redo_overflow(ctx, first_slot, count, nullptr);
}

int mml_features_download(mml_ctx* ctx, int slot, int kind, float* xyz, int capacity, int* n) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE((kind == 0 || kind == 1) && n, MML_ERR_INVALID, "bad kind");
    int h = 0;
    MML_HIP(hipMemcpyAsync(&h, ctx->ft_n + kind * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    MML_REQUIRE(h >= 0, MML_ERR_CAPACITY, "down-sample overflowed max_features / voxel capacity");
    *n = h;
    if (!xyz || h == 0) return MML_OK;
    MML_REQUIRE(capacity >= h, MML_ERR_CAPACITY, "download capacity too small");
    std::vector<float4> tmp(h);
    MML_HIP(hipMemcpyAsync(tmp.data(), ctx->ft_xyz[kind] + (size_t)slot * ctx->MF, sizeof(float4) * h,
                           hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    for (int i = 0; i < h; ++i) {
        xyz[3 * i] = tmp[i].x;
        xyz[3 * i + 1] = tmp[i].y;
        xyz[3 * i + 2] = tmp[i].z;
    }
    return MML_OK;
}

int mml_features_upload(mml_ctx* ctx, int slot, int kind, const float* xyz, int n) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE((kind == 0 || kind == 1) && n >= 0 && (n == 0 || xyz), MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(n <= ctx->MF, MML_ERR_CAPACITY, "more features than max_features");
    std::vector<float4> tmp(n ? n : 1);
    for (int i = 0; i < n; ++i) tmp[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f);
    if (n)
        MML_HIP(hipMemcpyAsync(ctx->ft_xyz[kind] + (size_t)slot * ctx->MF, tmp.data(), sizeof(float4) * n,
                               hipMemcpyHostToDevice, MML_STREAM(ctx)));
    MML_HIP(hipMemcpyAsync(ctx->ft_n + kind * ctx->B + slot, &n, sizeof(int), hipMemcpyHostToDevice, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    return MML_OK;
}

int mml_map_set_local(mml_ctx* ctx, int kind, const float* xyz, int m) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(m >= 0 && (m == 0 || xyz), MML_ERR_INVALID, "bad map arguments");
    MML_HIP(hipSetDevice(ctx->device));
    return mml_build_grid(ctx, kind, xyz, m);
}

int mml_map_increment_local(mml_ctx* ctx, int slot, const double* T_wl, int* n_corner_map, int* n_surf_map) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(T_wl, MML_ERR_INVALID, "null transform");
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    int n[2] = {0, 0};
    rc = mml_map_upkeep_increment(ctx, slot, T_wl, n);
    if (n_corner_map) *n_corner_map = n[0];
    if (n_surf_map) *n_surf_map = n[1];
    return rc;
}

int mml_map_local_reset(mml_ctx* ctx) {
    if (!ctx) return MML_ERR_INVALID;
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    memset(ctx->ring_n, 0, sizeof(ctx->ring_n));
    ctx->local_map_id = 0;
    return MML_OK;
}

int mml_map_local_download(mml_ctx* ctx, int kind, float* xyz, int capacity, int* n) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE((kind == 0 || kind == 1) && n, MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(ctx->have_map[kind], MML_ERR_STATE, "no local map");
    MML_HIP(hipSetDevice(ctx->device));
    const int m = ctx->grid[kind].m;
    *n = m;
    if (!xyz || m == 0) return MML_OK;
    MML_REQUIRE(capacity >= m, MML_ERR_CAPACITY, "capacity below the map size");
    std::vector<float4> tmp((size_t)m);
    MML_HIP(hipMemcpyAsync(tmp.data(), ctx->map_tmp + (size_t)kind * ctx->MM, sizeof(float4) * (size_t)m, hipMemcpyDeviceToHost,
                           MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    for (int i = 0; i < m; ++i) {
        xyz[3 * i] = tmp[i].x;
        xyz[3 * i + 1] = tmp[i].y;
        xyz[3 * i + 2] = tmp[i].z;
    }
    return MML_OK;
}

int mml_map_global_append(mml_ctx* ctx, int slot, const double* T_wl) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(T_wl, MML_ERR_INVALID, "null transform");
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    return mml_cube_store_append(ctx, slot, T_wl);
}

int mml_map_global_increment(mml_ctx* ctx, const double* T_wl, int* n_corner, int* n_surf) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(T_wl, MML_ERR_INVALID, "null transform");
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    int n[2] = {0, 0};
    rc = mml_cube_store_increment(ctx, T_wl, n);
    if (n_corner) *n_corner = n[0];
    if (n_surf) *n_surf = n[1];
    return rc;
}

int mml_map_global_download(mml_ctx* ctx, int kind, float* xyz, int* cube, int capacity, int* n, int* cen) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE((kind == 0 || kind == 1) && n, MML_ERR_INVALID, "bad arguments");
    MML_HIP(hipSetDevice(ctx->device));
    return mml_cube_store_download(ctx, kind, xyz, cube, capacity, n, cen);
}

int mml_map_global_reset(mml_ctx* ctx) {
    if (!ctx) return MML_ERR_INVALID;
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    return mml_cube_store_reset(ctx);
}

int mml_map_set_global(mml_ctx* ctx, int kind, const float* xyz, const int* cube, int m, const int* cen) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(m >= 0 && (m == 0 || (xyz && cube)), MML_ERR_INVALID, "bad global map arguments");
    MML_HIP(hipSetDevice(ctx->device));
    mml_sync_all(ctx);
    return mml_build_global_grid(ctx, kind, xyz, cube, m, cen);
}

int mml_knn5(mml_ctx* ctx, int kind, const float* q, int nq, float max_d2, int* idx, float* d2) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE((kind == 0 || kind == 1) && nq >= 0 && (nq == 0 || (q && idx && d2)), MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(ctx->have_map[kind], MML_ERR_STATE, "mml_knn5 before mml_map_set_local");
    if (nq == 0) return MML_OK;
    MML_HIP(hipSetDevice(ctx->device));
    float* d_q = nullptr;
    int* d_idx = nullptr;
    float* d_d2 = nullptr;
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&d_q), sizeof(float) * 3 * nq));
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&d_idx), sizeof(int) * 5 * nq));
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&d_d2), sizeof(float) * 5 * nq));
    MML_HIP(hipMemcpyAsync(d_q, q, sizeof(float) * 3 * nq, hipMemcpyHostToDevice, MML_STREAM(ctx)));
    int rc = mml_launch_knn5(ctx, kind, d_q, nq, max_d2, d_idx, d_d2);
    if (rc == MML_OK) {
        hipMemcpyAsync(idx, d_idx, sizeof(int) * 5 * nq, hipMemcpyDeviceToHost, MML_STREAM(ctx));
        hipMemcpyAsync(d2, d_d2, sizeof(float) * 5 * nq, hipMemcpyDeviceToHost, MML_STREAM(ctx));
    }
    hipError_t e = hipStreamSynchronize(MML_STREAM(ctx));
    hipFree(d_q);
    hipFree(d_idx);
    hipFree(d_d2);
    if (e != hipSuccess) {
        ctx->err = std::string("mml_knn5: ") + hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    return rc;
}

static void finish_stats(const double* s, mml_assoc_stats* o) {
    o->n_line = (int)s[0];
    o->n_plane = (int)s[1];
    o->n_line_used = (int)s[2];
    o->n_plane_used = (int)s[3];
    for (int k = 0; k < 9; ++k) o->normal_gram[k] = s[4 + k];
    // checkLocalizability (Estimator.cpp:536-565): smallest singular value of the M x 3 normal matrix =
    // sqrt(lambda_min(gram)); 3x3 symmetric eigenvalues by the trigonometric closed form (host, once per call)
    if (!(o->n_plane > 10)) {
        o->min_singular = -1;
    } else {
        const double* A = o->normal_gram;
        double p1 = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        double lam_min;
        double q = (A[0] + A[4] + A[8]) / 3.0;
        double p2 = (A[0] - q) * (A[0] - q) + (A[4] - q) * (A[4] - q) + (A[8] - q) * (A[8] - q) + 2 * p1;
        double p = sqrt(p2 / 6.0);
        if (p < 1e-300) {
            lam_min = q;
        } else {
            double Bm[9];
            for (int k = 0; k < 9; ++k) Bm[k] = (A[k] - ((k % 4 == 0) ? q : 0.0)) / p;
            double detB = Bm[0] * (Bm[4] * Bm[8] - Bm[5] * Bm[7]) - Bm[1] * (Bm[3] * Bm[8] - Bm[5] * Bm[6]) +
                          Bm[2] * (Bm[3] * Bm[7] - Bm[4] * Bm[6]);
            double r = detB / 2.0;
            r = r < -1 ? -1 : (r > 1 ? 1 : r);
            double phi = acos(r) / 3.0;
            lam_min = q + 2 * p * cos(phi + 2.0 * M_PI / 3.0);
        }
        o->min_singular = sqrt(lam_min > 0 ? lam_min : 0.0);
    }
    o->is_degenerate = o->min_singular < 3.0 ? 1 : 0;  // Estimator.cpp:772-775
}

static int associate_enqueue(mml_ctx* ctx, int first_slot, int count, const double* T_wl, double thres_dist, bool with_stats) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(T_wl != nullptr, MML_ERR_INVALID, "null T_wl");
    MML_REQUIRE(ctx->have_map[0] && ctx->have_map[1], MML_ERR_STATE, "mml_associate before both maps were set");
    double* d_T = ctx->d_pose_in + 64 * (size_t)first_slot;
    int rc = upload_doubles(ctx, d_T, T_wl, 16 * (size_t)count);
    if (rc != MML_OK) return rc;
    return mml_launch_associate(ctx, first_slot, count, d_T, thres_dist, with_stats);
}

int mml_associate(mml_ctx* ctx, int first_slot, int count, const double* T_wl, double thres_dist,
                  mml_assoc_stats* stats) {
    int rc = associate_enqueue(ctx, first_slot, count, T_wl, thres_dist, true);
    if (rc != MML_OK) return rc;
    if (stats) {
        double* h = stage_alloc(ctx, 16 * (size_t)count);  // pinned
        MML_HIP(hipMemcpyAsync(h, ctx->assoc_stats + 16 * (size_t)first_slot, sizeof(double) * 16 * (size_t)count,
                               hipMemcpyDeviceToHost, MML_STREAM(ctx)));
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
        for (int i = 0; i < count; ++i) finish_stats(h + 16 * i, &stats[i]);
    }
    return MML_OK;
}

int mml_factors_upload(mml_ctx* ctx, int slot, int kind, const double* rec, int n) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE((kind == 0 || kind == 1) && n >= 0 && (n == 0 || rec), MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(n <= ctx->MF, MML_ERR_CAPACITY, "more factors than max_features");
    ctx->stats_stale[slot] = 1;  // (the slot's statistics no longer describe its factors: recomputed when asked for)
    if (kind == 0) {
        std::vector<MmlLineFactor> h(n ? n : 1);
        for (int i = 0; i < n; ++i) {
            const double* r = rec + 10 * (size_t)i;
            for (int c = 0; c < 3; ++c) {
                h[i].ori[c] = (float)r[c];
                h[i].p1[c] = (float)r[3 + c];
                h[i].p2[c] = (float)r[6 + c];
            }
            h[i].src = i;
            h[i].error = r[9];
        }
        if (n) MML_HIP(hipMemcpyAsync(ctx->lf + (size_t)slot * ctx->MF, h.data(), sizeof(MmlLineFactor) * n, hipMemcpyHostToDevice, MML_STREAM(ctx)));
        MML_HIP(hipMemcpyAsync(ctx->ft_n + slot, &n, sizeof(int), hipMemcpyHostToDevice, MML_STREAM(ctx)));
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));  // (the staging vector goes out of scope)
    } else {
        std::vector<MmlPlaneFactor> h(n ? n : 1);
        for (int i = 0; i < n; ++i) {
            const double* r = rec + 10 * (size_t)i;
            for (int c = 0; c < 3; ++c) {
                h[i].ori[c] = (float)r[c];
                h[i].proj[c] = r[3 + c];
                h[i].omega[c] = (float)r[6 + c];
            }
            h[i].src = i;
            h[i]._pad = 0;
            h[i].error = r[9];
        }
        if (n) MML_HIP(hipMemcpyAsync(ctx->pf + (size_t)slot * ctx->MF, h.data(), sizeof(MmlPlaneFactor) * n, hipMemcpyHostToDevice, MML_STREAM(ctx)));
        MML_HIP(hipMemcpyAsync(ctx->ft_n + ctx->B + slot, &n, sizeof(int), hipMemcpyHostToDevice, MML_STREAM(ctx)));
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    }
    return MML_OK;
}

int mml_factors_download(mml_ctx* ctx, int slot, int kind, double* out, int* src, int capacity, int* n) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE((kind == 0 || kind == 1) && n, MML_ERR_INVALID, "bad arguments");
    int nf = 0;
    MML_HIP(hipMemcpyAsync(&nf, ctx->ft_n + kind * ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    MML_REQUIRE(nf >= 0, MML_ERR_CAPACITY, "feature stack overflowed");
    int cnt = 0;
    if (kind == 0) {
        std::vector<MmlLineFactor> h(nf ? nf : 1);
        if (nf) MML_HIP(hipMemcpy(h.data(), ctx->lf + (size_t)slot * ctx->MF, sizeof(MmlLineFactor) * nf, hipMemcpyDeviceToHost));
        for (int i = 0; i < nf; ++i) {
            if (h[i].src < 0) continue;
            if (cnt < capacity && out) {
                double* o = out + 10 * cnt;
                for (int c = 0; c < 3; ++c) {
                    o[c] = h[i].ori[c];
                    o[3 + c] = h[i].p1[c];
                    o[6 + c] = h[i].p2[c];
                }
                o[9] = h[i].error;
                if (src) src[cnt] = h[i].src;
            }
            ++cnt;
        }
    } else {
        std::vector<MmlPlaneFactor> h(nf ? nf : 1);
        if (nf) MML_HIP(hipMemcpy(h.data(), ctx->pf + (size_t)slot * ctx->MF, sizeof(MmlPlaneFactor) * nf, hipMemcpyDeviceToHost));
        for (int i = 0; i < nf; ++i) {
            if (h[i].src < 0) continue;
            if (cnt < capacity && out) {
                double* o = out + 10 * cnt;
                for (int c = 0; c < 3; ++c) {
                    o[c] = h[i].ori[c];
                    o[3 + c] = h[i].proj[c];
                    o[6 + c] = h[i].omega[c];
                }
                o[9] = h[i].error;
                if (src) src[cnt] = h[i].src;
            }
            ++cnt;
        }
    }
    *n = cnt;
    MML_REQUIRE(!out || cnt <= capacity, MML_ERR_CAPACITY, "factor download capacity too small");
    return MML_OK;
}

int mml_linearize_record(mml_ctx* ctx, int slot, const double* x, const double* T_bl, double plan_weight_tan,
                         double huber_delta, double* d_record) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(x && T_bl && d_record, MML_ERR_INVALID, "null argument");
    double h[22];
    memcpy(h, x, sizeof(double) * 6);
    memcpy(h + 6, T_bl, sizeof(double) * 16);
    double* d_par = ctx->d_pose_in + 64 * (size_t)slot;
    int rc = upload_doubles(ctx, d_par, h, 22);
    if (rc != MML_OK) return rc;
    return mml_launch_linearize(ctx, slot, d_par, d_par + 6, plan_weight_tan, huber_delta, d_record);
}

int mml_linearize_window(mml_ctx* ctx, int first_slot, int frames, int x_stride, const double* x, const double* T_bl,
                         double plan_weight_tan, double huber_delta, double* records) {
    CHECK_SLOTS(first_slot, frames);
    MML_REQUIRE(x && T_bl && records && x_stride >= 6 && frames <= 8, MML_ERR_INVALID, "bad arguments");
    double h[6 * 8 + 16];
    for (int f = 0; f < frames; ++f) memcpy(h + 6 * f, x + (size_t)x_stride * f, sizeof(double) * 6);
    memcpy(h + 6 * frames, T_bl, sizeof(double) * 16);
    // 64 doubles of parameter space per slot: poses of the window, then T_bl, in the first slot's block
    double* d_par = ctx->d_pose_in + 64 * (size_t)first_slot;
    int rc = upload_doubles(ctx, d_par, h, 6 * (size_t)frames + 16);
    if (rc != MML_OK) return rc;
    double* d_rec = ctx->d_rec + 32 * (size_t)first_slot;
    rc = mml_launch_linearize(ctx, first_slot, d_par, d_par + 6 * frames, plan_weight_tan, huber_delta, d_rec, frames);
    if (rc != MML_OK) return rc;
    MML_HIP(hipMemcpyAsync(records, d_rec, sizeof(double) * 32 * frames, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    return MML_OK;
}

int mml_linearize(mml_ctx* ctx, int slot, const double* x, const double* T_bl, double plan_weight_tan,
                  double huber_delta, double* H, double* g, double* cost) {
    CHECK_SLOTS(slot, 1);
    MML_REQUIRE(H && g && cost, MML_ERR_INVALID, "null output");
    double* d_rec = ctx->d_rec + 32 * (size_t)slot;
    int rc = mml_linearize_record(ctx, slot, x, T_bl, plan_weight_tan, huber_delta, d_rec);
    if (rc != MML_OK) return rc;
    double rec[32];
    MML_HIP(hipMemcpyAsync(rec, d_rec, sizeof(rec), hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            H[6 * a + b] = rec[k];
            H[6 * b + a] = rec[k];
            ++k;
        }
    for (int a = 0; a < 6; ++a) g[a] = rec[21 + a];
    *cost = rec[27];
    return MML_OK;
}

// enqueue only: upload x and T_bl for slots [first, first+count), launch the solver on the current lane
static int solve_enqueue(mml_ctx* ctx, int first_slot, int count, int window, const double* T_bl,
                         const mml_solve_opts* opts, const double* x, bool want_trace) {
    int rc = upload_doubles(ctx, ctx->d_x + 6 * (size_t)first_slot, x, 6 * (size_t)count);
    if (rc != MML_OK) return rc;
    double* d_Tbl = ctx->d_pose_in + 64 * (size_t)(first_slot + count) - 16;  // tail of this call's slice
    rc = upload_doubles(ctx, d_Tbl, T_bl, 16);
    if (rc != MML_OK) return rc;
    return mml_launch_solve(ctx, first_slot, count, window, d_Tbl, *opts, want_trace);
}

int mml_solve(mml_ctx* ctx, int first_slot, int count, int window, const double* T_bl, const mml_solve_opts* opts,
              double* x, mml_solve_summary* summaries, double* trace) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(T_bl && opts && x, MML_ERR_INVALID, "null argument");
    MML_REQUIRE(opts->max_num_iterations >= 0 && opts->max_num_iterations <= 64, MML_ERR_INVALID,
                "max_num_iterations must be in [0, 64]");
    int rc = solve_enqueue(ctx, first_slot, count, window, T_bl, opts, x, trace != nullptr);
    if (rc != MML_OK) return rc;
    const int nprob = count / window;
    double* hx = stage_alloc(ctx, 6 * (size_t)count + 8 * (size_t)nprob);  // pinned read-back area: poses, then summaries
    double* hs = hx + 6 * (size_t)count;
    MML_HIP(hipMemcpyAsync(hx, ctx->d_x + 6 * (size_t)first_slot, sizeof(double) * 6 * count, hipMemcpyDeviceToHost,
                           MML_STREAM(ctx)));
    MML_HIP(hipMemcpyAsync(hs, ctx->d_summ + 8 * (size_t)first_slot, sizeof(double) * 8 * (size_t)nprob, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    if (trace)
        MML_HIP(hipMemcpyAsync(trace, ctx->d_trace + (size_t)first_slot * 6 * 64, sizeof(double) * (size_t)nprob * opts->max_num_iterations * 6 * window,
                               hipMemcpyDeviceToHost, MML_STREAM(ctx)));
    MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
    if (window > 1 && !trace) {  // frame-parallel path: the first chunk of rounds ran; go on where a problem has not stopped
        bool open = false;
        for (int p = 0; p < nprob; ++p) open = open || hs[8 * p + 5] != 0.0;
        if (open) {
            const double* d_Tbl = ctx->d_pose_in + 64 * (size_t)(first_slot + count) - 16;
            rc = mml_window_solve_continue(ctx, first_slot, count, window, d_Tbl, *opts);
            if (rc != MML_OK) return rc;
            MML_HIP(hipMemcpyAsync(hx, ctx->d_x + 6 * (size_t)first_slot, sizeof(double) * 6 * count, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
            MML_HIP(hipMemcpyAsync(hs, ctx->d_summ + 8 * (size_t)first_slot, sizeof(double) * 8 * (size_t)nprob, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
            MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
        }
    }
    memcpy(x, hx, sizeof(double) * 6 * count);
    if (summaries)
        for (int p = 0; p < nprob; ++p) {
            summaries[p].iterations = (int)hs[8 * p];
            summaries[p].successful = (int)hs[8 * p + 1];
            summaries[p].initial_cost = hs[8 * p + 2];
            summaries[p].final_cost = hs[8 * p + 3];
            summaries[p].termination = (int)hs[8 * p + 4];
        }
    return MML_OK;
}

// ---- small host-side SO(3) helpers for the orchestration entry points (outside the kernels) ----------------
static void so3_exp_h(const double* phi, double* q) {  // sophus/so3.hpp:585-622
    double th2 = (phi[0] * phi[0] + phi[1] * phi[1]) + phi[2] * phi[2];
    double imag, real;
    if (th2 < 1e-20) {
        double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        double th = sqrt(th2), half = 0.5 * th;
        imag = sin(half) / th;
        real = cos(half);
    }
    q[0] = imag * phi[0];
    q[1] = imag * phi[1];
    q[2] = imag * phi[2];
    q[3] = real;
}
static void so3_log_h(const double* qin, double* phi) {  // sophus/so3.hpp:247-287
    double nq = sqrt((qin[0] * qin[0] + qin[2] * qin[2]) + (qin[1] * qin[1] + qin[3] * qin[3]));
    double q[4] = {qin[0] / nq, qin[1] / nq, qin[2] / nq, qin[3] / nq};
    double sn = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    double w = q[3], f;
    if (sn < 1e-20) {
        f = 2.0 / w - (2.0 / 3.0) * sn / (w * w * w);
    } else {
        double n = sqrt(sn);
        if (fabs(w) < 1e-10)
            f = (w > 0 ? M_PI : -M_PI) / n;
        else
            f = 2.0 * atan(n / w) / n;
    }
    phi[0] = f * q[0];
    phi[1] = f * q[1];
    phi[2] = f * q[2];
}
static void quat_to_R_h(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// T_bl = inverse(exTlb) (Estimator.cpp:157-159 / :1155-1156)
static void invert_extrinsic(const double* exTlb, double* T_bl) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T_bl[4 * r + c] = exTlb[4 * c + r];
    for (int r = 0; r < 3; ++r)
        T_bl[4 * r + 3] = -1.0 * ((T_bl[4 * r] * exTlb[3] + T_bl[4 * r + 1] * exTlb[7]) + T_bl[4 * r + 2] * exTlb[11]);
    T_bl[12] = T_bl[13] = T_bl[14] = 0;
    T_bl[15] = 1;
}
// transformTobeMapped = [Q * exRbl, Q * exPbl + P] (Estimator.cpp:1268-1270) from x = [t, phi]
static void pose_to_Twl(const double* q, const double* P, const double* T_bl, double* T) {
    double R[9];
    quat_to_R_h(q, R);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
            T[4 * r + c] = (R[3 * r] * T_bl[c] + R[3 * r + 1] * T_bl[4 + c]) + R[3 * r + 2] * T_bl[8 + c];
        T[4 * r + 3] = ((R[3 * r] * T_bl[3] + R[3 * r + 1] * T_bl[7]) + R[3 * r + 2] * T_bl[11]) + P[r];
    }
    T[12] = T[13] = T[14] = 0;
    T[15] = 1;
}

int mml_estimate(mml_ctx* ctx, int first_slot, int count, const double* exTlb, double* P, double* Q, int max_outer,
                 int inner_iters, mml_estimate_info* info) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(exTlb && P && Q && max_outer >= 1 && inner_iters >= 0, MML_ERR_INVALID, "bad arguments");
    MML_REQUIRE(ctx->have_map[0] && ctx->have_map[1], MML_ERR_STATE, "mml_estimate before both maps were set");
    double T_bl[16];
    invert_extrinsic(exTlb, T_bl);
    std::vector<int> done(count, 0), outer(count, 0), degen(count, 0);
    std::vector<double> x(6 * (size_t)count), Twl(16 * (size_t)count);
    std::vector<mml_assoc_stats> st(count);
    std::vector<int> last_fn(2 * (size_t)count, 0);  // the slots' stack sizes as the last result records reported them
    bool have_fn = false;
    double thres = 25.0;  // Estimator.cpp:1207
    mml_solve_opts so;
    so.max_num_iterations = inner_iters;
    so.fixed_iterations = 0;
    so.huber_delta = 0.1 / 1.5e-3;  // :1221
    so.plan_weight_tan = 0.0;       // :1206
    for (int it = 0; it < max_outer; ++it) {
        bool any = false;
        for (int i = 0; i < count; ++i)
            if (!done[i]) any = true;
        if (!any) break;
        for (int i = 0; i < count; ++i) {
            so3_log_h(Q + 4 * i, &x[6 * i + 3]);  // vector2double :937-950
            x[6 * i] = P[3 * i];
            x[6 * i + 1] = P[3 * i + 1];
            x[6 * i + 2] = P[3 * i + 2];
            pose_to_Twl(Q + 4 * i, P + 3 * i, T_bl, &Twl[16 * i]);
        }
        // association and solve are enqueued back to back and read back together: one host synchronisation per outer
        // iteration (the statistics of the association only feed the degeneracy flag, nothing the solve waits for)
        std::vector<double> xs = x;
        int rc;
        if (count <= kPackMaxSlots) {
            // (the live path: one copy down -- T_wl | x | T_bl --, one record per slot up -- pose, stack sizes, statistics; see mml_step)
            const size_t c = (size_t)count, n = 22 * c + 16;
            double* hp = stage_alloc(ctx, n);
            double* dp = ctx->d_pose_in + 64 * (size_t)first_slot;
            memcpy(hp, Twl.data(), sizeof(double) * 16 * c);
            memcpy(hp + 16 * c, x.data(), sizeof(double) * 6 * c);
            memcpy(hp + 22 * c, T_bl, sizeof(double) * 16);
            MML_HIP(hipMemcpyAsync(dp, hp, sizeof(double) * n, hipMemcpyHostToDevice, MML_STREAM(ctx)));
            rc = mml_launch_associate(ctx, first_slot, count, dp, thres, true);
            if (rc != MML_OK) return rc;
            thres = (it == 0) ? 10.0 : 1.0;  // :1377-1381
            double* d_res = ctx->d_result + MML_SOLVE_RESULT * (size_t)first_slot;
            rc = mml_launch_solve(ctx, first_slot, count, 1, dp + 22 * c, so, false, dp + 16 * c, d_res);
            if (rc != MML_OK) return rc;
            double* h_res = stage_alloc(ctx, MML_SOLVE_RESULT * c);
            MML_HIP(hipMemcpyAsync(h_res, d_res, sizeof(double) * MML_SOLVE_RESULT * c, hipMemcpyDeviceToHost, MML_STREAM(ctx)));
            MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
            for (int i = 0; i < count; ++i) {
                const double* r = h_res + MML_SOLVE_RESULT * (size_t)i;
                finish_stats(r + 8, &st[i]);
                memcpy(&xs[6 * (size_t)i], r, sizeof(double) * 6);
                last_fn[2 * i] = (int)r[6];
                last_fn[2 * i + 1] = (int)r[7];
            }
            have_fn = true;
        } else {
        rc = mml_associate(ctx, first_slot, count, Twl.data(), thres, nullptr);
        if (rc != MML_OK) return rc;
        thres = (it == 0) ? 10.0 : 1.0;  // :1377-1381
        rc = solve_enqueue(ctx, first_slot, count, 1, T_bl, &so, xs.data(), false);
        if (rc != MML_OK) return rc;
        double* h_back = stage_alloc(ctx, 22 * (size_t)count);  // pinned: stats (16 per slot), then poses (6 per slot)
        MML_HIP(hipMemcpyAsync(h_back, ctx->assoc_stats + 16 * (size_t)first_slot, sizeof(double) * 16 * count, hipMemcpyDeviceToHost,
                               MML_STREAM(ctx)));
        MML_HIP(hipMemcpyAsync(h_back + 16 * (size_t)count, ctx->d_x + 6 * (size_t)first_slot, sizeof(double) * 6 * count,
                               hipMemcpyDeviceToHost, MML_STREAM(ctx)));
        MML_HIP(hipStreamSynchronize(MML_STREAM(ctx)));
        for (int i = 0; i < count; ++i) finish_stats(h_back + 16 * (size_t)i, &st[i]);
        memcpy(xs.data(), h_back + 16 * (size_t)count, sizeof(double) * 6 * count);
        }
        for (int i = 0; i < count; ++i) {
            if (done[i]) continue;
            if (st[i].is_degenerate) degen[i] = 1;
            double qa[4];
            so3_exp_h(&xs[6 * i + 3], qa);  // double2vector :952-964
            const double* qb = Q + 4 * i;
            // angularDistance (:1444): d = qb * conj(qa); 2 atan2(|d.vec|, |d.w|)
            double cx = -qa[0], cy = -qa[1], cz = -qa[2], cw = qa[3];
            double dx = qb[3] * cx + qb[0] * cw + qb[1] * cz - qb[2] * cy;
            double dy = qb[3] * cy + qb[1] * cw + qb[2] * cx - qb[0] * cz;
            double dz = qb[3] * cz + qb[2] * cw + qb[0] * cy - qb[1] * cx;
            double dw = qb[3] * cw - qb[0] * cx - qb[1] * cy - qb[2] * cz;
            double deltaR = (2.0 * atan2(sqrt((dx * dx + dy * dy) + dz * dz), fabs(dw))) * 180.0 / M_PI;
            double ex = P[3 * i] - xs[6 * i], ey = P[3 * i + 1] - xs[6 * i + 1], ez = P[3 * i + 2] - xs[6 * i + 2];
            double deltaT = sqrt((ex * ex + ey * ey) + ez * ez);
            for (int c = 0; c < 3; ++c) P[3 * i + c] = xs[6 * i + c];
            for (int c = 0; c < 4; ++c) Q[4 * i + c] = qa[c];
            outer[i] = it + 1;
            if ((deltaR < 0.05 && deltaT < 0.05) || (it + 1) == max_outer) done[i] = 1;  // :1448
        }
    }
    if (info && have_fn) {
        for (int i = 0; i < count; ++i) {
            info[i].outer_iterations = outer[i];
            info[i].is_degenerate = degen[i];
            info[i].n_corner_feat = last_fn[2 * i];
            info[i].n_surf_feat = last_fn[2 * i + 1];
        }
    } else if (info) {
        std::vector<int> fn(2 * (size_t)ctx->B);
        MML_HIP(hipMemcpy(fn.data(), ctx->ft_n, sizeof(int) * fn.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; ++i) {
            info[i].outer_iterations = outer[i];
            info[i].is_degenerate = degen[i];
            info[i].n_corner_feat = fn[first_slot + i];
            info[i].n_surf_feat = fn[ctx->B + first_slot + i];
        }
    }
    return MML_OK;
}

int mml_step(mml_ctx* ctx, int first_slot, int count, const double* dR, const double* dt, const double* exTlb,
             double thres_dist, int gn_iters, double* x_inout) {
    CHECK_SLOTS(first_slot, count);
    MML_REQUIRE(dR && dt && exTlb && x_inout, MML_ERR_INVALID, "null argument");
    MML_REQUIRE(gn_iters >= 0 && gn_iters <= 64, MML_ERR_INVALID, "gn_iters must be in [0, 64]");
    MML_REQUIRE(ctx->have_map[0] && ctx->have_map[1], MML_ERR_STATE, "mml_step before both maps were set");
    double T_bl[16];
    invert_extrinsic(exTlb, T_bl);
    mml_solve_opts so;
    so.max_num_iterations = gn_iters;
    so.fixed_iterations = 1;
    so.huber_delta = 0.1 / 1.5e-3;
    so.plan_weight_tan = 0.0;
    // Sub-batches on independent streams (`lanes`, mml_set_lanes): contiguous slot ranges, each the whole chain on its own
    // stream, so that the tail of one kernel overlaps the head of another.  Scans are independent, results identical
    // (tests/test_gpu_shapes.py, bench.py replica_check).  What the lanes do NOT buy is co-residency of unlike kernels: every
    // kernel of the chain fills the CUs' registers / LDS by itself at these batch sizes.  Round 5 measured the other
    // schedule -- pieces of 512 .. 2048 scans, stream 0 extracting piece i + 1 while other streams run the back end of piece
    // i, 2 .. 5 streams, three stage-to-stream maps: 306 k .. 330 k scans/s against 355 k for two lanes (HISTORY.md).
    const int lanes = (count >= 64 && ctx->lanes_enabled) ? ctx->n_lanes : 1;
    const int n_pieces = lanes;
    const int chunk = (count + n_pieces - 1) / n_pieces;
    std::vector<double> Twl(16 * (size_t)chunk);
    // (the staging ring: a wrap in the middle of the call would drain the streams; wrap now if this call does not fit)
    {
        const size_t need = (7 + 16 + 12 + 6 + 8) * (size_t)count + 64 * (size_t)n_pieces + 1024;
        if (ctx->stage_cursor + need > ctx->h_stage_doubles && need <= ctx->h_stage_doubles) {
            int rw = mml_sync_all(ctx);
            if (rw != MML_OK) return rw;
            ctx->stage_cursor = 0;
        }
    }
    double* h_x = stage_alloc(ctx, 7 * (size_t)count + 1);  // pinned read-back area: poses, then the two stack-size arrays
    int* h_ftn = reinterpret_cast<int*>(h_x + 6 * (size_t)count);
    int rc = MML_OK;
    // A handful of slots (the live path): every small transfer is a launch of its own in the dependency chain (a blit kernel of
    // ~4 us plus the gap in front of it), and a one-scan step made eight of them.  All parameter blocks of the call -- sweep
    // motions, association poses, start poses, T_bl -- go down in ONE copy into the call's slice of d_pose_in, and everything read
    // back afterwards (poses, stack sizes) comes up in ONE record per slot that k_solve leaves (MML_SOLVE_RESULT doubles).
    const bool packed = count <= kPackMaxSlots;
    double *dp = nullptr, *h_res = nullptr;
    if (packed) {
        const size_t c = (size_t)count, n = 34 * c + 16;  // und [0, 12c) | T_wl [12c, 28c) | x [28c, 34c) | T_bl [34c, 34c + 16)
        double* hp = stage_alloc(ctx, n);
        dp = ctx->d_pose_in + 64 * (size_t)first_slot;
        for (size_t i = 0; i < c; ++i) {
            memcpy(hp + 12 * i, dR + 9 * i, sizeof(double) * 9);
            memcpy(hp + 12 * i + 9, dt + 3 * i, sizeof(double) * 3);
            double q[4];
            so3_exp_h(x_inout + 6 * i + 3, q);
            pose_to_Twl(q, x_inout + 6 * i, T_bl, hp + 12 * c + 16 * i);
        }
        memcpy(hp + 28 * c, x_inout, sizeof(double) * 6 * c);
        memcpy(hp + 34 * c, T_bl, sizeof(double) * 16);
        if (hipMemcpyAsync(dp, hp, sizeof(double) * n, hipMemcpyHostToDevice, MML_STREAM(ctx)) != hipSuccess) return MML_ERR_HIP;
        h_res = stage_alloc(ctx, MML_SOLVE_RESULT * c);
    }
    // Entry points outside mml_step enqueue on stream 0 (mml_scan_upload's copies, a staged mml_extract ...): the other streams
    // start behind whatever stream 0 holds at this point.  (Found by running 80 slots on the default two lanes straight
    // after their uploads: the last slot's copy was still in flight when lane 1 began to bucket it.)
    if (lanes > 1) {
        if (hipEventRecord(ctx->lane_mark[0], ctx->streams[0]) != hipSuccess) return MML_ERR_HIP;
        for (int l = 1; l < mml_ctx::MAX_LANES; ++l)
            if (hipStreamWaitEvent(ctx->streams[l], ctx->lane_mark[0], 0) != hipSuccess) return MML_ERR_HIP;
    }
    auto piece_span = [&](int p, int& f, int& c) {
        f = first_slot + p * chunk;
        c = std::min(chunk, first_slot + count - f);
        return c > 0;
    };
    auto run_stage = [&](int stage, int f, int c) -> int {
        const int off = f - first_slot;
        switch (stage) {
            case 0: return mml_launch_extract(ctx, f, c, false);
            case 1:
                if (packed) return mml_launch_undistort(ctx, f, c, dp);
                return mml_undistort(ctx, f, c, dR + 9 * (size_t)off, dt + 3 * (size_t)off);
            case 2: return mml_launch_downsample(ctx, f, c);
            case 3:
                if (packed) return mml_launch_associate(ctx, f, c, dp + 12 * (size_t)count, thres_dist, false);
                for (int i = 0; i < c; ++i) {
                    double q[4];
                    so3_exp_h(x_inout + 6 * (size_t)(off + i) + 3, q);
                    pose_to_Twl(q, x_inout + 6 * (size_t)(off + i), T_bl, &Twl[16 * (size_t)i]);
                }
                return associate_enqueue(ctx, f, c, Twl.data(), thres_dist, false);  // (nobody reads the statistics of a step)
            default: {
                if (packed) {
                    int r = mml_launch_solve(ctx, f, c, 1, dp + 34 * (size_t)count, so, false, dp + 28 * (size_t)count,
                                             ctx->d_result + MML_SOLVE_RESULT * (size_t)f);
                    if (r != MML_OK) return r;
                    if (hipMemcpyAsync(h_res, ctx->d_result + MML_SOLVE_RESULT * (size_t)f, sizeof(double) * MML_SOLVE_RESULT * c,
                                       hipMemcpyDeviceToHost, MML_STREAM(ctx)) != hipSuccess)
                        return MML_ERR_HIP;
                    return MML_OK;
                }
                int r = solve_enqueue(ctx, f, c, 1, T_bl, &so, x_inout + 6 * (size_t)off, false);
                if (r != MML_OK) return r;
                hipError_t e = hipMemcpyAsync(h_x + 6 * (size_t)off, ctx->d_x + 6 * (size_t)f, sizeof(double) * 6 * c, hipMemcpyDeviceToHost,
                                              MML_STREAM(ctx));
                if (e != hipSuccess) {
                    ctx->err = std::string("mml_step read-back: ") + hipGetErrorString(e);
                    return MML_ERR_HIP;
                }
                return MML_OK;
            }
        }
    };
    {
        // The stages are enqueued stage by stage across the lanes (each lane's stream keeps its own order): every lane has
        // its first kernels in its queue within a few tens of microseconds, instead of lane 3 waiting for the host to finish
        // enqueueing the whole chains of lanes 0..2.
        for (int stage = 0; stage < 5 && rc == MML_OK; ++stage) {
            for (int l = 0; l < n_pieces && rc == MML_OK; ++l) {
                int f, c;
                if (!piece_span(l, f, c)) break;
                ctx->cur = l;
                rc = run_stage(stage, f, c);
            }
        }
    }
    // the per-slot stack sizes ride back with the poses: a negative count is the down-sampler's overflow mark
    if (rc == MML_OK) {
        ctx->cur = 0;
        for (int l = 1; l < mml_ctx::MAX_LANES && rc == MML_OK; ++l)
            if (hipStreamSynchronize(ctx->streams[l]) != hipSuccess) rc = MML_ERR_HIP;
        for (int kind = 0; kind < 2 && rc == MML_OK && !packed; ++kind)
            if (hipMemcpyAsync(h_ftn + (size_t)kind * count, ctx->ft_n + kind * ctx->B + first_slot, sizeof(int) * count,
                               hipMemcpyDeviceToHost, MML_STREAM(ctx)) != hipSuccess)
                rc = MML_ERR_HIP;
    }
    ctx->cur = 0;
    int rs = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    if (rs != MML_OK) return rs;
    if (packed)  // the result records: pose (6), the two stack sizes, (association statistics: nobody reads them here)
        for (int i = 0; i < count; ++i) {
            memcpy(h_x + 6 * (size_t)i, h_res + MML_SOLVE_RESULT * (size_t)i, sizeof(double) * 6);
            h_ftn[i] = (int)h_res[MML_SOLVE_RESULT * (size_t)i + 6];
            h_ftn[count + i] = (int)h_res[MML_SOLVE_RESULT * (size_t)i + 7];
        }
    // A negative stack size is the down-sampler's overflow mark.  Slots whose labelled cloud is merely too dense for the
    // LDS sort are redone through the global-sort filter and re-registered one by one (rare: > 8192 corner- or
    // surf-labelled points in one scan); a slot that exceeds max_features stays failed.  Either way every other slot of
    // the batch gets its pose, and a failed slot keeps the pose it came with.
    std::vector<int> bad;
    for (int i = 0; i < count; ++i)
        if (h_ftn[i] < 0 || h_ftn[count + i] < 0) bad.push_back(i);
    std::vector<char> failed((size_t)count, 0);
    for (int i : bad) {
        std::vector<int> redone;
        rc = mml_downsample_redo_overflow(ctx, first_slot + i, 1, &redone);
        bool ok = rc == MML_OK && !redone.empty();
        if (ok) {
            int fn[2] = {-1, -1};  // (on the ctx stream: the lanes are non-blocking streams, the null stream does not order with them)
            if (hipMemcpyAsync(&fn[0], ctx->ft_n + first_slot + i, sizeof(int), hipMemcpyDeviceToHost, MML_STREAM(ctx)) != hipSuccess ||
                hipMemcpyAsync(&fn[1], ctx->ft_n + ctx->B + first_slot + i, sizeof(int), hipMemcpyDeviceToHost, MML_STREAM(ctx)) != hipSuccess ||
                hipStreamSynchronize(MML_STREAM(ctx)) != hipSuccess)
                return MML_ERR_HIP;
            ok = fn[0] >= 0 && fn[1] >= 0;
        }
        if (ok) {
            double q[4], T1[16];
            so3_exp_h(x_inout + 6 * (size_t)i + 3, q);
            pose_to_Twl(q, x_inout + 6 * (size_t)i, T_bl, T1);
            rc = mml_associate(ctx, first_slot + i, 1, T1, thres_dist, nullptr);
            if (rc != MML_OK) return rc;
            double xs[6];
            memcpy(xs, x_inout + 6 * (size_t)i, sizeof(xs));
            rc = mml_solve(ctx, first_slot + i, 1, 1, T_bl, &so, xs, nullptr, nullptr);
            if (rc != MML_OK) return rc;
            memcpy(h_x + 6 * (size_t)i, xs, sizeof(xs));
        } else {
            failed[i] = 1;
        }
    }
    int n_failed = 0, first_failed = -1;
    for (int i = 0; i < count; ++i) {
        if (failed[i]) {
            if (first_failed < 0) first_failed = first_slot + i;
            ++n_failed;
            continue;
        }
        memcpy(x_inout + 6 * (size_t)i, h_x + 6 * (size_t)i, sizeof(double) * 6);
    }
    if (n_failed) {
        ctx->err = "down-sample overflowed max_features in " + std::to_string(n_failed) + " slot(s), first: slot " +
                   std::to_string(first_failed) + " (the other slots' poses were returned)";
        return MML_ERR_CAPACITY;
    }
    return MML_OK;
}

int mml_time_offset_search(mml_ctx* ctx, const float* velo_xyz, int n_velo, const float* tf, const float* livox_xyz,
                           int n_livox, int search_resolution, int sliced_points, float* nn_d2, double* window_error,
                           int capacity, int* n_windows, int* best_window, double* lowest_error) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(n_velo >= 0 && n_livox >= 0 && (n_velo == 0 || velo_xyz) && (n_livox == 0 || livox_xyz), MML_ERR_INVALID,
                "bad point buffers");
    MML_REQUIRE(search_resolution >= 1 && sliced_points >= 1, MML_ERR_INVALID, "search_resolution / sliced_points must be >= 1");
    MML_REQUIRE(n_windows && best_window && lowest_error, MML_ERR_INVALID, "null output");
    MML_REQUIRE(n_velo <= ctx->MM, MML_ERR_CAPACITY, "Velodyne cloud exceeds max_map_points");
    MML_REQUIRE(n_velo > 0 || n_livox == 0, MML_ERR_INVALID, "nearest-neighbour search in an empty cloud");
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    ctx->cur = 0;
    // windows: cnt = 0, 1, ... while cnt * res + sliced < n_livox
    int nwin = 0;
    if (n_livox > sliced_points) nwin = (n_livox - sliced_points - 1) / search_resolution + 1;
    *n_windows = nwin;
    *best_window = -1;
    *lowest_error = 1000000.0;
    if (n_livox == 0) return MML_OK;
    // a one-off calibration step: its buffers live for the call only
    MmlGrid g;
    float4* d_v4 = nullptr;
    float *d_v = nullptr, *d_l = nullptr, *d_nn = nullptr, *d_tf = nullptr;
    double* d_err = nullptr;
    std::vector<void*> owned;
    auto take = [&](void** p, size_t bytes) {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return false;
        owned.push_back(*p);
        return true;
    };
    auto release = [&]() {
        for (void* p : owned) hipFree(p);
    };
    const size_t nv = (size_t)(n_velo > 0 ? n_velo : 1), nl = (size_t)n_livox;
    bool ok = take((void**)&g.pts, sizeof(float4) * nv) && take((void**)&g.cell_start, sizeof(int) * (4 * (size_t)ctx->MM + 4096 + 2)) &&
              take((void**)&d_v4, sizeof(float4) * nv) && take((void**)&d_v, sizeof(float) * 3 * nv) &&
              take((void**)&d_l, sizeof(float) * 3 * nl) && take((void**)&d_nn, sizeof(float) * nl) &&
              take((void**)&d_err, sizeof(double) * (size_t)(nwin > 0 ? nwin : 1)) && take((void**)&d_tf, sizeof(float) * 16);
    if (!ok) {
        release();
        ctx->err = "mml_time_offset_search: device allocation failed";
        return MML_ERR_HIP;
    }
    hipStream_t s = MML_STREAM(ctx);
    hipError_t e = hipSuccess;
    if (n_velo) e = hipMemcpyAsync(d_v, velo_xyz, sizeof(float) * 3 * nv, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_l, livox_xyz, sizeof(float) * 3 * nl, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && tf) e = hipMemcpyAsync(d_tf, tf, sizeof(float) * 16, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess)
        rc = mml_launch_time_offset(ctx, g, d_v4, d_v, n_velo, tf ? d_tf : nullptr, d_l, n_livox, search_resolution, sliced_points,
                                    nwin, d_nn, d_err);
    std::vector<double> herr((size_t)(nwin > 0 ? nwin : 1));
    if (e == hipSuccess && rc == MML_OK && nn_d2) e = hipMemcpyAsync(nn_d2, d_nn, sizeof(float) * nl, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == MML_OK && nwin) e = hipMemcpyAsync(herr.data(), d_err, sizeof(double) * nwin, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    release();
    if (e != hipSuccess) {
        ctx->err = std::string("mml_time_offset_search: ") + hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    if (rc != MML_OK) return rc;
    for (int c = 0; c < nwin; ++c) {  // :1141-1150
        if (window_error && c < capacity) window_error[c] = herr[c];
        if (herr[c] < *lowest_error) {
            *lowest_error = herr[c];
            *best_window = c;
        }
    }
    return MML_OK;
}

int mml_set_lanes(mml_ctx* ctx, int lanes) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(lanes >= 1 && lanes <= mml_ctx::MAX_LANES, MML_ERR_INVALID, "lanes must be in [1, 8]");
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    ctx->n_lanes = lanes;
    return MML_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------------
int mml_profile_enable(mml_ctx* ctx, int on) {
    if (!ctx) return MML_ERR_INVALID;
    ctx->profiling = on != 0;
    return MML_OK;
}

static int drain_pending(mml_ctx* ctx) {
    int rc0 = mml_sync_all(ctx);
    if (rc0 != MML_OK) return rc0;
    for (auto& pe : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            ctx->stages[pe.stage].total_ms += ms;
            ctx->stages[pe.stage].launches += 1;
        }
        ctx->event_pool.push_back(pe.a);
        ctx->event_pool.push_back(pe.b);
    }
    ctx->pending.clear();
    return MML_OK;
}

int mml_profile_reset(mml_ctx* ctx) {
    if (!ctx) return MML_ERR_INVALID;
    int rc = drain_pending(ctx);
    for (auto& s : ctx->stages) {
        s.total_ms = 0;
        s.launches = 0;
    }
    return rc;
}

int mml_profile_get(mml_ctx* ctx, mml_profile* out) {
    if (!ctx || !out) return MML_ERR_INVALID;
    int rc = drain_pending(ctx);
    if (rc != MML_OK) return rc;
    out->n_stages = (int)std::min<size_t>(ctx->stages.size(), MML_MAX_STAGES);
    for (int i = 0; i < out->n_stages; ++i) {
        out->name[i] = ctx->stages[i].name.c_str();
        out->total_ms[i] = ctx->stages[i].total_ms;
        out->launches[i] = ctx->stages[i].launches;
    }
    return MML_OK;
}

int mml_extract_queue_counts(mml_ctx* ctx, int slot, int* redo, int* brk) {
    CHECK_SLOTS(slot, 1);
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    if (brk) MML_HIP(hipMemcpy(brk, ctx->brk_cnt + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (redo) MML_HIP(hipMemcpy(redo, ctx->brk_cnt + ctx->B + slot, sizeof(int), hipMemcpyDeviceToHost));  // (redo_cnt = brk_cnt + B)
    return MML_OK;
}

int mml_associate_far_count(mml_ctx* ctx, int* n) {
    if (!ctx || !n) return MML_ERR_INVALID;
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    int h[mml_ctx::MAX_LANES];
    MML_HIP(hipMemcpy(h, ctx->d_misc + 32, sizeof(h), hipMemcpyDeviceToHost));  // (one counter per stream lane: map_assoc.hip hard_count)
    *n = 0;
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l) *n += h[l];
    return MML_OK;
}

int mml_device_info(mml_ctx* ctx, char* name, int name_cap, int* cus, size_t* hbm_bytes) {
    if (!ctx) return MML_ERR_INVALID;
    hipDeviceProp_t prop;
    MML_HIP(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap > 0) {
        strncpy(name, prop.name, name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return MML_OK;
}

}  // extern "C"

namespace {
__global__ void k_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
// The copy a streaming kernel of this library can actually match: 16 bytes per lane, four independent loads in flight per
// thread, non-temporal loads and stores (nothing is re-read), one workgroup per 16 KB so that the whole array is in flight on
// 8 wavefronts per SIMD (12 VGPRs).  MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; k_undistort sustains 5.8.
typedef float copy_v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy16_nt(const copy_v4f* __restrict__ a, copy_v4f* __restrict__ b, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    copy_v4f v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (base + 256 * r < n) v[r] = __builtin_nontemporal_load(a + base + 256 * r);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (base + 256 * r < n) __builtin_nontemporal_store(v[r], b + base + 256 * r);
}
}  // namespace

extern "C" int mml_copy_bandwidth(mml_ctx* ctx, size_t bytes, int reps, double* gbps) {
    if (!ctx || !gbps || reps <= 0) return MML_ERR_INVALID;
    MML_HIP(hipSetDevice(ctx->device));
    size_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&a), n * 16));
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&b), n * 16));
    MML_HIP(hipMemsetAsync(a, 1, n * 16, MML_STREAM(ctx)));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // two forms, the better one is reported: a grid-stride loop of plain 16-byte copies, and the non-temporal one above
    double best = 0.0;
    hipError_t e = hipSuccess;
    for (int form = 0; form < 2 && e == hipSuccess; ++form) {
        auto launch = [&]() {
            if (form == 0)
                hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, MML_STREAM(ctx), a, b, n);
            else
                hipLaunchKernelGGL(k_copy16_nt, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, MML_STREAM(ctx),
                                   reinterpret_cast<const copy_v4f*>(a), reinterpret_cast<copy_v4f*>(b), n);
        };
        launch();
        hipEventRecord(e0, MML_STREAM(ctx));
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1, MML_STREAM(ctx));
        e = hipStreamSynchronize(MML_STREAM(ctx));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) best = std::max(best, (2.0 * n * 16 * reps) / (ms * 1e-3) / 1e9);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(a);
    hipFree(b);
    if (e != hipSuccess) {
        ctx->err = hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    *gbps = best;
    return MML_OK;
}

namespace {
// tools/issue_probe.hip in small: eight independent chains per lane of one instruction class, every SIMD eight wavefronts deep
template <int KIND>
__global__ __launch_bounds__(256) void k_issue_rate(float* out, float a, float b, int ia) {
    float x[8];
    int u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = (float)(threadIdx.x + i);
        u[i] = (int)threadIdx.x * 7 + i;
    }
    for (int it = 0; it < 2048; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);
            if constexpr (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ia));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int mml_issue_rate(mml_ctx* ctx, int kind, int reps, double* wave_instr_per_s) {
    if (!ctx || !wave_instr_per_s || reps <= 0 || kind < 0 || kind > 1) return MML_ERR_INVALID;
    MML_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    MML_HIP(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 64;  // eight workgroups of four wavefronts per CU, eight rounds of them
    float* d = nullptr;
    MML_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(float) * 256 * (size_t)blocks));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        if (kind == 0)
            hipLaunchKernelGGL(k_issue_rate<0>, dim3(blocks), dim3(256), 0, MML_STREAM(ctx), d, 0.999f, 0.001f, 3);
        else
            hipLaunchKernelGGL(k_issue_rate<1>, dim3(blocks), dim3(256), 0, MML_STREAM(ctx), d, 0.999f, 0.001f, 3);
    };
    launch();
    double best = 0.0;
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps && e == hipSuccess; ++r) {
        hipEventRecord(e0, MML_STREAM(ctx));
        launch();
        hipEventRecord(e1, MML_STREAM(ctx));
        e = hipStreamSynchronize(MML_STREAM(ctx));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && ms > 0) best = std::max(best, (double)blocks * 4 * 2048 * 8 / (ms * 1e-3));
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    if (e != hipSuccess) {
        ctx->err = hipGetErrorString(e);
        return MML_ERR_HIP;
    }
    *wave_instr_per_s = best;
    return MML_OK;
}

int mml_stage_begin(mml_ctx* ctx, const char* name) {
    if (!ctx->profiling) return -1;
    int idx = -1;
    for (size_t i = 0; i < ctx->stages.size(); ++i)
        if (ctx->stages[i].name == name) idx = (int)i;
    if (idx < 0) {
        MmlStageTimer t;
        t.name = name;
        ctx->stages.push_back(t);
        idx = (int)ctx->stages.size() - 1;
    }
    mml_ctx::Pending pe;
    pe.stage = idx;
    auto get = [&]() {
        hipEvent_t e;
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            hipEventCreate(&e);
        }
        return e;
    };
    pe.a = get();
    pe.b = get();
    hipEventRecord(pe.a, MML_STREAM(ctx));
    ctx->pending.push_back(pe);
    return (int)ctx->pending.size() - 1;
}

void mml_stage_end(mml_ctx* ctx, int token) {
    if (token < 0) return;
    hipEventRecord(ctx->pending[token].b, MML_STREAM(ctx));
}

int mml_sync_all(mml_ctx* ctx) {
    for (int l = 0; l < mml_ctx::MAX_LANES; ++l) MML_HIP(hipStreamSynchronize(ctx->streams[l]));
    MML_HIP(hipStreamSynchronize(ctx->copy_stream));
    for (auto& u : ctx->uploads) ctx->upload_event_pool.push_back(u.done);
    ctx->uploads.clear();
    return MML_OK;
}

// Orders every lane behind the batch uploads still in flight on slots [first, first + count); finished uploads retire.
int mml_uploads_wait(mml_ctx* ctx, int first, int count) {
    for (size_t i = 0; i < ctx->uploads.size();) {
        mml_ctx::Upload& u = ctx->uploads[i];
        if (hipEventQuery(u.done) == hipSuccess) {
            ctx->upload_event_pool.push_back(u.done);
            ctx->uploads.erase(ctx->uploads.begin() + i);
            continue;
        }
        if (u.first < first + count && first < u.first + u.count)
            for (int l = 0; l < mml_ctx::MAX_LANES; ++l) MML_HIP(hipStreamWaitEvent(ctx->streams[l], u.done, 0));
        ++i;
    }
    return MML_OK;
}
