// mml_internal.h -- private definitions shared by the HIP translation units of libmmloam_hip.so.
// gfx950 (MI355X) only.  All device code is compiled with -ffp-contract=off: the reference arithmetic
// is x86-64 SSE2 without FMA and the parity tests are bit-exact on float paths.
#ifndef MML_INTERNAL_H
#define MML_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "mmloam_hip.h"

#define MML_WAVE 64

namespace mml_und {
// A double literal as a scalar-register operand.  gfx950 has no 64-bit literal in VOP3: left alone the compiler materialises
// each constant of a polynomial with two v_mov_b32 next to its FMA (three vector instructions per term in kernels that are
// bound by vector issue); from an SGPR pair the term is ONE v_fma_f64 and the two s_mov_b32 go to the scalar unit.
__device__ __forceinline__ double sconst(double v) {
    asm("" : "+s"(v));
    return v;
}
}  // namespace mml_und
// One-pass bucketing (feature.hip k_assign_onepass): a scan line is stored as up to MML_SEG_MAX segments, one per MML_OP_BLK-point
// block of the raw scan; MML_SEG_FLAT bounds (blocks x lines) of one sensor.
#define MML_SEG_MAX 16
#define MML_OP_BLK 4096
#define MML_SEG_FLAT 512
#define MML_VOXEL_LDS_CAP 8192  // labelled points per (slot, kind) that k_voxel sorts in LDS; more go the global-sort way

// ---- factor records kept on the device (SoA would save little: every field is read once per GN pass) ----
struct MmlLineFactor {   // Estimator.h:59-84 FeatureLine (values are floats in the reference, :256-271)
    float ori[3];
    float p1[3];
    float p2[3];
    int src;             // feature index, -1 = no factor for this feature
    double error;        // FeatureLine::ComputeError
};
struct MmlPlaneFactor {  // Estimator.h:105-122 FeaturePlanVec with sqrt_info replaced by omega
    float ori[3];
    float omega[3];
    double proj[3];
    double error;
    int src;
    int _pad;
};

struct MmlGrid {         // radix-sorted uniform grid over one map cloud (replaces pcl::KdTreeFLANN)
    float4* pts = nullptr;      // sorted by cell key: x, y, z, original index (bit pattern)
    int* cell_start = nullptr;  // ncell + 1
    uint16_t* tags = nullptr;   // optional: cube index of every sorted point (global map), else nullptr
    int m = 0;
    float origin[3] = {0.f, 0.f, 0.f};
    float cell = 1.f;
    float inv_cell = 1.f;
    int dim[3] = {1, 1, 1};
    int ncell = 1;
};

struct MmlStageTimer {
    std::string name;
    double total_ms = 0;
    long launches = 0;
};

struct MmlComm;   // comm.hip: RCCL communicator + window-solve buffers
struct MmlFwDev;  // fullwindow_dev.hip: parameter / scratch buffers of the device-resident full-window solve

struct mml_ctx {
    mml_config cfg;
    MmlComm* comm = nullptr;
    MmlFwDev* fwdev = nullptr;
    // frame-parallel window solve (solve.hip): one state machine copy, 4 counters and two record buffers per slot
    void* wstate = nullptr;
    double* wrec = nullptr;
    double* waux = nullptr;
    bool window_frame_parallel = true;
    bool assoc_group_search = true;  // mml_associate on <= 8 slots: 16 lanes per feature from ring 0 (map_assoc.hip)
    struct WinGraph {  // captured launch chain of one frame-parallel window solve
        int first, count, W, max_iters, fixed;
        double huber, w_tan;
        hipStream_t stream;
        const double* Tbl;
        bool second;
        hipGraphExec_t exec;
    };
    std::vector<WinGraph> win_graphs;
    int device = 0;
    // `lanes`: independent HIP streams.  Entry points enqueue on lane `cur` (0 unless mml_step is pipelining
    // sub-batches); per-call scratch is sliced by slot index so lanes never share a byte.
    static constexpr int MAX_LANES = 8;
    hipStream_t streams[MAX_LANES] = {};
    // mml_scan_upload_batch copies on its own stream, so that the copy of one slot range runs under the kernels of
    // another; an entry point that touches a slot range first makes the lanes wait for the uploads still in flight on it
    hipStream_t copy_stream = nullptr;
    hipEvent_t lane_mark[MAX_LANES] = {};
    struct Upload {
        int first, count;
        hipEvent_t done;
    };
    std::vector<Upload> uploads;
    std::vector<hipEvent_t> upload_event_pool;
    int n_lanes = 1;
    int cur = 0;
    bool lanes_enabled = true;
    std::string err;

    int B = 0, NV = 0, NL = 0, NT = 0, L = 0, MF = 0;

    // inputs
    float4* velo_in = nullptr;            // B * NV
    mml_livox_point* livox_in = nullptr;  // B * NL
    int* d_n_in = nullptr;                // B * 2 (velo, livox)
    std::vector<int> h_n_in;
    std::vector<char> stats_stale;        // B: 1 while assoc_stats of the slot do not describe its current factors (mml_ensure_assoc_stats)
    std::vector<char> raw_extracted;      // B: 1 while the slot's raw buffers are the ones its extracted state was made from

    // per raw point scratch
    uint8_t* raw_line = nullptr;  // B * NT   line id or 255
    float* raw_ori = nullptr;     // B * NV   -atan2(y,x) as float

    // line-bucketed points
    float4* ln_pts = nullptr;   // B * NT
    int* ln_gidx = nullptr;     // B * NT  fused index of the point: >= 0 kept, -1 dropped, -2 Livox beyond far_th
    int* ln_rel = nullptr;      // B * NT  bits of its in-sweep time, normal_x (two 4-byte arrays: the undistortion reads the time only, the
                                //         label pass the index only; as one 8-byte record each of them fetched both)
    int* line_start = nullptr;  // B * L   start of the line in LINE ORDER (the index space of ln_curv / ln_refl / ln_attr)
    int* line_len = nullptr;    // B * L
    // where the points of a line are STORED (ln_pts / ln_gidx / ln_rel / ln_label): segment s of a line covers its line indices
    // [seg_cum[s], seg_cum[s + 1]) at storage positions seg_pos[s] ...  (three-pass bucketing: one segment, = line_start)
    int* seg_cum = nullptr;     // B * L * (MML_SEG_MAX + 1)
    int* seg_pos = nullptr;     // B * L * MML_SEG_MAX
    int* seg_n = nullptr;       // B * L
    int* seg_flat = nullptr;    // B * 2 * MML_SEG_FLAT: the segment starts of a sensor's region in storage order (block-major)
    int* seg_flat_n = nullptr;  // B * 2 * 2: entries | lines per block
    // per 64 line indices the one segment boundary they may cross and the storage offsets on either side (k_seg_records):
    // seg_rs for the rounds of k_stencil's tiles (indices 64 r - 5 ...), seg_rw for aligned windows (64 w ...); record
    // (line, r) sits at  slot * seg_rstride + (line_start[line] >> 6) + 6 * line + r
    int4* seg_rs = nullptr;
    int4* seg_rw = nullptr;
    int seg_rstride = 0;
    unsigned long long* op_agg = nullptr;  // B * 2 * MML_SEG_MAX: block aggregates of the one-pass bucketing (epoch-tagged)
    unsigned op_epoch = 0;
    bool onepass = false;       // the ring layout takes the one-pass bucketing
    float* ln_curv = nullptr;
    float* ln_refl = nullptr;
    uint16_t* ln_attr = nullptr;
    int* crop_cnt = nullptr;     // crop pass block counts (8 ints per block)
    int* blk_cnt = nullptr;      // assign pass histograms
    int* assign_aux = nullptr;   // 8 ints per slot
    unsigned* sel_scratch = nullptr;  // 4 x B*NT unsigned: k_select scratch for lines beyond the LDS budget
    int sel_cap = 0;
    int sel_cap_velo = 0;
    unsigned* brk_queue = nullptr;  // B * NT: queued break-point candidates of k_stencil
    int* brk_cnt = nullptr;         // 2 B: break-point queue sizes, then redo queue sizes
    int* queue_off = nullptr;       // 2 (B + MAX_LANES + 1): per launch the offsets of the slots' redo / break-point queues in their concatenation
    unsigned* redo_queue = nullptr; // B * NT: points k_stencil left to k_stencil_redo
    uint8_t* sel_done = nullptr;       // B * L: lines finished by k_select_part
    bool select_part = true;
    int* sel_list = nullptr;           // 2 * B * L * 2 ints: (slot, line) lists of the lines left to k_select (rings | Livox lines)
    int* sel_list_cnt = nullptr;       // 2 B ints
    unsigned char* st_exit = nullptr;  // B * (NT / 256 + L + 8): k_stencil segment mode, exit offsets of the stride walk per tile
    int* vx_big = nullptr;             // 2 B: per launch (indexed by its first slot) the slots k_voxel<512> left to the large form: count | list

    // combined (pre-crop) cloud, velo part at [0, NV), livox part at [NV, NT)
    int* cb_n = nullptr;  // B * 2

    // The fused cropped cloud [velo_combine ; livox_combine] has no storage of its own: a kept point lives at its
    // line-bucketed position (ln_pts, undistorted in place) and ln_gidx says where it sits in the fused order; that order
    // is materialised only at the API boundary (downloads) and as the tie-break of the voxel sort.
    uint8_t* ln_line = nullptr;   // B * NT  ring / Livox line (normal_y) of an UPLOADED cloud (an extracted one has its line table)
    int* slot_flags = nullptr;    // B * 2   [0] bit 0: filled by mml_cloud_upload, bit 1: undistorted (in-sweep time reads 1);
                                  //         [1] bit 1 as it was when the current mml_undistort started
    uint8_t* ln_label = nullptr;  // B * NT  0 none / 1 corner / 2 surf (normal_z) for kept points; 0x81 / 0x82 for labelled Livox
                                  //         points beyond far_th (counted in livox_*_num and aligned by the GICP refresh, not fused)
    int* fu_info = nullptr;  // B * 8: n_points, n_velo, vc, vs, lc, ls, fused corner, fused surf

    // down-sampled feature stacks: kind 0 corner, 1 surf
    float4* ft_xyz[2] = {nullptr, nullptr};  // B * MF
    int* ft_n = nullptr;                     // 2 * B
    unsigned long long* vx_keys = nullptr;   // voxel sort scratch: B * 2 * VX_CAP
    int* vx_gidx = nullptr;                  // B * 2 * VX_CAP: the fused index of every listed point (label_append), parallel to the lists in vx_keys
    // batched global-sort down-sampler (map_upkeep.hip mml_downsample_big), allocated on first use
    unsigned long long* seg_keys = nullptr;
    unsigned* seg_vals = nullptr;
    float4* seg_cat = nullptr;
    int* seg_flag = nullptr;
    int* seg_meta = nullptr;
    void* seg_tmp[8] = {};
    size_t seg_tmp_bytes[8] = {};
    int VX_CAP = 0;                          // label-list stride per (slot, kind) = NT: every labelled point is listed

    // factors
    MmlLineFactor* lf = nullptr;   // B * MF
    MmlPlaneFactor* pf = nullptr;  // B * MF
    int* work_off = nullptr;       // 2 * B + 1 chunk offsets for k_associate
    float* hard_knn = nullptr;     // 10 floats per queued feature
    int4* hard_list = nullptr;     // B * MF * 2 queued (slot, kind, feature) triples
    double* assoc_stats = nullptr; // B * 16: n_line, n_plane, n_line_used, n_plane_used, gram[9], ...

    // maps
    MmlGrid grid[2];
    bool have_map[2] = {false, false};
    // global cube map (a12): tagged grids over the concatenated cube clouds
    MmlGrid ggrid[2];
    bool have_gmap[2] = {false, false};
    float4* gmap_orig[2] = {nullptr, nullptr};
    uint16_t* gtag_orig[2] = {nullptr, nullptr};
    int* cube_cnt[2] = {nullptr, nullptr};  // 4851 ints each
    int gmap_cap[2] = {0, 0};
    int cen[3] = {10, 5, 10};  // laserCloudCen{Width,Height,Depth}_last (Map_Manager.h:113-115)
    // device-side local map upkeep (Estimator::MapIncrementLocal): ring of key scans' features in the world frame
    static constexpr int LOCAL_WINDOW = 50;  // localMapWindowSize, Estimator.h:326
    float4* ring[2] = {nullptr, nullptr};    // LOCAL_WINDOW x MF each
    float4* ring_cat = nullptr;              // concatenation scratch, LOCAL_WINDOW x MF
    int* vox_flag = nullptr;                 // head flags / positions, 2 x (vox_cap + 1)
    size_t vox_cap = 0;                      // points ring_cat / vox_flag are sized for
    int ring_n[2][LOCAL_WINDOW] = {};
    long local_map_id = 0;                   // localMapID
    // device-side MAP_MANAGER cube stores (map_global.hip): live points + cube tags, pending world-frame features
    float4* gs_pts[2] = {nullptr, nullptr};
    uint16_t* gs_tag[2] = {nullptr, nullptr};
    float4* gs_pts2[2] = {nullptr, nullptr};   // ping-pong for compaction
    uint16_t* gs_tag2[2] = {nullptr, nullptr};
    int gs_n[2] = {0, 0};
    float4* gp_pts[2] = {nullptr, nullptr};    // laserCloud{Corner,Surf}_to_map
    int gp_n[2] = {0, 0};
    int gp_cap = 0;
    int gs_cen[3] = {10, 5, 10};               // laserCloudCen{Width,Height,Depth} of the live store
    int* gs_work = nullptr;                    // histograms / flags / bbox keys / scan scratch
    unsigned long long* gs_keys = nullptr;     // 64-bit sort keys, 2 x MM
    void* wire_stage = nullptr;              // raw message bytes on their way in / out (one slot at a time)
    size_t wire_stage_bytes = 0;
    int local_map_n[2] = {0, 0};
    float4* map_tmp = nullptr;
    unsigned* map_keys = nullptr;
    unsigned* map_keys2 = nullptr;
    unsigned* map_vals = nullptr;
    unsigned* map_vals2 = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int MM = 0;

    // solver state
    double* d_x = nullptr;        // B * 6
    double* d_pose_in = nullptr;  // B * 32 generic double params (T_wl, dR/dt ...)
    double* d_result = nullptr;      // B * MML_SOLVE_RESULT: k_solve's result records (pose, stack sizes, association statistics) for small calls
    double* d_summ = nullptr;     // B * 8
    double* d_trace = nullptr;    // B * 6 * MAX_ITERS
    double* d_und = nullptr;      // B * 8 per-scan undistortion constants
    double* d_rec = nullptr;      // B * 32
    float* d_extr = nullptr;      // 16 floats
    int* d_misc = nullptr;        // misc ints

    // pinned host staging
    double* h_stage = nullptr;
    size_t h_stage_doubles = 0;
    size_t stage_cursor = 0;

    // profiling
    bool profiling = false;
    std::vector<MmlStageTimer> stages;
    struct Pending {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
};

// pcl::VoxelGrid::applyFilter (PCL 1.8.1 voxel_grid.hpp): a bounding box of more than INT_MAX voxels ("Leaf size is too small for the
// input dataset. Integer indices would overflow.") returns the cloud UNFILTERED.  The device filters then key every point by its
// own ordinal, which is the same thing: every voxel holds one point, in input order.
__host__ __device__ inline bool mml_voxel_grid_overflows(const float* mn, const float* mx, float inv) {
    const long long dx = static_cast<long long>((mx[0] - mn[0]) * inv) + 1, dy = static_cast<long long>((mx[1] - mn[1]) * inv) + 1,
                    dz = static_cast<long long>((mx[2] - mn[2]) * inv) + 1;
    return dx * dy * dz > 2147483647LL;
}
#define MML_STREAM(ctx) ((ctx)->streams[(ctx)->cur])
int mml_sync_all(mml_ctx* ctx);
double* mml_stage_alloc(mml_ctx* ctx, size_t doubles);  // pinned staging ring (capi.hip): small read-backs land here
int mml_uploads_wait(mml_ctx* ctx, int first, int count);

#define MML_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return MML_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define MML_REQUIRE(cond, code, msg) \
    do {                             \
        if (!(cond)) {               \
            ctx->err = (msg);        \
            return (code);           \
        }                            \
    } while (0)

// profiling bracket (no-op unless enabled)
int mml_stage_begin(mml_ctx* ctx, const char* name);
void mml_stage_end(mml_ctx* ctx, int token);
struct MmlStageScope {
    mml_ctx* c;
    int t;
    MmlStageScope(mml_ctx* ctx, const char* name) : c(ctx), t(mml_stage_begin(ctx, name)) {}
    ~MmlStageScope() { mml_stage_end(c, t); }
};

// launchers implemented in the .hip files (all asynchronous on ctx->stream)
int mml_launch_extract(mml_ctx* ctx, int first, int count, bool have_extrinsic);
int mml_launch_cloud_decode(mml_ctx* ctx, int slot, const float* d_raw, int n, int n_velo);
int mml_launch_undistort(mml_ctx* ctx, int first, int count, const double* d_params);
int mml_launch_time_offset(mml_ctx* ctx, MmlGrid& g, float4* d_velo4, const float* d_velo_xyz, int n_velo, const float* d_tf,
                           const float* d_livox_xyz, int n_livox, int res, int sliced, int nwin, float* d_nn, double* d_err);
int mml_launch_downsample(mml_ctx* ctx, int first, int count);
// labelled corner points per (slot, kind) that the LDS sort of k_voxel takes (surf: MML_VOXEL_LDS_CAP)
inline int mml_voxel_cap_corner(const mml_ctx* ctx) { return (ctx->NT > 65536 || MML_VOXEL_LDS_CAP < 2048) ? MML_VOXEL_LDS_CAP : 2048; }
int mml_build_grid(mml_ctx* ctx, int kind, const float* h_xyz, int m);
int mml_build_grid_device(mml_ctx* ctx, int kind, int m);
int mml_downsample_big(mml_ctx* ctx, int first, int count);
int mml_downsample_redo_overflow(mml_ctx* ctx, int first, int count, std::vector<int>* redone);
int mml_map_upkeep_increment(mml_ctx* ctx, int slot, const double* T_wl, int* n_out);
int mml_build_global_grid(mml_ctx* ctx, int kind, const float* h_xyz, const int* h_cube, int m, const int* cen);
int mml_build_global_grid_device(mml_ctx* ctx, int kind, const float4* d_pts, const uint16_t* d_tags, const int* d_cnt, int m,
                                 const int* cen);
int mml_cube_store_append(mml_ctx* ctx, int slot, const double* T_wl);
int mml_cube_store_increment(mml_ctx* ctx, const double* T_wl, int* n_out);
int mml_cube_store_download(mml_ctx* ctx, int kind, float* xyz, int* cube, int capacity, int* n, int* cen);
int mml_cube_store_reset(mml_ctx* ctx);
int mml_launch_knn5(mml_ctx* ctx, int kind, const float* d_q, int nq, float max_d2, int* d_idx, float* d_d2);
// with_stats = false (mml_step, which reads none of them): the per-slot statistics (counts, normal Gram matrix) are left stale and
// computed by mml_ensure_assoc_stats when a consumer asks (mml_linearize*, the next mml_associate with statistics)
int mml_launch_associate(mml_ctx* ctx, int first, int count, const double* d_Twl, double thres_dist, bool with_stats = true);
int mml_ensure_assoc_stats(mml_ctx* ctx, int first, int count);
#define MML_SOLVE_RESULT 24  // doubles per problem of k_solve's result record: x (6), stack sizes (2), association statistics (16)
int mml_launch_solve(mml_ctx* ctx, int first, int count, int window, const double* d_Tbl, mml_solve_opts opts,
                     bool want_trace, const double* d_x_in = nullptr, double* d_result = nullptr);
int mml_window_solve_continue(mml_ctx* ctx, int first, int count, int window, const double* d_Tbl, mml_solve_opts opts);
int mml_feature_init(mml_ctx* ctx);
void mml_fullwindow_dev_release(mml_ctx* ctx);
int mml_launch_detect_line(mml_ctx* ctx, int n, uint16_t* d_final);
int mml_launch_raw_lines(mml_ctx* ctx, int slot);  // raw_line[] of one slot (ring / line id per raw point), for the GICP refresh
int mml_launch_linearize(mml_ctx* ctx, int slot, const double* d_x, const double* d_Tbl, double w_tan,
                         double huber, double* d_record, int frames = 1);

#endif
