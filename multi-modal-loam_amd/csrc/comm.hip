// comm.hip -- SURVEY.md section 8(e): the multi-GPU side of the C-ABI.  One process per GPU, one mml_ctx per process,
// one RCCL communicator per ctx; collectives are enqueued on the ctx stream (xGMI underneath on an 8 x MI355X node).
//   mml_window_solve_allgather   the joint window solve of Estimator::Estimate (Estimator.cpp:1265-1299 evaluate the frames
//                                one after the other, :1425-1432 solve): every rank evaluates its own frames, the 32-double
//                                normal-equation records travel by ncclAllGather, every rank advances the same
//                                device-resident dogleg state machine (k_window_round, solve.hip).  No host round trip
//                                between the evaluations.
//   mml_comm_broadcast_features  the map-update exchange: the key scan's down-sampled stacks go from the rank that owns
//                                the scan to every replica before MapIncrementLocal (Estimator.cpp:1083-1085,1125-1130)
//   mml_comm_broadcast_local_map the replicated local map, once (e.g. the 10 M-point map of BASELINE configs[4])
#include <rccl/rccl.h>
#include <string.h>

#include "mml_internal.h"

size_t mml_window_state_bytes();
int mml_launch_window_round(mml_ctx* ctx, int first, int n_local, int rank, int W, const double* d_Tbl, mml_solve_opts opts,
                            int round, bool do_eval, const double* d_x_all, double* d_rec_all, void* d_state, double* d_aux);
int mml_window_state_read(mml_ctx* ctx, const void* d_state, int W, double* x_window, mml_solve_summary* summ, double initial_cost);

struct MmlComm {
    ncclComm_t comm = nullptr;   // null for a loopback group (mml_comm_init_loopback)
    bool loopback = false;
    int n_ranks = 0, rank = 0;
    double* d_x_all = nullptr;    // 8 x 6
    double* d_rec_all = nullptr;  // 8 x 32
    void* d_state = nullptr;
    double* d_aux = nullptr;      // 8 doubles
    double* d_Tbl = nullptr;      // 16
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define MML_NCCL(call)                                                                \
    do {                                                                              \
        ncclResult_t r_ = (call);                                                     \
        if (r_ != ncclSuccess) {                                                      \
            ctx->err = std::string(#call) + ": " + ncclGetErrorString(r_);            \
            return MML_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

static_assert(MML_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as an opaque 128-byte block");

extern "C" int mml_comm_destroy(mml_ctx* ctx);

static int comm_buffers(mml_ctx* ctx, MmlComm* c) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_x_all), sizeof(double) * 6 * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_rec_all), sizeof(double) * 32 * 8);
    if (e == hipSuccess) e = hipMalloc(&c->d_state, mml_window_state_bytes());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_aux), sizeof(double) * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->d_Tbl), sizeof(double) * 16);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        ctx->err = std::string("mml_comm_init: ") + hipGetErrorString(e);
        mml_comm_destroy(ctx);
        return MML_ERR_HIP;
    }
    return MML_OK;
}

// The three rank-local parts of the joint window solve; the exchange between them is the transport's (RCCL or loopback).
// begin: own poses into this rank's section of the window vector, T_bl, a clean state
static int win_begin(mml_ctx* ctx, int first_slot, int n_local, const double* T_bl, const mml_solve_opts* opts, const double* x_local) {
    MmlComm* c = ctx->comm;
    MML_REQUIRE(first_slot >= 0 && n_local >= 1 && first_slot + n_local <= ctx->B, MML_ERR_INVALID, "slot range out of bounds");
    MML_REQUIRE(c->n_ranks * n_local <= 8, MML_ERR_INVALID, "window = n_ranks * n_local must not exceed 8 frames");
    MML_REQUIRE(T_bl && opts && x_local, MML_ERR_INVALID, "null argument");
    MML_REQUIRE(opts->max_num_iterations >= 0 && opts->max_num_iterations <= 64, MML_ERR_INVALID, "max_num_iterations must be in [0, 64]");
    MML_HIP(hipSetDevice(ctx->device));
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    double* h = mml_stage_alloc(ctx, 6 * 8 + 16);  // (pinned: the copies below are asynchronous)
    memcpy(h, x_local, sizeof(double) * 6 * n_local);
    memcpy(h + 6 * n_local, T_bl, sizeof(double) * 16);
    MML_HIP(hipMemcpyAsync(c->d_x_all + 6 * (size_t)c->rank * n_local, h, sizeof(double) * 6 * n_local, hipMemcpyHostToDevice, s));
    MML_HIP(hipMemcpyAsync(c->d_Tbl, h + 6 * n_local, sizeof(double) * 16, hipMemcpyHostToDevice, s));
    MML_HIP(hipMemsetAsync(c->d_state, 0, mml_window_state_bytes(), s));
    MML_HIP(hipMemsetAsync(c->d_aux, 0, sizeof(double) * 8, s));
    MML_HIP(hipMemsetAsync(c->d_rec_all, 0, sizeof(double) * 32 * 8, s));
    MML_HIP(hipEventRecord(c->ev0, s));
    return MML_OK;
}
static int win_round(mml_ctx* ctx, int first_slot, int n_local, const mml_solve_opts* opts, int r, bool do_eval) {
    MmlComm* c = ctx->comm;
    MML_HIP(hipSetDevice(ctx->device));
    ctx->cur = 0;
    return mml_launch_window_round(ctx, first_slot, n_local, c->rank, c->n_ranks * n_local, c->d_Tbl, *opts, r, do_eval, c->d_x_all,
                                   c->d_rec_all, c->d_state, c->d_aux);
}
static int win_end(mml_ctx* ctx, int n_local, int rounds, double* x_local, double* x_window, mml_solve_summary* summary,
                   mml_window_timing* timing) {
    MmlComm* c = ctx->comm;
    const int W = c->n_ranks * n_local;
    MML_HIP(hipSetDevice(ctx->device));
    hipStream_t s = MML_STREAM(ctx);
    MML_HIP(hipEventRecord(c->ev1, s));
    double aux[8];
    MML_HIP(hipMemcpyAsync(aux, c->d_aux, sizeof(aux), hipMemcpyDeviceToHost, s));
    std::vector<double> xw(6 * (size_t)W);
    int rc = mml_window_state_read(ctx, c->d_state, W, xw.data(), summary, 0.0);  // synchronises the stream
    if (rc != MML_OK) return rc;
    if (summary) summary->initial_cost = aux[0];
    memcpy(x_local, xw.data() + 6 * (size_t)c->rank * n_local, sizeof(double) * 6 * n_local);
    if (x_window) memcpy(x_window, xw.data(), sizeof(double) * 6 * W);
    if (timing) {
        float ms = 0;
        MML_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        timing->evaluations = (int)aux[1];
        timing->rounds = rounds;
        timing->exchanges = rounds;  // the pose all-gather + one record all-gather per evaluating round
        timing->device_ms = ms;
    }
    return MML_OK;
}
// the receiving side of the local-map broadcast: every replica builds its own grids (the sort is deterministic: bit-identical)
static int local_map_received(mml_ctx* ctx, const int* m, bool is_root) {
    if (!is_root)
        for (int kind = 0; kind < 2; ++kind) {
            int rc = mml_build_grid_device(ctx, kind, m[kind]);
            if (rc != MML_OK) return rc;
        }
    return mml_sync_all(ctx);
}

extern "C" {

int mml_comm_unique_id(uint8_t* id) {
    if (!id) return MML_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return MML_ERR_HIP;
    memcpy(id, u.internal, MML_COMM_ID_BYTES);
    return MML_OK;
}

int mml_rccl_version(int* version) {
    if (!version) return MML_ERR_INVALID;
    return ncclGetVersion(version) == ncclSuccess ? MML_OK : MML_ERR_HIP;
}

int mml_comm_destroy(mml_ctx* ctx) {
    if (!ctx) return MML_ERR_INVALID;
    MmlComm* c = ctx->comm;
    if (!c) return MML_OK;
    hipSetDevice(ctx->device);
    mml_sync_all(ctx);
    if (c->comm) ncclCommDestroy(c->comm);
    void* ptrs[] = {c->d_x_all, c->d_rec_all, c->d_state, c->d_aux, c->d_Tbl};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    delete c;
    ctx->comm = nullptr;
    return MML_OK;
}

int mml_comm_init(mml_ctx* ctx, int n_ranks, int rank, const uint8_t* id) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(id && n_ranks >= 1 && rank >= 0 && rank < n_ranks, MML_ERR_INVALID, "bad communicator arguments");
    MML_REQUIRE(ctx->comm == nullptr, MML_ERR_STATE, "the context already has a communicator");
    MML_HIP(hipSetDevice(ctx->device));
    MmlComm* c = new MmlComm();
    ctx->comm = c;
    c->n_ranks = n_ranks;
    c->rank = rank;
    ncclUniqueId u;
    memcpy(u.internal, id, MML_COMM_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        c->comm = nullptr;
        mml_comm_destroy(ctx);
        return MML_ERR_HIP;
    }
    return comm_buffers(ctx, c);
}

int mml_comm_info(mml_ctx* ctx, int* n_ranks, int* rank) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(ctx->comm != nullptr, MML_ERR_STATE, "no communicator (mml_comm_init)");
    if (n_ranks) *n_ranks = ctx->comm->n_ranks;
    if (rank) *rank = ctx->comm->rank;
    return MML_OK;
}

int mml_window_solve_allgather(mml_ctx* ctx, int first_slot, int n_local, const double* T_bl, const mml_solve_opts* opts,
                               double* x_local, double* x_window, mml_solve_summary* summary, mml_window_timing* timing) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(ctx->comm != nullptr, MML_ERR_STATE, "no communicator (mml_comm_init)");
    MmlComm* c = ctx->comm;
    MML_REQUIRE(!c->loopback, MML_ERR_STATE, "a loopback group is driven through mml_window_solve_allgather_loopback");
    int rc = win_begin(ctx, first_slot, n_local, T_bl, opts, x_local);
    if (rc != MML_OK) return rc;
    hipStream_t s = MML_STREAM(ctx);
    // every rank needs the whole window's starting point (the dogleg works on the joint parameter vector)
    MML_NCCL(ncclAllGather(c->d_x_all + 6 * (size_t)c->rank * n_local, c->d_x_all, 6 * (size_t)n_local, ncclDouble, c->comm, s));
    // round 0 evaluates at x0; rounds 1 .. R-1 advance with the gathered records and evaluate the next candidate; the
    // last round only advances.  Every trust-region iteration costs one evaluation, so max_iterations + 2 rounds is
    // the longest the state machine can run; ranks hold identical states and therefore stop together.
    const int rounds = opts->max_num_iterations + 2;
    for (int r = 0; r < rounds; ++r) {
        const bool do_eval = r + 1 < rounds;
        rc = win_round(ctx, first_slot, n_local, opts, r, do_eval);
        if (rc != MML_OK) return rc;
        if (do_eval)
            MML_NCCL(ncclAllGather(c->d_rec_all + 32 * (size_t)c->rank * n_local, c->d_rec_all, 32 * (size_t)n_local, ncclDouble, c->comm, s));
    }
    return win_end(ctx, n_local, rounds, x_local, x_window, summary, timing);
}

// ---- loopback group: N ranks = N contexts of ONE process on ONE device, every collective executed by the calling thread as
// device-to-device copies between the ranks' buffers.  RCCL refuses two ranks on one device, so this is how the rank > 0 side
// of the path -- the section offsets of the gather buffers, "linearise only my frames", a broadcast whose root is another
// rank -- runs on a single-GPU box: the kernels, the buffers and the state machine are the ones mml_window_solve_allgather
// drives, only the transport differs.  Test / bring-up entry points, not a deployment mode.
static int comm_init_loopback_ranks(mml_ctx** ctxs, int n_ranks) {
    for (int r = 0; r < n_ranks; ++r) {
        mml_ctx* ctx = ctxs[r];
        MML_REQUIRE(ctx->comm == nullptr, MML_ERR_STATE, "the context already has a communicator");
        MML_REQUIRE(ctx->device == ctxs[0]->device, MML_ERR_INVALID, "a loopback group lives on one device");
        MML_HIP(hipSetDevice(ctx->device));
        MmlComm* c = new MmlComm();
        ctx->comm = c;
        c->loopback = true;
        c->n_ranks = n_ranks;
        c->rank = r;
        int rc = comm_buffers(ctx, c);
        if (rc != MML_OK) return rc;
    }
    return MML_OK;
}

// all or nothing: a rank that cannot join (it already has a communicator, sits on another device, or runs out of memory) leaves
// the ranks before it without one as well, so that the call can be retried
int mml_comm_init_loopback(mml_ctx** ctxs, int n_ranks) {
    if (!ctxs || n_ranks < 1 || n_ranks > 8) return MML_ERR_INVALID;
    for (int r = 0; r < n_ranks; ++r)
        if (!ctxs[r]) return MML_ERR_INVALID;
    for (int r = 0; r < n_ranks; ++r)
        if (ctxs[r]->comm != nullptr) {
            ctxs[r]->err = "the context already has a communicator";
            return MML_ERR_STATE;
        }
    const int rc = comm_init_loopback_ranks(ctxs, n_ranks);
    if (rc != MML_OK)
        for (int r = 0; r < n_ranks; ++r) mml_comm_destroy(ctxs[r]);  // (no-op for a rank that never got one)
    return rc;
}

// all-gather of `count` doubles per rank at buf(rank) + rank * count, by copies; every stream is drained before and after
static int loopback_allgather(mml_ctx** ctxs, int n_ranks, double* (*buf)(MmlComm*), size_t count) {
    for (int r = 0; r < n_ranks; ++r) {
        int rc = mml_sync_all(ctxs[r]);
        if (rc != MML_OK) return rc;
    }
    for (int src = 0; src < n_ranks; ++src)
        for (int dst = 0; dst < n_ranks; ++dst) {
            if (dst == src) continue;
            mml_ctx* ctx = ctxs[dst];
            MML_HIP(hipMemcpyAsync(buf(ctxs[dst]->comm) + (size_t)src * count, buf(ctxs[src]->comm) + (size_t)src * count,
                                   sizeof(double) * count, hipMemcpyDeviceToDevice, MML_STREAM(ctx)));
        }
    for (int r = 0; r < n_ranks; ++r) {
        int rc = mml_sync_all(ctxs[r]);
        if (rc != MML_OK) return rc;
    }
    return MML_OK;
}

int mml_window_solve_allgather_loopback(mml_ctx** ctxs, int n_ranks, const int* first_slot, int n_local, const double* T_bl,
                                        const mml_solve_opts* opts, const double* x_window_in, double* x_window_out,
                                        mml_solve_summary* summaries) {
    if (!ctxs || n_ranks < 1 || n_ranks > 8 || !first_slot || !x_window_in || !x_window_out) return MML_ERR_INVALID;
    for (int r = 0; r < n_ranks; ++r) {
        mml_ctx* ctx = ctxs[r];
        if (!ctx) return MML_ERR_INVALID;
        MML_REQUIRE(ctx->comm && ctx->comm->loopback && ctx->comm->n_ranks == n_ranks && ctx->comm->rank == r, MML_ERR_STATE,
                    "not the loopback group these contexts were initialised as (mml_comm_init_loopback)");
    }
    {   // (before anything is sized by n_local: a negative count must come back as an error, not as a huge memcpy)
        mml_ctx* ctx = ctxs[0];
        MML_REQUIRE(n_local >= 1 && n_ranks * n_local <= 8, MML_ERR_INVALID, "window = n_ranks * n_local must be 1 .. 8 frames");
        MML_REQUIRE(T_bl && opts, MML_ERR_INVALID, "null argument");
        MML_REQUIRE(opts->max_num_iterations >= 0 && opts->max_num_iterations <= 64, MML_ERR_INVALID, "max_num_iterations must be in [0, 64]");
    }
    const int W = n_ranks * n_local;
    std::vector<double> xl(6 * (size_t)n_local);
    for (int r = 0; r < n_ranks; ++r) {
        memcpy(xl.data(), x_window_in + 6 * (size_t)r * n_local, sizeof(double) * 6 * n_local);
        int rc = win_begin(ctxs[r], first_slot[r], n_local, T_bl, opts, xl.data());
        if (rc != MML_OK) return rc;
    }
    int rc = loopback_allgather(ctxs, n_ranks, [](MmlComm* c) { return c->d_x_all; }, 6 * (size_t)n_local);
    if (rc != MML_OK) return rc;
    const int rounds = opts->max_num_iterations + 2;
    for (int r = 0; r < rounds; ++r) {
        const bool do_eval = r + 1 < rounds;
        for (int k = 0; k < n_ranks; ++k) {
            rc = win_round(ctxs[k], first_slot[k], n_local, opts, r, do_eval);
            if (rc != MML_OK) return rc;
        }
        if (do_eval) {
            rc = loopback_allgather(ctxs, n_ranks, [](MmlComm* c) { return c->d_rec_all; }, 32 * (size_t)n_local);
            if (rc != MML_OK) return rc;
        }
    }
    for (int r = 0; r < n_ranks; ++r) {
        rc = win_end(ctxs[r], n_local, rounds, xl.data(), x_window_out + 6 * (size_t)W * r, summaries ? summaries + r : nullptr, nullptr);
        if (rc != MML_OK) return rc;
    }
    return MML_OK;
}

int mml_comm_broadcast_features(mml_ctx* ctx, int slot, int root) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(ctx->comm != nullptr, MML_ERR_STATE, "no communicator (mml_comm_init)");
    MmlComm* c = ctx->comm;
    MML_REQUIRE(!c->loopback, MML_ERR_STATE, "a loopback group is driven through the *_loopback entry points");
    MML_REQUIRE(slot >= 0 && slot < ctx->B && root >= 0 && root < c->n_ranks, MML_ERR_INVALID, "bad slot / root");
    MML_HIP(hipSetDevice(ctx->device));
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    // counts first (device ints, no host read), then both stacks at their fixed capacity: 2 x max_features x 16 B
    // (256 KB at the defaults) is below the size at which trimming to the live count would pay for a host round trip
    for (int kind = 0; kind < 2; ++kind) {
        int* n = ctx->ft_n + kind * ctx->B + slot;
        MML_NCCL(ncclBroadcast(n, n, 1, ncclInt32, root, c->comm, s));
        float4* p = ctx->ft_xyz[kind] + (size_t)slot * ctx->MF;
        MML_NCCL(ncclBroadcast(p, p, 4 * (size_t)ctx->MF, ncclFloat32, root, c->comm, s));
    }
    return MML_OK;
}

int mml_comm_broadcast_local_map(mml_ctx* ctx, int root) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(ctx->comm != nullptr, MML_ERR_STATE, "no communicator (mml_comm_init)");
    MmlComm* c = ctx->comm;
    MML_REQUIRE(!c->loopback, MML_ERR_STATE, "a loopback group is driven through the *_loopback entry points");
    MML_REQUIRE(root >= 0 && root < c->n_ranks, MML_ERR_INVALID, "bad root");
    MML_REQUIRE(c->rank != root || (ctx->have_map[0] && ctx->have_map[1]), MML_ERR_STATE, "the root rank has no local map to broadcast");
    MML_HIP(hipSetDevice(ctx->device));
    int rc = mml_sync_all(ctx);
    if (rc != MML_OK) return rc;
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    // sizes through the device (RCCL moves device memory), then the two source clouds, then every replica builds its
    // own grid: the sort is deterministic, so the replicas are bit-identical
    int m[2] = {ctx->grid[0].m, ctx->grid[1].m};
    MML_HIP(hipMemcpyAsync(ctx->d_misc, m, sizeof(m), hipMemcpyHostToDevice, s));
    MML_NCCL(ncclBroadcast(ctx->d_misc, ctx->d_misc, 2, ncclInt32, root, c->comm, s));
    MML_HIP(hipMemcpyAsync(m, ctx->d_misc, sizeof(m), hipMemcpyDeviceToHost, s));
    MML_HIP(hipStreamSynchronize(s));
    MML_REQUIRE(m[0] >= 0 && m[1] >= 0 && m[0] <= ctx->MM && m[1] <= ctx->MM, MML_ERR_CAPACITY, "broadcast map exceeds max_map_points");
    for (int kind = 0; kind < 2; ++kind) {
        float4* p = ctx->map_tmp + (size_t)kind * ctx->MM;
        if (m[kind] > 0) MML_NCCL(ncclBroadcast(p, p, 4 * (size_t)m[kind], ncclFloat32, root, c->comm, s));
    }
    return local_map_received(ctx, m, c->rank == root);
}

int mml_comm_broadcast_features_loopback(mml_ctx** ctxs, int n_ranks, int slot, int root) {
    if (!ctxs || n_ranks < 1 || n_ranks > 8 || root < 0 || root >= n_ranks) return MML_ERR_INVALID;
    for (int r = 0; r < n_ranks; ++r) {
        mml_ctx* ctx = ctxs[r];
        if (!ctx) return MML_ERR_INVALID;
        MML_REQUIRE(ctx->comm && ctx->comm->loopback && ctx->comm->n_ranks == n_ranks && ctx->comm->rank == r, MML_ERR_STATE,
                    "not the loopback group these contexts were initialised as (mml_comm_init_loopback)");
        MML_REQUIRE(slot >= 0 && slot < ctx->B && ctx->MF == ctxs[root]->MF, MML_ERR_INVALID, "bad slot / unequal max_features");
        int rc = mml_sync_all(ctx);
        if (rc != MML_OK) return rc;
    }
    mml_ctx* src = ctxs[root];
    for (int r = 0; r < n_ranks; ++r) {
        if (r == root) continue;
        mml_ctx* ctx = ctxs[r];
        for (int kind = 0; kind < 2; ++kind) {  // what the two ncclBroadcast calls of mml_comm_broadcast_features move
            MML_HIP(hipMemcpyAsync(ctx->ft_n + kind * ctx->B + slot, src->ft_n + kind * src->B + slot, sizeof(int), hipMemcpyDeviceToDevice,
                                   MML_STREAM(ctx)));
            MML_HIP(hipMemcpyAsync(ctx->ft_xyz[kind] + (size_t)slot * ctx->MF, src->ft_xyz[kind] + (size_t)slot * src->MF,
                                   sizeof(float4) * (size_t)ctx->MF, hipMemcpyDeviceToDevice, MML_STREAM(ctx)));
        }
        int rc = mml_sync_all(ctx);
        if (rc != MML_OK) return rc;
    }
    return MML_OK;
}

int mml_comm_broadcast_local_map_loopback(mml_ctx** ctxs, int n_ranks, int root) {
    if (!ctxs || n_ranks < 1 || n_ranks > 8 || root < 0 || root >= n_ranks) return MML_ERR_INVALID;
    for (int r = 0; r < n_ranks; ++r) {
        mml_ctx* ctx = ctxs[r];
        if (!ctx) return MML_ERR_INVALID;
        MML_REQUIRE(ctx->comm && ctx->comm->loopback && ctx->comm->n_ranks == n_ranks && ctx->comm->rank == r, MML_ERR_STATE,
                    "not the loopback group these contexts were initialised as (mml_comm_init_loopback)");
        MML_REQUIRE(r != root || (ctx->have_map[0] && ctx->have_map[1]), MML_ERR_STATE, "the root rank has no local map to broadcast");
        int rc = mml_sync_all(ctx);
        if (rc != MML_OK) return rc;
    }
    mml_ctx* src = ctxs[root];
    int m[2] = {src->grid[0].m, src->grid[1].m};
    for (int r = 0; r < n_ranks; ++r) {
        mml_ctx* ctx = ctxs[r];
        MML_REQUIRE(m[0] >= 0 && m[1] >= 0 && m[0] <= ctx->MM && m[1] <= ctx->MM, MML_ERR_CAPACITY, "broadcast map exceeds max_map_points");
        if (r != root)
            for (int kind = 0; kind < 2; ++kind)
                if (m[kind] > 0)
                    MML_HIP(hipMemcpyAsync(ctx->map_tmp + (size_t)kind * ctx->MM, src->map_tmp + (size_t)kind * src->MM,
                                           sizeof(float4) * (size_t)m[kind], hipMemcpyDeviceToDevice, MML_STREAM(ctx)));
        int rc = local_map_received(ctx, m, r == root);
        if (rc != MML_OK) return rc;
    }
    return MML_OK;
}

}  // extern "C"
