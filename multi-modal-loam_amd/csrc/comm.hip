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
    ncclComm_t comm = nullptr;
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

extern "C" {

int mml_comm_unique_id(uint8_t* id) {
    if (!id) return MML_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return MML_ERR_HIP;
    memcpy(id, u.internal, MML_COMM_ID_BYTES);
    return MML_OK;
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
    MML_REQUIRE(first_slot >= 0 && n_local >= 1 && first_slot + n_local <= ctx->B, MML_ERR_INVALID, "slot range out of bounds");
    const int W = c->n_ranks * n_local;
    MML_REQUIRE(W <= 8, MML_ERR_INVALID, "window = n_ranks * n_local must not exceed 8 frames");
    MML_REQUIRE(T_bl && opts && x_local, MML_ERR_INVALID, "null argument");
    MML_REQUIRE(opts->max_num_iterations >= 0 && opts->max_num_iterations <= 64, MML_ERR_INVALID, "max_num_iterations must be in [0, 64]");
    MML_HIP(hipSetDevice(ctx->device));
    ctx->cur = 0;
    hipStream_t s = MML_STREAM(ctx);
    // own poses into this rank's section of the window vector, T_bl, a clean state
    double h[6 * 8 + 16];
    memcpy(h, x_local, sizeof(double) * 6 * n_local);
    memcpy(h + 6 * n_local, T_bl, sizeof(double) * 16);
    MML_HIP(hipMemcpyAsync(c->d_x_all + 6 * (size_t)c->rank * n_local, h, sizeof(double) * 6 * n_local, hipMemcpyHostToDevice, s));
    MML_HIP(hipMemcpyAsync(c->d_Tbl, h + 6 * n_local, sizeof(double) * 16, hipMemcpyHostToDevice, s));
    MML_HIP(hipMemsetAsync(c->d_state, 0, mml_window_state_bytes(), s));
    MML_HIP(hipMemsetAsync(c->d_aux, 0, sizeof(double) * 8, s));
    MML_HIP(hipMemsetAsync(c->d_rec_all, 0, sizeof(double) * 32 * 8, s));
    MML_HIP(hipEventRecord(c->ev0, s));
    // every rank needs the whole window's starting point (the dogleg works on the joint parameter vector)
    MML_NCCL(ncclAllGather(c->d_x_all + 6 * (size_t)c->rank * n_local, c->d_x_all, 6 * (size_t)n_local, ncclDouble, c->comm, s));
    // round 0 evaluates at x0; rounds 1 .. R-1 advance with the gathered records and evaluate the next candidate; the
    // last round only advances.  Every trust-region iteration costs one evaluation, so max_iterations + 2 rounds is
    // the longest the state machine can run; ranks hold identical states and therefore stop together.
    const int rounds = opts->max_num_iterations + 2;
    for (int r = 0; r < rounds; ++r) {
        const bool do_eval = r + 1 < rounds;
        int rc = mml_launch_window_round(ctx, first_slot, n_local, c->rank, W, c->d_Tbl, *opts, r, do_eval, c->d_x_all, c->d_rec_all,
                                         c->d_state, c->d_aux);
        if (rc != MML_OK) return rc;
        if (do_eval)
            MML_NCCL(ncclAllGather(c->d_rec_all + 32 * (size_t)c->rank * n_local, c->d_rec_all, 32 * (size_t)n_local, ncclDouble, c->comm, s));
    }
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

int mml_comm_broadcast_features(mml_ctx* ctx, int slot, int root) {
    if (!ctx) return MML_ERR_INVALID;
    MML_REQUIRE(ctx->comm != nullptr, MML_ERR_STATE, "no communicator (mml_comm_init)");
    MmlComm* c = ctx->comm;
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
    if (c->rank != root)
        for (int kind = 0; kind < 2; ++kind) {
            rc = mml_build_grid_device(ctx, kind, m[kind]);
            if (rc != MML_OK) return rc;
        }
    return mml_sync_all(ctx);
}

}  // extern "C"
