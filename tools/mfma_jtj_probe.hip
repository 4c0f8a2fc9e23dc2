// mfma_jtj_probe.hip -- the measurement behind DESIGN.md's "MFMA is not used" (north_star: "MFMA only for the small dense 6x6 /
// 15x15 accumulate where it actually helps -- choices evidenced by rocprof").
// Workload: the normal-equation accumulate of k_solve (solve.hip / lidar_eval.h): per window problem R Jacobian rows of 6 doubles
// and a residual; wanted are the 21 + 6 + 1 sums of H = J^T J (upper triangle), g = J^T r and r^T r.
//   k_jtj_fma    the product's form: one row per lane per step, 28 fused multiply-adds into 28 private sums, butterfly + LDS reduction
//   k_jtj_mfma   the matrix-core form: J' = [J | r | 0] (R x 8), H' = J'^T J' (8 x 8 holds all 28 sums) by v_mfma_f64_4x4x4f64
//                (4 blocks of 4 x 4 x 4 = the four 4 x 4 tiles of H'), rows staged through LDS so that a lane can pick the
//                operand element the instruction's layout wants from it
// Both are checked against a host sum; times are HIP events over `reps` launches (and the rocprofv3 rows of the same run).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_jtj_probe.hip      Run: ./mfma_probe [problems] [rows]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void k_jtj_fma(const double* __restrict__ rows, int R, double* __restrict__ out) {
    __shared__ double s_red[4][28];
    const double* p = rows + (size_t)blockIdx.x * R * 8;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0.0;
    for (int t = threadIdx.x; t < R; t += 256) {
        const double4 a = *reinterpret_cast<const double4*>(p + 8 * (size_t)t);
        const double4 b = *reinterpret_cast<const double4*>(p + 8 * (size_t)t + 4);
        const double j[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        const double r = b.z;
        int k = 0;
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int v = u; v < 6; ++v) acc[k++] = fma(j[u], j[v], acc[k]);
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[21 + u] = fma(j[u], r, acc[21 + u]);
        acc[27] = fma(r, r, acc[27]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 28; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 28) out[(size_t)blockIdx.x * 28 + threadIdx.x] = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// The operand layout of v_mfma_f64_4x4x4f64 is found by trial: the three 2-bit fields of the lane number are (element row i,
// block b, reduction index k) in one of six orders; `layout` picks the order, main() keeps the one whose result is right.
__device__ __host__ inline void lane_fields(int lane, int layout, int& f_i, int& f_b, int& f_k) {
    const int f[3] = {lane & 3, (lane >> 2) & 3, (lane >> 4) & 3};
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};  // which field is i, b, k
    f_i = f[perm[layout][0]];
    f_b = f[perm[layout][1]];
    f_k = f[perm[layout][2]];
}
__global__ __launch_bounds__(256) void k_jtj_mfma(const double* __restrict__ rows, int R, double* __restrict__ out, int layout) {
    __shared__ double s_tile[4][64][9];  // one 64-row tile per wavefront, padded row stride
    __shared__ double s_acc[4][64];
    const double* p = rows + (size_t)blockIdx.x * R * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int e_i, blk, e_k;
    lane_fields(lane, layout, e_i, blk, e_k);
    const int bi = blk >> 1, bj = blk & 1;                 // tile (bi, bj) of the 8 x 8 result
    double acc = 0.0;
    for (int r0 = wave * 64; r0 < R; r0 += 256) {
        const int row = r0 + lane;
        double4 a = make_double4(0, 0, 0, 0), b = make_double4(0, 0, 0, 0);
        if (row < R) {
            a = *reinterpret_cast<const double4*>(p + 8 * (size_t)row);
            b = *reinterpret_cast<const double4*>(p + 8 * (size_t)row + 4);
        }
        double* d = s_tile[wave][lane];
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = 0.0;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; ++g) {  // four staged rows per instruction
            const double av = s_tile[wave][4 * g + e_k][4 * bi + e_i];
            const double bv = s_tile[wave][4 * g + e_k][4 * bj + e_i];
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    s_acc[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 64) out[(size_t)blockIdx.x * 64 + threadIdx.x] = (s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + (s_acc[2][threadIdx.x] + s_acc[3][threadIdx.x]);
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 1024, R = argc > 2 ? atoi(argv[2]) : 1152, reps = 20;
    std::vector<double> h((size_t)P * R * 8);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((int)(s >> 8) % 2001 - 1000) * 1e-3;
    }
    for (size_t i = 7; i < h.size(); i += 8) h[i] = 0.0;
    double *d_rows, *d_fma, *d_mfma;
    CK(hipMalloc(&d_rows, h.size() * 8));
    CK(hipMalloc(&d_fma, (size_t)P * 28 * 8));
    CK(hipMalloc(&d_mfma, (size_t)P * 64 * 8));
    CK(hipMemcpy(d_rows, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    // host reference for problem 0
    double ref[8][8] = {};
    for (int r = 0; r < R; ++r)
        for (int u = 0; u < 8; ++u)
            for (int v = 0; v < 8; ++v) ref[u][v] += h[8 * (size_t)r + u] * h[8 * (size_t)r + v];
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_fma = 0;
    hipLaunchKernelGGL(k_jtj_fma, dim3(P), dim3(256), 0, 0, d_rows, R, d_fma);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_jtj_fma, dim3(P), dim3(256), 0, 0, d_rows, R, d_fma);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_fma, e0, e1));
    std::vector<double> of(28), om(64);
    CK(hipMemcpy(of.data(), d_fma, 28 * 8, hipMemcpyDeviceToHost));
    double err_fma = 0;
    {
        int k = 0;
        for (int u = 0; u < 6; ++u)
            for (int v = u; v < 6; ++v) err_fma = fmax(err_fma, fabs(of[k++] - ref[u][v]));
        for (int u = 0; u < 6; ++u) err_fma = fmax(err_fma, fabs(of[21 + u] - ref[u][6]));
        err_fma = fmax(err_fma, fabs(of[27] - ref[6][6]));
    }
    double err_mfma[6];
    float ms_mf[6];
    for (int layout = 0; layout < 6; ++layout) {
        err_mfma[layout] = 1e30;
        hipLaunchKernelGGL(k_jtj_mfma, dim3(P), dim3(256), 0, 0, d_rows, R, d_mfma, layout);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_jtj_mfma, dim3(P), dim3(256), 0, 0, d_rows, R, d_mfma, layout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_mf[layout], e0, e1));
        CK(hipMemcpy(om.data(), d_mfma, 64 * 8, hipMemcpyDeviceToHost));
        // result element (i, j) of block b: its lane is again one of the six field orders
        for (int dl = 0; dl < 6; ++dl) {
            double e = 0;
            for (int lane = 0; lane < 64; ++lane) {
                int i, b, j;
                lane_fields(lane, dl, i, b, j);
                e = fmax(e, fabs(om[lane] - ref[4 * (b >> 1) + i][4 * (b & 1) + j]));
            }
            err_mfma[layout] = fmin(err_mfma[layout], e);
        }
    }
    const double bytes = (double)P * R * 64, flop = (double)P * R * 2 * 28;
    printf("problems %d, rows %d (%.1f MB of Jacobian rows per launch)\n", P, R, bytes / 1e6);
    printf("k_jtj_fma : %.3f ms per launch, %.0f GB/s, %.2f useful TFLOP/s, max |err| %.2e\n", ms_fma / reps, bytes / (ms_fma / reps * 1e-3) / 1e9,
           flop / (ms_fma / reps * 1e-3) / 1e12, err_fma);
    for (int layout = 0; layout < 6; ++layout)
        printf("k_jtj_mfma (operand field order %d): %.3f ms per launch, %.0f GB/s, max |err| %.2e%s\n", layout, ms_mf[layout] / reps,
               bytes / (ms_mf[layout] / reps * 1e-3) / 1e9, err_mfma[layout], err_mfma[layout] < 1e-6 ? "  <- the instruction's layout" : "");
    return 0;
}
