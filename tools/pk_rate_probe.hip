// Issue rate of packed fp32 arithmetic on gfx950: the same number of INSTRUCTIONS issued as v_fma_f32 and as v_pk_fma_f32
// (eight independent accumulator chains per lane, enough wavefronts to fill every SIMD).  If a packed instruction holds the
// SIMD as long as a scalar one the two kernels take the same time; if it holds it twice as long, packing saves nothing.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/pk_rate_probe.hip -o pk_rate_probe && ./pk_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 8192;
__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 av = f2{a, a}, bv = f2{b, b};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    f2 s = f2{0, 0};
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main() {
    float* d;
    const int blocks = 256 * 8;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {
        float ms[2];
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            if (v == 0)
                hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f);
            else
                hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
        const double instr = (double)blocks * 4 /*waves*/ * ITER * 8;
        printf("pass %d: v_fma_f32 %.3f ms (%.3g wave-instr/s)   v_pk_fma_f32 %.3f ms (%.3g wave-instr/s)   ratio %.2f\n", pass, ms[0],
               instr / (ms[0] * 1e-3), ms[1], instr / (ms[1] * 1e-3), ms[1] / ms[0]);
    }
    return 0;
}
