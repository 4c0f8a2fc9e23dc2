// Measured instruction-issue rates of gfx950 (MI355X): wave64 instructions per second for the instruction classes the kernels of
// this repository are made of.  Every kernel runs `CHAINS` independent dependency chains per lane (so that latency is hidden inside
// one wavefront) on enough wavefronts to fill every SIMD several times over, and is timed with HIP events at 1 / 2 / 4 / 8 wavefronts per SIMD (set through the LDS a workgroup asks for).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -ffp-contract=off tools/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
// Output (kept as profiles/r06_issue_probe.txt): per class the wave-instructions per second of the whole device, the same per
// SIMD per clock at the clock the run sustained (measured with s_memrealtime against wall_clock64), and the implied cycles per
// wave64 instruction.  bench.py reads the v_fma_f32 line as its VALU issue peak.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

constexpr int ITER = 4096;
constexpr int CHAINS = 8;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k_probe(float* out, float a, float b, int ia) {
    extern __shared__ float lds[];  // >= 256 * 9 floats; the launch sizes it to set the number of workgroups a CU holds
    const int t = threadIdx.x;
    float x[CHAINS];
    double d[CHAINS];
    int u[CHAINS];
    f2 p[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
        x[i] = t + i;
        d[i] = t + i;
        u[i] = t * 7 + i;
        p[i] = f2{(float)t + i, (float)i};
        lds[t * 9 + i] = t + i;
    }
    __syncthreads();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if constexpr (KIND == 0) x[i] = __builtin_fmaf(x[i], a, b);                      // v_fma_f32
            if constexpr (KIND == 1) x[i] = x[i] * a;                                        // v_mul_f32
            if constexpr (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ia));   // v_add_u32 (the C form folds into one multiply)
            if constexpr (KIND == 3) u[i] = (u[i] > ia) ? u[i] - 3 : ia;                     // v_cmp + v_cndmask (+ v_sub): 3 instructions
            if constexpr (KIND == 4) d[i] = __builtin_fma(d[i], (double)a, (double)b);       // v_fma_f64
            if constexpr (KIND == 5) p[i] = __builtin_elementwise_fma(p[i], f2{a, a}, f2{b, b});  // v_pk_fma_f32
            if constexpr (KIND == 6) u[i] = __builtin_amdgcn_ubfe(u[i], 3, 9) + u[i];        // v_bfe_u32 + v_add
            if constexpr (KIND == 7) x[i] = __builtin_amdgcn_rcpf(x[i]);                     // v_rcp_f32 (transcendental unit)
            if constexpr (KIND == 8) x[i] = lds[((t + (int)x[i]) & 255) * 9 + i];            // ds_read_b32 (+ address arithmetic)
            if constexpr (KIND == 9) u[i] = __builtin_amdgcn_mov_dpp(u[i], 0x111, 0xf, 0xf, false) + ia;  // v_mov_dpp row_shr:1 + v_add
            if constexpr (KIND == 10) u[i] = __popcll(__ballot(u[i] > ia)) + u[i];            // v_cmp -> s_bcnt1 -> v_add (SALU round trip)
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i] + (float)d[i] + (float)u[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + t] = s;
}

__global__ void k_clock(unsigned long long* out, int spin) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    const unsigned long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[0] = w1 - w0;
        out[1] = c1 - c0;
        out[2] = (unsigned long long)x;
    }
}

struct Kind {
    const char* name;
    double per_iter;  // wave-instructions of the class per chain step
    void (*fn)(float*, float, float, int);
};

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clockRate %d kHz, wall clock %d kHz\n", prop.name, cus, prop.clockRate, wall_khz);
    float* d;
    const int max_blocks = cus * 64;  // (the largest launch below: 8 workgroups per CU x 8 rounds)
    hipMalloc(&d, sizeof(float) * max_blocks * 256);
    // the shader clock the device sustains under a VALU load: clock64() ticks (s_memtime, shader clock) against wall_clock64
    unsigned long long* dc;
    hipMalloc(&dc, 64);
    double shader_hz = 0;
    {
        hipLaunchKernelGGL(k_probe<0>, dim3(cus * 16), dim3(256), 18 * 1024, 0, d, 0.999f, 0.001f, 3);  // warm the clocks
        hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dc, 4000000);
        hipDeviceSynchronize();
        unsigned long long h[3];
        hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost);
        shader_hz = (double)h[1] / ((double)h[0] / (wall_khz * 1e3));
        printf("clock64 ran at %.3f GHz against the wall clock (s_memtime: the constant 100 MHz counter on gfx9 if this prints 0.1)\n", shader_hz * 1e-9);
    }
    const Kind kinds[] = {
        {"v_fma_f32", 1, k_probe<0>},       {"v_mul_f32", 1, k_probe<1>},         {"v_add_u32", 1, k_probe<2>},
        {"v_cmp+v_cndmask+v_sub (3)", 3, k_probe<3>}, {"v_fma_f64", 1, k_probe<4>}, {"v_pk_fma_f32", 1, k_probe<5>},
        {"v_bfe_u32+v_add (2)", 2, k_probe<6>}, {"v_rcp_f32", 1, k_probe<7>},     {"ds_read_b32 chain (+4 VALU)", 1, k_probe<8>},
        {"v_mov_dpp+v_add (2)", 2, k_probe<9>}, {"ballot+s_bcnt1+v_add", 1, k_probe<10>},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-30s %8s %14s %14s %12s\n", "class", "waves/SIMD", "ms", "wave-instr/s", "cycles/instr @2.4GHz");
    for (const Kind& k : kinds) {
        for (int occ : {1, 2, 4, 8}) {  // workgroups of 4 wavefronts (one per SIMD); `occ` of them fit a CU's 160 KB of LDS
            const size_t lds_bytes = occ == 1 ? 96 * 1024 : occ == 2 ? 60 * 1024 : occ == 4 ? 36 * 1024 : 18 * 1024;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            const int blocks = cus * occ * 8;  // eight rounds of resident workgroups
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), lds_bytes, 0, d, 0.999f, 0.001f, 3);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double instr = (double)blocks * 4 * ITER * CHAINS * k.per_iter;
            const double rate = instr / (best * 1e-3);
            // cycles a SIMD spends per wave-instruction if the device ran at 2.4 GHz: 4 SIMDs per CU
            const double cyc = (cus * 4.0 * 2.4e9) / rate;
            if (hipGetLastError() != hipSuccess) {
                printf("%-30s %8d launch failed\n", k.name, occ);
                continue;
            }
            printf("%-30s %8d %14.3f %14.4g %12.2f\n", k.name, occ, best, rate, cyc);
        }
    }
    return 0;
}
