// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate of the whole chip (what the power/clock management
// actually delivers), with zero and with random operands.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* clk) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(tid * 8 + i) & 0xffff];
        b[i] = in[(tid * 8 + 4 + i) & 0xffff];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if (tid == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 512, iters = 20000;
    float *in, *out;
    unsigned long long* clk;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipMalloc(&clk, 16);
    float* h = (float*)malloc(65536 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 65536; ++i) h[i] = mode ? (float)rand() / RAND_MAX * 2.f - 1.f : 0.f;
        hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, in, out, iters, clk);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2];
            hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double flop = (double)wgs * 4 * iters * 16 * 4096.0;
            printf("%s wgs=%d  %.3f ms  %.1f TFLOP/s   shader clock %.3f GHz (cycles %llu / 100MHz ticks %llu)\n",
                   mode ? "random" : "zeros ", wgs, ms, flop / ms / 1e9, (double)c[0] / ((double)c[1] * 10.0), c[0], c[1]);
        }
    }
    return 0;
}
