// Micro-benchmark: does non-matrix work issued between v_mfma_f32_32x32x2_f32 slow the matrix pipe?
// Per MFMA: NV independent VALU adds, NS SALU adds, ND ds_read_b128 (template parameters).  2 waves / SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NS, int ND>
__global__ void __launch_bounds__(256) mix(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(tid * 8 + i) & 0xffff];
        b[i] = in[(tid * 8 + 4 + i) & 0xffff];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned v0 = tid, v1 = tid * 3, v2 = tid * 5, v3 = tid * 7;
    unsigned s0 = iters;
    float4 d = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (k & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
                    else asm volatile("v_add_u32 %0, %0, %1" : "+v"(v2) : "v"(v3));
                }
#pragma unroll
                for (int k = 0; k < NS; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0));
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    f4 t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((tid * 16 + k * 4096 + u * 1024) & 16383)));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (ND) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = d.x;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[threadIdx.x + blockIdx.x * 256] = s + v0 + v2 + s0;
}

template <int NV, int NS, int ND>
void run(const float* in, float* out) {
    const int wgs = 512, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mix<NV, NS, ND>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)wgs * 4 * iters * 16 * 4096.0;
    printf("per MFMA: %d VALU %d SALU %d ds_read_b128 : %.3f ms  %.1f TFLOP/s\n", NV, NS, ND, best, flop / best / 1e9);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, 512 * 256 * 4);
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    run<0, 0, 0>(in, out);
    run<1, 0, 0>(in, out);
    run<2, 0, 0>(in, out);
    run<4, 0, 0>(in, out);
    run<8, 0, 0>(in, out);
    run<0, 2, 0>(in, out);
    run<0, 4, 0>(in, out);
    run<0, 8, 0>(in, out);
    run<0, 0, 1>(in, out);
    run<2, 2, 1>(in, out);
    run<4, 4, 1>(in, out);
    return 0;
}
