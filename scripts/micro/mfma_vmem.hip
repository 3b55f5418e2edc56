// Micro-benchmark: cost of buffer_load_dword issued between v_mfma_f32_32x32x2_f32 -- all 64 lanes active vs 4 lanes
// (exec-masked) vs all lanes out of range.  2 waves / SIMD, loads hit L2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NL, int MODE>   // MODE 0: all lanes, 1: exec = 4 lanes, 2: all lanes out of range, 3: 64 lanes x 64 different rows
__global__ void __launch_bounds__(256) mix(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(tid * 8 + i) & 0xffff];
        b[i] = in[(tid * 8 + 4 + i) & 0xffff];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 22, 0x00020000);
    unsigned voff = MODE == 2 ? 0x80000000u : (MODE == 3 ? (unsigned)((tid & 63) * 4096 + blockIdx.x % 64 * 4) : (unsigned)((tid & 63) * 4 + (blockIdx.x % 64) * 256));
    const unsigned long long mask = MODE == 1 ? 0x0000000100010101ull : ~0ull;
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    unsigned long long save;
                    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tbuffer_load_dword %0, %2, %4, 0 offen\n\ts_mov_b64 exec, %1"
                                 : "+v"(d[(u * 4 + i + k) & 7]), "=&s"(save)
                                 : "v"(voff), "s"(mask), "s"(rs)
                                 : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        asm volatile("s_waitcnt vmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += d[i];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[threadIdx.x + blockIdx.x * 256] = s;
}

template <int NL, int MODE>
void run(const float* in, float* out) {
    const int wgs = 512, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mix<NL, MODE>), dim3(wgs), dim3(256), 0, 0, in, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = (double)wgs * 4 * iters * 16 * 4096.0;
    const char* names[] = {"all 64 lanes, coalesced", "4 lanes (exec-masked)", "all lanes out of range", "64 lanes x 64 rows"};
    printf("%d buffer_load_dword per MFMA, %-26s: %.3f ms  %.1f TFLOP/s\n", NL, names[MODE], best, flop / best / 1e9);
}

int main() {
    float *in, *out;
    (void)hipMalloc(&in, 1 << 22);
    (void)hipMalloc(&out, 512 * 256 * 4);
    (void)hipMemset(in, 0, 1 << 22);
    run<0, 0>(in, out);
    run<0, 0>(in, out);
    run<1, 0>(in, out);
    run<1, 1>(in, out);
    run<1, 2>(in, out);
    run<1, 3>(in, out);
    run<2, 0>(in, out);
    run<2, 1>(in, out);
    run<2, 2>(in, out);
    return 0;
}
