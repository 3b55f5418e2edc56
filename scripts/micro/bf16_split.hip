// Micro-benchmark behind DESIGN.md section 7 ("next on the kernel side"): can the fp32 convolutions run on the bf16 matrix
// pipe without leaving fp32 accuracy?  x = h + m + l with three bf16 pieces (8 + 8 + 8 mantissa bits) is an exact split of
// an fp32 value; a product a*b then needs the piece pairs (h,h) | (h,m) (m,h) | (h,l) (m,m) (l,h) to reach 2^-24.
// Measures (1) the sustained rate of v_mfma_f32_32x32x16_bf16 on the whole chip and of the 3- and 6-instruction groups
// that stand for one fp32 K=16 step, against v_mfma_f32_32x32x2_f32; (2) the error of a 32x32 tile with K = 2304 (the
// residual-block convolution) against a float64 host reference for fp32 MFMA, 1, 3 and 6 products.
// Build: make -C scripts/micro bf16_split ; run on the GPU box: scripts/micro/bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// ---- (1) rates ------------------------------------------------------------------------------------------------------
template <int MODE>   // 0: fp32 32x32x2 ; 1: bf16 32x32x16 ; 2: bf16 16x16x32 ; 3: f16 32x32x16
__global__ void __launch_bounds__(256) rate_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (MODE == 0) {
        float a[4], b[4];
        for (int i = 0; i < 4; ++i) {
            a[i] = in[(tid * 8 + i) & 0xffff];
            b[i] = in[(tid * 8 + 4 + i) & 0xffff];
        }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    } else if (MODE == 3) {
        f16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (_Float16)in[(tid * 64 + i * 8 + j) & 0xffff];
                b[i][j] = (_Float16)in[(tid * 64 + 32 + i * 8 + j) & 0xffff];
            }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    } else if (MODE == 2) {   // same FLOP per iteration: 32 instructions of 16x16x32 (16 accumulators of 4 registers)
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (__bf16)in[(tid * 64 + i * 8 + j) & 0xffff];
                b[i][j] = (__bf16)in[(tid * 64 + 32 + i * 8 + j) & 0xffff];
            }
        f32x4 c4[16];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) c4[i][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u * 2 + (i >> 3)) & 3], b[i & 3], c4[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc[i & 3][r] += c4[i][r];
    } else {
        bf16x8 a[4], b[4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) {
                a[i][j] = (__bf16)in[(tid * 64 + i * 8 + j) & 0xffff];
                b[i][j] = (__bf16)in[(tid * 64 + 32 + i * 8 + j) & 0xffff];
            }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
}

// ---- (2) accuracy: one wave = one 32x32 tile, C = A[32][K] * B[K][32] -------------------------------------------------
// operand layout of the 32x32 MFMAs: lane l holds row/column l % 32; fp32 x2: k = l / 32; bf16 x16: k = 8 * (l / 32) + j
// accumulator: c[r] = C[(r / 4) * 8 + (l / 32) * 4 + r % 4][l % 32]
template <int NPROD>   // 0: fp32 MFMA ; 1, 3, 6: bf16 piece products
__global__ void __launch_bounds__(64) tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K,
                                                  float sa = 1.f, float sb = 1.f) {
    const int l = threadIdx.x, lo = l & 31, hi = l >> 5;
    const float* a = A + (size_t)blockIdx.x * 32 * K;
    const float* b = B + (size_t)blockIdx.x * K * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (NPROD == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[lo * K + k + hi], b[(k + hi) * 32 + lo], acc, 0, 0, 0);
    } else if (NPROD < 0) {   // two fp16 pieces per operand (11 + 11 significand bits), operands scaled by powers of two, 3 products
        for (int k = 0; k < K; k += 16) {
            f16x8 ah, al, bh, bl;
            for (int j = 0; j < 8; ++j) {
                const float x = a[lo * K + k + hi * 8 + j] * sa, y = b[(k + hi * 8 + j) * 32 + lo] * sb;
                ah[j] = (_Float16)x;
                al[j] = (_Float16)(x - (float)ah[j]);
                bh[j] = (_Float16)y;
                bl[j] = (_Float16)(y - (float)bh[j]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            if (NPROD == -4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc[r] *= 1.f / (sa * sb);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 h, m, q;
                split3(a[lo * K + k + hi * 8 + j], h, m, q);
                ah[j] = h; am[j] = m; al[j] = q;
                split3(b[(k + hi * 8 + j) * 32 + lo], h, m, q);
                bh[j] = h; bm[j] = m; bl[j] = q;
            }
            // smallest terms first
            if (NPROD >= 6) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            }
            if (NPROD >= 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    float* c = C + (size_t)blockIdx.x * 1024;
    for (int r = 0; r < 16; ++r) c[((r / 4) * 8 + hi * 4 + r % 4) * 32 + lo] = acc[r];
}

static float gauss() {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (float)rand() / RAND_MAX;
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main() {
    // (1)
    const int wgs = 512, iters = 20000;
    float *in, *out;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    std::vector<float> h(65536);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double rate[4] = {0, 0, 0, 0};
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            else if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(rate_kernel<3>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double kdepth = mode ? 16 : 2;      // (mode 2: 32 instructions of 16x16x32 = the FLOP of 16 of 32x32x16)
            const double flop = (double)wgs * 4 * iters * 16 * (2.0 * 32 * 32 * kdepth);
            rate[mode] = flop / ms / 1e9;
            const char* mn[4] = {"v_mfma_f32_32x32x2_f32  ", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_f16 "};
            printf("%s  %.3f ms  %.1f TFLOP/s\n", mn[mode], ms, rate[mode]);
        }
    printf("fp32-equivalent rate of the bf16 pipe: 3 products %.1f TFLOP/s, 6 products %.1f TFLOP/s (fp32 MFMA %.1f)\n", rate[1] / 3,
           rate[1] / 6, rate[0]);

    // (2)
    const int T = 64, K = 2304;
    std::vector<float> A((size_t)T * 32 * K), B((size_t)T * K * 32);
    for (auto& v : A) v = fmaxf(gauss(), 0.f);        // activations after InstanceNorm + ReLU
    for (auto& v : B) v = 0.02f * gauss();            // weights ~ N(0, 0.02)
    std::vector<double> ref((size_t)T * 1024);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)A[((size_t)t * 32 + i) * K + k] * (double)B[((size_t)t * K + k) * 32 + j];
                ref[(size_t)t * 1024 + i * 32 + j] = s;
            }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dC, (size_t)T * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)T * 1024);
    const char* names[6] = {"fp32 MFMA 32x32x2        ", "bf16 (h,h)       1 product", "bf16 split       3 products", "bf16 split       6 products",
                            "fp16 2-piece     3 products", "fp16 2-piece     4 products"};
    float amax = 0.f, bmax = 0.f;
    for (auto v : A) amax = fmaxf(amax, fabsf(v));
    for (auto v : B) bmax = fmaxf(bmax, fabsf(v));
    const float sa = exp2f(14.f - ceilf(log2f(amax))), sb = exp2f(14.f - ceilf(log2f(bmax)));
    printf("fp16 scales: activations max %.3f x %.0f, weights max %.4f x %.0f\n", amax, sa, bmax, sb);
    for (int v = 0; v < 6; ++v) {
        if (v == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
        if (v == 1) hipLaunchKernelGGL(tile_kernel<1>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
        if (v == 2) hipLaunchKernelGGL(tile_kernel<3>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
        if (v == 3) hipLaunchKernelGGL(tile_kernel<6>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
        if (v == 4) hipLaunchKernelGGL(tile_kernel<-3>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, sa, sb);
        if (v == 5) hipLaunchKernelGGL(tile_kernel<-4>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K, sa, sb);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0, mx = 0;
        for (size_t i = 0; i < C.size(); ++i) {
            const double d = (double)C[i] - ref[i];
            num += d * d;
            den += ref[i] * ref[i];
            if (fabs(d) > mx) mx = fabs(d);
        }
        printf("%s  rel. L2 error %.3e   max abs error %.3e   (K = %d, |C| rms %.3f)\n", names[v], sqrt(num / den), mx, K, sqrt(den / C.size()));
    }
    return 0;
}
