// Shared helpers for libpcgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/pcgan_hip.h"

namespace pcgan {

// thread-local error text behind pcgan_last_error()
void set_error(const char* fmt, ...);

#define PCGAN_CHECK(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            pcgan::set_error(__VA_ARGS__);    \
            return 1;                         \
        }                                     \
    } while (0)

#define PCGAN_LAUNCH_CHECK()                                                      \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            pcgan::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                             hipGetErrorString(e__));                             \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case PCGAN_ACT_RELU: return v > 0.f ? v : 0.f;
        case PCGAN_ACT_LRELU: return v > 0.f ? v : v * slope;
        case PCGAN_ACT_TANH: return tanhf(v);
        case PCGAN_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
    switch (act) {
        case PCGAN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case PCGAN_ACT_LRELU: return y > 0.f ? 1.f : slope;
        case PCGAN_ACT_TANH: return 1.f - y * y;
        case PCGAN_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// wave64 all-reduce sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); scratch: >= 16 floats of LDS.
// Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

}  // namespace pcgan
