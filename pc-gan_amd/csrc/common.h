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

// ---- kernel timer (measurement only; off unless pcgan_timer_enable was called) --------------------------------------------------
// HIP events on the LAUNCH stream around every launch of the kernels bench.py's roofline block reports (the three residual-block
// convolutions), whichever host path issued them (per-op call or composite).  kind < 0: no-op.
enum { TIMER_RES_FWD = 0, TIMER_RES_DGRAD = 1, TIMER_RES_WGRAD = 2, TIMER_RES_WGRAD_MAIN = 3, TIMER_KINDS = 4 };
struct TimerScope {
    int kind, slot;
    hipStream_t st;
    TimerScope(int kind, hipStream_t st);
    ~TimerScope();
};

// the residual-block convolution (256 -> 256, 3x3, stride 1, reflection padding 1): the kind to time it under, or -1
static inline int timer_kind_res(const pcgan_conv_desc* d, int kind) {
    return (d && d->K == 256 && d->C == 256 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad_mode == 1 && d->pad == 1) ? kind : -1;
}

// ---- routing options (include/pcgan_hip.h: pcgan_set_option) ------------------------------------------------------------------------
// The library reads NO environment variables (round 4): the few A/B switches that live below the C-ABI are explicit options with
// measured-best defaults, set by the host through pcgan_set_option(key, value) before the calls they affect.
enum { OPT_BSPLIT_HALO = 0, OPT_WGRAD_GEN = 1, OPT_WGRAD_PADCOPY = 2, OPT_WGRAD_CW = 3, OPT_HGEMM_BF16 = 4, OPT_WGD_LOOK = 5, OPT_WGRAD_DIRECT = 6, OPT_HGEMM_TILE = 7, OPT_HGEMM_KS = 8, OPT_WGRAD_ROWRING = 9, OPT_COUNT = 10 };
int option(int id);

// ---- non-finite sentinel of the fp16 route --------------------------------------------------------------------------------------------
// An operand element larger than the maximum its scale was derived from (a stale `_pcgan_amax`: the tensor was rewritten behind the
// host's back) overflows its fp16 piece to inf, and every product it enters becomes inf / NaN.  The epilogues of the fp16-route kernels
// therefore count workgroups that produced a non-finite result into ONE device word the host registered (pcgan_set_nonfinite_counter;
// null = off) and the host raises when it next looks (hip/ops.py: check_nonfinite) -- loud instead of silent.  One class test per
// result and one ballot per wave: nothing in the K loop.
unsigned* nonfinite_counter();
// wgrad_direct.hip: fixed-order sum of [split][tap][k][c] partial sums, scaled back by 1 / (scales[0] * scales[1]), into dW[k][c][tap]
int launch_wgd_reduce(const float* part, float* dw, const float* scales, int splits, int K, int C, int accumulate, hipStream_t st);
__device__ __forceinline__ bool is_nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ void report_nonfinite(unsigned* counter, bool bad) {
    if (counter != nullptr && __builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(counter, 1u);
}

// out[row] = largest |w| of every row of w[K][C][T] seen as the forward GEMM's A (by_c = 0: row = k) or the data gradient's (by_c = 1: row = c)
int launch_weight_row_absmax(const float* w, int K, int C, int T, int by_c, float* out, hipStream_t st);

// igemm_conv.hip: fold of the padded-grid gradient of a ReflectionPad2d input (also the last step of thin_conv.hip's head data gradient)
int launch_reflect_fold(const void* padded, void* dx, int NC, int H, int W, int pad, int dtype, hipStream_t st);

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

// ---- activation storage types ------------------------------------------------------------------------------------
// PCGAN_F32: fp32 tensors.  PCGAN_BF16: activations and their gradients are stored as bf16 in HBM (half the bytes of the
// HBM-bound kernels) while every kernel computes in fp32 registers; statistics, losses, parameters, parameter gradients and
// optimizer state stay fp32.  ld1 / ld4 / st1 / st4 are the only places that know the storage width: 4 consecutive elements
// are one 16-byte (fp32) or one 8-byte (bf16) access.  fp32 -> bf16 is round-to-nearest-even (a plain cast: v_cvt_pk_bf16_f32,
// NaN stays NaN).
typedef __bf16 bf16;

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16* p, float v) { *p = (bf16)v; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, const float4& v) {
    typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o;
    o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}

// run `...` with T = the storage type of `dtype` (host side of every entry point that takes a dtype)
#define PCGAN_DTYPE_SWITCH(dtype, T, ...)                                   \
    do {                                                                    \
        if ((dtype) == PCGAN_F32) {                                         \
            typedef float T;                                                \
            __VA_ARGS__;                                                    \
        } else if ((dtype) == PCGAN_BF16) {                                 \
            typedef pcgan::bf16 T;                                          \
            __VA_ARGS__;                                                    \
        } else {                                                            \
            pcgan::set_error("unknown dtype %d (PCGAN_F32 / PCGAN_BF16)", (int)(dtype)); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

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

// Two fp16 pieces, x * 2^e = h + l (11 + 11 significand bits), products (l,h) (h,l) (h,h): HALF the matrix instructions of the
// three-piece bf16 split.  scripts/micro/bf16_split, K = 2304 against float64: relative L2 error 5.3e-7 (fp32 MFMA 6.1e-7, bf16 x 6
// 7.0e-7).  fp16 has 5 exponent bits, so each operand tensor is scaled by a power of two that puts its largest magnitude in
// (2^13, 2^14] (the largest magnitude is computed on the device: pcgan_absmax, or handed over by the producing kernel); the
// accumulators are scaled back (exactly) in the epilogue.  An element below 2^-17 of the tensor's largest loses its low piece
// (error <= 2^-39 of the largest magnitude per element).
__device__ __forceinline__ float pow2_scale(float amax) {
    if (!(amax > 0.f)) return 1.f;
    int e;
    frexpf(amax, &e);                     // amax = m * 2^e, m in [0.5, 1)
    e = 14 - e;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.f, e);
}
// largest of n partial maxima (n small: every thread reads them all)
__device__ __forceinline__ float max_of_partials(const float* __restrict__ p, int n) {
    float m = 0.f;
    for (int i = 0; i < n; ++i) m = fmaxf(m, p[i]);
    return m;
}
static constexpr int WEIGHT_AMAX_SLOTS = 64;     // partial maxima kept for a weight tensor
// a thread's share of n partial maxima, strided over the workgroup's nt threads (block_max finishes it).  16-byte loads, four in
// flight, when the pointer allows: the element-per-iteration loop took one L2 round trip per element -- 16 in a row for the 8192
// per-plane maxima of a residual-block tensor and 512 threads, 3-6 % of a 0.09 ms convolution (scripts/namax_probe.py)
__device__ __forceinline__ float thread_max_of_partials(const float* __restrict__ p, int n, int tid, int nt) {
    float m = 0.f;
    if ((reinterpret_cast<size_t>(p) & 15) != 0) {
        for (int i = tid; i < n; i += nt) m = fmaxf(m, p[i]);
        return m;
    }
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int n4 = n >> 2;
    int i = tid;
    for (; i + 3 * nt < n4; i += 4 * nt) {
        const float4 a = p4[i], b = p4[i + nt], c = p4[i + 2 * nt], d = p4[i + 3 * nt];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
    }
    for (; i < n4; i += nt) {
        const float4 a = p4[i];
        m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    }
    for (int j = (n4 << 2) + tid; j < n; j += nt) m = fmaxf(m, p[j]);
    return m;
}

__device__ __forceinline__ void split2h(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}

// largest v over the workgroup, thread 0 gets it (others: a partial value)
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

}  // namespace pcgan
