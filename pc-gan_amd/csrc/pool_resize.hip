// Pooling and bilinear resize (HBM-bound gathers).  Backward passes are written as
// GATHERS over the input grid (each input element sums the few outputs that reference
// it) so they are deterministic and need no atomics.
//
// Reference: nn.MaxPool2d in models/resnet.py:138 and models/networks.py:1225-1235,
// nn.AvgPool2d(H)/nn.MaxPool2d(H) in models/networks.py:1056-1059,
// F.interpolate(mode='bilinear', align_corners=True) in util/util.py:111-117.
#include "common.h"
#include <float.h>

namespace pcgan {

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int32_t* __restrict__ arg,
                                   int H, int W, int k, int stride, int pad, int P, int Q, size_t total) {
    const size_t nc = blockIdx.y;   // plane; blockIdx.x = chunk of the output plane (32-bit index math)
    for (int me = blockIdx.x * blockDim.x + threadIdx.x; me < P * Q; me += gridDim.x * blockDim.x) {
        const int p = me / Q;
        const int q = me - p * Q;
        const size_t i = nc * (size_t)P * Q + me;
        const T* xp = x + nc * (size_t)H * W;
        const int y0 = p * stride - pad, x0 = q * stride - pad;
        float best = -FLT_MAX;
        int bi = -1;
        for (int r = 0; r < k; ++r) {
            const int iy = y0 + r;
            if (iy < 0 || iy >= H) continue;
            for (int s = 0; s < k; ++s) {
                const int ix = x0 + s;
                if (ix < 0 || ix >= W) continue;
                const float v = ld1(xp + iy * W + ix);
                if (bi < 0 || v > best || v != v) {  // first maximum in scan order; NaN propagates
                    best = v;
                    bi = iy * W + ix;
                }
            }
        }
        st1(y + i, best);        // (a stored bf16 value is exact again in bf16: max of stored values)
        arg[i] = bi;
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ arg, T* __restrict__ dx,
                                   int H, int W, int k, int stride, int pad, int P, int Q, size_t total) {
    // blockIdx.y = plane, blockIdx.x = 256-element chunk of the plane: no 64-bit divisions per element
    const size_t nc = blockIdx.y;
    for (int me = blockIdx.x * blockDim.x + threadIdx.x; me < H * W; me += gridDim.x * blockDim.x) {
        const int iy = me / W;
        const int ix = me - iy * W;
        const size_t i = nc * (size_t)H * W + me;
        // output rows p with p*stride - pad <= iy <= p*stride - pad + k - 1
        int p_lo = (iy + pad - k + 1 + stride - 1);
        p_lo = p_lo <= 0 ? 0 : p_lo / stride;
        int p_hi = (iy + pad) / stride;
        if (p_hi > P - 1) p_hi = P - 1;
        int q_lo = (ix + pad - k + 1 + stride - 1);
        q_lo = q_lo <= 0 ? 0 : q_lo / stride;
        int q_hi = (ix + pad) / stride;
        if (q_hi > Q - 1) q_hi = Q - 1;
        float acc = 0.f;
        const size_t ob = nc * (size_t)P * Q;
        for (int p = p_lo; p <= p_hi; ++p)
            for (int q = q_lo; q <= q_hi; ++q)
                if (arg[ob + p * Q + q] == me) acc += ld1(dy + ob + p * Q + q);
        st1(dx + i, acc);
    }
}

// one wave per plane
template <typename T>
__global__ void global_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int32_t* __restrict__ arg,
                                       int NC, int HW, int is_max) {
    const int plane = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (plane >= NC) return;
    const int lane = threadIdx.x & 63;
    const T* xp = x + (size_t)plane * HW;
    if (is_max) {
        float best = -FLT_MAX;
        int bi = 0x7fffffff;
        for (int i = lane; i < HW; i += 64) {
            const float v = ld1(xp + i);
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { st1(y + plane, best); arg[plane] = bi; }
    } else {
        float s = 0.f;
        for (int i = lane; i < HW; i += 64) s += ld1(xp + i);
        s = wave_sum(s);
        if (lane == 0) st1(y + plane, s / (float)HW);
    }
}

template <typename T>
__global__ void global_pool_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ arg,
                                       T* __restrict__ dx, int HW, int is_max, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t plane = i / HW;
        const int e = (int)(i - plane * HW);
        if (is_max) st1(dx + i, (arg[plane] == e) ? ld1(dy + plane) : 0.f);
        else st1(dx + i, ld1(dy + plane) / (float)HW);
    }
}

// source index arithmetic identical to ATen's area_pixel_compute_source_index for
// align_corners=True: scale = (in-1)/(out-1) in fp32, src = scale * dst
__device__ __forceinline__ void bilin_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int P, int Q,
                                    float sh, float sw, size_t total) {
    const size_t nc = blockIdx.y;   // plane; blockIdx.x = chunk of the output plane (32-bit index math)
    for (int me = blockIdx.x * blockDim.x + threadIdx.x; me < P * Q; me += gridDim.x * blockDim.x) {
        const int p = me / Q;
        const int q = me - p * Q;
        const size_t i = nc * (size_t)P * Q + me;
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(p, sh, H, y0, y1, ly);
        bilin_src(q, sw, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const T* xp = x + nc * (size_t)H * W;
        st1(y + i, hy * (hx * ld1(xp + y0 * W + x0) + lx * ld1(xp + y0 * W + x1)) +
                       ly * (hx * ld1(xp + y1 * W + x0) + lx * ld1(xp + y1 * W + x1)));
    }
}

// gather form of the backward: for input row iy the output rows p whose (y0,y1) touch it
// lie in [ceil((iy-1)/sh), floor((iy+1)/sh)]; every candidate is re-derived exactly.
template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int P, int Q,
                                    float sh, float sw, size_t total) {
    const size_t nc = blockIdx.y;   // plane; blockIdx.x = chunk of the plane (32-bit index math)
    for (int me = blockIdx.x * blockDim.x + threadIdx.x; me < H * W; me += gridDim.x * blockDim.x) {
        const int iy = me / W;
        const int ix = me - iy * W;
        const size_t i = nc * (size_t)H * W + me;
        int p_lo = 0, p_hi = P - 1, q_lo = 0, q_hi = Q - 1;
        if (sh > 0.f) {
            p_lo = (int)floorf((float)(iy - 1) / sh) - 1;
            p_hi = (int)ceilf((float)(iy + 1) / sh) + 1;
            if (p_lo < 0) p_lo = 0;
            if (p_hi > P - 1) p_hi = P - 1;
        }
        if (sw > 0.f) {
            q_lo = (int)floorf((float)(ix - 1) / sw) - 1;
            q_hi = (int)ceilf((float)(ix + 1) / sw) + 1;
            if (q_lo < 0) q_lo = 0;
            if (q_hi > Q - 1) q_hi = Q - 1;
        }
        const T* dp = dy + nc * (size_t)P * Q;
        float acc = 0.f;
        for (int p = p_lo; p <= p_hi; ++p) {
            int y0, y1;
            float ly;
            bilin_src(p, sh, H, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            for (int q = q_lo; q <= q_hi; ++q) {
                int x0, x1;
                float lx;
                bilin_src(q, sw, W, x0, x1, lx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx != 0.f) acc += wy * wx * ld1(dp + p * Q + q);
            }
        }
        st1(dx + i, acc);
    }
}

static inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b == 0 ? 1 : b));
}

}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_maxpool_fwd(const void* x, void* y, int32_t* argmax, int NC, int H, int W, int k, int stride,
                                 int pad, int P, int Q, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && y && argmax && NC > 0 && H > 0 && W > 0 && k > 0 && stride > 0, "maxpool_fwd: bad arguments");
    PCGAN_CHECK(P == (H + 2 * pad - k) / stride + 1 && Q == (W + 2 * pad - k) / stride + 1,
                "maxpool_fwd: output dims do not match (floor mode)");
    const size_t total = (size_t)NC * P * Q;
    PCGAN_CHECK(NC <= 65535, "maxpool_fwd: more than 65535 planes");
    const int bx = (P * Q + 255) / 256;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(bx > 1024 ? 1024 : bx, NC), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)x, (T*)y, argmax, H, W, k, stride, pad, P, Q, total));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_maxpool_bwd(const void* dy, const int32_t* argmax, void* dx, int NC, int H, int W, int k,
                                 int stride, int pad, int P, int Q, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && argmax && dx && NC > 0, "maxpool_bwd: bad arguments");
    const size_t total = (size_t)NC * H * W;
    PCGAN_CHECK(NC <= 65535, "maxpool_bwd: more than 65535 planes");
    const int bx = (H * W + 255) / 256;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(bx > 1024 ? 1024 : bx, NC), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)dy, argmax, (T*)dx, H, W, k, stride, pad, P, Q, total));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_global_pool_fwd(const void* x, void* y, int32_t* argmax, int NC, int HW, int is_max, int dtype,
                                     pcgan_stream_t s) {
    PCGAN_CHECK(x && y && NC > 0 && HW > 0 && (!is_max || argmax), "global_pool_fwd: bad arguments");
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(global_pool_fwd_kernel<T>, dim3((NC + 3) / 4), dim3(256), 0, (hipStream_t)s, (const T*)x,
                                                    (T*)y, argmax, NC, HW, is_max));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_global_pool_bwd(const void* dy, const int32_t* argmax, void* dx, int NC, int HW, int is_max, int dtype,
                                     pcgan_stream_t s) {
    PCGAN_CHECK(dy && dx && NC > 0 && HW > 0 && (!is_max || argmax), "global_pool_bwd: bad arguments");
    const size_t total = (size_t)NC * HW;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(global_pool_bwd_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)dy, argmax, (T*)dx, HW, is_max, total));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int pcgan_bilinear_fwd(const void* x, void* y, int NC, int H, int W, int P, int Q, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && y && NC > 0 && H > 0 && W > 0 && P > 0 && Q > 0, "bilinear_fwd: bad arguments");
    const size_t total = (size_t)NC * P * Q;
    PCGAN_CHECK(NC <= 65535, "bilinear_fwd: more than 65535 planes");
    const int bx = (P * Q + 255) / 256;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(bilinear_fwd_kernel<T>, dim3(bx > 1024 ? 1024 : bx, NC), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)x, (T*)y, H, W, P, Q, ac_scale(H, P), ac_scale(W, Q), total));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bilinear_bwd(const void* dy, void* dx, int NC, int H, int W, int P, int Q, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && dx && NC > 0 && H > 0 && W > 0 && P > 0 && Q > 0, "bilinear_bwd: bad arguments");
    const size_t total = (size_t)NC * H * W;
    PCGAN_CHECK(NC <= 65535, "bilinear_bwd: more than 65535 planes");
    const int bx = (H * W + 255) / 256;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(bilinear_bwd_kernel<T>, dim3(bx > 1024 ? 1024 : bx, NC), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)dy, (T*)dx, H, W, P, Q, ac_scale(H, P), ac_scale(W, Q), total));
    PCGAN_LAUNCH_CHECK();
    return 0;
}
