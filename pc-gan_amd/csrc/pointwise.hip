// Pointwise / small-reduction kernels (all HBM-bound, float4-vectorised where the
// shape allows): activation forward/backward, the z-channel concatenation of the
// generator / discriminator input, bias-gradient channel sums, Dropout2d scaling.
#include "common.h"

namespace pcgan {

static inline int ew_blocks(size_t n_vec) {
    size_t b = (n_vec + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b == 0 ? 1 : b));
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float slope) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = act_apply(x[i], act, slope);
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               size_t n, int act, float slope) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 v = reinterpret_cast<const float4*>(y)[i];
        g.x *= act_grad_from_out(v.x, act, slope); g.y *= act_grad_from_out(v.y, act, slope);
        g.z *= act_grad_from_out(v.z, act, slope); g.w *= act_grad_from_out(v.w, act, slope);
        reinterpret_cast<float4*>(dx)[i] = g;
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = dy[i] * act_grad_from_out(y[i], act, slope);
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = reinterpret_cast<const float4*>(a)[i];
        const float4 v = reinterpret_cast<const float4*>(b)[i];
        u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
        reinterpret_cast<float4*>(y)[i] = u;
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a[i] + b[i];
}

// y = alpha * (s ? s[0] : 1) * x   (upstream scalar gradient of a loss lives on the device)
__global__ void scale_kernel(const float* __restrict__ x, const float* __restrict__ sdev, float alpha,
                             float* __restrict__ y, size_t n) {
    const float k = alpha * (sdev ? sdev[0] : 1.f);
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = reinterpret_cast<const float4*>(x)[i];
        u.x *= k; u.y *= k; u.z *= k; u.w *= k;
        reinterpret_cast<float4*>(y)[i] = u;
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i] * k;
}

// blockIdx.x = output plane (n, c_out), blockIdx.y = 4096-element chunk of it (128-bit accesses when HW % 4 == 0)
__global__ void concat_z_kernel(const float* __restrict__ img, const float* __restrict__ z, float* __restrict__ out,
                                int C, int nz, int HW, int z_batch) {
    const int Ct = C + nz;
    const int n = blockIdx.x / Ct, c = blockIdx.x % Ct;
    const int lo = blockIdx.y * 4096, hi = lo + 4096 < HW ? lo + 4096 : HW;
    float* op = out + (size_t)blockIdx.x * HW;
    const float* ip = c < C ? img + ((size_t)n * C + c) * HW : nullptr;
    const float v = c < C ? 0.f : z[(z_batch == 1 ? 0 : n) * nz + (c - C)];
    if ((HW & 3) == 0) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += blockDim.x * 4)
            *reinterpret_cast<float4*>(op + i) = ip ? *reinterpret_cast<const float4*>(ip + i) : make_float4(v, v, v, v);
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) op[i] = ip ? ip[i] : v;
    }
}

// out[c] = sum_{n,hw} x[n][c][hw]; stage 1: one workgroup per (n,c) plane, stage 2: sum over n
__global__ void __launch_bounds__(256) plane_sum_kernel(const float* __restrict__ x, float* __restrict__ part, int HW) {
    __shared__ float scratch[16];
    const float* xp = x + (size_t)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0) {       // 128-bit loads (planes are 16-byte aligned then)
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += xp[i];
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sum_over_n_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int C, int accumulate) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per channel
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += part[n * C + c];
    s = wave_sum(s);
    if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

__global__ void channel_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y,
                                     int HW, float scale) {
    const float m = mask[blockIdx.x] * scale;
    const float* xp = x + (size_t)blockIdx.x * HW;
    float* yp = y + (size_t)blockIdx.x * HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) yp[i] = xp[i] * m;
}

}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_act_fwd(const float* x, float* y, size_t n, int act, float slope, pcgan_stream_t s) {
    PCGAN_CHECK(x && y, "act_fwd: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, x, y, n, act, slope);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float slope,
                             pcgan_stream_t s) {
    PCGAN_CHECK(dy && y && dx, "act_bwd: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, dy, y, dx, n, act,
                       slope);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_add(const float* a, const float* b, float* y, size_t n, pcgan_stream_t s) {
    PCGAN_CHECK(a && b && y, "add: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, a, b, y, n);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_scale(const float* x, const float* scalar_dev, float alpha, float* y, size_t n,
                           pcgan_stream_t s) {
    PCGAN_CHECK(x && y, "scale: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, x, scalar_dev, alpha, y, n);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_concat_z(const float* img, const float* z, float* out, int N, int C, int nz, int HW,
                              int z_batch, pcgan_stream_t s) {
    PCGAN_CHECK(img && z && out && N > 0 && C > 0 && nz > 0 && HW > 0, "concat_z: bad arguments");
    PCGAN_CHECK(z_batch == 1 || z_batch == N, "concat_z: z batch %d must be 1 or N=%d", z_batch, N);
    hipLaunchKernelGGL(concat_z_kernel, dim3(N * (C + nz), (HW + 4095) / 4096), dim3(256), 0, (hipStream_t)s, img, z, out, C,
                       nz, HW, z_batch);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_channel_sum(const float* x, float* out, float* scratch_nc, int N, int C, int HW, int accumulate,
                                 pcgan_stream_t s) {
    PCGAN_CHECK(x && out && scratch_nc && N > 0 && C > 0 && HW > 0, "channel_sum: bad arguments");
    hipLaunchKernelGGL(plane_sum_kernel, dim3(N * C), dim3(HW >= 1024 ? 256 : 64), 0, (hipStream_t)s, x, scratch_nc,
                       HW);
    PCGAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_over_n_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)s, scratch_nc, out, N, C, accumulate);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_sum_planes(const float* part_nc, float* out, int N, int C, int accumulate, pcgan_stream_t s) {
    PCGAN_CHECK(part_nc && out && N > 0 && C > 0, "sum_planes: bad arguments");
    hipLaunchKernelGGL(sum_over_n_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)s, part_nc, out, N, C, accumulate);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_channel_scale(const float* x, const float* mask_nc, float* y, int NC, int HW, float scale,
                                   pcgan_stream_t s) {
    PCGAN_CHECK(x && mask_nc && y && NC > 0 && HW > 0, "channel_scale: bad arguments");
    hipLaunchKernelGGL(channel_scale_kernel, dim3(NC), dim3(HW >= 1024 ? 256 : 64), 0, (hipStream_t)s, x, mask_nc, y,
                       HW, scale);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
