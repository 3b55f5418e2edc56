// Pointwise / small-reduction kernels (all HBM-bound, 4 elements per access where the
// shape allows): activation forward/backward, the z-channel concatenation of the
// generator / discriminator input, bias-gradient channel sums, Dropout2d scaling, storage casts.
// Every kernel is a template over the activation storage type T (float or bf16, common.h); arithmetic is fp32.
#include "common.h"

namespace pcgan {

static inline int ew_blocks(size_t n_vec) {
    size_t b = (n_vec + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b == 0 ? 1 : b));
}

template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, int act, float slope) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = ld4(x + 4 * i);
        v.x = act_apply(v.x, act, slope); v.y = act_apply(v.y, act, slope);
        v.z = act_apply(v.z, act, slope); v.w = act_apply(v.w, act, slope);
        st4(y + 4 * i, v);
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
        st1(y + i, act_apply(ld1(x + i), act, slope));
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                               size_t n, int act, float slope) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 g = ld4(dy + 4 * i);
        const float4 v = ld4(y + 4 * i);
        g.x *= act_grad_from_out(v.x, act, slope); g.y *= act_grad_from_out(v.y, act, slope);
        g.z *= act_grad_from_out(v.z, act, slope); g.w *= act_grad_from_out(v.w, act, slope);
        st4(dx + 4 * i, g);
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride)
        st1(dx + i, ld1(dy + i) * act_grad_from_out(ld1(y + i), act, slope));
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t n) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = ld4(a + 4 * i);
        const float4 v = ld4(b + 4 * i);
        u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
        st4(y + 4 * i, u);
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) st1(y + i, ld1(a + i) + ld1(b + i));
}

// y = alpha * (s ? s[0] : 1) * x   (upstream scalar gradient of a loss lives on the device, fp32);
// TX -> TY also serves as the storage cast (alpha = 1, no scalar)
template <typename TX, typename TY>
__global__ void scale_kernel(const TX* __restrict__ x, const float* __restrict__ sdev, float alpha,
                             TY* __restrict__ y, size_t n) {
    const float k = alpha * (sdev ? sdev[0] : 1.f);
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = ld4(x + 4 * i);
        u.x *= k; u.y *= k; u.z *= k; u.w *= k;
        st4(y + 4 * i, u);
    }
    for (size_t i = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) st1(y + i, ld1(x + i) * k);
}

// blockIdx.x = output plane (n, c_out), blockIdx.y = 4096-element chunk of it (4-element accesses when HW % 4 == 0)
template <typename T>
__global__ void concat_z_kernel(const T* __restrict__ img, const float* __restrict__ z, T* __restrict__ out,
                                int C, int nz, int HW, int z_batch) {
    const int Ct = C + nz;
    const int n = blockIdx.x / Ct, c = blockIdx.x % Ct;
    const int lo = blockIdx.y * 4096, hi = lo + 4096 < HW ? lo + 4096 : HW;
    T* op = out + (size_t)blockIdx.x * HW;
    const T* ip = c < C ? img + ((size_t)n * C + c) * HW : nullptr;
    const float v = c < C ? 0.f : z[(z_batch == 1 ? 0 : n) * nz + (c - C)];     // ratings stay fp32 on the host side
    if ((HW & 3) == 0) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += blockDim.x * 4)
            st4(op + i, ip ? ld4(ip + i) : make_float4(v, v, v, v));
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) st1(op + i, ip ? ld1(ip + i) : v);
    }
}

// out[c] = sum_{n,hw} x[n][c][hw]; stage 1: one workgroup per (n,c) plane, stage 2: sum over n (fp32 partials and result)
template <typename T>
__global__ void __launch_bounds__(256) plane_sum_kernel(const T* __restrict__ x, float* __restrict__ part, int HW) {
    __shared__ float scratch[16];
    const T* xp = x + (size_t)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0) {       // 4-element loads (planes are aligned then)
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = ld4(xp + 4 * i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += ld1(xp + i);
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

template <typename T>
__global__ void channel_scale_kernel(const T* __restrict__ x, const float* __restrict__ mask, T* __restrict__ y,
                                     int HW, float scale) {
    const float m = mask[blockIdx.x] * scale;
    const T* xp = x + (size_t)blockIdx.x * HW;
    T* yp = y + (size_t)blockIdx.x * HW;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            float4 v = ld4(xp + 4 * i);
            v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            st4(yp + 4 * i, v);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) st1(yp + i, ld1(xp + i) * m);
    }
}

}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_act_fwd(const void* x, void* y, size_t n, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && y, "act_fwd: null pointer");
    if (n == 0) return 0;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(act_fwd_kernel<T>, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)x, (T*)y, n, act, slope));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_act_bwd(const void* dy, const void* y, void* dx, size_t n, int act, float slope, int dtype,
                             pcgan_stream_t s) {
    PCGAN_CHECK(dy && y && dx, "act_bwd: null pointer");
    if (n == 0) return 0;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)dy, (const T*)y, (T*)dx, n, act, slope));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_add(const void* a, const void* b, void* y, size_t n, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(a && b && y, "add: null pointer");
    if (n == 0) return 0;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(add_kernel<T>, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)a, (const T*)b, (T*)y, n));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_scale(const void* x, const float* scalar_dev, float alpha, void* y, size_t n, int dtype,
                           pcgan_stream_t s) {
    PCGAN_CHECK(x && y, "scale: null pointer");
    if (n == 0) return 0;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL((scale_kernel<T, T>), dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s,
                                                    (const T*)x, scalar_dev, alpha, (T*)y, n));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_cast(const void* x, int dtype_x, void* y, int dtype_y, size_t n, pcgan_stream_t s) {
    PCGAN_CHECK(x && y, "cast: null pointer");
    PCGAN_CHECK((dtype_x == PCGAN_F32 || dtype_x == PCGAN_BF16) && (dtype_y == PCGAN_F32 || dtype_y == PCGAN_BF16), "cast: unknown dtype");
    if (n == 0) return 0;
    const dim3 grid(ew_blocks(n / 4 + 1)), block(256);
    hipStream_t st = (hipStream_t)s;
    const float* none = nullptr;
    if (dtype_x == PCGAN_F32 && dtype_y == PCGAN_BF16)
        hipLaunchKernelGGL((scale_kernel<float, bf16>), grid, block, 0, st, (const float*)x, none, 1.f, (bf16*)y, n);
    else if (dtype_x == PCGAN_BF16 && dtype_y == PCGAN_F32)
        hipLaunchKernelGGL((scale_kernel<bf16, float>), grid, block, 0, st, (const bf16*)x, none, 1.f, (float*)y, n);
    else if (dtype_x == PCGAN_F32)
        hipLaunchKernelGGL((scale_kernel<float, float>), grid, block, 0, st, (const float*)x, none, 1.f, (float*)y, n);
    else
        hipLaunchKernelGGL((scale_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)x, none, 1.f, (bf16*)y, n);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_concat_z(const void* img, const float* z, void* out, int N, int C, int nz, int HW,
                              int z_batch, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(img && z && out && N > 0 && C > 0 && nz > 0 && HW > 0, "concat_z: bad arguments");
    PCGAN_CHECK(z_batch == 1 || z_batch == N, "concat_z: z batch %d must be 1 or N=%d", z_batch, N);
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(concat_z_kernel<T>, dim3(N * (C + nz), (HW + 4095) / 4096), dim3(256), 0,
                                                    (hipStream_t)s, (const T*)img, z, (T*)out, C, nz, HW, z_batch));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_channel_sum(const void* x, float* out, float* scratch_nc, int N, int C, int HW, int accumulate,
                                 int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && out && scratch_nc && N > 0 && C > 0 && HW > 0, "channel_sum: bad arguments");
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(plane_sum_kernel<T>, dim3(N * C), dim3(HW >= 1024 ? 256 : 64), 0, (hipStream_t)s,
                                                    (const T*)x, scratch_nc, HW));
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

extern "C" int pcgan_channel_scale(const void* x, const float* mask_nc, void* y, int NC, int HW, float scale, int dtype,
                                   pcgan_stream_t s) {
    PCGAN_CHECK(x && mask_nc && y && NC > 0 && HW > 0, "channel_scale: bad arguments");
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(channel_scale_kernel<T>, dim3(NC), dim3(HW >= 1024 ? 256 : 64), 0, (hipStream_t)s,
                                                    (const T*)x, mask_nc, (T*)y, HW, scale));
    PCGAN_LAUNCH_CHECK();
    return 0;
}
