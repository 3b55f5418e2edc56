// Normalisation kernels (HBM-bound): InstanceNorm2d(affine=False, track_running_stats=True)
// and train-mode BatchNorm2d(affine=True), each with the following activation and an
// optional residual fused into the apply pass.
//
// Both norms are built from per-(n,c)-plane statistics: a plane is contiguous in NCHW, so
// one workgroup streams one plane with coalesced float4 loads, reduces with wave64
// shuffles + one LDS hop, and the batch-norm statistics are the Chan merge of the N
// plane statistics of a channel (exact, order-fixed, no atomics).
//
// Reference: models/networks.py:22-34 (get_norm_layer), models/resnet.py:47-71.
#include "common.h"

namespace pcgan {

// one workgroup per plane; mean and M2 = sum (x - mean)^2 by an exact two-pass
__global__ void __launch_bounds__(256) plane_stats_kernel(const float* __restrict__ x, float* __restrict__ mean_nc,
                                                          float* __restrict__ m2_nc, int HW) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    const float* xp = x + plane * (size_t)HW;
    float s = 0.f;
    if ((HW & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += xp[i];
    }
    const float mean = block_sum(s, scratch) / (float)HW;
    float q = 0.f;
    if ((HW & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float a = xp[i] - mean;
            q += a * a;
        }
    }
    const float m2 = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        mean_nc[plane] = mean;
        m2_nc[plane] = m2;
    }
}

__global__ void bn_merge_kernel(const float* __restrict__ mean_nc, const float* __restrict__ m2_nc,
                                float* __restrict__ mean_c, float* __restrict__ var_c, float* running_mean,
                                float* running_var, int N, int C, int HW, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float msum = 0.f;
    for (int n = 0; n < N; ++n) msum += mean_nc[n * C + c];
    const float mean = msum / (float)N;
    float m2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float d = mean_nc[n * C + c] - mean;
        m2 += m2_nc[n * C + c] + d * d * (float)HW;
    }
    const float cnt = (float)N * (float)HW;
    if (mean_c) mean_c[c] = mean;
    if (var_c) var_c[c] = m2 / cnt;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (cnt - 1.f));
}

__global__ void in_running_kernel(const float* __restrict__ mean_nc, const float* __restrict__ m2_nc,
                                  float* running_mean, float* running_var, int N, int C, int HW,
                                  float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float ms = 0.f, vs = 0.f;
    for (int n = 0; n < N; ++n) {
        // each of the N copies is updated, then averaged (torch instance_norm semantics)
        ms += (1.f - momentum) * running_mean[c] + momentum * mean_nc[n * C + c];
        vs += (1.f - momentum) * running_var[c] + momentum * (m2_nc[n * C + c] / (float)(HW - 1));
    }
    running_mean[c] = ms / (float)N;
    running_var[c] = vs / (float)N;
}

struct NormArgs {
    const float* x;
    const float* y;
    const float* dy;
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    const float* residual;
    const float* s1;
    const float* s2;
    float* out;
    float* out2;
    int N, C, HW, per_plane, act;
    float eps, slope, inv_cnt;
};

__device__ __forceinline__ void plane_coeffs(const NormArgs& a, size_t plane, float& mean, float& rstd,
                                             float& g, float& b) {
    const int c = (int)(plane % a.C);
    const size_t si = a.per_plane ? plane : (size_t)c;
    mean = a.mean[si];
    const float var = a.per_plane ? a.var[si] / (float)a.HW : a.var[si];  // per-plane passes M2
    rstd = rsqrtf(var + a.eps);
    g = a.gamma ? a.gamma[c] : 1.f;
    b = a.beta ? a.beta[c] : 0.f;
}

__global__ void __launch_bounds__(256) norm_act_fwd_kernel(NormArgs a) {
    const size_t plane = blockIdx.x;
    float mean, rstd, g, b;
    plane_coeffs(a, plane, mean, rstd, g, b);
    const float sc = rstd * g, sh = b - mean * rstd * g;
    const float* xp = a.x + plane * (size_t)a.HW;
    const float* rp = a.residual ? a.residual + plane * (size_t)a.HW : nullptr;
    float* yp = a.out + plane * (size_t)a.HW;
    if ((a.HW & 3) == 0) {
        for (int i = threadIdx.x; i < (a.HW >> 2); i += blockDim.x) {
            float4 v = reinterpret_cast<const float4*>(xp)[i];
            v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
            if (rp) {
                const float4 r = reinterpret_cast<const float4*>(rp)[i];
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            v.x = act_apply(v.x, a.act, a.slope); v.y = act_apply(v.y, a.act, a.slope);
            v.z = act_apply(v.z, a.act, a.slope); v.w = act_apply(v.w, a.act, a.slope);
            reinterpret_cast<float4*>(yp)[i] = v;
        }
    } else {
        for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
            float v = xp[i] * sc + sh;
            if (rp) v += rp[i];
            yp[i] = act_apply(v, a.act, a.slope);
        }
    }
}

// s1[plane] = sum g, s2[plane] = sum g * xhat, g = dy * act'(y)
__global__ void __launch_bounds__(256) norm_bwd_stats_kernel(NormArgs a) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    float mean, rstd, g_, b_;
    plane_coeffs(a, plane, mean, rstd, g_, b_);
    const float* xp = a.x + plane * (size_t)a.HW;
    const float* dp = a.dy + plane * (size_t)a.HW;
    const float* yp = (a.act != PCGAN_ACT_NONE) ? a.y + plane * (size_t)a.HW : nullptr;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
        float g = dp[i];
        if (yp) g *= act_grad_from_out(yp[i], a.act, a.slope);
        s1 += g;
        s2 += g * ((xp[i] - mean) * rstd);
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        a.out[plane] = s1;
        a.out2[plane] = s2;
    }
}

__global__ void bn_bwd_reduce_kernel(const float* __restrict__ s1_nc, const float* __restrict__ s2_nc,
                                     float* __restrict__ s1_c, float* __restrict__ s2_c, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, b = 0.f;
    for (int n = 0; n < N; ++n) {
        a += s1_nc[n * C + c];
        b += s2_nc[n * C + c];
    }
    s1_c[c] = a;
    s2_c[c] = b;
}

__global__ void __launch_bounds__(256) norm_bwd_apply_kernel(NormArgs a) {
    const size_t plane = blockIdx.x;
    float mean, rstd, g_, b_;
    plane_coeffs(a, plane, mean, rstd, g_, b_);
    const int c = (int)(plane % a.C);
    const size_t si = a.per_plane ? plane : (size_t)c;
    const float m1 = a.s1[si] * a.inv_cnt, m2 = a.s2[si] * a.inv_cnt;
    const float k = rstd * g_;
    const float* xp = a.x + plane * (size_t)a.HW;
    const float* dp = a.dy + plane * (size_t)a.HW;
    const float* yp = (a.act != PCGAN_ACT_NONE) ? a.y + plane * (size_t)a.HW : nullptr;
    float* op = a.out + plane * (size_t)a.HW;
    float* rp = a.out2 ? a.out2 + plane * (size_t)a.HW : nullptr;
    for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
        float g = dp[i];
        if (yp) g *= act_grad_from_out(yp[i], a.act, a.slope);
        const float xh = (xp[i] - mean) * rstd;
        op[i] = k * (g - m1 - xh * m2);
        if (rp) rp[i] = g;
    }
}

static inline int plane_threads(int HW) { return HW >= 1024 ? 256 : (HW >= 256 ? 128 : 64); }

}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_plane_stats(const float* x, float* mean_nc, float* m2_nc, int NC, int HW, pcgan_stream_t s) {
    PCGAN_CHECK(x && mean_nc && m2_nc && NC > 0 && HW > 0, "plane_stats: bad arguments");
    hipLaunchKernelGGL(plane_stats_kernel, dim3(NC), dim3(plane_threads(HW)), 0, (hipStream_t)s, x, mean_nc, m2_nc,
                       HW);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_merge(const float* mean_nc, const float* m2_nc, float* mean_c, float* var_c,
                              float* running_mean, float* running_var, int N, int C, int HW, float momentum,
                              pcgan_stream_t s) {
    PCGAN_CHECK(mean_nc && m2_nc && N > 0 && C > 0 && HW > 0, "bn_merge: bad arguments");
    hipLaunchKernelGGL(bn_merge_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, mean_nc, m2_nc, mean_c,
                       var_c, running_mean, running_var, N, C, HW, momentum);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_in_running_update(const float* mean_nc, const float* m2_nc, float* running_mean,
                                       float* running_var, int N, int C, int HW, float momentum,
                                       pcgan_stream_t s) {
    PCGAN_CHECK(mean_nc && m2_nc && running_mean && running_var && N > 0 && C > 0 && HW > 1,
                "in_running_update: bad arguments");
    hipLaunchKernelGGL(in_running_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, mean_nc, m2_nc,
                       running_mean, running_var, N, C, HW, momentum);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_act_fwd(const float* x, const float* mean, const float* var, const float* gamma,
                                  const float* beta, const float* residual, float* y, int N, int C, int HW,
                                  int per_plane, float eps, int act, float slope, pcgan_stream_t s) {
    PCGAN_CHECK(x && mean && var && y && N > 0 && C > 0 && HW > 0, "norm_act_fwd: bad arguments");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.residual = residual; a.out = y;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    hipLaunchKernelGGL(norm_act_fwd_kernel, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_bwd_stats(const float* dy, const float* x, const float* y, const float* mean,
                                    const float* var, float* s1_nc, float* s2_nc, int N, int C, int HW,
                                    int per_plane, float eps, int act, float slope, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean && var && s1_nc && s2_nc, "norm_bwd_stats: null pointer");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "norm_bwd_stats: activation mask needs y");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean; a.var = var; a.out = s1_nc; a.out2 = s2_nc;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    hipLaunchKernelGGL(norm_bwd_stats_kernel, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_bwd_reduce(const float* s1_nc, const float* s2_nc, float* s1_c, float* s2_c, int N, int C,
                                   pcgan_stream_t s) {
    PCGAN_CHECK(s1_nc && s2_nc && s1_c && s2_c, "bn_bwd_reduce: null pointer");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, s1_nc, s2_nc, s1_c,
                       s2_c, N, C);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_bwd_apply(const float* dy, const float* x, const float* y, const float* mean,
                                    const float* var, const float* gamma, const float* s1, const float* s2,
                                    float* dx, float* d_residual, int N, int C, int HW, int per_plane, float eps,
                                    int act, float slope, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean && var && s1 && s2 && dx, "norm_bwd_apply: null pointer");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "norm_bwd_apply: activation mask needs y");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean; a.var = var; a.gamma = gamma; a.s1 = s1; a.s2 = s2;
    a.out = dx; a.out2 = d_residual;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    a.inv_cnt = 1.f / (per_plane ? (float)HW : (float)N * (float)HW);
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
