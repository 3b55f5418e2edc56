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

// ---- "last arriver finishes the channel": the batch-norm reductions over the N planes of a channel without a second launch ----------
// The per-plane kernels below (one workgroup per (n, c) plane) can be given a ticket word per channel.  Every workgroup publishes its
// plane's partial result with WRITE-THROUGH stores (agent-scope relaxed atomic stores = `global_store ... sc1`), waits for them
// (s_waitcnt vmcnt(0)) and makes one returning agent-scope atomicAdd on ticket[c]; the workgroup whose add completes the channel's N
// arrivals reads the partials with sc1 loads (they bypass the non-coherent caches) and finishes the channel with the arithmetic of the
// stand-alone merge kernels (same order: the result does not depend on which workgroup is last).  No fences: a release fence per
// workgroup writes back the whole L2 of its XCD (MI355X_MICROARCH.md: 1.7 - 6.5 us each) -- the first version of this kernel with
// __threadfence() made the step 2.7 ms SLOWER than the three-launch form it replaces.  The arrival that finds old + 1 == N is the last
// of its call and puts the ticket back to 0 (an agent-scope atomic store, complete when the launch is): the decision does not depend on
// the history of the ticket array, so calls with DIFFERENT batch sizes may share it (the partial last batch of an epoch, test() and
// get_current_visuals() at another N: round 3's never-cleared form, (old + 1) % N == 0, picked a wrong last arriver for the rest of the
// run once a call with another N had moved the counter off a multiple of N) and a captured hipGraph replays correctly; the calls that
// share a ticket array run in stream order (a BatchNorm net never runs two passes at once: its running statistics must stay ordered
// anyway), so the reset of one launch is visible to the next.
struct BnTicket {
    unsigned* ticket;       // [C] arrival counters of this layer and pass direction, or null: no merge in this launch
    int N, C;
};
// true in every thread of the workgroup that completed channel c (call after the plane's partials were stored by thread 0)
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// call with the plane's partials already stored by thread 0 through st_wt
__device__ __forceinline__ bool bn_last_arriver(const BnTicket& t, int c, int* flag_lds) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the write-through stores have left before the ticket moves
        const unsigned old = __hip_atomic_fetch_add(&t.ticket[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = (old + 1u) == (unsigned)t.N;
        if (last) __hip_atomic_store(&t.ticket[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = last;
    }
    __syncthreads();
    return *flag_lds != 0;
}
// Chan merge of the N plane statistics of channel c by ONE wave (lane = threadIdx.x < 64): the arithmetic of bn_merge_kernel
template <bool WT>      // WT: the partials were published by other workgroups of THIS launch (sc1 loads); else by an earlier launch
__device__ __forceinline__ void bn_merge_channel(const float* mean_nc, const float* m2_nc, float* __restrict__ mean_c,
                                                 float* __restrict__ var_c, float* running_mean, float* running_var, int N, int C, int HW,
                                                 float momentum, int c, int lane) {
    float msum = 0.f;
    for (int n = lane; n < N; n += 64) msum += WT ? ld_wt(mean_nc + n * C + c) : mean_nc[n * C + c];
    const float mean = wave_sum(msum) / (float)N;
    float m2 = 0.f;
    for (int n = lane; n < N; n += 64) {
        const float d = (WT ? ld_wt(mean_nc + n * C + c) : mean_nc[n * C + c]) - mean;
        m2 += (WT ? ld_wt(m2_nc + n * C + c) : m2_nc[n * C + c]) + d * d * (float)HW;
    }
    m2 = wave_sum(m2);
    if (lane != 0) return;
    const float cnt = (float)N * (float)HW;
    if (mean_c) mean_c[c] = mean;
    if (var_c) var_c[c] = m2 / cnt;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (cnt - 1.f));
}
struct BnMergeOut {
    float *mean_c, *var_c, *running_mean, *running_var;
    long long* batches;
    float momentum;
};

// one workgroup per plane; mean and M2 = sum (x - mean)^2 by an exact two-pass
template <typename T>
__global__ void __launch_bounds__(256) plane_stats_kernel(const T* __restrict__ x, float* __restrict__ mean_nc,
                                                          float* __restrict__ m2_nc, int HW, BnTicket tk = BnTicket{nullptr, 0, 0},
                                                          BnMergeOut mo = BnMergeOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f}) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    const T* xp = x + plane * (size_t)HW;
    float s = 0.f;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = ld4(xp + 4 * i);
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += ld1(xp + i);
    }
    const float mean = block_sum(s, scratch) / (float)HW;
    float q = 0.f;
    if ((HW & 3) == 0) {
        for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
            const float4 v = ld4(xp + 4 * i);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float a = ld1(xp + i) - mean;
            q += a * a;
        }
    }
    const float m2 = block_sum(q, scratch);
    if (threadIdx.x == 0) {
        if (tk.ticket != nullptr) {
            st_wt(mean_nc + plane, mean);
            st_wt(m2_nc + plane, m2);
        } else {
            mean_nc[plane] = mean;
            m2_nc[plane] = m2;
        }
    }
    if (tk.ticket != nullptr) {      // batch norm: the workgroup that completes its channel merges the N plane statistics (bn_merge_kernel)
        __shared__ int last_flag;
        const int c = (int)(plane % (size_t)tk.C);
        if (bn_last_arriver(tk, c, &last_flag) && threadIdx.x < 64) {
            bn_merge_channel<true>(mean_nc, m2_nc, mo.mean_c, mo.var_c, mo.running_mean, mo.running_var, tk.N, tk.C, HW, mo.momentum, c, threadIdx.x);
            if (c == 0 && threadIdx.x == 0 && mo.batches) mo.batches[0] += 1;       // num_batches_tracked: once per call
        }
    }
}

// one wave per channel: lanes stride over the N plane statistics, shuffle-reduce (was one thread per channel
// walking N entries serially: 18 us of pure latency per call)
__global__ void bn_merge_kernel(const float* __restrict__ mean_nc, const float* __restrict__ m2_nc,
                                float* __restrict__ mean_c, float* __restrict__ var_c, float* running_mean,
                                float* running_var, int N, int C, int HW, float momentum) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    bn_merge_channel<false>(mean_nc, m2_nc, mean_c, var_c, running_mean, running_var, N, C, HW, momentum, c, threadIdx.x & 63);
}

__global__ void in_running_kernel(const float* __restrict__ mean_nc, const float* __restrict__ m2_nc,
                                  float* running_mean, float* running_var, int N, int C, int HW,
                                  float momentum) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    const float rm = running_mean[c], rv = running_var[c];
    float ms = 0.f, vs = 0.f;
    for (int n = lane; n < N; n += 64) {
        // each of the N copies is updated, then averaged (torch instance_norm semantics)
        ms += (1.f - momentum) * rm + momentum * mean_nc[n * C + c];
        vs += (1.f - momentum) * rv + momentum * (m2_nc[n * C + c] / (float)(HW - 1));
    }
    ms = wave_sum(ms);
    vs = wave_sum(vs);
    if (lane == 0) {
        running_mean[c] = ms / (float)N;
        running_var[c] = vs / (float)N;
    }
}

struct NormArgs {
    const void* x;          // activation tensors: storage type T of the kernel template
    const void* y;
    const void* dy;
    const float* mean;
    const float* var;
    const float* gamma;
    const float* beta;
    const void* residual;
    const float* s1;
    const float* s2;
    void* out;              // T (norm_act_fwd, norm_bwd_apply) or float (norm_bwd_stats)
    void* out2;
    float* pmax;            // norm_act_fwd / norm_bwd_apply: [N*C] largest magnitude of each plane of `out`, or null
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

template <typename T>
__global__ void __launch_bounds__(256) norm_act_fwd_kernel(NormArgs a) {
    const size_t plane = blockIdx.x;
    float mean, rstd, g, b;
    plane_coeffs(a, plane, mean, rstd, g, b);
    const float sc = rstd * g, sh = b - mean * rstd * g;
    const T* xp = (const T*)a.x + plane * (size_t)a.HW;
    const T* rp = a.residual ? (const T*)a.residual + plane * (size_t)a.HW : nullptr;
    T* yp = (T*)a.out + plane * (size_t)a.HW;
    float am = 0.f;
    if ((a.HW & 3) == 0) {
        for (int i = threadIdx.x; i < (a.HW >> 2); i += blockDim.x) {
            float4 v = ld4(xp + 4 * i);
            v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
            if (rp) {
                const float4 r = ld4(rp + 4 * i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            v.x = act_apply(v.x, a.act, a.slope); v.y = act_apply(v.y, a.act, a.slope);
            v.z = act_apply(v.z, a.act, a.slope); v.w = act_apply(v.w, a.act, a.slope);
            st4(yp + 4 * i, v);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
            float v = ld1(xp + i) * sc + sh;
            if (rp) v += ld1(rp + i);
            v = act_apply(v, a.act, a.slope);
            st1(yp + i, v);
            am = fmaxf(am, fabsf(v));
        }
    }
    if (a.pmax) {      // plane maximum for the fp16 route of the convolution that reads y (see instnorm_fwd_fused_kernel)
        __shared__ float scratch[16];
        am = block_max(am, scratch);
        if (threadIdx.x == 0) a.pmax[plane] = am;
    }
}

// s1[plane] = sum g, s2[plane] = sum g * xhat, g = dy * act'(y)
template <typename T>
__global__ void __launch_bounds__(256) norm_bwd_stats_kernel(NormArgs a, BnTicket tk = BnTicket{nullptr, 0, 0}, float* s1_c = nullptr,
                                                             float* s2_c = nullptr) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    float mean, rstd, g_, b_;
    plane_coeffs(a, plane, mean, rstd, g_, b_);
    const T* xp = (const T*)a.x + plane * (size_t)a.HW;
    const T* dp = (const T*)a.dy + plane * (size_t)a.HW;
    const T* yp = (a.act != PCGAN_ACT_NONE) ? (const T*)a.y + plane * (size_t)a.HW : nullptr;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
        float g = ld1(dp + i);
        if (yp) g *= act_grad_from_out(ld1(yp + i), a.act, a.slope);
        s1 += g;
        s2 += g * ((ld1(xp + i) - mean) * rstd);
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        if (tk.ticket != nullptr) {
            st_wt((float*)a.out + plane, s1);
            st_wt((float*)a.out2 + plane, s2);
        } else {
            ((float*)a.out)[plane] = s1;
            ((float*)a.out2)[plane] = s2;
        }
    }
    if (tk.ticket != nullptr) {      // batch norm: the last arriver sums the channel's N plane sums (bn_bwd_reduce_kernel)
        __shared__ int last_flag;
        const int c = (int)(plane % (size_t)tk.C);
        if (bn_last_arriver(tk, c, &last_flag) && threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const float* s1_nc = (const float*)a.out;
            const float* s2_nc = (const float*)a.out2;
            float u = 0.f, v = 0.f;
            for (int n = lane; n < tk.N; n += 64) {
                u += ld_wt(s1_nc + n * tk.C + c);
                v += ld_wt(s2_nc + n * tk.C + c);
            }
            u = wave_sum(u);
            v = wave_sum(v);
            if (lane == 0) {
                s1_c[c] = u;
                s2_c[c] = v;
            }
        }
    }
}

__global__ void bn_bwd_reduce_kernel(const float* __restrict__ s1_nc, const float* __restrict__ s2_nc,
                                     float* __restrict__ s1_c, float* __restrict__ s2_c, int N, int C) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    float a = 0.f, b = 0.f;
    for (int n = lane; n < N; n += 64) {
        a += s1_nc[n * C + c];
        b += s2_nc[n * C + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        s1_c[c] = a;
        s2_c[c] = b;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) norm_bwd_apply_kernel(NormArgs a) {
    const size_t plane = blockIdx.x;
    float mean, rstd, g_, b_;
    plane_coeffs(a, plane, mean, rstd, g_, b_);
    const int c = (int)(plane % a.C);
    const size_t si = a.per_plane ? plane : (size_t)c;
    const float m1 = a.s1[si] * a.inv_cnt, m2 = a.s2[si] * a.inv_cnt;
    const float k = rstd * g_;
    const T* xp = (const T*)a.x + plane * (size_t)a.HW;
    const T* dp = (const T*)a.dy + plane * (size_t)a.HW;
    const T* yp = (a.act != PCGAN_ACT_NONE) ? (const T*)a.y + plane * (size_t)a.HW : nullptr;
    T* op = (T*)a.out + plane * (size_t)a.HW;
    T* rp = a.out2 ? (T*)a.out2 + plane * (size_t)a.HW : nullptr;
    float am = 0.f;
    for (int i = threadIdx.x; i < a.HW; i += blockDim.x) {
        float g = ld1(dp + i);
        if (yp) g *= act_grad_from_out(ld1(yp + i), a.act, a.slope);
        const float xh = (ld1(xp + i) - mean) * rstd;
        const float o = k * (g - m1 - xh * m2);
        st1(op + i, o);
        am = fmaxf(am, fabsf(o));
        if (rp) st1(rp + i, g);
    }
    if (a.pmax) {
        __shared__ float scratch[16];
        am = block_max(am, scratch);
        if (threadIdx.x == 0) a.pmax[plane] = am;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused BatchNorm for small tensors (N * HW <= 8192 per channel: the Elo encoder's 14x14 / 7x7 maps, the PatchGAN's
// 16x16 / 15x15): ONE workgroup per channel does statistics (exact two-pass over the channel's N planes, which stay in
// L2), the running-statistics update and normalise + affine (+ residual) + activation -- one launch instead of
// plane_stats + bn_merge + norm_act (+ the batch counter); these layers are launch-latency bound, not bandwidth bound.
// Backward likewise: sum g, sum g * xhat, then dx (and the raw masked gradient for a residual branch).
// ------------------------------------------------------------------------------------------------
// element access of the fused BatchNorm kernels: the 4 waves take the N planes of channel c round-robin, lanes stride
// over a plane with 128-bit accesses when HW % 4 == 0 (V = 4) -- f(index, value...) is called once per element
template <int V, typename F>
__device__ __forceinline__ void bn_for_each(int N, int C, int HW, int c, F f) {
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    for (int n = wv; n < N; n += 4) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int k = ln * V; k < HW; k += 64 * V) f(base + k);
    }
}

// V consecutive elements as fp32 (V = 4: one 16-byte / 8-byte access, V = 1: scalar)
template <int V, typename T>
__device__ __forceinline__ void ldv(const T* p, float (&v)[V]) {
    if (V == 4) {
        const float4 t = ld4(p);
        v[0] = t.x; v[1 % V] = t.y; v[2 % V] = t.z; v[3 % V] = t.w;
    } else {
        v[0] = ld1(p);
    }
}
template <int V, typename T>
__device__ __forceinline__ void stv(T* p, const float (&v)[V]) {
    if (V == 4) st4(p, make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]));
    else st1(p, v[0]);
}

template <int V, typename T>
__global__ void __launch_bounds__(256) bn_fwd_fused_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ res,
                                                           T* __restrict__ y, float* __restrict__ mean_c,
                                                           float* __restrict__ var_c, float* running_mean, float* running_var,
                                                           long long* batches, int N, int C, int HW, float momentum, float eps,
                                                           int act, float slope, float* __restrict__ y_cmax) {
    __shared__ float scratch[16];
    const int c = blockIdx.x;
    const int cnt = N * HW;
    float s = 0.f;
    bn_for_each<V>(N, C, HW, c, [&](size_t i) {
        float v[V];
        ldv<V>(x + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) s += v[e];
    });
    const float mean = block_sum(s, scratch) / (float)cnt;
    float q = 0.f;
    bn_for_each<V>(N, C, HW, c, [&](size_t i) {
        float v[V];
        ldv<V>(x + i, v);
#pragma unroll
        for (int e = 0; e < V; ++e) q += (v[e] - mean) * (v[e] - mean);
    });
    const float m2 = block_sum(q, scratch);
    const float var = m2 / (float)cnt;
    if (threadIdx.x == 0) {
        mean_c[c] = mean;
        var_c[c] = var;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / ((float)cnt - 1.f));
        if (batches && c == 0) batches[0] += 1;
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = rsqrtf(var + eps) * g, sh = b - mean * sc;
    float am = 0.f;
    bn_for_each<V>(N, C, HW, c, [&](size_t i) {
        float v[V], r[V];
        ldv<V>(x + i, v);
        if (res) ldv<V>(res + i, r);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = v[e] * sc + sh;
            if (res) t += r[e];
            v[e] = act_apply(t, act, slope);
            am = fmaxf(am, fabsf(v[e]));
        }
        stv<V>(y + i, v);
    });
    if (y_cmax) {      // largest magnitude of this channel of y (partial maxima for the fp16 route of the next convolution)
        am = block_max(am, scratch);
        if (threadIdx.x == 0) y_cmax[c] = am;
    }
}

template <int V, typename T>
__global__ void __launch_bounds__(256) bn_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const T* __restrict__ y, const float* __restrict__ mean_c,
                                                           const float* __restrict__ var_c, const float* __restrict__ gamma,
                                                           T* __restrict__ dx, T* __restrict__ dres,
                                                           float* __restrict__ s1_c, float* __restrict__ s2_c, int N, int C,
                                                           int HW, float eps, int act, float slope, float* __restrict__ dx_cmax) {
    __shared__ float scratch[16];
    const int c = blockIdx.x;
    const int cnt = N * HW;
    const float mean = mean_c[c], rstd = rsqrtf(var_c[c] + eps);
    float s1 = 0.f, s2 = 0.f;
    bn_for_each<V>(N, C, HW, c, [&](size_t i) {
        float d[V], xv[V], yv[V];
        ldv<V>(dy + i, d);
        ldv<V>(x + i, xv);
        if (act != PCGAN_ACT_NONE) ldv<V>(y + i, yv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float g = d[e];
            if (act != PCGAN_ACT_NONE) g *= act_grad_from_out(yv[e], act, slope);
            s1 += g;
            s2 += g * ((xv[e] - mean) * rstd);
        }
    });
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
        s1_c[c] = s1;
        s2_c[c] = s2;
    }
    if (!dx && !dres) return;
    const float m1 = s1 / (float)cnt, m2 = s2 / (float)cnt;
    const float kk = rstd * (gamma ? gamma[c] : 1.f);
    float am = 0.f;
    bn_for_each<V>(N, C, HW, c, [&](size_t i) {
        float d[V], xv[V], yv[V], o[V], gr[V];
        ldv<V>(dy + i, d);
        ldv<V>(x + i, xv);
        if (act != PCGAN_ACT_NONE) ldv<V>(y + i, yv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float g = d[e];
            if (act != PCGAN_ACT_NONE) g *= act_grad_from_out(yv[e], act, slope);
            gr[e] = g;
            o[e] = kk * (g - m1 - ((xv[e] - mean) * rstd) * m2);
            am = fmaxf(am, fabsf(o[e]));
        }
        if (dx) stv<V>(dx + i, o);
        if (dres) stv<V>(dres + i, gr);
    });
    if (dx && dx_cmax) {
        am = block_max(am, scratch);
        if (threadIdx.x == 0) dx_cmax[c] = am;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused instance norm: the whole (n,c) plane lives in the workgroup's registers, so forward is ONE read
// + ONE write (statistics, normalise, residual, activation) and backward ONE read of dy/x(/y) + ONE write.
// T threads x E float4 per thread cover planes up to T*E*4 elements (HW % 4 == 0).
// ------------------------------------------------------------------------------------------------
template <int E, typename T>
__global__ void __launch_bounds__(1024) instnorm_fwd_fused_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                   T* __restrict__ y, float* __restrict__ mean_nc,
                                                                   float* __restrict__ m2_nc, int HW, float eps, int act,
                                                                   float slope, float* __restrict__ y_pmax) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    const int NT = blockDim.x, n4 = HW >> 2;
    const T* xp = x + plane * (size_t)HW;
    float4 v[E];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = threadIdx.x + k * NT;
        v[k] = i < n4 ? ld4(xp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = block_sum(s, scratch) / (float)HW;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if (threadIdx.x + k * NT < n4) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float m2 = block_sum(q, scratch);
    const float rstd = rsqrtf(m2 / (float)HW + eps);
    const float sh = -mean * rstd;
    const T* rp = res ? res + plane * (size_t)HW : nullptr;
    T* yp = y + plane * (size_t)HW;
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < n4) {
            float4 o = make_float4(v[k].x * rstd + sh, v[k].y * rstd + sh, v[k].z * rstd + sh, v[k].w * rstd + sh);
            if (rp) {
                const float4 r = ld4(rp + 4 * i);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            o.x = act_apply(o.x, act, slope); o.y = act_apply(o.y, act, slope);
            o.z = act_apply(o.z, act, slope); o.w = act_apply(o.w, act, slope);
            st4(yp + 4 * i, o);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (threadIdx.x == 0) {
        mean_nc[plane] = mean;
        m2_nc[plane] = m2;
    }
    // largest magnitude of y over this plane: the fp16 route of the convolution that reads y scales by the largest of them
    // (bf16x6_conv.hip; a plain store per plane -- one atomic maximum for the tensor serialises 8192 workgroups on one address)
    if (y_pmax) {
        am = block_max(am, scratch);
        if (threadIdx.x == 0) y_pmax[plane] = am;
    }
}

template <int E, typename T>
__global__ void __launch_bounds__(1024) instnorm_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const T* __restrict__ y, const float* __restrict__ mean_nc,
                                                                   const float* __restrict__ m2_nc, T* __restrict__ dx,
                                                                   float* __restrict__ dx_psum, int HW, float eps, int act,
                                                                   float slope, float* __restrict__ dx_pmax) {
    __shared__ float scratch[16];
    const size_t plane = blockIdx.x;
    const int NT = blockDim.x, n4 = HW >> 2;
    const float mean = mean_nc[plane];
    const float rstd = rsqrtf(m2_nc[plane] / (float)HW + eps);
    const T* dp = dy + plane * (size_t)HW;
    const T* xp = x + plane * (size_t)HW;
    const T* yp = (act != PCGAN_ACT_NONE) ? y + plane * (size_t)HW : nullptr;
    float4 g[E], xh[E];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < n4) {
            g[k] = ld4(dp + 4 * i);
            const float4 xv = ld4(xp + 4 * i);
            if (yp) {
                const float4 yv = ld4(yp + 4 * i);
                g[k].x *= act_grad_from_out(yv.x, act, slope); g[k].y *= act_grad_from_out(yv.y, act, slope);
                g[k].z *= act_grad_from_out(yv.z, act, slope); g[k].w *= act_grad_from_out(yv.w, act, slope);
            }
            xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
            s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
        } else {
            g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[k] = g[k];
        }
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    const float m1 = s1 / (float)HW, mm2 = s2 / (float)HW;
    T* op = dx + plane * (size_t)HW;
    float ps = 0.f, am = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int i = threadIdx.x + k * NT;
        if (i < n4) {
            const float4 o = make_float4(rstd * (g[k].x - m1 - xh[k].x * mm2), rstd * (g[k].y - m1 - xh[k].y * mm2),
                                         rstd * (g[k].z - m1 - xh[k].z * mm2), rstd * (g[k].w - m1 - xh[k].w * mm2));
            st4(op + 4 * i, o);
            ps += (o.x + o.y) + (o.z + o.w);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    // sum of dx over the plane, out of the registers that just stored it: the bias gradient of the convolution in front of
    // this norm is the sum of these over n (true value 0 -- the norm cancels the bias --, fp32 noise as in the reference);
    // saves the separate 33.5 MB read per layer that pcgan_channel_sum would need
    if (dx_psum) {
        ps = block_sum(ps, scratch);
        if (threadIdx.x == 0) dx_psum[plane] = ps;
    }
    if (dx_pmax) {     // largest magnitude of dx over this plane, as in the forward kernel
        am = block_max(am, scratch);
        if (threadIdx.x == 0) dx_pmax[plane] = am;
    }
}

// Planes of up to 1024 elements (the residual blocks' 32x32 maps: 8192 planes per launch): ONE WAVE per plane, four planes per workgroup.
// Every reduction is a wave reduction (no barrier, no LDS): the workgroup-per-plane form above spends its time in three barrier
// pairs per 4 KB plane.  Same arithmetic (exact two-pass mean / M2), other summation order.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256) instnorm_fwd_wave_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                                 float* __restrict__ mean_nc, float* __restrict__ m2_nc, int planes, int HW,
                                                                 float eps, int act, float slope, float* __restrict__ y_pmax) {
    const int lane = threadIdx.x & 63;
    const size_t plane = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= (size_t)planes) return;
    const int n4 = HW >> 2;
    const T* xp = x + plane * (size_t)HW;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * 64;
        v[k] = i < n4 ? ld4(xp + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const float mean = wave_sum(s) / (float)HW;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (lane + k * 64 < n4) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float m2 = wave_sum(q);
    const float rstd = rsqrtf(m2 / (float)HW + eps);
    const float sh = -mean * rstd;
    const T* rp = res ? res + plane * (size_t)HW : nullptr;
    T* yp = y + plane * (size_t)HW;
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * 64;
        if (i < n4) {
            float4 o = make_float4(v[k].x * rstd + sh, v[k].y * rstd + sh, v[k].z * rstd + sh, v[k].w * rstd + sh);
            if (rp) {
                const float4 r = ld4(rp + 4 * i);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            o.x = act_apply(o.x, act, slope); o.y = act_apply(o.y, act, slope);
            o.z = act_apply(o.z, act, slope); o.w = act_apply(o.w, act, slope);
            st4(yp + 4 * i, o);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (y_pmax) am = wave_max(am);
    if (lane == 0) {
        mean_nc[plane] = mean;
        m2_nc[plane] = m2;
        if (y_pmax) y_pmax[plane] = am;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) instnorm_bwd_wave_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                                 const float* __restrict__ mean_nc, const float* __restrict__ m2_nc,
                                                                 T* __restrict__ dx, float* __restrict__ dx_psum, int planes, int HW, float eps,
                                                                 int act, float slope, float* __restrict__ dx_pmax) {
    const int lane = threadIdx.x & 63;
    const size_t plane = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= (size_t)planes) return;
    const int n4 = HW >> 2;
    const float mean = mean_nc[plane];
    const float rstd = rsqrtf(m2_nc[plane] / (float)HW + eps);
    const T* dp = dy + plane * (size_t)HW;
    const T* xp = x + plane * (size_t)HW;
    const T* yp = (act != PCGAN_ACT_NONE) ? y + plane * (size_t)HW : nullptr;
    float4 g[4], xh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * 64;
        if (i < n4) {
            g[k] = ld4(dp + 4 * i);
            const float4 xv = ld4(xp + 4 * i);
            if (yp) {
                const float4 yv = ld4(yp + 4 * i);
                g[k].x *= act_grad_from_out(yv.x, act, slope); g[k].y *= act_grad_from_out(yv.y, act, slope);
                g[k].z *= act_grad_from_out(yv.z, act, slope); g[k].w *= act_grad_from_out(yv.w, act, slope);
            }
            xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
            s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
            s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
        } else {
            g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[k] = g[k];
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float m1 = s1 / (float)HW, mm2 = s2 / (float)HW;
    T* op = dx + plane * (size_t)HW;
    float ps = 0.f, am = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * 64;
        if (i < n4) {
            const float4 o = make_float4(rstd * (g[k].x - m1 - xh[k].x * mm2), rstd * (g[k].y - m1 - xh[k].y * mm2),
                                         rstd * (g[k].z - m1 - xh[k].z * mm2), rstd * (g[k].w - m1 - xh[k].w * mm2));
            st4(op + 4 * i, o);
            ps += (o.x + o.y) + (o.z + o.w);
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (dx_psum) ps = wave_sum(ps);
    if (dx_pmax) am = wave_max(am);
    if (lane == 0) {
        if (dx_psum) dx_psum[plane] = ps;
        if (dx_pmax) dx_pmax[plane] = am;
    }
}

// (threads, float4-per-thread) for a plane of HW elements; E == 0: plane too large / not a multiple of 4
static inline void fused_plan(int HW, int* T, int* E) {  // T: threads
    *T = 0;
    *E = 0;
    if (HW & 3) return;
    const int n4 = HW >> 2;
    if (n4 <= 256) { *T = n4 <= 64 ? 64 : (n4 <= 128 ? 128 : 256); *E = 1; }
    else if (n4 <= 1024) { *T = 256; *E = 4; }
    else if (n4 <= 4096) { *T = 1024; *E = 4; }
    else if (n4 <= 16384) { *T = 1024; *E = 16; }
}

static inline int plane_threads(int HW) { return HW >= 1024 ? 256 : (HW >= 256 ? 128 : 64); }

}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_plane_stats(const void* x, float* mean_nc, float* m2_nc, int NC, int HW, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && mean_nc && m2_nc && NC > 0 && HW > 0, "plane_stats: bad arguments");
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(plane_stats_kernel<T>, dim3(NC), dim3(plane_threads(HW)), 0, (hipStream_t)s,
                                                    (const T*)x, mean_nc, m2_nc, HW));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_stats_merged(const void* x, float* mean_nc, float* m2_nc, float* mean_c, float* var_c, float* running_mean,
                                     float* running_var, long long* batches, unsigned int* ticket, int N, int C, int HW, float momentum,
                                     int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && mean_nc && m2_nc && mean_c && var_c && ticket && N > 0 && C > 0 && HW > 0 && (long long)N * HW > 1, "bn_stats_merged: bad arguments");
    const BnTicket tk{ticket, N, C};
    const BnMergeOut mo{mean_c, var_c, running_mean, running_var, batches, momentum};
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(plane_stats_kernel<T>, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s,
                                                    (const T*)x, mean_nc, m2_nc, HW, tk, mo));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_bwd_stats_reduced(const void* dy, const void* x, const void* y, const float* mean_c, const float* var_c, float* s1_nc,
                                          float* s2_nc, float* s1_c, float* s2_c, unsigned int* ticket, int N, int C, int HW, float eps, int act,
                                          float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean_c && var_c && s1_nc && s2_nc && s1_c && s2_c && ticket && N > 0 && C > 0 && HW > 0, "bn_bwd_stats_reduced: bad arguments");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "bn_bwd_stats_reduced: activation mask needs y");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean_c; a.var = var_c; a.out = s1_nc; a.out2 = s2_nc;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = 0; a.eps = eps; a.act = act; a.slope = slope;
    const BnTicket tk{ticket, N, C};
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(norm_bwd_stats_kernel<T>, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a, tk, s1_c, s2_c));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_merge(const float* mean_nc, const float* m2_nc, float* mean_c, float* var_c,
                              float* running_mean, float* running_var, int N, int C, int HW, float momentum,
                              pcgan_stream_t s) {
    PCGAN_CHECK(mean_nc && m2_nc && N > 0 && C > 0 && HW > 0, "bn_merge: bad arguments");
    hipLaunchKernelGGL(bn_merge_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)s, mean_nc, m2_nc, mean_c,
                       var_c, running_mean, running_var, N, C, HW, momentum);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_in_running_update(const float* mean_nc, const float* m2_nc, float* running_mean,
                                       float* running_var, int N, int C, int HW, float momentum,
                                       pcgan_stream_t s) {
    PCGAN_CHECK(mean_nc && m2_nc && running_mean && running_var && N > 0 && C > 0 && HW > 1,
                "in_running_update: bad arguments");
    hipLaunchKernelGGL(in_running_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)s, mean_nc, m2_nc,
                       running_mean, running_var, N, C, HW, momentum);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_act_fwd(const void* x, const float* mean, const float* var, const float* gamma,
                                  const float* beta, const void* residual, void* y, float* y_pmax, int N, int C, int HW,
                                  int per_plane, float eps, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && mean && var && y && N > 0 && C > 0 && HW > 0, "norm_act_fwd: bad arguments");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.residual = residual; a.out = y; a.pmax = y_pmax;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(norm_act_fwd_kernel<T>, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_bwd_stats(const void* dy, const void* x, const void* y, const float* mean,
                                    const float* var, float* s1_nc, float* s2_nc, int N, int C, int HW,
                                    int per_plane, float eps, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean && var && s1_nc && s2_nc, "norm_bwd_stats: null pointer");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "norm_bwd_stats: activation mask needs y");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean; a.var = var; a.out = s1_nc; a.out2 = s2_nc;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(norm_bwd_stats_kernel<T>, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_bn_bwd_reduce(const float* s1_nc, const float* s2_nc, float* s1_c, float* s2_c, int N, int C,
                                   pcgan_stream_t s) {
    PCGAN_CHECK(s1_nc && s2_nc && s1_c && s2_c, "bn_bwd_reduce: null pointer");
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)s, s1_nc, s2_nc, s1_c,
                       s2_c, N, C);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_norm_bwd_apply(const void* dy, const void* x, const void* y, const float* mean,
                                    const float* var, const float* gamma, const float* s1, const float* s2,
                                    void* dx, void* d_residual, float* dx_pmax, int N, int C, int HW, int per_plane, float eps,
                                    int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean && var && s1 && s2 && dx, "norm_bwd_apply: null pointer");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "norm_bwd_apply: activation mask needs y");
    NormArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.dy = dy; a.mean = mean; a.var = var; a.gamma = gamma; a.s1 = s1; a.s2 = s2;
    a.out = dx; a.out2 = d_residual; a.pmax = dx_pmax;
    a.N = N; a.C = C; a.HW = HW; a.per_plane = per_plane; a.eps = eps; a.act = act; a.slope = slope;
    a.inv_cnt = 1.f / (per_plane ? (float)HW : (float)N * (float)HW);
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL(norm_bwd_apply_kernel<T>, dim3(N * C), dim3(plane_threads(HW)), 0, (hipStream_t)s, a));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static void launch_instnorm_fwd(int E, int NT, int planes, hipStream_t st, const void* x, const void* residual, void* y, float* mean_nc,
                                float* m2_nc, int HW, float eps, int act, float slope, float* amax) {
    const T* xp = (const T*)x;
    const T* rp = (const T*)residual;
    T* yp = (T*)y;
    if (E == 1 && planes >= 1024)      // small planes, many of them: one wave per plane
        hipLaunchKernelGGL((instnorm_fwd_wave_kernel<T>), dim3((planes + 3) / 4), dim3(256), 0, st, xp, rp, yp, mean_nc, m2_nc, planes, HW, eps, act, slope, amax);
    else if (E == 1) hipLaunchKernelGGL((instnorm_fwd_fused_kernel<1, T>), dim3(planes), dim3(NT), 0, st, xp, rp, yp, mean_nc, m2_nc, HW, eps, act, slope, amax);
    else if (E == 4) hipLaunchKernelGGL((instnorm_fwd_fused_kernel<4, T>), dim3(planes), dim3(NT), 0, st, xp, rp, yp, mean_nc, m2_nc, HW, eps, act, slope, amax);
    else hipLaunchKernelGGL((instnorm_fwd_fused_kernel<16, T>), dim3(planes), dim3(NT), 0, st, xp, rp, yp, mean_nc, m2_nc, HW, eps, act, slope, amax);
}

extern "C" int pcgan_instnorm_fwd(const void* x, const void* residual, void* y, float* mean_nc, float* m2_nc, float* y_pmax, int N,
                                  int C, int HW, float eps, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(x && y && mean_nc && m2_nc && N > 0 && C > 0 && HW > 0, "instnorm_fwd: bad arguments");
    int NT, E;
    fused_plan(HW, &NT, &E);
    hipStream_t st = (hipStream_t)s;
    PCGAN_CHECK(!y_pmax || E != 0, "instnorm_fwd: plane maxima come out of the register-resident kernel only (pcgan_instnorm_fused)");
    if (E == 0) {  // plane does not fit the register-resident kernel: statistics pass + apply pass
        if (pcgan_plane_stats(x, mean_nc, m2_nc, N * C, HW, dtype, s)) return 1;
        return pcgan_norm_act_fwd(x, mean_nc, m2_nc, nullptr, nullptr, residual, y, nullptr, N, C, HW, 1, eps, act, slope, dtype, s);
    }
    PCGAN_DTYPE_SWITCH(dtype, T, launch_instnorm_fwd<T>(E, NT, N * C, st, x, residual, y, mean_nc, m2_nc, HW, eps, act, slope, y_pmax));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_instnorm_fused(int HW) {
    int NT, E;
    fused_plan(HW, &NT, &E);
    return E != 0;
}

template <typename T>
static void launch_instnorm_bwd(int E, int NT, int planes, hipStream_t st, const void* dy, const void* x, const void* y, const float* mean_nc,
                                const float* m2_nc, void* dx, float* dx_psum, int HW, float eps, int act, float slope, float* amax) {
    const T* dp = (const T*)dy;
    const T* xp = (const T*)x;
    const T* yp = (const T*)y;
    T* op = (T*)dx;
    if (E == 1 && planes >= 1024)
        hipLaunchKernelGGL((instnorm_bwd_wave_kernel<T>), dim3((planes + 3) / 4), dim3(256), 0, st, dp, xp, yp, mean_nc, m2_nc, op, dx_psum, planes, HW, eps, act, slope, amax);
    else if (E == 1) hipLaunchKernelGGL((instnorm_bwd_fused_kernel<1, T>), dim3(planes), dim3(NT), 0, st, dp, xp, yp, mean_nc, m2_nc, op, dx_psum, HW, eps, act, slope, amax);
    else if (E == 4) hipLaunchKernelGGL((instnorm_bwd_fused_kernel<4, T>), dim3(planes), dim3(NT), 0, st, dp, xp, yp, mean_nc, m2_nc, op, dx_psum, HW, eps, act, slope, amax);
    else hipLaunchKernelGGL((instnorm_bwd_fused_kernel<16, T>), dim3(planes), dim3(NT), 0, st, dp, xp, yp, mean_nc, m2_nc, op, dx_psum, HW, eps, act, slope, amax);
}

extern "C" int pcgan_instnorm_bwd(const void* dy, const void* x, const void* y, const float* mean_nc,
                                  const float* m2_nc, void* dx, float* dx_psum, float* dx_pmax, float* ws_s1s2, int N, int C, int HW,
                                  float eps, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean_nc && m2_nc && dx, "instnorm_bwd: null pointer");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "instnorm_bwd: activation mask needs y");
    int NT, E;
    fused_plan(HW, &NT, &E);
    hipStream_t st = (hipStream_t)s;
    PCGAN_CHECK(E != 0 || !dx_psum, "instnorm_bwd: plane sums of dx come out of the register-resident kernel only (pcgan_instnorm_fused)");
    PCGAN_CHECK(E != 0 || !dx_pmax, "instnorm_bwd: plane maxima come out of the register-resident kernel only (pcgan_instnorm_fused)");
    if (E == 0) {
        PCGAN_CHECK(ws_s1s2, "instnorm_bwd: the two-pass fallback needs 2*N*C floats of workspace");
        if (pcgan_norm_bwd_stats(dy, x, y, mean_nc, m2_nc, ws_s1s2, ws_s1s2 + (size_t)N * C, N, C, HW, 1, eps, act, slope, dtype, s))
            return 1;
        return pcgan_norm_bwd_apply(dy, x, y, mean_nc, m2_nc, nullptr, ws_s1s2, ws_s1s2 + (size_t)N * C, dx, nullptr, nullptr, N, C,
                                    HW, 1, eps, act, slope, dtype, s);
    }
    PCGAN_DTYPE_SWITCH(dtype, T, launch_instnorm_bwd<T>(E, NT, N * C, st, dy, x, y, mean_nc, m2_nc, dx, dx_psum, HW, eps, act, slope, dx_pmax));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static void launch_bn_fwd(bool v4, hipStream_t st, const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                          float* mean_c, float* var_c, float* running_mean, float* running_var, long long* batches, int N, int C, int HW,
                          float momentum, float eps, int act, float slope, float* y_cmax) {
    if (v4)
        hipLaunchKernelGGL((bn_fwd_fused_kernel<4, T>), dim3(C), dim3(256), 0, st, (const T*)x, gamma, beta, (const T*)residual, (T*)y, mean_c,
                           var_c, running_mean, running_var, batches, N, C, HW, momentum, eps, act, slope, y_cmax);
    else
        hipLaunchKernelGGL((bn_fwd_fused_kernel<1, T>), dim3(C), dim3(256), 0, st, (const T*)x, gamma, beta, (const T*)residual, (T*)y, mean_c,
                           var_c, running_mean, running_var, batches, N, C, HW, momentum, eps, act, slope, y_cmax);
}

extern "C" int pcgan_bn_fwd_fused(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                                  float* mean_c, float* var_c, float* running_mean, float* running_var, long long* batches,
                                  float* y_cmax, int N, int C, int HW, float momentum, float eps, int act, float slope, int dtype,
                                  pcgan_stream_t s) {
    PCGAN_CHECK(x && y && mean_c && var_c && N > 0 && C > 0 && HW > 0 && (long long)N * HW > 1, "bn_fwd_fused: bad arguments");
    PCGAN_DTYPE_SWITCH(dtype, T, launch_bn_fwd<T>((HW & 3) == 0, (hipStream_t)s, x, gamma, beta, residual, y, mean_c, var_c, running_mean,
                                                  running_var, batches, N, C, HW, momentum, eps, act, slope, y_cmax));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static void launch_bn_bwd(bool v4, hipStream_t st, const void* dy, const void* x, const void* y, const float* mean_c, const float* var_c,
                          const float* gamma, void* dx, void* dres, float* s1_c, float* s2_c, int N, int C, int HW, float eps, int act, float slope,
                          float* dx_cmax) {
    if (v4)
        hipLaunchKernelGGL((bn_bwd_fused_kernel<4, T>), dim3(C), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y, mean_c, var_c, gamma,
                           (T*)dx, (T*)dres, s1_c, s2_c, N, C, HW, eps, act, slope, dx_cmax);
    else
        hipLaunchKernelGGL((bn_bwd_fused_kernel<1, T>), dim3(C), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y, mean_c, var_c, gamma,
                           (T*)dx, (T*)dres, s1_c, s2_c, N, C, HW, eps, act, slope, dx_cmax);
}

extern "C" int pcgan_bn_bwd_fused(const void* dy, const void* x, const void* y, const float* mean_c, const float* var_c,
                                  const float* gamma, void* dx, void* dres, float* s1_c, float* s2_c, float* dx_cmax, int N, int C, int HW,
                                  float eps, int act, float slope, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(dy && x && mean_c && var_c && s1_c && s2_c && N > 0 && C > 0 && HW > 0, "bn_bwd_fused: bad arguments");
    PCGAN_CHECK(act == PCGAN_ACT_NONE || y, "bn_bwd_fused: activation mask needs y");
    PCGAN_DTYPE_SWITCH(dtype, T, launch_bn_bwd<T>((HW & 3) == 0, (hipStream_t)s, dy, x, y, mean_c, var_c, gamma, dx, dres, s1_c, s2_c, N, C, HW,
                                                  eps, act, slope, dx_cmax));
    PCGAN_LAUNCH_CHECK();
    return 0;
}
