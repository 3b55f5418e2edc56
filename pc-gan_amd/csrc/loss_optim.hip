// Fused loss + loss-gradient reductions and the fused Adam update (HBM-bound).
//
// Reference: GANLoss / nn.BCELoss  models/networks.py:386-420,
//            nn.L1Loss / nn.MSELoss models/wsgan_emb_model.py:141-149,
//            torch.optim.Adam      models/wsgan_emb_model.py:153-156.
// Reductions are two-stage (per-workgroup partial -> one finishing workgroup), so the
// result is bitwise reproducible run to run.
#include "common.h"

namespace pcgan {

enum { LOSS_BCE = 0, LOSS_L1 = 1, LOSS_MSE = 2 };
static constexpr int LOSS_BLOCKS = 256;

// a (prediction / image) and grad are activation tensors of storage type T; the BCE target is a per-sample fp32 vector, the
// L1 / MSE target b an activation tensor; partial sums and the loss are fp32
template <int KIND, typename T>
__global__ void __launch_bounds__(256) loss_partial_kernel(const T* __restrict__ a, const void* __restrict__ bv,
                                                           T* __restrict__ grad, float* __restrict__ part, size_t n,
                                                           int per_n, float gs) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float p = ld1(a + i);
        if (KIND == LOSS_BCE) {
            const float t = ((const float*)bv)[i / per_n];
            float lp = logf(p), lq = logf(1.f - p);
            lp = lp < -100.f ? -100.f : lp;
            lq = lq < -100.f ? -100.f : lq;
            acc -= t * lp + (1.f - t) * lq;
            if (grad) {
                float den = (1.f - p) * p;
                den = den < 1e-12f ? 1e-12f : den;
                st1(grad + i, gs * (p - t) / den);
            }
        } else if (KIND == LOSS_L1) {
            const float d = p - ld1((const T*)bv + i);
            acc += fabsf(d);
            if (grad) st1(grad + i, d > 0.f ? gs : (d < 0.f ? -gs : 0.f));
        } else {
            const float d = p - ld1((const T*)bv + i);
            acc += d * d;
            if (grad) st1(grad + i, 2.f * gs * d);
        }
    }
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) loss_finish_kernel(const float* __restrict__ part, int nparts, float inv_n,
                                                          float* __restrict__ loss) {
    __shared__ float scratch[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
    acc = block_sum(acc, scratch);
    if (threadIdx.x == 0) loss[0] = acc * inv_n;
}

template <int KIND>
static int run_loss(const void* a, const void* b, float* loss, void* grad, size_t n, int per_n, float gscale,
                    void* ws, size_t ws_bytes, int dtype, hipStream_t st) {
    PCGAN_CHECK(a && b && loss && n > 0, "loss: bad arguments");
    PCGAN_CHECK(ws && ws_bytes >= LOSS_BLOCKS * sizeof(float), "loss: workspace too small");
    int blocks = (int)((n + 255) / 256);
    if (blocks > LOSS_BLOCKS) blocks = LOSS_BLOCKS;
    const float gs = gscale / (float)n;
    PCGAN_DTYPE_SWITCH(dtype, T, hipLaunchKernelGGL((loss_partial_kernel<KIND, T>), dim3(blocks), dim3(256), 0, st, (const T*)a, b, (T*)grad,
                                                    (float*)ws, n, per_n, gs));
    PCGAN_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, 1.f / (float)n, loss);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)   with m, v updated first (torch.optim.Adam)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps,
                            const float* __restrict__ lr_dev, const int* __restrict__ step_dev, int step) {
    if (lr_dev) lr = lr_dev[0];
    if (step_dev) step = step_dev[0];
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - beta1);  // lerp form used by torch
        vi = vi * beta2 + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}
__global__ void incr_kernel(int* c) { c[0] += 1; }

}  // namespace pcgan

using namespace pcgan;

extern "C" size_t pcgan_loss_workspace_bytes(size_t n) {
    (void)n;
    return LOSS_BLOCKS * sizeof(float);
}

extern "C" int pcgan_bce_loss(const void* pred, const float* target_n, float* loss, void* grad, int N, int per_n,
                              float gscale, void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s) {
    PCGAN_CHECK(N > 0 && per_n > 0, "bce_loss: bad shape");
    return run_loss<LOSS_BCE>(pred, target_n, loss, grad, (size_t)N * per_n, per_n, gscale, ws, ws_bytes, dtype, (hipStream_t)s);
}
extern "C" int pcgan_l1_loss(const void* a, const void* b, float* loss, void* grad_a, size_t n, float gscale,
                             void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s) {
    return run_loss<LOSS_L1>(a, b, loss, grad_a, n, 1, gscale, ws, ws_bytes, dtype, (hipStream_t)s);
}
extern "C" int pcgan_mse_loss(const void* a, const void* b, float* loss, void* grad_a, size_t n, float gscale,
                              void* ws, size_t ws_bytes, int dtype, pcgan_stream_t s) {
    return run_loss<LOSS_MSE>(a, b, loss, grad_a, n, 1, gscale, ws, ws_bytes, dtype, (hipStream_t)s);
}

extern "C" int pcgan_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, int step, pcgan_stream_t s) {
    PCGAN_CHECK(param && grad && exp_avg && exp_avg_sq && step >= 1, "adam_step: bad arguments");
    if (n == 0) return 0;
    size_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, (const float*)nullptr, (const int*)nullptr, step);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// graph-capturable variant: learning rate and step counter live in device memory; the
// counter is incremented on the stream before the update reads it.
extern "C" int pcgan_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                   const float* lr_dev, int* step_dev, float beta1, float beta2, float eps,
                                   pcgan_stream_t s) {
    PCGAN_CHECK(param && grad && exp_avg && exp_avg_sq && lr_dev && step_dev, "adam_step_dev: bad arguments");
    hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, step_dev);
    PCGAN_LAUNCH_CHECK();
    if (n == 0) return 0;
    size_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, n,
                       0.f, beta1, beta2, eps, lr_dev, (const int*)step_dev, 0);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
