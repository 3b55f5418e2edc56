// Composite entry points: a whole fused site of the generator per host call (include/pcgan_hip.h, "composite").
//
// The per-op C-ABI costs the host one Python -> ctypes round trip per launch; at config 2 the 18 ResnetBlocks x 2 generator passes are
// ~290 such calls per step.  pcgan_resblock_fwd / pcgan_resblock_bwd issue exactly the launches the per-op sequence issues -- same
// kernels, same arguments, same order on the same streams -- so their results are bit-identical to it (tests/test_gpu_composite.py);
// the only arithmetic that moves is the skip connection's `grad +=`, which becomes the epilogue of the first convolution's data
// gradient (one fp32 add either way).
#include "common.h"

namespace {
inline pcgan_conv_desc conv_of(const pcgan_resblock_desc* d) {
    pcgan_conv_desc c;
    c.N = d->N; c.C = d->C; c.H = d->H; c.W = d->W; c.K = d->C; c.R = 3; c.S = 3;
    c.stride = 1; c.pad = 1; c.pad_mode = 1; c.P = d->H; c.Q = d->W; c.dtype = d->dtype;
    return c;
}
inline bool half_of(const pcgan_resblock_desc* d) { return d->dtype == PCGAN_BF16; }
}  // namespace

extern "C" int pcgan_event_create(pcgan_event_t* ev) {
    PCGAN_CHECK(ev, "event_create: null pointer");
    hipEvent_t e;
    hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    PCGAN_CHECK(r == hipSuccess, "event_create: %s", hipGetErrorString(r));
    *ev = (pcgan_event_t)e;
    return 0;
}

extern "C" int pcgan_event_destroy(pcgan_event_t ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
    return 0;
}

extern "C" int pcgan_resblock_supported(const pcgan_resblock_desc* d) {
    if (!d || d->N < 1 || d->C < 1 || d->H < 3 || d->W < 3 || (d->dtype != PCGAN_F32 && d->dtype != PCGAN_BF16)) return 0;
    const pcgan_conv_desc c = conv_of(d);
    if (half_of(d))      // bf16 tensors: the one-product forms of the same kernels (no operand maxima, no scaling)
        return pcgan_conv2d_bsplit_supported(&c) && pcgan_conv2d_bsplit_dgrad_supported(&c) && pcgan_conv2d_hsplit_wgrad_supported(&c) &&
               pcgan_instnorm_fused(d->H * d->W);
    return pcgan_conv2d_hsplit_supported(&c, PCGAN_PASS_FWD) && pcgan_conv2d_hsplit_supported(&c, PCGAN_PASS_BWD_DATA) &&
           pcgan_conv2d_hsplit_wgrad_supported(&c) && pcgan_instnorm_fused(d->H * d->W);
}

extern "C" size_t pcgan_resblock_wgrad_workspace_bytes(const pcgan_resblock_desc* d) {
    if (!pcgan_resblock_supported(d)) return 0;
    const pcgan_conv_desc c = conv_of(d);
    return pcgan_conv2d_hsplit_wgrad_workspace_bytes(&c);
}

#define STEP(call)            \
    do {                      \
        const int r__ = (call); \
        if (r__) return r__;  \
    } while (0)

extern "C" int pcgan_resblock_fwd(const pcgan_resblock_desc* d, const void* x, const float* x_amax, int n_xamax, const void* pk1,
                                  const float* b1, const void* pk2, const float* b2, float* rm1, float* rv1, float* rm2, float* rv2,
                                  void* y1, void* h, void* y2, void* out, float* stats, float* amax, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_resblock_supported(d), "resblock_fwd: unsupported shape");
    const bool half = half_of(d);
    PCGAN_CHECK(x && (half || (x_amax && n_xamax > 0)) && pk1 && pk2 && y1 && h && y2 && out && stats && (half || amax), "resblock_fwd: null pointer");
    const pcgan_conv_desc c = conv_of(d);
    const int N = d->N, C = d->C, HW = d->H * d->W, NC = N * C;
    float *mean1 = stats, *m21 = stats + NC, *mean2 = stats + 2 * (size_t)NC, *m22 = stats + 3 * (size_t)NC;
    if (half) {      // bf16 tensors (pk1 / pk2: pcgan_conv2d_bsplit_pack): same sequence, no operand maxima
        STEP(pcgan_conv2d_fwd_bsplit(&c, x, pk1, b1, y1, PCGAN_ACT_NONE, 0.f, s));
        STEP(pcgan_instnorm_fwd(y1, nullptr, h, mean1, m21, nullptr, N, C, HW, d->eps, PCGAN_ACT_RELU, 0.f, PCGAN_BF16, s));
        if (rm1 && rv1) STEP(pcgan_in_running_update(mean1, m21, rm1, rv1, N, C, HW, d->momentum, s));
        STEP(pcgan_conv2d_fwd_bsplit(&c, h, pk2, b2, y2, PCGAN_ACT_NONE, 0.f, s));
        STEP(pcgan_instnorm_fwd(y2, x, out, mean2, m22, nullptr, N, C, HW, d->eps, PCGAN_ACT_NONE, 0.f, PCGAN_BF16, s));
        if (rm2 && rv2) STEP(pcgan_in_running_update(mean2, m22, rm2, rv2, N, C, HW, d->momentum, s));
        return 0;
    }
    float *h_amax = amax, *o_amax = amax + NC;
    STEP(pcgan_conv2d_fwd_hsplit(&c, x, x_amax, n_xamax, pk1, b1, y1, PCGAN_ACT_NONE, 0.f, s));
    STEP(pcgan_instnorm_fwd(y1, nullptr, h, mean1, m21, h_amax, N, C, HW, d->eps, PCGAN_ACT_RELU, 0.f, PCGAN_F32, s));
    if (rm1 && rv1) STEP(pcgan_in_running_update(mean1, m21, rm1, rv1, N, C, HW, d->momentum, s));
    STEP(pcgan_conv2d_fwd_hsplit(&c, h, h_amax, NC, pk2, b2, y2, PCGAN_ACT_NONE, 0.f, s));
    STEP(pcgan_instnorm_fwd(y2, x, out, mean2, m22, o_amax, N, C, HW, d->eps, PCGAN_ACT_NONE, 0.f, PCGAN_F32, s));
    if (rm2 && rv2) STEP(pcgan_in_running_update(mean2, m22, rm2, rv2, N, C, HW, d->momentum, s));
    return 0;
}

extern "C" int pcgan_resblock_bwd(const pcgan_resblock_desc* d, const void* dout, const void* x, const float* x_amax, int n_xamax,
                                  const void* y1, const void* h, const float* h_amax, const void* y2, const float* stats, const void* pk1b,
                                  const void* pk2b, float* dw1, float* db1, float* dw2, float* db2, void* dy2, void* dh, void* dy1, void* dx,
                                  float* scratch, void* wgrad_ws, size_t wgrad_ws_bytes, pcgan_stream_t s, pcgan_stream_t side,
                                  pcgan_event_t fork_event) {
    PCGAN_CHECK(pcgan_resblock_supported(d), "resblock_bwd: unsupported shape");
    const bool half = half_of(d);
    PCGAN_CHECK(dout && x && (half || (x_amax && n_xamax > 0 && h_amax)) && y1 && h && y2 && stats && pk1b && pk2b && dw1 && dw2 && dy2 && dh && dy1 &&
                    dx && scratch && wgrad_ws && fork_event,
                "resblock_bwd: null pointer");
    PCGAN_CHECK(side && side != s, "resblock_bwd: the parameter-gradient stream must be a second stream");
    const pcgan_conv_desc c = conv_of(d);
    const int N = d->N, C = d->C, HW = d->H * d->W, NC = N * C;
    const float *mean1 = stats, *m21 = stats + NC, *mean2 = stats + 2 * (size_t)NC, *m22 = stats + 3 * (size_t)NC;
    float *psum = scratch, *dy2_amax = scratch + NC, *dy1_amax = scratch + 2 * (size_t)NC, *psum1 = scratch + 3 * (size_t)NC;
    hipStream_t ms = (hipStream_t)s, ss = (hipStream_t)side;
    hipEvent_t ev = (hipEvent_t)fork_event;
    if (half) {      // bf16 tensors (pk1b / pk2b: pcgan_conv2d_bsplit_dgrad_pack): the same sequence; the skip connection's gradient is
                     // added by pcgan_add after the data gradient (the per-op path's autograd sum: one bf16 rounding of the fp32 sum
                     // either way), into dx from dh, which is free by then
        STEP(pcgan_instnorm_bwd(dout, y2, nullptr, mean2, m22, dy2, psum, nullptr, nullptr, N, C, HW, d->eps, PCGAN_ACT_NONE, 0.f, PCGAN_BF16, s));
        PCGAN_CHECK(hipEventRecord(ev, ms) == hipSuccess && hipStreamWaitEvent(ss, ev, 0) == hipSuccess, "resblock_bwd: stream fork failed");
        STEP(pcgan_conv2d_bwd_weight_hsplit(&c, h, nullptr, 0, dy2, nullptr, 0, dw2, 1, wgrad_ws, wgrad_ws_bytes, side));
        if (db2) STEP(pcgan_sum_planes(psum, db2, N, C, 1, side));
        STEP(pcgan_conv2d_bwd_data_bsplit(&c, dy2, pk2b, dh, s));
        STEP(pcgan_instnorm_bwd(dh, y1, h, mean1, m21, dy1, psum1, nullptr, nullptr, N, C, HW, d->eps, PCGAN_ACT_RELU, 0.f, PCGAN_BF16, s));
        PCGAN_CHECK(hipEventRecord(ev, ms) == hipSuccess && hipStreamWaitEvent(ss, ev, 0) == hipSuccess, "resblock_bwd: stream fork failed");
        STEP(pcgan_conv2d_bwd_weight_hsplit(&c, x, nullptr, 0, dy1, nullptr, 0, dw1, 1, wgrad_ws, wgrad_ws_bytes, side));
        if (db1) STEP(pcgan_sum_planes(psum1, db1, N, C, 1, side));
        STEP(pcgan_conv2d_bwd_data_bsplit(&c, dy1, pk1b, dh, s));
        STEP(pcgan_add(dh, dout, dx, (size_t)NC * HW, PCGAN_BF16, s));
        return 0;
    }
    // second half of the block: out = IN(y2) + x
    STEP(pcgan_instnorm_bwd(dout, y2, nullptr, mean2, m22, dy2, psum, dy2_amax, nullptr, N, C, HW, d->eps, PCGAN_ACT_NONE, 0.f, PCGAN_F32, s));
    PCGAN_CHECK(hipEventRecord(ev, ms) == hipSuccess && hipStreamWaitEvent(ss, ev, 0) == hipSuccess, "resblock_bwd: stream fork failed");
    STEP(pcgan_conv2d_bwd_weight_hsplit(&c, h, h_amax, NC, dy2, dy2_amax, NC, dw2, 1, wgrad_ws, wgrad_ws_bytes, side));
    if (db2) STEP(pcgan_sum_planes(psum, db2, N, C, 1, side));
    STEP(pcgan_conv2d_bwd_data_hsplit(&c, dy2, dy2_amax, NC, pk2b, dh, s));
    // first half: h = relu(IN(y1))
    STEP(pcgan_instnorm_bwd(dh, y1, h, mean1, m21, dy1, psum1, dy1_amax, nullptr, N, C, HW, d->eps, PCGAN_ACT_RELU, 0.f, PCGAN_F32, s));
    PCGAN_CHECK(hipEventRecord(ev, ms) == hipSuccess && hipStreamWaitEvent(ss, ev, 0) == hipSuccess, "resblock_bwd: stream fork failed");
    STEP(pcgan_conv2d_bwd_weight_hsplit(&c, x, x_amax, n_xamax, dy1, dy1_amax, NC, dw1, 1, wgrad_ws, wgrad_ws_bytes, side));
    if (db1) STEP(pcgan_sum_planes(psum1, db1, N, C, 1, side));
    // ... and the skip connection's gradient summed in the epilogue
    STEP(pcgan_conv2d_bwd_data_hsplit_add(&c, dy1, dy1_amax, NC, pk1b, dout, dx, s));
    return 0;
}

// ---- a run of ResnetBlocks per call (round 4) -----------------------------------------------------------------------------------------
// The generator's nine blocks are one chain: block i reads block i - 1's output and its plane maxima.  pcgan_restrunk_fwd / _bwd issue
// the launches of `nblocks` consecutive pcgan_resblock_fwd / _bwd calls from ONE host call (18 -> 2 calls per generator pass and
// direction, one autograd node instead of nine); same kernels, arguments, order and streams: bit-identical again.  Per-block tensors are
// slices of stacked buffers ([nblocks][...]); per-block parameter pointers come as host arrays.
extern "C" int pcgan_restrunk_fwd(const pcgan_resblock_desc* d, int nblocks, const void* x, const float* x_amax, int n_xamax,
                                  const void* const* pk1, const float* const* b1, const void* const* pk2, const float* const* b2,
                                  float* const* rm1, float* const* rv1, float* const* rm2, float* const* rv2, void* y1, void* h, void* y2,
                                  void* out, float* stats, float* amax, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_resblock_supported(d) && nblocks >= 1, "restrunk_fwd: unsupported shape or empty chain");
    PCGAN_CHECK(x && (half_of(d) || (x_amax && n_xamax > 0)) && pk1 && pk2 && b1 && b2 && rm1 && rv1 && rm2 && rv2 && y1 && h && y2 && out && stats && amax,
                "restrunk_fwd: null pointer");
    const size_t NC = (size_t)d->N * d->C, el = NC * d->H * d->W * (half_of(d) ? 2 : 4);
    for (int i = 0; i < nblocks; ++i) {
        const void* xi = i == 0 ? x : (const char*)out + (size_t)(i - 1) * el;
        const float* xa = i == 0 ? x_amax : amax + (size_t)(i - 1) * 2 * NC + NC;
        STEP(pcgan_resblock_fwd(d, xi, xa, i == 0 ? n_xamax : (int)NC, pk1[i], b1[i], pk2[i], b2[i], rm1[i], rv1[i], rm2[i], rv2[i],
                                (char*)y1 + i * el, (char*)h + i * el, (char*)y2 + i * el, (char*)out + i * el, stats + (size_t)i * 4 * NC,
                                amax + (size_t)i * 2 * NC, s));
    }
    return 0;
}

// dx receives the gradient of the chain's input.  dy2 / dy1: [nblocks] temporaries (read by the parameter-gradient stream after the
// call returns); dh: one temporary; dxs: [2] ping-pong buffers for the gradient handed from block to block (all on stream `s`).
extern "C" int pcgan_restrunk_bwd(const pcgan_resblock_desc* d, int nblocks, const void* dout, const void* x, const float* x_amax, int n_xamax,
                                  const void* y1, const void* h, const void* y2, const void* out, const float* stats, const float* amax,
                                  const void* const* pk1b, const void* const* pk2b, float* const* dw1, float* const* db1, float* const* dw2,
                                  float* const* db2, void* dy2, void* dh, void* dy1, void* dxs, void* dx, float* scratch, void* wgrad_ws,
                                  size_t wgrad_ws_bytes, pcgan_stream_t s, pcgan_stream_t side, pcgan_event_t fork_event) {
    PCGAN_CHECK(pcgan_resblock_supported(d) && nblocks >= 1, "restrunk_bwd: unsupported shape or empty chain");
    PCGAN_CHECK(dout && x && (half_of(d) || (x_amax && n_xamax > 0)) && y1 && h && y2 && out && stats && amax && pk1b && pk2b && dw1 && db1 && dw2 && db2 &&
                    dy2 && dh && dy1 && dxs && dx && scratch && wgrad_ws && fork_event,
                "restrunk_bwd: null pointer");
    const size_t NC = (size_t)d->N * d->C, el = NC * d->H * d->W * (half_of(d) ? 2 : 4);
    const void* g = dout;
    for (int i = nblocks - 1; i >= 0; --i) {
        const void* xi = i == 0 ? x : (const char*)out + (size_t)(i - 1) * el;
        const float* xa = i == 0 ? x_amax : amax + (size_t)(i - 1) * 2 * NC + NC;
        void* dxi = i == 0 ? dx : (char*)dxs + (size_t)(i & 1) * el;
        STEP(pcgan_resblock_bwd(d, g, xi, xa, i == 0 ? n_xamax : (int)NC, (const char*)y1 + i * el, (const char*)h + i * el,
                                amax + (size_t)i * 2 * NC, (const char*)y2 + i * el, stats + (size_t)i * 4 * NC, pk1b[i], pk2b[i], dw1[i], db1[i],
                                dw2[i], db2[i], (char*)dy2 + i * el, dh, (char*)dy1 + i * el, dxi, scratch + (size_t)i * 5 * NC, wgrad_ws,
                                wgrad_ws_bytes, s, side, fork_event));
        g = dxi;
    }
    return 0;
}
