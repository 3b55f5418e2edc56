// libpcgan_hip.so -- convolutions that GATHER at most 4 channels, on the f16 matrix pipe (fp32 tensors, gfx950 only).
//
// The 7x7 stems of the generator (4 -> 64 @128x128, reflection padding; models/networks.py:578-581) and of the Elo encoder's ResNet-18
// (3 -> 64, stride 2 @224x224; models/resnet.py:108-111), the first PatchGAN layer (4 -> 64, 4x4 stride 2; models/networks.py:753-755)
// and the DATA GRADIENT of the generator's 64 -> 3 head (models/networks.py:603-605: a forward-form convolution of the 3-channel dy with
// flipped weights) have K = taps x 4 <= 196 and write or read a 64-channel tensor: their bound is the HBM time of that tensor (134 MB for
// the generator's outermost layers = 27 us), not the matrix pipe -- the fp32-MFMA kernel igemm2_kernel<.., 4> took 123-238 us.
//
// One workgroup = 8 x 32 output pixels x 64 output channels.  Its input window ((8 - 1) stride + R) x ((32 - 1) stride + S) positions x 4
// channels is loaded ONCE (reflection / zero padding applied by the loading thread), scaled by the tensor's power-of-two scale, split into
// two fp16 pieces and written to LDS as [position][4 channels]: with the reduction index ordered k = 4 tap + channel, the 8 consecutive k
// an MFMA lane feeds are TWO ADJACENT TAPS of one pixel = two 8-byte LDS reads at addresses that differ from pixel to pixel by a
// constant -- no im2col, no per-tap gather, every input element split once.  The weights never touch LDS: the pack kernel stores them
// pre-scaled (one power of two per output row), pre-split and in MFMA fragment order, and each wave (32 output channels x 4 rows of 32
// pixels) loads its 13 stages x 2 pieces into registers while the window is on its way.  Three v_mfma_f32_32x32x16_f16 per 32x32 block
// and stage ((l,h) (h,l) (h,h), fp32 accumulators), scaled back exactly in the epilogue; non-finite sentinel as in the other fp16-route
// kernels (common.h).
#include "common.h"

namespace pcgan {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned THIN_OOB = 0x80000000u;      // a buffer offset that fails the range check: the load returns 0

struct ThinArgs {
    const float* X;          // [N][Cin][H][W], Cin <= 4
    float* Y;                // [N][M][P][Q]
    const u32x4* A;          // packed weights: [M / 32][stage][piece][64 lanes] x 16 bytes
    const float* rowmax;     // [M] largest |w| of every output row (written by the pack)
    const float* bias;       // [M] or null
    const float* x_amax;     // partial maxima of |x|
    int x_namax;
    int N, Cin, H, W, M, P, Q, pad, reflect, tilesX, tilesY, act;
    float slope;
    unsigned x_bytes;
    unsigned* ovf;
};

__device__ __forceinline__ constexpr int tap_off(int t, int T, int S, int WC) { return t < T ? (t / S) * WC + (t % S) : 0; }

template <int R, int S, int ST>
__global__ void __launch_bounds__(256) thin_conv_kernel(ThinArgs a) {
    constexpr int TH = 8, TW = 32, T = R * S;
    constexpr int WR = (TH - 1) * ST + R, WC = (TW - 1) * ST + S, NPOS = WR * WC;
    constexpr int NST = (T * 4 + 15) / 16;          // 16-deep stages: 4 taps x 4 channels
    constexpr int PER = (NPOS + 255) / 256;         // window positions per thread
    __shared__ __attribute__((aligned(16))) u32x2 Wh[NPOS], Wl[NPOS];     // [position] = 4 fp16 (channels 0-3) of the high / low piece
    __shared__ float isw[64];
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, ph = wave >> 1;        // the wave's 32 output channels of the 64, its 4 rows of the 8
    const int bid = blockIdx.x, mt = blockIdx.y;
    const int tx = bid % a.tilesX, ty = (bid / a.tilesX) % a.tilesY, n = bid / (a.tilesX * a.tilesY);
    const int oy0 = ty * TH, ox0 = tx * TW;

    // the wave's weight fragments: stage-major, 1 KB per (stage, piece) and wave
    // (a rolling prefetch, AHEAD stages in front of their use: all 13 stages up front cost 198 registers = two workgroups per CU; the
    // occupancy is worth more than the early loads -- 0.079 -> 0.062 ms for the generator's stem at three per CU)
#ifndef THIN_AHEAD
#define THIN_AHEAD 2
#endif
    constexpr int AHEAD = THIN_AHEAD < NST ? THIN_AHEAD : NST, NH = AHEAD;
    f16x8 Ah[NST], Al[NST];
    const u32x4* Ap = a.A + ((size_t)(mt * 2 + wm) * NST * 2) * 64 + lane;
#pragma unroll
    for (int st = 0; st < NH; ++st) {
        Ah[st] = __builtin_bit_cast(f16x8, Ap[(st * 2 + 0) * 64]);
        Al[st] = __builtin_bit_cast(f16x8, Ap[(st * 2 + 1) * 64]);
    }

    // window loads first (their latency covers the scale reduction below)
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)a.x_bytes, 0x00020000);
    const int HW = a.H * a.W;
    float xv[PER][4];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid + i * 256;
        const int wy = p / WC, wx = p - wy * WC;
        int iy = oy0 * ST - a.pad + wy, ix = ox0 * ST - a.pad + wx;
        bool ok = p < NPOS;
        if (a.reflect) {      // ReflectionPad2d: -k -> k, H - 1 + k -> H - 1 - k; positions only ragged tiles reach are clamped (never stored)
            iy = iy < 0 ? -iy : (iy >= a.H ? 2 * a.H - 2 - iy : iy);
            ix = ix < 0 ? -ix : (ix >= a.W ? 2 * a.W - 2 - ix : ix);
            iy = min(max(iy, 0), a.H - 1);
            ix = min(max(ix, 0), a.W - 1);
        } else {
            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        }
        const unsigned vo = ok ? (unsigned)(iy * a.W + ix) * 4u : THIN_OOB;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            xv[i][c] = c < a.Cin ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, vo, (unsigned)((n * a.Cin + c) * HW) * 4u, 0)) : 0.f;
    }

    const float sx = pow2_scale(block_max(thread_max_of_partials(a.x_amax, a.x_namax, tid, 256), red));
    if (tid < 64) isw[tid] = 1.f / (pow2_scale(a.rowmax[mt * 64 + tid]) * sx);

#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int p = tid + i * 256;
        f16x4 h, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            _Float16 x, y;
            split2h(xv[i][c] * sx, x, y);
            h[c] = x;
            l[c] = y;
        }
        if (p < NPOS) {
            Wh[p] = __builtin_bit_cast(u32x2, h);
            Wl[p] = __builtin_bit_cast(u32x2, l);
        }
    }
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // lane (lo, hi): pixel column lo of the wave's rows ph * 4 + j; k half hi = taps 4 st + 2 hi, + 1
    const int base = (ph * 4 * ST) * WC + lo * ST;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        if (st + AHEAD < NST) {
            Ah[st + AHEAD] = __builtin_bit_cast(f16x8, Ap[((st + AHEAD) * 2 + 0) * 64]);
            Al[st + AHEAD] = __builtin_bit_cast(f16x8, Ap[((st + AHEAD) * 2 + 1) * 64]);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int o0 = hi ? tap_off(st * 4 + 2, T, S, WC) : tap_off(st * 4 + 0, T, S, WC);
        const int o1 = hi ? tap_off(st * 4 + 3, T, S, WC) : tap_off(st * 4 + 1, T, S, WC);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = base + j * ST * WC;
            u32x4 bh, bl;
            const u32x2 h0 = Wh[q + o0], h1 = Wh[q + o1], l0 = Wl[q + o0], l1 = Wl[q + o1];
            bh.x = h0.x; bh.y = h0.y; bh.z = h1.x; bh.w = h1.y;
            bl.x = l0.x; bl.y = l0.y; bl.z = l1.x; bl.w = l1.y;
            const f16x8 Bh = __builtin_bit_cast(f16x8, bh), Bl = __builtin_bit_cast(f16x8, bl);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[st], Bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[st], Bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[st], Bh, acc[j], 0, 0, 0);
        }
    }

    // epilogue: acc[j][r] = y[n][mt * 64 + wm * 32 + (r / 4) * 8 + hi * 4 + r % 4][oy0 + ph * 4 + j][ox0 + lo]
    bool bad = false;
    const int ox = ox0 + lo;
    const size_t PQ = (size_t)a.P * a.Q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int oy = oy0 + ph * 4 + j;
        if (oy >= a.P || ox >= a.Q) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + (r >> 2) * 8 + hi * 4 + (r & 3), m = mt * 64 + ml;
            const float av = acc[j][r] * isw[ml];
            bad |= is_nonfinite(av);
            a.Y[((size_t)n * a.M + m) * PQ + (size_t)oy * a.Q + ox] = act_apply(av + (a.bias ? a.bias[m] : 0.f), a.act, a.slope);
        }
    }
    report_nonfinite(a.ovf, bad);
}

// packed[(m / 32)][stage][piece][lane = hi * 32 + lo][8 fp16]: row m = 32 (m / 32) + lo, k = 16 stage + 8 hi + i = 4 tap + channel;
// value = piece of w_row[k] * pow2_scale(rowmax[row]).  transposed = 0: forward, row = output channel k_o, w[k_o][c][tap];
// transposed = 1: data gradient as a forward-form convolution, row = input channel c_i, gathered channel = k_o, w[k_o][c_i][T - 1 - tap]
__global__ void thin_pack_kernel(const float* __restrict__ w, const float* __restrict__ rowmax, u32x4* __restrict__ out, int M, int G, int T,
                                 int nst, int transposed, int Kout, int Cin, int total) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int ln = e & 63, piece = (e >> 6) & 1, st = (e >> 7) % nst, mb = (e >> 7) / nst;
    const int lo = ln & 31, hi = ln >> 5, m = mb * 32 + lo;
    const float s = m < M ? pow2_scale(rowmax[m]) : 1.f;
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = st * 16 + hi * 8 + i, tap = k >> 2, g = k & 3;
        float x = 0.f;
        if (m < M && tap < T && g < G) x = transposed ? w[((size_t)g * Cin + m) * T + (T - 1 - tap)] : w[((size_t)m * Cin + g) * T + tap];
        _Float16 h, l;
        split2h(x * s, h, l);
        v[i] = piece ? l : h;
    }
    (void)Kout;
    out[e] = __builtin_bit_cast(u32x4, v);
}

struct Geo {
    int G, M, Hin, Win, Hout, Wout, pad, reflect, R, S, ST, nst, transposed;
    bool fold;        // reflect data gradient: the kernel writes the gradient of the PADDED input, reflect_fold_kernel folds it
};

// the forward-form convolution a (descriptor, pass) pair maps to; false = not one of this kernel's shapes
bool thin_geometry(const pcgan_conv_desc* d, int pass, Geo* g) {
    if (!d || d->dtype != PCGAN_F32) return false;
    const bool k77 = d->R == 7 && d->S == 7, k44 = d->R == 4 && d->S == 4;
    if (pass == PCGAN_PASS_FWD) {
        if (d->C > 4 || d->K % 64 != 0 || d->K > 256) return false;
        if (!((k77 && (d->stride == 1 || d->stride == 2)) || (k44 && d->stride == 2))) return false;
        if (d->pad_mode == 1 && (d->stride != 1 || d->pad >= d->H || d->pad >= d->W)) return false;
        *g = Geo{d->C, d->K, d->H, d->W, d->P, d->Q, d->pad, d->pad_mode == 1, d->R, d->S, d->stride, (d->R * d->S * 4 + 15) / 16, 0, false};
        return true;
    }
    if (pass == PCGAN_PASS_BWD_DATA) {       // dx = forward-form convolution of dy (K gathered channels) with flipped weights, stride 1
        if (d->K > 4 || d->C % 64 != 0 || d->C > 256 || !k77 || d->stride != 1) return false;
        const bool refl = d->pad_mode == 1;
        const int pad = refl ? d->R - 1 : d->R - 1 - d->pad;
        if (pad < 0) return false;
        *g = Geo{d->K, d->C, d->P, d->Q, refl ? d->H + 2 * d->pad : d->H, refl ? d->W + 2 * d->pad : d->W, pad, false, d->R, d->S, 1,
                 (d->R * d->S * 4 + 15) / 16, 1, refl};
        return true;
    }
    return false;
}

size_t thin_body_bytes(const Geo& g) { return (size_t)(g.M / 32) * g.nst * 2 * 64 * 16; }

template <int R, int S, int ST>
int launch_thin(const ThinArgs& a, int mtiles, hipStream_t st) {
    hipLaunchKernelGGL((thin_conv_kernel<R, S, ST>), dim3((unsigned)(a.N * a.tilesX * a.tilesY), (unsigned)mtiles), dim3(256), 0, st, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

int run_thin(const Geo& g, int N, const void* x, const float* x_amax, int n_amax, const void* packed, const float* bias, void* y, int act,
             float slope, hipStream_t st) {
    ThinArgs a;
    memset(&a, 0, sizeof(a));
    a.X = (const float*)x; a.Y = (float*)y; a.A = (const u32x4*)packed;
    a.rowmax = (const float*)((const char*)packed + thin_body_bytes(g));
    a.bias = bias; a.x_amax = x_amax; a.x_namax = n_amax;
    a.N = N; a.Cin = g.G; a.H = g.Hin; a.W = g.Win; a.M = g.M; a.P = g.Hout; a.Q = g.Wout; a.pad = g.pad; a.reflect = g.reflect;
    a.tilesX = (g.Wout + 31) / 32; a.tilesY = (g.Hout + 7) / 8; a.act = act; a.slope = slope;
    const size_t xb = (size_t)N * g.G * g.Hin * g.Win * 4;
    PCGAN_CHECK(xb < 0x7fffffffull, "conv2d thin: input of %zu bytes exceeds the 31-bit buffer range", xb);
    PCGAN_CHECK((size_t)N * a.tilesX * a.tilesY < 0x7fffffffull, "conv2d thin: too many tiles");
    a.x_bytes = (unsigned)xb;
    a.ovf = nonfinite_counter();
    const int mt = g.M / 64;
    if (g.R == 7 && g.ST == 1) return launch_thin<7, 7, 1>(a, mt, st);
    if (g.R == 7 && g.ST == 2) return launch_thin<7, 7, 2>(a, mt, st);
    if (g.R == 4 && g.ST == 2) return launch_thin<4, 4, 2>(a, mt, st);
    set_error("conv2d thin: no kernel for %dx%d stride %d", g.R, g.S, g.ST);
    return 1;
}

}  // namespace
}  // namespace pcgan

using namespace pcgan;

extern "C" int pcgan_conv2d_thin_supported(const pcgan_conv_desc* d, int pass) {
    Geo g;
    return thin_geometry(d, pass, &g) ? 1 : 0;
}

extern "C" size_t pcgan_conv2d_thin_packed_bytes(const pcgan_conv_desc* d, int pass) {
    Geo g;
    if (!thin_geometry(d, pass, &g)) return 0;
    return thin_body_bytes(g) + (size_t)g.M * sizeof(float);
}

extern "C" size_t pcgan_conv2d_thin_workspace_bytes(const pcgan_conv_desc* d, int pass) {
    Geo g;
    if (!thin_geometry(d, pass, &g) || !g.fold) return 0;
    return (size_t)d->N * g.M * g.Hout * g.Wout * sizeof(float);
}

extern "C" int pcgan_conv2d_thin_pack(const pcgan_conv_desc* d, int pass, const float* w, void* packed, pcgan_stream_t s) {
    Geo g;
    PCGAN_CHECK(thin_geometry(d, pass, &g), "conv2d_thin_pack: unsupported shape (pcgan_conv2d_thin_supported)");
    PCGAN_CHECK(w && packed, "conv2d_thin_pack: null pointer");
    hipStream_t st = (hipStream_t)s;
    float* rowmax = (float*)((char*)packed + thin_body_bytes(g));
    if (launch_weight_row_absmax(w, d->K, d->C, d->R * d->S, g.transposed, rowmax, st)) return 2;
    const int total = (g.M / 32) * g.nst * 2 * 64;
    hipLaunchKernelGGL(thin_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (const float*)rowmax, (u32x4*)packed, g.M,
                       g.G, g.R * g.S, g.nst, g.transposed, d->K, d->C, total);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_fwd_thin(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const void* packed,
                                     const float* bias, void* y, int act, float slope, pcgan_stream_t s) {
    Geo g;
    PCGAN_CHECK(thin_geometry(d, PCGAN_PASS_FWD, &g), "conv2d_fwd_thin: unsupported shape (pcgan_conv2d_thin_supported)");
    PCGAN_CHECK(x && x_amax && n_amax > 0 && packed && y, "conv2d_fwd_thin: null pointer");
    return run_thin(g, d->N, x, x_amax, n_amax, packed, bias, y, act, slope, (hipStream_t)s);
}

extern "C" int pcgan_conv2d_bwd_data_thin(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed,
                                          void* dx, void* ws, size_t ws_bytes, pcgan_stream_t s) {
    Geo g;
    PCGAN_CHECK(thin_geometry(d, PCGAN_PASS_BWD_DATA, &g), "conv2d_bwd_data_thin: unsupported shape (pcgan_conv2d_thin_supported)");
    PCGAN_CHECK(dy && dy_amax && n_amax > 0 && packed && dx, "conv2d_bwd_data_thin: null pointer");
    if (!g.fold) return run_thin(g, d->N, dy, dy_amax, n_amax, packed, nullptr, dx, PCGAN_ACT_NONE, 0.f, (hipStream_t)s);
    PCGAN_CHECK(ws && ws_bytes >= pcgan_conv2d_thin_workspace_bytes(d, PCGAN_PASS_BWD_DATA), "conv2d_bwd_data_thin: workspace too small (%zu)",
                ws_bytes);
    if (int e = run_thin(g, d->N, dy, dy_amax, n_amax, packed, nullptr, ws, PCGAN_ACT_NONE, 0.f, (hipStream_t)s)) return e;
    return launch_reflect_fold(ws, dx, d->N * d->C, d->H, d->W, d->pad, PCGAN_F32, (hipStream_t)s);
}
