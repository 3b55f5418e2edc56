// Weight gradient of the residual-block convolution (3x3, stride 1, reflection padding 1; fp32 tensors, fp16 two-piece route) in the
// "image-innermost" form (round 4).  Replaces autograd's weight gradient of nn.Conv2d in ResnetBlock (models/networks.py:616-652).
//
// Why another form.  dW[k][c][r][s] = sum over (n, y, x) of dy[n][k][y][x] * xpad[n][c][y + r][x + s] is a GEMM whose reduction index
// is the PIXEL.  hsplit_wgrad_kernel (bf16x6_conv.hip) feeds the matrix pipe 8 consecutive x per lane, so the nine taps of a channel
// are nine differently aligned runs of the same row: every (channel, tap) column gathers, scales, splits and stores its own copy of x
// (9 x the loads, 9 x the split arithmetic, 9 x the LDS stores), and the wave tile of 64 x 64 needs 8 ds_read_b128 per 12 MFMAs -- its
// matrix pipe is busy 38 % of the active cycles (profiles/r04_counters_residual_convs.txt), LDS traffic and vector ALU are what it
// waits for.  Here the 8 values a lane hands to the MFMA are the SAME pixel of 8 different IMAGES.  A tap then moves the pixel, not the
// position inside the 16-byte record: every operand fragment -- dy and all nine shifted views of x -- is an ALIGNED 16-byte record that
// a lane loads straight from memory into the register the MFMA reads.  No LDS, no barrier, no split arithmetic and no gather tables in
// the loop; every element of x and dy is scaled and split ONCE, by a transposing pre-pass.
//
//   pre-pass   wgd_pack_kernel: x, dy (NCHW fp32) -> two fp16 pieces each, laid out [group of 8 images][pixel][channel][8 images]; x with
//              its reflection padding materialised ((H + 2) x (W + 2) pixels), so the main kernel has no border logic.  HBM-bound:
//              reads each tensor once, writes the same number of bytes.
//   main       wgd_main_kernel: ONE WAVE per workgroup, one wave per SIMD (the accumulators of a 128 x 96 tile are 192 registers);
//              tile = 128 output channels x (32 input channels x the 3 taps of one tap row); K step = one pixel x 16 images (two
//              groups: lanes 0-31 / 32-63); 4 + 3 operand fragments x 2 pieces for 36 MFMAs.  The reduction (pixels x image-group pairs)
//              is split so that tiles x splits fills the chip's SIMDs once; partial sums [split][tap][k][c].
//   reduce     wgd_reduce_kernel: fixed-order sum of the partials, scale back (exact powers of two), transposition into dW[k][c][r][s],
//              optional accumulation into the optimizer's gradient buffer, non-finite sentinel.
#include "common.h"

namespace pcgan {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned WGD_OOB = 0x80000000u;

struct WgdGeom {
    int N, C, K, H, W;      // images, input channels, output channels, image size (output = input size)
    int G;                  // groups of 8 images (N / 8)
    int Hp, Wp;             // padded size (H + 2, W + 2)
};

// ---- pre-pass ------------------------------------------------------------------------------------------------------------------------
// One workgroup = one group of 8 images x one image row x 32 channels; thread (ch = tid / 8, xq = tid % 8 + 8 * pass) loads the 16 bytes
// (4 consecutive x) of each of the 8 images -- a wave reads 8 runs of 128 bytes --, scales, splits and stores, per x, the 16-byte record
// of 8 images: consecutive channels are consecutive records (a wave writes 8 runs of 128 bytes per piece and x).
// PAD = 1: the tensor is x; rows and columns are written at +1 and the mirrored border rows / columns are written by the thread that
// holds their source (row 1 -> padded row 0, row H - 2 -> padded row H + 1, column 1 -> padded column 0, column W - 2 -> W + 1).
template <int PAD>
__global__ void __launch_bounds__(256) wgd_pack_kernel(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                       const float* __restrict__ amax, int namax, float* __restrict__ scale_out, int Cn, int H,
                                                       int W) {
    __shared__ float scratch[16];
    const int tid = threadIdx.x;
    const float m = thread_max_of_partials(amax, namax, tid, 256);
    const float sc = pow2_scale(block_max(m, scratch));
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) scale_out[0] = sc;
    const int g = blockIdx.z, y = blockIdx.y, c = blockIdx.x * 32 + (tid >> 3);
    if (c >= Cn) return;
    const int Hp = H + 2 * PAD, Wp = W + 2 * PAD;
    const size_t plane = (size_t)H * W;
    for (int xq = tid & 7; xq * 4 < W; xq += 8) {
        const int x0 = xq * 4;
        float v[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* p = src + ((size_t)(g * 8 + i) * Cn + c) * plane + (size_t)y * W + x0;
            if (x0 + 3 < W) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                v[i][0] = q.x; v[i][1] = q.y; v[i][2] = q.z; v[i][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = x0 + j < W ? p[j] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            if (x >= W) break;
            h16x8 vh, vl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                _Float16 a, b;
                split2h(v[i][j] * sc, a, b);
                vh[i] = a;
                vl[i] = b;
            }
            // padded coordinates this element lands on: itself, and its mirror images across the border
            int ys[2], nys = 0, xs[2], nxs = 0;
            ys[nys++] = y + PAD;
            xs[nxs++] = x + PAD;
            if (PAD) {
                if (y == 1) ys[nys++] = 0;
                if (y == H - 2) ys[nys++] = H + 1;
                if (x == 1) xs[nxs++] = 0;
                if (x == W - 2) xs[nxs++] = W + 1;
            }
            for (int a = 0; a < nys; ++a)
                for (int b = 0; b < nxs; ++b) {
                    const size_t rec = (((size_t)g * Hp + ys[a]) * Wp + xs[b]) * Cn + c;
                    *reinterpret_cast<h16x8*>(hi + rec * 8) = vh;
                    *reinterpret_cast<h16x8*>(lo + rec * 8) = vl;
                }
        }
    }
}

// ---- main kernel -----------------------------------------------------------------------------------------------------------------------
struct WgdMainArgs {
    const _Float16 *dyh, *dyl;      // [G][H * W][K][8]
    const _Float16 *xh, *xl;        // [G][Hp * Wp][C][8]
    float* part;                    // [splits][9][K][C]
    WgdGeom g;
    int tiles, splits, steps, steps_per_split;      // tiles = (K / 128) * (C / 32) * 3; steps = (G / 2) * H * (W + 2) virtual steps in all
    int nwg;                        // tiles * splits (the grid is rounded up to a multiple of 8)
    unsigned dy_bytes, x_bytes;     // bytes of ONE piece array
};

template <int LOOK>      // K steps the operand loads run ahead of the MFMAs
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) wgd_main_kernel(WgdMainArgs a) {
    // workgroups go to the 8 XCDs round-robin: give each XCD a contiguous run of (split, tile) pairs -- the tiles of one split read the
    // same dy records and neighbouring x records through one L2
    const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int split = wg / a.tiles, tile = wg - split * a.tiles;
    const WgdGeom& g0 = a.g;
    const int nrh = g0.K / 128;
    const int rh = tile % nrh, tr = (tile / nrh) % 3, cb = tile / (3 * nrh);      // row block (128 output channels), tap row, block of 32 input channels
    const int lane = threadIdx.x, l32 = lane & 31, h = lane >> 5;
    const WgdGeom& g = a.g;
    const int HW = g.H * g.W, HWp = g.Hp * g.Wp;
    const __amdgpu_buffer_rsrc_t rDh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.dyh), 0, (int)a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rDl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.dyl), 0, (int)a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rXh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xh), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rXl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xl), 0, (int)a.x_bytes, 0x00020000);
    // lane parts of the byte offsets: the lane's image group (h) and its row / channel
    unsigned voA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voA[i] = (unsigned)(((size_t)h * HW * g.K + rh * 128 + i * 32 + l32) * 16);
    const unsigned voB = (unsigned)(((size_t)h * HWp * g.C + cb * 32 + l32) * 16);

    // The reduction runs over VIRTUAL steps (group pair, image row y, xv = -2 .. W - 1): step (y, xv) loads ONE new x record per piece --
    // padded column xv + 2 of padded row y + tr -- and the three tap columns of pixel (y, xv) are the last three loaded (a window that
    // slides along the row: 2 instead of 6 x loads per step; this kernel is bound by the bytes it pulls out of L2, 16.8-18.8 TB/s
    // chip-wide, not by its matrix instructions); xv < 0 are the two warm-up steps of a row (loads only), and a split starts two steps
    // early in the same mode.
    const int RW = g.W + 2;                          // virtual steps per image row
    const int s0 = split * a.steps_per_split;
    const int s1 = min(a.steps, s0 + a.steps_per_split);
    // Operand ring of R = LOOK + 3 slots: the loads run LOOK steps ahead of the MFMAs and the slots of the two previous steps stay intact,
    // because their x records are the other two tap columns of the step being multiplied -- the window slides by renaming, no register
    // moves.  The loop body is R steps, every slot index a compile-time constant.  The position of the next step to load is kept as
    // scalars (group pair, image row, xv) and advanced by compare-and-add: nothing but the loads, their two scalar offsets and the
    // MFMAs is left in the loop (a single wave per SIMD has nobody to hide vector-ALU work behind).
    constexpr int R = LOOK + 3;
    struct Ops {
        h16x8 A[4][2], Bn[2];
    };
    Ops ops[R];
    const int first = s0 - 2;                        // two warm-up steps refill the window in front of the split's first product
    int lst = first, lgp, ly, lxv;                   // next step to load and its position
    {
        const int st = first < 0 ? 0 : first;
        const int row = st / RW;
        lxv = st - row * RW - 2;      // (first < 0 only for split 0: its steps -2, -1 are dead and do not move the position)
        lgp = row / g.H;
        ly = row - lgp * g.H;
    }
    const unsigned KB = (unsigned)(g.K * 16), CB = (unsigned)(g.C * 16);
    auto load = [&](Ops& o) {
        // branch-free (one basic block per loop body, so that the issue-order hints below can place the loads between the MFMAs): a
        // step that must not load gets the out-of-range bit OR-ed into its lane offsets and reads zeros
        const int live = (lst >= 0) & (lst < s1);
        const int mul = live & (lxv >= 0) & (lst >= s0);
        const unsigned killA = (unsigned)(mul - 1) & WGD_OOB, killB = (unsigned)(live - 1) & WGD_OOB;
        const unsigned soA = ((unsigned)((2 * lgp) * HW + ly * g.W + lxv) * KB) & (unsigned)(-mul);
        const unsigned soB = ((unsigned)(((2 * lgp) * g.Hp + ly + tr) * g.Wp + lxv + 2) * CB) & (unsigned)(-live);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o.A[i][0] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rDh, voA[i] | killA, soA, 0));
            o.A[i][1] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rDl, voA[i] | killA, soA, 0));
        }
        o.Bn[0] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rXh, voB | killB, soB, 0));
        o.Bn[1] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(rXl, voB | killB, soB, 0));
        ++lst;
        const int adv = lst > 0;                     // (the dead steps in front of split 0 do not move the position)
        lxv += adv;
        const int wrap = lxv == g.W;
        lxv -= wrap * RW;
        ly += wrap;
        const int wrap2 = ly == g.H;
        ly -= wrap2 * g.H;
        lgp += wrap2;
    };
    f32x16 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) ops[q].Bn[p][e] = (_Float16)0.f;
#pragma unroll
    for (int q = 0; q < LOOK; ++q) load(ops[q]);
    const int n = s1 - first;
    const int nround = (n + R - 1) / R * R;          // (a dead step loads zeros through the out-of-range offset and multiplies them)
    for (int t = 0; t < nround; t += R) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            load(ops[(q + LOOK) % R]);
            const Ops& o = ops[q];
            const Ops& w1 = ops[(q + R - 1) % R];    // the previous step's record: tap column 1
            const Ops& w0 = ops[(q + R - 2) % R];    // two steps back: tap column 0
            // (l, h) (h, l) (h, h): smallest terms first.  A warm-up step multiplies too -- by the zeros its dy fragments were loaded as
            // (2 of W + 2 steps: a branch around the MFMAs made the compiler copy the accumulators, 512 registers and spills)
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.A[i][PA[p]], w0.Bn[PB[p]], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.A[i][PA[p]], w1.Bn[PB[p]], acc[i][1], 0, 0, 0);
                    acc[i][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.A[i][PA[p]], o.Bn[PB[p]], acc[i][2], 0, 0, 0);
                }
            // issue order: the step's 10 loads spread between its 36 MFMAs (a burst of loads in front of them stalls the wave at the
            // texture-address queue while the matrix pipe of its SIMD has nothing else to run)
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial sums: part[split][tap = tr * 3 + j][k][c]; acc[i][j][r] is (row (r / 4) * 8 + h * 4 + r % 4 of block i, column l32)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float* out = a.part + (((size_t)split * 9 + tr * 3 + j) * g.K + rh * 128) * g.C + cb * 32 + l32;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(size_t)(i * 32 + (r >> 2) * 8 + h * 4 + (r & 3)) * g.C] = acc[i][j][r];
    }
}

// dW[k][c][tap] (+)= (sum over splits, in order, of part[split][tap][k][c]) / (sx * sdy); one thread per (tap, k, 4 consecutive c):
// 16-byte loads along c, `splits` of them in flight
__global__ void __launch_bounds__(256) wgd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, const float* __restrict__ scales,
                                                         int splits, int K, int C, int accumulate, unsigned* ovf) {
    const size_t KC = (size_t)K * C, per_split = 9 * KC;
    const size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;      // element index inside one split's [tap][k][c]
    const float inv = (1.f / scales[0]) * (1.f / scales[1]);
    bool bad = false;
    if (q < per_split) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = part + q;
#pragma unroll 8
        for (int sp = 0; sp < splits; ++sp) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sp * per_split);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int t = (int)(q / KC);
        const size_t kc = q - (size_t)t * KC;
        const float v[4] = {s.x * inv, s.y * inv, s.z * inv, s.w * inv};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* o = dw + (kc + i) * 9 + t;
            bad |= is_nonfinite(v[i]);
            *o = accumulate ? *o + v[i] : v[i];
        }
    }
    report_nonfinite(ovf, bad);
}

// the reduce alone, for the other producer of [split][tap][k][c] partial sums (wgrad_rowring.hip)
int launch_wgd_reduce(const float* part, float* dw, const float* scales, int splits, int K, int C, int accumulate, hipStream_t st) {
    const size_t quads = (size_t)9 * K * C / 4;
    hipLaunchKernelGGL(wgd_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, part, dw, scales, splits, K, C, accumulate,
                       nonfinite_counter());
    PCGAN_LAUNCH_CHECK();
    return 0;
}

static inline bool wgd_shape(const pcgan_conv_desc* d) {
    return d && d->dtype == PCGAN_F32 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode == 1 && d->P == d->H && d->Q == d->W &&
           d->H >= 3 && d->W >= 3 && d->N >= 16 && d->N % 16 == 0 && d->K % 128 == 0 && d->C % 32 == 0;
}
struct WgdPlan {
    WgdGeom g;
    int tiles, splits, steps, per;
    size_t dy_piece, x_piece, part_bytes, total;
};
static bool wgd_plan(const pcgan_conv_desc* d, WgdPlan* p) {
    if (!wgd_shape(d)) return false;
    p->g = WgdGeom{d->N, d->C, d->K, d->H, d->W, d->N / 8, d->H + 2, d->W + 2};
    p->tiles = (d->K / 128) * (d->C / 32) * 3;
    p->steps = (p->g.G / 2) * d->H * (d->W + 2);      // virtual steps: W products + 2 window warm-ups per image row
    int cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    int splits = (4 * cus) / p->tiles;      // one wave per SIMD, the whole chip in one round
    splits = splits < 1 ? 1 : splits;
    if (splits > p->steps / 8) splits = p->steps / 8 > 0 ? p->steps / 8 : 1;
    p->per = (p->steps + splits - 1) / splits;
    p->splits = (p->steps + p->per - 1) / p->per;
    p->dy_piece = (size_t)d->N * d->K * d->H * d->W * 2;
    p->x_piece = (size_t)d->N * d->C * p->g.Hp * p->g.Wp * 2;
    p->part_bytes = (size_t)p->splits * 9 * d->K * d->C * 4;
    p->total = 256 + 2 * align_up(p->dy_piece, 256) + 2 * align_up(p->x_piece, 256) + align_up(p->part_bytes, 256);
    return p->dy_piece < 0x7fffffffull && p->x_piece < 0x7fffffffull;
}

}  // namespace pcgan

extern "C" int pcgan_conv2d_wgrad_direct_supported(const pcgan_conv_desc* d) {
    pcgan::WgdPlan p;
    return pcgan::wgd_plan(d, &p) ? 1 : 0;
}

extern "C" size_t pcgan_conv2d_wgrad_direct_workspace_bytes(const pcgan_conv_desc* d) {
    pcgan::WgdPlan p;
    return pcgan::wgd_plan(d, &p) ? p.total : 0;
}

extern "C" int pcgan_conv2d_bwd_weight_direct(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy,
                                              const float* dy_amax, int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes,
                                              pcgan_stream_t s) {
    using namespace pcgan;
    WgdPlan p;
    PCGAN_CHECK(wgd_plan(d, &p), "conv2d_bwd_weight_direct: unsupported shape (3x3 stride 1 reflection padding 1, fp32, N %% 16 == 0, K %% 128 == 0, C %% 32 == 0)");
    PCGAN_CHECK(x && dy && dw && ws && x_amax && dy_amax && n_xamax > 0 && n_dyamax > 0, "conv2d_bwd_weight_direct: null pointer");
    PCGAN_CHECK(ws_bytes >= p.total, "conv2d_bwd_weight_direct: workspace too small (%zu < %zu)", ws_bytes, p.total);
    hipStream_t st = (hipStream_t)s;
    char* w = (char*)ws;
    float* scales = (float*)w;      // [0] = x, [1] = dy
    w += 256;
    _Float16* dyh = (_Float16*)w; w += align_up(p.dy_piece, 256);
    _Float16* dyl = (_Float16*)w; w += align_up(p.dy_piece, 256);
    _Float16* xh = (_Float16*)w; w += align_up(p.x_piece, 256);
    _Float16* xl = (_Float16*)w; w += align_up(p.x_piece, 256);
    float* part = (float*)w;
    {
        TimerScope whole(timer_kind_res(d, TIMER_RES_WGRAD), st);
        const dim3 gx((unsigned)((d->C + 31) / 32), (unsigned)d->H, (unsigned)p.g.G), gd((unsigned)((d->K + 31) / 32), (unsigned)d->H, (unsigned)p.g.G);
        hipLaunchKernelGGL((wgd_pack_kernel<1>), gx, dim3(256), 0, st, (const float*)x, xh, xl, x_amax, n_xamax, scales, d->C, d->H, d->W);
        hipLaunchKernelGGL((wgd_pack_kernel<0>), gd, dim3(256), 0, st, (const float*)dy, dyh, dyl, dy_amax, n_dyamax, scales + 1, d->K, d->H, d->W);
        WgdMainArgs a;
        a.dyh = dyh; a.dyl = dyl; a.xh = xh; a.xl = xl; a.part = part;
        a.g = p.g;
        a.tiles = p.tiles; a.splits = p.splits; a.steps = p.steps; a.steps_per_split = p.per;
        a.nwg = p.tiles * p.splits;
        a.dy_bytes = (unsigned)p.dy_piece;
        a.x_bytes = (unsigned)p.x_piece;
        const unsigned grid = (unsigned)((a.nwg + 7) / 8 * 8);
        {
            TimerScope main_only(timer_kind_res(d, TIMER_RES_WGRAD_MAIN), st);
            switch (option(OPT_WGD_LOOK)) {
                case 2: hipLaunchKernelGGL((wgd_main_kernel<2>), dim3(grid), dim3(64), 0, st, a); break;
                case 4: hipLaunchKernelGGL((wgd_main_kernel<4>), dim3(grid), dim3(64), 0, st, a); break;
                default: hipLaunchKernelGGL((wgd_main_kernel<3>), dim3(grid), dim3(64), 0, st, a); break;
            }
        }
        const size_t quads = (size_t)9 * d->K * d->C / 4;
        hipLaunchKernelGGL(wgd_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, (const float*)part, dw, (const float*)scales, p.splits,
                           d->K, d->C, accumulate, nonfinite_counter());
    }
    PCGAN_LAUNCH_CHECK();
    return 0;
}
