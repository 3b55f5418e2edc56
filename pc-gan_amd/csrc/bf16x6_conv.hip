// Stride-1 convolution forward -- and the data and weight gradients of the reflection-padded 3x3 convolution -- on the bf16
// matrix pipe, in two precisions that share one kernel skeleton (template parameters NP = pieces per operand, TA = storage):
//   NP = 3, TA = float  fp32 tensors, fp32-level accuracy (the default route of the residual-block convolutions, below);
//   NP = 1, TA = bf16   the bf16 path (desc.dtype = PCGAN_BF16): activations stored as bf16, weights rounded to bf16 by the
//                       pack kernel, ONE product per term, fp32 accumulators -- plain mixed precision, 6x fewer MFMAs.
//
// An fp32 value is the exact sum of three bf16 pieces, x = h + m + l (8 + 8 + 8 significand bits).  A product a*b then needs
// the piece pairs (h,h) | (h,m) (m,h) | (h,l) (m,m) (l,h) to keep every term above 2^-24 |a||b|; everything is accumulated in
// the fp32 accumulators of v_mfma_f32_32x32x16_bf16.  scripts/micro/bf16_split measures, for K = 2304 (the residual-block
// convolution): relative L2 error 7.0e-7 against float64, fp32 MFMA 6.1e-7; sustained rate of the six instructions that stand
// for one fp32 K = 16 step 304 TFLOP/s fp32-equivalent against 155 TFLOP/s of v_mfma_f32_32x32x2_f32.  This kernel: 0.171-0.179 ms
// on the residual convolution (fp32 implicit GEMM: 0.269 ms).
//
// Replaces the same call sites as the fp32 implicit GEMM (nn.ReflectionPad2d + nn.Conv2d of the ResnetBlocks,
// models/networks.py:621-648; stride-1 nn.Conv2d elsewhere) when the gathered channel count is a multiple of 16.
//
// Y[m][pix] = sum_k A[m][k] * G(k, pix), K ordered (16-channel chunk, tap, channel).  Workgroup = 128 output channels x 128
// pixels, 4 waves of 64 x 64 (2 x 2 accumulators), one K stage = 16 channels of one tap:
//   weights  pre-split by the pack kernel, stored [piece][M tile][stage][k half][128 rows][8 bf16]: a stage is 3 coalesced
//            16-byte loads per thread that go to LDS unchanged;
//   pixels   8 channels of one pixel per thread (lanes along pixels: coalesced), split into the three pieces in registers,
//            three 16-byte LDS writes;
//   LDS      [piece][k half][row or pixel][8 bf16]: every access 16 bytes, 16 consecutive lanes = 256 contiguous bytes;
//   per wave and stage 12 ds_read_b128 and 24 MFMAs (smallest terms first); global loads run three stages ahead, the LDS reads
//   of the next stage sit under the MFMAs of the current one (operands double-buffered in registers), one barrier per stage.
#include "common.h"
#include <stdlib.h>

namespace pcgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned BS_OOB = 0x80000000u;
static constexpr int BS_MAXTAP = 25;

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

struct BsplitArgs {
    const void* X;       // [N][C][H][W], storage type TA
    const void* A;       // packed weights, see above
    const float* bias;   // [M] or null
    void* Y;             // [N][M][P][Q], storage type TA (weight gradient: fp32 partial sums)
    int N, C, H, W, M, R, S, pad, reflect, P, Q;
    int nMt, nst, act;
    float slope;
    unsigned x_bytes, a_bytes;
    // data gradient of the reflection-padded 3x3 convolution: three row classes (rows without a mirror image | row 1 | row H-2),
    // each with its own packed weights (the row mirror is folded into them) and its own run of pixel tiles in the grid
    int tstart[4];       // first pixel tile of each phase, tstart[3] = total
    unsigned phase_bytes;
    // weight gradient: blockIdx.y takes stages [y * nst_split, (y + 1) * nst_split) of the pixel reduction and writes a raw partial sum
    int nst_split;
};

enum { BS_FWD_ZERO = 0, BS_FWD_REFLECT = 1, BS_DGRAD_REFLECT = 2, BS_WGRAD = 3 };

// one 16-byte entry (8 values) of every piece of a packed weight image: entry e of piece p sits at A + p * per_piece + 8 e
__device__ __forceinline__ void pack_store8(__bf16* __restrict__ A, size_t e, size_t per_piece, int np, float wscale, const float (&v)[8]) {
    if (np == 2) {       // two fp16 pieces of the scaled value
        f16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 x, y;
            split2h(v[j] * wscale, x, y);
            h[j] = x;
            l[j] = y;
        }
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(A) + 8 * e) = h;
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(A) + per_piece + 8 * e) = l;
        return;
    }
    bf16x8 h, mm, l;     // np == 1: the weight rounded to nearest-even bf16; np == 3: the exact three-piece split
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 x, y, z;
        split3(v[j], x, y, z);
        h[j] = x;
        mm[j] = y;
        l[j] = z;
    }
    *reinterpret_cast<bf16x8*>(A + 8 * e) = h;
    if (np == 3) {
        *reinterpret_cast<bf16x8*>(A + per_piece + 8 * e) = mm;
        *reinterpret_cast<bf16x8*>(A + 2 * per_piece + 8 * e) = l;
    }
}

// weights w[M][C][R][S] -> [piece][mt][stage][half][BM][8] bf16 (BM = 128 << bm_shift), stage = chunk * T + tap, k in stage =
// channel in chunk
// np = 2: two fp16 pieces of w[m][..] * pow2_scale(rowmax[m]) (the fp16 route: ONE power of two per output row, so that a filter
// row far below the tensor's largest weight keeps its 22 bits -- the epilogue divides row m by the same power)
__global__ void bsplit_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ A, int M, int C, int T, int nMt, int nst,
                                   int bm_shift, int np, const float* __restrict__ rowmax = nullptr) {
    const int BM = 128 << bm_shift;
    const size_t per_piece = (size_t)nMt * nst * 16 * BM;
    // a thread builds one 16-byte entry (8 consecutive channels of a row and tap) of every piece: 8 loads in flight, one store per piece
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < per_piece / 8; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e & (BM - 1)), half = (int)((e >> (7 + bm_shift)) & 1);
        const size_t q = e >> (8 + bm_shift);
        const int st = (int)(q % nst), mt = (int)(q / nst);
        const int m = mt * BM + row, c0 = (st / T) * 16 + half * 8, tap = st % T;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = m < M ? w[((size_t)m * C + c0 + j) * T + tap] : 0.f;
        const float wscale = (np == 2 && m < M) ? pow2_scale(rowmax[m]) : 1.f;
        pack_store8(A, e, per_piece, np, wscale, v);
    }
}

// partial maxima of |x|: out[blockIdx.x] = the largest magnitude this workgroup saw (consumers take the largest of the partials)
template <typename TA>
__global__ void __launch_bounds__(256) absmax_kernel(const TA* __restrict__ x, size_t n, float* __restrict__ out) {
    float m = 0.f;
    // scalar head up to a 16-byte boundary (a weight tensor may be a view into the optimizer's flat buffer), vector body, scalar tail
    size_t head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / sizeof(TA);
    head = head < n ? head : n;
    const TA* xb = x + head;
    const size_t nb = n - head, n4 = nb / 4, stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {       // four independent 16-byte loads in flight per thread
        const float4 a = ld4(xb + 4 * i), b = ld4(xb + 4 * (i + stride)), c = ld4(xb + 4 * (i + 2 * stride)), d = ld4(xb + 4 * (i + 3 * stride));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
    }
    for (; i < n4; i += stride) {
        const float4 v = ld4(xb + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) m = fmaxf(m, fabsf(ld1(x + threadIdx.x)));
        if (threadIdx.x < (nb & 3)) m = fmaxf(m, fabsf(ld1(xb + 4 * n4 + threadIdx.x)));
    }
    __shared__ float red[16];
    m = block_max(m, red);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// Audit of operand maxima (round 4): `claimed` = the partial maxima a tensor carries (its producer's, or an earlier absmax pass),
// `fresh` = the partials of an absmax pass made NOW.  Both are maxima over the same stored values, so their largest entries must be
// EQUAL; counts[0] += 1 when the tensor holds a larger value than claimed (an fp16 piece would overflow), counts[1] += 1 when its
// largest value is below 2^-8 of the claim (the scaled operand sits >= 8 bits under the fp16 target range: low pieces go subnormal and
// precision is lost SILENTLY -- the case the non-finite sentinel cannot see), counts[2] += 1 for any other mismatch.
__global__ void __launch_bounds__(256) amax_audit_kernel(const float* __restrict__ claimed, int nc, const float* __restrict__ fresh, int nf,
                                                         unsigned* __restrict__ counts) {
    __shared__ float red[16];
    float c = 0.f, f = 0.f;
    for (int i = threadIdx.x; i < nc; i += 256) c = fmaxf(c, claimed[i]);
    for (int i = threadIdx.x; i < nf; i += 256) f = fmaxf(f, fresh[i]);
    c = block_max(c, red);
    __syncthreads();
    f = block_max(f, red);
    if (threadIdx.x == 0 && f != c) {
        if (!(f <= c)) atomicAdd(counts + 0, 1u);                  // larger than claimed (or NaN)
        else if (f < c * 0.00390625f) atomicAdd(counts + 1, 1u);    // under-scaled by 2^8 or more
        else atomicAdd(counts + 2, 1u);
    }
}

// largest magnitude of every ROW of a convolution's weight matrix, w[K][C][T]: by_c = 0 the rows of the forward GEMM (output channel k:
// C * T contiguous values), by_c = 1 the rows of the data gradient (input channel c: K runs of T values).  One workgroup per row.
__global__ void __launch_bounds__(256) weight_row_absmax_kernel(const float* __restrict__ w, int K, int C, int T, int by_c, float* __restrict__ out) {
    const int row = blockIdx.x;
    float m = 0.f;
    if (!by_c) {
        const float* p = w + (size_t)row * C * T;
        for (int i = threadIdx.x; i < C * T; i += 256) m = fmaxf(m, fabsf(p[i]));
    } else {
        for (int i = threadIdx.x; i < K * T; i += 256) {
            const int k = i / T, t = i - k * T;
            m = fmaxf(m, fabsf(w[((size_t)k * C + row) * T + t]));
        }
    }
    __shared__ float red[16];
    m = block_max(m, red);
    if (threadIdx.x == 0) out[row] = m;
}

// ---- weight gradient as the same GEMM with the roles turned: rows = output channels k (operand A = dy, re-split per call), columns
// = (c, r, s), reduction = (n, y, x) in stages of 16 consecutive x.  The reflection padding is materialised once (xpad), so that
// the gather address is separable: column part (c, r, s) in the lane's offset, reduction part (n, y, x) in the scalar offset.
template <typename TA>
__global__ void bsplit_pad_reflect_kernel(const TA* __restrict__ x, TA* __restrict__ xp, int H, int W, int pad, int reflect = 1) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const TA* src = x + (size_t)blockIdx.y * H * W;
    TA* dst = xp + (size_t)blockIdx.y * Hp * Wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hp * Wp; i += gridDim.x * blockDim.x) {
        int y = i / Wp - pad, xx = i % Wp - pad;
        if (!reflect) {      // zero padding
            const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            TA v;
            st1(&v, 0.f);
            dst[i] = in ? src[y * W + xx] : v;
            continue;
        }
        y = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
        xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
        dst[i] = src[y * W + xx];
    }
}

// the same copy for small planes (the residual blocks: 8192 planes of 34 x 34): one WAVE per plane, a lane writes PAIRS of neighbouring
// elements (padded width even: a pair never leaves its row, every pair is 8- / 4-byte aligned), row / column advanced without
// a division.  The element-per-thread form above spent its time in 40960 workgroups of one element per thread (0.030 ms).
template <typename TA>
__global__ void __launch_bounds__(256) bsplit_pad_wave_kernel(const TA* __restrict__ x, TA* __restrict__ xp, int planes, int H, int W, int pad, int reflect) {
    const int lane = threadIdx.x & 63;
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, half = Wp >> 1;
    const TA* src = x + (size_t)plane * H * W;
    TA* dst = xp + (size_t)plane * Hp * Wp;
    int row = lane / half, cp = lane - row * half;
    const int drow = 64 / half, dcp = 64 - drow * half;
    typedef TA pair_t __attribute__((ext_vector_type(2)));
    while (row < Hp) {
        int y = row - pad, x0 = 2 * cp - pad, x1 = x0 + 1;
        bool in0 = (unsigned)y < (unsigned)H, in1 = in0;
        if (reflect) {
            y = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
            x0 = x0 < 0 ? -x0 : (x0 >= W ? 2 * (W - 1) - x0 : x0);
            x1 = x1 < 0 ? -x1 : (x1 >= W ? 2 * (W - 1) - x1 : x1);
            in0 = in1 = true;
        } else {
            in0 = in0 && (unsigned)x0 < (unsigned)W;
            in1 = in1 && (unsigned)x1 < (unsigned)W;
        }
        TA zero;
        st1(&zero, 0.f);
        pair_t v;
        v.x = in0 ? src[y * W + x0] : zero;
        v.y = in1 ? src[y * W + x1] : zero;
        *reinterpret_cast<pair_t*>(dst + row * Wp + 2 * cp) = v;
        cp += dcp;
        row += drow;
        if (cp >= half) {
            cp -= half;
            ++row;
        }
    }
}

static void launch_pad(const void* x, void* xpad, int planes, int H, int W, int pad, int reflect, bool half, hipStream_t st) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    if (Wp % 2 == 0 && Wp <= 128 && Hp * Wp <= 8192 && planes >= 1024) {
        const dim3 grid((planes + 3) / 4);
        if (half) hipLaunchKernelGGL(bsplit_pad_wave_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, (bf16*)xpad, planes, H, W, pad, reflect);
        else hipLaunchKernelGGL(bsplit_pad_wave_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)xpad, planes, H, W, pad, reflect);
        return;
    }
    const dim3 pgrid((Hp * Wp + 255) / 256, planes);
    if (half) hipLaunchKernelGGL(bsplit_pad_reflect_kernel<bf16>, pgrid, dim3(256), 0, st, (const bf16*)x, (bf16*)xpad, H, W, pad, reflect);
    else hipLaunchKernelGGL(bsplit_pad_reflect_kernel<float>, pgrid, dim3(256), 0, st, (const float*)x, (float*)xpad, H, W, pad, reflect);
}

// dy[N][K][HW] -> [piece][stage][half][BM][8] bf16 pieces, stage = 16 consecutive elements of the (n, y, x) reduction
template <typename TA>
__global__ void bsplit_pack_dy_kernel(const TA* __restrict__ dy, __bf16* __restrict__ A, int K, int HW, int nst, int bm_shift, int np) {
    const int BM = 128 << bm_shift;
    const size_t per_piece = (size_t)nst * 16 * BM;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_piece; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), row = (int)((i >> 3) & (BM - 1)), half = (int)((i >> (10 + bm_shift)) & 1);
        const size_t st = i >> (11 + bm_shift);
        const size_t e = st * 16 + half * 8 + j;
        const size_t n = e / HW, r = e - n * HW;
        const float v = row < K ? ld1(dy + (n * K + row) * HW + r) : 0.f;
        __bf16 h, mm, l;
        split3(v, h, mm, l);
        A[i] = h;
        if (np == 3) {
            A[per_piece + i] = mm;
            A[2 * per_piece + i] = l;
        }
    }
}

// dw[i] (+)= sum over the splits in a fixed order
__global__ void bsplit_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, size_t total, int accumulate) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    float a0 = 0.f, a1 = 0.f;
    int sp = 0;
    for (; sp + 1 < splits; sp += 2) {
        a0 += part[(size_t)sp * total + i];
        a1 += part[(size_t)(sp + 1) * total + i];
    }
    if (sp < splits) a0 += part[(size_t)sp * total + i];
    const float v = a0 + a1;
    dw[i] = accumulate ? dw[i] + v : v;
}

// BM = 128: 4 waves, two workgroups per CU, each thread gathers 8 channels of its pixel per stage.
// BM = 256: 8 waves (4 x 2 of 64 x 64) share ONE gathered / split pixel tile for all 256 output channels: half the gathers, split
//           arithmetic and pixel LDS writes per MFMA; each thread gathers 4 channels; one workgroup per CU.
// data-gradient weights of the reflect-padded 3x3 convolution: A[phase][piece][mt][stage][half][BM][8], rows = input channels c,
// k = (16-chunk of output channels, tap (r', s'), channel), value = wf[c][k][r'][s'] = w[k][c][2-r'][2-s'] with the row mirror
// folded in: row class 1 (row 1) reads row 0 through tap r'=0 for itself AND for padded row -1: wf'[0] = wf[0] + wf[2];
// row class 2 (row H-2): wf'[2] = wf[2] + wf[0].
// nphase = 1: only row class 0, the plain flipped weights (all the window kernel reads)
__global__ void bsplit_pack_dgrad_kernel(const float* __restrict__ w, __bf16* __restrict__ A, int K, int C, int nMt, int nst, int bm_shift,
                                         int np, const float* __restrict__ rowmax = nullptr, int nphase = 3) {
    const int BM = 128 << bm_shift;      // (rowmax: per INPUT channel c -- the rows of the data gradient -- as in bsplit_pack_kernel)
    const size_t per_piece = (size_t)nMt * nst * 16 * BM, per_phase = (size_t)np * per_piece;
    // (nphase row classes x per_piece / 8 entries of 8 values; each entry writes its np pieces.  A folded row-class weight is at most
    // twice the largest weight: still far inside the fp16 range)
    const size_t per_phase8 = per_piece / 8;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)nphase * per_phase8; i += (size_t)gridDim.x * blockDim.x) {
        const int phase = (int)(i / per_phase8);
        const size_t e = i - (size_t)phase * per_phase8;
        const int row = (int)(e & (BM - 1)), half = (int)((e >> (7 + bm_shift)) & 1);
        const size_t q = e >> (8 + bm_shift);
        const int st = (int)(q % nst), mt = (int)(q / nst);
        const int c = mt * BM + row, k0 = (st / 9) * 16 + half * 8, tap = st % 9;
        const int rp = tap / 3, sp = tap - rp * 3;
        const bool fold = (phase == 1 && rp == 0) || (phase == 2 && rp == 2);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.f;
            if (c < C) {
                const float* wk = w + ((size_t)(k0 + j) * C + c) * 9;
                v[j] = wk[(2 - rp) * 3 + (2 - sp)];
                if (fold) v[j] += wk[rp * 3 + (2 - sp)];   // + wf[2 - rp][sp]
            }
        }
        const float wscale = (np == 2 && c < C) ? pow2_scale(rowmax[c]) : 1.f;
        pack_store8(A + (size_t)phase * per_phase, e, per_piece, np, wscale, v);
    }
}

template <int MODE, int BM, int NP, typename TA>
__global__ void __launch_bounds__(BM * 2) bsplit_conv_fwd_kernel(BsplitArgs a) {
    static_assert((NP == 3 && sizeof(TA) == 4) || (NP == 1 && sizeof(TA) == 2), "3 pieces of fp32 tensors, or bf16 tensors as they are");
    constexpr unsigned ES = sizeof(TA);
    constexpr bool REFLECT = MODE == BS_FWD_REFLECT;
    constexpr bool DGRAD = MODE == BS_DGRAD_REFLECT;
    constexpr bool WGRAD = MODE == BS_WGRAD;
    constexpr int NT = BM * 2;              // threads
    constexpr int KB = 2048 / NT;           // channels of one pixel a thread gathers per stage (8 or 4)
    constexpr unsigned ASTAGE = BM * 32;    // bytes of one stage of one piece of the weights
    __shared__ __attribute__((aligned(16))) bf16x8 As[2][NP][2 * BM];   // [buffer][piece][half * BM + row]
    __shared__ __attribute__((aligned(16))) bf16x8 Bs[2][NP][256];      // [buffer][piece][half * 128 + pixel]
    __shared__ unsigned offT[BS_MAXTAP][128];

    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wp = wave & 1;
    const int mt = blockIdx.x % a.nMt;
    int pt = blockIdx.x / a.nMt;
    int phase = 0;
    if (DGRAD) {
        phase = (pt >= a.tstart[1]) + (pt >= a.tstart[2]);
        pt -= a.tstart[phase];
    }
    const int Hs = DGRAD ? (phase == 0 ? a.H - 2 : 1) : a.P;      // rows per image of this phase's pixel list
    const int T = a.R * a.S, PQ = WGRAD ? a.C * T : Hs * a.Q, Ptot = WGRAD ? PQ : a.N * PQ;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;          // WGRAD: padded planes of xpad
    const int HW4 = a.H * a.W * (int)ES;      // bytes of one channel plane
    const int pl = tid & 127;
    const int kq = __builtin_amdgcn_readfirstlane(tid >> 7);      // which KB-channel slice of the 16-channel stage
    const int half = (kq * KB) >> 3;

    // gather offsets (bytes, channel 0) of this workgroup's 128 pixels for every tap
    {
        const int pg = pt * 128 + pl;
        const bool pv = pg < Ptot;
        const int n = pv ? pg / PQ : 0, rem = pv ? pg - n * PQ : 0;
        int py = rem / a.Q;
        const int px = rem - py * a.Q;
        if (DGRAD) py = phase == 0 ? (py == 0 ? 0 : (py == Hs - 1 ? a.H - 1 : py + 1)) : (phase == 1 ? 1 : a.H - 2);
        const unsigned nbase = (unsigned)n * (unsigned)a.C * (unsigned)(a.H * a.W);
        if (WGRAD) {   // column (c, r, s) -> offset of xpad[0][c][r][s]; the reduction part comes through the scalar offset
            const int c = pg / T, tap = pg - c * T, r = tap / a.S, sx = tap - r * a.S;
            if (kq == 0) offT[0][pl] = pv ? (unsigned)((c * Hp + r) * Wp + sx) * ES : BS_OOB;
        }
        for (int t = kq; !WGRAD && t < T; t += NT / 128) {
            const int r = t / a.S, s = t - r * a.S;
            int iy = py - a.pad + r, ix = px - a.pad + s;
            bool ok = pv;
            if (REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
            }
            offT[t][pl] = ok ? (nbase + (unsigned)(iy * a.W + ix)) * ES : BS_OOB;
            if (DGRAD) {   // column mirror: column 1 also receives padded column -1 (source column 0 through tap s'=2), column W-2 padded column W
                const int ix2 = (px == 1 && s == 2) ? 0 : ((px == a.W - 2 && s == 0) ? a.W - 1 : -1);
                const bool ok2 = pv & (ix2 >= 0) & ((unsigned)iy < (unsigned)a.H);
                offT[9 + t][pl] = ok2 ? (nbase + (unsigned)(iy * a.W + ix2)) * ES : BS_OOB;
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, (int)a.x_bytes, 0x00020000);
    // one activation element: its fp32 value -- except on the one-piece route without mirror sums (RAW), where the stored
    // bf16 pattern goes to LDS as it is (zero-extended here, truncated again in stash: no shift, no conversion instruction)
    constexpr bool RAW = NP == 1 && !DGRAD;
    auto ldx = [&](unsigned voff, unsigned soff) -> float {
        if constexpr (ES == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rX, voff, soff, 0));
        else if constexpr (RAW) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rX, voff, soff, 0));
        else return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rX, voff, soff, 0) << 16);
    };
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)a.a_bytes, 0x00020000);
    const unsigned piece_bytes = (unsigned)a.nMt * (unsigned)a.nst * ASTAGE;
    const unsigned a_tile = (DGRAD ? (unsigned)phase * a.phase_bytes : 0u) + (unsigned)mt * (unsigned)a.nst * ASTAGE;

    struct Stage {
        u32x4 ap[NP];
        float b[KB];
        float b2[DGRAD ? KB : 1];     // column-mirror source (two lanes per image row are in range)
    };
    const int nst_here = WGRAD ? min(a.nst_split, a.nst - (int)blockIdx.y * a.nst_split) : a.nst;
    const int st0 = WGRAD ? (int)blockIdx.y * a.nst_split : 0;
    auto load = [&](Stage& r, int s) {
        const bool live = s < nst_here;
        const int gs = st0 + (live ? s : 0);
        const unsigned avo = live ? (unsigned)tid * 16u : BS_OOB;
        const unsigned aso = a_tile + (unsigned)gs * ASTAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p) r.ap[p] = __builtin_amdgcn_raw_buffer_load_b128(rA, avo, aso + p * piece_bytes, 0);
        if (WGRAD) {   // 16 consecutive x of image n, row y: scalar offset of xpad[n][0][y][x0], the thread's KB values are consecutive
            const int e0 = gs * 16, hw = a.H * a.W;
            const int n = e0 / hw, rem = e0 - n * hw, y = rem / a.W, x0 = rem - y * a.W;
            const unsigned bvo = live ? offT[0][pl] : BS_OOB;
            const unsigned bso = (unsigned)(((n * a.C) * Hp + y) * Wp + x0 + kq * KB) * ES;
#pragma unroll
            for (int j = 0; j < KB; ++j) r.b[j] = ldx(bvo, bso + j * ES);
            return;
        }
        const int cc = gs / T, tap = gs - cc * T;
        const unsigned bvo = live ? offT[tap][pl] : BS_OOB;
        const unsigned bso = live ? (unsigned)(cc * 16 + kq * KB) * (unsigned)HW4 : 0u;
#pragma unroll
        for (int j = 0; j < KB; ++j) r.b[j] = ldx(bvo, bso + j * HW4);
        if (DGRAD) {
            const unsigned bvo2 = live ? offT[9 + tap][pl] : BS_OOB;
#pragma unroll
            for (int j = 0; j < KB; ++j) r.b2[j] = ldx(bvo2, bso + j * HW4);
        }
    };
    auto stash = [&](const Stage& r, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(&As[buf][p][tid]) = r.ap[p];
        typedef __bf16 bfv __attribute__((ext_vector_type(KB)));
        bfv h, m, l;
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const float v = DGRAD ? r.b[j] + r.b2[j] : r.b[j];
            if constexpr (NP == 3) {
                __bf16 x, y, z;
                split3(v, x, y, z);
                h[j] = x;
                m[j] = y;
                l[j] = z;
            } else if constexpr (RAW) {
                h[j] = __builtin_bit_cast(__bf16, (unsigned short)__float_as_uint(v));
            } else {
                h[j] = (__bf16)v;      // the mirror sum of the data gradient is rounded once
            }
        }
        // this thread's KB consecutive k of pixel pl: offset (kq * KB) % 8 inside the pixel's 8-wide half
        const int sub = (kq * KB) & 7;
        *reinterpret_cast<bfv*>(reinterpret_cast<__bf16*>(&Bs[buf][0][half * 128 + pl]) + sub) = h;
        if constexpr (NP == 3) {
            *reinterpret_cast<bfv*>(reinterpret_cast<__bf16*>(&Bs[buf][1][half * 128 + pl]) + sub) = m;
            *reinterpret_cast<bfv*>(reinterpret_cast<__bf16*>(&Bs[buf][2][half * 128 + pl]) + sub) = l;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Operands {
        bf16x8 A[NP][2], B[NP][2];
    };
    auto fetch = [&](Operands& o, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o.A[p][i] = As[buf][p][hi * BM + wm * 64 + i * 32 + lo];
                o.B[p][i] = Bs[buf][p][hi * 128 + wp * 64 + i * 32 + lo];
            }
    };
    auto mma = [&](const Operands& o) {
        // smallest terms first: (l,h) (h,l) (m,m) | (m,h) (h,m) | (h,h); the four accumulators take turns, so consecutive
        // MFMAs are independent
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = (NP == 3 ? 0 : 5); q < 6; ++q)       // one piece: only the (h, h) product
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[PA[q]][i], o.B[PB[q]][j], acc[i][j], 0, 0, 0);
    };

    // Software pipeline, stage pair unrolled so that every register set is static:
    //   global loads run three stages ahead of the MFMAs (two register sets r0 / r1),
    //   LDS holds stages s+1 and s+2 while the MFMAs of stage s run out of registers (two operand sets),
    //   so the LDS reads of the next stage and the writes of the one after sit under the matrix instructions; one barrier per stage.
    // An odd stage count is rounded up: dead stages load zeros (out-of-range offsets) and add nothing.
    Stage r0, r1;
    Operands oa, ob;
    load(r0, 0);
    load(r1, 1);
    stash(r0, 0);
    __syncthreads();
    load(r0, 2);
    fetch(oa, 0);
    stash(r1, 1);
    __syncthreads();
    load(r1, 3);
    const int nst2 = (nst_here + 1) & ~1;
    // issue order inside a stage (a hint the scheduler follows where dependences allow): every MFMA is followed by its share of
    // the other work -- LDS reads of the next stage first, then the split arithmetic and LDS writes of the stage after, then the
    // global loads three stages ahead (0.178 -> 0.171 ms)
    auto interleave = [&]() {
        if constexpr (NP == 3) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                if (q < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                 // two VALU
                if (q >= 12 && q < 18) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one LDS write
                if (q >= 14 && q < 14 + 3 + KB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one global load
            }
        } else {   // 4 MFMAs per stage: one LDS read, then the two LDS writes / the global loads behind each of them
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (q < 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, (1 + KB * (DGRAD ? 2 : 1) + 3) / 4, 0);
            }
        }
    };
    for (int s = 0; s < nst2; s += 2) {
        fetch(ob, 1);         // operands of stage s+1
        mma(oa);              // stage s
        stash(r0, 0);         // stage s+2 -> buffer 0 (its stage s was read before the last barrier)
        load(r0, s + 4);
        interleave();
        __syncthreads();
        fetch(oa, 0);         // operands of stage s+2
        mma(ob);              // stage s+1
        stash(r1, 1);         // stage s+3 -> buffer 1
        load(r1, s + 5);
        interleave();
        __syncthreads();
    }

    // epilogue: acc[i][j][r] = Y[m0 + wm*64 + i*32 + (r/4)*8 + hi*4 + r%4][pixel wp*64 + j*32 + lo]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pg = pt * 128 + wp * 64 + j * 32 + lo;
        if (pg >= Ptot) continue;
        const int n = pg / PQ;
        int rem = pg - n * PQ;
        if (DGRAD) {
            const int sy = rem / a.Q, x = rem - sy * a.Q;
            const int y = phase == 0 ? (sy == 0 ? 0 : (sy == Hs - 1 ? a.H - 1 : sy + 1)) : (phase == 1 ? 1 : a.H - 2);
            rem = y * a.Q + x;
        }
        const int PQo = WGRAD ? PQ : a.P * a.Q;
        const size_t yo = (WGRAD ? (size_t)blockIdx.y : (size_t)n) * a.M * PQo + rem;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + wm * 64 + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                if (m < a.M) {
                    const float v = act_apply(acc[i][j][r] + (a.bias ? a.bias[m] : 0.f), a.act, a.slope);
                    if constexpr (WGRAD) ((float*)a.Y)[yo + (size_t)m * PQo] = v;      // fp32 partial sums
                    else st1((TA*)a.Y + yo + (size_t)m * PQo, v);
                }
            }
    }
}

// ---- "halo" form of the reflection-padded 3x3 stride-1 convolution (forward and data gradient), image width 32 or 64 -----------
// The kernel above gathers every input element once per tap (9 loads, 9 splits, 9 LDS writes per element and workgroup).  Here a
// pixel tile is RT = 128 / W full image rows, and per 16-channel chunk its (RT + 2) x (W + 2) window is loaded, split and written
// to LDS ONCE; the nine taps read it at shifted addresses ([piece][k half][window pixel][8 bf16]: a tap is a constant added to the
// lane's LDS address, 32 consecutive lanes still read 512 contiguous bytes).  The packed weights and the MFMA order are those of the
// kernel above; a wave owns 32 output channels x all 128 pixels and loads its weight fragments from memory into registers (no
// weight stage in LDS, one barrier per 16-channel chunk: see the A operand below).
//   forward        the window holds the reflection-padded input (the mirror is applied when the window is built);
//   data gradient  the window holds dy with a ring of zeros; the contributions of the padded rows / columns -1 and H / W, which
//                  fold onto rows / columns 1 and H-2 / W-2, become two extra window rows and columns of SUMS
//                  (dy[2] + dy[0] for output row 1 through tap r' = 2, dy[H-3] + dy[H-1] for row H-2 through r' = 0; columns
//                  alike; the corners sum four sources): a lane on row 1 / column 1 / ... reads the sum entry instead of the shifted
//                  one.  One set of plain flipped weights serves every row (the kernel above needs three row classes).
enum { BH_FWD = 0, BH_DGRAD = 1 };

struct HaloArgs {
    const void* X;       // [N][C][H][W], storage type TA (data gradient: dy, C = the convolution's output channels)
    const void* A;       // packed weights [piece][M tile][chunk * 9 + tap][k half][256 rows][8 bf16]
    const float* bias;   // [M] or null
    void* Y;             // [N][M][H][W]
    int N, C, H, M, nMt, nch, act;
    float slope;
    unsigned x_bytes, a_bytes;
    const float* x_amax;   // PK_F16X2: x_namax partial maxima of |X| (device), and the largest magnitude of every weight ROW [M]
    const float* w_amax;   //           (the pack call scaled row m by pow2_scale(w_amax[m]); the epilogue divides by it)
    int x_namax;
    unsigned* ovf;         // non-finite sentinel (common.h), PK_F16X2 only; may be null
    const void* R;         // optional [N][M][H][W] tensor added to the result (after bias / activation): the skip connection's gradient
                           // summed into the data gradient of a residual block's first convolution (autograd's `grad +=` pass)
};

// piece kinds: what an operand element becomes on its way to the matrix pipe
enum { PK_BF16X3 = 0,      // fp32 tensors, three bf16 pieces, six products (exact terms above 2^-24)
       PK_BF16 = 1,        // bf16 tensors as they are, one product
       PK_F16X2 = 2 };     // fp32 tensors, two scaled fp16 pieces, three products

template <int MODE, int PK, typename TA, int QW>
__global__ void __launch_bounds__(512) bsplit_halo_kernel(HaloArgs a) {
    constexpr int NP = PK == PK_BF16X3 ? 3 : (PK == PK_F16X2 ? 2 : 1);
    static_assert((PK != PK_BF16 && sizeof(TA) == 4) || (PK == PK_BF16 && sizeof(TA) == 2), "pieces of fp32 tensors, or bf16 tensors as they are");
    static_assert(QW == 32 || QW == 64, "image width");
    constexpr unsigned ES = sizeof(TA);
    constexpr bool DG = MODE == BH_DGRAD;
    constexpr int BM = 256, NT = 512;
    constexpr int RT = 128 / QW;                    // image rows of a pixel tile
    constexpr int WR = RT + 2, WC = QW + 2;         // window = tile + a ring of one pixel
    constexpr int HR = WR + (DG ? 2 : 0), QH = WC + (DG ? 2 : 0);     // + the two sum rows / columns of the data gradient
    constexpr int NPX = HR * QH;
    constexpr int NBASE = WR * WC, NRB = (NBASE * 4 + NT - 1) / NT;   // window entries x 4 channel quads, rounds over the threads
    constexpr int NPATCH = DG ? 2 * WR + 2 * QH : 0, NRP = DG ? (NPATCH * 4 + NT - 1) / NT : 0;
    constexpr unsigned ASTAGE = BM * 32;            // bytes of one stage of one piece of the weights
    constexpr unsigned XPIECE = 2 * NPX * 16, XBUF = NP * XPIECE;
    __shared__ __attribute__((aligned(16))) bf16x8 Xs[2 * NP * 2 * NPX];  // [buffer][piece][half][window entry]
    __shared__ float red_scratch[16];

    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NI = 1, NJ = 4;                   // a wave's tile: 32 output channels x all 128 pixels (its weight fragments are its own: no wave loads another's)
    const int mt = blockIdx.x % a.nMt, pt = blockIdx.x / a.nMt;
    const int TPI = a.H / RT;
    const int n = pt / TPI, y0 = (pt - n * TPI) * RT;
    const int HW = a.H * QW;
    char* const xs_bytes = reinterpret_cast<char*>(&Xs[0]);

    // ---- window builder tables: which source element(s) a thread loads per chunk and where their pieces go
    unsigned bvo[NRB], blds[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        const int u = tid + i * NT;
        const bool valid = u < NBASE * 4;
        const int q = u / NBASE, e = u - q * NBASE, hb = e / WC, wb = e - hb * WC;
        int iy = y0 - 1 + hb, ix = wb - 1;
        bool ok = valid;
        if (DG) {
            ok = ok & ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)QW);
        } else {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= QW ? 2 * (QW - 1) - ix : ix;
        }
        bvo[i] = ok ? (unsigned)(q * 4 * HW + iy * QW + ix) * ES : BS_OOB;
        blds[i] = valid ? (unsigned)(((q >> 1) * NPX + hb * QH + wb) * 16 + (q & 1) * 8) : 0xffffffffu;
    }
    unsigned pvo[DG ? NRP : 1][4], plds[DG ? NRP : 1];
    if constexpr (DG) {
        const bool use_lo = y0 <= 1 && 1 < y0 + RT;               // the tile holds row 1: sum row {2, 0}
        const bool use_hi = y0 <= a.H - 2 && a.H - 2 < y0 + RT;   // the tile holds row H-2: sum row {H-3, H-1}
#pragma unroll
        for (int i = 0; i < NRP; ++i) {
            const int u = tid + i * NT;
            const bool valid = u < NPATCH * 4;
            const int q = u / NPATCH, pe = u - q * NPATCH;
            int h, w;
            if (pe < 2 * WR) {
                h = pe >> 1;
                w = WC + (pe & 1);
            } else {
                const int p2 = pe - 2 * WR;
                h = WR + p2 / QH;
                w = p2 - (p2 / QH) * QH;
            }
            int r0, r1, c0, c1;
            if (h < WR) {
                r0 = y0 - 1 + h;
                r0 = (unsigned)r0 < (unsigned)a.H ? r0 : -1;
                r1 = -1;
            } else if (h == WR) {
                r0 = use_hi ? a.H - 3 : -1;
                r1 = use_hi ? a.H - 1 : -1;
            } else {
                r0 = use_lo ? 2 : -1;
                r1 = use_lo ? 0 : -1;
            }
            if (w < WC) {
                c0 = w - 1;
                c0 = (unsigned)c0 < (unsigned)QW ? c0 : -1;
                c1 = -1;
            } else if (w == WC) {
                c0 = QW - 3;
                c1 = QW - 1;
            } else {
                c0 = 2;
                c1 = 0;
            }
            const int rr[4] = {r0, r0, r1, r1}, cc[4] = {c0, c1, c0, c1};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pvo[i][k] = (valid && rr[k] >= 0 && cc[k] >= 0) ? (unsigned)(q * 4 * HW + rr[k] * QW + cc[k]) * ES : BS_OOB;
            plds[i] = valid ? (unsigned)(((q >> 1) * NPX + h * QH + w) * 16 + (q & 1) * 8) : 0xffffffffu;
        }
    }

    // ---- LDS addresses of this lane's two pixel columns (j = 0, 1) of the B operand, per tap
    unsigned boff[NJ], rowb[DG ? NJ : 1][3], colb[DG ? NJ : 1][3];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = j * 32 + lo, ty = p / QW, tx = p - ty * QW;
        boff[j] = (unsigned)((hi * NPX + ty * QH + tx) * 16);
        if constexpr (DG) {
            const int y = y0 + ty;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int rs = (y == 1 && t == 2) ? WR + 1 : ((y == a.H - 2 && t == 0) ? WR : ty + t);
                const int cs = (tx == 1 && t == 2) ? WC + 1 : ((tx == QW - 2 && t == 0) ? WC : tx + t);
                rowb[j][t] = (unsigned)(rs * QH * 16);
                colb[j][t] = (unsigned)((hi * NPX + cs) * 16);
            }
        }
    }

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.X), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.A), 0, (int)a.a_bytes, 0x00020000);
    // an activation element as loaded: fp32 value; bf16 storage: the stored pattern, zero-extended (RAW: it goes to LDS unchanged)
    auto ldraw = [&](unsigned voff, unsigned soff) -> unsigned {
        if constexpr (ES == 4) return __builtin_amdgcn_raw_buffer_load_b32(rX, voff, soff, 0);
        else return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rX, voff, soff, 0);
    };
    auto tofloat = [&](unsigned raw) -> float { return __uint_as_float(ES == 4 ? raw : raw << 16); };
    const unsigned piece_bytes = (unsigned)a.nMt * (unsigned)(a.nch * 9) * ASTAGE;
    const unsigned a_tile = (unsigned)mt * (unsigned)(a.nch * 9) * ASTAGE;
    const int nst = a.nch * 9;

    // write 4 consecutive channels of one window entry (8 bytes per piece)
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
    float sx = 1.f;
    if constexpr (PK == PK_F16X2) {      // largest of the partial maxima the producer left (one per plane, or a single value)
        const float m = thread_max_of_partials(a.x_amax, a.x_namax, tid, NT);
        sx = pow2_scale(block_max(m, red_scratch));
        __syncthreads();
    }
    auto put_split = [&](unsigned lds, int buf, const float (&v)[4]) {
        char* dst = xs_bytes + (unsigned)buf * XBUF + lds;
        if constexpr (PK == PK_F16X2) {
            hf4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 x, y;
                split2h(v[j] * sx, x, y);
                h[j] = x;
                l[j] = y;
            }
            *reinterpret_cast<hf4*>(dst) = h;
            *reinterpret_cast<hf4*>(dst + XPIECE) = l;
        } else {
            bf4 h, m, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (NP == 3) {
                    __bf16 x, y, z;
                    split3(v[j], x, y, z);
                    h[j] = x;
                    m[j] = y;
                    l[j] = z;
                } else {
                    h[j] = (__bf16)v[j];
                }
            }
            *reinterpret_cast<bf4*>(dst) = h;
            if constexpr (NP == 3) {
                *reinterpret_cast<bf4*>(dst + XPIECE) = m;
                *reinterpret_cast<bf4*>(dst + 2 * XPIECE) = l;
            }
        }
    };
    unsigned tb[NRB][4];
    auto base_load = [&](int ch) {       // (a chunk past the end: out-of-range offsets, zeros come back)
        const bool live = ch < a.nch;
        const unsigned so = live ? (unsigned)((n * a.C + ch * 16) * HW) * ES : 0u;
#pragma unroll
        for (int i = 0; i < NRB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) tb[i][j] = ldraw(live ? bvo[i] : BS_OOB, so + (unsigned)(j * HW) * ES);
    };
    auto base_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NRB; ++i) {
            if (blds[i] == 0xffffffffu) continue;
            if constexpr (PK == PK_BF16) {     // stored bf16 patterns as they are
                typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                us4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (unsigned short)tb[i][j];
                *reinterpret_cast<us4*>(xs_bytes + (unsigned)buf * XBUF + blds[i]) = v;
            } else {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tofloat(tb[i][j]);
                put_split(blds[i], buf, v);
            }
        }
    };
    unsigned tp[DG ? NRP : 1][4][4];
    auto patch_load = [&](int ch) {
        if constexpr (DG) {
            const bool live = ch < a.nch;
            const unsigned so = live ? (unsigned)((n * a.C + ch * 16) * HW) * ES : 0u;
#pragma unroll
            for (int i = 0; i < NRP; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) tp[i][k][j] = ldraw(live ? pvo[i][k] : BS_OOB, so + (unsigned)(j * HW) * ES);
        }
    };
    auto patch_write = [&](int buf) {
        if constexpr (DG) {
#pragma unroll
            for (int i = 0; i < NRP; ++i) {
                if (plds[i] == 0xffffffffu) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = (tofloat(tp[i][0][j]) + tofloat(tp[i][1][j])) + (tofloat(tp[i][2][j]) + tofloat(tp[i][3][j]));
                put_split(plds[i], buf, v);
            }
        }
    };

    // the A operand (weights) never touches LDS: the packed image holds, per stage and piece, [k half][256 rows][8 values] -- exactly
    // the 16 bytes a lane feeds to the MFMA (row wave * 32 + lo, k half hi) -- so every wave loads ITS fragments from memory
    // (32 rows x 16 bytes contiguous per half-wave; every byte of the image is loaded by exactly one wave of the workgroup) two or three stages ahead.
    // No weight stage in LDS means no barrier per stage (one per 16-channel chunk, for the window), none of this chip's slow LDS stores
    // (~80 B/clk), half the LDS reads, and waves that drift apart so that one wave's window work sits under another's MFMAs.
    struct OpA {
        bf16x8 A[NP][NI];
    };
    const unsigned a_lane = (unsigned)((hi * BM + wave * 32 + lo) * 16);
    auto aload = [&](OpA& o, int s) {
        const bool live = s < nst;
        const unsigned avo = live ? a_lane : BS_OOB;
        const unsigned aso = a_tile + (unsigned)(live ? s : 0) * ASTAGE;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                o.A[p][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rA, avo, aso + p * piece_bytes + i * 512, 0));
    };

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct OpB {
        bf16x8 B[NP][NJ];
    };
    // B operand of one stage: pixels of tap (tr, ts) from window buffer xbuf
    auto fetch = [&](OpB& o, int xbuf, int tr, int ts) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            unsigned ad;
            if constexpr (DG) ad = rowb[j][tr] + colb[j][ts];
            else ad = boff[j] + (unsigned)((tr * QH + ts) * 16);
#pragma unroll
            for (int p = 0; p < NP; ++p)
                o.B[p][j] = *reinterpret_cast<const bf16x8*>(xs_bytes + ad + (unsigned)xbuf * XBUF + p * XPIECE);
        }
    };
    auto mma = [&](const OpA& oa, const OpB& ob) {
        if constexpr (PK == PK_F16X2) {       // (l,h) (h,l) (h,h)
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, oa.A[PA[q]][i]),
                                                                            __builtin_bit_cast(f16x8, ob.B[PB[q]][j]), acc[i][j], 0, 0, 0);
        } else {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = (NP == 3 ? 0 : 5); q < 6; ++q)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa.A[PA[q]][i], ob.B[PB[q]][j], acc[i][j], 0, 0, 0);
        }
    };
    // issue order inside a stage (a hint): every MFMA is followed by its share of the other work
    auto interleave = [&]() {
        if constexpr (NP == 2) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                if (q < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // LDS reads of the next stage first
                if (q < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // weight fragments two stages ahead
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                   // VALU (split arithmetic of a window)
                if (q >= 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // LDS writes
                if (q >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);       // window loads
            }
        } else if constexpr (NP == 3) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                if (q < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // LDS reads of the next stage first
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                   // VALU (split arithmetic of a window)
                if (q >= 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // LDS writes
                if (q >= 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // global loads
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
            }
        }
    };

    // window of chunk 0, then the stage loop unrolled over a PAIR of chunks (18 stages) so that window buffer, tap and register sets
    // are all static.  Stage t of the pair: MFMAs of stage t out of registers | B fetch (LDS) of stage t+1 | weight fragments of
    // stage t+NA-1 from memory | window of the next chunk:
    //   t = 0 / 9    loads of the window proper                       (registers only)
    //   t = 3 / 12   its split + LDS writes; loads of the sum rows / columns (data gradient)
    //   t = 6 / 15   their split + LDS writes
    // ONE barrier per chunk, at the end of t = 7 / 16: behind every wave's window writes and its last fetch from the window before
    // (during t = 7 / 16), in front of the first fetch from the new window (during t = 8 / 17) and of the next writes (t = 3 / 12).
    constexpr int NA = NP == 3 ? 2 : 3;       // register sets of weight fragments (18 = 0 mod NA)
    base_load(0);
    patch_load(0);
    OpA oa[NA];
    OpB ob[2];
#pragma unroll
    for (int i = 0; i < NA - 1; ++i) aload(oa[i], i);
    base_write(0);
    patch_write(0);
    __syncthreads();
    fetch(ob[0], 0, 0, 0);
    for (int c = 0; c < a.nch; c += 2) {
        const int s0 = c * 9;
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            const int tn = (t + 1) % 18, tapn = tn % 9;
#ifndef HALO_ABL
#define HALO_ABL 0
#endif
            if (HALO_ABL != 2) aload(oa[(t + NA - 1) % NA], s0 + t + NA - 1);                   // weight fragments of stage t+NA-1
            if (HALO_ABL != 3) fetch(ob[(t + 1) & 1], tn / 9, tapn / 3, tapn % 3);              // pixels of stage t+1
            if (HALO_ABL != 4) mma(oa[t % NA], ob[t & 1]);                                      // stage t
            if (t == 0) base_load(c + 1);
            if (t == 9) base_load(c + 2);
            if (t == 3 || t == 12) {
                base_write(t == 3 ? 1 : 0);
                patch_load(t == 3 ? c + 1 : c + 2);
            }
            if (t == 6 || t == 15) patch_write(t == 6 ? 1 : 0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);      // nothing moves across a stage boundary (MFMAs of the next stage would wait on its own LDS reads)
            if (t == 7 || t == 16) {
                if (HALO_ABL != 1) __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // epilogue: acc[i][j][r] = Y[m0 + wave*32 + (r/4)*8 + hi*4 + r%4][pixel j*32 + lo]; a tile is RT full rows of image n
    const float isx = 1.f / sx;    // powers of two: exact
    float iswr[NI][16];            // ... and one per weight ROW (the pack call scaled row m by pow2_scale(w_amax[m]))
    if constexpr (PK == PK_F16X2) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + wave * 32 + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                iswr[i][r] = m < a.M ? 1.f / pow2_scale(a.w_amax[m]) : 1.f;
            }
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const size_t yo = (size_t)n * a.M * HW + (size_t)y0 * QW + j * 32 + lo;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + wave * 32 + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                if (m < a.M) {
                    const float av = PK == PK_F16X2 ? (acc[i][j][r] * isx) * iswr[i][r] : acc[i][j][r];
                    if constexpr (PK == PK_F16X2) bad |= is_nonfinite(av);
                    float v = act_apply(av + (a.bias ? a.bias[m] : 0.f), a.act, a.slope);
                    if (a.R) v += ld1((const TA*)a.R + yo + (size_t)m * HW);
                    st1((TA*)a.Y + yo + (size_t)m * HW, v);
                }
            }
    }
    if constexpr (PK == PK_F16X2) report_nonfinite(a.ovf, bad);
}

// ---- fp16 two-piece route (fp32 tensors) / bf16 one-product route (bf16 tensors): weight gradient of a padded convolution -----------
// dW[k][(c, r, s)] = sum over (n, y, x) of dy[n][k][y][x] * xpad[n][c][y * STRIDE + r][x * STRIDE + s]: rows = output channels (one tile of
// BM = 128 or 256), columns = (c, tap) in tiles of 128, reduction over output pixels in stages of 16 consecutive x (output width a
// multiple of 16).  xpad = the input with its padding materialised once (reflection or zeros), so that the gather address separates into
// a column part (lane offset) and a pixel part (scalar offset).  Both operands are split on their way to LDS (no packed copy of dy):
//   A  dy[n][row][y][x0 + 8 half .. + 8): two 16-byte loads per thread, one 16-byte LDS write per piece;
//   B  xpad[n][c][y STRIDE + r][(x0 + KB q + j) STRIDE + s], j < KB = 2048 / threads: KB element loads (the tap shifts the alignment), one
//      8- or 16-byte LDS write per piece;
// 12 MFMAs (fp16 route; bf16: 4) and 8 (4) ds_read_b128 per wave and stage; blockIdx.y takes a range of stages and writes a raw partial
// sum, combined in a fixed order by bsplit_wgrad_reduce_kernel.  Replaces autograd's weight gradient of nn.Conv2d /
// nn.ConvTranspose2d of the generator's down / up-sampling layers, the residual blocks and the PatchGAN (models/networks.py:584-648, 734-763).
struct HWgradArgs {
    const void* XP;        // padded input [N][C][Hp][Wp], storage type TA
    const void* DY;        // [N][K][P][Q]
    float* part;           // [splits][K][C * T]
    int N, C, K, P, Q, Hp, Wp, R, S, nst, nst_split;
    unsigned xp_bytes, dy_bytes;
    const float* x_amax;   // fp16 route: partial maxima of |x| and |dy| (device)
    const float* dy_amax;
    int x_namax, dy_namax;
    int ntile, nwg;        // column tiles, workgroups that have work (the grid is padded to a multiple of 8)
    unsigned* ovf;         // non-finite sentinel (common.h), fp32 tensors only; may be null
    int reflect_inline;    // stride 1, 3x3, reflection padding 1: XP is the UNPADDED input (Hp = H, Wp = W) and the mirror is applied
                           // in the gather -- a per-stage row select and one register move at the two image edges -- instead of by a padded copy
    int splits, nmt;       // splits of the pixel reduction; row tiles of BM output channels (K > 256: the PatchGAN's 512-channel layer)
    int Qs, pad;           // GEN kernels: Q rounded up to a multiple of 16 (the stages of a row; dy beyond column Q enters as zero) and the
                           // ZERO padding (0 or 1) applied inside the gather -- XP is the unpadded input (Hp = H, Wp = W), no padded copy
};

// NC = column tiles of 128 per workgroup (2 with the 256-row tile: the dy tile is loaded and split once for 256 columns)
// GEN (zero padding <= 1, fp32 or bf16 tensors): the padding is applied inside the gather (rows outside the image select an
// out-of-range offset, the at most one column per side is zeroed in registers when the run is split / stored) and the output width
// may be ragged (stages of 16 columns per output row, the dy values beyond column Q masked to zero: the PatchGAN's 15 x 15 layer;
// bf16 tensors then load dy by 2-byte elements, a ragged row starts on a 2-byte boundary)
template <int BM, int STRIDE, typename TA, int NC, int GEN = 0>
__global__ void __launch_bounds__(BM * 2) hsplit_wgrad_kernel(HWgradArgs a) {
    constexpr int NT = BM * 2;
    constexpr int CW = 128 * NC;                // columns per workgroup
    constexpr bool HALF = sizeof(TA) == 2;      // bf16 tensors: one piece, one product, no scaling
    constexpr int NP = HALF ? 1 : 2;
    constexpr unsigned ES = sizeof(TA);
    constexpr int KB = 16 * CW / NT;            // consecutive output pixels of its column a thread gathers per stage (4 or 8)
    // the two k halves of a row are written by neighbouring lanes: 128 bytes of padding between the halves put them on disjoint banks
    constexpr int AH = BM + 8;
    __shared__ __attribute__((aligned(16))) bf16x8 As[2][NP][2 * AH];     // [buffer][piece][half * AH + row]
    constexpr int BH = CW + 8;
    __shared__ __attribute__((aligned(16))) bf16x8 Bs[2][NP][2 * BH];     // [buffer][piece][half * BH + column]
    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wp = wave & 1;
    const int T = a.R * a.S, CT = a.C * T, PQ = a.P * a.Q;
    // workgroups go to the 8 XCDs round-robin: give each XCD a CONTIGUOUS run of (split, column tile) pairs, so that the column tiles
    // of one split -- which all read the same dy rows -- share one L2 (dispatch order put them on all eight: dy crossed the fabric 8x)
    const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int bx = wg % a.ntile, byz = wg / a.ntile;
    const int by = byz % a.splits, m0 = (byz / a.splits) * BM;      // (the column tiles of one (row tile, split) pair are neighbours: same dy rows)

    float sx = 1.f, sdy = 1.f;
    if constexpr (!HALF) {
        const float m = thread_max_of_partials(a.x_amax, a.x_namax, tid, NT), g = thread_max_of_partials(a.dy_amax, a.dy_namax, tid, NT);
        float* scratch = reinterpret_cast<float*>(&As[0][0][0]);
        sx = pow2_scale(block_max(m, scratch));
        sdy = pow2_scale(block_max(g, scratch));
        __syncthreads();
    }

    // loaders: neighbouring lanes read neighbouring bytes of one row / one column's pixel run (64 contiguous bytes per 4 lanes of fp32
    // data: a wave's load touches 16 lines, not 32-64 -- the vector memory path, not the matrix pipe, was the limit of this kernel).
    // fp32 A: rows tid / 4 and NT / 4 + tid / 4, floats 4 * (tid % 4) ..+3 of the stage's 16;  bf16 A: row tid / 2, 8 values
    // B: column tid / BT, pixels KB * (tid % BT) ..+KB-1
    constexpr int AR = NT / 4;
    constexpr int BT = 16 / KB;
    const int arow = HALF ? tid >> 1 : tid >> 2, aq = HALF ? (tid & 1) * 2 : tid & 3;
    const int ahalf = aq >> 1, asub = (aq & 1) * 4;
    const unsigned avo = m0 + arow < a.K ? (unsigned)((m0 + arow) * PQ + aq * 4) * ES : BS_OOB;
    const unsigned avo1 = (!HALF && m0 + arow + AR < a.K) ? (unsigned)((m0 + arow + AR) * PQ + aq * 4) * ES : BS_OOB;
    const int bcol = tid / BT, bq = tid % BT;
    const int col = bx * CW + bcol;
    unsigned bvo = BS_OOB;
    int tr = 1, fixl = 0, fixr = 0;      // inline reflection: this thread's tap row; whether its first / last element can fall on column -1 / W
    int lm = 0, rmk = 0;                 // GEN: elements of the run that lie in the zero padding in the first / last stage of a row (bit j), bit 8 = rotate
    if (col < CT) {
        const int c = col / T, tap = col - c * T, r = tap / a.S, s = tap - r * a.S;
        if constexpr (GEN) {
            const int cs = bq * KB * STRIDE + s - a.pad;      // column of the run's first element in the row's first stage (-1 at most)
            bvo = (unsigned)(((c * a.Hp + r) * a.Wp + cs) * (int)ES);      // (may wrap: its sum with the stage's part below does not)
            tr = r;
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                lm |= (cs + j * STRIDE < 0) ? 1 << j : 0;
                rmk |= ((a.Qs - 16) * STRIDE + cs + j * STRIDE >= a.Wp) ? 1 << j : 0;
            }
            // a run whose first element is column -1 is loaded one element late and rotated when it is split: in the first row of the
            // tensor its offset would be "-4", which does not wrap in the hardware's range check (every later element of the run,
            // reached through the instruction's immediate offset, would read as 0 too)
            if (cs < 0) lm |= 256;
        } else
        if (a.reflect_inline) {      // row term chosen per stage (ro0 / ro1 / ro2 below); a left-edge lane's run starts at column -1
            bvo = (unsigned)((c * a.Hp) * a.Wp + s - 1 + bq * KB) * ES;      // (may be "-4": the per-stage sum below is not)
            tr = r;
            fixl = (s == 0 && bq == 0);
            fixr = (s == 2 && bq == BT - 1);
        } else {
            bvo = (unsigned)((c * a.Hp + r) * a.Wp + s + bq * KB * STRIDE) * ES;
        }
    }
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.DY), 0, (int)a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.XP), 0, (int)a.xp_bytes, 0x00020000);

    const int st0 = by * a.nst_split;
    const int nst_here = min(a.nst_split, a.nst - st0);
    // position of the next stage to LOAD (scalar): image, output row, first output column
    int ln, ly, lx;
    {
        const int Qrow = GEN ? a.Qs : a.Q;      // stage positions count the (padded) row width
        const int e0 = st0 * 16;
        ln = e0 / (a.P * Qrow);
        const int rem = e0 - ln * (a.P * Qrow);
        ly = rem / Qrow;
        lx = rem - ly * Qrow;
    }
    struct Stage {
        unsigned a[HALF ? 4 : 8];     // 8 consecutive dy values of this thread's row (bf16: packed pairs)
        unsigned b[KB];               // KB consecutive output pixels of this thread's column
        int fix;                      // inline reflection: 1 = the run starts at column -1, loaded from column 0 instead (rotate), 2 = the last is column W
                                      // GEN: bit j = element j of the run lies in the zero padding, bit 8 = the run was loaded from column 0 (rotate)
        int am;                       // GEN: how many of this thread's 4 consecutive dy values lie inside the row (>= 4: all)
    };
    int lcount = 0;
#ifndef WG_ABL
#define WG_ABL 0      // timing-only ablations (results wrong by construction): 1 no global loads, 2 no split / LDS writes, 3 no LDS reads, 4 no MFMAs, 5 no barriers
#endif
    auto load = [&](Stage& r) {
        const bool live = WG_ABL == 1 ? false : lcount < nst_here;
        const unsigned aso = (unsigned)(ln * a.K * PQ + ly * a.Q + lx) * ES;
        unsigned bso = (unsigned)(((ln * a.C) * a.Hp + ly * STRIDE) * a.Wp + lx * STRIDE) * ES;
        unsigned bvt = bvo;
        r.fix = 0;
        r.am = 4;
        if constexpr (GEN) {
            const int yy = ly * STRIDE - a.pad;      // source row of tap row 0 (scalar)
            const bool rowok = (unsigned)(yy + tr) < (unsigned)a.Hp;
            r.fix = (lx == 0 ? lm : 0) | (lx == a.Qs - 16 ? rmk : 0);
            bvt = rowok ? bvo + (unsigned)((((ln * a.C) * a.Hp + yy) * a.Wp + lx * STRIDE) * (int)ES) + ((r.fix & 256) ? STRIDE * ES : 0u) : BS_OOB;
            bso = 0;
            r.am = a.Q - lx - aq * 4;
        } else
        if constexpr (STRIDE == 1) {
            if (a.reflect_inline) {      // source row of tap row tr under reflection padding 1 (scalars per stage), selected by the lane's tap row
                const int y0 = ly == 0 ? 1 : ly - 1, y2 = ly == a.Hp - 1 ? a.Hp - 2 : ly + 1;
                const unsigned base = (unsigned)((ln * a.C) * a.Hp * a.Wp + lx) * ES;
                const unsigned ro0 = base + (unsigned)(y0 * a.Wp) * ES, ro1 = base + (unsigned)(ly * a.Wp) * ES, ro2 = base + (unsigned)(y2 * a.Wp) * ES;
                r.fix = (lx == 0 && fixl) ? 1 : ((lx + 16 == a.Q && fixr) ? 2 : 0);
                // a run that would start at column -1 is loaded from column 0 and rotated when it is split (the first row of the tensor
                // has nothing in front of it, and an offset of "-4" does not wrap in the hardware's range check: the whole load would
                // return 0); a run that ends at column W reads one element of the next row -- in range, or zero at the very end
                bvt = bvo + (tr == 0 ? ro0 : (tr == 1 ? ro1 : ro2)) + (r.fix == 1 ? ES : 0u);
                bso = 0;
            }
        }
        const unsigned av = live ? avo : BS_OOB, bv = live ? bvt : BS_OOB;
        if constexpr (HALF) {
            if (GEN && a.Qs != a.Q) {      // ragged rows start on 2-byte boundaries: element loads, the values beyond column Q enter as zero
                const int am = a.Q - lx - aq * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned e0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rD, av, aso + (unsigned)(2 * i) * ES, 0);
                    const unsigned e1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rD, av, aso + (unsigned)(2 * i + 1) * ES, 0);
                    r.a[i] = (2 * i < am ? (e0 & 0xffffu) : 0u) | (2 * i + 1 < am ? e1 << 16 : 0u);
                }
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rD, av, aso, 0);
                r.a[0] = v.x; r.a[1] = v.y; r.a[2] = v.z; r.a[3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < KB; ++j) r.b[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rX, bv, bso + (unsigned)(j * STRIDE) * ES, 0);
        } else {
            const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rD, av, aso, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(rD, live ? avo1 : BS_OOB, aso, 0);
            r.a[0] = v0.x; r.a[1] = v0.y; r.a[2] = v0.z; r.a[3] = v0.w;
            r.a[4 % (HALF ? 4 : 8)] = v1.x; r.a[5 % (HALF ? 4 : 8)] = v1.y; r.a[6 % (HALF ? 4 : 8)] = v1.z; r.a[7 % (HALF ? 4 : 8)] = v1.w;
            if constexpr (STRIDE == 1) {      // KB consecutive floats (any 4-byte alignment): 16 bytes per load
#pragma unroll
                for (int j = 0; j < KB; j += 4) {
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rX, bv, bso + (unsigned)j * ES, 0);
                    r.b[j] = w.x; r.b[j + 1] = w.y; r.b[j + 2] = w.z; r.b[j + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < KB; ++j) r.b[j] = __builtin_amdgcn_raw_buffer_load_b32(rX, bv, bso + (unsigned)(j * STRIDE) * ES, 0);
            }
        }
        ++lcount;
        lx += 16;
        if (lx == (GEN ? a.Qs : a.Q)) {
            lx = 0;
            if (++ly == a.P) {
                ly = 0;
                ++ln;
            }
        }
    };
    // the thread's KB consecutive k (pixels) of column bcol: k half (bq * KB) / 8, offset (bq * KB) % 8 inside it
    const int bhalf = (bq * KB) >> 3, bsub = (bq * KB) & 7;
    // fp16 route: the pieces of a stage in registers (split), then to LDS (write) -- two steps, a barrier apart in the loop below
    typedef _Float16 hf4 __attribute__((ext_vector_type(4)));
    typedef _Float16 hfK __attribute__((ext_vector_type(KB)));
    struct Pieces {
        hf4 ah[2], al[2];      // rows arow and arow + AR: 4 values each
        hfK bh, bl;
    };
    auto split = [&](const Stage& r, Pieces& q) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 x, y;
                unsigned aj = r.a[(i * 4 + j) % (HALF ? 4 : 8)];
                if constexpr (GEN) aj = j < r.am ? aj : 0u;                // dy beyond the row's last column
                split2h(__uint_as_float(aj) * sdy, x, y);
                q.ah[i][j] = x;
                q.al[i][j] = y;
            }
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            unsigned bj = r.b[j];
            if constexpr (GEN) {
                bj = (r.fix & 256) ? (j == 0 ? 0u : r.b[j == 0 ? 0 : j - 1]) : bj;      // run loaded one element late
                bj = (r.fix >> j & 1) ? 0u : bj;                            // zero padding
            } else {
            if (j == 0) bj = r.fix == 1 ? r.b[1] : bj;                     // run loaded from column 0: wanted (col 1, col 0, col 1, col 2, ...)
            else bj = r.fix == 1 ? r.b[j - 1] : bj;
            if (j == KB - 1) bj = r.fix == 2 ? r.b[KB - 3] : bj;           // column W mirrors to column W - 2
            }
            _Float16 x, y;
            split2h(__uint_as_float(bj) * sx, x, y);
            q.bh[j] = x;
            q.bl[j] = y;
        }
    };
    auto write = [&](const Pieces& q, int buf) {
        if (WG_ABL == 2) {
            asm volatile("" :: "v"(q.ah[0]), "v"(q.bh));
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<hf4*>(reinterpret_cast<_Float16*>(&As[buf][0][ahalf * AH + arow + i * AR]) + asub) = q.ah[i];
            *reinterpret_cast<hf4*>(reinterpret_cast<_Float16*>(&As[buf][NP - 1][ahalf * AH + arow + i * AR]) + asub) = q.al[i];
        }
        *reinterpret_cast<hfK*>(reinterpret_cast<_Float16*>(&Bs[buf][0][bhalf * BH + bcol]) + bsub) = q.bh;
        *reinterpret_cast<hfK*>(reinterpret_cast<_Float16*>(&Bs[buf][NP - 1][bhalf * BH + bcol]) + bsub) = q.bl;
    };

    auto stash = [&](const Stage& r, int buf) {
        if (WG_ABL == 2) {
            asm volatile("" :: "v"(r.a[0]), "v"(r.b[0]));
            return;
        }
        if constexpr (!HALF) {
            Pieces q;
            split(r, q);
            write(q, buf);
        } else {                   // stored bf16 patterns as they are
            u32x4 v;
            v.x = r.a[0]; v.y = r.a[1]; v.z = r.a[2]; v.w = r.a[3];
            *reinterpret_cast<u32x4*>(&As[buf][0][ahalf * AH + arow]) = v;
            typedef unsigned short usK __attribute__((ext_vector_type(KB)));
            usK w;
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                unsigned bj = r.b[j];
                if constexpr (GEN) {
                    bj = (r.fix & 256) ? (j == 0 ? 0u : r.b[j == 0 ? 0 : j - 1]) : bj;      // run loaded one element late
                    bj = (r.fix >> j & 1) ? 0u : bj;                                         // zero padding
                } else if constexpr (STRIDE == 1) {                                          // inline reflection, as in split() above
                    if (j == 0) bj = r.fix == 1 ? r.b[1] : bj;
                    else bj = r.fix == 1 ? r.b[j - 1] : bj;
                    if (j == KB - 1) bj = r.fix == 2 ? r.b[KB - 3] : bj;
                }
                w[j] = (unsigned short)bj;
            }
            *reinterpret_cast<usK*>(reinterpret_cast<unsigned short*>(&Bs[buf][0][bhalf * BH + bcol]) + bsub) = w;
        }
    };
    constexpr int NJ = 2 * NC;                  // 32-column blocks of a wave (its half of the workgroup's columns)
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Operands {
        bf16x8 A[NP][2], B[NP][NJ];
    };
    auto fetch = [&](Operands& o, int buf) {
        if (WG_ABL == 3) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) o.A[p][i] = As[buf][p][hi * AH + wm * 64 + i * 32 + lo];
#pragma unroll
            for (int j = 0; j < NJ; ++j) o.B[p][j] = Bs[buf][p][hi * BH + wp * (CW / 2) + j * 32 + lo];
        }
    };
    auto mma = [&](const Operands& o) {
        if (WG_ABL == 4) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(o.A[p][i]));
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(o.B[p][j]));
            }
            return;
        }
        if constexpr (HALF) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[0][i], o.B[0][j], acc[i][j], 0, 0, 0);
        } else {      // (l,h) (h,l) (h,h)
            constexpr int PA[3] = {NP - 1, 0, 0}, PB[3] = {0, NP - 1, 0};
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, o.A[PA[q]][i]),
                                                                            __builtin_bit_cast(f16x8, o.B[PB[q]][j]), acc[i][j], 0, 0, 0);
        }
    };
    auto interleave = [&]() {
        if constexpr (HALF) {
#pragma unroll
            for (int q = 0; q < 4 * NC; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 12 * NC; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                if (q < 4 + 4 * NC) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // LDS reads of the next stage first
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                   // split arithmetic
                if (q >= 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // LDS writes
                if (q >= 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // global loads
            }
        }
    };

    // the software pipeline of the convolution kernels above: global loads three stages ahead, LDS one, operands in registers
    Stage rg[2];
    Operands op[2];
    load(rg[0]);
    load(rg[1]);
    stash(rg[0], 0);
    __syncthreads();
    load(rg[0]);
    fetch(op[0], 0);
    stash(rg[1], 1);
    __syncthreads();
    load(rg[1]);
    const int nst2 = (nst_here + 1) & ~1;
    if constexpr (HALF) {
        for (int s = 0; s < nst2; s += 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fetch(op[(t + 1) & 1], (t + 1) & 1);     // operands of stage s+t+1
                mma(op[t]);                              // stage s+t
                stash(rg[t], t);                         // stage s+t+2
                load(rg[t]);                             // stage s+t+4
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // fp16 route: TWO SLOTS per stage, and the two waves of a SIMD (w and w + 4) run them in opposite order --
        //   slot X   the 12 MFMAs of stage k (operands in registers) with the split arithmetic of stage k+2 between them
        //   slot Y   pieces of stage k+2 to LDS, global loads of stage k+4, LDS reads of the operands of stage k+1
        // waves 4-7 start one slot late, so that in every slot one wave of each SIMD feeds the matrix pipe while the other works
        // the LDS / memory side (with all eight waves in the same phase the two kinds of work ran one after the other: the kernel
        // took the SUM of its MFMA time and its load / split / LDS time).  Buffer k & 1 holds stage k: written in the Y slots
        // 2k-3 (waves 0-3) and 2k-2 (waves 4-7), read in the Y slots 2k-1 and 2k, rewritten from slot 2k+1 on; a barrier ends
        // every slot.
        const bool late = wave >= NT / 128;
        if (late) {
            __builtin_amdgcn_sched_barrier(0);
            if (WG_ABL != 5) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int s = 0; s < nst2; s += 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                Pieces q;
                split(rg[t], q);                         // stage s+t+2
                mma(op[0]);                              // stage s+t
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (WG_ABL != 5) __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                write(q, t);                             // stage s+t+2 over stage s+t
                load(rg[t]);                             // stage s+t+4
                fetch(op[0], (t + 1) & 1);               // operands of stage s+t+1
                __builtin_amdgcn_sched_barrier(0);
                if (WG_ABL != 5) __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!late) {
            __builtin_amdgcn_sched_barrier(0);
            if (WG_ABL != 5) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: acc[i][j][r] = part[split][row wm*64 + i*32 + (r/4)*8 + hi*4 + r%4][column wp*64 + j*32 + lo]
    const float isx = 1.f / sx, isd = 1.f / sdy;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cg = bx * CW + wp * (CW / 2) + j * 32 + lo;
        if (cg >= CT) continue;
        float* out = a.part + (size_t)by * a.K * CT + cg;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
                const float v = (acc[i][j][r] * isx) * isd;
                if (m < a.K) {
                    out[(size_t)m * CT] = v;
                    if constexpr (!HALF) bad |= is_nonfinite(v);
                }
            }
    }
    if constexpr (!HALF) report_nonfinite(a.ovf, bad);
}

int launch_weight_row_absmax(const float* w, int K, int C, int T, int by_c, float* out, hipStream_t st) {
    hipLaunchKernelGGL(weight_row_absmax_kernel, dim3((unsigned)(by_c ? C : K)), dim3(256), 0, st, w, K, C, T, by_c, out);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

static int bsplit_check(const pcgan_conv_desc* d) {
    PCGAN_CHECK(d, "conv2d_bsplit: null descriptor");
    PCGAN_CHECK(d->dtype == PCGAN_F32 || d->dtype == PCGAN_BF16, "conv2d_bsplit: dtype %d", d->dtype);
    PCGAN_CHECK(d->stride == 1 && d->C % 16 == 0 && d->R * d->S <= BS_MAXTAP && d->K >= 32, "conv2d_bsplit: unsupported shape");
    PCGAN_CHECK(d->P == d->H + 2 * d->pad - d->R + 1 && d->Q == d->W + 2 * d->pad - d->S + 1, "conv2d_bsplit: output dims");
    PCGAN_CHECK(d->pad_mode == 0 || (d->pad < d->H && d->pad < d->W), "conv2d_bsplit: reflection pad too large");
    PCGAN_CHECK((size_t)d->N * d->C * d->H * d->W * 4 < 0x80000000ull, "conv2d_bsplit: input beyond 2 GiB");
    return 0;
}

// pieces per operand / bytes per activation element of the descriptor's storage type
static inline int np_of(const pcgan_conv_desc* d) { return d->dtype == PCGAN_BF16 ? 1 : 3; }
static inline size_t es_of(const pcgan_conv_desc* d) { return d->dtype == PCGAN_BF16 ? 2 : 4; }

// launch bsplit_conv_fwd_kernel<MODE, BM, NP, TA> for the tile / storage type at hand
template <int MODE>
static void launch_bsplit(const pcgan_conv_desc* d, int bm, dim3 grid, hipStream_t st, const BsplitArgs& a) {
    if (d->dtype == PCGAN_BF16) {
        if (bm == 256) hipLaunchKernelGGL((bsplit_conv_fwd_kernel<MODE, 256, 1, bf16>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bsplit_conv_fwd_kernel<MODE, 128, 1, bf16>), grid, dim3(256), 0, st, a);
    } else {
        if (bm == 256) hipLaunchKernelGGL((bsplit_conv_fwd_kernel<MODE, 256, 3, float>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bsplit_conv_fwd_kernel<MODE, 128, 3, float>), grid, dim3(256), 0, st, a);
    }
}

// the halo kernel takes a layer when a pixel tile is whole image rows and the chunks come in pairs; option "bsplit_halo" = 0 keeps
// every layer on the per-tap gather kernel (A/B measurement)
static bool halo_enabled() { return option(OPT_BSPLIT_HALO) != 0; }
static bool halo_geometry(int chan, int rows_out, int H, int W) {
    return (W == 32 || W == 64) && H >= 4 && H % (128 / W) == 0 && chan % 32 == 0 && rows_out % 256 == 0;
}
static bool halo_shape(int chan, int rows_out, int H, int W) { return halo_enabled() && halo_geometry(chan, rows_out, H, W); }

template <int MODE>
static void launch_halo(const pcgan_conv_desc* d, int W, dim3 grid, hipStream_t st, const HaloArgs& a, bool f16 = false) {
    if (d->dtype == PCGAN_BF16) {
        if (W == 32) hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_BF16, bf16, 32>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_BF16, bf16, 64>), grid, dim3(512), 0, st, a);
    } else if (f16) {
        if (W == 32) hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_F16X2, float, 32>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_F16X2, float, 64>), grid, dim3(512), 0, st, a);
    } else {
        if (W == 32) hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_BF16X3, float, 32>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((bsplit_halo_kernel<MODE, PK_BF16X3, float, 64>), grid, dim3(512), 0, st, a);
    }
}

}  // namespace pcgan

extern "C" int pcgan_conv2d_bsplit_supported(const pcgan_conv_desc* d) {
    return d && d->stride == 1 && d->C % 16 == 0 && d->R * d->S <= pcgan::BS_MAXTAP && d->K >= 32 &&
           (size_t)d->N * d->C * d->H * d->W * 4 < 0x80000000ull && (d->pad_mode == 0 || (d->pad < d->H && d->pad < d->W));
}

// M tile: 256 rows (8 waves share one gathered pixel tile) when the output channels fill it, else 128
static inline int bsplit_bm(const pcgan_conv_desc* d) { return d->K % 256 == 0 ? 256 : 128; }

extern "C" size_t pcgan_conv2d_bsplit_packed_bytes(const pcgan_conv_desc* d) {
    if (!pcgan_conv2d_bsplit_supported(d)) return 0;
    const int bm = bsplit_bm(d);
    const size_t nMt = (d->K + bm - 1) / bm, nst = (size_t)(d->C / 16) * d->R * d->S;
    return (size_t)pcgan::np_of(d) * nMt * nst * 32 * bm;
}

extern "C" int pcgan_conv2d_bsplit_pack(const pcgan_conv_desc* d, const float* w, void* packed, pcgan_stream_t s) {
    if (pcgan::bsplit_check(d)) return 1;
    PCGAN_CHECK(w && packed, "conv2d_bsplit_pack: null pointer");
    const int bm = bsplit_bm(d);
    const int T = d->R * d->S, nMt = (d->K + bm - 1) / bm, nst = (d->C / 16) * T;
    const size_t per_piece = (size_t)nMt * nst * 16 * bm;
    const int blocks = (int)((per_piece / 8 + 255) / 256 > 4096 ? 4096 : (per_piece / 8 + 255) / 256);      // 8 values per thread
    hipLaunchKernelGGL(pcgan::bsplit_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (__bf16*)packed, d->K, d->C, T, nMt, nst,
                       bm == 256 ? 1 : 0, pcgan::np_of(d));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_fwd_bsplit(const pcgan_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                                       int act, float slope, pcgan_stream_t s) {
    if (pcgan::bsplit_check(d)) return 1;
    PCGAN_CHECK(x && packed && y, "conv2d_fwd_bsplit: null pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_FWD), (hipStream_t)s);
    pcgan::BsplitArgs a;
    a.X = x; a.A = packed; a.bias = bias; a.Y = y;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->K; a.R = d->R; a.S = d->S; a.pad = d->pad; a.reflect = d->pad_mode;
    a.P = d->P; a.Q = d->Q;
    const int bm = bsplit_bm(d);
    a.nMt = (d->K + bm - 1) / bm;
    a.nst = (d->C / 16) * d->R * d->S;
    a.act = act; a.slope = slope;
    a.x_bytes = (unsigned)((size_t)d->N * d->C * d->H * d->W * pcgan::es_of(d));
    a.tstart[0] = a.tstart[1] = a.tstart[2] = a.tstart[3] = 0;
    a.phase_bytes = 0;
    a.nst_split = 0;
    const size_t ab = (size_t)pcgan::np_of(d) * a.nMt * a.nst * 32 * bm;
    PCGAN_CHECK(ab < 0x80000000ull, "conv2d_fwd_bsplit: packed weights beyond 2 GiB");
    a.a_bytes = (unsigned)ab;
    const long ptiles = ((long)d->N * d->P * d->Q + 127) / 128;
    const dim3 grid((unsigned)(ptiles * a.nMt));
    if (d->pad_mode == 1 && d->R == 3 && d->S == 3 && d->pad == 1 && pcgan::halo_shape(d->C, d->K, d->H, d->W)) {
        pcgan::HaloArgs h{};
        h.X = x; h.A = packed; h.bias = bias; h.Y = y;
        h.N = d->N; h.C = d->C; h.H = d->H; h.M = d->K; h.nMt = a.nMt; h.nch = d->C / 16; h.act = act; h.slope = slope;
        h.x_bytes = a.x_bytes; h.a_bytes = a.a_bytes; h.x_amax = h.w_amax = nullptr; h.x_namax = 0;
        pcgan::launch_halo<pcgan::BH_FWD>(d, d->W, grid, (hipStream_t)s, h);
        PCGAN_LAUNCH_CHECK();
        return 0;
    }
    if (d->pad_mode == 1) pcgan::launch_bsplit<pcgan::BS_FWD_REFLECT>(d, bm, grid, (hipStream_t)s, a);
    else pcgan::launch_bsplit<pcgan::BS_FWD_ZERO>(d, bm, grid, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// ---- data gradient of the reflection-padded 3x3 stride-1 convolution ---------------------------------------------------------
extern "C" int pcgan_conv2d_bsplit_dgrad_supported(const pcgan_conv_desc* d) {
    return d && d->stride == 1 && d->pad_mode == 1 && d->pad == 1 && d->R == 3 && d->S == 3 && d->K % 16 == 0 && d->C >= 32 &&
           d->H >= 4 && d->W >= 4 && d->P == d->H && d->Q == d->W && (size_t)d->N * d->K * d->H * d->W * 4 < 0x80000000ull;
}

static inline int bsplit_dgrad_bm(const pcgan_conv_desc* d) { return d->C % 256 == 0 ? 256 : 128; }

extern "C" size_t pcgan_conv2d_bsplit_dgrad_packed_bytes(const pcgan_conv_desc* d) {
    if (!pcgan_conv2d_bsplit_dgrad_supported(d)) return 0;
    const int bm = bsplit_dgrad_bm(d);
    const size_t nMt = (d->C + bm - 1) / bm, nst = (size_t)(d->K / 16) * 9;
    return 3 * (size_t)pcgan::np_of(d) * nMt * nst * 32 * bm;
}

extern "C" int pcgan_conv2d_bsplit_dgrad_pack(const pcgan_conv_desc* d, const float* w, void* packed, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_bsplit_dgrad_supported(d), "conv2d_bsplit_dgrad_pack: unsupported shape");
    PCGAN_CHECK(w && packed, "conv2d_bsplit_dgrad_pack: null pointer");
    const int bm = bsplit_dgrad_bm(d);
    const int nMt = (d->C + bm - 1) / bm, nst = (d->K / 16) * 9;
    const size_t total = 3 * (size_t)nMt * nst * 2 * bm;      // 16-byte entries of the three row classes
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pcgan::bsplit_pack_dgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (__bf16*)packed, d->K, d->C, nMt, nst,
                       bm == 256 ? 1 : 0, pcgan::np_of(d));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_bwd_data_bsplit(const pcgan_conv_desc* d, const void* dy, const void* packed, void* dx, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_bsplit_dgrad_supported(d), "conv2d_bwd_data_bsplit: unsupported shape");
    PCGAN_CHECK(d->dtype == PCGAN_F32 || d->dtype == PCGAN_BF16, "conv2d_bwd_data_bsplit: dtype %d", d->dtype);
    PCGAN_CHECK(dy && packed && dx, "conv2d_bwd_data_bsplit: null pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_DGRAD), (hipStream_t)s);
    pcgan::BsplitArgs a;
    a.X = dy; a.A = packed; a.bias = nullptr; a.Y = dx;
    a.N = d->N; a.C = d->K; a.H = d->H; a.W = d->W; a.M = d->C; a.R = 3; a.S = 3; a.pad = 1; a.reflect = 1;
    a.P = d->H; a.Q = d->W;
    const int bm = bsplit_dgrad_bm(d);
    a.nMt = (d->C + bm - 1) / bm;
    a.nst = (d->K / 16) * 9;
    a.act = PCGAN_ACT_NONE; a.slope = 0.f;
    a.x_bytes = (unsigned)((size_t)d->N * d->K * d->H * d->W * pcgan::es_of(d));
    const size_t per_phase = (size_t)pcgan::np_of(d) * a.nMt * a.nst * 32 * bm;
    PCGAN_CHECK(3 * per_phase < 0x80000000ull, "conv2d_bwd_data_bsplit: packed weights beyond 2 GiB");
    a.phase_bytes = (unsigned)per_phase;
    a.a_bytes = (unsigned)(3 * per_phase);
    a.nst_split = 0;
    if (pcgan::halo_shape(d->K, d->C, d->H, d->W)) {      // plain flipped weights = row class 0 of the packed buffer
        pcgan::HaloArgs h{};
        h.X = dy; h.A = packed; h.bias = nullptr; h.Y = dx;
        h.N = d->N; h.C = d->K; h.H = d->H; h.M = d->C; h.nMt = a.nMt; h.nch = d->K / 16; h.act = PCGAN_ACT_NONE; h.slope = 0.f;
        h.x_bytes = a.x_bytes; h.a_bytes = (unsigned)per_phase; h.x_amax = h.w_amax = nullptr; h.x_namax = 0;
        const dim3 hgrid((unsigned)((long)d->N * d->H * d->W / 128 * a.nMt));
        pcgan::launch_halo<pcgan::BH_DGRAD>(d, d->W, hgrid, (hipStream_t)s, h);
        PCGAN_LAUNCH_CHECK();
        return 0;
    }
    const long rows[3] = {(long)d->H - 2, 1, 1};
    long t = 0;
    for (int p = 0; p < 3; ++p) {
        a.tstart[p] = (int)t;
        t += ((long)d->N * rows[p] * d->W + 127) / 128;
    }
    a.tstart[3] = (int)t;
    const dim3 grid((unsigned)(t * a.nMt));
    pcgan::launch_bsplit<pcgan::BS_DGRAD_REFLECT>(d, bm, grid, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------
extern "C" int pcgan_conv2d_bsplit_wgrad_supported(const pcgan_conv_desc* d) {
    return d && d->stride == 1 && d->pad_mode == 1 && d->R == 3 && d->S == 3 && d->pad == 1 && d->W % 16 == 0 && d->K >= 32 &&
           d->P == d->H && d->Q == d->W && (size_t)d->N * d->C * (d->H + 2) * (d->W + 2) * 4 < 0x80000000ull &&
           (size_t)d->N * d->H * d->W * 3 * 2 * (size_t)(d->K % 256 == 0 ? 256 : 128) < 0x80000000ull;
}

static inline int bsplit_wgrad_splits(const pcgan_conv_desc* d, int bm, int* nst_split) {
    const int nst = d->N * d->H * d->W / 16;
    const long tiles = (long)((d->C * 9 + 127) / 128) * ((d->K + bm - 1) / bm);
    long want = (bm == 256 ? 256 : 512) / tiles;          // one round of resident workgroups
    if (want < 1) want = 1;
    if (want > nst / 8) want = nst / 8 > 0 ? nst / 8 : 1;
    *nst_split = (int)((nst + want - 1) / want);
    return (nst + *nst_split - 1) / *nst_split;
}

extern "C" size_t pcgan_conv2d_bsplit_wgrad_workspace_bytes(const pcgan_conv_desc* d) {
    if (!pcgan_conv2d_bsplit_wgrad_supported(d)) return 0;
    const int bm = d->K % 256 == 0 ? 256 : 128;
    int per;
    const int splits = bsplit_wgrad_splits(d, bm, &per);
    const size_t xpad = pcgan::align_up((size_t)d->N * d->C * (d->H + 2) * (d->W + 2) * 4, 256);
    const size_t nMt = (d->K + bm - 1) / bm;
    const size_t packed = pcgan::align_up(3 * nMt * (size_t)(d->N * d->H * d->W / 16) * 32 * bm, 256);
    const size_t part = (size_t)splits * nMt * bm * d->C * 9 * 4;
    return xpad + packed + part;       // (sized for the fp32 / 3-piece case; the bf16 path uses less of it)
}

extern "C" int pcgan_conv2d_bwd_weight_bsplit(const pcgan_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate,
                                              void* ws, size_t ws_bytes, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_bsplit_wgrad_supported(d), "conv2d_bwd_weight_bsplit: unsupported shape");
    PCGAN_CHECK(d->dtype == PCGAN_F32 || d->dtype == PCGAN_BF16, "conv2d_bwd_weight_bsplit: dtype %d", d->dtype);
    PCGAN_CHECK(x && dy && dw && ws && ws_bytes >= pcgan_conv2d_bsplit_wgrad_workspace_bytes(d), "conv2d_bwd_weight_bsplit: null pointer or small workspace");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_WGRAD), (hipStream_t)s);
    const int bm = d->K % 256 == 0 ? 256 : 128;
    PCGAN_CHECK(d->K % bm == 0, "conv2d_bwd_weight_bsplit: output channels must fill the %d-row tile", bm);
    hipStream_t st = (hipStream_t)s;
    const bool half = d->dtype == PCGAN_BF16;
    const int np = pcgan::np_of(d);
    int per;
    const int splits = bsplit_wgrad_splits(d, bm, &per);
    const int nst = d->N * d->H * d->W / 16, nMt = d->K / bm;
    const size_t xpad_bytes = pcgan::align_up((size_t)d->N * d->C * (d->H + 2) * (d->W + 2) * 4, 256);
    const size_t packed_bytes = pcgan::align_up(3 * (size_t)nMt * nst * 32 * bm, 256);
    void* xpad = ws;
    __bf16* packed = (__bf16*)((char*)ws + xpad_bytes);
    float* part = (float*)((char*)ws + xpad_bytes + packed_bytes);
    PCGAN_CHECK(d->N * d->C <= 65535, "conv2d_bwd_weight_bsplit: more than 65535 planes");
    pcgan::launch_pad(x, xpad, d->N * d->C, d->H, d->W, 1, 1, half, st);
    PCGAN_LAUNCH_CHECK();
    PCGAN_CHECK(nMt == 1, "conv2d_bwd_weight_bsplit: more than one %d-row tile of output channels is not built", bm);
    const size_t per_piece = (size_t)nst * 16 * bm;
    const dim3 ygrid((unsigned)((per_piece + 255) / 256 > 8192 ? 8192 : (per_piece + 255) / 256));
    if (half) hipLaunchKernelGGL(pcgan::bsplit_pack_dy_kernel<pcgan::bf16>, ygrid, dim3(256), 0, st, (const pcgan::bf16*)dy, packed, d->K, d->H * d->W, nst,
                                 bm == 256 ? 1 : 0, np);
    else hipLaunchKernelGGL(pcgan::bsplit_pack_dy_kernel<float>, ygrid, dim3(256), 0, st, (const float*)dy, packed, d->K, d->H * d->W, nst,
                            bm == 256 ? 1 : 0, np);
    PCGAN_LAUNCH_CHECK();
    pcgan::BsplitArgs a;
    a.X = xpad; a.A = packed; a.bias = nullptr; a.Y = part;
    a.N = d->N; a.C = d->C; a.H = d->H; a.W = d->W; a.M = d->K; a.R = 3; a.S = 3; a.pad = 1; a.reflect = 1;
    a.P = 1; a.Q = d->C * 9;
    a.nMt = nMt;
    a.nst = nst;
    a.nst_split = per;
    a.act = PCGAN_ACT_NONE; a.slope = 0.f;
    a.x_bytes = (unsigned)((size_t)d->N * d->C * (d->H + 2) * (d->W + 2) * pcgan::es_of(d));
    a.a_bytes = (unsigned)((size_t)np * nMt * nst * 32 * bm);
    a.tstart[0] = a.tstart[1] = a.tstart[2] = a.tstart[3] = 0;
    a.phase_bytes = 0;
    const dim3 grid((unsigned)(((d->C * 9 + 127) / 128) * nMt), (unsigned)splits);
    pcgan::launch_bsplit<pcgan::BS_WGRAD>(d, bm, grid, st, a);
    PCGAN_LAUNCH_CHECK();
    const size_t total = (size_t)d->K * d->C * 9;
    hipLaunchKernelGGL(pcgan::bsplit_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, dw, splits, total, accumulate);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// ---- fp16 two-piece route of the reflection-padded 3x3 convolution (forward + data gradient), fp32 tensors -------------------------
// packed buffer: [2 pieces][M tile][stage][k half][256 rows][8 fp16] (data gradient: the plain flipped weights -- the window kernel
// has no row classes) followed by the largest magnitude of every weight ROW (one float per produced channel, padded to 256 bytes):
// row m is scaled by its own power of two, pow2_scale(rowmax[m]), which the epilogue divides out again -- exactly
static size_t hsplit_body_bytes(const pcgan_conv_desc* d, int pass) {
    const int rows = pass == PCGAN_PASS_FWD ? d->K : d->C, chan = pass == PCGAN_PASS_FWD ? d->C : d->K;
    const size_t nMt = (rows + 255) / 256, nst = (size_t)(chan / 16) * 9;
    return 2 * nMt * nst * 32 * 256;
}

extern "C" int pcgan_conv2d_hsplit_supported(const pcgan_conv_desc* d, int pass) {
    if (!d || d->dtype != PCGAN_F32 || d->stride != 1 || d->pad_mode != 1 || d->pad != 1 || d->R != 3 || d->S != 3) return 0;
    if (d->P != d->H || d->Q != d->W || (size_t)d->N * (d->C > d->K ? d->C : d->K) * d->H * d->W * 4 >= 0x80000000ull) return 0;
    if (pass == PCGAN_PASS_FWD) return pcgan::halo_geometry(d->C, d->K, d->H, d->W) && hsplit_body_bytes(d, pass) < 0x80000000ull;
    if (pass == PCGAN_PASS_BWD_DATA) return pcgan::halo_geometry(d->K, d->C, d->H, d->W) && hsplit_body_bytes(d, pass) < 0x80000000ull;
    return 0;
}

static size_t hsplit_tail_bytes(const pcgan_conv_desc* d, int pass) {
    return pcgan::align_up((size_t)(pass == PCGAN_PASS_FWD ? d->K : d->C) * 4, 256);
}
extern "C" size_t pcgan_conv2d_hsplit_packed_bytes(const pcgan_conv_desc* d, int pass) {
    return pcgan_conv2d_hsplit_supported(d, pass) ? hsplit_body_bytes(d, pass) + hsplit_tail_bytes(d, pass) : 0;
}

extern "C" int pcgan_absmax_slots(size_t n) {
    const size_t want = (n / 4 + 1023) / 1024;       // ~4 vector loads per thread
    return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

extern "C" int pcgan_absmax(const void* x, size_t n, int dtype, float* out, int slots, pcgan_stream_t s) {
    PCGAN_CHECK(x && out && n > 0 && slots > 0 && slots <= 1024, "absmax: null pointer, empty tensor or bad slot count");
    PCGAN_CHECK(dtype == PCGAN_F32 || dtype == PCGAN_BF16, "absmax: dtype %d", dtype);
    hipStream_t st = (hipStream_t)s;
    const dim3 grid((unsigned)slots);
    if (dtype == PCGAN_BF16) hipLaunchKernelGGL(pcgan::absmax_kernel<pcgan::bf16>, grid, dim3(256), 0, st, (const pcgan::bf16*)x, n, out);
    else hipLaunchKernelGGL(pcgan::absmax_kernel<float>, grid, dim3(256), 0, st, (const float*)x, n, out);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_amax_audit(const float* claimed, int n_claimed, const float* fresh, int n_fresh, unsigned int* counts, pcgan_stream_t s) {
    PCGAN_CHECK(claimed && fresh && counts && n_claimed > 0 && n_fresh > 0, "amax_audit: null pointer or empty maxima");
    hipLaunchKernelGGL(pcgan::amax_audit_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, claimed, n_claimed, fresh, n_fresh, counts);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_hsplit_pack(const pcgan_conv_desc* d, int pass, const float* w, void* packed, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_hsplit_supported(d, pass), "conv2d_hsplit_pack: unsupported shape or pass");
    PCGAN_CHECK(w && packed, "conv2d_hsplit_pack: null pointer");
    hipStream_t st = (hipStream_t)s;
    const size_t body = hsplit_body_bytes(d, pass);
    float* amax = (float*)((char*)packed + body);     // the tail: the largest magnitude of every row of this pass's weight matrix
    hipLaunchKernelGGL(pcgan::weight_row_absmax_kernel, dim3((unsigned)(pass == PCGAN_PASS_FWD ? d->K : d->C)), dim3(256), 0, st, w, d->K, d->C, 9,
                       pass == PCGAN_PASS_FWD ? 0 : 1, amax);
    PCGAN_LAUNCH_CHECK();
    if (pass == PCGAN_PASS_FWD) {
        const int nMt = (d->K + 255) / 256, nst = (d->C / 16) * 9;
        const size_t per_piece = (size_t)nMt * nst * 16 * 256;
        const int blocks = (int)((per_piece / 8 + 255) / 256 > 4096 ? 4096 : (per_piece / 8 + 255) / 256);
        hipLaunchKernelGGL(pcgan::bsplit_pack_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)packed, d->K, d->C, 9, nMt, nst, 1, 2, amax);
    } else {
        const int nMt = (d->C + 255) / 256, nst = (d->K / 16) * 9;
        const size_t total = (size_t)nMt * nst * 2 * 256;       // 16-byte entries of one row class
        const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(pcgan::bsplit_pack_dgrad_kernel, dim3(blocks), dim3(256), 0, st, w, (__bf16*)packed, d->K, d->C, nMt, nst, 1, 2, amax, 1);
    }
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_fwd_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const void* packed,
                                       const float* bias, void* y, int act, float slope, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_hsplit_supported(d, PCGAN_PASS_FWD), "conv2d_fwd_hsplit: unsupported shape");
    PCGAN_CHECK(x && x_amax && n_amax > 0 && packed && y, "conv2d_fwd_hsplit: null pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_FWD), (hipStream_t)s);
    const size_t body = hsplit_body_bytes(d, PCGAN_PASS_FWD);
    pcgan::HaloArgs h{};
    h.X = x; h.A = packed; h.bias = bias; h.Y = y;
    h.N = d->N; h.C = d->C; h.H = d->H; h.M = d->K; h.nMt = (d->K + 255) / 256; h.nch = d->C / 16; h.act = act; h.slope = slope;
    h.x_bytes = (unsigned)((size_t)d->N * d->C * d->H * d->W * 4);
    h.a_bytes = (unsigned)body;
    h.x_amax = x_amax;
    h.x_namax = n_amax;
    h.w_amax = (const float*)((const char*)packed + body);
    h.ovf = pcgan::nonfinite_counter();
    const dim3 grid((unsigned)((long)d->N * d->H * d->W / 128 * h.nMt));
    pcgan::launch_halo<pcgan::BH_FWD>(d, d->W, grid, (hipStream_t)s, h, true);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_bwd_data_hsplit(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed,
                                            void* dx, pcgan_stream_t s) {
    return pcgan_conv2d_bwd_data_hsplit_add(d, dy, dy_amax, n_amax, packed, nullptr, dx, s);
}

extern "C" int pcgan_conv2d_bwd_data_hsplit_add(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax, const void* packed,
                                                const void* add, void* dx, pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_hsplit_supported(d, PCGAN_PASS_BWD_DATA), "conv2d_bwd_data_hsplit: unsupported shape");
    PCGAN_CHECK(dy && dy_amax && n_amax > 0 && packed && dx, "conv2d_bwd_data_hsplit: null pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_DGRAD), (hipStream_t)s);
    const size_t body = hsplit_body_bytes(d, PCGAN_PASS_BWD_DATA);
    pcgan::HaloArgs h{};
    h.X = dy; h.A = packed; h.bias = nullptr; h.Y = dx;
    h.N = d->N; h.C = d->K; h.H = d->H; h.M = d->C; h.nMt = (d->C + 255) / 256; h.nch = d->K / 16; h.act = PCGAN_ACT_NONE; h.slope = 0.f;
    h.x_bytes = (unsigned)((size_t)d->N * d->K * d->H * d->W * 4);
    h.a_bytes = (unsigned)body;              // plain flipped weights
    h.x_amax = dy_amax;
    h.x_namax = n_amax;
    h.w_amax = (const float*)((const char*)packed + body);
    h.ovf = pcgan::nonfinite_counter();
    h.R = add;
    const dim3 grid((unsigned)((long)d->N * d->H * d->W / 128 * h.nMt));
    pcgan::launch_halo<pcgan::BH_DGRAD>(d, d->W, grid, (hipStream_t)s, h, true);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

// weight gradient on the fp16 route (fp32 tensors) / bf16 route (bf16 tensors): padded copy of x (workspace), the kernel above over
// splits of the pixel reduction, reduce
// the general form of the kernel (GEN): zero padding of at most 1 applied inside the gather (no padded copy), ragged output width, row
// tiles.  Option "wgrad_gen" = 0 keeps the padded copy (A/B measurement; the ragged widths and K > 256 then leave this route)
static bool hsplit_wgrad_gen(const pcgan_conv_desc* d) {
    if (!pcgan::option(pcgan::OPT_WGRAD_GEN) || d->pad_mode != 0 || d->pad > 1) return false;
    return d->dtype == PCGAN_F32 || d->dtype == PCGAN_BF16;      // (bf16 tensors with ragged rows: element loads of dy, rows start on 2-byte boundaries)
}

extern "C" int pcgan_conv2d_hsplit_wgrad_inline(const pcgan_conv_desc* d) {
    if (!d || !pcgan_conv2d_hsplit_wgrad_supported(d)) return 0;
    const bool inline_reflect = d->stride == 1 && d->pad_mode == 1 && d->pad == 1 && d->R == 3 && d->S == 3 && d->H >= 2 &&
                                d->W >= 16 && d->P == d->H && d->Q == d->W && !pcgan::option(pcgan::OPT_WGRAD_PADCOPY);
    return (inline_reflect || d->pad == 0 || hsplit_wgrad_gen(d)) ? 1 : 0;
}

extern "C" int pcgan_conv2d_hsplit_wgrad_supported(const pcgan_conv_desc* d) {
    if (!d || (d->dtype != PCGAN_F32 && d->dtype != PCGAN_BF16)) return 0;
    if (d->stride != 1 && d->stride != 2) return 0;
    if (d->pad_mode == 1 && (d->stride != 1 || d->pad >= d->H || d->pad >= d->W)) return 0;
    if (d->K < 32 || d->R * d->S > 49 || d->P < 1 || d->Q < 1) return 0;
    const bool gen = hsplit_wgrad_gen(d);
    const int Qs = (d->Q + 15) & ~15;
    if (!gen && (d->Q % 16 != 0 || d->K > 256)) return 0;
    if ((Qs - d->Q) * 4 > Qs) return 0;           // ragged rows: at least three quarters of every 16-column stage are real columns
    if (d->P != (d->H + 2 * d->pad - d->R) / d->stride + 1 || d->Q != (d->W + 2 * d->pad - d->S) / d->stride + 1) return 0;
    // the gather reads xpad rows up to (P-1) stride + R - 1 and columns up to (Q-1) stride + S - 1: inside the padded plane by the two lines above
    return (size_t)d->N * d->C * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * 4 < 0x80000000ull && (size_t)d->N * d->K * d->P * d->Q * 4 < 0x80000000ull &&
           d->N * d->C <= 65535;
}

static inline int hsplit_wgrad_bm(const pcgan_conv_desc* d) { return d->K > 128 ? 256 : 128; }

// 256 columns per workgroup with the 256-row tile when that still leaves at least 8 column tiles: the dy tile is loaded, split and
// written to LDS once for 256 columns and a wave's 64 x 128 tile needs 12 LDS operand reads for 24 MFMAs instead of 8 for 12.
// bf16 tensors since round 2 (26.5 -> 26.0 ms per bf16 step).  fp32 tensors: the kernel ALONE gains (238 VGPRs, no spills: residual
// shape 0.153 -> 0.139 ms incl. the reduce of twice as many split partials), the STEP loses 1 % (1206 -> 1194 img/s, three interleaved
// pairs on one box: beside the data-gradient kernel of the main stream the wide workgroups take 0.292 instead of 0.277 ms and the
// parameter-gradient stream is the longer one) -- so fp32 tensors keep 128 columns; option "wgrad_cw" = 256 selects the wide form (A/B).
static inline int hsplit_wgrad_cw(const pcgan_conv_desc* d) {
    const int force = pcgan::option(pcgan::OPT_WGRAD_CW);      // 0: as measured best; 256: fp32 too; 128: bf16 too
    const bool wide = (d->dtype == PCGAN_BF16 && force != 128) || (d->dtype == PCGAN_F32 && d->pad_mode == 1 && force == 256);
    return wide && hsplit_wgrad_bm(d) == 256 && d->C * d->R * d->S >= 8 * 256 ? 256 : 128;
}

static inline int hsplit_wgrad_splits(const pcgan_conv_desc* d, int* nst_split) {
    const int nst = d->N * d->P * ((d->Q + 15) / 16);
    const int cw = hsplit_wgrad_cw(d);
    const int bmr = hsplit_wgrad_bm(d);
    const long tiles = (long)((d->C * d->R * d->S + cw - 1) / cw) * ((d->K + bmr - 1) / bmr);      // column tiles x row tiles
    long want = (hsplit_wgrad_bm(d) == 256 ? 256 : 512) / tiles;          // one round of resident workgroups
    if (want < 1) want = 1;
    if (want > nst / 8) want = nst / 8 > 0 ? nst / 8 : 1;
    *nst_split = (int)((nst + want - 1) / want);
    return (nst + *nst_split - 1) / *nst_split;
}

extern "C" size_t pcgan_conv2d_hsplit_wgrad_workspace_bytes(const pcgan_conv_desc* d) {
    if (!pcgan_conv2d_hsplit_wgrad_supported(d)) return 0;
    int per;
    const int splits = hsplit_wgrad_splits(d, &per);
    const size_t xpad = hsplit_wgrad_gen(d) ? 0 : pcgan::align_up((size_t)d->N * d->C * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * 4, 256);
    const size_t own = xpad + (size_t)splits * d->K * d->C * d->R * d->S * 4;
    // (option "wgrad_direct": the residual-block shape is handed to the image-innermost form of wgrad_direct.hip, which needs more room)
    const size_t direct = pcgan::option(pcgan::OPT_WGRAD_DIRECT) ? pcgan_conv2d_wgrad_direct_workspace_bytes(d) : 0;
    // (option "wgrad_rowring": the same shapes' row-ring form, wgrad_rowring.hip)
    const size_t ring = pcgan::option(pcgan::OPT_WGRAD_ROWRING) ? pcgan_conv2d_wgrad_rowring_workspace_bytes(d) : 0;
    const size_t other = direct > ring ? direct : ring;
    return own > other ? own : other;
}

extern "C" int pcgan_conv2d_bwd_weight_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy,
                                              const float* dy_amax, int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes,
                                              pcgan_stream_t s) {
    PCGAN_CHECK(pcgan_conv2d_hsplit_wgrad_supported(d), "conv2d_bwd_weight_hsplit: unsupported shape");
    const bool half = d->dtype == PCGAN_BF16;
    PCGAN_CHECK(x && dy && dw && ws && ws_bytes >= pcgan_conv2d_hsplit_wgrad_workspace_bytes(d), "conv2d_bwd_weight_hsplit: null pointer or small workspace");
    PCGAN_CHECK(half || (x_amax && dy_amax && n_xamax > 0 && n_dyamax > 0), "conv2d_bwd_weight_hsplit: fp32 tensors need their operand maxima");
    if (pcgan::option(pcgan::OPT_WGRAD_DIRECT) && pcgan_conv2d_wgrad_direct_supported(d))
        return pcgan_conv2d_bwd_weight_direct(d, x, x_amax, n_xamax, dy, dy_amax, n_dyamax, dw, accumulate, ws, ws_bytes, s);
    if (pcgan::option(pcgan::OPT_WGRAD_ROWRING) && pcgan_conv2d_wgrad_rowring_supported(d))
        return pcgan_conv2d_bwd_weight_rowring(d, x, x_amax, n_xamax, dy, dy_amax, n_dyamax, dw, accumulate, ws, ws_bytes, s);
    hipStream_t st = (hipStream_t)s;
    const bool res_like = pcgan::timer_kind_res(d, 0) == 0;
    pcgan::TimerScope timer(res_like ? pcgan::TIMER_RES_WGRAD : -1, st);       // padded copy + main kernel + reduce
    int per;
    const int splits = hsplit_wgrad_splits(d, &per);
    const int Hp = d->H + 2 * d->pad, Wp = d->W + 2 * d->pad;
    const size_t es = half ? 2 : 4;
    const bool gen = hsplit_wgrad_gen(d);
    const size_t xpad_bytes = gen ? 0 : pcgan::align_up((size_t)d->N * d->C * Hp * Wp * 4, 256);
    void* xpad = ws;
    float* part = (float*)((char*)ws + xpad_bytes);
    const void* xin = x;
    // the residual-block shape (fp32, 3x3, stride 1, reflection padding 1): the mirror is applied inside the gather, no padded copy
    const bool inline_reflect = d->stride == 1 && d->pad_mode == 1 && d->pad == 1 && d->R == 3 && d->S == 3 && d->H >= 2 && d->W >= 16 &&
                                d->P == d->H && d->Q == d->W && !pcgan::option(pcgan::OPT_WGRAD_PADCOPY);
    int Hx = Hp, Wx = Wp;
    if (inline_reflect || gen) {
        Hx = d->H;
        Wx = d->W;
    } else if (d->pad > 0) {
        pcgan::launch_pad(x, xpad, d->N * d->C, d->H, d->W, d->pad, d->pad_mode, half, st);
        PCGAN_LAUNCH_CHECK();
        xin = xpad;
    }
    pcgan::HWgradArgs a;
    a.XP = xin; a.DY = dy; a.part = part;
    a.N = d->N; a.C = d->C; a.K = d->K; a.P = d->P; a.Q = d->Q; a.Hp = Hx; a.Wp = Wx; a.R = d->R; a.S = d->S;
    a.reflect_inline = inline_reflect ? 1 : 0;
    a.Qs = (d->Q + 15) & ~15;
    a.pad = d->pad;
    a.splits = splits;
    a.nmt = (d->K + hsplit_wgrad_bm(d) - 1) / hsplit_wgrad_bm(d);
    a.nst = d->N * d->P * (a.Qs / 16);
    a.nst_split = per;
    a.xp_bytes = (unsigned)((size_t)d->N * d->C * Hx * Wx * es);
    a.dy_bytes = (unsigned)((size_t)d->N * d->K * d->P * d->Q * es);
    a.x_amax = x_amax; a.x_namax = n_xamax; a.dy_amax = dy_amax; a.dy_namax = n_dyamax;
    a.ovf = half ? nullptr : pcgan::nonfinite_counter();
    const int cw = hsplit_wgrad_cw(d);
    a.ntile = (d->C * d->R * d->S + cw - 1) / cw;
    a.nwg = a.ntile * splits * a.nmt;
    const dim3 grid((unsigned)((a.nwg + 7) & ~7));
    const int bm = hsplit_wgrad_bm(d);
#define LWG(BMV, SV) do { if (half) { if (cw == 256) hipLaunchKernelGGL((pcgan::hsplit_wgrad_kernel<256, SV, pcgan::bf16, 2, 1>), grid, dim3(512), 0, st, a); \
                                      else hipLaunchKernelGGL((pcgan::hsplit_wgrad_kernel<BMV, SV, pcgan::bf16, 1, 1>), grid, dim3(BMV * 2), 0, st, a); } \
                           else hipLaunchKernelGGL((pcgan::hsplit_wgrad_kernel<BMV, SV, float, 1, 1>), grid, dim3(BMV * 2), 0, st, a); } while (0)
#define LWH(BMV, SV, NCV) do { if (half) hipLaunchKernelGGL((pcgan::hsplit_wgrad_kernel<BMV, SV, pcgan::bf16, NCV>), grid, dim3(BMV * 2), 0, st, a); \
                               else hipLaunchKernelGGL((pcgan::hsplit_wgrad_kernel<BMV, SV, float, NCV>), grid, dim3(BMV * 2), 0, st, a); } while (0)
    {
    pcgan::TimerScope timer_main(res_like ? pcgan::TIMER_RES_WGRAD_MAIN : -1, st);
    if (gen) {
        if (bm == 256 && d->stride == 1) LWG(256, 1);
        else if (bm == 256) LWG(256, 2);
        else if (d->stride == 1) LWG(128, 1);
        else LWG(128, 2);
    } else
    if (bm == 256 && cw == 256 && d->stride == 1) LWH(256, 1, 2);
    else if (bm == 256 && cw == 256) LWH(256, 2, 2);
    else if (bm == 256 && d->stride == 1) LWH(256, 1, 1);
    else if (bm == 256) LWH(256, 2, 1);
    else if (d->stride == 1) LWH(128, 1, 1);
    else LWH(128, 2, 1);
    }
#undef LWH
#undef LWG
    PCGAN_LAUNCH_CHECK();
    const size_t total = (size_t)d->K * d->C * d->R * d->S;
    hipLaunchKernelGGL(pcgan::bsplit_wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part, dw, splits, total, accumulate);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
