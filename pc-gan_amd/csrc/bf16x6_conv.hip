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

namespace pcgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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

// weights w[M][C][R][S] -> [piece][mt][stage][half][BM][8] bf16 (BM = 128 << bm_shift), stage = chunk * T + tap, k in stage =
// channel in chunk
__global__ void bsplit_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ A, int M, int C, int T, int nMt, int nst,
                                   int bm_shift, int np) {
    const int BM = 128 << bm_shift;
    const size_t per_piece = (size_t)nMt * nst * 16 * BM;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_piece; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), row = (int)((i >> 3) & (BM - 1)), half = (int)((i >> (10 + bm_shift)) & 1);
        const size_t q = i >> (11 + bm_shift);
        const int st = (int)(q % nst), mt = (int)(q / nst);
        const int m = mt * BM + row, c = (st / T) * 16 + half * 8 + j, tap = st % T;
        const float v = m < M ? w[((size_t)m * C + c) * T + tap] : 0.f;
        __bf16 h, mm, l;
        split3(v, h, mm, l);
        A[i] = h;                      // (np == 1: the weight rounded to nearest-even bf16)
        if (np == 3) {
            A[per_piece + i] = mm;
            A[2 * per_piece + i] = l;
        }
    }
}

// ---- weight gradient as the same GEMM with the roles turned: rows = output channels k (operand A = dy, re-split per call), columns
// = (c, r, s), reduction = (n, y, x) in stages of 16 consecutive x.  The reflection padding is materialised once (xpad), so that
// the gather address is separable: column part (c, r, s) in the lane's offset, reduction part (n, y, x) in the scalar offset.
template <typename TA>
__global__ void bsplit_pad_reflect_kernel(const TA* __restrict__ x, TA* __restrict__ xp, int H, int W, int pad) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const TA* src = x + (size_t)blockIdx.y * H * W;
    TA* dst = xp + (size_t)blockIdx.y * Hp * Wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Hp * Wp; i += gridDim.x * blockDim.x) {
        int y = i / Wp - pad, xx = i % Wp - pad;
        y = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
        xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
        dst[i] = src[y * W + xx];
    }
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
__global__ void bsplit_pack_dgrad_kernel(const float* __restrict__ w, __bf16* __restrict__ A, int K, int C, int nMt, int nst, int bm_shift,
                                         int np) {
    const int BM = 128 << bm_shift;
    const size_t per_piece = (size_t)nMt * nst * 16 * BM, per_phase = (size_t)np * per_piece;
    // (3 row classes x per_piece entries; each entry writes its np pieces)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 3 * per_piece; i += (size_t)gridDim.x * blockDim.x) {
        const int phase = (int)(i / per_piece);
        const size_t e = i - (size_t)phase * per_piece;
        const int j = (int)(e & 7), row = (int)((e >> 3) & (BM - 1)), half = (int)((e >> (10 + bm_shift)) & 1);
        const size_t q = e >> (11 + bm_shift);
        const int st = (int)(q % nst), mt = (int)(q / nst);
        const int c = mt * BM + row, k = (st / 9) * 16 + half * 8 + j, tap = st % 9;
        const int rp = tap / 3, sp = tap - rp * 3;
        float v = 0.f;
        if (c < C) {
            const float* wk = w + ((size_t)k * C + c) * 9;
            v = wk[(2 - rp) * 3 + (2 - sp)];
            if ((phase == 1 && rp == 0) || (phase == 2 && rp == 2)) v += wk[rp * 3 + (2 - sp)];   // + wf[2 - rp][sp]
        }
        __bf16 h, mm, l;
        split3(v, h, mm, l);
        __bf16* out = A + (size_t)phase * per_phase;
        out[e] = h;
        if (np == 3) {
            out[per_piece + e] = mm;
            out[2 * per_piece + e] = l;
        }
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
    const int blocks = (int)((per_piece + 255) / 256 > 4096 ? 4096 : (per_piece + 255) / 256);
    hipLaunchKernelGGL(pcgan::bsplit_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (__bf16*)packed, d->K, d->C, T, nMt, nst,
                       bm == 256 ? 1 : 0, pcgan::np_of(d));
    PCGAN_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcgan_conv2d_fwd_bsplit(const pcgan_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                                       int act, float slope, pcgan_stream_t s) {
    if (pcgan::bsplit_check(d)) return 1;
    PCGAN_CHECK(x && packed && y, "conv2d_fwd_bsplit: null pointer");
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
    const size_t total = 3 * (size_t)nMt * nst * 16 * bm;
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
    const int per_plane = (d->H + 2) * (d->W + 2);
    const dim3 pgrid((per_plane + 255) / 256, d->N * d->C);
    if (half) hipLaunchKernelGGL(pcgan::bsplit_pad_reflect_kernel<pcgan::bf16>, pgrid, dim3(256), 0, st, (const pcgan::bf16*)x, (pcgan::bf16*)xpad, d->H, d->W, 1);
    else hipLaunchKernelGGL(pcgan::bsplit_pad_reflect_kernel<float>, pgrid, dim3(256), 0, st, (const float*)x, (float*)xpad, d->H, d->W, 1);
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
