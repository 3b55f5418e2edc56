// Implicit-GEMM convolution family for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_32x32x2_f32, exact f32 fma chain).
//
//   forward          Y[n][m][oy][ox] = act( sum_k A[m][k] * G(k; n,oy,ox) + bias[m] )
//   backward-data    the same kernels with the transposed gather; stride phases (only the structurally non-zero taps)
//                    and, for reflection padding, row classes with folded weights are phases of ONE launch
//   backward-weight  Wp[m][k] = sum_pix dY[m][pix] * G(k; pix)   (split over pixel ranges, deterministic second-pass
//                    sum, no atomics)
//
// Layout: activations NCHW fp32.  The GEMM "N" dimension is the flattened pixel index, so consecutive lanes touch
// consecutive addresses of one channel plane (coalesced loads and epilogue stores: the 32x32 accumulator has its COLUMN
// on the lane, so the pixel is the column and the output channel the row).
//
// Kernels:  igemm2_kernel / wgrad2_kernel  -- channel counts that are multiples of 16 / 64: the step's hot kernels
//           igemm_kernel  / wgrad_kernel   -- generic K order (3-/4-channel stems, odd channel counts, > 25 taps)
//           smallm_*                       -- <= 4 output channels (vector ALU)
//           repack_* / pack_strip / transpose4 -- weight packing (pcgan_conv2d_pack_weights, once per optimizer step)
// What bounds them and why they are written the way they are: DESIGN.md section 3.
//
// Reference call sites replaced: see include/pcgan_hip.h.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace pcgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_FWD_ZERO = 0, MODE_FWD_REFLECT = 1, MODE_BWD = 2, MODE_BWD_REFLECT = 3 };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned OOB = 0x80000000u;  // byte offset beyond any tensor (< 2 GiB): hardware returns 0

// Range-checked buffer loads: an invalid lane gets voffset = OOB and reads 0 -- no exec-mask
// branches, no 64-bit address arithmetic in the gather.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 ld_b128(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Activation tensors (x, y, dy, dx) are stored as TA = float or bf16 (common.h); the gathers go through range-checked buffer
// loads with BYTE offsets, so every offset of an activation tensor is scaled by ES = sizeof(TA).  Weights (packed A operands),
// partial sums and weight gradients are always fp32.
template <typename TA>
__device__ __forceinline__ float ldx(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ float ldx<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return ld_b32(r, voff, soff); }
template <>
__device__ __forceinline__ float ldx<bf16>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0) << 16);
}
// four consecutive elements
template <typename TA>
__device__ __forceinline__ float4 ldx4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ float4 ldx4<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <>
__device__ __forceinline__ float4 ldx4<bf16>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
}

// ---- bf16 tensors on the fp32-MFMA kernels without a conversion instruction per element ---------------------------------
// A gathered bf16 element arrives zero-extended in the LOW half of a VGPR; the fp32 value it stands for has those 16 bits in
// the HIGH half.  Shifting in the vector ALU costs one VALU instruction per element, which these VALU-starved loops feel
// (4 VALU per MFMA = 18 %; measured +20-30 % on igemm2, +95 % on wgrad2 with the shift).  Instead the LDS
// operand tiles of the ACTIVATIONS are zeroed once per workgroup and every element is stored with a 16-bit LDS write into the
// high half of its fp32 slot (ds_write_b16 / ds_write_b16_d16_hi: same instruction count as the 32-bit store they replace).
//   ldr / ldr4   raw gathered element(s): the fp32 value itself, or (bf16) the zero-extended / packed 16-bit pattern(s)
//   put1 / put4 / put4p  store raw element(s) into consecutive fp32 LDS slots
//   raw2f        the fp32 value of a raw element (where arithmetic on it is needed before the store)
template <typename TA>
__device__ __forceinline__ float ldr(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <>
__device__ __forceinline__ float ldr<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return ld_b32(r, voff, soff); }
template <>
__device__ __forceinline__ float ldr<bf16>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
}
template <typename TA>
__device__ __forceinline__ float raw2f(float raw) {
    if constexpr (sizeof(TA) == 2) return __uint_as_float(__float_as_uint(raw) << 16);
    else return raw;
}
template <typename TA>
__device__ __forceinline__ void put1(float* slot, float raw) {
    if constexpr (sizeof(TA) == 2) reinterpret_cast<unsigned short*>(slot)[1] = (unsigned short)__float_as_uint(raw);
    else *slot = raw;
}
template <typename TA>
__device__ __forceinline__ void put4(float* slot, float r0, float r1, float r2, float r3) {
    if constexpr (sizeof(TA) == 2) {
        unsigned short* p = reinterpret_cast<unsigned short*>(slot);
        p[1] = (unsigned short)__float_as_uint(r0);
        p[3] = (unsigned short)__float_as_uint(r1);
        p[5] = (unsigned short)__float_as_uint(r2);
        p[7] = (unsigned short)__float_as_uint(r3);
    } else {
        *reinterpret_cast<float4*>(slot) = make_float4(r0, r1, r2, r3);
    }
}
// four CONSECUTIVE elements: fp32: four values; bf16: two dwords of two packed elements each (in .x, .y)
template <typename TA>
__device__ __forceinline__ float4 ldr4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(TA) == 2) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), 0.f, 0.f);
    } else {
        return ldx4<float>(r, voff, soff);
    }
}
template <typename TA>
__device__ __forceinline__ void put4p(float* slot, const float4& raw) {
    if constexpr (sizeof(TA) == 2) {
        unsigned short* p = reinterpret_cast<unsigned short*>(slot);
        const unsigned a = __float_as_uint(raw.x), b = __float_as_uint(raw.y);
        p[1] = (unsigned short)a;
        p[3] = (unsigned short)(a >> 16);       // ds_write_b16_d16_hi
        p[5] = (unsigned short)b;
        p[7] = (unsigned short)(b >> 16);
    } else {
        *reinterpret_cast<float4*>(slot) = raw;
    }
}
// zero an LDS tile (all threads of the workgroup; the caller's next barrier publishes it)
template <typename TA>
__device__ __forceinline__ void zero_tile(float* tile, int n) {
    if constexpr (sizeof(TA) == 2)
        for (int i = threadIdx.x; i < n; i += blockDim.x) tile[i] = 0.f;
}

struct PhaseArgs {
    const float* A;  // [M][Kp], k = (tap_index * Cgp + c)
    int Kp;
    int Hs, Ws;      // pixel sub-grid of this phase
    int fy, fx;      // output coordinate = sub * ostep + f
    int r0, s0, nR, nS;  // taps: r = r0 + i*tstep (i < nR), s = s0 + j*tstep (j < nS)
    int Ptot;        // N * Hs * Ws
    const float* As; // small-M strip kernel: weights as [c][sj][8][4] (row taps and outputs zero-padded)
    int ymap;        // 1: sub-grid row sy is output row {0, pad+1 .. H-2-pad, H-1}[sy] (rows without a mirror image)
};

struct IgemmArgs {
    const void* X;      // gathered tensor [N][Cg][Hg][Wg], storage type TA
    void* Y;            // output tensor   [N][M][Yh][Yw], storage type TA
    int dtype;          // PCGAN_F32 / PCGAN_BF16: which TA instantiation runs
    const float* bias;  // [M] or null
    int M, N, Cg, Cgp, Hg, Wg;
    int Yh, Yw;
    int ostep, sl, pad, tstep;
    int act;
    float slope;
    unsigned x_bytes;
    int nphase;
    int rowfold;  // MODE_BWD_REFLECT: the row mirrors are folded into per-phase weights, only column mirrors are gathered
    int chunked;  // K order of ph[].A: 1 = (16-channel chunk, tap, channel) -> igemm2_kernel, 0 = (tap, channel)
    int ksplit;   // > 1: blockIdx.z takes a contiguous range of K stages and stores a raw partial sum
    float* Ypart; // [ksplit][N][M][Yh][Yw] partial sums (then reduced + bias + activation by splitk_reduce)
    int tstart[17];  // igemm2_kernel: first pixel tile of each phase in the linearised grid (no empty workgroups)
    PhaseArgs ph[16];
    // fp16 two-piece form (hgemm_kernel): partial maxima of |X| and the largest |weight| (device); hsplit selects the kernel
    const float* x_amax;
    const float* w_amax;
    int x_namax, hsplit;
    unsigned* ovf;      // non-finite sentinel (common.h); may be null
};

struct Geom {
    int Hg, Wg, sl, pad;
};

// spatial offset of tap (r, s) for the pixel (py, px) of this thread
template <int MODE>
__device__ __forceinline__ bool tap_offset(const Geom& a, int py, int px, int r, int s, int& off) {
    if (MODE == MODE_BWD || MODE == MODE_BWD_REFLECT) {
        const int ty = py + a.pad - r, tx = px + a.pad - s;
        const int oy = ty >> a.sl, ox = tx >> a.sl;  // divisible by construction of the phase
        off = oy * a.Wg + ox;
        return (ty >= 0) & (tx >= 0) & (oy < a.Hg) & (ox < a.Wg);  // '&': straight-line code, no branches
    } else {
        int iy = (py << a.sl) - a.pad + r;
        int ix = (px << a.sl) - a.pad + s;
        if (MODE == MODE_FWD_REFLECT) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.Hg ? 2 * (a.Hg - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.Wg ? 2 * (a.Wg - 1) - ix : ix;
            off = iy * a.Wg + ix;
            return true;
        } else {
            off = iy * a.Wg + ix;
            return ((unsigned)iy < (unsigned)a.Hg) & ((unsigned)ix < (unsigned)a.Wg);
        }
    }
}

// wave-uniform iterator over the K slots (ri, sj, c) with c fastest
struct KIter {
    int ri, sj, c;
    __device__ __forceinline__ void advance(int n, int Cgp, int nS) {
        c += n;
        while (c >= Cgp) {
            c -= Cgp;
            if (++sj == nS) {
                sj = 0;
                ++ri;
            }
        }
    }
};

// Generic-K-order kernel: K ordered (tap, channel) with the channel count padded to 4, so a 16-deep K stage may
// straddle filter taps (3-/4-channel stems, odd channel counts, > 25 taps).  Block tile BM (output channels) x BP
// (pixels), K stage 16, 4 waves, double-buffered LDS, one barrier per stage.  LDS images (all accesses 128-bit):
//   As[row][20]      : 16 k of one output channel per row (+4 floats pad => ds_read_b128 conflict-free)
//   Bs[k/4][pix][4]  : 4 consecutive k of one pixel per 16-byte slot
// The MFMA consumes K in a permuted order (half-wave h takes k = 4*(2q+h)+j in step (q,j)); A and B use the same
// permutation so the sum is unchanged.  The layers that matter for the step time use igemm2_kernel below.
template <int MODE, int BM, int BP, typename TA>
__global__ void __launch_bounds__(256) igemm_kernel(IgemmArgs a) {
    static_assert(MODE != MODE_BWD_REFLECT, "the mirror-gather data gradient exists only in the chunked-K kernel");
    constexpr unsigned ES = sizeof(TA);
    constexpr int WM = (BM == 128 || (BM == 64 && BP == 64)) ? 2 : 1;  // waves along M
    constexpr int WP = 4 / WM;                                         // waves along pixels
    constexpr int WMT = BM / WM, WPT = BP / WP;
    static_assert(WMT % 32 == 0 && WPT % 32 == 0, "wave tile must be a multiple of 32x32");
    constexpr int MI = WMT / 32, PJ = WPT / 32;
    constexpr int AP = 20;
    constexpr int KPT = BP / 16;                 // K slots per thread per stage (8 or 4)
    constexpr int ACH = (BM * 4 + 255) / 256;    // float4 chunks of A per thread
    __shared__ __attribute__((aligned(16))) float As[2][BM * AP];
    __shared__ __attribute__((aligned(16))) float Bs[2][4 * BP * 4];

    const PhaseArgs& P = a.ph[blockIdx.y];
    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WP, wp = wave % WP;
    const int nMt = (a.M + BM - 1) / BM;
    const int mt = blockIdx.x % nMt, pt = blockIdx.x / nMt;
    const int m0 = mt * BM, p0 = pt * BP;
    const int Ptot = P.Ptot, Kp = P.Kp;
    if (p0 >= Ptot) return;  // phases of unequal size share one grid
    const int ph_r0 = P.r0, ph_s0 = P.s0, ph_nR = P.nR, ph_nS = P.nS, ph_Ws = P.Ws, ph_fy = P.fy, ph_fx = P.fx;

    const Geom g{a.Hg, a.Wg, a.sl, a.pad};
    const int HsWs = P.Hs * ph_Ws;
    const int HgWg = a.Hg * a.Wg;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(P.A, (unsigned)a.M * (unsigned)Kp * 4u);

    // --- this thread's gather pixel -------------------------------------------------
    const int pl = tid % BP;
    const int pg = p0 + pl;
    const bool pvalid = pg < Ptot;
    int vbase = 0, py = 0, px = 0;
    if (pvalid) {
        const int gn = pg / HsWs;
        const int rem = pg - gn * HsWs;
        const int sy = rem / ph_Ws;
        py = sy * a.ostep + ph_fy;
        px = (rem - sy * ph_Ws) * a.ostep + ph_fx;
        vbase = gn * a.Cg * HgWg;
    }
    const int ksub = __builtin_amdgcn_readfirstlane(tid / BP);  // which KPT-slice of the stage this wave gathers

    // K-stage range of this workgroup (split-K: small problems are cut along K to fill the 256 CUs)
    const int nst_all = (Kp + 15) / 16;
    const int nst_per = a.ksplit > 1 ? (nst_all + a.ksplit - 1) / a.ksplit : nst_all;
    const int st_begin = a.ksplit > 1 ? (int)blockIdx.z * nst_per : 0;
    const int st_end = st_begin + nst_per < nst_all ? st_begin + nst_per : nst_all;

    KIter it{0, 0, 0};
    it.advance(st_begin * 16 + ksub * KPT, a.Cgp, ph_nS);

    float4 areg[ACH];
    float breg[KPT];
    bool a_ok[ACH];
    unsigned a_off[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int q = tid + 256 * j;
        const int row = q >> 2, kc = (q & 3) * 4;
        a_ok[j] = (row < BM) & (m0 + row < a.M);
        a_off[j] = (unsigned)((m0 + row) * Kp + kc) * 4u;
    }

    auto load_stage = [&](int k0) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int kc = ((tid + 256 * j) & 3) * 4;
            areg[j] = ld_b128(rA, (a_ok[j] & (k0 + kc < Kp)) ? a_off[j] + (unsigned)k0 * 4u : OOB);
        }
        KIter e = it;
        unsigned voff = OOB;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if (i == 0 || e.c == 0) {  // wave-uniform: the tap changed
                voff = OOB;
                if (e.ri < ph_nR) {
                    int off;
                    const bool ok = tap_offset<MODE>(g, py, px, ph_r0 + e.ri * a.tstep, ph_s0 + e.sj * a.tstep, off);
                    voff = (ok && pvalid) ? (unsigned)(vbase + off) * ES : OOB;
                }
            }
            breg[i] = (e.c < a.Cg) ? ldr<TA>(rX, voff, (unsigned)(e.c * HgWg) * ES) : 0.f;
            e.advance(1, a.Cgp, ph_nS);
        }
        it.advance(16, a.Cgp, ph_nS);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int q = tid + 256 * j;
            const int row = q >> 2, kc = (q & 3) * 4;
            if (BM * 4 >= 256 || row < BM) *reinterpret_cast<float4*>(&As[buf][row * AP + kc]) = areg[j];
        }
#pragma unroll
        for (int gq = 0; gq < KPT / 4; ++gq)
            put4<TA>(&Bs[buf][((ksub * (KPT / 4) + gq) * BP + pl) * 4], breg[gq * 4 + 0], breg[gq * 4 + 1], breg[gq * 4 + 2], breg[gq * 4 + 3]);
    };

    f32x16 acc[MI][PJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float av0[MI][4], bv0[PJ][4], av1[MI][4], bv1[PJ][4];
    auto read_ops = [&](int buf, int q, float (&av)[MI][4], float (&bv)[PJ][4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(&As[buf][(wm * WMT + i * 32 + lo) * AP + (2 * q + hi) * 4]);
            av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(&Bs[buf][((2 * q + hi) * BP + wp * WPT + j * 32 + lo) * 4]);
            bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
        }
    };
    auto mfma_group = [&](const float (&av)[MI][4], const float (&bv)[PJ][4]) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][jj], bv[j][jj], acc[i][j], 0, 0, 0);
    };
    if (st_begin < st_end) {  // (empty K range of a split-K tail: accumulators stay zero, stored below)
        if constexpr (sizeof(TA) == 2) {
            zero_tile<TA>(&Bs[0][0], 2 * 4 * BP * 4);
            __syncthreads();
        }
        load_stage(st_begin * 16);
        store_stage(0);
        __syncthreads();
        read_ops(0, 0, av0, bv0);
        for (int st = st_begin; st < st_end; ++st) {
            const int buf = (st - st_begin) & 1;
            const bool more = st + 1 < st_end;
            read_ops(buf, 1, av1, bv1);
            if (more) load_stage((st + 1) * 16);
            mfma_group(av0, bv0);
            mfma_group(av1, bv1);
            if (more) store_stage(buf ^ 1);
            __syncthreads();
            if (more) read_ops(buf ^ 1, 0, av0, bv0);
        }
    }

    // --- epilogue: bias + activation, NCHW store (pixel on the lane -> coalesced) -----
    const int YhYw = a.Yh * a.Yw;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int pix = p0 + wp * WPT + j * 32 + lo;
        if (pix >= Ptot) continue;
        const int n = pix / HsWs;
        const int rem = pix - n * HsWs;
        const int sy = rem / ph_Ws;
        const int oy = sy * a.ostep + ph_fy;
        const int ox = (rem - sy * ph_Ws) * a.ostep + ph_fx;
        if (a.ksplit > 1) {  // raw partial sum; bias / activation happen in splitk_reduce_kernel
            float* Yp = a.Ypart + ((size_t)blockIdx.z * a.N + n) * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < a.M) Yp[(size_t)m * YhYw] = acc[i][j][r];
                }
            }
            continue;
        }
        TA* Yp = (TA*)a.Y + (size_t)n * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < a.M) {
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[m];
                    v = act_apply(v, a.act, a.slope);
                    st1(Yp + (size_t)m * YhYw, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Chunked-K kernel (C % 16 == 0, <= 25 filter taps): the hot kernel of the step.
//
// K order (16-channel chunk, tap, channel-in-chunk): one K stage = 16 channels of ONE filter tap, consecutive
// stages walk the taps of the same 16 channel planes (L2-resident).
//
// What bounds this kernel (measured, scripts/micro/mfma_mix.hip): on one SIMD every vector-ALU, LDS and
// vector-memory instruction issued between two v_mfma costs the matrix pipe ~4-6 cycles -- they do not hide
// under the 64 cycles of a 32x32x2 fp32 MFMA; only scalar instructions are free.  So the loop is written to
// need as few non-scalar instructions per stage as possible:
//   * the gather offset of (pixel, tap) -- padding / reflection / stride arithmetic, validity in bit 31 -- is
//     tabulated once per workgroup in LDS: a stage needs ONE 4-byte LDS read (+1 VALU for its address);
//   * channel and K offsets go through the scalar offset operand of the buffer loads;
//   * the LDS buffer index is a compile-time constant (loop unrolled by two), so every LDS address is a
//     per-thread base register + immediate;
//   * the K iterator lives in SGPRs.
// Per wave and stage (128x128 tile): 32 MFMA, 9 LDS reads, 4 LDS writes, 10 global loads, ~2 VALU.
//
// Pipeline (a wave issues in order and stops at every wait, so each wait must come long after its request):
//     first half of the MFMA chain (operands av0/bv0, already in registers)
//         + LDS reads of this stage's second-half operands av1/bv1
//         + LDS write of stage t+1 (its global loads were issued one stage ago)
//         + global gathers of stage t+2
//     barrier  (stage t+1 is now visible; nobody still reads the buffer written next)
//     second half of the chain (av1/bv1)
//         + LDS reads of stage t+1's first-half operands av0/bv0
//         + offset-table read for the gathers of stage t+3
// so the LDS write -> barrier -> LDS read latency chain of a hand-over sits under matrix instructions instead
// of between two stages.  Two LDS buffers suffice (the buffer written in stage t was last read before the
// barrier of stage t-1).  Stages past the end of the K range are gathered as all-out-of-range (zeros) and
// written to LDS but never consumed.  Source order in the loop IS the issue order (sched_barrier(0) per slot).
static constexpr int NTAP_FWD = 25;   // filter taps the offset table holds (5x5)
static constexpr int NTAP_MIR = 9;    // ... for the fused reflect data gradient (4 source combinations)
static constexpr int NTAP_CG4 = 49;   // ... for 3-/4-channel tensors (7x7 stems)

// index along one axis of the gathered tensor for filter tap `tap`, or 0xffffffff if the tap falls outside.
// Forward modes: p = output coordinate.  Backward modes: base = p + pad (or the padded-grid index of the mirror
// image of p for the fused reflect gradient; base_ok = false if there is none).
template <int MODE>
__device__ __forceinline__ unsigned axis_entry(int p, int base, bool base_ok, int tap, int n, int sl, int pad) {
    int i;
    bool ok;
    if (MODE == MODE_BWD || MODE == MODE_BWD_REFLECT) {
        const int t = base - tap;
        i = t >> sl;  // divisible by construction of the phase
        ok = base_ok & (t >= 0) & (i < n);
    } else {
        i = (p << sl) - pad + tap;
        if (MODE == MODE_FWD_REFLECT) {
            i = i < 0 ? -i : i;
            i = i >= n ? 2 * (n - 1) - i : i;
            ok = true;
        } else {
            ok = (unsigned)i < (unsigned)n;
        }
    }
    return ok ? (unsigned)i : 0xffffffffu;
}

__device__ __forceinline__ float4 ld_b128s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Pixel enumeration of a phase: index -> (image, y, x) on the output grid (row-major, lanes stay coalesced).
__device__ __forceinline__ int refl_inner(int i, int n, int p) { return i == 0 ? 0 : (i == n - 2 * p - 1 ? n - 1 : i + p); }
__device__ __forceinline__ void pix_coord(const IgemmArgs& a, const PhaseArgs& P, int pg, int& n, int& py, int& px) {
    const int HsWs = P.Hs * P.Ws;
    n = pg / HsWs;
    const int rem = pg - n * HsWs;
    const int sy = rem / P.Ws;
    py = P.ymap ? refl_inner(sy, a.Yh, a.pad) : sy * a.ostep + P.fy;
    px = (rem - sy * P.Ws) * a.ostep + P.fx;
}

// CPS = channels per K stage: 16 (chunked order, channel count a multiple of 16) or 4 (image-like tensors of 3-4 channels,
// K order (tap, channel) with the channels padded to 4: one stage = 4 filter taps x 4 channels, up to 7x7 taps).
template <int MODE, int BM, int BP, int CPS, typename TA>
__global__ void __launch_bounds__(256) igemm2_kernel(IgemmArgs a) {
    constexpr unsigned ES = sizeof(TA);
    constexpr int WM = (BM == 128 || (BM == 64 && BP == 64)) ? 2 : 1;
    constexpr int WP = 4 / WM;
    constexpr int WMT = BM / WM, WPT = BP / WP;
    constexpr int MI = WMT / 32, PJ = WPT / 32;
    constexpr int AP = 20;
    constexpr int KPT = BP / 16;
    constexpr int ACH = (BM * 4 + 255) / 256;
    constexpr bool MIR = MODE == MODE_BWD_REFLECT;
    static_assert(CPS == 16 || (CPS == 4 && !MIR), "4-channel stages: forward and plain data gradient only");
    constexpr int TROWS = (MIR ? NTAP_MIR : (CPS == 4 ? NTAP_CG4 : NTAP_FWD)) + 1;   // + one all-out-of-range row for dead stages
    constexpr int NCOMB = MIR ? 4 : 1;
    __shared__ __attribute__((aligned(16))) float As[2][BM * AP];
    __shared__ __attribute__((aligned(16))) float Bs[2][4 * BP * 4];
    __shared__ unsigned offT[NCOMB][TROWS][BP];
    __shared__ __attribute__((aligned(16))) float biasS[BM];

    const int nMt = (a.M + BM - 1) / BM;
    const int mt = blockIdx.x % nMt;
    int pt = blockIdx.x / nMt;
    int phase = 0;
    while (phase + 1 < a.nphase && pt >= a.tstart[phase + 1]) ++phase;   // grid.x = all phases' tiles back to back
    pt -= a.tstart[phase];
    const PhaseArgs& P = a.ph[phase];
    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WP, wp = wave % WP;
    const int Ptot = P.Ptot, Kp = P.Kp;
    const int m0 = mt * BM, p0 = pt * BP;
    const int ph_nS = P.nS, ph_Ws = P.Ws, ph_fy = P.fy, ph_fx = P.fx;
    const int T = P.nR * ph_nS;
    const int HsWs = P.Hs * ph_Ws;
    const int HgWg4 = a.Hg * a.Wg * (int)ES;     // bytes of one channel plane
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(P.A, (unsigned)a.M * (unsigned)Kp * 4u);

    // --- gather-offset table of this workgroup's BP pixels ------------------------------
    const int pl = tid % BP;
    int myr = -1, mxr = -1;
    {
        const int pg = p0 + pl;
        const bool pvalid = pg < Ptot;
        int gn = 0, py = 0, px = 0;
        if (pvalid) pix_coord(a, P, pg, gn, py, px);
        const unsigned vbase = (unsigned)gn * (unsigned)a.Cg * (unsigned)(a.Hg * a.Wg);
        if (MIR && pvalid) {  // padded row j holds input row reflect(j - pad): row py also appears at these padded rows
            if (py >= 1 && py <= a.pad) myr = a.pad - py;
            else if (py >= a.Yh - 1 - a.pad && py <= a.Yh - 2) myr = a.pad + 2 * (a.Yh - 1) - py;
            if (px >= 1 && px <= a.pad) mxr = a.pad - px;
            else if (px >= a.Yw - 1 - a.pad && px <= a.Yw - 2) mxr = a.pad + 2 * (a.Yw - 1) - px;
            if (a.rowfold) myr = -1;   // this phase's weights already carry the row mirror
        }
        for (int t = tid / BP; t <= T; t += 256 / BP) {
            const int ri = t / ph_nS, sj = t - ri * ph_nS;
            const int r = P.r0 + ri * a.tstep, sx = P.s0 + sj * a.tstep;
            const bool live = pvalid && t < T;
            const unsigned y = axis_entry<MODE>(py, py + a.pad, true, r, a.Hg, a.sl, a.pad);
            const unsigned x = axis_entry<MODE>(px, px + a.pad, true, sx, a.Wg, a.sl, a.pad);
            offT[0][t][pl] = (live && y != 0xffffffffu && x != 0xffffffffu) ? (vbase + y * (unsigned)a.Wg + x) * ES : OOB;
            if (MIR) {
                const unsigned yb = axis_entry<MODE>(py, myr, myr >= 0, r, a.Hg, a.sl, a.pad);
                const unsigned xb = axis_entry<MODE>(px, mxr, mxr >= 0, sx, a.Wg, a.sl, a.pad);
                offT[NCOMB > 1 ? 1 : 0][t][pl] = (live && y != 0xffffffffu && xb != 0xffffffffu) ? (vbase + y * (unsigned)a.Wg + xb) * ES : OOB;
                offT[NCOMB > 1 ? 2 : 0][t][pl] = (live && yb != 0xffffffffu && x != 0xffffffffu) ? (vbase + yb * (unsigned)a.Wg + x) * ES : OOB;
                offT[NCOMB > 1 ? 3 : 0][t][pl] = (live && yb != 0xffffffffu && xb != 0xffffffffu) ? (vbase + yb * (unsigned)a.Wg + xb) * ES : OOB;
            }
        }
        if (tid < BM) biasS[tid] = (a.bias != nullptr && m0 + tid < a.M) ? a.bias[m0 + tid] : 0.f;
        zero_tile<TA>(&Bs[0][0], 2 * 4 * BP * 4);     // (published by the barrier in front of the first stage)
    }
    const int ksub = __builtin_amdgcn_readfirstlane(tid / BP);

    const int nst_all = (Kp + 15) >> 4;
    const int nst_per = a.ksplit > 1 ? (nst_all + a.ksplit - 1) / a.ksplit : nst_all;
    const int st_begin = a.ksplit > 1 ? (int)blockIdx.z * nst_per : 0;
    const int st_end = st_begin + nst_per < nst_all ? st_begin + nst_per : nst_all;

    // load-side iterator (scalar; runs two stages ahead of the MFMA chain)
    int it_c, it_tap, it_k0;
    if (CPS == 4) {
        it_tap = st_begin * 4;
        it_c = 0;
    } else {
        const int cc0 = st_begin / T;
        it_tap = st_begin - cc0 * T;
        it_c = cc0 * 16;
    }
    it_k0 = st_begin * 16;
    unsigned a_base[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int q = tid + 256 * j;
        const int row = q >> 2, kc = (q & 3) * 4;
        a_base[j] = ((row < BM) & (m0 + row < a.M)) ? (unsigned)((m0 + row) * Kp + kc) * 4u : OOB;
    }

    float4 areg[ACH];
    float breg[KPT];
    float bmir[MIR ? 3 : 1][MIR ? KPT : 1];
    unsigned vo[NCOMB];          // gather offsets of the next load (bit 31 = out of range)
    unsigned vo_b = OOB;         // CPS 4, BP 128: offset of this thread's second filter tap
    int vo_c = 0, vo_k0 = 0;     // channel chunk / A column of the stage `vo` belongs to

    // offset-table read for the stage the iterator points at + iterator advance
    auto next_offsets = [&](auto nm_tag) {
        constexpr int NM = decltype(nm_tag)::value;
        if constexpr (CPS == 4) {   // this thread's KPT K-slots = KPT / 4 consecutive taps x 4 channels
            const int tA = it_tap + ksub * (KPT / 4);
            vo[0] = offT[0][tA < T ? tA : T][pl];
            if (KPT == 8) vo_b = offT[0][tA + 1 < T ? tA + 1 : T][pl];
            vo_k0 = it_k0;
            it_tap += 4;
            it_k0 += 16;
            return;
        }
        const int row = it_c >= a.Cg ? T : it_tap;     // stage past the end of K: the all-out-of-range row
        const unsigned* tp = &offT[0][0][pl] + row * BP;
        vo[0] = tp[0];
        if constexpr (NM >= 1) vo[NCOMB > 1 ? 1 : 0] = tp[(NCOMB > 1 ? 1 : 0) * TROWS * BP];
        if constexpr (NM >= 3) {
            vo[NCOMB > 1 ? 2 : 0] = tp[(NCOMB > 1 ? 2 : 0) * TROWS * BP];
            vo[NCOMB > 1 ? 3 : 0] = tp[(NCOMB > 1 ? 3 : 0) * TROWS * BP];
        }
        vo_c = it_c;
        vo_k0 = it_k0;
        const int t1 = it_tap + 1;
        const bool wr = t1 == T;
        it_tap = wr ? 0 : t1;
        it_c += wr ? 16 : 0;
        it_k0 += 16;
    };
    auto load_a = [&](int j) { areg[j] = ld_b128s(rA, a_base[j], (unsigned)vo_k0 * 4u); };
    auto load_b = [&](int i, auto nm_tag) {
        constexpr int NM = decltype(nm_tag)::value;
        if constexpr (CPS == 4) {
            const unsigned v = (i & 3) < a.Cg ? (i < 4 ? vo[0] : vo_b) : OOB;   // 3-channel tensors: the pad channel reads 0
            breg[i] = ldr<TA>(rX, v, (unsigned)((i & 3) * HgWg4));
            return;
        }
        const unsigned so = (unsigned)((vo_c + ksub * KPT + i) * HgWg4);
        breg[i] = ldr<TA>(rX, vo[0], so);
        if constexpr (NM >= 1) bmir[0][i] = ldr<TA>(rX, vo[NCOMB > 1 ? 1 : 0], so);
        if constexpr (NM >= 3) {
            bmir[1][i] = ldr<TA>(rX, vo[NCOMB > 1 ? 2 : 0], so);
            bmir[2][i] = ldr<TA>(rX, vo[NCOMB > 1 ? 3 : 0], so);
        }
    };
    auto store_a = [&](int buf, int j) {
        const int q = tid + 256 * j;
        const int row = q >> 2, kc = (q & 3) * 4;
        if (BM * 4 >= 256 || row < BM) *reinterpret_cast<float4*>(&As[buf][row * AP + kc]) = areg[j];
    };
    auto store_b = [&](int buf, int gq, auto nm_tag) {
        constexpr int NM = decltype(nm_tag)::value;
        float* slot = &Bs[buf][((ksub * (KPT / 4) + gq) * BP + pl) * 4];
        if constexpr (NM == 0) {       // raw elements straight into their slots
            put4<TA>(slot, breg[gq * 4 + 0], breg[gq * 4 + 1], breg[gq * 4 + 2], breg[gq * 4 + 3]);
        } else {                       // border workgroups of the fused reflect gradient: sum the mirror images as fp32 values
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = gq * 4 + e;
                v[e] = raw2f<TA>(breg[i]);
                if constexpr (NM == 1) v[e] += raw2f<TA>(bmir[0][i]);
                if constexpr (NM == 3) v[e] += (raw2f<TA>(bmir[0][i]) + raw2f<TA>(bmir[1][i])) + raw2f<TA>(bmir[2][i]);
            }
            *reinterpret_cast<float4*>(slot) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };

    f32x16 acc[MI][PJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float av0[MI][4], bv0[PJ][4], av1[MI][4], bv1[PJ][4];
    auto read_a = [&](int buf, int q, int i, float (&av)[MI][4]) {
        const float4 t = *reinterpret_cast<const float4*>(&As[buf][(wm * WMT + i * 32 + lo) * AP + (2 * q + hi) * 4]);
        av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
    };
    auto read_b = [&](int buf, int q, int j, float (&bv)[PJ][4]) {
        const float4 t = *reinterpret_cast<const float4*>(&Bs[buf][((2 * q + hi) * BP + wp * WPT + j * 32 + lo) * 4]);
        bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
    };
    // g-th matrix instruction of a half stage; consecutive ones hit different accumulators
    auto mfma_one = [&](int g, const float (&av)[MI][4], const float (&bv)[PJ][4]) {
        const int jj = g / (MI * PJ), i = (g / PJ) % MI, j = g % PJ;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][jj], bv[j][jj], acc[i][j], 0, 0, 0);
    };

    auto run = [&](auto nm_tag) {
        constexpr int NH = MI * PJ * 4;                      // matrix instructions per half stage
        // non-MFMA work of the first half, in issue order: operand reads (second half of this stage), LDS writes of
        // stage st+1, global loads of stage st+2
        constexpr int I_RA = 0, I_RB = I_RA + MI, I_WA = I_RB + PJ, I_WB = I_WA + ACH, I_LA = I_WB + KPT / 4,
                      I_LB = I_LA + ACH, NI1 = I_LB + KPT;
        // second half: operand reads of stage st+1 (first half), offset-table read for stage st+3
        constexpr int J_RA = 0, J_RB = J_RA + MI, J_TA = J_RB + PJ, NI2 = J_TA + 1;
        if (st_begin >= st_end) return;
        __syncthreads();                                     // table visible
        next_offsets(nm_tag);
#pragma unroll
        for (int j = 0; j < ACH; ++j) load_a(j);             // stage 0
#pragma unroll
        for (int i = 0; i < KPT; ++i) load_b(i, nm_tag);
        next_offsets(nm_tag);
#pragma unroll
        for (int j = 0; j < ACH; ++j) store_a(0, j);
#pragma unroll
        for (int gq = 0; gq < KPT / 4; ++gq) store_b(0, gq, nm_tag);
#pragma unroll
        for (int j = 0; j < ACH; ++j) load_a(j);             // stage 1
#pragma unroll
        for (int i = 0; i < KPT; ++i) load_b(i, nm_tag);
        next_offsets(nm_tag);                                // offsets of stage 2
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MI; ++i) read_a(0, 0, i, av0);
#pragma unroll
        for (int j = 0; j < PJ; ++j) read_b(0, 0, j, bv0);

        auto stage = [&](auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
            for (int g = 0; g < NH; ++g) {
                mfma_one(g, av0, bv0);
#pragma unroll
                for (int k = 0; k < NI1; ++k) {
                    if (k * NH / NI1 != g) continue;
                    if (k < I_RB) read_a(buf, 1, k - I_RA, av1);
                    else if (k < I_WA) read_b(buf, 1, k - I_RB, bv1);
                    else if (k < I_WB) store_a(buf ^ 1, k - I_WA);
                    else if (k < I_LA) store_b(buf ^ 1, k - I_WB, nm_tag);
                    else if (k < I_LB) load_a(k - I_LA);
                    else load_b(k - I_LB, nm_tag);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NH; ++g) {
                mfma_one(g, av1, bv1);
#pragma unroll
                for (int k = 0; k < NI2; ++k) {
                    if (k * NH / NI2 != g) continue;
                    if (k < J_RB) read_a(buf ^ 1, 0, k - J_RA, av0);
                    else if (k < J_TA) read_b(buf ^ 1, 0, k - J_RB, bv0);
                    else next_offsets(nm_tag);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int st = st_begin; st < st_end; st += 2) {
            stage(std::integral_constant<int, 0>{});
            if (st + 1 < st_end) stage(std::integral_constant<int, 1>{});
        }
    };
    if (MIR) {  // workgroup-uniform: how many mirror images do its pixels receive at most?
        const int any2 = __syncthreads_or((myr >= 0) & (mxr >= 0));
        const int any1 = __syncthreads_or((myr >= 0) | (mxr >= 0));
        if (any2 || (any1 && !a.rowfold)) run(std::integral_constant<int, 3>{});
        else if (any1) run(std::integral_constant<int, 1>{});      // column mirrors only (table slot 1)
        else run(std::integral_constant<int, 0>{});
    } else {
        run(std::integral_constant<int, 0>{});
    }

    // --- epilogue: bias (from LDS) + activation, NCHW store (pixel on the lane -> coalesced) ------
    const int YhYw = a.Yh * a.Yw;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int pix = p0 + wp * WPT + j * 32 + lo;
        if (pix >= Ptot) continue;
        int n, oy, ox;
        pix_coord(a, P, pix, n, oy, ox);
        if (a.ksplit > 1) {
            float* Yp = a.Ypart + ((size_t)blockIdx.z * a.N + n) * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < a.M) Yp[(size_t)m * YhYw] = acc[i][j][r];
                }
            }
            continue;
        }
        TA* Yp = (TA*)a.Y + (size_t)n * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ml = wm * WMT + i * 32 + 8 * rq + 4 * hi;      // 4 consecutive output channels
                const float4 bq = *reinterpret_cast<const float4*>(&biasS[ml]);
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + ml + e;
                    if (m < a.M) st1(Yp + (size_t)m * YhYw, act_apply(acc[i][j][rq * 4 + e] + bb[e], a.act, a.slope));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// fp16 two-piece form of the chunked-K kernel (fp32 tensors; forward with zero / reflection padding and the plain data gradient,
// any stride, <= 25 taps, channel count a multiple of 16): the same gather tables, K order, phases and split-K as igemm2_kernel,
// but both operands are scaled by a power of two and split into two fp16 pieces on their way to LDS (x * 2^e = h + l, common.h
// pow2_scale / split2h) and a 16-deep K stage is THREE v_mfma_f32_32x32x16_f16 per 32 x 32 block -- (l,h) (h,l) (h,h) -- instead of
// eight v_mfma_f32_32x32x2_f32.  Measured error at the fp32 kernel's level (scripts/micro/bf16_split: 5.3e-7 at K = 2304, fp32
// MFMA 6.1e-7).  a.x_amax[0 .. x_namax) are partial maxima of |X| (device), a.w_amax the largest |weight|.
//   LDS images [piece][k half][row or pixel][8 fp16]: every operand read is one conflict-free ds_read_b128;
//   per wave and stage (128 x 128 tile): 12 MFMA, 8 LDS reads; weights 2 x 16-byte loads, pixels 8 x 4-byte gathers per thread;
//   global loads two stages ahead in registers, LDS one stage ahead, operands of the next stage read under this stage's MFMAs.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// TA = bf16 (the bf16 path, desc.dtype = PCGAN_BF16): the stored bf16 activations go to LDS as they are, the fp32 weights are rounded
// to bf16 on their way there, ONE v_mfma_f32_32x32x16_bf16 per block and stage, no scaling -- plain mixed precision as in the
// one-product form of the residual-convolution kernels (bf16x6_conv.hip).
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

template <int MODE, int BM, int BP, typename TA>
__global__ void __launch_bounds__(256) hgemm_kernel(IgemmArgs a) {
    static_assert(MODE == MODE_FWD_ZERO || MODE == MODE_FWD_REFLECT || MODE == MODE_BWD, "forward and plain data gradient");
    constexpr bool HALF = sizeof(TA) == 2;      // bf16 tensors: one piece, one product
    constexpr int NP = HALF ? 1 : 2;
    constexpr unsigned ES = sizeof(TA);
    constexpr int WM = (BM == 128 || (BM == 64 && BP == 64)) ? 2 : 1;
    constexpr int WP = 4 / WM;
    constexpr int WMT = BM / WM, WPT = BP / WP;
    constexpr int MI = WMT / 32, PJ = WPT / 32;
    constexpr int KPT = BP / 16;        // channels of its pixel a thread gathers per stage (8 or 4)
    constexpr int ACH = BM * 4 / 256;   // float4 of the weight tile per thread (2 or 1)
    constexpr int TROWS = NTAP_FWD + 1; // + one all-out-of-range row for dead stages
    // four neighbouring lanes write the two k halves of one weight row: 128 bytes of padding between the halves put them on disjoint banks
    constexpr int AH = BM + 8;
    __shared__ __attribute__((aligned(16))) f16x8 As[2][NP][2 * AH];     // [buffer][piece][k half * AH + row]
    __shared__ __attribute__((aligned(16))) f16x8 Bs[2][NP][2 * BP];     // [buffer][piece][k half * BP + pixel]
    __shared__ unsigned offT[TROWS][BP];
    __shared__ __attribute__((aligned(16))) float biasS[BM];
    __shared__ __attribute__((aligned(16))) float iswS[BM];     // fp16 route: 1 / (the power of two row m0 + i of the weights was scaled by)

    const int nMt = (a.M + BM - 1) / BM;
    const int mt = blockIdx.x % nMt;
    int pt = blockIdx.x / nMt;
    int phase = 0;
    while (phase + 1 < a.nphase && pt >= a.tstart[phase + 1]) ++phase;
    pt -= a.tstart[phase];
    const PhaseArgs& P = a.ph[phase];
    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WP, wp = wave % WP;
    const int Ptot = P.Ptot, Kp = P.Kp;
    const int m0 = mt * BM, p0 = pt * BP;
    const int ph_nS = P.nS;
    const int T = P.nR * ph_nS;
    const int HgWg4 = a.Hg * a.Wg * (int)ES;     // bytes of one channel plane
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(P.A, (unsigned)a.M * (unsigned)Kp * 4u);

    float sx = 1.f;
    // gather-offset table of this workgroup's BP pixels (as in igemm2_kernel)
    const int pl = tid % BP;
    {
        const int pg = p0 + pl;
        const bool pvalid = pg < Ptot;
        int gn = 0, py = 0, px = 0;
        if (pvalid) pix_coord(a, P, pg, gn, py, px);
        const unsigned vbase = (unsigned)gn * (unsigned)a.Cg * (unsigned)(a.Hg * a.Wg);
        for (int t = tid / BP; t <= T; t += 256 / BP) {
            const int ri = t / ph_nS, sj = t - ri * ph_nS;
            const int r = P.r0 + ri * a.tstep, sxx = P.s0 + sj * a.tstep;
            const bool live = pvalid && t < T;
            const unsigned y = axis_entry<MODE>(py, py + a.pad, true, r, a.Hg, a.sl, a.pad);
            const unsigned x = axis_entry<MODE>(px, px + a.pad, true, sxx, a.Wg, a.sl, a.pad);
            offT[t][pl] = (live && y != 0xffffffffu && x != 0xffffffffu) ? (vbase + y * (unsigned)a.Wg + x) * ES : OOB;
        }
    }
    const int ksub = __builtin_amdgcn_readfirstlane(tid / BP);
    __syncthreads();

    const int nst_all = Kp >> 4;
    const int nst_per = a.ksplit > 1 ? (nst_all + a.ksplit - 1) / a.ksplit : nst_all;
    const int st_begin = a.ksplit > 1 ? (int)blockIdx.z * nst_per : 0;
    const int st_end = st_begin + nst_per < nst_all ? st_begin + nst_per : nst_all;
    const int nst_here = st_end > st_begin ? st_end - st_begin : 0;

    // load-side iterator (scalar): tap, first channel and weight column of the next stage to load
    int it_tap, it_c, it_k0, it_left = nst_here;
    {
        const int cc0 = st_begin / T;
        it_tap = st_begin - cc0 * T;
        it_c = cc0 * 16;
        it_k0 = st_begin * 16;
    }
    unsigned a_base[ACH];
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int q = tid + 256 * j;
        const int row = q >> 2, kc = (q & 3) * 4;
        a_base[j] = (m0 + row < a.M) ? (unsigned)((m0 + row) * Kp + kc) * 4u : OOB;
    }
    struct Stage {
        u32x4 av[ACH];      // 4 consecutive k of this thread's weight row(s)
        unsigned bv[KPT];   // KPT consecutive channels of this thread's pixel
    };
    auto load = [&](Stage& r) {
        const bool live = it_left > 0;
        const unsigned vo = offT[live ? it_tap : T][pl];
        const unsigned so = (unsigned)((it_c + ksub * KPT) * HgWg4);
#pragma unroll
        for (int j = 0; j < ACH; ++j) r.av[j] = __builtin_amdgcn_raw_buffer_load_b128(rA, live ? a_base[j] : OOB, (unsigned)it_k0 * 4u, 0);
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            if constexpr (HALF) r.bv[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rX, vo, so + (unsigned)(i * HgWg4), 0);
            else r.bv[i] = __builtin_amdgcn_raw_buffer_load_b32(rX, vo, so + (unsigned)(i * HgWg4), 0);
        }
        --it_left;
        const int t1 = it_tap + 1;
        const bool wr = t1 == T;
        it_tap = wr ? 0 : t1;
        it_c += wr ? 16 : 0;
        it_k0 += 16;
    };
    auto stash = [&](const Stage& r, int buf) {
        if constexpr (HALF) {
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                const int q = tid + 256 * j;
                const int row = q >> 2, kc = (q & 3) * 4;
                bf16x4v h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)__uint_as_float(r.av[j][e]);      // weights: round to nearest even
                *reinterpret_cast<bf16x4v*>(reinterpret_cast<__bf16*>(&As[buf][0][(kc >> 3) * AH + row]) + (kc & 4)) = h;
            }
            typedef unsigned short usK __attribute__((ext_vector_type(KPT)));
            usK v;
#pragma unroll
            for (int e = 0; e < KPT; ++e) v[e] = (unsigned short)r.bv[e];                     // stored bf16 patterns as they are
            if constexpr (KPT == 8) *reinterpret_cast<usK*>(&Bs[buf][0][ksub * BP + pl]) = v;
            else *reinterpret_cast<usK*>(reinterpret_cast<unsigned short*>(&Bs[buf][0][(ksub >> 1) * BP + pl]) + (ksub & 1) * 4) = v;
            return;
        }
        // the weights arrive PRE-SPLIT (pcgan_conv2d_hgemm_pack: they only change once per optimizer step while every net runs 2-4
        // times in between): the 16 bytes a thread loaded are [4 x fp16 high pieces | 4 x fp16 low pieces] of 4 consecutive k of its
        // row, scaled by the same power of two the epilogue divides by -- two 8-byte LDS stores, no arithmetic
#pragma unroll
        for (int j = 0; j < ACH; ++j) {
            const int q = tid + 256 * j;
            const int row = q >> 2, kc = (q & 3) * 4;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 h, l;
            h[0] = r.av[j][0]; h[1] = r.av[j][1];
            l[0] = r.av[j][2]; l[1] = r.av[j][3];
            _Float16* d0 = reinterpret_cast<_Float16*>(&As[buf][0][(kc >> 3) * AH + row]) + (kc & 4);
            _Float16* d1 = reinterpret_cast<_Float16*>(&As[buf][NP - 1][(kc >> 3) * AH + row]) + (kc & 4);
            *reinterpret_cast<u32x2*>(d0) = h;
            *reinterpret_cast<u32x2*>(d1) = l;
        }
        if constexpr (KPT == 8) {
            f16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                split2h(__uint_as_float(r.bv[e]) * sx, x, y);
                h[e] = x;
                l[e] = y;
            }
            Bs[buf][0][ksub * BP + pl] = h;
            Bs[buf][NP - 1][ksub * BP + pl] = l;
        } else {
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 x, y;
                split2h(__uint_as_float(r.bv[e]) * sx, x, y);
                h[e] = x;
                l[e] = y;
            }
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(&Bs[buf][0][(ksub >> 1) * BP + pl]) + (ksub & 1) * 4) = h;
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(&Bs[buf][NP - 1][(ksub >> 1) * BP + pl]) + (ksub & 1) * 4) = l;
        }
    };

    f32x16 acc[MI][PJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    struct Operands {
        f16x8 A[NP][MI], B[NP][PJ];
    };
    auto fetch = [&](Operands& o, int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int i = 0; i < MI; ++i) o.A[p][i] = As[buf][p][hi * AH + wm * WMT + i * 32 + lo];
#pragma unroll
            for (int j = 0; j < PJ; ++j) o.B[p][j] = Bs[buf][p][hi * BP + wp * WPT + j * 32 + lo];
        }
    };
    auto mma = [&](const Operands& o) {      // (l,h) (h,l) (h,h): smallest terms first
        if constexpr (HALF) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, o.A[0][i]), __builtin_bit_cast(bf16x8v, o.B[0][j]),
                                                                         acc[i][j], 0, 0, 0);
            return;
        }
        constexpr int PA[3] = {NP - 1, 0, 0}, PB[3] = {0, NP - 1, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.A[PA[q]][i], o.B[PB[q]][j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() {
        constexpr int NM = (HALF ? 1 : 3) * MI * PJ, NRD = NP * (MI + PJ);
#pragma unroll
        for (int q = 0; q < NM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, (NRD + NM - 1) / NM, 0);          // LDS reads of the next stage first
            __builtin_amdgcn_sched_group_barrier(0x002, 48 / NM + 1, 0);                  // split arithmetic
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                            // LDS writes
            __builtin_amdgcn_sched_group_barrier(0x020, (ACH + KPT + NM - 1) / NM, 0);    // global loads
        }
    };

    Stage rg[2];
    if (nst_here > 0) {      // the first two stages' global loads go out before the scale reduction below (their latency covers it)
        load(rg[0]);
        load(rg[1]);
    }
    // operand scales: the largest of the partial maxima the producer of X left; the weights were scaled ROW BY ROW by the pack call
    // (a.w_amax[m] = largest magnitude of row m), the epilogue divides each row by its own power of two
    if constexpr (!HALF) {
        const float m = thread_max_of_partials(a.x_amax, a.x_namax, tid, 256);
        sx = pow2_scale(block_max(m, biasS));
        if (tid < BM) iswS[tid] = m0 + tid < a.M ? 1.f / pow2_scale(a.w_amax[m0 + tid]) : 1.f;
        __syncthreads();      // (biasS was the reduction's scratch)
    }
    if (tid < BM) biasS[tid] = (a.bias != nullptr && m0 + tid < a.M) ? a.bias[m0 + tid] : 0.f;
    if (nst_here == 0) __syncthreads();      // (with stages, the barriers below order biasS before the epilogue)
    if (nst_here > 0) {
        Operands op[2];
        stash(rg[0], 0);
        __syncthreads();
        load(rg[0]);
        fetch(op[0], 0);
        stash(rg[1], 1);
        __syncthreads();
        load(rg[1]);
        const int nst2 = (nst_here + 1) & ~1;      // an odd count is rounded up: the dead stage gathered zeros
        for (int s = 0; s < nst2; s += 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fetch(op[t ^ 1], t ^ 1);           // operands of stage s+t+1
                mma(op[t]);                        // stage s+t
                stash(rg[t], t);                   // stage s+t+2
                load(rg[t]);                       // stage s+t+4
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // epilogue (as igemm2_kernel): scale back (powers of two: exact), bias + activation or raw partial sum of a K split
    const float isx = 1.f / sx;
    const int YhYw = a.Yh * a.Yw;
    bool bad = false;
    if constexpr (!HALF) {       // (before the ragged-tile `continue`s below: every lane of the wave takes part in the ballot)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) bad |= is_nonfinite(acc[i][j][r]);
        report_nonfinite(a.ovf, bad);
    }
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int pix = p0 + wp * WPT + j * 32 + lo;
        if (pix >= Ptot) continue;
        int n, oy, ox;
        pix_coord(a, P, pix, n, oy, ox);
        if (a.ksplit > 1) {
            float* Yp = a.Ypart + ((size_t)blockIdx.z * a.N + n) * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int m = m0 + ml;
                    if (m < a.M) Yp[(size_t)m * YhYw] = (acc[i][j][r] * isx) * (HALF ? 1.f : iswS[ml]);
                }
            continue;
        }
        TA* Yp = (TA*)a.Y + (size_t)n * a.M * YhYw + oy * a.Yw + ox;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m0 + ml < a.M) st1(Yp + (size_t)(m0 + ml) * YhYw, act_apply((acc[i][j][r] * isx) * (HALF ? 1.f : iswS[ml]) + biasS[ml], a.act, a.slope));
            }
    }
}

struct WgradArgs {
    const void* dY;   // [N][M][Ho][Wo], storage type TA
    const void* X;    // [N][Cg][Hg][Wg], storage type TA
    int dtype;
    float* Wp;        // [splits][M][Kp]   (k = tap*Cgp + c)
    int M, Kp, N, Cg, Cgp, Hg, Wg, Ho, Wo;
    int sl, pad, S;
    int magicS;  // ceil(65536 / S): tap / S == (tap * magicS) >> 16 for tap <= 512
    int Ptot, chunks_per_split;
    unsigned x_bytes, dy_bytes;
};

// ------------------------------------------------------------------------------------
// Small-M path (M <= 4 output channels): the generator head (64->3), the last PatchGAN / Elo-head conv
// (->1) and every data gradient that lands on an image (3-4 channels).  A 32-row MFMA tile would be >= 87 %
// padding there, so these run on the vector ALU: one thread = one pixel x 4 outputs, weights broadcast
// through the scalar cache as [k][4] rows, gathers coalesced along pixels.  Bound: L1/TA (one 4-byte
// gather per 4 FMA).
// ------------------------------------------------------------------------------------
__global__ void transpose4_kernel(const float* __restrict__ A, float* __restrict__ At, int M, int Kp) {
    const int total = Kp * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i >> 2, m = i & 3;
        At[i] = m < M ? A[(size_t)m * Kp + k] : 0.f;
    }
}

// strip-kernel weights: Ws[c][sj][8][4] from A[m][(ri*nS + sj)*Cgp + c] (zero for ri >= nR, m >= M)
__global__ void pack_strip_kernel(const float* __restrict__ A, float* __restrict__ Ws, int M, int Cg, int Cgp, int nR, int nS) {
    const int total = Cg * nS * 32;
    const int Kp = nR * nS * Cgp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i & 3, ri = (i >> 2) & 7, cs = i >> 5;
        const int c = cs / nS, sj = cs - c * nS;
        Ws[i] = (m < M && ri < nR) ? A[(size_t)m * Kp + (ri * nS + sj) * Cgp + c] : 0.f;
    }
}

static constexpr unsigned SM_INV = 0x40000000u;  // row/column marker: any sum with it is >= 1 GiB => out of range

// 64 pixels per workgroup; the 4 waves split the channels of the gathered tensor and are summed through
// LDS.  Loop order channel -> tap keeps one channel's (R x S) neighbourhood L1-resident across its taps;
// the separable gather offsets (row part, column part) are tabulated per pixel in LDS once per workgroup.
template <int MODE, typename TA>
__global__ void __launch_bounds__(256) smallm_conv_kernel(IgemmArgs a) {
    constexpr unsigned ES = sizeof(TA);
    __shared__ unsigned rowoff[12][64], coloff[12][64];
    __shared__ float red[3][4][64];
    const PhaseArgs& P = a.ph[blockIdx.y];
    const int Ptot = P.Ptot;
    const int p0 = blockIdx.x * 64;
    if (p0 >= Ptot) return;
    const int tid = threadIdx.x;
    const int pl = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = p0 + pl;
    const bool pvalid = pg < Ptot;
    const int HsWs = P.Hs * P.Ws, HgWg = a.Hg * a.Wg;
    int n = 0, py = 0, px = 0;
    if (pvalid) {
        n = pg / HsWs;
        const int rem = pg - n * HsWs;
        const int sy = rem / P.Ws;
        py = sy * a.ostep + P.fy;
        px = (rem - sy * P.Ws) * a.ostep + P.fx;
    }
    // separable offset tables (bytes): wave w fills entries w, w+4, w+8 of its pixel lane
    for (int i = wave; i < P.nR; i += 4) {
        const int r = P.r0 + i * a.tstep;
        int iy;
        bool ok = true;
        if (MODE == MODE_BWD) {
            const int ty = py + a.pad - r;
            iy = ty >> a.sl;
            ok = ty >= 0 && iy < a.Hg;
        } else {
            iy = (py << a.sl) - a.pad + r;
            if (MODE == MODE_FWD_REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.Hg ? 2 * (a.Hg - 1) - iy : iy;
            } else {
                ok = (unsigned)iy < (unsigned)a.Hg;
            }
        }
        rowoff[i][pl] = ok ? (unsigned)(iy * a.Wg) * ES : SM_INV;
    }
    for (int j = wave; j < P.nS; j += 4) {
        const int s = P.s0 + j * a.tstep;
        int ix;
        bool ok = true;
        if (MODE == MODE_BWD) {
            const int tx = px + a.pad - s;
            ix = tx >> a.sl;
            ok = tx >= 0 && ix < a.Wg;
        } else {
            ix = (px << a.sl) - a.pad + s;
            if (MODE == MODE_FWD_REFLECT) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.Wg ? 2 * (a.Wg - 1) - ix : ix;
            } else {
                ok = (unsigned)ix < (unsigned)a.Wg;
            }
        }
        coloff[j][pl] = ok ? (unsigned)ix * ES : SM_INV;
    }
    __syncthreads();
    const unsigned vbase = pvalid ? (unsigned)(n * a.Cg * HgWg) * ES : SM_INV;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const float4* __restrict__ At = reinterpret_cast<const float4*>(P.A);  // [Kp][4]
    const int cpw = (a.Cg + 3) >> 2;
    const int c_lo = wave * cpw;
    const int c_hi = (c_lo + cpw < a.Cg) ? c_lo + cpw : a.Cg;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
        const unsigned soff = (unsigned)(c * HgWg) * ES;
        for (int ri = 0; ri < P.nR; ++ri) {
            const unsigned ro = vbase + rowoff[ri][pl];
            const float4* __restrict__ wrow = At + (ri * P.nS) * a.Cgp + c;
#pragma unroll 4
            for (int sj = 0; sj < P.nS; ++sj) {
                const float x = ldx<TA>(rX, ro + coloff[sj][pl], soff);
                const float4 w = wrow[sj * a.Cgp];  // wave-uniform address -> scalar load
                acc0 += x * w.x; acc1 += x * w.y; acc2 += x * w.z; acc3 += x * w.w;
            }
        }
    }
    if (wave > 0) {
        red[wave - 1][0][pl] = acc0; red[wave - 1][1][pl] = acc1; red[wave - 1][2][pl] = acc2; red[wave - 1][3][pl] = acc3;
    }
    __syncthreads();
    if (wave > 0 || !pvalid) return;
    const float out[4] = {acc0 + (red[0][0][pl] + red[1][0][pl]) + red[2][0][pl], acc1 + (red[0][1][pl] + red[1][1][pl]) + red[2][1][pl],
                          acc2 + (red[0][2][pl] + red[1][2][pl]) + red[2][2][pl], acc3 + (red[0][3][pl] + red[1][3][pl]) + red[2][3][pl]};
    const int YhYw = a.Yh * a.Yw;
    TA* Yp = (TA*)a.Y + (size_t)n * a.M * YhYw + py * a.Yw + px;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m < a.M) {
            float v = out[m];
            if (a.bias) v += a.bias[m];
            st1(Yp + (size_t)m * YhYw, act_apply(v, a.act, a.slope));
        }
    }
}

// Strip variant: one thread = PX vertically consecutive pixels of one column x MO outputs; lanes run along the row, so
// every gather instruction reads consecutive addresses.  For a fixed (channel, filter column) the PX + NR - 1 input
// values above/below the strip are loaded ONCE into registers and reused by all NR row taps of all PX pixels (sliding
// window): MO * NR * PX fused multiply-adds per PX + NR - 1 gathers instead of MO per gather, which moves the kernel
// from the L1/TA bound of smallm_conv_kernel towards the vector-ALU bound.  Weights come in through the scalar cache
// ([c][sj][8][4], wave-uniform addresses, no branches).  The 4 waves split the channels and are summed through LDS.
// Needs unit pixel stride along the column in the gathered tensor: forward with stride 1, or any data-gradient phase.
template <int MODE, int NR, int MO, typename TA>
__global__ void __launch_bounds__(256) smallm_strip_kernel(IgemmArgs a) {
    constexpr unsigned ES = sizeof(TA);
    constexpr int PX = 8, NW = PX + NR - 1;
    constexpr bool BWD = MODE == MODE_BWD;
    __shared__ unsigned coltab[12][64];
    __shared__ float red[3][MO * PX][64];
    const PhaseArgs& P = a.ph[blockIdx.y];
    const int spc = (P.Hs + PX - 1) / PX;   // strips per column
    const int nstrips = a.N * spc * P.Ws;
    if ((int)(blockIdx.x * 64) >= nstrips) return;
    const int tid = threadIdx.x;
    const int pl = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sg = blockIdx.x * 64 + pl;
    const bool svalid = sg < nstrips;
    const int HgWg = a.Hg * a.Wg;
    int n = 0, sy0 = 0, sx = 0;
    if (svalid) {
        n = sg / (spc * P.Ws);
        const int rem = sg - n * spc * P.Ws;
        const int ss = rem / P.Ws;
        sy0 = ss * PX;
        sx = rem - ss * P.Ws;
    }
    const int px = sx * a.ostep + P.fx;
    for (int j = wave; j < P.nS; j += 4) {   // column part of the gather offset, per filter column
        const int sc = P.s0 + j * a.tstep;
        int ix;
        bool ok = svalid;
        if (BWD) {
            const int tx = px + a.pad - sc;
            ix = tx >> a.sl;
            ok = ok && tx >= 0 && ix < a.Wg;
        } else {
            ix = px - a.pad + sc;
            if (MODE == MODE_FWD_REFLECT) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.Wg ? 2 * (a.Wg - 1) - ix : ix;
            } else {
                ok = ok && (unsigned)ix < (unsigned)a.Wg;
            }
        }
        coltab[j][pl] = ok ? (unsigned)ix * ES : SM_INV;
    }
    // row part (+ image base): window position k holds input row y0 + k; pixel j and row tap ri meet at k = j + ri
    // (forward) or k = j - ri + NR - 1 (data gradient: source row = sub-grid row + q0 - ri)
    unsigned rowoff[NW];
    {
        const int y0 = BWD ? sy0 + ((P.fy + a.pad - P.r0) >> a.sl) - (NR - 1) : sy0 - a.pad + P.r0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int iy = y0 + k;
            if (MODE == MODE_FWD_REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.Hg ? 2 * (a.Hg - 1) - iy : iy;
            }
            const bool ok = (unsigned)iy < (unsigned)a.Hg;   // (reflect: strips past the last row are never stored)
            rowoff[k] = ok ? (unsigned)(n * a.Cg * HgWg + iy * a.Wg) * ES : SM_INV;
        }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const int nS = P.nS;
    // channels: split over blockIdx.z (few-strip launches, partial sums reduced by splitk_reduce_kernel), then over waves
    const int cps = a.ksplit > 1 ? (a.Cg + a.ksplit - 1) / a.ksplit : a.Cg;
    const int cz0 = (int)blockIdx.z * cps;
    const int cz1 = cz0 + cps < a.Cg ? cz0 + cps : a.Cg;
    const int cpw = (cz1 - cz0 + 3) >> 2;
    const int c_lo = cz0 + wave * cpw;
    const int c_hi = (c_lo + cpw < cz1) ? c_lo + cpw : cz1;
    float acc[MO][PX];
#pragma unroll
    for (int m = 0; m < MO; ++m)
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[m][j] = 0.f;

    const float4* __restrict__ Ws4 = reinterpret_cast<const float4*>(P.As);   // [c][sj][8] float4
    auto issue = [&](float (&buf)[NW], float4 (&wb)[NR], int c, int sj) {
        const unsigned co = coltab[sj][pl];
        const unsigned so = (unsigned)(c * HgWg) * ES;
#pragma unroll
        for (int k = 0; k < NW; ++k) buf[k] = ldx<TA>(rX, rowoff[k] + co, so);
        const float4* __restrict__ wr = Ws4 + (size_t)(c * nS + sj) * 8;   // wave-uniform -> scalar loads, no branches
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) wb[ri] = wr[ri];
    };
    auto compute = [&](const float (&buf)[NW], const float4 (&wb)[NR]) {
#pragma unroll
        for (int ri = 0; ri < NR; ++ri) {
            const float w[4] = {wb[ri].x, wb[ri].y, wb[ri].z, wb[ri].w};
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const float x = buf[BWD ? j - ri + NR - 1 : j + ri];
#pragma unroll
                for (int m = 0; m < MO; ++m) acc[m][j] += x * w[m];
            }
        }
    };
    const int T = (c_hi - c_lo) * nS;
    if (T > 0) {
        float b0[NW], b1[NW];
        float4 w0[NR], w1[NR];
        int c = c_lo, sj = 0;           // (c, sj) of the stage being issued
        auto adv = [&](int& cx, int& sx_) {
            if (++sx_ == nS) {
                sx_ = 0;
                ++cx;
            }
        };
        issue(b0, w0, c, sj);
        adv(c, sj);
        for (int t = 0; t < T; t += 2) {
            if (t + 1 < T) {
                issue(b1, w1, c, sj);
                adv(c, sj);
            }
            compute(b0, w0);
            if (t + 1 < T) {
                if (t + 2 < T) {
                    issue(b0, w0, c, sj);
                    adv(c, sj);
                }
                compute(b1, w1);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MO; ++m)
#pragma unroll
            for (int j = 0; j < PX; ++j) red[wave - 1][m * PX + j][pl] = acc[m][j];
    }
    __syncthreads();
    if (wave > 0 || !svalid) return;
    const int YhYw = a.Yh * a.Yw;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        if (sy0 + j >= P.Hs) break;
        const int py = (sy0 + j) * a.ostep + P.fy;
        const size_t yo = (size_t)n * a.M * YhYw + py * a.Yw + px;
#pragma unroll
        for (int m = 0; m < MO; ++m) {
            if (m < a.M) {
                float v = acc[m][j] + (red[0][m * PX + j][pl] + red[1][m * PX + j][pl]) + red[2][m * PX + j][pl];
                if (a.ksplit > 1) {   // raw fp32 partial sum; bias / activation happen in splitk_reduce_kernel
                    a.Ypart[(size_t)blockIdx.z * a.N * a.M * YhYw + yo + (size_t)m * YhYw] = v;
                    continue;
                }
                if (a.bias) v += a.bias[m];
                st1((TA*)a.Y + yo + (size_t)m * YhYw, act_apply(v, a.act, a.slope));
            }
        }
    }
}

// Variant for phases with few taps (<= 9, e.g. the stride phases of 4x4/s2 and 11x11/s4 data gradients): one
// thread per pixel, tap-outer loop; no tables, no cross-wave reduction.
template <int MODE, typename TA>
__global__ void __launch_bounds__(256) smallm_conv_fewtaps_kernel(IgemmArgs a) {
    constexpr unsigned ES = sizeof(TA);
    const PhaseArgs& P = a.ph[blockIdx.y];
    const int Ptot = P.Ptot;
    if ((int)(blockIdx.x * 256) >= Ptot) return;
    const int pg = blockIdx.x * 256 + threadIdx.x;
    const bool pvalid = pg < Ptot;
    const Geom g{a.Hg, a.Wg, a.sl, a.pad};
    const int HsWs = P.Hs * P.Ws, HgWg = a.Hg * a.Wg;
    int n = 0, py = 0, px = 0;
    if (pvalid) {
        n = pg / HsWs;
        const int rem = pg - n * HsWs;
        const int sy = rem / P.Ws;
        py = sy * a.ostep + P.fy;
        px = (rem - sy * P.Ws) * a.ostep + P.fx;
    }
    const int vbase = n * a.Cg * HgWg;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const float4* __restrict__ At = reinterpret_cast<const float4*>(P.A);  // [Kp][4]
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int ri = 0; ri < P.nR; ++ri) {
        for (int sj = 0; sj < P.nS; ++sj) {
            int off;
            const bool ok = tap_offset<MODE>(g, py, px, P.r0 + ri * a.tstep, P.s0 + sj * a.tstep, off) && pvalid;
            const unsigned voff = ok ? (unsigned)(vbase + off) * ES : OOB;
            const int kbase = (ri * P.nS + sj) * a.Cgp;
            int c = 0;
            for (; c + 8 <= a.Cg; c += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = ldx<TA>(rX, voff, (unsigned)((c + u) * HgWg) * ES);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 w = At[kbase + c + u];  // wave-uniform address -> scalar load
                    acc0 += x[u] * w.x; acc1 += x[u] * w.y; acc2 += x[u] * w.z; acc3 += x[u] * w.w;
                }
            }
            for (; c < a.Cg; ++c) {
                const float x = ldx<TA>(rX, voff, (unsigned)(c * HgWg) * ES);
                const float4 w = At[kbase + c];
                acc0 += x * w.x; acc1 += x * w.y; acc2 += x * w.z; acc3 += x * w.w;
            }
        }
    }
    if (!pvalid) return;
    const int YhYw = a.Yh * a.Yw;
    TA* Yp = (TA*)a.Y + (size_t)n * a.M * YhYw + py * a.Yw + px;
    const float out[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m < a.M) {
            float v = out[m];
            if (a.bias) v += a.bias[m];
            st1(Yp + (size_t)m * YhYw, act_apply(v, a.act, a.slope));
        }
    }
}

// Weight gradient for M <= 4: Wp[split][m][kb..kb+15] = sum_pix dY[m][pix] * G(k; pix).  One workgroup per
// 16-column slab of K (one tap, 16 channels: needs Cgp % 16 == 0) and pixel split; every thread keeps the
// 4x16 partial sums of its pixels in registers and the workgroup reduces them once at the end.
template <int MODE, typename TA>
__global__ void __launch_bounds__(256) smallm_wgrad_kernel(WgradArgs a) {
    constexpr unsigned ES = sizeof(TA);
    __shared__ float red[4][64];
    const int tid = threadIdx.x;
    const int kb = blockIdx.x * 16;
    const int tap = kb / a.Cgp, c0 = kb - tap * a.Cgp;
    const int r = (tap * a.magicS) >> 16, s = tap - r * a.S;
    const Geom g{a.Hg, a.Wg, a.sl, a.pad};
    const int HoWo = a.Ho * a.Wo, HgWg = a.Hg * a.Wg;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dY, a.dy_bytes);
    float acc[4][16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
    const int pbeg = blockIdx.y * a.chunks_per_split * 32;
    int pend = pbeg + a.chunks_per_split * 32;
    if (pend > a.Ptot) pend = a.Ptot;
    for (int pg = pbeg + tid; pg < pend; pg += 256) {
        const int n = pg / HoWo;
        const int rem = pg - n * HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        int off;
        const bool ok = tap_offset<MODE>(g, oy, ox, r, s, off);
        const unsigned voff = ok ? (unsigned)(n * a.Cg * HgWg + off) * ES : OOB;
        const unsigned yoff = (unsigned)(n * a.M * HoWo + rem) * ES;
        float dy[4], x[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) dy[m] = ldx<TA>(rY, m < a.M ? yoff : OOB, (unsigned)(m * HoWo) * ES);
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = ldx<TA>(rX, (c0 + j < a.Cg) ? voff : OOB, (unsigned)((c0 + j) * HgWg) * ES);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[m][j] += dy[m] * x[j];
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float v = wave_sum(acc[m][j]);
            if (lane == 0) red[wave][m * 16 + j] = v;
        }
    __syncthreads();
    if (tid < 64) {
        const int m = tid >> 4, j = tid & 15;
        if (m < a.M && kb + j < a.Kp)
            a.Wp[((size_t)blockIdx.y * a.M + m) * a.Kp + kb + j] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// y = act( sum_s part[s] + bias[channel] ) over the split-K partial sums
template <typename TA>
__global__ void splitk_reduce_kernel(const float* __restrict__ part, TA* __restrict__ y, const float* __restrict__ bias,
                                     int ks, size_t n, int M, int HW, int act, float slope) {
    const size_t n4 = (n & 3) ? 0 : (n >> 2);  // 16-byte path only when every split's base stays aligned
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 acc = reinterpret_cast<const float4*>(part)[i];
        for (int s = 1; s < ks; ++s) {
            const float4 v = reinterpret_cast<const float4*>(part + (size_t)s * n)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = i * 4 + u;
            if (bias) o[u] += bias[(e / HW) % M];
            o[u] = act_apply(o[u], act, slope);
        }
        st4(y + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
    for (size_t e = (n4 << 2) + blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < n; e += stride) {
        float acc = 0.f;
        for (int s = 0; s < ks; ++s) acc += part[(size_t)s * n + e];
        if (bias) acc += bias[(e / HW) % M];
        st1(y + e, act_apply(acc, act, slope));
    }
}

// ------------------------------------------------------------------------------------
// weight re-layout kernels
// ------------------------------------------------------------------------------------
// forward: A[k][tap][c] (Cgp-padded) from w[K][C][R][S]
// chunked != 0 (igemm2_kernel): K order (16-channel chunk, tap, channel-in-chunk), so the 9..49 taps of one
// channel chunk are consecutive K stages and re-read the same small input tile from L1/L2 instead of
// streaming the whole input once per tap from beyond L2
__global__ void repack_fwd_kernel(const float* __restrict__ w, float* __restrict__ A, int K, int C, int Cgp,
                                  int RS, int chunked) {
    const int Kp = RS * Cgp;
    const size_t total = (size_t)K * Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / Kp);
        const int j = (int)(i - (size_t)k * Kp);
        int tap, c;
        if (chunked) {
            const int t2 = j >> 4;
            const int cc = t2 / RS;
            tap = t2 - cc * RS;
            c = cc * 16 + (j & 15);
        } else {
            tap = j / Cgp;
            c = j - tap * Cgp;
        }
        A[i] = (c < C) ? w[((size_t)k * C + c) * RS + tap] : 0.f;
    }
}
// backward-data, one stride phase: A[c][(ri,sj)][k] (Kgp-padded) from w[K][C][R][S]
__global__ void repack_bwd_kernel(const float* __restrict__ w, float* __restrict__ A, int K, int C, int Kgp,
                                  int R, int S, int r0, int s0, int tstep, int nR, int nS, int chunked, int fold) {
    const int Kp = nR * nS * Kgp;
    const size_t total = (size_t)C * Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / Kp);
        const int j = (int)(i - (size_t)c * Kp);
        int t, k;
        if (chunked) {
            const int t2 = j >> 4;
            const int kc = t2 / (nR * nS);
            t = t2 - kc * (nR * nS);
            k = kc * 16 + (j & 15);
        } else {
            t = j / Kgp;
            k = j - t * Kgp;
        }
        const int ri = t / nS, sj = t - ri * nS;
        const int r = r0 + ri * tstep, s = s0 + sj * tstep;
        float v = (k < K) ? w[(((size_t)k * C + c) * R + r) * S + s] : 0.f;
        // reflection pad 1, 3 taps: output row 1 also receives padded row 0 = input row 1 through tap 0, from the
        // source row that its tap 2 reads (fold 1); output row H-2 the other way round (fold 2)
        if (k < K && fold == 1 && r == R - 1) v += w[(((size_t)k * C + c) * R + 0) * S + s];
        if (k < K && fold == 2 && r == 0) v += w[(((size_t)k * C + c) * R + (R - 1)) * S + s];
        A[i] = v;
    }
}

// all phases / row classes of one data-gradient weight pack in ONE launch (blockIdx.y = entry): these kernels are
// launch-latency bound (a few microseconds each, up to 16 per layer)
struct PackBwdArgs {
    const float* w;
    int K, C, Kgp, R, S, tstep, chunked, n;
    struct Entry {
        float* A;
        int r0, s0, nR, nS, fold;
    } e[16];
};
__global__ void repack_bwd_multi_kernel(PackBwdArgs a) {
    const PackBwdArgs::Entry& E = a.e[blockIdx.y];
    const int RS = E.nR * E.nS;
    const int Kp = RS * a.Kgp;
    const size_t total = (size_t)a.C * Kp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i / Kp);
        const int j = (int)(i - (size_t)c * Kp);
        int t, k;
        if (a.chunked) {
            const int t2 = j >> 4;
            const int kc = t2 / RS;
            t = t2 - kc * RS;
            k = kc * 16 + (j & 15);
        } else {
            t = j / a.Kgp;
            k = j - t * a.Kgp;
        }
        const int ri = t / E.nS, sj = t - ri * E.nS;
        const int r = E.r0 + ri * a.tstep, sx = E.s0 + sj * a.tstep;
        float v = 0.f;
        if (k < a.K) {
            const float* wk = a.w + ((size_t)k * a.C + c) * a.R * a.S;
            v = wk[r * a.S + sx];
            if (E.fold == 1 && r == a.R - 1) v += wk[sx];                       // see repack_bwd_kernel
            if (E.fold == 2 && r == 0) v += wk[(a.R - 1) * a.S + sx];
        }
        E.A[i] = v;
    }
}

// fold the gradient of a reflection-padded tensor back onto the unpadded tensor
// grid = (row groups, planes): one thread = 4 consecutive pixels of one row, plane-local 32-bit index math
template <typename TA>
__global__ void reflect_fold_kernel(const TA* __restrict__ t, TA* __restrict__ dx, int NC, int H, int W,
                                    int pad) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int W4 = (W + 3) >> 2;
    const TA* tp = t + (size_t)blockIdx.y * Hp * Wp;
    TA* dp = dx + (size_t)blockIdx.y * H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W4; i += gridDim.x * blockDim.x) {
        const int y = i / W4, x0 = (i - y * W4) * 4;
        int ys[3], ny = 0;
        ys[ny++] = y + pad;
        if (y >= 1 && y <= pad) ys[ny++] = pad - y;
        if (y >= H - 1 - pad && y <= H - 2) ys[ny++] = pad + 2 * (H - 1) - y;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a) {
            const TA* row = tp + ys[a] * Wp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                if (x >= W) continue;
                float v = ld1(row + x + pad);
                if (x >= 1 && x <= pad) v += ld1(row + pad - x);
                if (x >= W - 1 - pad && x <= W - 2) v += ld1(row + pad + 2 * (W - 1) - x);
                acc[j] += v;
            }
        }
        if ((W & 3) == 0) {
            st4(dp + y * W + x0, make_float4(acc[0], acc[1], acc[2], acc[3]));
        } else {
            for (int j = 0; j < 4 && x0 + j < W; ++j) st1(dp + y * W + x0 + j, acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------
// backward-weight
// ------------------------------------------------------------------------------------

// Wp[m][kcol] = sum over a pixel range of dY[m][pix] * G(kcol; pix).  Tile BM x 128 (kcol), stage = 32
// pixels.  LDS rows hold 32 pixels of one m / one kcol at pitch 36 floats (ds_read_b128 conflict-free);
// the MFMA consumes the pixels in the same permuted order for both operands.
// VECA: dY planes are a multiple of 4 pixels, so a thread fetches 4 consecutive pixels with one 16-byte load.
// KMODE: 0 = generic (Cgp % 8 == 0), 1 = SMALLC (per-thread tap), 2 = ONETAP (Cgp % 128 == 0: the whole 128-column
// tile lies inside one filter tap -> one spatial offset per stage, straight-line code, interleaved schedule)
template <int MODE, int BM, int KMODE, bool VECA, typename TA>
__global__ void __launch_bounds__(256) wgrad_kernel(WgradArgs a) {
    constexpr unsigned ES = sizeof(TA);
    constexpr bool SMALLC = KMODE == 1;
    constexpr bool ONETAP = KMODE == 2;
    constexpr int BN = 128;
    constexpr int WM = (BM == 128) ? 2 : 1;
    constexpr int WN = 4 / WM;
    constexpr int WMT = BM / WM, WNT = BN / WN;
    constexpr int MI = WMT / 32, NJ = WNT / 32;
    constexpr int PT = 36;
    constexpr int AR = BM / 8, BR = BN / 8;
    constexpr int AV = BM / 32;  // float4 chunks of dY per thread (VECA)
    __shared__ __attribute__((aligned(16))) float As[2][BM * PT];
    __shared__ __attribute__((aligned(16))) float Gs[2][BN * PT];

    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int pl = tid & 31, rg = tid >> 5;

    const int nKt = (a.Kp + BN - 1) / BN;
    const int kt = blockIdx.x % nKt, mt = blockIdx.x / nKt;
    const int m0 = mt * BM, kb = kt * BN;
    const int split = blockIdx.y;
    const int HoWo = a.Ho * a.Wo, HgWg = a.Hg * a.Wg;
    const Geom g{a.Hg, a.Wg, a.sl, a.pad};
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dY, a.dy_bytes);

    // K-column bookkeeping.  Fast path (Cgp % 8 == 0): for row-group offset i the tap of column
    // kb + 8*i + rg is block-uniform and c = c_i + rg.
    int tap_b = 0, c_b = 0;
    if (!SMALLC) {
        tap_b = kb / a.Cgp;
        c_b = kb - tap_b * a.Cgp;
    }

    float areg[AR], breg[BR];
    auto load_stage = [&](int chunk) {
        // ---- dY tile -----------------------------------------------------------------
        if (VECA) {
            const int pc = (tid & 7) * 4;  // same pixel quad for all of this thread's rows
            const int pg = chunk * 32 + pc;
            unsigned vb = OOB;
            {
                const int n = pg / HoWo;
                const unsigned vv = (unsigned)((n * a.M + m0) * HoWo + (pg - n * HoWo)) * ES;
                vb = (pg < a.Ptot) ? vv : OOB;
            }
#pragma unroll
            for (int j = 0; j < AV; ++j) {
                const int row = (tid >> 3) + 32 * j;
                const float4 v = ldr4<TA>(rY, ((vb != OOB) & (m0 + row < a.M)) ? vb + (unsigned)(row * HoWo) * ES : OOB, 0u);
                areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
            }
        }
        // ---- this thread's gather pixel ------------------------------------------------
        const int pg = chunk * 32 + pl;
        const bool pvalid = pg < a.Ptot;
        // (computed for out-of-range pixels too: selects instead of branches keep the stage one basic block)
        const int n = pg / HoWo;
        const int rem = pg - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        if (!VECA) {
            const unsigned vb = pvalid ? (unsigned)((n * a.M + m0 + rg) * HoWo + rem) * ES : OOB;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                areg[i] = ldr<TA>(rY, (pvalid & (m0 + rg + 8 * i < a.M)) ? vb : OOB, (unsigned)(8 * i * HoWo) * ES);
        }
        const int vbase = n * a.Cg * HgWg;
        if (ONETAP) {
            const int r = (tap_b * a.magicS) >> 16;
            const int s = tap_b - r * a.S;
            int off;
            const bool ok = tap_offset<MODE>(g, oy, ox, r, s, off) & pvalid;
            const unsigned voff = ok ? (unsigned)(vbase + off + rg * HgWg) * ES : OOB;
#pragma unroll
            for (int i = 0; i < BR; ++i)
                breg[i] = ldr<TA>(rX, (c_b + 8 * i + rg < a.Cg) ? voff : OOB, (unsigned)((c_b + 8 * i) * HgWg) * ES);
        } else if (!SMALLC) {
            int tap = tap_b, c = c_b;
            unsigned voff = OOB;
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                if (i == 0 || c == 0) {  // block-uniform: the tap changed (c is a multiple of 8)
                    const int r = (tap * a.magicS) >> 16;
                    const int s = tap - r * a.S;
                    int off;
                    const bool ok = tap_offset<MODE>(g, oy, ox, r, s, off) & pvalid;
                    voff = ok ? (unsigned)(vbase + off + rg * HgWg) * ES : OOB;
                }
                const bool okc = (c + rg < a.Cg) & (kb + 8 * i + rg < a.Kp);
                breg[i] = ldr<TA>(rX, okc ? voff : OOB, (unsigned)(c * HgWg) * ES);
                c += 8;
                if (c >= a.Cgp) {
                    c -= a.Cgp;
                    ++tap;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int kcol = kb + 8 * i + rg;
                const int tap = kcol / a.Cgp, c = kcol - tap * a.Cgp;
                const int r = (tap * a.magicS) >> 16;
                const int s = tap - r * a.S;
                int off;
                const bool ok = tap_offset<MODE>(g, oy, ox, r, s, off) && pvalid && kcol < a.Kp && c < a.Cg;
                breg[i] = ldr<TA>(rX, ok ? (unsigned)(vbase + c * HgWg + off) * ES : OOB, 0u);
            }
        }
    };
    auto store_stage = [&](int buf) {
        if (VECA) {
#pragma unroll
            for (int j = 0; j < AV; ++j)
                put4p<TA>(&As[buf][((tid >> 3) + 32 * j) * PT + (tid & 7) * 4],
                          make_float4(areg[4 * j + 0], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]));
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) put1<TA>(&As[buf][(rg + 8 * i) * PT + pl], areg[i]);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) put1<TA>(&Gs[buf][(rg + 8 * i) * PT + pl], breg[i]);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (a.Ptot + 31) / 32;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;
    const int nst = c_end - c_begin;
    if (nst > 0) {
        if constexpr (sizeof(TA) == 2) {
            zero_tile<TA>(&As[0][0], 2 * BM * PT);
            zero_tile<TA>(&Gs[0][0], 2 * BN * PT);
            __syncthreads();
        }
        load_stage(c_begin);
        store_stage(0);
        __syncthreads();
        auto compute = [&](int buf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float av[MI][4], bv[NJ][4];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const float4 t = *reinterpret_cast<const float4*>(&As[buf][(wm * WMT + i * 32 + lo) * PT + (2 * q + hi) * 4]);
                    av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float4 t = *reinterpret_cast<const float4*>(&Gs[buf][(wn * WNT + j * 32 + lo) * PT + (2 * q + hi) * 4]);
                    bv[j][0] = t.x; bv[j][1] = t.y; bv[j][2] = t.z; bv[j][3] = t.w;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][jj], bv[j][jj], acc[i][j], 0, 0, 0);
            }
        };
        // same fine-grained interleave as the forward kernel: the next stage's gathers and their address
        // arithmetic are issued between this stage's MFMAs (a wave cannot issue past a waiting MFMA)
        constexpr int NMFMA = MI * NJ * 16;
        constexpr int NLD = (VECA ? AV : AR) + BR;
        for (int st = 0; st + 1 < nst; ++st) {
            const int buf = st & 1;
            load_stage(c_begin + st + 1);
            compute(buf);
            if (ONETAP) {
#pragma unroll
                for (int gI = 0; gI < NMFMA; ++gI) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (gI < 4 * (MI + NJ)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (gI < NMFMA / 2) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, (NLD + NMFMA / 2 - 1) / (NMFMA / 2), 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            store_stage(buf ^ 1);
            __syncthreads();
        }
        compute((nst - 1) & 1);
    }
    // partial tile store: row = m, column = k (lane) -> coalesced
    float* Wp = a.Wp + (size_t)split * a.M * a.Kp;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int kcol = kb + wn * WNT + j * 32 + lo;
        if (kcol >= a.Kp) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < a.M) Wp[(size_t)m * a.Kp + kcol] = acc[i][j][r];
            }
        }
    }
}

// Weight gradient for <= 3 output channels, stride 1, <= NT x NT taps (the generator head 64 -> 3, 7x7): sliding-window
// strips like smallm_strip_kernel.  One workgroup = one input channel c and a range of strips; one thread = 8 vertically
// consecutive output pixels of one column (lanes along the row => coalesced).  The 8 x M values of dY are loaded once per
// strip; for each filter column the 8 + NT - 1 input values are loaded once and reused by all NT row taps:
// M * NT * 8 fused multiply-adds per 8 + NT - 1 gathers.  Every thread keeps the NT x NT x M partial sums of ITS pixels in
// registers; the workgroup reduces them once at the end (wave shuffles, then LDS) and writes Wp[split][m][tap * Cgp + c].
template <int MODE, int NT, typename TA>
__global__ void __launch_bounds__(256) smallm_wgrad_strip_kernel(WgradArgs a) {
    constexpr unsigned ES = sizeof(TA);
    constexpr int PX = 8, NW = PX + NT - 1, MO = 3;
    constexpr int SG = NT > 4 ? 4 : NT;     // filter columns per workgroup (blockIdx.z picks the group): keeps the
                                            // accumulators at SG x NT x 3 registers so that 2-3 waves fit a SIMD
    __shared__ float red[4][SG * NT * MO];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const int s_lo = blockIdx.z * SG;
    const int R = a.Kp / a.Cgp / a.S;       // taps: R x S
    const int HoWo = a.Ho * a.Wo, HgWg = a.Hg * a.Wg;
    const int spc = (a.Ho + PX - 1) / PX;   // strips per column
    const int nstrips = a.N * spc * a.Wo;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dY, a.dy_bytes);
    float acc[SG][NT][MO];   // [s - s_lo][r][m]
#pragma unroll
    for (int sj = 0; sj < SG; ++sj)
#pragma unroll
        for (int ri = 0; ri < NT; ++ri)
#pragma unroll
            for (int m = 0; m < MO; ++m) acc[sj][ri][m] = 0.f;
    const int sbeg = blockIdx.y * a.chunks_per_split;      // (strips per split)
    int send = sbeg + a.chunks_per_split;
    if (send > nstrips) send = nstrips;
    for (int sg = sbeg + tid; sg < send; sg += 256) {
        const int n = sg / (spc * a.Wo);
        const int rem = sg - n * spc * a.Wo;
        const int ss = rem / a.Wo;
        const int ox = rem - ss * a.Wo;
        const int oy0 = ss * PX;
        // input rows under the strip
        unsigned rowoff[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int iy = oy0 - a.pad + k;
            if (MODE == MODE_FWD_REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.Hg ? 2 * (a.Hg - 1) - iy : iy;
            }
            rowoff[k] = ((unsigned)iy < (unsigned)a.Hg) ? (unsigned)((n * a.Cg + c) * HgWg + iy * a.Wg) * ES : SM_INV;
        }
        auto col_off = [&](int sj) {
            int ix = ox - a.pad + s_lo + sj;
            if (MODE == MODE_FWD_REFLECT) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.Wg ? 2 * (a.Wg - 1) - ix : ix;
            }
            return (s_lo + sj < a.S && (unsigned)ix < (unsigned)a.Wg) ? (unsigned)ix * ES : SM_INV;
        };
        float xin[2][NW];
        {
            const unsigned co = col_off(0);
#pragma unroll
            for (int k = 0; k < NW; ++k) xin[0][k] = ldx<TA>(rX, rowoff[k] + co, 0u);
        }
        // dY of the strip (rows past the end: 0)
        float dyv[MO][PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = (oy0 + j < a.Ho) ? (unsigned)(n * a.M * HoWo + (oy0 + j) * a.Wo + ox) * ES : OOB;
#pragma unroll
            for (int m = 0; m < MO; ++m) dyv[m][j] = m < a.M ? ldx<TA>(rY, vo, (unsigned)(m * HoWo) * ES) : 0.f;
        }
#pragma unroll
        for (int sj = 0; sj < SG; ++sj) {
            if (sj + 1 < SG) {   // next filter column in flight while this one is consumed
                const unsigned co = col_off(sj + 1);
#pragma unroll
                for (int k = 0; k < NW; ++k) xin[(sj + 1) & 1][k] = ldx<TA>(rX, rowoff[k] + co, 0u);
            }
            // One v_fmac_f32 per term, written out.  WORKAROUND, cause not established: left to itself the compiler pairs the
            // accumulators into v_pk_fma_f32 with operand selects, and THAT build of this kernel returned different sums from run to
            // run whenever an f16-MFMA kernel of another stream shared the CUs (scripts/diag_race.py: 30 / 30; alone, or beside
            // fp32-MFMA / copy kernels, exact).  Round 4's ISA study (scripts/micro/head_wgrad_isa.md) shows the compiler's wait counts
            // are correct (no read or overwrite of a register with an outstanding load) and that the one form unique to that build is
            // `op_sel:[0,1,0]` (high dword of src1 broadcast) -- absent from every other kernel; tests/test_isa_guard.py bans it from
            // the library.  Same arithmetic, same order.
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int j = 0; j < PX; ++j)
#pragma unroll
                    for (int m = 0; m < MO; ++m)
                        asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[sj][ri][m]) : "v"(dyv[m][j]), "v"(xin[sj & 1][j + ri]));
        }
    }
    // workgroup reduction: 6 shuffle steps inside each wave, then the 4 waves through LDS
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int sj = 0; sj < SG; ++sj)
#pragma unroll
        for (int ri = 0; ri < NT; ++ri)
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                float v = acc[sj][ri][m];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
                if (lane == 0) red[wave][(sj * NT + ri) * MO + m] = v;
            }
    __syncthreads();
    if (tid < SG * NT * MO) {
        const int m = tid % MO, ri = (tid / MO) % NT, sj = s_lo + tid / (MO * NT);
        if (m < a.M && ri < R && sj < a.S) {
            const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            a.Wp[((size_t)blockIdx.y * a.M + m) * a.Kp + (ri * a.S + sj) * a.Cgp + c] = v;
        }
    }
}

// ------------------------------------------------------------------------------------
// Weight gradient, pipelined version for tap-aligned K tiles (padded channel count a multiple of 128, or 64):
// same recipe as igemm2_kernel -- few non-scalar instructions per matrix instruction, every wait long after its
// request, source order = issue order.
//   tile  : BM output channels x 128 K-columns (one filter tap x 128 channels, or two taps x 64), reduction over
//           pixels in stages of 32, split over pixel ranges (blockIdx.y) with a deterministic second-pass sum
//   offsets: the per-pixel gather offsets (padding / reflection / stride arithmetic, two integer divisions) are
//           computed by all 256 threads for 8 stages at a time into a 16-slot LDS ring; a stage then needs
//           NT + 1 four-byte LDS reads
//   stage t: group 0/1 of the MFMA chain + LDS writes of stage t+1 (loaded one stage ago)
//            group 2/3 + global loads of stage t+2; the barrier sits between group 2 and 3, the operands of
//            stage t+1's first group are read under group 3
template <int MODE, int BM, bool VECA, int NT, typename TA>
__global__ void __launch_bounds__(256) wgrad2_kernel(WgradArgs a) {
    constexpr unsigned ES = sizeof(TA);
    constexpr int BN = 128;
    constexpr int WM = (BM == 128) ? 2 : 1;
    constexpr int WN = 4 / WM;
    constexpr int WMT = BM / WM, WNT = BN / WN;
    constexpr int MI = WMT / 32, NJ = WNT / 32;
    constexpr int PT = 36;
    constexpr int AR = BM / 8, BR = BN / 8;
    constexpr int AV = BM / 32;
    constexpr int NA = VECA ? AV : AR;          // dY loads / LDS writes per thread and stage
    constexpr int RING = 16;
    __shared__ __attribute__((aligned(16))) float As[2][BM * PT];
    __shared__ __attribute__((aligned(16))) float Gs[2][BN * PT];
    __shared__ unsigned xoffT[RING][NT][32];
    __shared__ unsigned yoffT[RING][32];

    const int tid = threadIdx.x;
    const int lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int pl = tid & 31, rg = tid >> 5;

    const int nKt = (a.Kp + BN - 1) / BN;
    const int kt = blockIdx.x % nKt, mt = blockIdx.x / nKt;
    const int m0 = mt * BM, kb = kt * BN;
    const int split = blockIdx.y;
    const int HoWo = a.Ho * a.Wo, HgWg = a.Hg * a.Wg;
    const int HoWo4 = HoWo * (int)ES, HgWg4 = HgWg * (int)ES;     // bytes of one plane
    const Geom g{a.Hg, a.Wg, a.sl, a.pad};
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dY, a.dy_bytes);
    const int tap_b = kb / a.Cgp, c_b = kb - tap_b * a.Cgp;
    const int ntaps = a.Kp / a.Cgp;

    const int nchunks = (a.Ptot + 31) / 32;
    const int c_begin = split * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;
    const int nst = c_end - c_begin;

    // offsets of 8 stages (relative chunks tb .. tb+7) -> ring slots
    auto refill = [&](int tb) {
        const int rel = tb + (tid >> 5);
        const int slot = rel & (RING - 1);
        const int pg = (c_begin + rel) * 32 + pl;
        const bool valid = (pg < a.Ptot) & (c_begin + rel < c_end);
        const int n = pg / HoWo;
        const int rem = pg - n * HoWo;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        yoffT[slot][pl] = valid ? (unsigned)((n * a.M + m0) * HoWo + rem) * ES : OOB;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int tap = tap_b + ti;
            const int r = tap / a.S, sx = tap - r * a.S;
            int off;
            const bool ok = tap_offset<MODE>(g, oy, ox, r, sx, off) & valid & (tap < ntaps);
            xoffT[slot][ti][pl] = ok ? (unsigned)(n * a.Cg * HgWg + off) * ES : OOB;
        }
    };

    // per-thread constant parts of the load offsets (bit 31 = row out of range)
    const int pc = (tid & 7) * 4;
    const unsigned yrow = VECA ? (unsigned)((tid >> 3) * HoWo4) : (unsigned)(rg * HoWo4);
    unsigned yflag[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = VECA ? (tid >> 3) + 32 * j : rg + 8 * j;
        yflag[j] = (m0 + row < a.M) ? 0u : OOB;
    }
    const unsigned xrow = (unsigned)(rg * HgWg4);

    float areg[VECA ? 4 * AV : AR], breg[BR];
    unsigned yo = OOB, xo[NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) xo[ti] = OOB;
    unsigned yraw = OOB, xraw[NT];
    auto read_offsets = [&](int rel) {   // ring-table entries of relative chunk `rel` (raw: used one group later)
        const int slot = rel & (RING - 1);
        yraw = yoffT[slot][VECA ? pc : pl];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) xraw[ti] = xoffT[slot][ti][pl];
    };
    auto combine_offsets = [&]() {
        yo = yraw + yrow;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) xo[ti] = xraw[ti] + xrow;
    };
    auto load_a = [&](int j) {
        if (VECA) {
            const float4 v = ldr4<TA>(rY, yo | yflag[j], (unsigned)(32 * j * HoWo4));
            areg[4 * j + 0] = v.x; areg[4 * j + 1] = v.y; areg[4 * j + 2] = v.z; areg[4 * j + 3] = v.w;
        } else {
            areg[j] = ldr<TA>(rY, yo | yflag[j], (unsigned)(8 * j * HoWo4));
        }
    };
    auto load_b = [&](int i) {
        constexpr int PER = BR / NT;   // K-columns (i) per tap
        breg[i] = ldr<TA>(rX, xo[i / PER], (unsigned)((c_b + 8 * (i % PER)) * HgWg4));
    };
    auto store_a = [&](int buf, int j) {
        if (VECA) {
            put4p<TA>(&As[buf][((tid >> 3) + 32 * j) * PT + pc], make_float4(areg[4 * j + 0], areg[4 * j + 1], areg[4 * j + 2], areg[4 * j + 3]));
        } else {
            put1<TA>(&As[buf][(rg + 8 * j) * PT + pl], areg[j]);
        }
    };
    auto store_b = [&](int buf, int i) { put1<TA>(&Gs[buf][(rg + 8 * i) * PT + pl], breg[i]); };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float av[2][MI][4], bv[2][NJ][4];   // operand sets of two consecutive MFMA groups
    auto read_op = [&](int buf, int q, int k, int set) {   // k-th operand read of group q: A rows first, then B
        if (k < MI) {
            const float4 t = *reinterpret_cast<const float4*>(&As[buf][(wm * WMT + k * 32 + lo) * PT + (2 * q + hi) * 4]);
            av[set][k][0] = t.x; av[set][k][1] = t.y; av[set][k][2] = t.z; av[set][k][3] = t.w;
        } else {
            const int j = k - MI;
            const float4 t = *reinterpret_cast<const float4*>(&Gs[buf][(wn * WNT + j * 32 + lo) * PT + (2 * q + hi) * 4]);
            bv[set][j][0] = t.x; bv[set][j][1] = t.y; bv[set][j][2] = t.z; bv[set][j][3] = t.w;
        }
    };
    auto mfma_one = [&](int gidx, int set) {
        const int jj = gidx / (MI * NJ), i = (gidx / NJ) % MI, j = gidx % NJ;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][i][jj], bv[set][j][jj], acc[i][j], 0, 0, 0);
    };

    if (nst > 0) {
        zero_tile<TA>(&As[0][0], 2 * BM * PT);
        zero_tile<TA>(&Gs[0][0], 2 * BN * PT);
        refill(0);
        refill(8);
        __syncthreads();
        read_offsets(0);
        combine_offsets();
#pragma unroll
        for (int j = 0; j < NA; ++j) load_a(j);
#pragma unroll
        for (int i = 0; i < BR; ++i) load_b(i);
        read_offsets(1);
        combine_offsets();
#pragma unroll
        for (int j = 0; j < NA; ++j) store_a(0, j);
#pragma unroll
        for (int i = 0; i < BR; ++i) store_b(0, i);
#pragma unroll
        for (int j = 0; j < NA; ++j) load_a(j);
#pragma unroll
        for (int i = 0; i < BR; ++i) load_b(i);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MI + NJ; ++k) read_op(0, 0, k, 0);

        constexpr int NG = MI * NJ * 4;            // matrix instructions per group
        constexpr int NOP = MI + NJ;               // operand reads per group
        constexpr int NWR = NA + BR, NLD = NA + BR;
        constexpr int NW0 = NWR / 2, NL0 = NLD / 2;
        auto stage = [&](int t, auto buf_tag) {
            constexpr int buf = decltype(buf_tag)::value;
            // group q computes with operand set q & 1 and issues: the reads of group q+1's operands, plus its share of
            // LDS writes (groups 0, 1) / global loads (groups 2, 3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n_extra = q == 0 ? NW0 + 1 : (q == 1 ? NWR - NW0 : (q == 2 ? NL0 + 1 : NLD - NL0));
                const int n_items = NOP + n_extra;
                if (q == 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int gI = 0; gI < NG; ++gI) {
                    mfma_one(gI, q & 1);
#pragma unroll
                    for (int k = 0; k < n_items; ++k) {
                        if (k * NG / n_items != gI) continue;
                        if (k < NOP) {
                            if (q < 3) read_op(buf, q + 1, k, (q + 1) & 1);
                            else read_op(buf ^ 1, 0, k, 0);
                        } else {
                            const int e = k - NOP;
                            if (q == 0) {
                                if (e == 0) read_offsets(t + 2);
                                else if (e - 1 < NA) store_a(buf ^ 1, e - 1);
                                else store_b(buf ^ 1, e - 1 - NA);
                            } else if (q == 1) {
                                const int w = NW0 + e;
                                if (w < NA) store_a(buf ^ 1, w);
                                else store_b(buf ^ 1, w - NA);
                            } else if (q == 2) {
                                if (e == 0) combine_offsets();
                                else if (e - 1 < NA) load_a(e - 1);
                                else load_b(e - 1 - NA);
                            } else {
                                const int l = NL0 + e;
                                if (l < NA) load_a(l);
                                else load_b(l - NA);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        for (int t = 0; t < nst; t += 2) {
            if ((t & 7) == 0 && t > 0) refill(t + 8);
            stage(t, std::integral_constant<int, 0>{});
            if (t + 1 < nst) stage(t + 1, std::integral_constant<int, 1>{});
        }
    }
    // partial tile store: row = m, column = k (lane) -> coalesced
    float* Wp = a.Wp + (size_t)split * a.M * a.Kp;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int kcol = kb + wn * WNT + j * 32 + lo;
        if (kcol >= a.Kp) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WMT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < a.M) Wp[(size_t)m * a.Kp + kcol] = acc[i][j][r];
            }
        }
    }
}

// dw[k][c][r][s] = sum_split Wp[split][k][tap*Cgp + c].  One workgroup = 64 consecutive partial-sum columns; its 4 waves
// take the splits round-robin (4 loads in flight per thread) and are combined in a fixed order: the sequential sum over up to
// 512 splits of the first version was pure load latency (23 us average, 38 us on the few-tile layers).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ Wp, float* __restrict__ dw, int splits, int K,
                                                           int C, int Cgp, int RS, int accumulate) {
    __shared__ float red[3][64];
    const int Kp = RS * Cgp;
    const int total = K * Kp;
    const int wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    float acc = 0.f;
    if (i < total) {
        const float* p = Wp + i;
        int sp = wave;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (; sp + 12 < splits; sp += 16) {
            a0 += p[(size_t)sp * total];
            a1 += p[(size_t)(sp + 4) * total];
            a2 += p[(size_t)(sp + 8) * total];
            a3 += p[(size_t)(sp + 12) * total];
        }
        for (; sp < splits; sp += 4) a0 += p[(size_t)sp * total];
        acc = (a0 + a1) + (a2 + a3);
    }
    if (wave > 0) red[wave - 1][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wave > 0 || i >= total) return;
    acc = (acc + red[0][threadIdx.x]) + (red[1][threadIdx.x] + red[2][threadIdx.x]);
    const int k = i / Kp;
    const int j = i - k * Kp;
    const int tap = j / Cgp, c = j - tap * Cgp;
    if (c >= C) return;
    float* o = dw + ((size_t)k * C + c) * RS + tap;
    *o = accumulate ? *o + acc : acc;   // accumulate: dw is the parameter's .grad buffer (fused "grad +=")
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static inline int round4(int v) { return (v + 3) & ~3; }
static inline size_t esz(const pcgan_conv_desc* d) { return d->dtype == PCGAN_BF16 ? 2 : 4; }   // bytes per activation element

// launch KERNEL<template arguments..., TA> with TA = the activation storage type `dt` (256 threads, stream `st`)
#define LAUNCH_TA(dt, KERNEL, GRID, ARG, ...)                                                                   \
    do {                                                                                                        \
        if ((dt) == PCGAN_BF16) hipLaunchKernelGGL((KERNEL<__VA_ARGS__, bf16>), GRID, dim3(256), 0, st, ARG);    \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__, float>), GRID, dim3(256), 0, st, ARG);                      \
    } while (0)

static int launch_splitk_reduce(int dtype, hipStream_t st, const float* part, void* y, const float* bias, int ks, size_t out_elems, int M,
                                int HW, int act, float slope) {
    size_t b = (out_elems / 4 + 255) / 256;
    b = b > 4096 ? 4096 : (b < 1 ? 1 : b);
    if (dtype == PCGAN_BF16)
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, dim3((unsigned)b), dim3(256), 0, st, part, (bf16*)y, bias, ks, out_elems, M, HW, act, slope);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((unsigned)b), dim3(256), 0, st, part, (float*)y, bias, ks, out_elems, M, HW, act, slope);
    PCGAN_LAUNCH_CHECK();
    return 0;
}

static int check_desc(const pcgan_conv_desc* d) {
    PCGAN_CHECK(d != nullptr, "conv: null descriptor");
    PCGAN_CHECK(d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0 && d->R > 0 && d->S > 0,
                "conv: non-positive dimension");
    PCGAN_CHECK(ilog2_exact(d->stride) >= 0 && d->stride <= 4, "conv: stride %d unsupported (1,2,4)", d->stride);
    PCGAN_CHECK(d->pad_mode == 0 || d->pad_mode == 1, "conv: pad_mode %d", d->pad_mode);
    PCGAN_CHECK(d->dtype == PCGAN_F32 || d->dtype == PCGAN_BF16, "conv: dtype %d (PCGAN_F32 / PCGAN_BF16)", d->dtype);
    const int P = (d->H + 2 * d->pad - d->R) / d->stride + 1, Q = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    PCGAN_CHECK(P == d->P && Q == d->Q, "conv: output dims %dx%d do not match geometry %dx%d", d->P, d->Q, P, Q);
    if (d->pad_mode == 1)
        PCGAN_CHECK(d->pad < d->H && d->pad < d->W, "conv: reflection pad %d >= input size", d->pad);
    // range-checked buffer loads address tensors with 32-bit byte offsets; the out-of-range marker is 2 GiB
    const size_t xin = (size_t)d->N * d->C * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * 4;
    const size_t yout = (size_t)d->N * d->K * d->P * d->Q * 4;
    PCGAN_CHECK(xin < (1ull << 31) && yout < (1ull << 31),
                "conv: tensor of %zu bytes exceeds the 2 GiB addressing limit of one launch (split the batch)",
                xin > yout ? xin : yout);
    PCGAN_CHECK((size_t)d->K * round4(d->C) * d->R * d->S * 4 < (1ull << 31) &&
                    (size_t)d->C * round4(d->K) * d->R * d->S * 4 < (1ull << 31), "conv: weight tensor too large");
    PCGAN_CHECK(d->R * d->S <= 512, "conv: filter too large");
    return 0;
}

// tile choice: the largest tile that still gives the 256 CUs >= ~1.5 workgroups each; problems with few
// pixels but a long K loop (the encoder's 7x7 / 14x14 stages, the PatchGAN's 16x16 stage) keep the big tile
// and are cut along K instead (split-K, partial sums reduced by splitk_reduce_kernel)
// chunked K order / igemm2_kernel: gathered channel count a multiple of 16, taps fit the LDS tables, MFMA path
static inline bool chunked_k(int Cg, int M, int R, int S) { return (Cg % 16) == 0 && M > 4 && R * S <= NTAP_FWD; }
// 4-channel stages (igemm2_kernel<.., 4>): 3-/4-channel gathered tensor, MFMA path, taps fit the table; weights stay in
// the generic (tap, channel) order
static inline bool cg4_k(int Cg, int M, int R, int S) { return round4(Cg) == 4 && M > 4 && R * S <= NTAP_CG4; }
static inline long tile_blocks(int M, int ptot_max, int nphase, int mm, int pp) {
    return (long)((M + mm - 1) / mm) * ((ptot_max + pp - 1) / pp) * nphase;
}
static inline long split_below() { return 192; }     // tiles (BP = 128) below which a long-K layer is cut along K (profiles/r01_tile_sweep.txt)
static inline bool may_split(int M, int ptot_max, int nphase) {
    if (option(OPT_HGEMM_KS) > 1) return M > 4;      // forced K split (measurement): every layer gets the room
    const int m = M > 64 ? 128 : (M > 32 ? 64 : 32);
    return M > 4 && tile_blocks(M, ptot_max, nphase, m, 128) < split_below();
}
static void choose_tile(int M, int ptot_max, int nphase, int nst, bool allow_split, int* bm, int* bp, int* ks) {
    int m = M > 64 ? 128 : (M > 32 ? 64 : 32);
    int p = 128;
    *ks = 1;
    // options "hgemm_tile" / "hgemm_ks" (measurement, scripts/sweep_hgemm.py): a forced tile and K split for layers with > 32 rows
    const int ft = option(OPT_HGEMM_TILE), fk = option(OPT_HGEMM_KS);
    if ((ft || fk) && M > 32) {
        if (ft) {
            m = ft / 1000;
            p = ft % 1000;
        } else {      // the heuristic's tile for an unsplit launch
            if (tile_blocks(M, ptot_max, nphase, m, p) < 384) p = 64;
            if (m == 128 && tile_blocks(M, ptot_max, nphase, m, p) < 384) m = 64;
        }
        if (M <= 64 && m == 128) m = 64;
        int k = fk > 0 ? fk : 1;
        if (!allow_split) k = 1;
        if (k > nst / 4) k = nst / 4 > 0 ? nst / 4 : 1;
        *bm = m;
        *bp = p;
        *ks = k;
        return;
    }
    if (allow_split && may_split(M, ptot_max, nphase) && nst >= 32) {
        const long b = tile_blocks(M, ptot_max, nphase, m, 128);
        int k = (int)(512 / b);
        if (k > 8) k = 8;
        if (k > nst / 8) k = nst / 8;
        if (k >= 2) {
            *bm = m;
            *bp = 128;
            *ks = k;
            return;
        }
    }
    if (m >= 64 && tile_blocks(M, ptot_max, nphase, m, p) < 384) p = 64;
    if (m == 128 && tile_blocks(M, ptot_max, nphase, m, p) < 384) m = 64;
    *bm = m;
    *bp = p;
}

template <int MODE>
static int launch_igemm(IgemmArgs& a, hipStream_t st, float* part_ws = nullptr, size_t part_bytes = 0) {
    int pmax = 0;
    for (int i = 0; i < a.nphase; ++i) pmax = a.ph[i].Ptot > pmax ? a.ph[i].Ptot : pmax;
    if (pmax <= 0 || a.nphase <= 0) return 0;
    if (a.M <= 4) {  // vector-ALU path; ph[].A already holds the transposed [Kp][4] weights
        PCGAN_CHECK(a.x_bytes < SM_INV, "small-M conv: gathered tensor must be < 1 GiB");
        for (int i = 0; i < a.nphase; ++i)
            PCGAN_CHECK(a.ph[i].nR <= 12 && a.ph[i].nS <= 12, "small-M conv: more than 12 taps per axis");
        {   // strip kernel: unit pixel stride along the column, columns long enough for 8-pixel strips, <= 7 row taps
            int maxR = 0, minH = 1 << 30, maxstrips = 0;
            for (int i = 0; i < a.nphase; ++i) {
                maxR = a.ph[i].nR > maxR ? a.ph[i].nR : maxR;
                minH = a.ph[i].Hs < minH ? a.ph[i].Hs : minH;
                const int ns = a.N * ((a.ph[i].Hs + 7) / 8) * a.ph[i].Ws;
                maxstrips = ns > maxstrips ? ns : maxstrips;
            }
            constexpr int SMODE = MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE;
            const bool unit = SMODE == MODE_BWD || (a.sl == 0 && a.ostep == 1);
            if (unit && maxR >= 3 && maxR <= 7 && minH >= 8) {   // (1-2 row taps: nothing to reuse)
                // few strips but many channels (the last PatchGAN conv, 512 -> 1 on 14x14): cut the channels over blockIdx.z
                const int wgs = ((maxstrips + 63) / 64) * a.nphase;
                int ks = 1;
                const size_t out_elems = (size_t)a.N * a.M * a.Yh * a.Yw;
                if (part_ws != nullptr && a.nphase == 1 && wgs < 128 && a.Cg >= 64) {
                    ks = 256 / wgs;
                    if (ks > 8) ks = 8;
                    if (ks > a.Cg / 16) ks = a.Cg / 16;
                    if ((size_t)ks * out_elems * 4 > part_bytes) ks = 1;
                }
                a.ksplit = ks;
                a.Ypart = part_ws;
                const dim3 gs((unsigned)((maxstrips + 63) / 64), (unsigned)a.nphase, (unsigned)ks);
#define LS(NRV) do { if (a.M <= 3) LAUNCH_TA(a.dtype, smallm_strip_kernel, gs, a, SMODE, NRV, 3); \
                     else LAUNCH_TA(a.dtype, smallm_strip_kernel, gs, a, SMODE, NRV, 4); } while (0)
                if (maxR <= 4) LS(4); else LS(7);
#undef LS
                PCGAN_LAUNCH_CHECK();
                if (ks > 1 && launch_splitk_reduce(a.dtype, st, part_ws, a.Y, a.bias, ks, out_elems, a.M, a.Yh * a.Yw, a.act, a.slope)) return 2;
                return 0;
            }
        }
        int maxtaps = 0;
        for (int i = 0; i < a.nphase; ++i) maxtaps = a.ph[i].nR * a.ph[i].nS > maxtaps ? a.ph[i].nR * a.ph[i].nS : maxtaps;
        if (maxtaps <= 9) {
            const dim3 g1((unsigned)((pmax + 255) / 256), (unsigned)a.nphase);
            LAUNCH_TA(a.dtype, smallm_conv_fewtaps_kernel, g1, a, (MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE));
            PCGAN_LAUNCH_CHECK();
            return 0;
        }
        const dim3 grid((unsigned)((pmax + 63) / 64), (unsigned)a.nphase);
        LAUNCH_TA(a.dtype, smallm_conv_kernel, grid, a, (MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE));
        PCGAN_LAUNCH_CHECK();
        return 0;
    }
    int bm, bp, ks;
    int nst_min = 1 << 30;
    for (int i = 0; i < a.nphase; ++i) nst_min = (a.ph[i].Kp + 15) / 16 < nst_min ? (a.ph[i].Kp + 15) / 16 : nst_min;
    const size_t out_elems = (size_t)a.N * a.M * a.Yh * a.Yw;
    choose_tile(a.M, pmax, a.nphase, nst_min, part_ws != nullptr, &bm, &bp, &ks);
    if (ks > 1 && (size_t)ks * out_elems * 4 > part_bytes) ks = 1;
    a.ksplit = ks;
    a.Ypart = part_ws;
    const dim3 grid((unsigned)(((a.M + bm - 1) / bm) * ((pmax + bp - 1) / bp)), (unsigned)a.nphase, (unsigned)ks);
    a.tstart[0] = 0;
    for (int i = 0; i < a.nphase; ++i) a.tstart[i + 1] = a.tstart[i] + (a.ph[i].Ptot + bp - 1) / bp;
    const dim3 grid2((unsigned)(((a.M + bm - 1) / bm) * a.tstart[a.nphase]), 1u, (unsigned)ks);
    const bool cg16 = a.chunked == 1, cg4 = a.chunked == 2;
    PCGAN_CHECK(cg16 || MODE != MODE_BWD_REFLECT, "igemm: fused reflect data-gradient needs K %% 16 == 0");
    if (cg4) {
        PCGAN_CHECK(a.Cgp == 4, "igemm: 4-channel stages need a 3-/4-channel tensor");
        for (int i = 0; i < a.nphase; ++i)
            PCGAN_CHECK(a.ph[i].nR * a.ph[i].nS <= NTAP_CG4, "igemm: 4-channel stages: more than %d taps", NTAP_CG4);
    }
    if (cg16) {
        PCGAN_CHECK((a.Cg % 16) == 0, "igemm: chunked K order needs a multiple of 16 channels");
        for (int i = 0; i < a.nphase; ++i)
            PCGAN_CHECK(a.ph[i].nR * a.ph[i].nS <= (MODE == MODE_BWD_REFLECT ? NTAP_MIR : NTAP_FWD) && (a.ph[i].Kp % 16) == 0,
                        "igemm: chunked K order: bad phase");
    }
    // bf16 tensors take the one-product bf16 MFMA form of the kernel whenever the shape allows (option "hgemm_bf16" = 0: the fp32 MFMA kernels)
    const bool half = a.dtype == PCGAN_BF16 && option(OPT_HGEMM_BF16) != 0;
    if ((half || (a.hsplit && a.dtype == PCGAN_F32)) && cg16 && MODE != MODE_BWD_REFLECT && bm >= 64) {
        // fp16 two-piece form (fp32 tensors) / bf16 form (bf16 tensors) of the same launch: same tiles, phases, K splits
        constexpr int HMODE = MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE;
#define LH(BMV, BPV) do { if (half) hipLaunchKernelGGL((hgemm_kernel<HMODE, BMV, BPV, bf16>), grid2, dim3(256), 0, st, a); \
                          else hipLaunchKernelGGL((hgemm_kernel<HMODE, BMV, BPV, float>), grid2, dim3(256), 0, st, a); } while (0)
        if (bm == 128 && bp == 128) LH(128, 128);
        else if (bm == 128) LH(128, 64);
        else if (bp == 128) LH(64, 128);
        else LH(64, 64);
#undef LH
        PCGAN_LAUNCH_CHECK();
        if (ks > 1 && launch_splitk_reduce(a.dtype, st, part_ws, a.Y, a.bias, ks, out_elems, a.M, a.Yh * a.Yw, a.act, a.slope)) return 2;
        return 0;
    }
#define LI(BMV, BPV)                                                                                   \
    do {                                                                                               \
        if (cg16) LAUNCH_TA(a.dtype, igemm2_kernel, grid2, a, MODE, BMV, BPV, 16);                      \
        else if (cg4) LAUNCH_TA(a.dtype, igemm2_kernel, grid2, a, (MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE), BMV, BPV, 4); \
        else LAUNCH_TA(a.dtype, igemm_kernel, grid, a, (MODE == MODE_BWD_REFLECT ? MODE_BWD : MODE), BMV, BPV); \
    } while (0)
    if (bm == 128 && bp == 128) LI(128, 128);
    else if (bm == 128) LI(128, 64);
    else if (bm == 64 && bp == 128) LI(64, 128);
    else if (bm == 64) LI(64, 64);
    else LI(32, 128);
#undef LI
    PCGAN_LAUNCH_CHECK();
    if (ks > 1 && launch_splitk_reduce(a.dtype, st, part_ws, a.Y, a.bias, ks, out_elems, a.M, a.Yh * a.Yw, a.act, a.slope)) return 2;
    return 0;
}

static inline bool smallm_wgrad(const pcgan_conv_desc* d) { return d->K <= 4 && (round4(d->C) % 16) == 0; }

// strip weight-gradient kernel: <= 3 output channels, stride 1, <= 7x7 taps, columns long enough for 8-pixel strips
static inline bool smallm_wgrad_strip(const pcgan_conv_desc* d) {
    return d->K <= 3 && d->stride == 1 && d->R <= 7 && d->S <= 7 && d->R >= 3 && d->P >= 16 && d->C >= 16;
}

static int wgrad_splits(const pcgan_conv_desc* d, int* chunks_per_split) {
    const int Cgp = round4(d->C);
    if (smallm_wgrad_strip(d)) {   // one workgroup per input channel and strip range; ~2048 workgroups, >= 4 strips per thread
        const int nstrips = d->N * ((d->P + 7) / 8) * d->Q;
        int splits = 1024 / d->C;
        if (splits > nstrips / 1024) splits = nstrips / 1024;
        if (splits < 1) splits = 1;
        const int sps = (nstrips + splits - 1) / splits;
        *chunks_per_split = sps;
        return (nstrips + sps - 1) / sps;
    }
    const int Kp = d->R * d->S * Cgp;
    if (smallm_wgrad(d)) {  // one workgroup per 16 K-columns and pixel split; aim at ~2048 workgroups
        const int chunks = (d->N * d->P * d->Q + 31) / 32;
        int splits = 2048 / (Kp / 16);
        if (splits > chunks / 64) splits = chunks / 64;  // >= 8 pixels per thread
        if (splits < 1) splits = 1;
        int cps = (chunks + splits - 1) / splits;
        *chunks_per_split = cps;
        return (chunks + cps - 1) / cps;
    }
    const int bm = d->K > 64 ? 128 : (d->K > 32 ? 64 : 32);
    const int tiles = ((d->K + bm - 1) / bm) * ((Kp + 127) / 128);
    const int Ptot = d->N * d->P * d->Q;
    const int chunks = (Ptot + 31) / 32;
    // Two workgroups fit on a CU (74-80 KB of LDS each): `slots` run at once.  The workgroup count tiles x splits is
    // kept just BELOW a whole number of rounds of slots -- a few workgroups over and the kernel waits for a nearly empty
    // extra round.  One round if it fills >= 90 % of the slots (fewest partial sums to write and reduce), else the
    // round count (<= 4) with the best fill.
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        slots = 2 * cus;
    }
    int splits = 1;
    {
        double best = -1.0;
        for (int r = 1; r <= 4; ++r) {
            const int sp = (r * slots) / tiles;
            if (sp < 1) continue;
            const double fill = (double)sp * tiles / ((double)r * slots);
            if (fill > best + 1e-9) {
                best = fill;
                splits = sp;
            }
            if (fill >= 0.9) break;
        }
    }
    if (splits > chunks / 8) splits = chunks / 8;  // at least 8 stages of work per block
    if (splits < 1) splits = 1;
    if (splits > 512) splits = 512;
    int cps = (chunks + splits - 1) / splits;
    splits = (chunks + cps - 1) / cps;
    *chunks_per_split = cps;
    return splits;
}

}  // namespace pcgan

using namespace pcgan;

// room for split-K partial sums (upper bound: 8 splits), 0 when the problem never splits
static size_t fwd_part_bytes(const pcgan_conv_desc* d) {
    const int ptot = d->N * d->P * d->Q;
    if (d->K <= 4) return (ptot <= 65536 && d->C >= 64) ? align_up((size_t)8 * d->K * ptot * 4, 256) : 0;   // strip kernel, channel split
    return may_split(d->K, ptot, 1) ? align_up((size_t)8 * d->N * d->K * d->P * d->Q * 4, 256) : 0;
}
static size_t bwd_part_bytes(const pcgan_conv_desc* d) {
    const int s = d->stride;
    const int pmax = d->N * ((d->H + s - 1) / s) * ((d->W + s - 1) / s);
    return may_split(d->C, pmax, s * s) ? align_up((size_t)8 * d->N * d->C * d->H * d->W * 4, 256) : 0;
}
static size_t fwd_base_bytes(const pcgan_conv_desc* d) {
    // A matrix (+ small-M path: RS*C*16 bytes for the [k][4] transposed weights and S*C*128 for the strip layout)
    return align_up((size_t)d->K * d->R * d->S * round4(d->C) * 4 + (size_t)d->R * d->S * round4(d->C) * 16 +
                    (d->K <= 4 ? (size_t)d->S * d->C * 128 : 0), 256);
}
// fused reflect data gradient (mirror images gathered on the unpadded grid) and its row-folded form (pad 1, 3 rows)
static bool bwd_fused_reflect(const pcgan_conv_desc* d) {
    return d->pad_mode == 1 && chunked_k(d->K, d->C, d->R, d->S) && d->R * d->S <= NTAP_MIR && d->H >= 2 * d->pad + 2 &&
           d->W >= 2 * d->pad + 2;
}
static bool bwd_rowfold(const pcgan_conv_desc* d) { return bwd_fused_reflect(d) && d->pad == 1 && d->R == 3 && d->stride == 1; }
static size_t bwd_base_bytes(const pcgan_conv_desc* d) {
    const size_t a = (size_t)d->C * d->R * d->S * round4(d->K) * 4;
    // small-M path: [k][4] transposed copies + strip layout (sum over the stride phases of nS = stride * S columns)
    return align_up((bwd_rowfold(d) ? 3 * a : a) + (size_t)d->R * d->S * round4(d->K) * 16 +
                    (d->C <= 4 ? (size_t)d->stride * d->S * d->K * 128 : 0), 256);
}

extern "C" size_t pcgan_conv2d_workspace_bytes(const pcgan_conv_desc* d, int pass) {
    if (!d) return 0;
    const size_t RS = (size_t)d->R * d->S;
    if (pass == PCGAN_PASS_FWD) return fwd_base_bytes(d) + fwd_part_bytes(d);
    if (pass == PCGAN_PASS_BWD_DATA) {
        size_t b = bwd_base_bytes(d) + bwd_part_bytes(d);
        if (d->pad_mode == 1)
            b += align_up((size_t)d->N * d->C * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * 4, 256);
        return b;
    }
    int cps;
    const int splits = wgrad_splits(d, &cps);
    return align_up((size_t)splits * d->K * RS * round4(d->C) * 4, 256);
}

// forward weight pack: A[K][Kp] (+ its [Kp][4] transpose behind it for the small-M path)
static int pack_fwd(const pcgan_conv_desc* d, const float* w, float* A, hipStream_t st) {
    const int Cgp = round4(d->C), RS = d->R * d->S;
    const size_t total = (size_t)d->K * RS * Cgp;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(repack_fwd_kernel, dim3(blocks), dim3(256), 0, st, w, A, d->K, d->C, Cgp, RS,
                       (int)chunked_k(d->C, d->K, d->R, d->S));
    PCGAN_LAUNCH_CHECK();
    if (d->K <= 4) {
        hipLaunchKernelGGL(transpose4_kernel, dim3((RS * Cgp * 4 + 255) / 256), dim3(256), 0, st, (const float*)A,
                           A + total, d->K, RS * Cgp);
        PCGAN_LAUNCH_CHECK();
        hipLaunchKernelGGL(pack_strip_kernel, dim3((d->C * d->S * 32 + 255) / 256), dim3(256), 0, st, (const float*)A,
                           A + total + (size_t)RS * Cgp * 4, d->K, d->C, Cgp, d->R, d->S);
        PCGAN_LAUNCH_CHECK();
    }
    return 0;
}

// operand maxima of the fp16 two-piece form (null: the fp32 MFMA kernels)
struct HsplitOpt {
    const float* x_amax;
    int n_amax;
    const float* w_amax;
};

static int conv2d_fwd_impl(const pcgan_conv_desc* d, const void* x, const float* w, const float* packed,
                           const float* bias, void* y, int act, float slope, void* ws, size_t ws_bytes,
                           pcgan_stream_t s, const HsplitOpt* hs = nullptr) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(x && (w || packed) && y, "conv2d_fwd: null pointer");
    PCGAN_CHECK(ws && ws_bytes >= pcgan_conv2d_workspace_bytes(d, PCGAN_PASS_FWD),
                "conv2d_fwd: workspace too small (%zu)", ws_bytes);
    hipStream_t st = (hipStream_t)s;
    const int Cgp = round4(d->C), RS = d->R * d->S;
    float* A = packed ? const_cast<float*>(packed) : (float*)ws;
    if (!packed && pack_fwd(d, w, A, st)) return 2;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.X = x; a.Y = y; a.bias = bias; a.dtype = d->dtype;
    a.M = d->K; a.N = d->N; a.Cg = d->C; a.Cgp = Cgp; a.Hg = d->H; a.Wg = d->W;
    a.Yh = d->P; a.Yw = d->Q;
    a.ostep = 1; a.sl = ilog2_exact(d->stride); a.pad = d->pad; a.tstep = 1;
    a.act = act; a.slope = slope;
    a.x_bytes = (unsigned)((size_t)d->N * d->C * d->H * d->W * esz(d));
    a.nphase = 1;
    a.chunked = chunked_k(d->C, d->K, d->R, d->S) ? 1 : (cg4_k(d->C, d->K, d->R, d->S) ? 2 : 0);
    if (hs) {
        a.hsplit = 1; a.x_amax = hs->x_amax; a.x_namax = hs->n_amax; a.w_amax = hs->w_amax;
        a.ovf = pcgan::nonfinite_counter();
    }
    PhaseArgs& p = a.ph[0];
    p.A = A; p.Kp = RS * Cgp; p.Hs = d->P; p.Ws = d->Q; p.fy = 0; p.fx = 0;
    if (d->K <= 4) {  // small-M path reads the weights as [k][4] / [c][ri][8][4]
        p.A = A + (size_t)d->K * RS * Cgp;
        p.As = p.A + (size_t)RS * Cgp * 4;
    }
    p.r0 = 0; p.s0 = 0; p.nR = d->R; p.nS = d->S; p.Ptot = d->N * d->P * d->Q;
    float* part = fwd_part_bytes(d) ? (float*)((char*)ws + fwd_base_bytes(d)) : nullptr;
    return d->pad_mode == 1 ? launch_igemm<MODE_FWD_REFLECT>(a, st, part, fwd_part_bytes(d))
                            : launch_igemm<MODE_FWD_ZERO>(a, st, part, fwd_part_bytes(d));
}

extern "C" int pcgan_conv2d_fwd(const pcgan_conv_desc* d, const void* x, const float* w, const float* bias,
                                void* y, int act, float slope, void* ws, size_t ws_bytes, pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(w, "conv2d_fwd: null weight pointer");
    return conv2d_fwd_impl(d, x, w, nullptr, bias, y, act, slope, ws, ws_bytes, s);
}
extern "C" int pcgan_conv2d_fwd_packed(const pcgan_conv_desc* d, const void* x, const float* packed,
                                       const float* bias, void* y, int act, float slope, void* ws, size_t ws_bytes,
                                       pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(packed, "conv2d_fwd_packed: null packed-weight pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_FWD), (hipStream_t)s);
    return conv2d_fwd_impl(d, x, nullptr, packed, bias, y, act, slope, ws, ws_bytes, s);
}

namespace pcgan {
// dx = the gradient of a reflection-padded tensor `padded` [NC][H + 2 pad][W + 2 pad] folded back onto the unpadded grid
int launch_reflect_fold(const void* padded, void* dx, int NC, int H, int W, int pad, int dtype, hipStream_t st) {
    PCGAN_CHECK(NC <= 65535, "conv2d_bwd_data: more than 65535 planes in the reflect fold");
    const int per_plane = H * ((W + 3) / 4);
    if (dtype == PCGAN_BF16)
        hipLaunchKernelGGL(reflect_fold_kernel<bf16>, dim3((per_plane + 255) / 256, NC), dim3(256), 0, st, (const bf16*)padded, (bf16*)dx, NC, H,
                           W, pad);
    else
        hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3((per_plane + 255) / 256, NC), dim3(256), 0, st, (const float*)padded, (float*)dx, NC,
                           H, W, pad);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
}  // namespace pcgan

static int conv2d_bwd_data_impl(const pcgan_conv_desc* d, const void* dy, const float* w, const float* packed,
                                const float* bias, void* dx, void* ws, size_t ws_bytes, pcgan_stream_t s, const HsplitOpt* hs = nullptr) {
    if (check_desc(d)) return 1;
    const bool pack_only = dx == nullptr;   // pcgan_conv2d_pack_weights: run the repack launches into `packed` only
    PCGAN_CHECK(pack_only ? (w && packed) : (dy && (w || packed)), "conv2d_bwd_data: null pointer");
    PCGAN_CHECK(pack_only || (ws && ws_bytes >= pcgan_conv2d_workspace_bytes(d, PCGAN_PASS_BWD_DATA)),
                "conv2d_bwd_data: workspace too small (%zu)", ws_bytes);
    PCGAN_CHECK(d->pad_mode == 0 || d->stride == 1, "conv2d_bwd_data: reflection padding needs stride 1");
    hipStream_t st = (hipStream_t)s;
    const int Kgp = round4(d->K), RS = d->R * d->S;
    float* Abase = packed ? const_cast<float*>(packed) : (float*)ws;
    const bool do_pack = pack_only || !packed;
    const size_t a_bytes = bwd_base_bytes(d);
    const bool smallm = d->C <= 4;
    size_t at_off = (size_t)d->C * RS * Kgp;  // transposed copies for the small-M path live behind the A's
    size_t as_off = at_off + (size_t)RS * Kgp * 4;   // ... and the strip layouts behind those
    // reflection: gather the mirror images directly (fused, needs K % 16 == 0 and H,W >= 2 pad + 2); otherwise
    // compute the gradient of the PADDED input (pad 0 on a larger grid) and fold it back
    const bool chunked = chunked_k(d->K, d->C, d->R, d->S);
    const bool fused = bwd_fused_reflect(d);
    const bool rowfold = bwd_rowfold(d);
    const bool reflect = d->pad_mode == 1 && !fused;
    const int H = reflect ? d->H + 2 * d->pad : d->H;
    const int W = reflect ? d->W + 2 * d->pad : d->W;
    const int pad = reflect ? 0 : d->pad;
    void* out = reflect ? (void*)((char*)ws + a_bytes) : dx;
    const int stv = d->stride;

    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.X = dy; a.Y = out; a.bias = bias; a.dtype = d->dtype;
    a.M = d->C; a.N = d->N; a.Cg = d->K; a.Cgp = Kgp; a.Hg = d->P; a.Wg = d->Q;
    a.Yh = H; a.Yw = W;
    a.ostep = stv; a.sl = ilog2_exact(stv); a.pad = pad; a.tstep = stv;
    a.act = PCGAN_ACT_NONE; a.slope = 0.f;
    a.chunked = chunked ? 1 : (cg4_k(d->K, d->C, d->R, d->S) ? 2 : 0);
    a.rowfold = rowfold;
    if (hs) {
        a.hsplit = 1; a.x_amax = hs->x_amax; a.x_namax = hs->n_amax; a.w_amax = hs->w_amax;
        a.ovf = pcgan::nonfinite_counter();
    }
    a.x_bytes = (unsigned)((size_t)d->N * d->K * d->P * d->Q * esz(d));

    if (rowfold) {
        // three row classes with their own weights: rows without a mirror image | row 1 | row H-2
        const size_t total = (size_t)d->C * RS * Kgp;
        PackBwdArgs pk;
        memset(&pk, 0, sizeof(pk));
        pk.w = w; pk.K = d->K; pk.C = d->C; pk.Kgp = Kgp; pk.R = d->R; pk.S = d->S; pk.tstep = 1; pk.chunked = 1; pk.n = 3;
        for (int f = 0; f < 3; ++f) {
            float* A = Abase + (size_t)f * total;
            pk.e[f].A = A; pk.e[f].r0 = 0; pk.e[f].s0 = 0; pk.e[f].nR = d->R; pk.e[f].nS = d->S; pk.e[f].fold = f;
            PhaseArgs& p = a.ph[a.nphase++];
            p.A = A; p.Kp = RS * Kgp; p.Ws = W; p.fx = 0; p.r0 = 0; p.s0 = 0; p.nR = d->R; p.nS = d->S;
            p.Hs = f == 0 ? H - 2 : 1;
            p.fy = f == 0 ? 0 : (f == 1 ? 1 : H - 2);
            p.ymap = f == 0 ? 1 : 0;
            p.Ptot = d->N * p.Hs * W;
        }
        if (do_pack) {
            const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
            hipLaunchKernelGGL(repack_bwd_multi_kernel, dim3(blocks, 3), dim3(256), 0, st, pk);
            PCGAN_LAUNCH_CHECK();
        }
        if (pack_only) return 0;
        return launch_igemm<MODE_BWD_REFLECT>(a, st, nullptr, 0);
    }

    // one phase per (iy % stride, ix % stride): only the taps that are structurally non-zero for it
    bool need_zero = false;
    size_t a_off = 0, max_total = 0;
    PackBwdArgs pk;
    memset(&pk, 0, sizeof(pk));
    pk.w = w; pk.K = d->K; pk.C = d->C; pk.Kgp = Kgp; pk.R = d->R; pk.S = d->S; pk.tstep = stv; pk.chunked = (int)chunked;
    for (int fy = 0; fy < stv; ++fy) {
        for (int fx = 0; fx < stv; ++fx) {
            const int r0 = (fy + pad) % stv, s0 = (fx + pad) % stv;
            const int nR = r0 < d->R ? (d->R - r0 + stv - 1) / stv : 0;
            const int nS = s0 < d->S ? (d->S - s0 + stv - 1) / stv : 0;
            const int Hs = fy < H ? (H - fy + stv - 1) / stv : 0;
            const int Ws = fx < W ? (W - fx + stv - 1) / stv : 0;
            if (Hs * Ws == 0) continue;
            if (nR * nS == 0) {  // e.g. 1x1 stride 2: these pixels receive nothing
                need_zero = true;
                continue;
            }
            float* A = Abase + a_off;
            const size_t total = (size_t)d->C * nR * nS * Kgp;
            a_off += total;
            if (!smallm) {   // packed together after the loop; the small-M layouts below are derived from A right away
                max_total = total > max_total ? total : max_total;
                PackBwdArgs::Entry& pe = pk.e[pk.n++];
                pe.A = A; pe.r0 = r0; pe.s0 = s0; pe.nR = nR; pe.nS = nS; pe.fold = 0;
            } else if (do_pack) {
                const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
                hipLaunchKernelGGL(repack_bwd_kernel, dim3(blocks), dim3(256), 0, st, w, A, d->K, d->C, Kgp, d->R,
                                   d->S, r0, s0, stv, nR, nS, (int)chunked, 0);
                PCGAN_LAUNCH_CHECK();
            }
            PhaseArgs& p = a.ph[a.nphase++];
            p.A = A; p.Kp = nR * nS * Kgp; p.Hs = Hs; p.Ws = Ws; p.fy = fy; p.fx = fx;
            if (smallm) {
                float* At = Abase + at_off;
                at_off += (size_t)nR * nS * Kgp * 4;
                float* As = Abase + as_off;
                as_off += (size_t)d->K * nS * 32;
                if (do_pack) {
                    hipLaunchKernelGGL(transpose4_kernel, dim3((nR * nS * Kgp * 4 + 255) / 256), dim3(256), 0, st,
                                       (const float*)A, At, d->C, nR * nS * Kgp);
                    PCGAN_LAUNCH_CHECK();
                    hipLaunchKernelGGL(pack_strip_kernel, dim3((d->K * nS * 32 + 255) / 256), dim3(256), 0, st,
                                       (const float*)A, As, d->C, d->K, Kgp, nR, nS);
                    PCGAN_LAUNCH_CHECK();
                }
                p.A = At;
                p.As = As;
            }
            p.r0 = r0; p.s0 = s0; p.nR = nR; p.nS = nS; p.Ptot = d->N * Hs * Ws;
        }
    }
    if (do_pack && pk.n > 0) {
        const int blocks = (int)((max_total + 255) / 256 > 2048 ? 2048 : (max_total + 255) / 256);
        hipLaunchKernelGGL(repack_bwd_multi_kernel, dim3(blocks, pk.n), dim3(256), 0, st, pk);
        PCGAN_LAUNCH_CHECK();
    }
    if (pack_only) return 0;
    if (need_zero) {
        // pixels that no phase writes would also miss the bias; never happens for the nets on the hot path
        PCGAN_CHECK(!bias, "conv2d_bwd_data: bias with uncovered phases is unsupported");
        hipError_t e = hipMemsetAsync(out, 0, (size_t)d->N * d->C * H * W * esz(d), st);
        PCGAN_CHECK(e == hipSuccess, "memset failed: %s", hipGetErrorString(e));
    }
    // split-K partials live behind the A matrices (not with the padded-grid fallback or uncovered phases)
    float* part = (bwd_part_bytes(d) && !reflect && !need_zero) ? (float*)((char*)ws + a_bytes) : nullptr;
    if (fused ? launch_igemm<MODE_BWD_REFLECT>(a, st, part, bwd_part_bytes(d)) : launch_igemm<MODE_BWD>(a, st, part, bwd_part_bytes(d)))
        return 2;
    if (reflect) return pcgan::launch_reflect_fold(out, dx, d->N * d->C, d->H, d->W, d->pad, d->dtype, st);
    return 0;
}

extern "C" int pcgan_conv2d_bwd_data(const pcgan_conv_desc* d, const void* dy, const float* w,
                                     const float* bias, void* dx, void* ws, size_t ws_bytes,
                                     pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(w && dx, "conv2d_bwd_data: null pointer");
    return conv2d_bwd_data_impl(d, dy, w, nullptr, bias, dx, ws, ws_bytes, s);
}
extern "C" int pcgan_conv2d_bwd_data_packed(const pcgan_conv_desc* d, const void* dy, const float* packed,
                                            const float* bias, void* dx, void* ws, size_t ws_bytes,
                                            pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(packed && dx, "conv2d_bwd_data_packed: null pointer");
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_DGRAD), (hipStream_t)s);
    return conv2d_bwd_data_impl(d, dy, nullptr, packed, bias, dx, ws, ws_bytes, s);
}

// ---- fp16 two-piece form of the packed calls (fp32 tensors, hgemm_kernel): same packed weights, workspace and results layout as
// pcgan_conv2d_fwd_packed / pcgan_conv2d_bwd_data_packed; x_amax[0 .. n_amax) partial maxima of |x| (pcgan_absmax or a producer's
// plane maxima) and w_amax[0 .. 64) partial maxima of |weight| (pcgan_absmax with 64 slots), both on the device
extern "C" int pcgan_conv2d_hgemm_supported(const pcgan_conv_desc* d, int pass) {
    if (!d || d->dtype != PCGAN_F32) return 0;
    if (pass == PCGAN_PASS_FWD) return chunked_k(d->C, d->K, d->R, d->S) && d->K > 32;
    if (pass == PCGAN_PASS_BWD_DATA) return d->pad_mode == 0 && chunked_k(d->K, d->C, d->R, d->S) && d->C > 32;
    return 0;
}
extern "C" int pcgan_conv2d_fwd_packed_hsplit(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_amax, const float* packed,
                                              const float* w_amax, const float* bias, void* y, int act, float slope, void* ws,
                                              size_t ws_bytes, pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(pcgan_conv2d_hgemm_supported(d, PCGAN_PASS_FWD), "conv2d_fwd_packed_hsplit: unsupported shape");
    PCGAN_CHECK(packed && x_amax && n_amax > 0 && w_amax, "conv2d_fwd_packed_hsplit: null pointer");
    const HsplitOpt hs = {x_amax, n_amax, w_amax};
    return conv2d_fwd_impl(d, x, nullptr, packed, bias, y, act, slope, ws, ws_bytes, s, &hs);
}
extern "C" int pcgan_conv2d_bwd_data_packed_hsplit(const pcgan_conv_desc* d, const void* dy, const float* dy_amax, int n_amax,
                                                   const float* packed, const float* w_amax, const float* bias, void* dx, void* ws,
                                                   size_t ws_bytes, pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(pcgan_conv2d_hgemm_supported(d, PCGAN_PASS_BWD_DATA), "conv2d_bwd_data_packed_hsplit: unsupported shape");
    PCGAN_CHECK(packed && dx && dy_amax && n_amax > 0 && w_amax, "conv2d_bwd_data_packed_hsplit: null pointer");
    const HsplitOpt hs = {dy_amax, n_amax, w_amax};
    return conv2d_bwd_data_impl(d, dy, nullptr, packed, bias, dx, ws, ws_bytes, s, &hs);
}

// in-place pre-split of one packed fp32 weight matrix A[M][Kp] for hgemm_kernel: every aligned group of 4 consecutive floats of row m
// (what one thread feeds to LDS per stage) becomes [4 fp16 high pieces][4 fp16 low pieces] of the values scaled by
// pow2_scale(rowmax[m]) -- one power of two per ROW, so a filter row far below the tensor's largest weight keeps its 22 bits
__global__ void __launch_bounds__(256) hgemm_presplit_kernel(float* __restrict__ A, int M, int Kp4, const float* __restrict__ rowmax) {
    const size_t n4 = (size_t)M * Kp4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float s = pow2_scale(rowmax[i / Kp4]);
        const float4 v = reinterpret_cast<const float4*>(A)[i];
        f16x4 h, l;
        _Float16 x, y;
        split2h(v.x * s, x, y); h[0] = x; l[0] = y;
        split2h(v.y * s, x, y); h[1] = x; l[1] = y;
        split2h(v.z * s, x, y); h[2] = x; l[2] = y;
        split2h(v.w * s, x, y); h[3] = x; l[3] = y;
        reinterpret_cast<f16x4*>(A)[2 * i] = h;
        reinterpret_cast<f16x4*>(A)[2 * i + 1] = l;
    }
}

extern "C" int pcgan_conv2d_hgemm_pack(const pcgan_conv_desc* d, int pass, const float* w, float* w_rowmax, float* packed,
                                       pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(pcgan_conv2d_hgemm_supported(d, pass), "conv2d_hgemm_pack: unsupported shape or pass");
    PCGAN_CHECK(w && w_rowmax && packed, "conv2d_hgemm_pack: null pointer");
    hipStream_t st = (hipStream_t)s;
    const bool fwd = pass == PCGAN_PASS_FWD;
    const int M = fwd ? d->K : d->C, RS = d->R * d->S;
    if (launch_weight_row_absmax(w, d->K, d->C, RS, fwd ? 0 : 1, w_rowmax, st)) return 2;
    if (pcgan_conv2d_pack_weights(d, pass, w, packed, s)) return 1;
    // the A matrices of the phases lie back to back at the start of the packed buffer (conv2d_fwd_impl / conv2d_bwd_data_impl): one
    // for the forward pass, one per (y mod stride, x mod stride) that owns taps for the data gradient
    struct Ph { size_t off; int kp; } ph[16];
    int nph = 0;
    if (fwd) {
        ph[nph++] = {0, RS * round4(d->C)};
    } else {
        const int stv = d->stride, Kgp = round4(d->K);
        size_t off = 0;
        for (int fy = 0; fy < stv; ++fy)
            for (int fx = 0; fx < stv; ++fx) {
                const int r0 = (fy + d->pad) % stv, s0 = (fx + d->pad) % stv;
                const int nR = r0 < d->R ? (d->R - r0 + stv - 1) / stv : 0, nS = s0 < d->S ? (d->S - s0 + stv - 1) / stv : 0;
                const int Hs = fy < d->H ? (d->H - fy + stv - 1) / stv : 0, Ws = fx < d->W ? (d->W - fx + stv - 1) / stv : 0;
                if (Hs * Ws == 0 || nR * nS == 0) continue;
                ph[nph++] = {off, nR * nS * Kgp};
                off += (size_t)d->C * nR * nS * Kgp;
            }
    }
    for (int i = 0; i < nph; ++i) {
        const size_t n4 = (size_t)M * (ph[i].kp / 4);
        const unsigned blocks = (unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
        hipLaunchKernelGGL(hgemm_presplit_kernel, dim3(blocks), dim3(256), 0, st, packed + ph[i].off, M, ph[i].kp / 4, (const float*)w_rowmax);
        PCGAN_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" size_t pcgan_conv2d_packed_bytes(const pcgan_conv_desc* d, int pass) {
    if (!d) return 0;
    if (pass == PCGAN_PASS_FWD) return fwd_base_bytes(d);
    if (pass == PCGAN_PASS_BWD_DATA) return bwd_base_bytes(d);
    return 0;
}
extern "C" int pcgan_conv2d_pack_weights(const pcgan_conv_desc* d, int pass, const float* w, float* packed,
                                         pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(w && packed, "conv2d_pack_weights: null pointer");
    if (pass == PCGAN_PASS_FWD) return pack_fwd(d, w, packed, (hipStream_t)s);
    PCGAN_CHECK(pass == PCGAN_PASS_BWD_DATA, "conv2d_pack_weights: pass %d has no packed weights", pass);
    return conv2d_bwd_data_impl(d, nullptr, w, packed, nullptr, nullptr, nullptr, 0, s);
}

extern "C" int pcgan_conv2d_bwd_weight(const pcgan_conv_desc* d, const void* x, const void* dy, float* dw,
                                       int accumulate, void* ws, size_t ws_bytes, pcgan_stream_t s) {
    if (check_desc(d)) return 1;
    PCGAN_CHECK(x && dy && dw, "conv2d_bwd_weight: null pointer");
    PCGAN_CHECK(ws && ws_bytes >= pcgan_conv2d_workspace_bytes(d, PCGAN_PASS_BWD_WEIGHT),
                "conv2d_bwd_weight: workspace too small (%zu)", ws_bytes);
    hipStream_t st = (hipStream_t)s;
    pcgan::TimerScope timer(pcgan::timer_kind_res(d, pcgan::TIMER_RES_WGRAD), st);
    const int Cgp = round4(d->C), RS = d->R * d->S;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dY = dy; a.X = x; a.Wp = (float*)ws; a.dtype = d->dtype;
    a.M = d->K; a.Kp = RS * Cgp; a.N = d->N; a.Cg = d->C; a.Cgp = Cgp;
    a.Hg = d->H; a.Wg = d->W; a.Ho = d->P; a.Wo = d->Q;
    a.sl = ilog2_exact(d->stride); a.pad = d->pad; a.S = d->S;
    a.magicS = (65536 + d->S - 1) / d->S;
    a.Ptot = d->N * d->P * d->Q;
    a.x_bytes = (unsigned)((size_t)d->N * d->C * d->H * d->W * esz(d));
    a.dy_bytes = (unsigned)((size_t)d->N * d->K * d->P * d->Q * esz(d));
    const int splits = wgrad_splits(d, &a.chunks_per_split);
    if (smallm_wgrad_strip(d)) {
        const dim3 sgrid((unsigned)d->C, (unsigned)splits, (unsigned)((d->R <= 4 && d->S <= 4) ? 1 : (d->S + 3) / 4));
#define LWS(MODE) do { if (d->R <= 4 && d->S <= 4) LAUNCH_TA(a.dtype, smallm_wgrad_strip_kernel, sgrid, a, MODE, 4); \
                       else LAUNCH_TA(a.dtype, smallm_wgrad_strip_kernel, sgrid, a, MODE, 7); } while (0)
        if (d->pad_mode == 1) LWS(MODE_FWD_REFLECT); else LWS(MODE_FWD_ZERO);
#undef LWS
        PCGAN_LAUNCH_CHECK();
        const size_t total = (size_t)d->K * RS * Cgp;
        const int blocks = (int)((total + 63) / 64);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, d->K, d->C,
                           Cgp, RS, accumulate);
        PCGAN_LAUNCH_CHECK();
        return 0;
    }
    if (smallm_wgrad(d)) {
        const dim3 sgrid((unsigned)(a.Kp / 16), (unsigned)splits);
        if (d->pad_mode == 1) LAUNCH_TA(a.dtype, smallm_wgrad_kernel, sgrid, a, MODE_FWD_REFLECT);
        else LAUNCH_TA(a.dtype, smallm_wgrad_kernel, sgrid, a, MODE_FWD_ZERO);
        PCGAN_LAUNCH_CHECK();
        const size_t total = (size_t)d->K * RS * Cgp;
        const int blocks = (int)((total + 63) / 64);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits, d->K, d->C,
                           Cgp, RS, accumulate);
        PCGAN_LAUNCH_CHECK();
        return 0;
    }
    const int bm = a.M > 64 ? 128 : (a.M > 32 ? 64 : 32);
    const dim3 grid((unsigned)(((a.M + bm - 1) / bm) * ((a.Kp + 127) / 128)), (unsigned)splits);
    const bool smallc = (Cgp % 8) != 0;
    const bool reflect = d->pad_mode == 1;
    const bool veca = ((d->P * d->Q) % 4) == 0;
    const int kmode = smallc ? 1 : ((Cgp % 128) == 0 ? 2 : 0);
    const bool w2 = (d->C % 64) == 0 && ((Cgp % 128) == 0 || Cgp == 64);
    if (w2) {
#define LW2(MODE, BMV, VA) do { if (Cgp == 64) LAUNCH_TA(a.dtype, wgrad2_kernel, grid, a, MODE, BMV, VA, 2); \
                                else LAUNCH_TA(a.dtype, wgrad2_kernel, grid, a, MODE, BMV, VA, 1); } while (0)
#define LW2_VA(MODE, BMV) do { if (veca) LW2(MODE, BMV, true); else LW2(MODE, BMV, false); } while (0)
#define LW2_BM(MODE) do { if (bm == 128) LW2_VA(MODE, 128); else if (bm == 64) LW2_VA(MODE, 64); else LW2_VA(MODE, 32); } while (0)
        if (reflect) LW2_BM(MODE_FWD_REFLECT); else LW2_BM(MODE_FWD_ZERO);
#undef LW2_BM
#undef LW2_VA
#undef LW2
    } else {
#define LW(MODE, BMV, KM, VA) LAUNCH_TA(a.dtype, wgrad_kernel, grid, a, MODE, BMV, KM, VA)
#define LW_VA(MODE, BMV, KM) do { if (veca) LW(MODE, BMV, KM, true); else LW(MODE, BMV, KM, false); } while (0)
#define LW_SC(MODE, BMV) do { if (kmode == 1) LW_VA(MODE, BMV, 1); else if (kmode == 2) LW_VA(MODE, BMV, 2); else LW_VA(MODE, BMV, 0); } while (0)
#define LW_BM(MODE) do { if (bm == 128) LW_SC(MODE, 128); else if (bm == 64) LW_SC(MODE, 64); else LW_SC(MODE, 32); } while (0)
    if (reflect) LW_BM(MODE_FWD_REFLECT); else LW_BM(MODE_FWD_ZERO);
#undef LW_BM
#undef LW_SC
#undef LW_VA
#undef LW
    }
    PCGAN_LAUNCH_CHECK();
    {
        const size_t total = (size_t)d->K * RS * Cgp;
        const int blocks = (int)((total + 63) / 64);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, dw, splits,
                           d->K, d->C, Cgp, RS, accumulate);
        PCGAN_LAUNCH_CHECK();
    }
    return 0;
}
