// Weight gradient of the residual-block convolution (3x3, stride 1, reflection padding 1; fp32 tensors, fp16 two-piece route) in the
// "row ring" form (round 4).  Replaces autograd's weight gradient of nn.Conv2d in ResnetBlock (models/networks.py:616-652).
//
// dW[k][c][r][s] = sum over (n, y, x) of dy[n][k][y][x] * xpad[n][c][y + r][x + s]: the reduction index is the PIXEL, 16 consecutive x of
// one image row per MFMA K step.  hsplit_wgrad_kernel (bf16x6_conv.hip) makes every (channel, tap) COLUMN gather, scale, split and store
// its own copy of x -- nine loads, nine splits and nine LDS stores per element, and its matrix pipe is busy 38 % of the time.  Here a
// workgroup's column tile is 32 input channels x ALL nine taps, and it walks DOWN a 16-pixel-wide strip of one image:
//   * a stage (n, x0, y) needs the padded rows y, y + 1, y + 2 of its 32 channels; the next stage needs y + 1 .. y + 3: ONE new row per
//     stage, kept in a ring of four row slots in LDS -- every element of x is loaded and split ONCE per strip (18 / 16 with the strip's
//     two border columns), not nine times;
//   * the three tap columns s = 0, 1, 2 are element shifts of a row, which would misalign the 16-byte record of 8 pixels a lane feeds the
//     MFMA; the row builder therefore writes THREE copies of the row, shifted by s (the shifted quads are assembled in registers from
//     the neighbouring lane's edge element: one shuffle each way) -- a tap is then a slot / copy index, every B fragment one aligned
//     ds_read_b128 of [khalf][channel][8 pixels], conflict-free;
//   * dy never touches LDS: wave w owns the output-channel rows w * 32 .. + 31 of the tile and a lane loads the 8 pixels of ITS row
//     straight from memory (two 16-byte loads), splits them in registers and has its A fragment.
// Per wave and stage: 27 MFMAs (9 taps x 3 piece products), 18 LDS reads, 8 + ~1 splits per lane, one barrier.  256 threads and
// 24.5 KB of LDS per workgroup, two workgroups per CU (one wave per SIMD each): one's epilogue / strip prologue runs under the other's
// MFMAs.  The strips are split over enough workgroups to fill the chip twice; partial sums [split][tap][k][c] (coalesced along c) are
// combined in a fixed order, scaled back and transposed into dW[k][c][r][s] by wgd_reduce_kernel (wgrad_direct.hip).
#include "common.h"

namespace pcgan {

typedef _Float16 rr_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rr_h4 __attribute__((ext_vector_type(4)));
typedef float rr_f16v __attribute__((ext_vector_type(16)));

struct RowRingArgs {
    const float* X;       // [N][C][H][W]
    const float* DY;      // [N][K][H][W]
    float* part;          // [splits][9][K][C]
    float* scales;        // [2]: the powers of two x and dy were scaled by (written by workgroup 0, read by the reduce kernel)
    const float* x_amax;  // partial maxima of |x| / |dy| (device)
    const float* dy_amax;
    int x_namax, dy_namax;
    int N, C, K, H, W;
    int nmt, ncb;         // row tiles of 128 output channels, column tiles of 32 input channels
    int strips, strips_per_split, nwg;
};

__global__ void __launch_bounds__(256, 2) rowring_wgrad_kernel(RowRingArgs a) {
    // [row slot][shift s][piece][k half][channel] records of 8 pixels, + 8 bytes per thread where the lanes without a quad of their own write
    // (branch-free row builder: a branch would cut the stage into basic blocks and the issue-order fences below work inside one)
    __shared__ __attribute__((aligned(16))) rr_h8 Xs[4 * 3 * 2 * 2 * 32 + 128];
    __shared__ float scratch[16];
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups go to the 8 XCDs round-robin: give each XCD a contiguous run of (split, tile) pairs (one split's tiles read the same rows)
    const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int tiles = a.nmt * a.ncb;
    const int split = wg / tiles, tile = wg - split * tiles;
    const int mt = tile % a.nmt, cb = tile / a.nmt;
    const int H = a.H, W = a.W;

    const float sx = pow2_scale(block_max(thread_max_of_partials(a.x_amax, a.x_namax, tid, 256), scratch));
    __syncthreads();
    const float sdy = pow2_scale(block_max(thread_max_of_partials(a.dy_amax, a.dy_namax, tid, 256), scratch));
    if (wg == 0 && tid == 0) {
        a.scales[0] = sx;
        a.scales[1] = sdy;
    }

    // row builder role of this thread: channel ch of the column tile; q = 1 .. 4 loads the quad x0 + 4 (q - 1) .. + 3, q = 0 the single
    // column left of the strip (mirrored at the image edge), q = 5 the single column right of it; q = 6, 7 load nothing
    const int ch = tid >> 3, q = tid & 7;
    char* const xs_bytes = reinterpret_cast<char*>(&Xs[0]);
    auto rec_off = [&](int slot, int s, int piece, int khalf, int c) -> unsigned {
        return (unsigned)((((((slot * 3 + s) * 2 + piece) * 2 + khalf) * 32) + c) * 16);
    };

    rr_f16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int wq = W >> 4;      // strips per image
    const int s_begin = split * a.strips_per_split;
    const int s_end = s_begin + a.strips_per_split < a.strips ? s_begin + a.strips_per_split : a.strips;
    const int HW = H * W;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((size_t)a.N * a.C * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.DY), 0, (int)((size_t)a.N * a.K * HW * 4), 0x00020000);
    typedef unsigned int rr_u4 __attribute__((ext_vector_type(4)));
    const unsigned vod = (unsigned)(((mt * 128 + wave * 32 + lo) * HW + 8 * hi) * 4);      // lane part of the dy offsets
    for (int strip = s_begin; strip < s_end; ++strip) {
        const int n = strip / wq, x0 = (strip - n * wq) << 4;
        // every thread loads ONE aligned quad per row (branch-free): q = 1 .. 4 their own, q = 0 / q >= 5 the quad that holds the single
        // column they contribute (the element is selected after the load)
        const int colq = q == 0 ? (x0 == 0 ? 1 : x0 - 1) : (q >= 5 ? (x0 + 16 >= W ? W - 2 : x0 + 16) : x0 + 4 * (q - 1));
        const int esel = colq & 3;
        const unsigned vox = (unsigned)((ch * HW + (colq & ~3)) * 4);
        const unsigned sox = (unsigned)((n * a.C + cb * 32) * HW * 4);
        const unsigned sod = (unsigned)((n * a.K * HW + x0) * 4);
        auto ldrow = [&](int py) -> rr_u4 {      // padded row py = unpadded row py - 1, mirrored at the image edge (py clamped to H + 1)
            int sy = (py > H + 1 ? H + 1 : py) - 1;
            sy = sy < 0 ? -sy : sy;
            sy = sy >= H ? 2 * (H - 1) - sy : sy;
            return __builtin_amdgcn_raw_buffer_load_b128(rX, vox, sox + (unsigned)(sy * W * 4), 0);
        };
        auto lddy = [&](int y, rr_u4& u0, rr_u4& u1) {
            const int yy = y < H ? y : H - 1;
            u0 = __builtin_amdgcn_raw_buffer_load_b128(rD, vod, sod + (unsigned)(yy * W * 4), 0);
            u1 = __builtin_amdgcn_raw_buffer_load_b128(rD, vod + 16u, sod + (unsigned)(yy * W * 4), 0);
        };
        // ---- row builder, in slices that the stage loop spreads between its MFMA groups -------------------------------------------
        const bool owns = q >= 1 && q <= 4;
        const int jq = q - 1;
        const unsigned wbase = owns ? (unsigned)(((jq >> 1) * 32 + ch) * 16 + (jq & 1) * 8) : (unsigned)(4 * 3 * 2 * 2 * 32 * 16 + tid * 8);
        const unsigned wmul = owns ? 1u : 0u;      // (the other lanes write their own 8 dump bytes whatever the slot / copy)
        struct RowRegs {      // one row's quad as packed fp16 pairs (plain 32-bit words: element shifts are alignbyte / shift-or on registers;
                              // with fp16 vectors or arrays here the compiler went through scratch memory to shift by one element)
            unsigned H01, H23, L01, L23, left, right;
        };
        auto pk = [](_Float16 lo16, _Float16 hi16) -> unsigned {
            return (unsigned)__builtin_bit_cast(unsigned short, lo16) | ((unsigned)__builtin_bit_cast(unsigned short, hi16) << 16);
        };
        auto split_row = [&](const rr_u4& raw, RowRegs& g) {
            float v[4] = {__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3])};
            const float pick = esel == 0 ? v[0] : (esel == 1 ? v[1] : (esel == 2 ? v[2] : v[3]));
            v[3] = q == 0 ? pick : v[3];
            v[0] = q >= 5 ? pick : v[0];
            _Float16 u0, u1, u2, u3, w0, w1, w2, w3;
            split2h(v[0] * sx, u0, w0);
            split2h(v[1] * sx, u1, w1);
            split2h(v[2] * sx, u2, w2);
            split2h(v[3] * sx, u3, w3);
            g.H01 = pk(u0, u1);
            g.H23 = pk(u2, u3);
            g.L01 = pk(w0, w1);
            g.L23 = pk(w2, w3);
        };
        auto edges = [&](RowRegs& g) {
            // edge elements of the neighbouring lanes (same channel: 8 consecutive lanes of one DPP row): (h, l) packed into one word
            const int e3 = (int)((g.H23 >> 16) | (g.L23 & 0xffff0000u));
            const int e0 = (int)((g.H01 & 0xffffu) | (g.L01 << 16));
            g.left = (unsigned)__builtin_amdgcn_update_dpp(0, e3, 0x111, 0xf, 0xf, false);      // row_shr:1: (h3, l3) of lane - 1
            g.right = (unsigned)__builtin_amdgcn_update_dpp(0, e0, 0x101, 0xf, 0xf, false);     // row_shl:1: (h0, l0) of lane + 1
        };
        // copy s holds xpad[x0 + p + s], p = 0 .. 15: s = 1 is this thread's own quad, s = 0 starts one element to the left, s = 2 one to the right
        typedef unsigned int rr_u2 __attribute__((ext_vector_type(2)));
        auto write_copy = [&](int slot, int sc, const RowRegs& g) {
            rr_u2 Hq, Lq;
            const unsigned Hmid = __builtin_amdgcn_alignbyte(g.H23, g.H01, 2), Lmid = __builtin_amdgcn_alignbyte(g.L23, g.L01, 2);      // (e1, e2)
            if (sc == 0) {
                Hq = rr_u2{(g.left & 0xffffu) | (g.H01 << 16), Hmid};
                Lq = rr_u2{(g.left >> 16) | (g.L01 << 16), Lmid};
            } else if (sc == 1) {
                Hq = rr_u2{g.H01, g.H23};
                Lq = rr_u2{g.L01, g.L23};
            } else {
                Hq = rr_u2{Hmid, (g.H23 >> 16) | (g.right << 16)};
                Lq = rr_u2{Lmid, (g.L23 >> 16) | (g.right & 0xffff0000u)};
            }
            *reinterpret_cast<rr_u2*>(xs_bytes + wbase + wmul * rec_off(slot, sc, 0, 0, 0)) = Hq;
            *reinterpret_cast<rr_u2*>(xs_bytes + wbase + wmul * rec_off(slot, sc, 1, 0, 0)) = Lq;
        };
        auto build = [&](int slot, const rr_u4& raw) {
            RowRegs g;
            split_row(raw, g);
            edges(g);
            write_copy(slot, 0, g);
            write_copy(slot, 1, g);
            write_copy(slot, 2, g);
        };
        rr_h8 Bh[3], Bl[3];
        auto rdB = [&](int buf, int ybase, int tap) {      // B fragment of tap (r, s) of the stage whose first padded row is ybase
            const int r = tap / 3, sc = tap % 3;
            Bh[buf] = *reinterpret_cast<const rr_h8*>(xs_bytes + rec_off((ybase + r) & 3, sc, 0, hi, lo));
            Bl[buf] = *reinterpret_cast<const rr_h8*>(xs_bytes + rec_off((ybase + r) & 3, sc, 1, hi, lo));
        };
        // One stage.  Entering stage y: rows y .. y + 2 are in LDS (row y + 2 since the barrier that ended stage y - 1), the fragments of
        // taps 0 and 1 are in flight, Acur holds dy of this stage, `din` dy of stage y + 1 and `xin` row y + 3 (both loaded a stage ago).
        // The stage loads dy of stage y + 2 and row y + 4 into `dout` / `xout` and leaves the next A fragment in Anext: the caller swaps
        // the roles from stage to stage, so no register copy has to wait for a load in flight.
        struct AFrag {
            rr_h8 h, l;
        };
        auto stage = [&](int y, const AFrag& Acur, AFrag& Anext, const rr_u4& din0, const rr_u4& din1, const rr_u4& xin, rr_u4& dout0, rr_u4& dout1,
                         rr_u4& xout) {
            __builtin_amdgcn_sched_barrier(0);
            lddy(y + 2, dout0, dout1);          // (rows / stages past the end: clamped, harmless)
            xout = ldrow(y + 4);
            const float dv[8] = {__uint_as_float(din0[0]), __uint_as_float(din0[1]), __uint_as_float(din0[2]), __uint_as_float(din0[3]),
                                 __uint_as_float(din1[0]), __uint_as_float(din1[1]), __uint_as_float(din1[2]), __uint_as_float(din1[3])};
            RowRegs g;
            const int wslot = (y + 3) & 3;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // group t: the LDS reads of tap t + 2 (taps 9, 10 = taps 0, 1 of the next stage: rows y + 1, written long ago), the three
                // MFMAs of tap t, and one slice of the vector work: an element of the next A fragment, a piece of the row builder
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < 9) rdB((t + 2) % 3, y, t + 2);
                else rdB((t + 2) % 3, y + 1, t + 2 - 9);
                rr_f16v c = acc[t];      // (l, h) (h, l) (h, h): smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Acur.l, Bh[t % 3], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Acur.h, Bl[t % 3], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(Acur.h, Bh[t % 3], c, 0, 0, 0);
                acc[t] = c;
                if (t < 8) {
                    _Float16 u, v;
                    split2h(dv[t] * sdy, u, v);
                    Anext.h[t] = u;
                    Anext.l[t] = v;
                }
                if (t == 0) split_row(xin, g);
                if (t == 1) edges(g);
                if (t == 2) write_copy(wslot, 0, g);     // (the stage past the last row writes a slot nobody reads before the next strip rebuilds it)
                if (t == 3) write_copy(wslot, 1, g);
                if (t == 4) write_copy(wslot, 2, g);
            }
            // End of the stage: row y + 3 must be written and every wave done reading the slot row y + 4 goes to.  The row's LDS writes
            // sit in groups 2 .. 4, at least the eight LDS reads of groups 5 .. 8 follow them (the fences keep the groups apart), and a
            // wave's LDS operations complete in order: waiting until at most 8 are outstanding covers the writes and leaves the prefetched
            // fragments of the next stage's taps 0 and 1 in flight across the barrier (__syncthreads would wait for them: lgkmcnt(0)).
            // The slots those reads and tap 8's touch are not the one the next stage writes.
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8), vmcnt / expcnt untouched
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- strip prologue: rows 0, 1, 2 into their slots; the A fragment of stage 0; dy of stage 1 and row 3 in registers -----------
        rr_u4 r0 = ldrow(0), r1 = ldrow(1), r2 = ldrow(2), xa = ldrow(3), xb2, dA0, dA1, da0, da1, db0, db1;
        lddy(0, dA0, dA1);
        lddy(1, da0, da1);
        __syncthreads();        // the previous strip's last stage has read its rows
        build(0, r0);
        build(1, r1);
        build(2, r2);
        AFrag A0, A1;
        {
            const float dv[8] = {__uint_as_float(dA0[0]), __uint_as_float(dA0[1]), __uint_as_float(dA0[2]), __uint_as_float(dA0[3]),
                                 __uint_as_float(dA1[0]), __uint_as_float(dA1[1]), __uint_as_float(dA1[2]), __uint_as_float(dA1[3])};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 u, v;
                split2h(dv[j] * sdy, u, v);
                A0.h[j] = u;
                A0.l[j] = v;
            }
        }
        __syncthreads();
        rdB(0, 0, 0);
        rdB(1, 0, 1);
        int y = 0;
#pragma unroll 1
        for (; y + 1 < H; y += 2) {
            stage(y, A0, A1, da0, da1, xa, db0, db1, xb2);
            stage(y + 1, A1, A0, db0, db1, xb2, da0, da1, xa);
        }
        if (y < H) stage(y, A0, A1, da0, da1, xa, db0, db1, xb2);
    }

    // partial sums: part[split][tap][k][c]; acc[t][r] is (row (r / 4) * 8 + hi * 4 + r % 4 of the wave's 32 rows, column lo)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* out = a.part + (((size_t)split * 9 + t) * a.K + mt * 128 + wave * 32) * a.C + cb * 32 + lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)((r >> 2) * 8 + hi * 4 + (r & 3)) * a.C] = acc[t][r];
    }
}

// ---- bf16 tensors: the same walk with ONE product per tap -------------------------------------------------------------------------------
// The stored bf16 patterns go to LDS and to the A fragment as they are (no scaling, no split, no vector arithmetic apart from the element
// shifts of the row copies), 9 MFMAs and 9 LDS reads per wave and stage.  A stage is then only ~300 matrix cycles long, far shorter than a
// trip to memory: dy fragments and x rows are loaded FOUR stages ahead into register rings (the stage loop is unrolled by four, ring slots
// are compile-time; stages past the last row multiply zeros: their dy loads are killed by an out-of-range offset).
typedef __bf16 rr_b8 __attribute__((ext_vector_type(8)));
typedef unsigned int rr_w4 __attribute__((ext_vector_type(4)));
typedef unsigned int rr_w2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256, 2) rowring_wgrad_bf16_kernel(RowRingArgs a) {
    __shared__ __attribute__((aligned(16))) rr_h8 Xs[4 * 3 * 2 * 32 + 128];      // [row slot][shift s][k half][channel] records of 8 pixels + dump bytes
    const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int tiles = a.nmt * a.ncb;
    const int split = wg / tiles, tile = wg - split * tiles;
    const int mt = tile % a.nmt, cb = tile / a.nmt;
    const int H = a.H, W = a.W, HW = H * W;
    if (wg == 0 && tid == 0) {
        a.scales[0] = 1.f;
        a.scales[1] = 1.f;
    }
    const int ch = tid >> 3, q = tid & 7;
    char* const xs_bytes = reinterpret_cast<char*>(&Xs[0]);
    auto rec_off = [&](int slot, int s, int khalf, int c) -> unsigned { return (unsigned)(((((slot * 3 + s) * 2 + khalf) * 32) + c) * 16); };
    rr_f16v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int wq = W >> 4;
    const int s_begin = split * a.strips_per_split;
    const int s_end = s_begin + a.strips_per_split < a.strips ? s_begin + a.strips_per_split : a.strips;
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((size_t)a.N * a.C * HW * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.DY), 0, (int)((size_t)a.N * a.K * HW * 2), 0x00020000);
    const unsigned vod = (unsigned)(((mt * 128 + wave * 32 + lo) * HW + 8 * hi) * 2);
    const bool owns = q >= 1 && q <= 4;
    const int jq = q - 1;
    const unsigned wbase = owns ? (unsigned)(((jq >> 1) * 32 + ch) * 16 + (jq & 1) * 8) : (unsigned)(4 * 3 * 2 * 32 * 16 + tid * 8);
    const unsigned wmul = owns ? 1u : 0u;
    for (int strip = s_begin; strip < s_end; ++strip) {
        const int n = strip / wq, x0 = (strip - n * wq) << 4;
        const int colq = q == 0 ? (x0 == 0 ? 1 : x0 - 1) : (q >= 5 ? (x0 + 16 >= W ? W - 2 : x0 + 16) : x0 + 4 * (q - 1));
        const int esel = colq & 3;
        const unsigned vox = (unsigned)((ch * HW + (colq & ~3)) * 2);
        const unsigned sox = (unsigned)((n * a.C + cb * 32) * HW * 2);
        const unsigned sod = (unsigned)((n * a.K * HW + x0) * 2);
        auto ldrow = [&](int py) -> rr_w2 {      // padded row py (clamped to H + 1), mirrored at the image edge
            int sy = (py > H + 1 ? H + 1 : py) - 1;
            sy = sy < 0 ? -sy : sy;
            sy = sy >= H ? 2 * (H - 1) - sy : sy;
            return __builtin_amdgcn_raw_buffer_load_b64(rX, vox, sox + (unsigned)(sy * W * 2), 0);
        };
        auto lddy = [&](int y) -> rr_w4 {        // dy of stage y; a stage past the last row loads zeros
            return __builtin_amdgcn_raw_buffer_load_b128(rD, y < H ? vod : 0x80000000u, y < H ? sod + (unsigned)(y * W * 2) : 0u, 0);
        };
        struct RowRegs {
            unsigned E01, E23, left, right;
        };
        auto take_row = [&](const rr_w2& raw, RowRegs& g) {
            const unsigned w = (esel & 2) ? raw[1] : raw[0];
            const unsigned pick = (esel & 1) ? (w >> 16) : (w & 0xffffu);
            g.E01 = q >= 5 ? ((raw[0] & 0xffff0000u) | pick) : raw[0];
            g.E23 = q == 0 ? ((raw[1] & 0xffffu) | (pick << 16)) : raw[1];
        };
        auto edges = [&](RowRegs& g) {
            g.left = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(g.E23 >> 16), 0x111, 0xf, 0xf, false);        // element 3 of lane - 1
            g.right = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(g.E01 & 0xffffu), 0x101, 0xf, 0xf, false);   // element 0 of lane + 1
        };
        auto write_copy = [&](int slot, int sc, const RowRegs& g) {
            const unsigned mid = __builtin_amdgcn_alignbyte(g.E23, g.E01, 2);
            rr_w2 v;
            if (sc == 0) v = rr_w2{g.left | (g.E01 << 16), mid};
            else if (sc == 1) v = rr_w2{g.E01, g.E23};
            else v = rr_w2{mid, (g.E23 >> 16) | (g.right << 16)};
            *reinterpret_cast<rr_w2*>(xs_bytes + wbase + wmul * rec_off(slot, sc, 0, 0)) = v;
        };
        auto build = [&](int slot, const rr_w2& raw) {
            RowRegs g;
            take_row(raw, g);
            edges(g);
            write_copy(slot, 0, g);
            write_copy(slot, 1, g);
            write_copy(slot, 2, g);
        };
        rr_h8 Bq[3];
        auto rdB = [&](int buf, int ybase, int tap) {
            const int r = tap / 3, sc = tap % 3;
            Bq[buf] = *reinterpret_cast<const rr_h8*>(xs_bytes + rec_off((ybase + r) & 3, sc, hi, lo));
        };
        // rings: dring[k] holds dy of the next stage with y % 4 == k, xring[k] the padded row that stage builds (row y + 3)
        rr_w4 dring[4];
        rr_w2 xring[4];
        const rr_w2 r0 = ldrow(0), r1 = ldrow(1), r2 = ldrow(2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dring[k] = lddy(k);
            xring[k] = ldrow(k + 3);
        }
        __syncthreads();        // the previous strip's last stage has read its rows
        build(0, r0);
        build(1, r1);
        build(2, r2);
        __syncthreads();
        rdB(0, 0, 0);
        rdB(1, 0, 1);
#pragma unroll 1
        for (int y0 = 0; y0 < H; y0 += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = y0 + k;
                const rr_b8 A = __builtin_bit_cast(rr_b8, dring[k]);
                RowRegs g;
                const int wslot = (y + 3) & 3;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 2 < 9) rdB((t + 2) % 3, y, t + 2);
                    else rdB((t + 2) % 3, y + 1, t + 2 - 9);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, __builtin_bit_cast(rr_b8, Bq[t % 3]), acc[t], 0, 0, 0);
                    if (t == 0) take_row(xring[k], g);
                    if (t == 1) edges(g);
                    if (t == 2) write_copy(wslot, 0, g);
                    if (t == 3) write_copy(wslot, 1, g);
                    if (t == 4) write_copy(wslot, 2, g);
                }
                __builtin_amdgcn_sched_barrier(0);
                dring[k] = lddy(y + 4);           // four stages ahead, into the ring slot this stage has just finished with
                xring[k] = ldrow(y + 7);
                // (as in the fp32 kernel: the row's writes are followed by at least the four reads of groups 5 .. 8)
                __builtin_amdgcn_s_waitcnt(0xC47F);      // lgkmcnt(4)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* out = a.part + (((size_t)split * 9 + t) * a.K + mt * 128 + wave * 32) * a.C + cb * 32 + lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)((r >> 2) * 8 + hi * 4 + (r & 3)) * a.C] = acc[t][r];
    }
}

struct RowRingPlan {
    int nmt, ncb, strips, per, splits;
    size_t part_bytes, total;
};
static bool rowring_plan(const pcgan_conv_desc* d, RowRingPlan* p) {
    if (!d || (d->dtype != PCGAN_F32 && d->dtype != PCGAN_BF16) || d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->pad_mode != 1 || d->P != d->H || d->Q != d->W)
        return false;
    if (d->H < 3 || d->W < 16 || d->W % 16 != 0 || d->K % 128 != 0 || d->C % 32 != 0 || d->N < 1) return false;
    if ((size_t)d->N * d->C * d->H * d->W * 4 >= 0x80000000ull || (size_t)d->N * d->K * d->H * d->W * 4 >= 0x80000000ull) return false;
    if (d->dtype == PCGAN_BF16 && option(OPT_WGRAD_ROWRING) == 3) return false;      // option value 3: fp32 tensors only (A/B measurement of the bf16 form)
    p->nmt = d->K / 128;
    p->ncb = d->C / 32;
    p->strips = d->N * (d->W / 16);
    const int tiles = p->nmt * p->ncb;
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    int want = (2 * cus) / tiles;      // two workgroups per CU, one round
    want = want < 1 ? 1 : (want > p->strips ? p->strips : want);
    p->per = (p->strips + want - 1) / want;
    p->splits = (p->strips + p->per - 1) / p->per;
    p->part_bytes = (size_t)p->splits * 9 * d->K * d->C * 4;
    p->total = 256 + align_up(p->part_bytes, 256);
    return true;
}

}  // namespace pcgan

extern "C" int pcgan_conv2d_wgrad_rowring_supported(const pcgan_conv_desc* d) {
    pcgan::RowRingPlan p;
    return pcgan::rowring_plan(d, &p) ? 1 : 0;
}

extern "C" size_t pcgan_conv2d_wgrad_rowring_workspace_bytes(const pcgan_conv_desc* d) {
    pcgan::RowRingPlan p;
    return pcgan::rowring_plan(d, &p) ? p.total : 0;
}

extern "C" int pcgan_conv2d_bwd_weight_rowring(const pcgan_conv_desc* d, const void* x, const float* x_amax, int n_xamax, const void* dy,
                                               const float* dy_amax, int n_dyamax, float* dw, int accumulate, void* ws, size_t ws_bytes,
                                               pcgan_stream_t s) {
    using namespace pcgan;
    RowRingPlan p;
    PCGAN_CHECK(rowring_plan(d, &p), "conv2d_bwd_weight_rowring: unsupported shape (3x3 stride 1 reflection padding 1, fp32, W %% 16 == 0, K %% 128 == 0, C %% 32 == 0)");
    const bool half = d->dtype == PCGAN_BF16;
    PCGAN_CHECK(x && dy && dw && ws && (half || (x_amax && dy_amax && n_xamax > 0 && n_dyamax > 0)), "conv2d_bwd_weight_rowring: null pointer");
    PCGAN_CHECK(ws_bytes >= p.total, "conv2d_bwd_weight_rowring: workspace too small (%zu < %zu)", ws_bytes, p.total);
    hipStream_t st = (hipStream_t)s;
    RowRingArgs a;
    a.X = (const float*)x; a.DY = (const float*)dy;
    a.scales = (float*)ws;
    a.part = (float*)((char*)ws + 256);
    a.x_amax = x_amax; a.dy_amax = dy_amax; a.x_namax = n_xamax; a.dy_namax = n_dyamax;
    a.N = d->N; a.C = d->C; a.K = d->K; a.H = d->H; a.W = d->W;
    a.nmt = p.nmt; a.ncb = p.ncb;
    a.strips = p.strips; a.strips_per_split = p.per;
    a.nwg = p.nmt * p.ncb * p.splits;
    {
        TimerScope whole(timer_kind_res(d, TIMER_RES_WGRAD), st);
        {
            TimerScope main_only(timer_kind_res(d, TIMER_RES_WGRAD_MAIN), st);
            if (half) hipLaunchKernelGGL(rowring_wgrad_bf16_kernel, dim3((unsigned)((a.nwg + 7) / 8 * 8)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(rowring_wgrad_kernel, dim3((unsigned)((a.nwg + 7) / 8 * 8)), dim3(256), 0, st, a);
        }
        if (launch_wgd_reduce(a.part, dw, a.scales, p.splits, d->K, d->C, accumulate, st)) return 2;
    }
    PCGAN_LAUNCH_CHECK();
    return 0;
}
