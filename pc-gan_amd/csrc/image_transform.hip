// Image pipeline of the loader on the GPU: antialiased bicubic resize -> crop -> horizontal flip -> [0,1] ->
// (x - 0.5) / 0.5, uint8 HWC in, fp32 NCHW out, one launch per batch of equally sized source images.
//
// Replaces (reference data/base_dataset.py:24-64, get_transform 'resize_and_crop' / 'crop'):
//   transforms.Resize([loadSize, loadSize], Image.BICUBIC) -> RandomCrop(fineSize) -> RandomHorizontalFlip ->
//   ToTensor -> Normalize((.5,.5,.5), (.5,.5,.5))   [+ the RGB -> gray mix of data/wsgan_emb_dataset.py:46-49]
// torchvision's Resize on a PIL image is Pillow's Image.resize: a separable two-pass filter in 22-bit fixed point with
// the intermediate image rounded back to uint8 (horizontal pass first).  The host builds the integer coefficient
// tables exactly as Pillow does (pcgan_amd/data/gpu_transform.py); the kernel is integer arithmetic up to the final
// normalisation, so results are BIT-EXACT with the PIL path.
//
// Bound: HBM (byte work, ~0.3 MB per image).  One workgroup = one image x a band of output rows:
//   phase 0  the source rows the band needs (one contiguous byte range of the image) and the coefficient rows of this
//            crop window are copied into LDS with aligned 4-byte loads;
//   phase A  horizontal pass LDS -> LDS (bytes), only the cropped columns, waves along rows, lanes along (column, channel);
//   phase B  vertical pass out of LDS, normalisation, rows of floats stored coalesced along x.
// The last operations must round like the separate torch ops they stand for: this file is compiled with
// -ffp-contract=off (Makefile), so no mul+add pair is fused into an fma.
#include "common.h"

namespace pcgan {

constexpr int IMG_PRECISION_BITS = 32 - 8 - 2;   // Pillow: Resample.c PRECISION_BITS
constexpr int IMG_LDS_BYTES = 64 * 1024;

__device__ __forceinline__ int clip8(int acc) {
    int v = acc >> IMG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct ImgArgs {
    const uint8_t* src;     // [n][H][W][3]
    const int* kh;          // [RW][ksh]  horizontal coefficients (fixed point)
    const int* bh;          // [RW][2]    first source column, tap count
    const int* kv;          // [RH][ksv]
    const int* bv;          // [RH][2]
    const int* aug;         // [n][4]     crop x0, crop y0, flip, destination image index
    float* out;             // [*][OC][FH][FW]
    size_t src_bytes;       // n * H * W * 3
    int H, W, RH, RW, FH, FW, ksh, ksv, OC, band, max_rows;
};

// LDS layout (ints first, then bytes): kh[FW][ksh] | bh[FW][2] | kv[band][ksv] | bv[band][2] | S[max_rows*W*3 + 8] | T[max_rows*FW*3]
static inline size_t img_lds_bytes(int W, int FW, int ksh, int ksv, int band, int rows) {
    return (size_t)4 * (FW * ksh + FW * 2 + band * ksv + band * 2) + align_up((size_t)rows * W * 3 + 8, 4) + (size_t)rows * FW * 3;
}

__global__ __launch_bounds__(256) void image_transform_kernel(ImgArgs a) {
    extern __shared__ int lds_i[];
    int* kh = lds_i;
    int* bh = kh + a.FW * a.ksh;
    int* kv = bh + a.FW * 2;
    int* bv = kv + a.band * a.ksv;
    uint8_t* S = (uint8_t*)(bv + a.band * 2);
    const int srow = a.W * 3, rowlen = a.FW * 3;
    uint8_t* T = S + (((size_t)a.max_rows * srow + 8 + 3) & ~(size_t)3);

    const int img = blockIdx.y;
    const int r0 = blockIdx.x * a.band;
    const int r1 = min(r0 + a.band, a.FH);
    const int cx = a.aug[img * 4 + 0], cy = a.aug[img * 4 + 1], flip = a.aug[img * 4 + 2], dst = a.aug[img * 4 + 3];
    const int ylo = a.bv[(cy + r0) * 2];
    const int yhi = a.bv[(cy + r1 - 1) * 2] + a.bv[(cy + r1 - 1) * 2 + 1];
    const int rows = min(yhi - ylo, a.max_rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // phase 0: coefficient rows of this window, and the byte range [first, first + rows * srow) of the batch, 4 bytes at a time
    for (int e = tid; e < a.FW * a.ksh; e += 256) kh[e] = a.kh[(size_t)cx * a.ksh + e];
    for (int e = tid; e < a.FW * 2; e += 256) bh[e] = a.bh[cx * 2 + e];
    for (int e = tid; e < (r1 - r0) * a.ksv; e += 256) kv[e] = a.kv[(size_t)(cy + r0) * a.ksv + e];
    for (int e = tid; e < (r1 - r0) * 2; e += 256) bv[e] = a.bv[(cy + r0) * 2 + e];
    const size_t first = ((size_t)img * a.H + ylo) * srow;
    const size_t base = first & ~(size_t)3;              // a.src is at least 4-byte aligned (checked by the caller)
    const int head = (int)(first - base);
    const int words = (head + rows * srow + 3) >> 2;
    const bool tail_inside = base + (size_t)words * 4 <= a.src_bytes;      // false only for the last rows of the batch
    if (tail_inside) {
        const uint32_t* g = (const uint32_t*)(a.src + base);
        int e = tid;
        for (; e + 7 * 256 < words; e += 8 * 256) {          // 8 independent loads in flight per thread
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = g[e + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) ((uint32_t*)S)[e + u * 256] = v[u];
        }
        for (; e < words; e += 256) ((uint32_t*)S)[e] = g[e];
    } else {
        for (int e = tid; e < words; e += 256) {
            const size_t off = base + (size_t)e * 4;
            uint32_t v = 0;
            if (off + 4 <= a.src_bytes) {
                v = *(const uint32_t*)(a.src + off);
            } else {                                        // last word of the batch: never read past the buffer
                for (int b = 0; b < 4; ++b)
                    if (off + b < a.src_bytes) v |= (uint32_t)a.src[off + b] << (8 * b);
            }
            ((uint32_t*)S)[e] = v;
        }
    }
    __syncthreads();

    // phase A: horizontal pass
    for (int y = wave; y < rows; y += 4) {
        const uint8_t* row = S + head + y * srow;
        for (int xc = lane; xc < rowlen; xc += 64) {
            const int x = xc / 3, c = xc - x * 3;
            const int xmin = bh[x * 2], cnt = bh[x * 2 + 1];
            const int* k = kh + x * a.ksh;
            const uint8_t* p = row + xmin * 3 + c;
            int acc = 1 << (IMG_PRECISION_BITS - 1);
            for (int i = 0; i < cnt; ++i) acc += (int)p[i * 3] * k[i];
            T[y * rowlen + xc] = (uint8_t)clip8(acc);
        }
    }
    __syncthreads();

    // phase B: vertical pass + normalisation, one thread per output pixel (all channels), lanes along x
    const size_t plane = (size_t)a.FH * a.FW;
    float* out = a.out + (size_t)dst * a.OC * plane;
    for (int r = wave; r < r1 - r0; r += 4) {
        const int oy = r0 + r;
        const int ymin = bv[r * 2], cnt = bv[r * 2 + 1];
        const int* k = kv + r * a.ksv;
        for (int ox = lane; ox < a.FW; ox += 64) {
            const int col = flip ? a.FW - 1 - ox : ox;
            const uint8_t* p = T + (ymin - ylo) * rowlen + col * 3;
            int acc0 = 1 << (IMG_PRECISION_BITS - 1), acc1 = acc0, acc2 = acc0;
            for (int j = 0; j < cnt; ++j) {
                const int kj = k[j];
                acc0 += (int)p[j * rowlen + 0] * kj;
                acc1 += (int)p[j * rowlen + 1] * kj;
                acc2 += (int)p[j * rowlen + 2] * kj;
            }
            // ToTensor: float32(v) / 255 ; Normalize: (t - 0.5) / 0.5 -- correctly rounded fp32 ops in the same order
            const float f0 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(acc0), 255.0f), 0.5f), 0.5f);
            const float f1 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(acc1), 255.0f), 0.5f), 0.5f);
            const float f2 = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(acc2), 255.0f), 0.5f), 0.5f);
            const size_t o = (size_t)oy * a.FW + ox;
            if (a.OC == 3) {
                out[o] = f0;
                out[o + plane] = f1;
                out[o + 2 * plane] = f2;
            } else {   // (A[0] * 0.299 + A[1] * 0.587) + A[2] * 0.114 as three products and two sums
                out[o] = __fadd_rn(__fadd_rn(__fmul_rn(f0, 0.299f), __fmul_rn(f1, 0.587f)), __fmul_rn(f2, 0.114f));
            }
        }
    }
}

}  // namespace pcgan

extern "C" int pcgan_image_transform_band(const pcgan_image_desc* d, const int* bv_host, int* band, int* max_rows) {
    PCGAN_CHECK(d && bv_host && band && max_rows, "image_transform_band: null argument");
    PCGAN_CHECK(d->H > 0 && d->W > 0 && d->RH > 0 && d->RW > 0 && d->ksize_h > 0 && d->ksize_v > 0, "image_transform_band: bad geometry");
    PCGAN_CHECK(d->FH > 0 && d->FH <= d->RH && d->FW > 0 && d->FW <= d->RW, "image_transform_band: crop %dx%d outside the resized image %dx%d",
                d->FH, d->FW, d->RH, d->RW);
    // tallest band of output rows whose source rows (for ANY crop offset), window coefficients and intermediate rows fit LDS
    // (at most 16 rows: a batch of 64 images then gives 512 workgroups, two per CU)
    for (int b = d->FH < 16 ? d->FH : 16; b >= 1; --b) {
        int worst = 0;
        for (int y0 = 0; y0 + b <= d->RH; ++y0) {
            const int rows = bv_host[(y0 + b - 1) * 2] + bv_host[(y0 + b - 1) * 2 + 1] - bv_host[y0 * 2];
            if (rows > worst) worst = rows;
        }
        if (worst > 0 && pcgan::img_lds_bytes(d->W, d->FW, d->ksize_h, d->ksize_v, b, worst) <= (size_t)pcgan::IMG_LDS_BYTES) {
            *band = b;
            *max_rows = worst;
            return 0;
        }
    }
    pcgan::set_error("image_transform_band: one output row needs more source rows than fit in LDS (%dx%d -> %dx%d)", d->H, d->W, d->RH, d->RW);
    return 1;
}

extern "C" int pcgan_image_transform(const pcgan_image_desc* d, const uint8_t* src, const int* kh, const int* bh, const int* kv,
                                     const int* bv, const int* aug, float* out, int n, int band, int max_rows, pcgan_stream_t s) {
    PCGAN_CHECK(d && src && kh && bh && kv && bv && aug && out, "image_transform: null argument");
    PCGAN_CHECK(n > 0 && n <= 65535, "image_transform: batch %d outside 1..65535", n);
    PCGAN_CHECK(d->H > 0 && d->W > 0 && d->RH > 0 && d->RW > 0 && d->ksize_h > 0 && d->ksize_v > 0, "image_transform: bad geometry");
    PCGAN_CHECK(d->FH > 0 && d->FH <= d->RH && d->FW > 0 && d->FW <= d->RW, "image_transform: crop %dx%d outside the resized image %dx%d",
                d->FH, d->FW, d->RH, d->RW);
    PCGAN_CHECK(d->out_channels == 3 || d->out_channels == 1, "image_transform: out_channels must be 3 or 1");
    PCGAN_CHECK(((uintptr_t)src & 3) == 0, "image_transform: src must be 4-byte aligned");
    const size_t lds = band > 0 && max_rows > 0 ? pcgan::img_lds_bytes(d->W, d->FW, d->ksize_h, d->ksize_v, band, max_rows) : 0;
    PCGAN_CHECK(lds > 0 && lds <= (size_t)pcgan::IMG_LDS_BYTES, "image_transform: band %d / rows %d do not fit LDS", band, max_rows);
    pcgan::ImgArgs a{src, kh, bh, kv, bv, aug, out, (size_t)n * d->H * d->W * 3, d->H, d->W, d->RH, d->RW, d->FH, d->FW, d->ksize_h, d->ksize_v,
                     d->out_channels, band, max_rows};
    const int bands = (d->FH + band - 1) / band;
    hipLaunchKernelGGL(pcgan::image_transform_kernel, dim3(bands, n), dim3(256), lds, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
