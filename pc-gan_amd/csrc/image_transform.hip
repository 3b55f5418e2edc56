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
// Bound: HBM (byte work, ~0.3 MB per image).  One workgroup = one image x a band of output rows: phase A runs the
// horizontal pass for the source rows the band needs (only the cropped columns) into LDS as bytes, phase B the vertical
// pass out of LDS, normalises and stores rows of 4-byte floats coalesced along x.
#include "common.h"

// The last operations must round like the separate torch ops they stand for: this file is compiled with
// -ffp-contract=off (Makefile), so no mul+add pair is fused into an fma.

namespace pcgan {

constexpr int IMG_PRECISION_BITS = 32 - 8 - 2;   // Pillow: Resample.c PRECISION_BITS

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
    int H, W, RH, RW, FH, FW, ksh, ksv, OC, band, max_rows;
};

__global__ __launch_bounds__(256) void image_transform_kernel(ImgArgs a) {
    extern __shared__ uint8_t T[];     // [rows][FW][3] horizontally resized bytes of this band's source rows
    const int img = blockIdx.y;
    const int r0 = blockIdx.x * a.band;
    const int r1 = min(r0 + a.band, a.FH);
    const int cx = a.aug[img * 4 + 0], cy = a.aug[img * 4 + 1], flip = a.aug[img * 4 + 2], dst = a.aug[img * 4 + 3];
    const int ylo = a.bv[(cy + r0) * 2];
    const int yhi = a.bv[(cy + r1 - 1) * 2] + a.bv[(cy + r1 - 1) * 2 + 1];
    const int rows = min(yhi - ylo, a.max_rows);
    const uint8_t* src = a.src + (size_t)img * a.H * a.W * 3;
    const int rowlen = a.FW * 3;

    // phase A: horizontal pass, element = (row, cropped column, channel); lanes run along (column, channel)
    for (int e = threadIdx.x; e < rows * rowlen; e += blockDim.x) {
        const int y = e / rowlen, xc = e - y * rowlen;
        const int x = xc / 3, c = xc - x * 3;
        const int rx = cx + x;
        const int xmin = a.bh[rx * 2], cnt = a.bh[rx * 2 + 1];
        const int* k = a.kh + (size_t)rx * a.ksh;
        const uint8_t* p = src + ((size_t)(ylo + y) * a.W + xmin) * 3 + c;
        int acc = 1 << (IMG_PRECISION_BITS - 1);
        for (int i = 0; i < cnt; ++i) acc += (int)p[i * 3] * k[i];
        T[e] = (uint8_t)clip8(acc);
    }
    __syncthreads();

    // phase B: vertical pass + normalisation, one thread per output pixel (all channels), lanes along x
    const size_t plane = (size_t)a.FH * a.FW;
    float* out = a.out + (size_t)dst * a.OC * plane;
    for (int e = threadIdx.x; e < (r1 - r0) * a.FW; e += blockDim.x) {
        const int oy = r0 + e / a.FW, ox = e % a.FW;
        const int col = flip ? a.FW - 1 - ox : ox;
        const int ry = cy + oy;
        const int ymin = a.bv[ry * 2], cnt = a.bv[ry * 2 + 1];
        const int* k = a.kv + (size_t)ry * a.ksv;
        const uint8_t* p = T + (size_t)(ymin - ylo) * rowlen + col * 3;
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
        } else {   // (A[0] * 0.299 + A[1] * 0.587) + A[2] * 0.114, no contraction
            out[o] = __fadd_rn(__fadd_rn(__fmul_rn(f0, 0.299f), __fmul_rn(f1, 0.587f)), __fmul_rn(f2, 0.114f));
        }
    }
}

}  // namespace pcgan

extern "C" int pcgan_image_transform_band(const pcgan_image_desc* d, const int* bv_host, int* band, int* max_rows) {
    PCGAN_CHECK(d && bv_host && band && max_rows, "image_transform_band: null argument");
    PCGAN_CHECK(d->FH > 0 && d->FH <= d->RH && d->FW > 0 && d->FW <= d->RW, "image_transform_band: crop %dx%d outside the resized image %dx%d",
                d->FH, d->FW, d->RH, d->RW);
    // tallest band of output rows whose source rows (for ANY crop offset) fit 48 KB of LDS
    const int rowlen = d->FW * 3, cap = 48 * 1024 / rowlen;
    for (int b = d->FH < 32 ? d->FH : 32; b >= 1; --b) {
        int worst = 0;
        for (int y0 = 0; y0 + b <= d->RH; ++y0) {
            const int rows = bv_host[(y0 + b - 1) * 2] + bv_host[(y0 + b - 1) * 2 + 1] - bv_host[y0 * 2];
            if (rows > worst) worst = rows;
        }
        if (worst <= cap) {
            *band = b;
            *max_rows = worst;
            return 0;
        }
    }
    pcgan::set_error("image_transform_band: one output row needs more source rows than fit in LDS (%d -> %d rows)", d->H, d->RH);
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
    PCGAN_CHECK(band > 0 && max_rows > 0 && (size_t)max_rows * d->FW * 3 <= 48 * 1024, "image_transform: band %d / rows %d do not fit LDS", band, max_rows);
    pcgan::ImgArgs a{src, kh, bh, kv, bv, aug, out, d->H, d->W, d->RH, d->RW, d->FH, d->FW, d->ksize_h, d->ksize_v, d->out_channels, band, max_rows};
    const int bands = (d->FH + band - 1) / band;
    hipLaunchKernelGGL(pcgan::image_transform_kernel, dim3(bands, n), dim3(256), (size_t)max_rows * d->FW * 3, (hipStream_t)s, a);
    PCGAN_LAUNCH_CHECK();
    return 0;
}
