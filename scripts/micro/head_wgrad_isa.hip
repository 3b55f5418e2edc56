// ISA diagnosis of round 3's determinism finding: the inner loop of smallm_wgrad_strip_kernel<MODE_FWD_REFLECT, 7, float>
// (csrc/igemm_conv.hip) in its two forms -- PK = 1: plain C (`acc += dy * x`, which the SLP vectoriser pairs into v_pk_fma_f32 with
// op_sel), PK = 0: one inline v_fmac_f32 per term (the shipped form).  Build the ISA of both and compare the wait counts / register
// pairs around the FMA block:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o head_wgrad_isa.s head_wgrad_isa.hip
// (scripts/micro/head_wgrad_isa.md holds what was found.)  Not part of the library.
#include <hip/hip_runtime.h>

static constexpr unsigned OOB = 0x80000000u, SM_INV = 0x40000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ldf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
struct WgradArgs {
    const void* dY;
    const void* X;
    int dtype;
    float* Wp;
    int M, Kp, N, Cg, Cgp, Hg, Wg, Ho, Wo;
    int sl, pad, S;
    int magicS;
    int Ptot, chunks_per_split;
    unsigned x_bytes, dy_bytes;
};

template <int NT, int PK>
__global__ void __launch_bounds__(256) strip_kernel(WgradArgs a) {
    constexpr unsigned ES = 4;
    constexpr int PX = 8, NW = PX + NT - 1, MO = 3;
    constexpr int SG = NT > 4 ? 4 : NT;
    __shared__ float red[4][SG * NT * MO];
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const int s_lo = blockIdx.z * SG;
    const int R = a.Kp / a.Cgp / a.S;
    const int HoWo = a.Ho * a.Wo, HgWg = a.Hg * a.Wg;
    const int spc = (a.Ho + PX - 1) / PX;
    const int nstrips = a.N * spc * a.Wo;
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.X, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.dY, a.dy_bytes);
    float acc[SG][NT][MO];
#pragma unroll
    for (int sj = 0; sj < SG; ++sj)
#pragma unroll
        for (int ri = 0; ri < NT; ++ri)
#pragma unroll
            for (int m = 0; m < MO; ++m) acc[sj][ri][m] = 0.f;
    const int sbeg = blockIdx.y * a.chunks_per_split;
    int send = sbeg + a.chunks_per_split;
    if (send > nstrips) send = nstrips;
    for (int sg = sbeg + tid; sg < send; sg += 256) {
        const int n = sg / (spc * a.Wo);
        const int rem = sg - n * spc * a.Wo;
        const int ss = rem / a.Wo;
        const int ox = rem - ss * a.Wo;
        const int oy0 = ss * PX;
        unsigned rowoff[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int iy = oy0 - a.pad + k;
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.Hg ? 2 * (a.Hg - 1) - iy : iy;
            rowoff[k] = ((unsigned)iy < (unsigned)a.Hg) ? (unsigned)((n * a.Cg + c) * HgWg + iy * a.Wg) * ES : SM_INV;
        }
        auto col_off = [&](int sj) {
            int ix = ox - a.pad + s_lo + sj;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.Wg ? 2 * (a.Wg - 1) - ix : ix;
            return (s_lo + sj < a.S && (unsigned)ix < (unsigned)a.Wg) ? (unsigned)ix * ES : SM_INV;
        };
        float xin[2][NW];
        {
            const unsigned co = col_off(0);
#pragma unroll
            for (int k = 0; k < NW; ++k) xin[0][k] = ldf(rX, rowoff[k] + co, 0u);
        }
        float dyv[MO][PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = (oy0 + j < a.Ho) ? (unsigned)(n * a.M * HoWo + (oy0 + j) * a.Wo + ox) * ES : OOB;
#pragma unroll
            for (int m = 0; m < MO; ++m) dyv[m][j] = m < a.M ? ldf(rY, vo, (unsigned)(m * HoWo) * ES) : 0.f;
        }
#pragma unroll
        for (int sj = 0; sj < SG; ++sj) {
            if (sj + 1 < SG) {
                const unsigned co = col_off(sj + 1);
#pragma unroll
                for (int k = 0; k < NW; ++k) xin[(sj + 1) & 1][k] = ldf(rX, rowoff[k] + co, 0u);
            }
#pragma unroll
            for (int ri = 0; ri < NT; ++ri)
#pragma unroll
                for (int j = 0; j < PX; ++j)
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        if (PK) acc[sj][ri][m] += dyv[m][j] * xin[sj & 1][j + ri];
                        else asm("v_fmac_f32 %0, %1, %2" : "+v"(acc[sj][ri][m]) : "v"(dyv[m][j]), "v"(xin[sj & 1][j + ri]));
                    }
        }
    }
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
template __global__ void strip_kernel<7, 1>(WgradArgs);
template __global__ void strip_kernel<7, 0>(WgradArgs);
