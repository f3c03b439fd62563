// Narrow-N CausalConv3d: the decoder's conv_out (128 channels -> 3, 3x3x3; modeling_enc_dec.py:317-321 through
// CausalConv3d.forward, modeling_causal_conv.py:116-146).  As an implicit GEMM it is N = 3 padded to 128 columns -- 0.39
// PFLOP of padded work per 768p video for 0.009 useful, and every activation fetched 27 times (once per tap).  Here:
//   * one workgroup = one 16 x 16 pixel tile over ALL frames of the chunk.  The 18 x 18 x 128-channel input halo of ONE
//     input frame is staged in LDS (81 KiB, LDS-DMA, XOR-swizzled per pixel) and feeds the three output frames it
//     belongs to (temporal taps 2, 1, 0 of outputs s-2, s-1, s): three rolling accumulator sets, each input frame is
//     read once per tile instead of 27 times;
//   * v_mfma_f32_16x16x32_bf16 with the FILTERS as the 16-row operand (rows >= 8 are zero registers, rows 0..7 come from
//     a 54-KiB LDS copy of the first eight filter rows: every column the 8-channel output pitch can hold is computed, so
//     any n_valid in 1..8 is exact) and 16 pixels of a tile row as the columns: a lane of the first
//     two 16-lane groups ends up with output channels 0-3 / 4-7 of its pixel = one 8-byte store each ([T][H][W][8] bf16,
//     the layout the blend / uint8 kernels read);
//   * K order per output = (dt, dh, dw, c) ascending, the implicit GEMM's order.
#include "common.h"
#include "pyflow_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int TS = 16;                        // tile side (pixels)
constexpr int HS = TS + 2;                    // halo side
constexpr int CIN = 128;
constexpr int HALO_BYTES = HS * HS * CIN * 2; // 82 944 = 81 pieces of 1 KiB
constexpr int WROW = 27 * CIN * 2 + 32;       // bytes per filter row in LDS (+32: rows land on different banks)
constexpr int NROW = 8;                       // filter rows staged and multiplied (= the output pitch)
constexpr int SMEM = HALO_BYTES + NROW * WROW;

struct NArgs {
    const bf16_t* X; const bf16_t* Wt; const float* bias; bf16_t* Y;
    int T, H, W, Hp, Wp;
    long long in_base_off, out_base_off;
    int Hop, Wop, ldw;
    float out_scale;
};

__global__ __launch_bounds__(256, 1) void conv_narrow_kernel(const NArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wl = smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = p.W / TS;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;

    // ---- filter rows 0..7 -> LDS (16-byte pieces, plain loads: 54 KiB once per workgroup)
    for (int i = tid; i < NROW * (27 * CIN / 8); i += 256) {
        const int row = i / (27 * CIN / 8), ck = i - row * (27 * CIN / 8);
        *(u32x4_t*)(wl + row * WROW + ck * 16) = *(const u32x4_t*)(p.Wt + (long long)row * p.ldw + ck * 8);
    }

    // ---- halo DMA geometry: piece pc (0..80) = 4 halo pixels x 256 B; lane -> pixel pc*4 + lane/16, LDS chunk lane%16
    //      holds source chunk (lane%16) ^ (pixel & 15)
    const long long frame_el = (long long)p.Hp * p.Wp * CIN;
    auto stage = [&](int slot) {
        const bf16_t* xf = p.X + p.in_base_off + (long long)slot * frame_el;
        for (int pc = wid; pc < 81; pc += 4) {
            const int hp = pc * 4 + (lane >> 4);
            const int hy = hp / HS, hx = hp - hy * HS;
            const int c = (lane & 15) ^ (hp & 15);
            glds16(xf + ((long long)(y0 + hy) * p.Wp + (x0 + hx)) * CIN + c * 8, halo + pc * 1024);
        }
    };

    // fragment roles: lane -> (row-of-operand i = lane & 15, k group kq = lane >> 4)
    const int li = lane & 15, kq = lane >> 4;
    f32x4v acc[3][4];                          // [temporal tap 2 / 1 / 0 = outputs s-2 / s-1 / s][tile row of this wave]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[a][g] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    float bq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bq[r] = (kq < 2 && p.bias) ? p.bias[4 * kq + r] : 0.f;

    const int S = p.T + 2;                     // input slots
    for (int s = 0; s < S; ++s) {
        __syncthreads();                       // everyone is done reading the previous halo (and, first time, the filters are written)
        stage(s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // outputs fed by this slot: a = 0 -> t = s-2 (dt = 2), a = 1 -> t = s-1 (dt = 1), a = 2 -> t = s (dt = 0)
        bool live[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { const int t = s - 2 + a; live[a] = t >= 0 && t < p.T; }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8_t wf[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int dt = 2 - a;
                    u32x4_t raw = (u32x4_t){0u, 0u, 0u, 0u};
                    if (li < NROW) raw = *(const u32x4_t*)(wl + li * WROW + (((dt * 9 + tap) * CIN) + kk * 32 + kq * 8) * 2);
                    wf[a] = __builtin_bit_cast(bf16x8_t, raw);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int hp = (wid * 4 + g + dy) * HS + li + dx;
                    const bf16x8_t xf = *(const bf16x8_t*)(halo + hp * 256 + (((kk * 4 + kq) ^ (hp & 15)) << 4));
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (live[a]) acc[a][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf, acc[a][g], 0, 0, 0);
                }
            }
        }
        // output frame s-2 is complete: lanes of k groups 0 / 1 hold channels 0-3 / 4-7 of pixel (row, li)
        const int t = s - 2;
        if (t >= 0 && kq < 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int y = y0 + wid * 4 + g, x = x0 + li;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (acc[0][g][r] + bq[r]) * p.out_scale;
                bf16_t* o = p.Y + p.out_base_off + (((long long)t * p.Hop + y) * p.Wop + x) * 8 + 4 * kq;
                *(uint2*)o = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            acc[0][g] = acc[1][g];
            acc[1][g] = acc[2][g];
            acc[2][g] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
    }
}

}  // namespace

int pf_set_err(const char* m);

// does this convolution fit the narrow kernel?  (3x3x3, 128 input channels, 1..8 output channels -- the kernel multiplies
// filter rows 0..7, the weight matrix has >= 8 rows (N) -- stored with pitch 8, unit strides, no shortcut add,
// 16-pixel-aligned frame).  n_valid == 0 means "all N columns" in pf_conv_desc and is NOT narrow.
bool pf_conv_narrow_supports(const pf_conv_desc* d) {
    return d->kt == 3 && d->kh == 3 && d->kw == 3 && d->Cin == CIN && d->n_valid >= 1 && d->n_valid <= NROW && d->N >= NROW &&
           d->Cout_pitch == 8 && d->Cg == 8 &&
           d->st == 1 && d->sh == 1 && d->sw == 1 && !(d->flags & PF_GEMM_GATE_RES) && d->out_t_shift == 0 &&
           (d->in_sh == 0 || d->in_sh == 1) && (d->in_sw == 0 || d->in_sw == 1) && (d->in_st == 0 || d->in_st == 1) &&
           d->H % TS == 0 && d->W_ % TS == 0 && d->T >= 1;
}

int pf_conv_narrow_launch(const pf_conv_desc* d, hipStream_t stream) {
    PF_SET_MAX_LDS_ONCE(conv_narrow_kernel, SMEM);
    NArgs a;
    a.X = (const bf16_t*)d->X; a.Wt = (const bf16_t*)d->W; a.bias = d->bias; a.Y = (bf16_t*)d->Y;
    a.T = d->T; a.H = d->H; a.W = d->W_; a.Hp = d->Hp; a.Wp = d->Wp;
    a.in_base_off = d->in_base_off; a.out_base_off = d->out_base_off;
    a.Hop = d->Hop; a.Wop = d->Wop; a.ldw = 27 * CIN;
    a.out_scale = d->out_scale == 0.f ? 1.f : d->out_scale;
    const int grid = (d->H / TS) * (d->W_ / TS);
    hipLaunchKernelGGL(conv_narrow_kernel, dim3(grid), dim3(256), SMEM, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pf_set_err(hipGetErrorString(e));
    return 0;
}
