// LAB COPY in product form (NOT built into the library, NOT YET RUN ON HARDWARE): lab/conv_halo_lab.hip's kernel behind the
// library's conv descriptor.  To adopt it after tools/gpu_r4_halo.sh has passed: move this file to pyramid-flow_amd/csrc/,
// add it to SRCS in the Makefile, declare pf_conv_halo_supports / pf_conv_halo_launch in gemm.hip next to the narrow-conv
// pair, and give conv_route() a route for it in front of the gemm8p / gemm256 routes:
//     if (g_halo_enabled && pf_conv_halo_supports(d)) return -2;            // ... and in pf_conv3d_bf16: if (route == -2) return pf_conv_halo_launch(d, stream);
// (conv_fuses_gn_stats() then answers 0 for these layers until the kernel's epilogue accumulates the sums as gemm256's does.)
// See lab/conv_halo_lab.hip for the design notes and lab/conv_halo_emulate.py for the index arithmetic check.
#include "common.h"
#include "pyflow_hip.h"

namespace {

typedef unsigned short bf16_bits;
#define DEV __device__ __forceinline__
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); } while (0)
DEV unsigned pack2_rne(float a, float b) {
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}

struct HArgs {
    const bf16_bits* X; const bf16_bits* W; const float* bias; const bf16_bits* res; bf16_bits* Y;
    int T, H, Wd, Hp, Wp, Hop, Wop, Cout_pitch;
    long long in_base_off, out_base_off;
};

constexpr int PH = 16, PW = 32;                  // patch of output pixels per workgroup
constexpr int HH = PH + 2, HW = PW + 2;          // halo
constexpr int HALO_PIX = HH * HW;                // 612
constexpr int HALO_CHUNKS = HALO_PIX * 4;        // 16-byte chunks of one (frame, quarter) halo: 2448
constexpr int HALO_PIECES = (HALO_CHUNKS + 63) / 64;      // 39 one-KiB pieces (the last one partial)
constexpr int HALO_BYTES = 40 * 1024;            // buffer size (>= 39 KiB: the partial piece spills into the pad)
constexpr int WS_BYTES = 128 * 64;               // one (tap, quarter) filter slice
constexpr int SMEM = 2 * HALO_BYTES + 4 * WS_BYTES;       // 112 KiB
constexpr int NSTAGE = 12, NSTEP = NSTAGE * 9;   // (dt, quarter) stages x 9 spatial taps

// pieces a wave issues per step, in program order: [filter piece of step s + 3][halo piece k * 8 + wid of the next stage, k < 5]
__host__ __device__ constexpr int halo_issued(int k) { return (k >= 0 && k < 5) ? 1 : 0; }
// outstanding pieces allowed when the filter piece of step s + 1 must have landed, at the end of step s (tap k):
// issued after it: the halo piece of step s - 2, both pieces of step s - 1, both pieces of step s
__host__ __device__ constexpr int wait_count(int k) { return halo_issued(k - 2) + 1 + halo_issued(k - 1) + 1 + halo_issued(k); }

template <int N>
DEV void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SKEW>
__global__ __launch_bounds__(512, 2) void conv_halo128_kernel(const HArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const s_halo = smem;                       // 2 buffers
    char* const s_w = smem + 2 * HALO_BYTES;         // 4 slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = SKEW ? (wid >> 2) : 0;           // wave group (SKEW): group 1 runs one barrier behind group 0
    const int wm = wid & 3, wn = wid >> 2;           // wave tile: patch rows 4 wm .. +4 (128 pixels), filters 64 wn .. +64
    // (with SKEW the two waves of a SIMD are wid and wid + 4 = the two filter halves of the same pixel rows)

    // ---- tile
    const int tiles_x = p.Wd / PW, tiles_y = p.H / PH;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; const int t = tile / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW;

    // ---- DMA geometry of the halo: piece pc covers LDS chunks [64 pc, 64 pc + 64); chunk g = 4 * halo pixel + slot
    // this wave's pieces of a stage: pc = k * 8 + wid for k = 0..4 (wave 7's k = 4 piece would be pc 39: it re-issues 38)
    unsigned hsrc[5];                                // byte offset of this lane's source chunk relative to the halo origin
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int pc = k * 8 + wid;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        int g = pc * 64 + lane;
        g = g < HALO_CHUNKS ? g : HALO_CHUNKS - 1;   // the partial piece: lanes past the end re-read the last chunk (lands in the pad)
        const int hp = g >> 2, slot = g & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int c = slot ^ ((hx >> 2) & 3);        // the chunk that belongs in this slot
        hsrc[k] = (unsigned)((hy * p.Wp + hx) * 256 + c * 16);
    }
    // filter slice: 8 pieces, wave wid owns LDS rows 16 wid .. +16; LDS row r = 64 wn' + 16 j + i holds filter
    // 64 wn' + 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) + (i & 3)      (gemm8p.hip's permutation: epilogue from registers)
    unsigned wsrc;
    {
        const int r = 16 * wid + (lane >> 2), slot = lane & 3;
        const int wn_ = r >> 6, j = (r >> 4) & 3, i = r & 15;
        const int n = 64 * wn_ + 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
        const int c = slot ^ ((r >> 2) & 3);
        wsrc = (unsigned)(n * (27 * 256) + c * 16);
    }
    const char* const Xb = (const char*)(p.X + p.in_base_off) + ((long long)t * p.Hp * p.Wp + (long long)y0 * p.Wp + x0) * 256;
    const char* const Wb = (const char*)p.W;

    auto issue_halo = [&](int stage, int k) {        // piece k of this wave for (dt, q) = (stage / 4, stage % 4)
        const int dt = stage >> 2, q = stage & 3;
        const char* src = Xb + (long long)dt * p.Hp * p.Wp * 256 + q * 64;
        int pc = k * 8 + wid;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        glds16(src + hsrc[k], s_halo + (stage & 1) * HALO_BYTES + pc * 1024);
    };
    auto issue_w = [&](int step) {                   // this wave's piece of the filter slice of `step`
        const int stage = step / 9, k = step - stage * 9;
        const int dt = stage >> 2, q = stage & 3;
        const int tap = dt * 9 + k;
        glds16(Wb + tap * 256 + q * 64 + wsrc, s_w + (step & 3) * WS_BYTES + wid * 1024);
    };

    // ---- fragment read offsets (bytes)
    const int fi = lane & 15, fc = lane >> 4;
    unsigned xoff[3][2];                             // [dw][x half]: column part of the halo address incl. the swizzled chunk
#pragma unroll
    for (int dw = 0; dw < 3; ++dw)
#pragma unroll
        for (int xh = 0; xh < 2; ++xh) {
            const int hx = 16 * xh + fi + dw;
            xoff[dw][xh] = (unsigned)(hx * 64 + ((fc ^ ((hx >> 2) & 3)) << 4));
        }
    const unsigned woff = (unsigned)((64 * wn + fi) * 64 + ((fc ^ ((fi >> 2) & 3)) << 4));   // + 16 j rows

    f32x4_t acc[8][4];                               // [pixel fragment f: patch row 4 wm + (f >> 1), x half f & 1][filter fragment j]
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[f][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t fx[8], fw[4];

    auto read_frags = [&](int stage, int step, int k) {
        const int dh = k / 3, dw = k - dh * 3;
        const char* hb = s_halo + (stage & 1) * HALO_BYTES;
        const char* wb = s_w + (step & 3) * WS_BYTES + woff;
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *(const bf16x8_t*)(wb + j * (16 * 64));
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int hy = 4 * wm + (f >> 1) + dh;
            fx[f] = *(const bf16x8_t*)(hb + hy * (HW * 64) + (dw == 0 ? xoff[0][f & 1] : (dw == 1 ? xoff[1][f & 1] : xoff[2][f & 1])));
        }
    };
    auto mfmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[f], acc[f][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: halo of stage 0, filter slices of steps 0..2; then: halo 0 + slice 0 landed
#pragma unroll
    for (int k = 0; k < 5; ++k) issue_halo(0, k);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    BAR();
    if (SKEW && grp == 1) BAR();                     // group 1 runs one barrier behind

    // ---- main loop.  Slot structure per step and group: [R: issue DMA, read fragments] barrier [M: 32 MFMAs] barrier.
    // SKEW: group 0's R(s) coincides with group 1's M(s - 1).  The counted wait for the NEXT step's operands sits at the
    // end of the slot that precedes the barrier before group 0's R(s + 1): M(s) for group 0, R(s) ... wait, group 1 is one
    // slot behind, so for group 1 that is the end of its R(s) -- see the header for the count.
    for (int stage = 0; stage < NSTAGE; ++stage) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int step = stage * 9 + k;
            // R slot
            if (step + 3 < NSTEP) issue_w(step + 3);
            if (k < 5 && stage + 1 < NSTAGE) issue_halo(stage + 1, k);
            read_frags(stage, step, k);
            const bool last_stage = stage == NSTAGE - 1;
            auto counted_wait = [&]() {
                // everything up to the filter piece of step + 1 (and with it the whole halo of the next stage) has landed;
                // in the last stage fewer pieces are issued per step: drain completely there (9 steps of 108)
                if (last_stage) { vmwait<0>(); return; }
                switch (k) {                          // k is the unrolled loop index: one case survives
                    case 0: vmwait<wait_count(0)>(); break;
                    case 1: vmwait<wait_count(1)>(); break;
                    case 2: vmwait<wait_count(2)>(); break;
                    case 3: vmwait<wait_count(3)>(); break;
                    case 4: vmwait<wait_count(4)>(); break;
                    case 5: vmwait<wait_count(5)>(); break;
                    case 6: vmwait<wait_count(6)>(); break;
                    case 7: vmwait<wait_count(7)>(); break;
                    default: vmwait<wait_count(8)>(); break;
                }
            };
            if (SKEW && grp == 1) counted_wait();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BAR();
            // M slot
            mfmas();
            if (!(SKEW && grp == 1)) counted_wait();
            BAR();
        }
    }
    if (SKEW && grp == 0) BAR();                     // matches group 1's extra barrier

    // ---- epilogue from registers: lane (fi = pixel in fragment, fc) owns filters 64 wn + 32 hsel + 8 fc + (0..7) of pixel
    // (patch row 4 wm + (f >> 1), x = 16 (f & 1) + fi) in acc[f][2 hsel] | acc[f][2 hsel + 1]
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
        const int n = 64 * wn + 32 * hsel + 8 * fc;
        const f32x4_t b0 = *(const f32x4_t*)(p.bias + n), b1 = *(const f32x4_t*)(p.bias + n + 4);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int y = y0 + 4 * wm + (f >> 1), x = x0 + 16 * (f & 1) + fi;
            const long long off = p.out_base_off + (((long long)t * p.Hop + y) * p.Wop + x) * p.Cout_pitch + n;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[f][2 * hsel][r] + b0[r]; v[4 + r] = acc[f][2 * hsel + 1][r] + b1[r]; }
            if (p.res) {
                const u32x4_t rr = *(const u32x4_t*)(p.res + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(rr[e] << 16);
                    v[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u);
                }
            }
            const u32x4_t o = (u32x4_t){pack2_rne(v[0], v[1]), pack2_rne(v[2], v[3]), pack2_rne(v[4], v[5]), pack2_rne(v[6], v[7])};
            *(u32x4_t*)(p.Y + off) = o;
        }
    }
}


}  // namespace

// 3 x 3 x 3 taps, 128 input channels at pitch 128, 128 filters all valid, plain output map, unit strides, patches of 16 x 32
bool pf_conv_halo_supports(const pf_conv_desc* d) {
    if (d->kt != 3 || d->kh != 3 || d->kw != 3 || d->Cin != 128 || d->N != 128) return false;
    if ((d->n_valid > 0 ? d->n_valid : d->N) != 128 || d->Cg != 128 || d->Cout_pitch % 8) return false;
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->out_t_shift != 0) return false;
    if ((d->in_sh > 1) || (d->in_sw > 1) || (d->in_st > 1)) return false;
    if (d->H % 16 || d->W_ % 32 || d->T <= 0) return false;
    if ((d->flags & ~PF_GEMM_GATE_RES) || (d->out_scale != 0.f && d->out_scale != 1.f)) return false;
    if ((d->flags & PF_GEMM_GATE_RES) && !d->res) return false;
    return true;
}

int pf_conv_halo_launch(const pf_conv_desc* d, hipStream_t stream) {
    PF_SET_MAX_LDS_ONCE((conv_halo128_kernel<1>), SMEM);
    HArgs a{};
    a.X = (const bf16_bits*)d->X; a.W = (const bf16_bits*)d->W; a.bias = d->bias;
    a.res = (d->flags & PF_GEMM_GATE_RES) ? (const bf16_bits*)d->res : nullptr;
    a.Y = (bf16_bits*)d->Y;
    a.T = d->T; a.H = d->H; a.Wd = d->W_; a.Hp = d->Hp; a.Wp = d->Wp; a.Hop = d->Hop; a.Wop = d->Wop;
    a.Cout_pitch = d->Cout_pitch;
    a.in_base_off = d->in_base_off; a.out_base_off = d->out_base_off;
    const int grid = d->T * (d->H / PH) * (d->W_ / PW);
    hipLaunchKernelGGL(conv_halo128_kernel<1>, dim3(grid), dim3(512), SMEM, stream, a);
    return 0;
}
