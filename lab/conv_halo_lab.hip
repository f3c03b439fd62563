// lab/conv_halo_lab.hip -- PROTOTYPE, NOT part of the library, NOT YET RUN ON HARDWARE (written at the end of round 3 after
// the GPU budget was spent; DESIGN.md section 8 item 1 has the budget this follows).  First task of the next round:
// `tools/gpu_r4_halo.sh` (build line below), which checks it against a naive direct convolution and times it next to
// pf_conv3d_bf16 of the shipped library on the decoder's full-resolution layer shape.
//
// What it is: a CausalConv3d (3 x 3 x 3 taps, 128 input channels, 128 filters: the decoder's full-resolution resnet convs,
// video_vae/modeling_causal_conv.py:116-146) as a DIRECT convolution with the input halo staged in LDS, instead of the
// implicit GEMM of gemm256.hip that fetches every activation once per tap.
//   * block tile 512 output pixels (a 16 x 32 patch of one output frame) x 128 filters; 8 waves = 4 (pixel rows) x 2
//     (filter halves); wave tile 128 pixels x 64 filters of v_mfma_f32_16x16x32_bf16 computing C^T (filters are the MFMA's
//     first operand), so a lane ends up with 8 consecutive filters of one pixel per pair of accumulators: the epilogue
//     runs from registers (gemm8p.hip's arrangement, including its filter-row permutation).
//   * K order: temporal tap dt -> 32-channel quarter q -> 9 spatial taps.  Per (dt, q) "stage" the 18 x 34-pixel halo of
//     the patch (x 32 channels = 38.25 KiB) is staged once; the 9 taps read it at shifted addresses.  Two halo buffers:
//     the next stage's halo is requested during the first five taps of the current stage.  The filters stream as
//     (tap, quarter) slices of 128 x 32 (8 KiB) through a 4-slot ring, three steps ahead.  LDS: 2 x 40 + 4 x 8 = 112 KiB.
//   * bank conflicts: a pixel's (or filter row's) 64 bytes hold four 16-byte chunks; chunk c of pixel x is stored at slot
//     c ^ ((x >> 2) & 3).  A 16-lane fragment group reads 16 consecutive x at 64-byte pitch: the (x & 3, (x >> 2) & 3) pairs
//     are distinct for ANY 16 consecutive x, i.e. for every tap shift.  The swizzle is applied on the DMA's source address.
//   * synchronisation: one "step" = one (stage, tap).  Two wave groups (waves 0-3 / 4-7: one wave of each SIMD) run one
//     barrier apart: a group reads fragments + issues DMA in one slot and runs its 32 MFMAs in the next, so each SIMD
//     always has one wave in its MFMA slot.  Counted s_waitcnt vmcnt(X) with X a compile-time function of the tap index
//     (the 9-tap loop is unrolled): in program order a wave issues per step [filter piece of step s+3][halo piece of the
//     next stage if tap < 5]; before step s+1 is read, everything up to the filter piece of s+1 must have landed.
//   SKEW = 0 builds the plain form (one group, one barrier per step) for bring-up.
// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../pyramid-flow_amd/csrc conv_halo_lab.hip -ldl -o conv_halo_lab
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pyflow_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_bits;

#define DEV __device__ __forceinline__
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); } while (0)

DEV void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
DEV unsigned pack2(float a, float b) {          // round to nearest even, two bf16 in one dword
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua += 0x7fffu + ((ua >> 16) & 1u);
    ub += 0x7fffu + ((ub >> 16) & 1u);
    return (ua >> 16) | (ub & 0xffff0000u);
}
DEV float bf16f(bf16_bits b) { return __uint_as_float(((unsigned)b) << 16); }

struct HArgs {
    const bf16_bits* X;      // channels-last, padded: element (frame f, padded row r, padded col c, channel ch) at
                             //   in_base_off + ((f * Hp + r) * Wp + c) * 128 + ch; (f, r, c) = (0, 0, 0) is tap (0, 0, 0) of
                             //   output pixel (t, y, x) = (0, 0, 0)
    const bf16_bits* W;      // [128 filters][27 taps][128 channels]
    const float* bias;       // [128]
    const bf16_bits* res;    // optional, indexed like Y
    bf16_bits* Y;            // element (t, y, x, n) at out_base_off + ((t * Hop + y) * Wop + x) * 128 + n
    int T, H, Wd;            // output grid; H % 16 == 0, Wd % 32 == 0
    int Hp, Wp, Hop, Wop;
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
constexpr int SMEM_T2 = 2 * HALO_BYTES + 8 * WS_BYTES;    // 144 KiB (two taps per step)
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
            const long long off = p.out_base_off + (((long long)t * p.Hop + y) * p.Wop + x) * 128 + n;
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
            const u32x4_t o = (u32x4_t){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            *(u32x4_t*)(p.Y + off) = o;
        }
    }
}

template <int SKEW>
__global__ __launch_bounds__(512, 2) void conv_halo128_t2_kernel(const HArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const s_halo = smem;                       // 2 buffers
    char* const s_w = smem + 2 * HALO_BYTES;         // 8 slots (two taps per step, slices two steps ahead)
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
        glds16(Wb + tap * 256 + q * 64 + wsrc, s_w + (step & 7) * WS_BYTES + wid * 1024);
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
    bf16x8_t fx[8], fw[2][4];        // pixel fragments shared by the two taps of a step (the second tap's are read during the first's MFMAs)

    auto read_w = [&](int u, int step) {
        const char* wb = s_w + (step & 7) * WS_BYTES + woff;
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[u][j] = *(const bf16x8_t*)(wb + j * (16 * 64));
    };
    auto read_x = [&](int stage, int k, int f0) {                     // pixel fragments f0 .. f0 + 3 of tap k
        const int dh = k / 3, dw = k - dh * 3;
        const char* hb = s_halo + (stage & 1) * HALO_BYTES;
#pragma unroll
        for (int f = f0; f < f0 + 4; ++f) {
            const int hy = 4 * wm + (f >> 1) + dh;
            fx[f] = *(const bf16x8_t*)(hb + hy * (HW * 64) + (dw == 0 ? xoff[0][f & 1] : (dw == 1 ? xoff[1][f & 1] : xoff[2][f & 1])));
        }
    };
    auto mfmas = [&](int u, int f0) {                                 // 16 MFMAs: pixel fragments f0 .. f0 + 3 x 4 filter fragments
#pragma unroll
        for (int f = f0; f < f0 + 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[u][j], fx[f], acc[f][j], 0, 0, 0);
    };

    // ---- two taps per step: the nine taps of a stage run as steps j = 0..4 = taps (0,1) (2,3) (4,5) (6,7) (8): half the
    // barriers per MFMA.  `step` below counts TAPS (ring slot = tap index & 7).  In program order a wave issues per step
    // [filter pieces of the taps of step j + 2][halo pieces of the next stage: 2, 2, 1, 0, 0]; before step j + 1 is read,
    // everything up to the filter pieces of step j + 1 must have landed: allowed outstanding = h(j-1) + f(j) + h(j).
    auto ntap = [](int j) { return j == 4 ? 1 : 2; };
    auto tap_of = [&](int stage, int j) { return stage * 9 + 2 * j; };
    auto issue_w_step = [&](int stage, int j) {              // filter pieces of step (stage, j), normalised over stage ends
        if (j >= 5) { j -= 5; ++stage; }
        if (stage >= NSTAGE) return;
        const int t0 = tap_of(stage, j);
        issue_w(t0);
        if (j < 4) issue_w(t0 + 1);
    };
    // prologue: halo of stage 0, filter pieces of steps 0 and 1; then: halo 0 + the slices of step 0 landed
#pragma unroll
    for (int k = 0; k < 5; ++k) issue_halo(0, k);
    issue_w_step(0, 0);
    issue_w_step(0, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    BAR();
    if (SKEW && grp == 1) BAR();
    for (int stage = 0; stage < NSTAGE; ++stage) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int t0 = tap_of(stage, j);
            // R slot
            issue_w_step(stage, j + 2);
            if (stage + 1 < NSTAGE) {
                if (j == 0) { issue_halo(stage + 1, 0); issue_halo(stage + 1, 1); }
                if (j == 1) { issue_halo(stage + 1, 2); issue_halo(stage + 1, 3); }
                if (j == 2) issue_halo(stage + 1, 4);
            }
            read_w(0, t0);
            read_x(stage, 2 * j, 0);
            read_x(stage, 2 * j, 4);
            if (j < 4) read_w(1, t0 + 1);
            const bool last_stages = stage >= NSTAGE - 2;    // fewer pieces are issued near the end: drain completely there
            auto counted_wait = [&]() {
                if (last_stages) { vmwait<0>(); return; }
                switch (j) {                                  // h(j-1) + f(j) + h(j), f(j) = taps of step j + 2
                    case 0: vmwait<0 + 2 + 2>(); break;
                    case 1: vmwait<2 + 2 + 2>(); break;
                    case 2: vmwait<2 + 1 + 1>(); break;
                    case 3: vmwait<1 + 2 + 0>(); break;
                    default: vmwait<0 + 2 + 0>(); break;
                }
            };
            if (SKEW && grp == 1) counted_wait();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BAR();
            // M slot
            // first tap; the second tap's pixel fragments replace the first's as soon as their MFMAs have issued and land
            // under the MFMAs that follow (reads only: no LDS hazard with the other group's R slot)
            __builtin_amdgcn_s_setprio(1);
            mfmas(0, 0);
            FENCE();
            if (j < 4) read_x(stage, 2 * j + 1, 0);
            FENCE();
            mfmas(0, 4);
            FENCE();
            if (j < 4) {
                read_x(stage, 2 * j + 1, 4);
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                FENCE();
                mfmas(1, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                FENCE();
                mfmas(1, 4);
            }
            __builtin_amdgcn_s_setprio(0);
            if (!(SKEW && grp == 1)) counted_wait();
            BAR();
        }
    }
    if (SKEW && grp == 0) BAR();

    // ---- epilogue from registers: lane (fi = pixel in fragment, fc) owns filters 64 wn + 32 hsel + 8 fc + (0..7) of pixel
    // (patch row 4 wm + (f >> 1), x = 16 (f & 1) + fi) in acc[f][2 hsel] | acc[f][2 hsel + 1]
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
        const int n = 64 * wn + 32 * hsel + 8 * fc;
        const f32x4_t b0 = *(const f32x4_t*)(p.bias + n), b1 = *(const f32x4_t*)(p.bias + n + 4);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const int y = y0 + 4 * wm + (f >> 1), x = x0 + 16 * (f & 1) + fi;
            const long long off = p.out_base_off + (((long long)t * p.Hop + y) * p.Wop + x) * 128 + n;
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
            const u32x4_t o = (u32x4_t){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            *(u32x4_t*)(p.Y + off) = o;
        }
    }
}

// ---- reference: one thread per (pixel, filter), fp32 accumulation in tap-major order
__global__ void naive_conv(const HArgs p, float* out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)p.T * p.H * p.Wd * 128;
    if (idx >= total) return;
    const int n = (int)(idx & 127);
    long long pix = idx >> 7;
    const int x = (int)(pix % p.Wd); pix /= p.Wd;
    const int y = (int)(pix % p.H); const int t = (int)(pix / p.H);
    float s = p.bias[n];
    for (int dt = 0; dt < 3; ++dt)
        for (int dh = 0; dh < 3; ++dh)
            for (int dw = 0; dw < 3; ++dw) {
                const bf16_bits* xp = p.X + p.in_base_off + (((long long)(t + dt) * p.Hp + y + dh) * p.Wp + x + dw) * 128;
                const bf16_bits* wp = p.W + ((long long)n * 27 + (dt * 3 + dh) * 3 + dw) * 128;
                for (int c = 0; c < 128; ++c) s += bf16f(xp[c]) * bf16f(wp[c]);
            }
    if (p.res) s += bf16f(p.res[p.out_base_off + (((long long)t * p.Hop + y) * p.Wop + x) * 128 + n]);
    out[idx] = s;
}

static bf16_bits f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_bits)(u >> 16); }
static float bf2f(bf16_bits b) { unsigned u = ((unsigned)b) << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

template <int SKEW>
static float run_halo(const HArgs& a, int iters) {
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)conv_halo128_kernel<SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM)); set = true; }
    const int grid = a.T * (a.H / PH) * (a.Wd / PW);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(conv_halo128_kernel<SKEW>, dim3(grid), dim3(512), SMEM, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(conv_halo128_kernel<SKEW>, dim3(grid), dim3(512), SMEM, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static float run_halo_t2(const HArgs& a, int iters) {
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)conv_halo128_t2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_T2)); set = true; }
    const int grid = a.T * (a.H / PH) * (a.Wd / PW);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(conv_halo128_t2_kernel<1>, dim3(grid), dim3(512), SMEM_T2, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(conv_halo128_t2_kernel<1>, dim3(grid), dim3(512), SMEM_T2, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    // default: the decoder's full-resolution layer on one tile-chunk window: 8 frames of 256 x 256, 128 -> 128 channels
    const int T = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 256, Wd = argc > 3 ? atoi(argv[3]) : 256;
    const int with_res = argc > 4 ? atoi(argv[4]) : 1;
    const int Hp = H + 2, Wp = Wd + 2, TF = T + 2;
    const size_t nx = (size_t)TF * Hp * Wp * 128, nw = (size_t)128 * 27 * 128, ny = (size_t)(T + 2) * Hp * Wp * 128;
    std::vector<bf16_bits> hx(nx, 0), hw(nw), hres(ny, 0);
    std::vector<float> hb(128);
    for (int f = 0; f < TF; ++f)
        for (int r = 1; r <= H; ++r)
            for (int c = 1; c <= Wd; ++c)
                for (int ch = 0; ch < 128; ++ch) hx[(((size_t)f * Hp + r) * Wp + c) * 128 + ch] = f2bf(frand());
    for (auto& v : hw) v = f2bf(frand() * 0.05f);
    for (auto& v : hb) v = frand();
    for (auto& v : hres) v = f2bf(frand());
    bf16_bits *dX, *dW, *dR, *dY, *dY2;
    float *dB, *dRef;
    CK(hipMalloc(&dX, nx * 2)); CK(hipMalloc(&dW, nw * 2)); CK(hipMalloc(&dR, ny * 2)); CK(hipMalloc(&dY, ny * 2)); CK(hipMalloc(&dY2, ny * 2));
    CK(hipMalloc(&dB, 128 * 4)); CK(hipMalloc(&dRef, (size_t)T * H * Wd * 128 * 4));
    CK(hipMemcpy(dX, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, hres.data(), ny * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), 128 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dY, 0, ny * 2)); CK(hipMemset(dY2, 0, ny * 2));
    HArgs a{};
    a.X = dX; a.W = dW; a.bias = dB; a.res = with_res ? dR : nullptr; a.Y = dY;
    a.T = T; a.H = H; a.Wd = Wd; a.Hp = Hp; a.Wp = Wp; a.Hop = Hp; a.Wop = Wp;
    a.in_base_off = 0;                                           // frame 0 = first cache slot, padded origin
    a.out_base_off = ((long long)2 * Hp * Wp + Wp + 1) * 128;    // interior pixel (0, 0) of slot 2
    {
        const long long total = (long long)T * H * Wd * 128;
        hipLaunchKernelGGL(naive_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, a, dRef);
        CK(hipDeviceSynchronize());
    }
    std::vector<float> ref((size_t)T * H * Wd * 128);
    CK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
    const double flops = 2.0 * T * H * Wd * 128.0 * 27 * 128;
    std::vector<bf16_bits> hy(ny);
    auto check = [&](const char* name, bf16_bits* dev) {
        CK(hipMemcpy(hy.data(), dev, ny * 2, hipMemcpyDeviceToHost));
        double num = 0, den = 0, mx = 0;
        for (int t = 0; t < T; ++t)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < Wd; ++x)
                    for (int n = 0; n < 128; ++n) {
                        const double r = ref[(((size_t)t * H + y) * Wd + x) * 128 + n];
                        const double v = bf2f(hy[a.out_base_off + (((size_t)t * Hp + y) * Wp + x) * 128 + n]);
                        num += (v - r) * (v - r); den += r * r; mx = fmax(mx, fabs(v - r));
                    }
        printf("%-40s rel-L2 vs the naive fp32 conv %.3e  max abs %.3e\n", name, sqrt(num / den), mx);
    };
    for (int skew = 0; skew < 2; ++skew) {
        CK(hipMemset(dY, 0, ny * 2));
        const float ms = skew ? run_halo<1>(a, 10) : run_halo<0>(a, 10);
        printf("conv_halo128_kernel<SKEW=%d>  T=%d %dx%d res=%d: %.3f ms  %.0f TFLOP/s\n", skew, T, H, Wd, with_res, ms, flops / ms / 1e9);
        check(skew ? "halo kernel, two wave groups" : "halo kernel, plain", dY);
    }
    {
        CK(hipMemset(dY, 0, ny * 2));
        const float ms = run_halo_t2(a, 10);
        printf("conv_halo128_t2_kernel<SKEW=1> (two taps per step)  T=%d %dx%d res=%d: %.3f ms  %.0f TFLOP/s\n", T, H, Wd, with_res, ms, flops / ms / 1e9);
        check("halo kernel, two taps per step", dY);
    }
    // the shipped library on the same problem
    void* lib = dlopen("pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
    if (!lib) lib = dlopen("../pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
    if (lib) {
        typedef int (*conv_fn)(const pf_conv_desc*, hipStream_t);
        conv_fn conv = (conv_fn)dlsym(lib, "pf_conv3d_bf16");
        pf_conv_desc d{};
        d.X = dX; d.W = dW; d.Y = dY2; d.bias = dB; d.res = with_res ? dR : nullptr;
        d.T = T; d.H = H; d.W_ = Wd; d.Hp = Hp; d.Wp = Wp; d.Cin = 128; d.kt = d.kh = d.kw = 3;
        d.in_base_off = 0; d.N = 128; d.n_valid = 128; d.st = d.sh = d.sw = 1; d.Cg = 128;
        d.Hop = Hp; d.Wop = Wp; d.Cout_pitch = 128; d.out_base_off = a.out_base_off;
        d.flags = with_res ? PF_GEMM_GATE_RES : 0; d.out_scale = 1.f;
        if (conv(&d, 0)) { printf("pf_conv3d_bf16 failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) conv(&d, 0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 10;
        printf("pf_conv3d_bf16 (libpyflow_hip.so): %.3f ms  %.0f TFLOP/s\n", ms, flops / ms / 1e9);
        check("library conv", dY2);
    } else {
        printf("libpyflow_hip.so not found: library timing skipped\n");
    }
    return 0;
}
