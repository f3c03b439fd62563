// LDS-halo direct convolution for the decoder's full-resolution CausalConv3d layers (round 4).
//
// Replaces, for 3 x 3 x 3 taps, 128 filters and 128 or 256 input channels (the conv1 / conv2 of up_blocks.3's resnets:
// video_vae/modeling_causal_conv.py:116-146, modeling_resnet.py:115-150), the implicit GEMM of gemm256.hip, which fetches
// every activation once per tap from L2 / the fabric (profiles/r03_pmc_vae_tile.json: 3.8 GB per launch for a layer whose
// algorithmic bytes are 1.08 GB) and spends its time ISSUING those LDS-DMA pieces.  Here the input halo of a patch is staged
// ONCE per (frame, 32-channel quarter) and the nine spatial taps read it at shifted LDS addresses:
//   * block tile 512 output pixels (a 16 x 32 patch of one output frame) x 128 filters; 8 waves = 4 (pixel rows) x 2
//     (filter halves); wave tile 128 pixels x 64 filters of v_mfma_f32_16x16x32_bf16 computing C^T (filters are the MFMA's
//     first operand), so a lane ends up with 8 consecutive filters of one pixel per pair of accumulators: the epilogue
//     (bias, shortcut add, bf16, GroupNorm statistics) runs from registers -- gemm8p.hip's arrangement and row permutation.
//   * K order: temporal tap dt -> 32-channel quarter q -> 9 spatial taps.  Per (dt, q) "stage" the 18 x 34-pixel halo of
//     the patch (x 32 channels = 38.25 KiB) is staged once.  Two halo buffers: the next stage's halo is requested during
//     the first five taps of the current stage.  The filters stream as (tap, quarter) slices of 128 x 32 (8 KiB) through
//     a 4-slot ring, three steps ahead.  LDS: 2 x 40 + 4 x 8 = 112 KiB, one workgroup per CU.
//   * bank conflicts: a pixel's (or filter row's) 64 bytes hold four 16-byte chunks; chunk c of pixel x is stored at slot
//     c ^ ((x >> 2) & 3).  A 16-lane fragment group reads 16 consecutive x at 64-byte pitch: the (x & 3, (x >> 2) & 3) pairs
//     are distinct for ANY 16 consecutive x, i.e. for every tap shift.  The swizzle is applied on the DMA's source address.
//   * synchronisation: one "step" = one (stage, tap).  Two wave groups (waves 0-3 / 4-7: one wave of each SIMD) run one
//     barrier apart: a group reads fragments + issues DMA in one slot and runs its 32 MFMAs in the next, so each SIMD
//     always has one wave in its MFMA slot.  Counted s_waitcnt vmcnt(X) with X a compile-time function of the tap index.
// Measured on MI355X (profiles/r04_conv_halo_lab.log, 8 frames of 256 x 256, 128 -> 128 channels + shortcut add):
// 0.441 ms = 1 051 TFLOP/s where the implicit GEMM runs 0.627 ms = 740 TFLOP/s; same result to the last bit of the
// rel-L2 (1.66e-3 against a naive fp32 convolution).  Index arithmetic emulated on the CPU in lab/conv_halo_emulate.py.
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
    int cin;                 // input channels = pitch of X and of a filter tap (128 / 256 / 512)
    int st, sh, sw, Cg, t_shift;   // output map of the upsamplers: column n = g Cg + c, g = (pt sh + ph) sw + pw -> pixel-shuffle /
                             //   depth-to-time placement (modeling_resnet.py:609-617, 716-729); 1, 1, 1, N, 0 = plain
    long long in_base_off, out_base_off;
    double* gn_stats;        // [frame][gn_C][2] (sum, sum of squares) of the stored output, or nullptr (pf_conv_desc.gn_stats)
    int gn_C;
};

constexpr int PH = 16, PW = 32;                  // patch of output pixels per workgroup
constexpr int HH = PH + 2, HW = PW + 2;          // halo
constexpr int HALO_PIX = HH * HW;                // 612
constexpr int HALO_CHUNKS = HALO_PIX * 4;        // 16-byte chunks of one (frame, quarter) halo: 2448
constexpr int HALO_PIECES = (HALO_CHUNKS + 63) / 64;      // 39 one-KiB pieces (the last one partial)
constexpr int HALO_BYTES = 40 * 1024;            // buffer size (>= 39 KiB: the partial piece spills into the pad)
constexpr int WS_BYTES = 128 * 64;               // one (tap, quarter) filter slice
constexpr int SMEM = 2 * HALO_BYTES + 4 * WS_BYTES;       // 112 KiB

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
    const int nblk = blockIdx.y;                     // block of 128 filters (N = 128 gridDim.y)
    // XCD-aware order (block b runs on XCD b % 8, each XCD has its own L2): an output frame's patch reads the input frames
    // t, t + 1, t + 2, so the SAME patch of consecutive frames shares two of its three halos -- XCD x owns the patches
    // x, x + 8, ... and its workgroups walk a patch's frames consecutively, which turns the temporal re-fetch into L2 hits
    // (profiles/r04_pmc_vae_tile.json before this order: 5.6 bytes fetched per byte written on the 128 -> 128 layers)
    const int npatch = tiles_x * tiles_y;
    if ((npatch & 7) == 0) {
        const int xcd = tile & 7, i = tile >> 3;
        const int pl = i / p.T;
        tile = (i - pl * p.T) * npatch + pl * 8 + xcd;            // (frame, patch) in the plain order below
    }
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y; const int t = tile / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW;
    const int nq = p.cin >> 5, lq = p.cin == 512 ? 4 : (p.cin == 256 ? 3 : 2);      // 32-channel quarters per temporal tap: 4 / 8 / 16 = 1 << lq
    const int NSTAGE = 3 * nq, NSTEP = NSTAGE * 9;   // (dt, quarter) stages x 9 spatial taps
    const int pix = p.cin * 2;                       // bytes per input pixel / per filter tap

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
        hsrc[k] = (unsigned)((hy * p.Wp + hx) * pix + c * 16);
    }
    // filter slice: 8 pieces, wave wid owns LDS rows 16 wid .. +16; LDS row r = 64 wn' + 16 j + i holds filter
    // 64 wn' + 32 (j >> 1) + 8 (i >> 2) + 4 (j & 1) + (i & 3)      (gemm8p.hip's permutation: epilogue from registers)
    unsigned wsrc;
    {
        const int r = 16 * wid + (lane >> 2), slot = lane & 3;
        const int wn_ = r >> 6, j = (r >> 4) & 3, i = r & 15;
        const int n = 64 * wn_ + 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
        const int c = slot ^ ((r >> 2) & 3);
        wsrc = (unsigned)(n * (27 * pix) + c * 16);
    }
    const char* const Xb = (const char*)(p.X + p.in_base_off) + ((long long)t * p.Hp * p.Wp + (long long)y0 * p.Wp + x0) * pix;
    const char* const Wb = (const char*)p.W + (long long)nblk * 128 * 27 * pix;      // this block's 128 filter rows

    auto issue_halo = [&](int stage, int k) {        // piece k of this wave for (dt, q) = (stage / nq, stage % nq)
        const int dt = stage >> lq, q = stage & (nq - 1);
        const char* src = Xb + (long long)dt * p.Hp * p.Wp * pix + q * 64;
        int pc = k * 8 + wid;
        pc = pc < HALO_PIECES ? pc : HALO_PIECES - 1;
        glds16(src + hsrc[k], s_halo + (stage & 1) * HALO_BYTES + pc * 1024);
    };
    auto issue_w = [&](int step) {                   // this wave's piece of the filter slice of `step`
        const int stage = step / 9, k = step - stage * 9;
        const int dt = stage >> lq, q = stage & (nq - 1);
        const int tap = dt * 9 + k;
        glds16(Wb + tap * pix + q * 64 + wsrc, s_w + (step & 3) * WS_BYTES + wid * 1024);
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
#ifdef PF_HALO_PRIO_FLIPS          // (lab: the per-slot priority flips of rounds 4-5)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[f], acc[f][j], 0, 0, 0);
#ifdef PF_HALO_PRIO_FLIPS
        __builtin_amdgcn_s_setprio(0);
#endif
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
#ifndef PF_HALO_PRIO_FLIPS
    // static priority for the second-dispatched wave group, no per-slot flips (as gemm8p.hip since round 6)
    if (SKEW && grp == 1) __builtin_amdgcn_s_setprio(1);
#endif

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
    // (patch row 4 wm + (f >> 1), x = 16 (f & 1) + fi) in acc[f][2 hsel] | acc[f][2 hsel + 1].
    // GroupNorm statistics (p.gn_stats, the [frame][channel][2] arena pf_gn_apply reads): sums of the values AS STORED (after
    // the bf16 rounding: what a pf_gn_stats pass would read back) over the lane's 8 pixels, over the 16 lanes that own the
    // same filters (4 shuffles), over the four pixel-row waves through LDS (free after the main loop), then ONE double
    // atomic per (filter, statistic) and 512-pixel tile.
    const bool do_stats = p.gn_stats != nullptr;       // workgroup-uniform
    float* const s_stat = (float*)smem;                // [wave 0..7][hsel][fc][e][2]: 8 x 64 x 2 floats = 4 KiB
    // Every load of the epilogue (bias, the shortcut's pieces) is issued before the first store, and the bf16 results wait in
    // registers (in place of the accumulators they came from) until both column halves are done: `res` may alias `Y` as far as
    // the compiler knows, so in the one-pass form (load, add, store per piece: rounds 4) it kept every shortcut load behind the
    // previous store and guarded it with s_waitcnt vmcnt(0) -- 16 dependent load + store round trips per tile.
    u32x4_t outp[2][8];
    // COALESCED EPILOGUE TRAFFIC (round 6, as in gemm8p.hip).  In the accumulator layout the four 16-byte pieces of a pixel's 64
    // bytes sit in lanes {x, x + 16, x + 32, x + 48}: adjacent lanes address different pixels (a channel pitch apart), every
    // store / shortcut load was 64 separate 16-byte transactions (~270-530 cycles per instruction and wave in the GEMM).  The
    // memory side uses the layout lane L = (pixel L >> 2, piece L & 3) -- a quad of lanes covers 64 contiguous bytes -- and
    // the registers move between the two layouts through the LDS crossbar (ds_bpermute_b32: no LDS memory, no barrier).
    const int fi2 = lane >> 2, fc2 = lane & 3;
    const int to_mem = (((lane & 3) << 4) | (lane >> 2)) << 2;        // memory-layout lane L reads accumulator-layout lane (L & 3) 16 + (L >> 2)
    const int to_acc = (((lane & 15) << 2) | (lane >> 4)) << 2;       // accumulator-layout lane (fc, fi) reads memory-layout lane 4 fi + fc
    auto xperm = [&](const u32x4_t v, int src) {
        u32x4_t o;
#pragma unroll
        for (int d_ = 0; d_ < 4; ++d_) o[d_] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)v[d_]);
        return o;
    };
    auto out_off = [&](int hsel, int f, bool& ok, int fi, int fc) -> long long {
        const int n = 128 * nblk + 64 * wn + 32 * hsel + 8 * fc;
        // output placement of this lane's 8 filters (one group: Cg % 8 == 0): frame t st + pt (+ t_shift; < 0 = dropped),
        // pixel (y sh + ph, x sw + pw), channel cc
        const int gg = n / p.Cg, cc = n - gg * p.Cg;
        const int shw = p.sh * p.sw;
        const int pt = gg / shw, g2 = gg - pt * shw;
        const int ph = g2 / p.sw, pw = g2 - ph * p.sw;
        const int tf = t * p.st + pt + p.t_shift;
        const int y = y0 + 4 * wm + (f >> 1), x = x0 + 16 * (f & 1) + fi;
        ok = tf >= 0;
        return p.out_base_off + (((long long)(tf < 0 ? 0 : tf) * p.Hop + (y * p.sh + ph)) * p.Wop + (x * p.sw + pw)) * p.Cout_pitch + cc;
    };
#pragma unroll
    for (int hsel = 0; hsel < 2; ++hsel) {
        const int n = 128 * nblk + 64 * wn + 32 * hsel + 8 * fc;
        const f32x4_t b0 = *(const f32x4_t*)(p.bias + n), b1 = *(const f32x4_t*)(p.bias + n + 4);
        u32x4_t rr[8];
        if (p.res) {
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                bool ok;
                rr[f] = *(const u32x4_t*)(p.res + out_off(hsel, f, ok, fi2, fc2));
            }
#pragma unroll
            for (int f = 0; f < 8; ++f) rr[f] = xperm(rr[f], to_acc);
        }
        float gs[8], gq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[f][2 * hsel][r] + b0[r]; v[4 + r] = acc[f][2 * hsel + 1][r] + b1[r]; }
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(rr[f][e] << 16);
                    v[2 * e + 1] += __uint_as_float(rr[f][e] & 0xffff0000u);
                }
            }
            const u32x4_t o = (u32x4_t){pack2_rne(v[0], v[1]), pack2_rne(v[2], v[3]), pack2_rne(v[4], v[5]), pack2_rne(v[6], v[7])};
            outp[hsel][f] = o;
            if (do_stats) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(o[e] << 16), hi = __uint_as_float(o[e] & 0xffff0000u);
                    gs[2 * e] += lo; gq[2 * e] += lo * lo;
                    gs[2 * e + 1] += hi; gq[2 * e + 1] += hi * hi;
                }
            }
        }
        if (do_stats) {
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gs[e] += __shfl_xor(gs[e], o_, 64);
                    gq[e] += __shfl_xor(gq[e], o_, 64);
                }
            if (fi == 0) {
                float* d_ = s_stat + ((wid * 2 + hsel) * 4 + fc) * 16;
#pragma unroll
                for (int e = 0; e < 8; ++e) { d_[2 * e] = gs[e]; d_[2 * e + 1] = gq[e]; }
            }
        }
    }
    {       // stores, pipelined by one item against the lane permutation of the next; the two halves of a pixel back to back
        u32x4_t cur = xperm(outp[0][0], to_mem);
#pragma unroll
        for (int i = 0; i < 16; ++i) {               // i = 2 f + hsel
            u32x4_t nxt = cur;
            if (i + 1 < 16) nxt = xperm(outp[(i + 1) & 1][(i + 1) >> 1], to_mem);
            bool ok;
            const long long off = out_off(i & 1, i >> 1, ok, fi2, fc2);
            if (ok) *(u32x4_t*)(p.Y + off) = cur;
            cur = nxt;
        }
    }
    if (do_stats) {
        BAR();
        if (tid < 256) {                              // (filter c, statistic k): add the four pixel-row waves of filter half wn_
            const int c = tid >> 1, k = tid & 1;
            const int wn_ = c >> 6, hsel = (c >> 5) & 1, fc_ = (c >> 3) & 3, e = c & 7;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += s_stat[(((wn_ * 4 + w) * 2 + hsel) * 4 + fc_) * 16 + 2 * e + k];
            atomicAdd(p.gn_stats + ((long long)t * p.gn_C + 128 * nblk + c) * 2 + k, (double)sum);
        }
    }
}


}  // namespace

// 3 x 3 x 3 taps, 128 / 256 / 512 input channels (= the input pitch), N = a multiple of 128 filters, all valid (one 128-filter
// block per blockIdx.y: the input halo is staged once per block), plain output map, unit strides, frames that are whole
// numbers of 16 x 32 patches; bias required (the decoder's convs all have one).  `wide` = the caller accepts N > 128.
bool pf_conv_halo_supports(const pf_conv_desc* d, bool wide) {
    if (d->kt != 3 || d->kh != 3 || d->kw != 3 || (d->Cin != 128 && d->Cin != 256 && d->Cin != 512) || !d->bias) return false;
    if (d->N % 128 || d->N <= 0 || (!wide && (d->N != 128 || d->Cin == 512))) return false;
    if ((d->n_valid > 0 ? d->n_valid : d->N) != d->N || d->Cout_pitch % 8 || d->Cg <= 0 || d->Cg % 8) return false;
    const bool plain = d->st == 1 && d->sh == 1 && d->sw == 1 && d->out_t_shift == 0;
    if (plain ? d->Cg != d->N : (!wide || d->N != d->st * d->sh * d->sw * d->Cg || (d->flags & PF_GEMM_GATE_RES))) return false;
    // a launch of at least half a round of the chip (the small 32 x 32 / 64 x 64-pixel layers stay with the implicit GEMM)
    if (d->N > 128 && (long long)d->T * (d->H / 16) * (d->W_ / 32) * (d->N / 128) < 128) return false;
    if (d->st < 1 || d->sh < 1 || d->sw < 1) return false;
    if ((d->in_sh > 1) || (d->in_sw > 1) || (d->in_st > 1)) return false;
    if (d->H % 16 || d->W_ % 32 || d->T <= 0) return false;
    if ((d->flags & ~PF_GEMM_GATE_RES) || (d->out_scale != 0.f && d->out_scale != 1.f)) return false;
    if ((d->flags & PF_GEMM_GATE_RES) && !d->res) return false;
    return true;
}

// gn_stats: the caller's arena when conv_fuses_gn_stats() (gemm.hip) said this launch accumulates it, else nullptr
int pf_conv_halo_launch(const pf_conv_desc* d, double* gn_stats, int gn_C, hipStream_t stream) {
    PF_SET_MAX_LDS_ONCE((conv_halo128_kernel<1>), SMEM);
    HArgs a{};
    a.X = (const bf16_bits*)d->X; a.W = (const bf16_bits*)d->W; a.bias = d->bias;
    a.res = (d->flags & PF_GEMM_GATE_RES) ? (const bf16_bits*)d->res : nullptr;
    a.Y = (bf16_bits*)d->Y;
    a.T = d->T; a.H = d->H; a.Wd = d->W_; a.Hp = d->Hp; a.Wp = d->Wp; a.Hop = d->Hop; a.Wop = d->Wop;
    a.Cout_pitch = d->Cout_pitch;
    a.cin = d->Cin;
    a.st = d->st; a.sh = d->sh; a.sw = d->sw; a.Cg = d->Cg; a.t_shift = d->out_t_shift;
    a.in_base_off = d->in_base_off; a.out_base_off = d->out_base_off;
    a.gn_stats = gn_stats; a.gn_C = gn_C;
    const int grid = d->T * (d->H / PH) * (d->W_ / PW);
    hipLaunchKernelGGL(conv_halo128_kernel<1>, dim3(grid, d->N / 128), dim3(512), SMEM, stream, a);
    return 0;
}
