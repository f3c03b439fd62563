// bf16 MFMA GEMM, 256 x BN block tile (BN = 128 / 192 / 256), 8 waves in a ping-pong schedule.
//
// Same contract as gemm.hip (C = epi(A . W^T), optional implicit-GEMM conv addressing) for the large
// DiT projections (flux_block.py:756-758, 816-835, 868-872, 914-942) and the VAE CausalConv3d
// (modeling_causal_conv.py:116-146).  Why a second kernel: the 128x128 / one-barrier-per-K-step structure of
// gemm.hip tops out near 0.85 PFLOP/s -- it needs 64 B/clk/CU of L2->LDS traffic at full MFMA rate (the vector
// memory path's limit) and drains every LDS-DMA at every barrier.  Here:
//   * block tile 256 x BN, BK = 64; BN = 192 divides every miniFLUX width (d = 1920 = 10 x 192), BN = 256 the
//     VAE filter counts; L2->LDS traffic per MFMA drops to 58 % / 50 % of the 128x128 tile's;
//   * 8 waves = 2 groups (M halves) x 4 waves (2 x 2, wave tile 64 x BN/2).  A SIMD hosts one wave of each group;
//     the groups run one "slot" apart: while group 0 issues MFMAs (s_setprio 1) for K-half h of tile t, group 1
//     issues its ds_read_b128 fragment reads and its share of the LDS-DMA prefetch, then they swap (one s_barrier
//     per slot).  The matrix pipe of every SIMD always has exactly one wave feeding it;
//   * LDS-DMA (global_load_lds_dwordx4) never drains to zero in the main loop: the A halves are triple-buffered
//     and B double-buffered (3*32 + 2*BN/8 KiB <= 160 KiB); per K-tile a wave issues NT = BN/64 pieces of B(t+1)
//     in its first load slot and 4 pieces of A(t+2) in its second, and waits ONCE per tile with vmcnt(4), i.e.
//     with the four A pieces still in flight across the barriers;
//   * epilogue per wave through a private LDS strip (no block barrier): fp32 accumulators -> whole 16-byte bf16
//     pieces with bias / GELU-tanh / gate*x+res / pixel-shuffle store mapping as in gemm.hip.
// Tile order: XCD-contiguous chunks (xcd_remap), inside a chunk groups of 4 M-tiles x all N-tiles, M fastest, so
// the 32 tiles resident on one XCD share 4 A panels and 8 W panels in that XCD's L2.
#include "common.h"
#include "pyflow_hip.h"
#include "gemm_args.h"

using namespace pfgemm;

namespace {

constexpr int BM = 256, BK = 64;
constexpr int A_HALF = 128 * BK * 2;      // 16 KiB: 128 rows x 128 B
constexpr int A_STAGE = 2 * A_HALF;       // 32 KiB
constexpr int A_BYTES = 3 * A_STAGE;      // 96 KiB
constexpr int GROUP_M = 4;

template <int BN>
struct Cfg {
    static constexpr int NT = BN / 64;                 // 32-column MFMA tiles per wave (wave tile 64 x BN/2)
    static constexpr int B_STAGE = BN * BK * 2;
    static constexpr int SMEM = A_BYTES + 2 * B_STAGE;
    static constexpr int SMEM3 = A_BYTES + 3 * B_STAGE;   // variant 4: B triple-buffered as well (fits for BN = 128 only)
    static constexpr int EPI_STRIDE = BN / 2 + 4;      // floats per staged row
    static constexpr int EPI_BYTES = 32 * EPI_STRIDE * 4;
    static_assert(8 * EPI_BYTES <= SMEM, "epilogue strips must fit");
};

#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PF_BARRIER()                         \
    do {                                     \
        PF_SCHED_FENCE();                    \
        __builtin_amdgcn_s_barrier();        \
        PF_SCHED_FENCE();                    \
    } while (0)

template <int BN, bool CONV, int V>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const Args p) {
    using C_ = Cfg<BN>;
    constexpr int NT = C_::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sB = smem + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wid >> 2, w4 = wid & 3, wm = w4 >> 1, wn = w4 & 1;

    // ---- tile mapping
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;      // N tail: clamped loads, masked stores
    const int TM = tiles_m * p.batch;
    const int nwg = TM * tiles_n;
    int t = xcd_remap(blockIdx.x, nwg);
    const int GM = p.group_m > 0 ? p.group_m : GROUP_M;
    const int group_sz = GM * tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GM;
    const int gm = min(TM - first_m, GM);
    const int r_in = t - grp * group_sz;
    const int tn = r_in / gm;
    const int tmm = first_m + (r_in - tn * gm);
    const int b = tmm / tiles_m, tm = tmm - b * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA sources.  A: group g loads its own half, wave w4 owns 1-KiB pieces ih = 4 w4 + j (rows 8 ih ..
    //      8 ih + 7 of the half).  B: pieces ib = g*BN/16 + w4*NT + j (rows 8 ib .. + 7 of the BN filter rows).
    //      lane -> row + lane/8, LDS chunk lane%8 holds source chunk (lane%8) ^ ((row >> 1) & 7).
    const bf16_t* A = p.A + (long long)b * p.sA;
    const bf16_t* asrc[4];
    const bf16_t* bsrc[NT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ih = w4 * 4 + j;
        const int c = (lane & 7) ^ (((ih & 1) << 2) + (lane >> 4));
        int m = m0 + g * 128 + 8 * ih + (lane >> 3);
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.cg.H * p.cg.W;
            const int tt = m / hw, rem = m - tt * hw;
            const int hh = rem / p.cg.W, ww = rem - hh * p.cg.W;
            asrc[j] = A + p.cg.base_off + (((long long)tt * p.cg.st * p.cg.Hp + hh * p.cg.sh) * p.cg.Wp + ww * p.cg.sw) * p.cg.Cin + c * 8;
        } else {
            asrc[j] = A + (long long)m * p.lda + c * 8;
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ib = g * (BN / 16) + w4 * NT + j;
        const int c = (lane & 7) ^ (((ib & 1) << 2) + (lane >> 4));
        int nrow = n0 + 8 * ib + (lane >> 3);
        nrow = nrow < p.N ? nrow : p.N - 1;
        bsrc[j] = p.W + (long long)nrow * p.ldw + c * 8;
    }
    const int nk = p.K / BK;

    auto issueA = [&](int kt, int stage) {
        long long aoff;
        if (CONV) {
            const int k0 = kt * BK;
            const int tap = k0 / p.cg.Cin, c0 = k0 - tap * p.cg.Cin;
            const int khw = p.cg.kh * p.cg.kw;
            const int dt = tap / khw, r2 = tap - dt * khw;
            const int dh = r2 / p.cg.kw, dw = r2 - dh * p.cg.kw;
            aoff = (((long long)dt * p.cg.Hp + dh) * p.cg.Wp + dw) * p.cg.Cin + c0;
        } else {
            aoff = (long long)kt * BK;
        }
        char* base = sA + stage * A_STAGE + g * A_HALF + w4 * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(asrc[j] + aoff, base + j * 1024);
    };
    auto issueB = [&](int kt, int buf) {
        char* base = sB + buf * C_::B_STAGE + (g * (BN / 16) + w4 * NT) * 1024;
#pragma unroll
        for (int j = 0; j < NT; ++j) glds16(bsrc[j] + (long long)kt * BK, base + j * 1024);
    };

    f32x16_t acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhi = lane >> 5, fswz = (lane >> 1) & 7;
    const int a_row_off = g * A_HALF + (wm * 64 + frow) * 128;
    const int b_row_off = (wn * (BN / 2) + frow) * 128;

    bf16x8_t af[2][2], bfr[NT][2];
    auto load_frags = [&](int stage, int buf, int h) {
        const char* sa = sA + stage * A_STAGE + a_row_off;
        const char* sb = sB + buf * C_::B_STAGE + b_row_off;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((2 * (2 * h + kk) + fhi) ^ fswz) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][kk] = *(const bf16x8_t*)(sa + i * 32 * 128 + ch);
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j][kk] = *(const bf16x8_t*)(sb + j * 32 * 128 + ch);
        }
    };
    auto mfma_slot = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][kk], bfr[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    {
    // ---- prologue: A(0), B(0), A(1) (variant 4 also B(1): B runs two K-tiles ahead like A)
    constexpr bool B3 = (V == 4);
    issueA(0, 0);
    issueB(0, 0);
    if (nk > 1) {
        issueA(1, 1);
        if (B3) {
            issueB(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PF_BARRIER();

    int stage = 0;                     // A stage of tile kt = kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = B3 ? stage : (kt & 1);
        const bool more2 = kt + 2 < nk;
        // ---- slot L0: fragments of K-half 0, prefetch B(kt+1) (variant 4: B(kt+2) into the stage tile kt-1 left)
        load_frags(stage, buf, 0);
        if (B3) {
            if (more2) issueB(kt + 2, stage == 0 ? 2 : stage - 1);
        } else if (kt + 1 < nk) {
            issueB(kt + 1, buf ^ 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PF_SCHED_FENCE();
        // ---- slot M0
        mfma_slot();
        PF_SCHED_FENCE();
        // ---- slot L1: fragments of K-half 1, prefetch A(kt+2)
        load_frags(stage, buf, 1);
        if (more2) issueA(kt + 2, stage == 0 ? 2 : stage - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (g == 1) {                  // group 1: this barrier is the one before group 0 reads tile kt+1
            if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B3 ? 4 + NT : 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PF_BARRIER();
        } else {
            PF_SCHED_FENCE();
        }
        // ---- slot M1
        mfma_slot();
        if (g == 0) {
            if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B3 ? 4 + NT : 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PF_BARRIER();
        } else {
            PF_SCHED_FENCE();
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    }
    PF_SCHED_FENCE();

    // ---- epilogue: per wave, two 32-row halves through a private LDS strip
    float* st = (float*)(smem + wid * C_::EPI_BYTES);
    constexpr int STR = C_::EPI_STRIDE;
    constexpr int CG = BN / 16;                 // 8-column groups per staged row
    const int wave_m0 = m0 + g * 128 + wm * 64;
    const int wave_n0 = n0 + wn * (BN / 2);
    // GroupNorm statistics of the stored output (conv only, p.gn_stats): a lane owns the same 8 columns in every item
    // (64 % CG == 0), so it sums its rows in registers; lanes -> waves -> one double atomic per (column, statistic) and tile
    constexpr bool STATS = CONV && (64 % CG == 0);
    const bool do_stats = STATS && p.gn_stats != nullptr;
    float gs[8], gq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fhi;
                st[row * STR + j * 32 + frow] = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // Two passes over the strip's items: every global load of the strip (bias, gate, the residual pieces) is issued
        // before its first store.  In the one-pass form (rounds 1-4) each item's loads stood behind the previous item's
        // store -- `res` / `bias` may alias `C` for all the compiler knows -- and it guarded them with s_waitcnt vmcnt(0):
        // three dependent load round trips + one store round trip per item.
        // (COLFIX: 64 % CG == 0 -- BN = 128 / 256 -- a lane owns the same 8 columns in every item: ONE bias / gate set)
        constexpr int NI = CG / 2;
        constexpr bool COLFIX = 64 % CG == 0;
        constexpr int NB = COLFIX ? 1 : NI;
        long long coffs[NI];
        bool oks[NI];
        u32x4_t rr[NI];
        f32x4_t bb0[NB], bb1[NB], gg0[NB], gg1[NB];
        const bool has_res = (p.flags & PF_GEMM_GATE_RES) != 0;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + lane;
            const int row = item / CG, cgi = item - row * CG;
            const int m = wave_m0 + i * 32 + row;
            const int n = wave_n0 + cgi * 8;
            bool ok = m < p.M && n < p.n_valid;
            long long coff = 0;
            if (ok) {
                if (CONV && p.om.mode == 1) {
                    const int hw = p.om.H * p.om.W;
                    const int tt = m / hw, rem = m - tt * hw;
                    const int hh = rem / p.om.W, ww = rem - hh * p.om.W;
                    const int gg = n / p.om.Cg, cc = n - gg * p.om.Cg;
                    const int shw = p.om.sh * p.om.sw;
                    const int pt = gg / shw, g2 = gg - pt * shw;
                    const int ph = g2 / p.om.sw, pw = g2 - ph * p.om.sw;
                    const int tf = tt * p.om.st + pt + p.om.t_shift;
                    ok = tf >= 0;
                    coff = p.om.base_off +
                           (((long long)(tf < 0 ? 0 : tf) * p.om.Hop + (hh * p.om.sh + ph)) * p.om.Wop + (ww * p.om.sw + pw)) *
                               p.om.Cout_pitch + cc;
                } else {
                    coff = (long long)b * p.sC + (long long)m * p.ldc + n;
                }
            }
            oks[it] = ok;
            coffs[it] = coff;
            const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, o4 = (f32x4_t){1.f, 1.f, 1.f, 1.f};
            rr[it] = (u32x4_t){0u, 0u, 0u, 0u};
            if (!COLFIX || it == 0) {
                const int ib = COLFIX ? 0 : it;
                const bool nok = COLFIX ? n < p.n_valid : ok;          // (COLFIX: the columns are valid or not for all items)
                bb0[ib] = bb1[ib] = z4;
                gg0[ib] = gg1[ib] = o4;
                if (nok) {
                    if (p.bias) { bb0[ib] = *(const f32x4_t*)(p.bias + n); bb1[ib] = *(const f32x4_t*)(p.bias + n + 4); }
                    if (has_res && p.gate) {
                        const float* gp = p.gate + (long long)b * p.gate_stride + n;
                        gg0[ib] = *(const f32x4_t*)gp;
                        gg1[ib] = *(const f32x4_t*)(gp + 4);
                    }
                }
            }
            if (ok && has_res) {
                const long long roff =
                    (CONV && p.om.mode == 1) ? coff : ((long long)b * p.sR + (long long)m * p.ldr + n);
                rr[it] = *(const u32x4_t*)(p.res + roff);
            }
        }
        // every loaded register is redefined HERE, outside the items' predicated blocks: the compiler then places its one wait
        // for the loads above in front of this point, and none of its own behind a store (where vmcnt(0) would also wait for
        // that store: gemm8p.hip, epilogue_tile, has the same statement for the same reason)
#pragma unroll
        for (int it = 0; it < NI; ++it) asm volatile("" : "+v"(rr[it]));
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) asm volatile("" : "+v"(bb0[ib]), "+v"(bb1[ib]), "+v"(gg0[ib]), "+v"(gg1[ib]));
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int item = it * 64 + lane;
            const int row = item / CG, cgi = item - row * CG;
            const int n = wave_n0 + cgi * 8;
            const f32x4_t v0 = *(const f32x4_t*)(st + row * STR + cgi * 8);
            const f32x4_t v1 = *(const f32x4_t*)(st + row * STR + cgi * 8 + 4);
            if (!oks[it]) continue;
            float v[8];
            const int ib = COLFIX ? 0 : it;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bb0[ib][e]; v[4 + e] = v1[e] + bb1[ib][e]; }
            if (n >= p.gelu_from) {
                act8(v, p.flags);
            }
            const long long coff = coffs[it];
            if (has_res) {
                float rv[8];
                unpack8(rr[it], rv);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = rv[e] + gg0[ib][e] * v[e]; v[4 + e] = rv[4 + e] + gg1[ib][e] * v[4 + e]; }
            }
            if (p.out_scale != 1.f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
            }
            if (p.flags & PF_GEMM_OUT_F32) {
                float* c = (float*)p.C + coff;
                *(f32x4_t*)c = (f32x4_t){v[0], v[1], v[2], v[3]};
                *(f32x4_t*)(c + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
            } else {
                const u32x4_t packed = pack8(v);
                *(u32x4_t*)((bf16_t*)p.C + coff) = packed;
                if (STATS) {
                    if (do_stats) {               // statistics of the values as stored (what pf_gn_stats would read back)
                        float sv[8];
                        unpack8(packed, sv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { gs[e] += sv[e]; gq[e] += sv[e] * sv[e]; }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (STATS) {
        if (do_stats) {                            // workgroup-uniform
#pragma unroll
            for (int off = CG; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gs[e] += __shfl_xor(gs[e], off, 64);
                    gq[e] += __shfl_xor(gq[e], off, 64);
                }
            if (lane < CG) {                       // the wave's sums of columns wave_n0 + 8 lane + e over its 64 rows
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    st[(lane * 8 + e) * 2] = gs[e];
                    st[(lane * 8 + e) * 2 + 1] = gq[e];
                }
            }
            PF_BARRIER();
            if (tid < 2 * BN) {
                const int c = tid >> 1, k = tid & 1;
                const int wn_ = c / (BN / 2), cc = c - wn_ * (BN / 2);
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) sum += ((const float*)(smem + (wn_ + 2 * w) * C_::EPI_BYTES))[cc * 2 + k];
                const int n = n0 + c;
                if (n < p.n_valid && m0 < p.M) {
                    const int frame = m0 / (p.om.H * p.om.W);      // a tile lies inside one frame (H * W % 256 == 0)
                    atomicAdd(p.gn_stats + ((long long)frame * p.gn_C + n) * 2 + k, (double)sum);
                }
            }
        }
    }
}

template <int BN, bool CONV, int V>
int launch(const Args& a, hipStream_t stream) {
    const int grid = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.batch;
    constexpr int SM = (V == 4) ? Cfg<BN>::SMEM3 : Cfg<BN>::SMEM;
    PF_SET_MAX_LDS_ONCE((gemm256_kernel<BN, CONV, V>), SM);
    hipLaunchKernelGGL((gemm256_kernel<BN, CONV, V>), dim3(grid), dim3(512), SM, stream, a);
    return 0;
}

}  // namespace

// Picks the tile width for the 256-row kernel, 0 = use the 128x128 kernel of gemm.hip.
// force: 0 auto, 128/192/256 = that width if it divides N, -1 = never.
int pf_gemm256_pick(long long M_total, int M, int batch, int N, int force) {
    if (force < 0) return 0;
    if (force > 0) return (N % force == 0 || (force == 256 && N % 8 == 0 && N > 256)) ? force : 0;
    // 256-wide tiles move the fewest LDS / L2 bytes per FLOP (the kernels are power-bound: DESIGN.md section 3); an N
    // tail is computed on clamped filter rows and masked at the store, so 256 is used whenever the padded columns
    // cost < 3 % (N = 13 440, 5 760, 7 680, ...); otherwise 192 (N = 1 920) or 128 (the full-resolution VAE convs).
    const int n256 = (N + 255) / 256 * 256;
    int bn = 0;
    if (N % 8 == 0 && (n256 - N) * 100 < 3 * N) bn = 256;
    else if (N % 192 == 0) bn = 192;
    else if (N % 128 == 0 && M_total >= 65536) bn = 128;
    if (!bn) return 0;
    // small problems: the 128x128 tiles (2 blocks / CU) fill the 256 CUs better
    auto tiles_of = [&](int w) { return (long long)((M + BM - 1) / BM) * batch * ((N + w - 1) / w); };
    if (tiles_of(bn) >= 192) return bn;
    // mid-size problems (a sequence-parallel rank at P = 8: M ~ 2 x 1 936 rows, N = 1 920): too few 256 x 192 tiles
    // for one round of the chip, but 256 x 128 tiles fill it (16 x 15 = 240) and still run the ping-pong kernel
    // instead of the 128 x 128 kernel that this shape used to fall back to (0.09 of peak in the round-1 profile)
    if (bn != 128 && N % 128 == 0 && tiles_of(128) >= 192) return 128;
    return 0;
}

int pf_gemm256_launch(const Args& a, int bn, bool conv, hipStream_t stream) {
    // BN = 128 leaves room for a third B stage (144 KiB): B then runs two K-tiles ahead like A (+2-5 %, V = 4)
    switch (bn) {
        case 128: return conv ? launch<128, true, 4>(a, stream) : launch<128, false, 4>(a, stream);
        case 192: return conv ? launch<192, true, 1>(a, stream) : launch<192, false, 1>(a, stream);
        case 256: return conv ? launch<256, true, 1>(a, stream) : launch<256, false, 1>(a, stream);
    }
    return -1;
}
