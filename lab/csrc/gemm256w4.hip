// bf16 MFMA GEMM, 256 x BN block tile, FOUR waves with 128 x BN/2 wave tiles (one wave per SIMD, whole register file).
//
// Same contract and LDS image as gemm256.hip; selected with pf_gemm_set_variant(3), NOT the default.  A 128 x 128 wave
// tile reads 0.5 KiB of LDS per MFMA where the 64 x 128 tiles of the 8-wave kernel read 0.75 KiB; the accumulators (256
// registers) live in the AGPR half of the unified file.  With one wave per SIMD nothing else covers the wave's own
// issue time, so every memory instruction sits in an MFMA gap: each 16-wide k-step issues its first MFMA, then one or
// two of {next k-step's fragment reads, LDS-DMA pieces of A(kt+2) in phase 0 / B(kt+2) after the phase-3 barrier}
// behind each following MFMA (sched_barrier(0) pins the order); LDS-DMA runs a full K-tile ahead (A triple-, B
// double-buffered, counted vmcnt), one barrier per K-tile.
//
// Measured (profiles/r01_gemm_operand_path_diagnostics.log, M = 30 976): equal to the 8-wave kernel at K >= 7 680
// (1.14-1.31 PFLOP/s), 3-4 % behind at K = 1 920 where the per-tile prologue / 4-wave epilogue weighs more, and behind
// on the smaller M of the early units: routing the K >= 7 680 GEMMs of a whole C3 video to it cost 2 % of their time
// (profiles/r01_c3_rocprofv3_kernel_stats_v4_w4_longK.csv), so the default stays the 8-wave kernel.  The DBG
// builds bound the loop from above: without the steady-state LDS-DMA loads it runs 1.52-1.61 PFLOP/s, without DMA
// and fragment reads 1.65-1.75 -- the 25-33 % are the ISSUE cost of the LDS-DMA pieces (16 x 1 KiB per wave and
// K-tile, tens of cycles each, MI355X_MICROARCH.md) which a lone wave per SIMD cannot hide behind MFMAs.  Moving the
// operands through registers instead (global_load_dwordx4 -> ds_write_b128, two LDS stages) was 10-17 % slower (the
// wide LDS stores cost 13+ cycles per wave-instruction); issuing DMA pieces from inside the 8-wave kernel's MFMA slots
// was 20 % slower.  hipBLASLt's hand-scheduled 256 x 256 x 64 / 16x16x32 / 4-wave kernel reaches 1.37-1.59 PFLOP/s on
// the same shapes (profiles/r01_vendor_library_compare.log): that is the remaining headroom.
#include <type_traits>
#include "common.h"
#include "pyflow_hip.h"
#include "gemm_args.h"

using namespace pfgemm;

namespace {

constexpr int BM = 256, BK = 64;
constexpr int A_HALF = 128 * BK * 2;
constexpr int A_STAGE = 2 * A_HALF;
constexpr int A_BYTES = 3 * A_STAGE;
constexpr int GROUP_M = 4;

template <int BN>
struct Cfg {
    static constexpr int NT = BN / 64;
    static constexpr int NB = BN / 32;                 // B pieces per wave per K-tile
    static constexpr int B_STAGE = BN * BK * 2;
    static constexpr int SMEM = A_BYTES + 2 * B_STAGE;
    static constexpr int EPI_STRIDE = BN / 2 + 4;
    static constexpr int EPI_BYTES = 32 * EPI_STRIDE * 4;
    static_assert(4 * EPI_BYTES <= SMEM, "epilogue strips must fit");
};

#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PF_BARRIER()                         \
    do {                                     \
        PF_SCHED_FENCE();                    \
        __builtin_amdgcn_s_barrier();        \
        PF_SCHED_FENCE();                    \
    } while (0)

// DBG (diagnostics only, results are garbage): 1 = steady-state LDS-DMA loads not issued, 2 = neither DMA nor fragment
// reads (pure MFMA + barrier loop on whatever the first fragments held): upper bounds for the memory system's share.
template <int BN, bool CONV, int DBG = 0>
__global__ __launch_bounds__(256, 1) void gemm256w4_kernel(const Args p) {
    using C_ = Cfg<BN>;
    constexpr int NT = C_::NT, NB = C_::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sB = smem + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int TM = tiles_m * p.batch;
    const int nwg = TM * tiles_n;
    int t = xcd_remap(blockIdx.x, nwg);
    const int group_sz = GROUP_M * tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GROUP_M;
    const int gm = min(TM - first_m, GROUP_M);
    const int r_in = t - grp * group_sz;
    const int tn = r_in / gm;
    const int tmm = first_m + (r_in - tn * gm);
    const int b = tmm / tiles_m, tm = tmm - b * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA sources: the two waves of M-half wm load its 16 pieces (8 each); every wave loads NB pieces of B
    const bf16_t* A = p.A + (long long)b * p.sA;
    const bf16_t* asrc[8];
    const bf16_t* bsrc[NB];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ih = wn * 8 + j;
        const int c = (lane & 7) ^ (((ih & 1) << 2) + (lane >> 4));
        int m = m0 + wm * 128 + 8 * ih + (lane >> 3);
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
    for (int j = 0; j < NB; ++j) {
        const int ib = wid * NB + j;
        const int c = (lane & 7) ^ (((ib & 1) << 2) + (lane >> 4));
        int nrow = n0 + 8 * ib + (lane >> 3);
        nrow = nrow < p.N ? nrow : p.N - 1;
        bsrc[j] = p.W + (long long)nrow * p.ldw + c * 8;
    }
    const int nk = p.K / BK;

    // K-tile offset of the A operand (implicit-GEMM: tap + channel block of this K-tile)
    auto a_koff = [&](int kt) -> long long {
        if (CONV) {
            const int k0 = kt * BK;
            const int tap = k0 / p.cg.Cin, c0 = k0 - tap * p.cg.Cin;
            const int khw = p.cg.kh * p.cg.kw;
            const int dt = tap / khw, r2 = tap - dt * khw;
            const int dh = r2 / p.cg.kw, dw = r2 - dh * p.cg.kw;
            return (((long long)dt * p.cg.Hp + dh) * p.cg.Wp + dw) * p.cg.Cin + c0;
        }
        return (long long)kt * BK;
    };
    auto issueA = [&](int kt, int stage) {
        const long long aoff = a_koff(kt);
        char* base = sA + stage * A_STAGE + wm * A_HALF + wn * 8192;
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(asrc[j] + aoff, base + j * 1024);
    };
    auto issueB = [&](int kt, int buf) {
        char* base = sB + buf * C_::B_STAGE + wid * NB * 1024;
#pragma unroll
        for (int j = 0; j < NB; ++j) glds16(bsrc[j] + (long long)kt * BK, base + j * 1024);
    };

    f32x16_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhi = lane >> 5, fswz = (lane >> 1) & 7;
    const int a_row_off = wm * A_HALF + frow * 128;
    const int b_row_off = (wn * (BN / 2) + frow) * 128;

    bf16x8_t fa[2][4], fb[2][NT];
    auto loadF = [&](int set, int stg, int bufi, int ph) {
        const char* sa = sA + stg * A_STAGE + a_row_off;
        const char* sb = sB + bufi * C_::B_STAGE + b_row_off;
        const int ch = ((2 * ph + fhi) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *(const bf16x8_t*)(sa + i * 32 * 128 + ch);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[set][j] = *(const bf16x8_t*)(sb + j * 32 * 128 + ch);
    };

    // One 16-wide k-step = 4*NT MFMAs on fragment set PH&1.  With a single wave per SIMD nothing else fills the matrix
    // pipe while this wave issues memory instructions, so they are spread between the MFMAs (an MFMA executes for 32
    // cycles after issue; independent LDS / LDS-DMA instructions issue underneath it): the next k-step's fragment reads
    // (4 + NT ds_read_b128), in phase 0 the 8 LDS-DMA loads of A for K-tile kt+2, in phase 3 -- after the K-tile barrier
    // -- the NB LDS-DMA loads of B for K-tile kt+2.  DMA: K-tile kt+2 exists; LOAD: K-tile kt+1 exists.
    // sched_barrier(0) after every instruction group pins exactly this order.
    auto phase = [&](auto PHc, auto DMAc, auto LOADc, const int kt, const int stage, const int stage_n, const int buf) {
        constexpr int PH = decltype(PHc)::value;
        constexpr bool DMA = decltype(DMAc)::value, LOAD = decltype(LOADc)::value;
        constexpr int cur = PH & 1, nxt = cur ^ 1;
        constexpr int NM = 4 * NT;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
        PF_SCHED_FENCE();
        const char* sa = nullptr;
        const char* sb = nullptr;
        int ch = 0;
        char* dbase = nullptr;
        long long aoff = 0;
        if constexpr (PH == 3) {
            if constexpr (LOAD) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PF_BARRIER();
                sa = sA + stage_n * A_STAGE + a_row_off;
                sb = sB + (buf ^ 1) * C_::B_STAGE + b_row_off;
                ch = (fhi ^ fswz) << 4;
                if constexpr (DMA) dbase = sB + buf * C_::B_STAGE + wid * NB * 1024;
            }
        } else {
            sa = sA + stage * A_STAGE + a_row_off;
            sb = sB + buf * C_::B_STAGE + b_row_off;
            ch = ((2 * (PH + 1) + fhi) ^ fswz) << 4;
            if constexpr (PH == 0 && DMA) {
                aoff = a_koff(kt + 2);
                dbase = sA + (stage == 0 ? 2 : stage - 1) * A_STAGE + wm * A_HALF + wn * 8192;
            }
        }
        constexpr int NDMA = (!DMA || DBG >= 1) ? 0 : (PH == 0 ? 8 : (PH == 3 ? NB : 0));
        constexpr int NLD = ((PH == 3 && !LOAD) || DBG >= 2) ? 0 : 4 + NT;
        constexpr int NOPS = NDMA + NLD;
        constexpr int RATE = (NOPS + NM - 2) / (NM - 1);          // memory instructions per MFMA slot (1 or 2)
        auto mem_op = [&](const int k) {
            if (k < NDMA) {
                if (PH == 0) glds16(asrc[k] + aoff, dbase + k * 1024);
                else glds16(bsrc[k] + (long long)(kt + 2) * BK, dbase + k * 1024);
            } else {
                const int f = k - NDMA;
                if (f < 4) fa[nxt][f] = *(const bf16x8_t*)(sa + f * 32 * 128 + ch);
                else fb[nxt][f - 4] = *(const bf16x8_t*)(sb + (f - 4) * 32 * 128 + ch);
            }
        };
        PF_SCHED_FENCE();
#pragma unroll
        for (int m = 1; m < NM; ++m) {
#pragma unroll
            for (int r = 0; r < RATE; ++r) {
                const int k = (m - 1) * RATE + r;
                if (k < NOPS) mem_op(k);
            }
            PF_SCHED_FENCE();
            const int i = m / NT, j = m - i * NT;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
            PF_SCHED_FENCE();
        }
    };
    using std::integral_constant;
    using T_ = integral_constant<bool, true>;
    using F_ = integral_constant<bool, false>;
    auto ktile = [&](auto DMAc, auto LOADc, const int kt, const int stage) -> int {
        const int buf = kt & 1;
        const int stage_n = stage == 2 ? 0 : stage + 1;
        phase(integral_constant<int, 0>{}, DMAc, LOADc, kt, stage, stage_n, buf);
        phase(integral_constant<int, 1>{}, DMAc, LOADc, kt, stage, stage_n, buf);
        phase(integral_constant<int, 2>{}, DMAc, LOADc, kt, stage, stage_n, buf);
        phase(integral_constant<int, 3>{}, DMAc, LOADc, kt, stage, stage_n, buf);
        return stage_n;
    };

    {
        issueA(0, 0);
        issueB(0, 0);
        if (nk > 1) {
            issueA(1, 1);
            issueB(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NB) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        PF_BARRIER();
        loadF(0, 0, 0, 0);
        PF_SCHED_FENCE();
        int stage = 0, kt = 0;
        for (; kt + 2 < nk; ++kt) stage = ktile(T_{}, T_{}, kt, stage);       // steady state
        if (kt + 1 < nk) {                                                      // K-tile nk-2: nothing left to request
            stage = ktile(F_{}, T_{}, kt, stage);
            ++kt;
        }
        stage = ktile(F_{}, F_{}, kt, stage);                                   // last K-tile
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PF_BARRIER();
    PF_SCHED_FENCE();

    // ---- epilogue: per wave, four 32-row quarters through a private LDS strip
    float* st = (float*)(smem + wid * C_::EPI_BYTES);
    constexpr int STR = C_::EPI_STRIDE;
    constexpr int CG = BN / 16;                 // 8-column groups per staged row
    const int wave_m0 = m0 + wm * 128;
    const int wave_n0 = n0 + wn * (BN / 2);
    // one 32-row quarter; called with a literal index so that the accumulators stay in registers (a loop the compiler
    // declines to unroll would index acc[] dynamically and push all 256 registers through scratch memory)
    auto quarter = [&](const f32x16_t (&aq)[NT], const int i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fhi;
                st[row * STR + j * 32 + frow] = aq[j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int it = 0; it < CG / 2; ++it) {
            const int item = it * 64 + lane;
            const int row = item / CG, cgi = item - row * CG;
            const int m = wave_m0 + i * 32 + row;
            const int n = wave_n0 + cgi * 8;
            const f32x4_t v0 = *(const f32x4_t*)(st + row * STR + cgi * 8);
            const f32x4_t v1 = *(const f32x4_t*)(st + row * STR + cgi * 8 + 4);
            if (m >= p.M || n >= p.n_valid) continue;
            float v[8];
            if (p.bias) {
                const f32x4_t b0 = *(const f32x4_t*)(p.bias + n), b1 = *(const f32x4_t*)(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
            }
            if (n >= p.gelu_from) {
                act8(v, p.flags);
            }
            long long coff;
            if (CONV && p.om.mode == 1) {
                const int hw = p.om.H * p.om.W;
                const int tt = m / hw, rem = m - tt * hw;
                const int hh = rem / p.om.W, ww = rem - hh * p.om.W;
                const int gg = n / p.om.Cg, cc = n - gg * p.om.Cg;
                const int shw = p.om.sh * p.om.sw;
                const int pt = gg / shw, g2 = gg - pt * shw;
                const int ph = g2 / p.om.sw, pw = g2 - ph * p.om.sw;
                const int tf = tt * p.om.st + pt + p.om.t_shift;
                if (tf < 0) continue;
                coff = p.om.base_off +
                       (((long long)tf * p.om.Hop + (hh * p.om.sh + ph)) * p.om.Wop + (ww * p.om.sw + pw)) *
                           p.om.Cout_pitch + cc;
            } else {
                coff = (long long)b * p.sC + (long long)m * p.ldc + n;
            }
            if (p.flags & PF_GEMM_GATE_RES) {
                float rv[8];
                const long long roff =
                    (CONV && p.om.mode == 1) ? coff : ((long long)b * p.sR + (long long)m * p.ldr + n);
                unpack8(*(const u32x4_t*)(p.res + roff), rv);
                if (p.gate) {
                    const float* gp = p.gate + (long long)b * p.gate_stride + n;
                    const f32x4_t g0 = *(const f32x4_t*)gp, g1 = *(const f32x4_t*)(gp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = rv[e] + g0[e] * v[e]; v[4 + e] = rv[4 + e] + g1[e] * v[4 + e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rv[e] + v[e];
                }
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
                *(u32x4_t*)((bf16_t*)p.C + coff) = pack8(v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    quarter(acc[0], 0);
    quarter(acc[1], 1);
    quarter(acc[2], 2);
    quarter(acc[3], 3);
}

template <int BN, bool CONV, int DBG = 0>
int launch4(const Args& a, hipStream_t stream) {
    const int grid = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.batch;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm256w4_kernel<BN, CONV, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            Cfg<BN>::SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256w4_kernel<BN, CONV, DBG>), dim3(grid), dim3(256), Cfg<BN>::SMEM, stream, a);
    return 0;
}

}  // namespace

int pf_gemm256w4_launch(const Args& a, int bn, bool conv, hipStream_t stream, int dbg) {
    if (dbg == 1 && !conv) return bn == 256 ? launch4<256, false, 1>(a, stream) : launch4<192, false, 1>(a, stream);
    if (dbg == 2 && !conv) return bn == 256 ? launch4<256, false, 2>(a, stream) : launch4<192, false, 2>(a, stream);
    if (bn == 256) return conv ? launch4<256, true>(a, stream) : launch4<256, false>(a, stream);
    if (bn == 192) return conv ? launch4<192, true>(a, stream) : launch4<192, false>(a, stream);
    return -1;
}
