// bf16 MFMA GEMM, 256 x BN block tile, FOUR waves with 128 x BN/2 wave tiles (one wave per SIMD, whole register file).
//
// Same contract and LDS image as gemm256.hip.  Why a third kernel: both MFMA kernels are power-bound on real data
// (DESIGN.md section 3), so the lever is energy per FLOP.  A 128 x 128 wave tile reads 0.5 KiB of LDS per MFMA where the
// 64 x 128 tiles of the 8-wave kernel read 0.75 KiB; the accumulators (256 registers) live in the unified VGPR/AGPR
// file.  With one wave per SIMD nothing else covers LDS latency, so the main loop is the register-prefetch pipeline:
// the fragments of 16-wide k-step p+1 are requested right after the first MFMA of step p (two fragment sets), LDS-DMA
// runs a full K-tile ahead (A triple-, B double-buffered, counted vmcnt), one barrier per K-tile.
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
};

#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PF_BARRIER()                         \
    do {                                     \
        PF_SCHED_FENCE();                    \
        __builtin_amdgcn_s_barrier();        \
        PF_SCHED_FENCE();                    \
    } while (0)

template <int BN, bool CONV>
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
    auto phase = [&](int set, auto&& between) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][0], fb[set][0], acc[0][0], 0, 0, 0);
        PF_SCHED_FENCE();
        between();
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (i | j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
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
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        const int stage_n = stage == 2 ? 0 : stage + 1;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int cur = ph & 1, nxt = cur ^ 1;
            phase(cur, [&]() {
                if (ph == 3) {
                    if (more1) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        PF_BARRIER();
                        if (more2) issueB(kt + 2, buf);
                        loadF(nxt, stage_n, buf ^ 1, 0);
                    }
                } else {
                    loadF(nxt, stage, buf, ph + 1);
                }
                if (ph == 0 && more2) issueA(kt + 2, stage == 0 ? 2 : stage - 1);
            });
        }
        stage = stage_n;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fhi;
                st[row * STR + j * 32 + frow] = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
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
    }
}

template <int BN, bool CONV>
int launch4(const Args& a, hipStream_t stream) {
    const int grid = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.batch;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm256w4_kernel<BN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            Cfg<BN>::SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256w4_kernel<BN, CONV>), dim3(grid), dim3(256), Cfg<BN>::SMEM, stream, a);
    return 0;
}

}  // namespace

int pf_gemm256w4_launch(const Args& a, int bn, bool conv, hipStream_t stream) {
    if (bn == 256) return conv ? launch4<256, true>(a, stream) : launch4<256, false>(a, stream);
    if (bn == 192) return conv ? launch4<192, true>(a, stream) : launch4<192, false>(a, stream);
    return -1;
}
