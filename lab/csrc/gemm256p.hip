// bf16 MFMA GEMM, 256 x BN block tile, 8 waves -- PERSISTENT form of gemm256.hip's default (one-barrier-per-K-tile) loop.
// Selected with pf_gemm_set_variant(10); NOT the default: measured neutral (-4 ... +2 % against the one-tile-per-workgroup
// kernel on the DiT shapes, +2-8 % for BN = 128 at K = 1 920; profiles/r01_microbench_gemm256_persistent.log), i.e. the
// prologue / epilogue of a tile is not what holds the K = 1 920 GEMMs back.  Kept as a tested variant.
//
// Hypothesis it tests: with 160 KiB of LDS a CU hosts ONE workgroup of gemm256_kernel, so a tile's prologue (first operand K-tiles
// through L2 / HBM: several microseconds with nothing to compute) and its epilogue (fp32 accumulators through LDS
// strips, 128 KiB of stores) are never overlapped with MFMA work: at K = 1 920 (30 K-tiles, ~55 us per tile) that is
// ~10 % of the kernel (profiles/r01_gemm_operand_path_diagnostics.log).  Here one workgroup per CU walks over its tiles
// (tile = blockIdx.x + i * gridDim.x, the same XCD-aware order) and, when a tile's K-loop ends, FIRST issues the LDS-DMA
// loads of the next tile's K-tiles 0 and 1, THEN runs the epilogue of the finished tile, so the next tile's operands
// land while the accumulators are stored.
//
// LDS plan (same 3 A stages + 2 B stages as gemm256.hip): a tile's K-tile kt lives in A stage kt % 3 and B stage
// (kt + 1) & 1, so the next tile's prefetch fills A0, A1 and B1; the epilogue strips (8 rows x BN/2 fp32 per wave and
// pass, 33 KiB per workgroup) sit in the A2 | B0 range that the prefetch does not touch.  After the loop's last barrier
// no wave reads operand LDS any more (group 1 passes it after its last fragment reads, group 0 after its last MFMA
// slot), so both the prefetch and the strips may start at once.  One vmcnt(0) + s_barrier separates two tiles.
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
    static constexpr int B_STAGE = BN * BK * 2;
    static constexpr int SMEM = A_BYTES + 2 * B_STAGE;
    static constexpr int STR = BN / 2 + 4;                 // floats per staged row
    static constexpr int EPI_BYTES = 8 * STR * 4;          // one 8-row pass of one wave
    static_assert(8 * EPI_BYTES <= A_STAGE + B_STAGE, "strips must fit into A2 | B0");
};

#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PF_BARRIER()                         \
    do {                                     \
        PF_SCHED_FENCE();                    \
        __builtin_amdgcn_s_barrier();        \
        PF_SCHED_FENCE();                    \
    } while (0)

template <int BN, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const Args p) {
    using C_ = Cfg<BN>;
    constexpr int NT = C_::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sB = smem + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wid >> 2, w4 = wid & 3, wm = w4 >> 1, wn = w4 & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int TM = tiles_m * p.batch;
    const int nwg = TM * tiles_n;
    const int GM = p.group_m > 0 ? p.group_m : GROUP_M;
    const int group_sz = GM * tiles_n;
    const int nk = p.K / BK;

    // ---- per-tile state: origin + LDS-DMA sources (same piece ownership and swizzle as gemm256.hip)
    int m0 = 0, n0 = 0, b = 0;
    const bf16_t* asrc[4];
    const bf16_t* bsrc[NT];
    auto setup = [&](const int tlin) {
        const int t = xcd_remap(tlin, nwg);
        const int grp = t / group_sz;
        const int first_m = grp * GM;
        const int gm = min(TM - first_m, GM);
        const int r_in = t - grp * group_sz;
        const int tn = r_in / gm;
        const int tmm = first_m + (r_in - tn * gm);
        b = tmm / tiles_m;
        const int tm = tmm - b * tiles_m;
        m0 = tm * BM;
        n0 = tn * BN;
        const bf16_t* A = p.A + (long long)b * p.sA;
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
    };
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
    auto prologue = [&]() {            // K-tiles 0 and 1 of the tile `setup` described: A0, B1, A1
        issueA(0, 0);
        issueB(0, 1);
        if (nk > 1) issueA(1, 1);
    };

    f32x16_t acc[2][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

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

    // ---- epilogue of the tile at (em0, en0, eb): 8 passes of 8 rows per wave through a private strip in A2 | B0
    constexpr int STR = C_::STR;
    constexpr int CG = BN / 16;                 // 8-column groups per staged row
    float* const st = (float*)(smem + 2 * A_STAGE + wid * C_::EPI_BYTES);
    auto epi_pass = [&](const f32x16_t (&aq)[NT], auto Qc, const int row0, const int em0, const int en0, const int eb) {
        constexpr int Q = decltype(Qc)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) st[(rr + 4 * fhi) * STR + j * 32 + frow] = aq[j][Q * 4 + rr];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int item = lane; item < 8 * CG; item += 64) {
            const int row = item / CG, cgi = item - row * CG;
            const int m = em0 + g * 128 + wm * 64 + row0 + Q * 8 + row;
            const int n = en0 + wn * (BN / 2) + cgi * 8;
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
                coff = (long long)eb * p.sC + (long long)m * p.ldc + n;
            }
            if (p.flags & PF_GEMM_GATE_RES) {
                float rv[8];
                const long long roff =
                    (CONV && p.om.mode == 1) ? coff : ((long long)eb * p.sR + (long long)m * p.ldr + n);
                unpack8(*(const u32x4_t*)(p.res + roff), rv);
                if (p.gate) {
                    const float* gp = p.gate + (long long)eb * p.gate_stride + n;
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
    using std::integral_constant;
    auto epilogue = [&](const int em0, const int en0, const int eb) {
        // literal indices: the accumulators must stay in registers
        epi_pass(acc[0], integral_constant<int, 0>{}, 0, em0, en0, eb);
        epi_pass(acc[0], integral_constant<int, 1>{}, 0, em0, en0, eb);
        epi_pass(acc[0], integral_constant<int, 2>{}, 0, em0, en0, eb);
        epi_pass(acc[0], integral_constant<int, 3>{}, 0, em0, en0, eb);
        epi_pass(acc[1], integral_constant<int, 0>{}, 32, em0, en0, eb);
        epi_pass(acc[1], integral_constant<int, 1>{}, 32, em0, en0, eb);
        epi_pass(acc[1], integral_constant<int, 2>{}, 32, em0, en0, eb);
        epi_pass(acc[1], integral_constant<int, 3>{}, 32, em0, en0, eb);
    };

    // ---- the tile walk
    int tlin = blockIdx.x;
    setup(tlin);
    prologue();
    zero_acc();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PF_BARRIER();
    for (;;) {
        int stage = 0;                     // A stage of K-tile kt = kt % 3; B stage (kt + 1) & 1
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = (kt + 1) & 1;
            const bool more2 = kt + 2 < nk;
            load_frags(stage, buf, 0);
            if (kt + 1 < nk) issueB(kt + 1, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PF_SCHED_FENCE();
            mfma_slot();
            PF_SCHED_FENCE();
            load_frags(stage, buf, 1);
            if (more2) issueA(kt + 2, stage == 0 ? 2 : stage - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (g == 1) {                  // group 1 runs one slot behind: its barrier of the K-tile is here
                if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PF_BARRIER();
            } else {
                PF_SCHED_FENCE();
            }
            mfma_slot();
            if (g == 0) {
                if (more2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PF_BARRIER();
            } else {
                PF_SCHED_FENCE();
            }
            stage = stage == 2 ? 0 : stage + 1;
        }
        PF_SCHED_FENCE();
        // every wave has passed the last K-tile's barrier: no operand LDS is read any more
        const int em0 = m0, en0 = n0, eb = b;
        tlin += gridDim.x;
        const bool has_next = tlin < nwg;
        if (has_next) {
            setup(tlin);
            prologue();                    // lands in A0 / A1 / B1 while the strips below use A2 | B0
        }
        PF_SCHED_FENCE();
        epilogue(em0, en0, eb);
        if (!has_next) break;
        zero_acc();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PF_BARRIER();                      // operands of the next tile landed for everyone; all strips are done
    }
}

template <int BN, bool CONV>
int launchp(const Args& a, hipStream_t stream) {
    const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.batch;
    static int n_cu = 0;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm256p_kernel<BN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            Cfg<BN>::SMEM);
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
        n_cu -= n_cu % 8;                  // keep blockIdx % 8 == tile index % 8 across the walk (XCD-contiguous chunks)
        if (n_cu <= 0) n_cu = 8;
        attr_set = true;
    }
    const int grid = tiles < n_cu ? tiles : n_cu;
    hipLaunchKernelGGL((gemm256p_kernel<BN, CONV>), dim3(grid), dim3(512), Cfg<BN>::SMEM, stream, a);
    return 0;
}

}  // namespace

int pf_gemm256p_launch(const Args& a, int bn, bool conv, hipStream_t stream) {
    switch (bn) {
        case 128: return conv ? launchp<128, true>(a, stream) : launchp<128, false>(a, stream);
        case 192: return conv ? launchp<192, true>(a, stream) : launchp<192, false>(a, stream);
        case 256: return conv ? launchp<256, true>(a, stream) : launchp<256, false>(a, stream);
    }
    return -1;
}
