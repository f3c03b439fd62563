// lab/gemm4w_lab.hip -- prototype of a FOUR-wave bf16 GEMM main loop for gfx950 (one wave per SIMD, 128 x 128 wave tiles),
// hand-placed statement by statement, next to the shipped pf_gemm_bf16 (gemm8p: eight waves, barrier-paced ping-pong).
//
// Why: the round-3 review measured the vendor library 15-20 % ahead of gemm8p on the DiT's shapes.  gemm8p's two waves per
// SIMD need eight barriers per K-tile and read 24 KiB of fragments per wave and K-tile for 1.05 MFLOP; this form reads
// 32 KiB for 2.1 MFLOP (-33 % LDS traffic per flop), has ONE barrier per K-tile and no second wave to arbitrate with -- but
// nothing hides an issue stall either, so every statement of the loop is placed by hand (lab/gen_gemm4w_body.py): each of the
// 64 v_mfma_f32_32x32x16_bf16 of a K-tile is followed by at most one of its 32 fragment reads, 16 LDS writes, 16 global loads.
//
//   block tile 256 x 256 x 64, waves 2 (M) x 2 (N); C^T accumulators (first MFMA operand = W rows) so that a lane ends up with
//   8 consecutive columns of a row after one permlane32_swap (the epilogue of the attention kernels);
//   accumulator file a[0:255] = the 4 x 4 blocks of 32 x 32 (block (jn, im) at 16 (4 jn + im)), addressed by name;
//   operands: buffer_load_dwordx4 (SRD + per-lane offset + scalar K offset) -> two staging sets of 16 x 4 registers ->
//   ds_write_b128 into two 64-KiB LDS buffers (rows of 128 B, chunk c of row r at c ^ ((r >> 1) & 7): conflict-free for the
//   writes' 8-lane groups and the reads' 16-lane groups), fragments by ds_read_b128 one k-step ahead.
//
// Build:  cd lab && hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../pyramid-flow_amd/csrc gemm4w_lab.hip -ldl -o gemm4w_lab
// Run:    lab/gemm4w_lab [M N K]      (defaults: the DiT's projections at L = 15 488, batch 2)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
#include "pyflow_hip.h"
#include "common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int BUF = 65536;                 // one LDS buffer: A rows [0, 32 KiB), W rows [32 KiB, 64 KiB)

// VAR (timing experiments, wrong results): 1 no LDS writes, 2 no global loads, 3 no fragment reads, 4 no barrier
template <int STAMP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                        int M, int N, int K, int lda, int ldw, int ldc, unsigned* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a255");                                        // the whole accumulator file is in use
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int T = tiles_m * tiles_n;
    const int t = xcd_remap(blockIdx.x, T);
    // groups of 4 row tiles x all column tiles (neighbouring workgroups share operand panels)
    const int GM = 4, gsz = GM * tiles_n, grp = t / gsz, first_m = grp * GM, gm = min(tiles_m - first_m, GM);
    const int r_in = t - grp * gsz, tn = r_in / gm, tm = first_m + (r_in - tn * gm);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk = K / BK;

    // ---- global side: wave w loads rows (8 w + i) * 8 + (lane >> 3), i = 0..7, 16 bytes (lane & 7) of the K-tile's 128
    // raw buffer descriptors (base, stride 0, 2 GiB of records, dword format): wave-uniform, four SGPRs each
    auto make_srd = [](const void* base) {
        const unsigned long long b = (unsigned long long)base;
        i32x4_t r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
        r[2] = 0x7fffffff;
        r[3] = 0x00020000;
        return r;
    };
    i32x4_t srdA = make_srd(A + (long long)m0 * lda), srdW = make_srd(W + (long long)n0 * ldw);
    unsigned voffA[8], voffW[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (8 * wid + i) * 8 + (lane >> 3);
        voffA[i] = (unsigned)(min(m0 + row, M - 1) - m0) * (unsigned)(lda * 2) + (lane & 7) * 16;
        voffW[i] = (unsigned)(min(n0 + row, N - 1) - n0) * (unsigned)(ldw * 2) + (lane & 7) * 16;
    }
    // ---- LDS side.  Writes: row = 8 j + (lane >> 3) with j = 8 w + i: chunk ^ ((row >> 1) & 7) = (lane & 7) ^ (4 (i & 1) + (lane >> 4))
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned wrv[2][2];                    // [buffer][i & 1]; + i * 1024 (+ 32768 for W rows) as the instruction's offset
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int par = 0; par < 2; ++par)
            wrv[b][par] = lds0 + b * BUF + wid * 8192 + (lane >> 3) * 128 + ((((lane & 7) ^ (4 * par + (lane >> 4))) & 7) << 4);
    // Fragment reads (32x32x16 operand: lane -> row lane & 31, 16-byte chunk 2 ks + (lane >> 5) of the row's 128 bytes)
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;
    unsigned rdA[2][4], rdW[2][4];         // [buffer][k-step]; + block * 4096 as the instruction's offset
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned f = (unsigned)(frow * 128 + (((2 * ks + hi) ^ swz) << 4));
            rdA[b][ks] = lds0 + b * BUF + wr * 16384 + f;
            rdW[b][ks] = lds0 + b * BUF + 32768 + wc * 16384 + f;
        }

    bf16x8_t fa[2][4], fb[2][4];           // fragment sets (k-step parity) x blocks
    u32x4_t st[2][16];                     // staging sets (tile parity) x loads (0..7 A rows, 8..15 W rows)
    unsigned kofs = 0;                     // byte offset of the K-tile the NEXT global loads fetch

#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");
#define VMC(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");
#define BARRIER() if (VAR != 4) __builtin_amdgcn_s_barrier();
#define MFMA(JN, IM, FS)                                                                                                  \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * (4 * (JN) + (IM))), "n"(16 * (4 * (JN) + (IM)) + 15), \
                 "v"(fb[FS][JN]), "v"(fa[FS][IM]));
#define RDA(BUFI, KS, BLK, FS) if (VAR != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[FS][BLK]) : "v"(rdA[BUFI][KS]), "n"((BLK) * 4096));
#define RDW(BUFI, KS, BLK, FS) if (VAR != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[FS][BLK]) : "v"(rdW[BUFI][KS]), "n"((BLK) * 4096));
#define WRA(BUFI, I) if (VAR != 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wrv[BUFI][(I) & 1]), "v"(st[BUFI][I]), "n"((I) * 1024) : "memory");
#define WRW(BUFI, I) if (VAR != 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wrv[BUFI][(I) & 1]), "v"(st[BUFI][8 + (I)]), "n"(32768 + (I) * 1024) : "memory");
#define LDA(SET, I) if (VAR != 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[SET][I]) : "v"(voffA[I]), "s"(srdA), "s"(kofs) : "memory");
#define LDW(SET, I) if (VAR != 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[SET][8 + (I)]) : "v"(voffW[I]), "s"(srdW), "s"(kofs) : "memory");

    // ---- accumulators = 0 (MFMAs of zero operands write the accumulator file without a register move)
    {
        bf16x8_t z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.0f;
        asm volatile("" : "+v"(z));
#define ZERO(X) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %2, 0" ::"n"(16 * (X)), "n"(16 * (X) + 15), "v"(z));
        ZERO(0) ZERO(1) ZERO(2) ZERO(3) ZERO(4) ZERO(5) ZERO(6) ZERO(7) ZERO(8) ZERO(9) ZERO(10) ZERO(11) ZERO(12) ZERO(13) ZERO(14) ZERO(15)
#undef ZERO
    }
    // ---- prologue: tiles 0 and 1 into the staging sets, tile 0 into LDS buffer 0, its first fragments
    // kofs = byte offset of the K-tile the next global loads fetch (tile 0, 1, 2 in the prologue, then tile kt + 3)
#define KSTEP() kofs += 128;
    // GENERATED PROLOGUE BEGIN (lab/gen_gemm4w_body.py)
        LDA(0, 0)
        LDA(0, 1)
        LDA(0, 2)
        LDA(0, 3)
        LDA(0, 4)
        LDA(0, 5)
        LDA(0, 6)
        LDA(0, 7)
        LDW(0, 0)
        LDW(0, 1)
        LDW(0, 2)
        LDW(0, 3)
        LDW(0, 4)
        LDW(0, 5)
        LDW(0, 6)
        LDW(0, 7)
        KSTEP() LDA(1, 0)
        LDA(1, 1)
        LDA(1, 2)
        LDA(1, 3)
        LDA(1, 4)
        LDA(1, 5)
        LDA(1, 6)
        LDA(1, 7)
        LDW(1, 0)
        LDW(1, 1)
        LDW(1, 2)
        LDW(1, 3)
        LDW(1, 4)
        LDW(1, 5)
        LDW(1, 6)
        LDW(1, 7)
        VMC(31) WRA(0, 0)
        VMC(30) WRA(0, 1)
        VMC(29) WRA(0, 2)
        VMC(28) WRA(0, 3)
        VMC(27) WRA(0, 4)
        VMC(26) WRA(0, 5)
        VMC(25) WRA(0, 6)
        VMC(24) WRA(0, 7)
        VMC(23) WRW(0, 0)
        VMC(22) WRW(0, 1)
        VMC(21) WRW(0, 2)
        VMC(20) WRW(0, 3)
        VMC(19) WRW(0, 4)
        VMC(18) WRW(0, 5)
        VMC(17) WRW(0, 6)
        VMC(16) WRW(0, 7)
        KSTEP() LDA(0, 0)
        LDA(0, 1)
        LDA(0, 2)
        LDA(0, 3)
        LDA(0, 4)
        LDA(0, 5)
        LDA(0, 6)
        LDA(0, 7)
        LDW(0, 0)
        LDW(0, 1)
        LDW(0, 2)
        LDW(0, 3)
        LDW(0, 4)
        LDW(0, 5)
        LDW(0, 6)
        LDW(0, 7)
        LGKM(0) BARRIER()
        RDW(0, 0, 0, 0)
        RDA(0, 0, 0, 0)
        RDA(0, 0, 1, 0)
        RDA(0, 0, 2, 0)
        RDA(0, 0, 3, 0)
        RDW(0, 0, 1, 0)
        RDW(0, 0, 2, 0)
        RDW(0, 0, 3, 0)
        // GENERATED PROLOGUE END
    // ---- main loop: nk even, >= 6.  Tile kt lives in LDS buffer kt & 1 and came through staging set kt & 1
    const unsigned t_loop0 = STAMP ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    for (int kt = 0; kt + 4 < nk; kt += 2) {
        // GENERATED STEADY0 BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(0, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 1, 3, 1)
        MFMA(2, 0, 0)
        VMC(31) WRA(1, 0)
        MFMA(2, 1, 0)
        VMC(30) WRA(1, 1)
        MFMA(2, 2, 0)
        VMC(29) WRA(1, 2)
        MFMA(2, 3, 0)
        VMC(28) WRA(1, 3)
        MFMA(3, 0, 0)
        VMC(27) WRA(1, 4)
        MFMA(3, 1, 0)
        VMC(26) WRA(1, 5)
        MFMA(3, 2, 0)
        KSTEP() LDA(1, 0)
        MFMA(3, 3, 0)
        LDA(1, 1)
        LGKM(6)
        MFMA(0, 0, 1)
        RDW(0, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(0, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(0, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(0, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(0, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(0, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(0, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(0, 2, 3, 0)
        MFMA(2, 0, 1)
        VMC(27) WRA(1, 6)
        MFMA(2, 1, 1)
        VMC(26) WRA(1, 7)
        MFMA(2, 2, 1)
        VMC(25) WRW(1, 0)
        MFMA(2, 3, 1)
        VMC(24) WRW(1, 1)
        MFMA(3, 0, 1)
        VMC(23) WRW(1, 2)
        MFMA(3, 1, 1)
        LDA(1, 2)
        MFMA(3, 2, 1)
        LDA(1, 3)
        MFMA(3, 3, 1)
        LDA(1, 4)
        LGKM(5)
        MFMA(0, 0, 0)
        RDW(0, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 3, 3, 1)
        MFMA(2, 0, 0)
        VMC(25) WRW(1, 3)
        MFMA(2, 1, 0)
        VMC(24) WRW(1, 4)
        MFMA(2, 2, 0)
        VMC(23) WRW(1, 5)
        MFMA(2, 3, 0)
        VMC(22) WRW(1, 6)
        MFMA(3, 0, 0)
        VMC(21) WRW(1, 7)
        MFMA(3, 1, 0)
        LDA(1, 5)
        MFMA(3, 2, 0)
        LDA(1, 6)
        MFMA(3, 3, 0)
        LDA(1, 7)
        LGKM(0) BARRIER()
        LGKM(5)
        MFMA(0, 0, 1)
        RDW(1, 0, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 0, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 0, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 0, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 0, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 0, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 0, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 0, 3, 0)
        MFMA(2, 0, 1)
        LDW(1, 0)
        MFMA(2, 1, 1)
        LDW(1, 1)
        MFMA(2, 2, 1)
        LDW(1, 2)
        MFMA(2, 3, 1)
        LDW(1, 3)
        MFMA(3, 0, 1)
        LDW(1, 4)
        MFMA(3, 1, 1)
        LDW(1, 5)
        MFMA(3, 2, 1)
        LDW(1, 6)
        MFMA(3, 3, 1)
        LDW(1, 7)
        // GENERATED STEADY0 END
        // GENERATED STEADY1 BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(1, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 1, 3, 1)
        MFMA(2, 0, 0)
        VMC(31) WRA(0, 0)
        MFMA(2, 1, 0)
        VMC(30) WRA(0, 1)
        MFMA(2, 2, 0)
        VMC(29) WRA(0, 2)
        MFMA(2, 3, 0)
        VMC(28) WRA(0, 3)
        MFMA(3, 0, 0)
        VMC(27) WRA(0, 4)
        MFMA(3, 1, 0)
        VMC(26) WRA(0, 5)
        MFMA(3, 2, 0)
        KSTEP() LDA(0, 0)
        MFMA(3, 3, 0)
        LDA(0, 1)
        LGKM(6)
        MFMA(0, 0, 1)
        RDW(1, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 2, 3, 0)
        MFMA(2, 0, 1)
        VMC(27) WRA(0, 6)
        MFMA(2, 1, 1)
        VMC(26) WRA(0, 7)
        MFMA(2, 2, 1)
        VMC(25) WRW(0, 0)
        MFMA(2, 3, 1)
        VMC(24) WRW(0, 1)
        MFMA(3, 0, 1)
        VMC(23) WRW(0, 2)
        MFMA(3, 1, 1)
        LDA(0, 2)
        MFMA(3, 2, 1)
        LDA(0, 3)
        MFMA(3, 3, 1)
        LDA(0, 4)
        LGKM(5)
        MFMA(0, 0, 0)
        RDW(1, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 3, 3, 1)
        MFMA(2, 0, 0)
        VMC(25) WRW(0, 3)
        MFMA(2, 1, 0)
        VMC(24) WRW(0, 4)
        MFMA(2, 2, 0)
        VMC(23) WRW(0, 5)
        MFMA(2, 3, 0)
        VMC(22) WRW(0, 6)
        MFMA(3, 0, 0)
        VMC(21) WRW(0, 7)
        MFMA(3, 1, 0)
        LDA(0, 5)
        MFMA(3, 2, 0)
        LDA(0, 6)
        MFMA(3, 3, 0)
        LDA(0, 7)
        LGKM(0) BARRIER()
        LGKM(5)
        MFMA(0, 0, 1)
        RDW(0, 0, 0, 0)
        MFMA(0, 1, 1)
        RDA(0, 0, 0, 0)
        MFMA(0, 2, 1)
        RDA(0, 0, 1, 0)
        MFMA(0, 3, 1)
        RDA(0, 0, 2, 0)
        MFMA(1, 0, 1)
        RDA(0, 0, 3, 0)
        MFMA(1, 1, 1)
        RDW(0, 0, 1, 0)
        MFMA(1, 2, 1)
        RDW(0, 0, 2, 0)
        MFMA(1, 3, 1)
        RDW(0, 0, 3, 0)
        MFMA(2, 0, 1)
        LDW(0, 0)
        MFMA(2, 1, 1)
        LDW(0, 1)
        MFMA(2, 2, 1)
        LDW(0, 2)
        MFMA(2, 3, 1)
        LDW(0, 3)
        MFMA(3, 0, 1)
        LDW(0, 4)
        MFMA(3, 1, 1)
        LDW(0, 5)
        MFMA(3, 2, 1)
        LDW(0, 6)
        MFMA(3, 3, 1)
        LDW(0, 7)
        // GENERATED STEADY1 END
    }
    {
        // tile nk - 4 (steady, buffer 0), then the three tiles that request nothing further
        // GENERATED STEADY0B BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(0, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 1, 3, 1)
        MFMA(2, 0, 0)
        VMC(31) WRA(1, 0)
        MFMA(2, 1, 0)
        VMC(30) WRA(1, 1)
        MFMA(2, 2, 0)
        VMC(29) WRA(1, 2)
        MFMA(2, 3, 0)
        VMC(28) WRA(1, 3)
        MFMA(3, 0, 0)
        VMC(27) WRA(1, 4)
        MFMA(3, 1, 0)
        VMC(26) WRA(1, 5)
        MFMA(3, 2, 0)
        KSTEP() LDA(1, 0)
        MFMA(3, 3, 0)
        LDA(1, 1)
        LGKM(6)
        MFMA(0, 0, 1)
        RDW(0, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(0, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(0, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(0, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(0, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(0, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(0, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(0, 2, 3, 0)
        MFMA(2, 0, 1)
        VMC(27) WRA(1, 6)
        MFMA(2, 1, 1)
        VMC(26) WRA(1, 7)
        MFMA(2, 2, 1)
        VMC(25) WRW(1, 0)
        MFMA(2, 3, 1)
        VMC(24) WRW(1, 1)
        MFMA(3, 0, 1)
        VMC(23) WRW(1, 2)
        MFMA(3, 1, 1)
        LDA(1, 2)
        MFMA(3, 2, 1)
        LDA(1, 3)
        MFMA(3, 3, 1)
        LDA(1, 4)
        LGKM(5)
        MFMA(0, 0, 0)
        RDW(0, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 3, 3, 1)
        MFMA(2, 0, 0)
        VMC(25) WRW(1, 3)
        MFMA(2, 1, 0)
        VMC(24) WRW(1, 4)
        MFMA(2, 2, 0)
        VMC(23) WRW(1, 5)
        MFMA(2, 3, 0)
        VMC(22) WRW(1, 6)
        MFMA(3, 0, 0)
        VMC(21) WRW(1, 7)
        MFMA(3, 1, 0)
        LDA(1, 5)
        MFMA(3, 2, 0)
        LDA(1, 6)
        MFMA(3, 3, 0)
        LDA(1, 7)
        LGKM(0) BARRIER()
        LGKM(5)
        MFMA(0, 0, 1)
        RDW(1, 0, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 0, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 0, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 0, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 0, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 0, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 0, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 0, 3, 0)
        MFMA(2, 0, 1)
        LDW(1, 0)
        MFMA(2, 1, 1)
        LDW(1, 1)
        MFMA(2, 2, 1)
        LDW(1, 2)
        MFMA(2, 3, 1)
        LDW(1, 3)
        MFMA(3, 0, 1)
        LDW(1, 4)
        MFMA(3, 1, 1)
        LDW(1, 5)
        MFMA(3, 2, 1)
        LDW(1, 6)
        MFMA(3, 3, 1)
        LDW(1, 7)
        // GENERATED STEADY0B END
        // GENERATED TAIL3 BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(1, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 1, 3, 1)
        MFMA(2, 0, 0)
        VMC(31) WRA(0, 0)
        MFMA(2, 1, 0)
        VMC(30) WRA(0, 1)
        MFMA(2, 2, 0)
        VMC(29) WRA(0, 2)
        MFMA(2, 3, 0)
        VMC(28) WRA(0, 3)
        MFMA(3, 0, 0)
        VMC(27) WRA(0, 4)
        MFMA(3, 1, 0)
        VMC(26) WRA(0, 5)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(6)
        MFMA(0, 0, 1)
        RDW(1, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 2, 3, 0)
        MFMA(2, 0, 1)
        VMC(25) WRA(0, 6)
        MFMA(2, 1, 1)
        VMC(24) WRA(0, 7)
        MFMA(2, 2, 1)
        VMC(23) WRW(0, 0)
        MFMA(2, 3, 1)
        VMC(22) WRW(0, 1)
        MFMA(3, 0, 1)
        VMC(21) WRW(0, 2)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        LGKM(5)
        MFMA(0, 0, 0)
        RDW(1, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 3, 3, 1)
        MFMA(2, 0, 0)
        VMC(20) WRW(0, 3)
        MFMA(2, 1, 0)
        VMC(19) WRW(0, 4)
        MFMA(2, 2, 0)
        VMC(18) WRW(0, 5)
        MFMA(2, 3, 0)
        VMC(17) WRW(0, 6)
        MFMA(3, 0, 0)
        VMC(16) WRW(0, 7)
        MFMA(3, 1, 0)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(0) BARRIER()
        LGKM(5)
        MFMA(0, 0, 1)
        RDW(0, 0, 0, 0)
        MFMA(0, 1, 1)
        RDA(0, 0, 0, 0)
        MFMA(0, 2, 1)
        RDA(0, 0, 1, 0)
        MFMA(0, 3, 1)
        RDA(0, 0, 2, 0)
        MFMA(1, 0, 1)
        RDA(0, 0, 3, 0)
        MFMA(1, 1, 1)
        RDW(0, 0, 1, 0)
        MFMA(1, 2, 1)
        RDW(0, 0, 2, 0)
        MFMA(1, 3, 1)
        RDW(0, 0, 3, 0)
        MFMA(2, 0, 1)
        MFMA(2, 1, 1)
        MFMA(2, 2, 1)
        MFMA(2, 3, 1)
        MFMA(3, 0, 1)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        // GENERATED TAIL3 END
        // GENERATED TAIL2 BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(0, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 1, 3, 1)
        MFMA(2, 0, 0)
        VMC(15) WRA(1, 0)
        MFMA(2, 1, 0)
        VMC(14) WRA(1, 1)
        MFMA(2, 2, 0)
        VMC(13) WRA(1, 2)
        MFMA(2, 3, 0)
        VMC(12) WRA(1, 3)
        MFMA(3, 0, 0)
        VMC(11) WRA(1, 4)
        MFMA(3, 1, 0)
        VMC(10) WRA(1, 5)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(6)
        MFMA(0, 0, 1)
        RDW(0, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(0, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(0, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(0, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(0, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(0, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(0, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(0, 2, 3, 0)
        MFMA(2, 0, 1)
        VMC(9) WRA(1, 6)
        MFMA(2, 1, 1)
        VMC(8) WRA(1, 7)
        MFMA(2, 2, 1)
        VMC(7) WRW(1, 0)
        MFMA(2, 3, 1)
        VMC(6) WRW(1, 1)
        MFMA(3, 0, 1)
        VMC(5) WRW(1, 2)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        LGKM(5)
        MFMA(0, 0, 0)
        RDW(0, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(0, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(0, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(0, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(0, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(0, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(0, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(0, 3, 3, 1)
        MFMA(2, 0, 0)
        VMC(4) WRW(1, 3)
        MFMA(2, 1, 0)
        VMC(3) WRW(1, 4)
        MFMA(2, 2, 0)
        VMC(2) WRW(1, 5)
        MFMA(2, 3, 0)
        VMC(1) WRW(1, 6)
        MFMA(3, 0, 0)
        VMC(0) WRW(1, 7)
        MFMA(3, 1, 0)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(0) BARRIER()
        LGKM(5)
        MFMA(0, 0, 1)
        RDW(1, 0, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 0, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 0, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 0, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 0, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 0, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 0, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 0, 3, 0)
        MFMA(2, 0, 1)
        MFMA(2, 1, 1)
        MFMA(2, 2, 1)
        MFMA(2, 3, 1)
        MFMA(3, 0, 1)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        // GENERATED TAIL2 END
        // GENERATED TAIL1 BEGIN (lab/gen_gemm4w_body.py)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(1, 1, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 1, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 1, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 1, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 1, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 1, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 1, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 1, 3, 1)
        MFMA(2, 0, 0)
        MFMA(2, 1, 0)
        MFMA(2, 2, 0)
        MFMA(2, 3, 0)
        MFMA(3, 0, 0)
        MFMA(3, 1, 0)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(0)
        MFMA(0, 0, 1)
        RDW(1, 2, 0, 0)
        MFMA(0, 1, 1)
        RDA(1, 2, 0, 0)
        MFMA(0, 2, 1)
        RDA(1, 2, 1, 0)
        MFMA(0, 3, 1)
        RDA(1, 2, 2, 0)
        MFMA(1, 0, 1)
        RDA(1, 2, 3, 0)
        MFMA(1, 1, 1)
        RDW(1, 2, 1, 0)
        MFMA(1, 2, 1)
        RDW(1, 2, 2, 0)
        MFMA(1, 3, 1)
        RDW(1, 2, 3, 0)
        MFMA(2, 0, 1)
        MFMA(2, 1, 1)
        MFMA(2, 2, 1)
        MFMA(2, 3, 1)
        MFMA(3, 0, 1)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        LGKM(0)
        MFMA(0, 0, 0)
        RDW(1, 3, 0, 1)
        MFMA(0, 1, 0)
        RDA(1, 3, 0, 1)
        MFMA(0, 2, 0)
        RDA(1, 3, 1, 1)
        MFMA(0, 3, 0)
        RDA(1, 3, 2, 1)
        MFMA(1, 0, 0)
        RDA(1, 3, 3, 1)
        MFMA(1, 1, 0)
        RDW(1, 3, 1, 1)
        MFMA(1, 2, 0)
        RDW(1, 3, 2, 1)
        MFMA(1, 3, 0)
        RDW(1, 3, 3, 1)
        MFMA(2, 0, 0)
        MFMA(2, 1, 0)
        MFMA(2, 2, 0)
        MFMA(2, 3, 0)
        MFMA(3, 0, 0)
        MFMA(3, 1, 0)
        MFMA(3, 2, 0)
        MFMA(3, 3, 0)
        LGKM(0)
        MFMA(0, 0, 1)
        MFMA(0, 1, 1)
        MFMA(0, 2, 1)
        MFMA(0, 3, 1)
        MFMA(1, 0, 1)
        MFMA(1, 1, 1)
        MFMA(1, 2, 1)
        MFMA(1, 3, 1)
        MFMA(2, 0, 1)
        MFMA(2, 1, 1)
        MFMA(2, 2, 1)
        MFMA(2, 3, 1)
        MFMA(3, 0, 1)
        MFMA(3, 1, 1)
        MFMA(3, 2, 1)
        MFMA(3, 3, 1)
        // GENERATED TAIL1 END
    }
#undef KSTEP

    if (STAMP) {
        const unsigned t_loop1 = (unsigned)__builtin_amdgcn_s_memtime();
        if (lane == 0) { dbg[(blockIdx.x * 4 + wid) * 2] = t_loop1 - t_loop0; dbg[(blockIdx.x * 4 + wid) * 2 + 1] = (unsigned)nk; }
    }
    // ---- epilogue: C^T blocks -> bf16 -> 16-byte stores (a lane pair exchanges halves: 8 consecutive columns each)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const int row_base = m0 + wr * 128 + frow, col_base = n0 + wc * 128 + 8 * hi;
#define STORE_BLOCK(JN, IM)                                                                                               \
    {                                                                                                                     \
        float c[16];                                                                                                      \
        asm volatile("v_accvgpr_read_b32 %0, a[%16]\n\tv_accvgpr_read_b32 %1, a[%17]\n\tv_accvgpr_read_b32 %2, a[%18]\n\tv_accvgpr_read_b32 %3, a[%19]\n\t" \
                     "v_accvgpr_read_b32 %4, a[%20]\n\tv_accvgpr_read_b32 %5, a[%21]\n\tv_accvgpr_read_b32 %6, a[%22]\n\tv_accvgpr_read_b32 %7, a[%23]\n\t" \
                     "v_accvgpr_read_b32 %8, a[%24]\n\tv_accvgpr_read_b32 %9, a[%25]\n\tv_accvgpr_read_b32 %10, a[%26]\n\tv_accvgpr_read_b32 %11, a[%27]\n\t" \
                     "v_accvgpr_read_b32 %12, a[%28]\n\tv_accvgpr_read_b32 %13, a[%29]\n\tv_accvgpr_read_b32 %14, a[%30]\n\tv_accvgpr_read_b32 %15, a[%31]" \
                     : "=v"(c[0]), "=v"(c[1]), "=v"(c[2]), "=v"(c[3]), "=v"(c[4]), "=v"(c[5]), "=v"(c[6]), "=v"(c[7]), "=v"(c[8]), "=v"(c[9]),   \
                       "=v"(c[10]), "=v"(c[11]), "=v"(c[12]), "=v"(c[13]), "=v"(c[14]), "=v"(c[15])                                              \
                     : "n"(16 * (4 * (JN) + (IM)) + 0), "n"(16 * (4 * (JN) + (IM)) + 1), "n"(16 * (4 * (JN) + (IM)) + 2), "n"(16 * (4 * (JN) + (IM)) + 3),     \
                       "n"(16 * (4 * (JN) + (IM)) + 4), "n"(16 * (4 * (JN) + (IM)) + 5), "n"(16 * (4 * (JN) + (IM)) + 6), "n"(16 * (4 * (JN) + (IM)) + 7),     \
                       "n"(16 * (4 * (JN) + (IM)) + 8), "n"(16 * (4 * (JN) + (IM)) + 9), "n"(16 * (4 * (JN) + (IM)) + 10), "n"(16 * (4 * (JN) + (IM)) + 11),   \
                       "n"(16 * (4 * (JN) + (IM)) + 12), "n"(16 * (4 * (JN) + (IM)) + 13), "n"(16 * (4 * (JN) + (IM)) + 14), "n"(16 * (4 * (JN) + (IM)) + 15)); \
        const int row = row_base + 32 * (IM);                                                                             \
        _Pragma("unroll") for (int q8 = 0; q8 < 2; ++q8) {                                                                \
            unsigned a0 = pack2(c[8 * q8 + 0], c[8 * q8 + 1]), a1 = pack2(c[8 * q8 + 2], c[8 * q8 + 3]);                  \
            unsigned b0 = pack2(c[8 * q8 + 4], c[8 * q8 + 5]), b1 = pack2(c[8 * q8 + 6], c[8 * q8 + 7]);                  \
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);                                       \
            const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);                                       \
            u32x4_t w;                                                                                                    \
            w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];                                                       \
            const int col = col_base + 32 * (JN) + 16 * q8;                                                               \
            if (row < M && col < N) *(u32x4_t*)(C + (long long)row * ldc + col) = w;                                      \
        }                                                                                                                 \
    }
    STORE_BLOCK(0, 0) STORE_BLOCK(0, 1) STORE_BLOCK(0, 2) STORE_BLOCK(0, 3) STORE_BLOCK(1, 0) STORE_BLOCK(1, 1) STORE_BLOCK(1, 2) STORE_BLOCK(1, 3)
    STORE_BLOCK(2, 0) STORE_BLOCK(2, 1) STORE_BLOCK(2, 2) STORE_BLOCK(2, 3) STORE_BLOCK(3, 0) STORE_BLOCK(3, 1) STORE_BLOCK(3, 2) STORE_BLOCK(3, 3)
#undef STORE_BLOCK
}

__global__ void fill_kernel(bf16_t* p, long long n, unsigned seed, float scale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (bf16_t)(((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
    }
}

}  // namespace

int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* what; };
    std::vector<Shape> shapes = {{30976, 7680, 1920, "MLP up  (N = 7680, K = 1920)"}, {30976, 1920, 7680, "MLP down (N = 1920, K = 7680)"},
                                 {30976, 5760, 1920, "K|V|Q   (N = 5760, K = 1920)"}, {30976, 1920, 1920, "attn out (N = 1920, K = 1920)"},
                                 {8192, 8192, 8192, "8192^3"}};
    if (argc == 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), "command line"}};
    void* lib = dlopen("pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
    if (!lib) lib = dlopen("../pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
    typedef int (*gemm_fn)(const pf_gemm_desc*, pf_stream_t);
    typedef long long (*ws_fn)(int, int, int, int);
    gemm_fn lib_gemm = lib ? (gemm_fn)dlsym(lib, "pf_gemm_bf16") : nullptr;
    ws_fn lib_ws = lib ? (ws_fn)dlsym(lib, "pf_gemm_workspace_bytes") : nullptr;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const int M = s.M, N = s.N, K = s.K;
        if (K % 128 || K < 384) { printf("K must be a multiple of 128, >= 384\n"); return 1; }
        bf16_t *A, *W, *C, *Cref;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
        CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&Cref, (size_t)M * N * 2));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, A, (long long)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, W, (long long)N * K, 2u, 0.05f);
        CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, st));
        const int T = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
        unsigned* dbg;
        CK(hipMalloc(&dbg, (size_t)T * 4 * 2 * 4));
        auto run4w = [&] { hipLaunchKernelGGL((gemm4w_kernel<0, 0>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); };
        const double flops = 2.0 * M * N * K;
        auto time_it = [&](auto&& f, int reps) {
            for (int i = 0; i < 3; ++i) f();
            CK(hipStreamSynchronize(st));
            std::vector<float> ms;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) f();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t / reps);
            }
            std::sort(ms.begin(), ms.end());
            return ms[2];
        };
        printf("%s: M = %d\n", s.what, M);
        float t_lib = 0;
        if (lib_gemm) {
            pf_gemm_desc d = {};
            d.A = A; d.W = W; d.C = Cref; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldw = K; d.ldc = N; d.batch = 1; d.gelu_from = -1;
            void* ws = nullptr;
            const long long wsb = lib_ws ? lib_ws(M, 1, N, K) : 0;
            if (wsb > 0) { CK(hipMalloc(&ws, wsb)); d.workspace = ws; d.workspace_bytes = wsb; }
            if (lib_gemm(&d, st)) { printf("pf_gemm_bf16 failed\n"); return 1; }
            t_lib = time_it([&] { lib_gemm(&d, st); }, 10);
            printf("   pf_gemm_bf16 (libpyflow_hip.so, gemm8p)      %.3f ms  %6.0f TFLOP/s\n", t_lib, flops / t_lib / 1e9);
            if (ws) CK(hipFree(ws));
        }
        run4w();
        CK(hipStreamSynchronize(st));
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { printf("gemm4w launch failed: %s\n", hipGetErrorString(le)); return 1; }
        const float t4 = time_it(run4w, 10);
        // compare with the library's result (different MFMA shape and summation order: bf16 rounding of nearly equal sums)
        double rel = -1, mx = 0;
        if (lib_gemm) {
            const size_t n = (size_t)M * N;
            std::vector<unsigned short> h1(n), h2(n);
            CK(hipMemcpy(h1.data(), C, n * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), Cref, n * 2, hipMemcpyDeviceToHost));
            double num = 0, den = 0;
            for (size_t i = 0; i < n; ++i) {
                union { unsigned u; float f; } a, b;
                a.u = (unsigned)h1[i] << 16; b.u = (unsigned)h2[i] << 16;
                const double dlt = (double)a.f - b.f;
                num += dlt * dlt; den += (double)b.f * b.f;
                if (std::fabs(dlt) > mx || dlt != dlt) mx = (dlt != dlt) ? 1e30 : std::fabs(dlt);
            }
            rel = std::sqrt(num / (den + 1e-30));
        }
        printf("   gemm4w (4 waves, hand-placed, plain epilogue) %.3f ms  %6.0f TFLOP/s   vs library result: rel-L2 %.2e max-abs %.3g   (%d tiles = %.2f rounds)\n",
               t4, flops / t4 / 1e9, rel, mx, T, T / 256.0);
        {
            const char* nm[5] = {"", ", NO LDS writes", ", NO global loads", ", NO fragment reads", ", NO barrier"};
            printf("   gemm4w main loop, cycles per K-tile and wave (s_memtime; the matrix pipe needs 2048):");
            for (int v = 0; v < 5; ++v) {
                switch (v) {
                    case 0: hipLaunchKernelGGL((gemm4w_kernel<1, 0>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 1: hipLaunchKernelGGL((gemm4w_kernel<1, 1>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 2: hipLaunchKernelGGL((gemm4w_kernel<1, 2>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 3: hipLaunchKernelGGL((gemm4w_kernel<1, 3>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    default: hipLaunchKernelGGL((gemm4w_kernel<1, 4>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                }
                CK(hipStreamSynchronize(st));
                std::vector<unsigned> hd((size_t)T * 8);
                CK(hipMemcpy(hd.data(), dbg, hd.size() * 4, hipMemcpyDeviceToHost));
                double cyc = 0, kt = 0;
                for (size_t w = 0; w < (size_t)T * 4; ++w) { cyc += hd[2 * w]; kt += hd[2 * w + 1]; }
                printf("  %.0f%s", cyc / kt, nm[v]);
            }
            printf("\n");
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(Cref)); CK(hipFree(dbg));
    }
    return 0;
}
