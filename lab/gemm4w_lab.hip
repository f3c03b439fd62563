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
constexpr int BUF = 65536;                 // one LDS buffer = 32 KiB of A rows + 32 KiB of W rows

// VAR (timing experiments, wrong results): 1 no LDS writes, 2 no global loads, 3 no fragment reads, 4 no barrier
// PERSIST: one workgroup per CU walks a list of output tiles (XCD x owns a contiguous chunk of the tile order, its workgroups take
// the chunk's tiles round-robin, as gemm8p does); the operand stream -- global loads three K-tiles ahead of the MFMAs, LDS writes
// one ahead -- runs on ACROSS tile boundaries, so that only the epilogue itself (accumulator reads, stores) is not under MFMAs.
template <int STAMP, int VAR = 0, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                        int M, int N, int K, int lda, int ldw, int ldc, unsigned* dbg, int stagger = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a255");                                        // the whole accumulator file is in use
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int T = tiles_m * tiles_n;
    const int nk = K / BK;
    int n_my = 1, n_max = 1, tile_first = 0, tile_step = 0;
    if (PERSIST) {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3, nslot = (nwg - xcd + 7) >> 3;
        const int cq = T >> 3, cr = T & 7, cs = xcd * cq + min(xcd, cr), clen = cq + (xcd < cr ? 1 : 0);
        n_my = slot < clen ? (clen - slot + nslot - 1) / nslot : 0;
        n_max = (clen + nslot - 1) / nslot;
        tile_first = cs + slot;
        tile_step = nslot;
    } else {
        tile_first = xcd_remap(blockIdx.x, T);
    }
    if (n_my == 0) return;
    if (PERSIST && stagger > 0 && n_my < n_max) {          // (only workgroups with a tile less than the busiest: their wait is free)
        // desynchronise the workgroups: all 256 of them reaching their epilogue together write 32 MB in one burst (12 000 cycles
        // of stores per tile at ~3 TB/s aggregate); phase (slot % 8) / 8 of a tile time, waited out here
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned)stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    // tile order: groups of 4 row tiles x all column tiles (neighbouring workgroups share operand panels)
    auto tile_mn = [&](int seq, int& m0, int& n0) {
        const int t = tile_first + seq * tile_step;
        const int GM = 4, gsz = GM * tiles_n, grp = t / gsz, first_m = grp * GM, gm = min(tiles_m - first_m, GM);
        const int r_in = t - grp * gsz, tn = r_in / gm, tm = first_m + (r_in - tn * gm);
        m0 = tm * BM;
        n0 = tn * BN;
    };

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
    i32x4_t srdA, srdW;
    unsigned voffA[8], voffW[8];
    // the LOAD stream's position: output tile ld_seq, K-tile ld_kt of it (it runs three K-tiles ahead of the MFMAs)
    int ld_seq = 0, ld_kt = 0;
    auto setup_load_tile = [&](int seq) {
        int m0, n0;
        tile_mn(seq, m0, n0);
        srdA = make_srd(A + (long long)m0 * lda);
        srdW = make_srd(W + (long long)n0 * ldw);
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (8 * wid + i) * 8 + (ln >> 3);
            voffA[i] = (unsigned)(min(m0 + row, M - 1) - m0) * (unsigned)(lda * 2) + (ln & 7) * 16;
            voffW[i] = (unsigned)(min(n0 + row, N - 1) - n0) * (unsigned)(ldw * 2) + (ln & 7) * 16;
        }
    };
    setup_load_tile(0);
    // ---- LDS side.  Writes: row = 8 j + (lane >> 3) with j = 8 w + i: chunk ^ ((row >> 1) & 7) = (lane & 7) ^ (4 (i & 1) + (lane >> 4))
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    // LDS map: A rows of buffer 0 | A rows of buffer 1 | W rows of buffer 0 | W rows of buffer 1 (32 KiB each): the buffer index is
    // part of the instructions' 16-bit offsets, not of the address registers
    unsigned wrA[2], wrW[2];               // [i & 1]; + b * 32768 + i * 1024 as the instruction's offset
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        wrA[par] = lds0 + wid * 8192 + (lane >> 3) * 128 + ((((lane & 7) ^ (4 * par + (lane >> 4))) & 7) << 4);
        wrW[par] = wrA[par] + 65536;
    }
    // Fragment reads (32x32x16 operand: lane -> row lane & 31, 16-byte chunk 2 ks + (lane >> 5) of the row's 128 bytes)
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;
    unsigned rdA[4], rdW[4];               // [k-step]; + b * 32768 + block * 4096 as the instruction's offset
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const unsigned f = (unsigned)(frow * 128 + (((2 * ks + hi) ^ swz) << 4));
        rdA[ks] = lds0 + wr * 16384 + f;
        rdW[ks] = lds0 + 65536 + wc * 16384 + f;
    }

    bf16x8_t fa[2][4], fb[2][4];           // fragment sets (k-step parity) x blocks
    u32x4_t st[2][16];                     // staging sets (tile parity) x loads (0..7 A rows, 8..15 W rows)
    unsigned kofs = 0;                     // byte offset of the K-tile the NEXT global loads fetch

// every accumulator-file register, as a clobber list: the MFMA statements below name their accumulators in the asm text, which
// the compiler cannot see -- with this list on each of them it cannot keep a value of its own in the accumulator file across
// any of them (without it hipcc parked a zero vector and two spills in a0..a7 of the persistent instantiation).  On the MFMAs of
// the main loop the list costs an s_nop in front of every one of them (+13 % cycles per K-tile): there it is left off, the list
// stays on the statements of the epilogue, and gen_gemm4w_body.py --check is what guards the loop
#define ALL_AGPRS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
    "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
    "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
    "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
    "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
    "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
    "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
    "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
    "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", \
    "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
    "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
    "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", \
    "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
    "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");
#define VMC(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");
#define BARRIER() if (VAR != 4) __builtin_amdgcn_s_barrier();
#define MFMA(JN, IM, FS)                                                                                                  \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(16 * (4 * (JN) + (IM))), "n"(16 * (4 * (JN) + (IM)) + 15), \
                 "v"(fb[FS][JN]), "v"(fa[FS][IM]));
#define RDA(BUFI, KS, BLK, FS) if (VAR != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[FS][BLK]) : "v"(rdA[KS]), "n"((BUFI) * 32768 + (BLK) * 4096));
#define RDW(BUFI, KS, BLK, FS) if (VAR != 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[FS][BLK]) : "v"(rdW[KS]), "n"((BUFI) * 32768 + (BLK) * 4096));
#define WRA(BUFI, I) if (VAR != 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wrA[(I) & 1]), "v"(st[BUFI][I]), "n"((BUFI) * 32768 + (I) * 1024) : "memory");
#define WRW(BUFI, I) if (VAR != 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(wrW[(I) & 1]), "v"(st[BUFI][8 + (I)]), "n"((BUFI) * 32768 + (I) * 1024) : "memory");
#define LDA(SET, I) if (VAR != 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[SET][I]) : "v"(voffA[I]), "s"(srdA), "s"(kofs) : "memory");
#define LDW(SET, I) if (VAR != 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(st[SET][8 + (I)]) : "v"(voffW[I]), "s"(srdW), "s"(kofs) : "memory");

    // ---- accumulators = 0 (MFMAs of zero operands write the accumulator file without a register move)
    // (the zero operand is rebuilt where it is needed: four registers kept alive across the loop were what tipped hipcc into
    // parking a value in a0..a3 -- the accumulator file belongs to the statements below, see gen_gemm4w_body.py --check)
#define ZERO(X) asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %2, 0" ::"n"(16 * (X)), "n"(16 * (X) + 15), "v"(zfrag) : ALL_AGPRS);
#define ZERO_ALL()                                                                                                              \
    {                                                                                                                           \
        bf16x8_t zfrag;                                                                                                         \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) zfrag[e] = (bf16_t)0.0f;                                                  \
        asm volatile("" : "+v"(zfrag));                                                                                         \
        ZERO(0) ZERO(1) ZERO(2) ZERO(3) ZERO(4) ZERO(5) ZERO(6) ZERO(7) ZERO(8) ZERO(9) ZERO(10) ZERO(11) ZERO(12) ZERO(13) ZERO(14) ZERO(15) \
    }
    ZERO_ALL()
    // ---- prologue: tiles 0 and 1 into the staging sets, tile 0 into LDS buffer 0, its first fragments
    // kofs = byte offset of the K-tile the next global loads fetch (tile 0, 1, 2 in the prologue, then tile kt + 3)
#define KSTEP()                                                                                          \
    {                                                                                                    \
        kofs += 128;                                                                                     \
        if (++ld_kt == nk) { ld_kt = 0; kofs = 0; if (++ld_seq < n_my) setup_load_tile(ld_seq); }          \
    }
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
    // ---- epilogue: C^T blocks -> bf16 -> 16-byte stores (a lane pair exchanges halves: 8 consecutive columns each)
    auto epilogue = [&](int seq) {
    int m0, n0;
    tile_mn(seq, m0, n0);
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
                       "n"(16 * (4 * (JN) + (IM)) + 12), "n"(16 * (4 * (JN) + (IM)) + 13), "n"(16 * (4 * (JN) + (IM)) + 14), "n"(16 * (4 * (JN) + (IM)) + 15)  \
                     : ALL_AGPRS);                                                                                        \
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
    ZERO_ALL()
    };
    // ---- main loop: nk even, >= 4.  Stream K-tile g lives in LDS buffer g & 1 and came through staging set g & 1 (nk even: every
    //      output tile starts in buffer 0)
    const unsigned t_loop0 = STAMP ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    for (int seq = 0; seq < n_my; ++seq) {
        const bool last = seq == n_my - 1;
        const int jend = last ? nk - 4 : nk;
        int j = 0;
        if (seq > 0 && jend >= 2) {
        // the two K-tiles behind an epilogue (vmcnt values count its 32 stores)
        // GENERATED POST0 BEGIN (lab/gen_gemm4w_body.py)
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
        VMC(63) WRA(1, 0)
        MFMA(2, 1, 0)
        VMC(62) WRA(1, 1)
        MFMA(2, 2, 0)
        VMC(61) WRA(1, 2)
        MFMA(2, 3, 0)
        VMC(60) WRA(1, 3)
        MFMA(3, 0, 0)
        VMC(59) WRA(1, 4)
        MFMA(3, 1, 0)
        VMC(58) WRA(1, 5)
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
        VMC(59) WRA(1, 6)
        MFMA(2, 1, 1)
        VMC(58) WRA(1, 7)
        MFMA(2, 2, 1)
        VMC(57) WRW(1, 0)
        MFMA(2, 3, 1)
        VMC(56) WRW(1, 1)
        MFMA(3, 0, 1)
        VMC(55) WRW(1, 2)
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
        VMC(57) WRW(1, 3)
        MFMA(2, 1, 0)
        VMC(56) WRW(1, 4)
        MFMA(2, 2, 0)
        VMC(55) WRW(1, 5)
        MFMA(2, 3, 0)
        VMC(54) WRW(1, 6)
        MFMA(3, 0, 0)
        VMC(53) WRW(1, 7)
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
        // GENERATED POST0 END
        // GENERATED POST1 BEGIN (lab/gen_gemm4w_body.py)
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
        VMC(63) WRA(0, 0)
        MFMA(2, 1, 0)
        VMC(62) WRA(0, 1)
        MFMA(2, 2, 0)
        VMC(61) WRA(0, 2)
        MFMA(2, 3, 0)
        VMC(60) WRA(0, 3)
        MFMA(3, 0, 0)
        VMC(59) WRA(0, 4)
        MFMA(3, 1, 0)
        VMC(58) WRA(0, 5)
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
        VMC(59) WRA(0, 6)
        MFMA(2, 1, 1)
        VMC(58) WRA(0, 7)
        MFMA(2, 2, 1)
        VMC(57) WRW(0, 0)
        MFMA(2, 3, 1)
        VMC(56) WRW(0, 1)
        MFMA(3, 0, 1)
        VMC(55) WRW(0, 2)
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
        VMC(57) WRW(0, 3)
        MFMA(2, 1, 0)
        VMC(56) WRW(0, 4)
        MFMA(2, 2, 0)
        VMC(55) WRW(0, 5)
        MFMA(2, 3, 0)
        VMC(54) WRW(0, 6)
        MFMA(3, 0, 0)
        VMC(53) WRW(0, 7)
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
        // GENERATED POST1 END
            j = 2;
        }
        for (; j < jend; j += 2) {
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
        if (last) {
        // the stream's last four K-tiles: one more steady one, then the three that request nothing further
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
        epilogue(seq);
    }
    if (STAMP) {
        const unsigned t_loop1 = (unsigned)__builtin_amdgcn_s_memtime();
        if (lane == 0) { dbg[(blockIdx.x * 4 + wid) * 2] = t_loop1 - t_loop0; dbg[(blockIdx.x * 4 + wid) * 2 + 1] = (unsigned)(nk * n_my); }
    }
#undef KSTEP

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
    setvbuf(stdout, nullptr, _IONBF, 0);
    struct Shape { int M, N, K; const char* what; };
    std::vector<Shape> shapes = {{30976, 7680, 1920, "MLP up  (N = 7680, K = 1920)"}, {30976, 1920, 7680, "MLP down (N = 1920, K = 7680)"},
                                 {30976, 5760, 1920, "K|V|Q   (N = 5760, K = 1920)"}, {30976, 1920, 1920, "attn out (N = 1920, K = 1920)"},
                                 {8192, 8192, 8192, "8192^3"}};
    if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), "command line"}};
    const int grid_override = argc >= 5 ? atoi(argv[4]) : 0;
    void* lib = getenv("GEMM4W_NO_LIB") ? nullptr : dlopen("pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
    if (!lib && !getenv("GEMM4W_NO_LIB")) lib = dlopen("../pyramid-flow_amd/libpyflow_hip.so", RTLD_NOW);
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
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    CK(hipFuncSetAttribute((const void*)gemm4w_kernel<1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
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
        CK(hipStreamSynchronize(st));
        printf("   [operands filled]\n");
        float t_lib = 0;
        if (lib_gemm) {
            pf_gemm_desc d = {};
            d.A = A; d.W = W; d.C = Cref; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldw = K; d.ldc = N; d.batch = 1; d.gelu_from = -1;
            void* ws = nullptr;
            const long long wsb = lib_ws ? lib_ws(M, 1, N, K) : 0;
            if (wsb > 0) { CK(hipMalloc(&ws, wsb)); d.workspace = ws; d.workspace_bytes = wsb; }
            if (lib_gemm(&d, st)) { printf("pf_gemm_bf16 failed\n"); return 1; }
            CK(hipStreamSynchronize(st));
            printf("   [library call returned]\n");
            t_lib = time_it([&] { lib_gemm(&d, st); }, 10);
            printf("   pf_gemm_bf16 (libpyflow_hip.so, gemm8p)      %.3f ms  %6.0f TFLOP/s\n", t_lib, flops / t_lib / 1e9);
            if (ws) CK(hipFree(ws));
        }
        auto compare = [&](double& rel, double& mx) {
            rel = -1; mx = 0;
            if (!lib_gemm) return;
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
                if (dlt != dlt) mx = 1e30; else if (std::fabs(dlt) > mx) mx = std::fabs(dlt);
            }
            rel = std::sqrt(num / (den + 1e-30));
        };
        printf("   [one launch per tile]\n");
        run4w();
        CK(hipStreamSynchronize(st));
        double rel1, mx1;
        compare(rel1, mx1);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { printf("gemm4w launch failed: %s\n", hipGetErrorString(le)); return 1; }
        const float t4 = time_it(run4w, 10);
        int ncu = 256;
        { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = grid_override ? grid_override : pr.multiProcessorCount; }
        auto run4wp = [&] { hipLaunchKernelGGL((gemm4w_kernel<0, 0, true>), dim3(ncu), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); };
        CK(hipMemsetAsync(C, 0xff, (size_t)M * N * 2, st));
        printf("   [persistent]\n");
        run4wp();
        CK(hipStreamSynchronize(st));
        { hipError_t le2 = hipGetLastError(); if (le2 != hipSuccess) printf("persistent launch failed: %s\n", hipGetErrorString(le2)); }
        double rel, mx;
        compare(rel, mx);
        const float t4p = time_it(run4wp, 10);
        const int stag = (K / BK) * 2600 / 8;
        auto run4ws = [&] { hipLaunchKernelGGL((gemm4w_kernel<0, 0, true>), dim3(ncu), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg, stag); };
        const float t4s = time_it(run4ws, 10);
        printf("   gemm4w, one launch per tile (plain epilogue)    %.3f ms  %6.0f TFLOP/s   vs library result: rel-L2 %.2e max-abs %.3g   (%d tiles = %.2f rounds)\n", t4, flops / t4 / 1e9, rel1, mx1, T, T / 256.0);
        printf("   gemm4w, persistent (%d workgroups)              %.3f ms  %6.0f TFLOP/s   vs library result: rel-L2 %.2e max-abs %.3g\n", ncu, t4p, flops / t4p / 1e9, rel, mx);
        printf("   gemm4w, persistent, starts staggered by slot %% 8    %.3f ms  %6.0f TFLOP/s\n", t4s, flops / t4s / 1e9);
        {
            const char* nm[6] = {"", ", NO LDS writes", ", NO global loads", ", NO fragment reads", ", NO barrier", " persistent (incl. epilogues)"};
            printf("   gemm4w main loop, cycles per K-tile and wave (s_memtime; the matrix pipe needs 2048):");
            for (int v = 0; v < 6; ++v) {
                if (v == 1) continue;         // (loads in flight into registers nobody reads: the compiler reuses them, e.g. for store addresses)
                if (v == 5) CK(hipMemsetAsync(dbg, 0, (size_t)T * 4 * 2 * 4, st));
                switch (v) {
                    case 0: hipLaunchKernelGGL((gemm4w_kernel<1, 0>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 1: hipLaunchKernelGGL((gemm4w_kernel<1, 1>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 2: hipLaunchKernelGGL((gemm4w_kernel<1, 2>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 3: hipLaunchKernelGGL((gemm4w_kernel<1, 3>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    case 4: hipLaunchKernelGGL((gemm4w_kernel<1, 4>), dim3(T), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
                    default: hipLaunchKernelGGL((gemm4w_kernel<1, 0, true>), dim3(ncu), dim3(256), 2 * BUF, st, A, W, C, M, N, K, K, K, N, dbg); break;
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
