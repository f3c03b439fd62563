// bf16 MFMA GEMM, persistent 256 x 256 block tile, v_mfma_f32_16x16x32_bf16, two ping-pong phases per K-tile.
//
// Same contract as gemm.hip / gemm256.hip (C = epi(A . W^T), optional implicit-GEMM conv addressing).  Serves the
// large DiT projections (flux_block.py:756-758, 816-835, 868-872, 914-942) and the wide VAE CausalConv3d layers
// (modeling_causal_conv.py:116-146).  What differs from gemm256.hip, and why:
//   * ONE persistent workgroup per CU walks a list of output tiles; the operand stream (LDS-DMA, 1 KiB pieces) is a
//     single pipeline that runs six 16-KiB units (1.5 K-tiles) ahead of the MFMAs and CROSSES tile boundaries: while a
//     tile's epilogue stores drain, the first K-tile of the next tile is already in LDS.  At K = 1920 (30 K-tiles) the
//     un-overlapped prologue + epilogue of gemm256.hip cost 10-15 % of a tile.
//   * The epilogue never touches LDS memory: the MFMAs compute C^T tiles (operands swapped), so a lane owns 4 consecutive
//     columns of one row per accumulator; the W rows are permuted on their way into LDS (the DMA source address is per
//     lane) such that a lane's two accumulators of a column half are 8 CONSECUTIVE columns -> bias / GELU / gate*x+res
//     on 16-byte register pieces, no barrier, no staging strips.
//   * 16x16x32 MFMAs on a 128 x 64 wave tile (2 x 4 waves).  A K-tile is TWO phases of 32 MFMAs (round 6; rounds 2-5 ran
//     four phases of 16): phase A = load slot {A sub 0, B sub 0, B sub 1: 16 fragment reads, two DMA units} | (A0 x B0),
//     (A0 x B1); phase B = load slot {A sub 1: 8 reads, two DMA units} | (A1 x B1), (A1 x B0) -- one A fragment set and
//     both B sets in registers, as before.  Measured with s_memtime stamps (tools/gemm8p_stamps.py,
//     profiles/r06_gemm8p_stamps.log): 2 920 cycles per K-tile with eight barriers, 2 550 with four, where the matrix pipe
//     needs 2 048 -- an interval costs ~75-125 cycles over its MFMAs whatever its length, so longer intervals waste less.
//   * Two wave groups (the M halves) run one barrier apart: while one group issues its 32 MFMAs the other reads
//     fragments and issues DMA (two barriers per phase).  Slot S = 2 gk + h issues units 2S + 6, 2S + 7; at its end the
//     units the NEXT slot reads have landed: h = 0 -> unit 4gk + 3 (four units = 8 pieces of this wave may be in flight:
//     s_waitcnt vmcnt(8)), h = 1 -> units <= 4gk + 6 (three units: vmcnt(6)).  A unit's LDS region is overwritten by
//     unit + 8, issued in the slot after its last read by group 0 = the interval in which group 1's reads of it complete:
//     every wave therefore waits for its fragment reads BEFORE its slot's barrier (lgkmcnt(0); the load slot is the short
//     side of the interval), so that no read is pending when the barrier releases the other group's DMA issue.
//   * STATIC PRIORITY: the second-dispatched wave group (waves 4-7) runs at s_setprio 1, no per-phase flips
//     (MI355X_MICROARCH.md "two waves per SIMD" item 4): +1-2 % on every DiT shape.
//   * TAIL SPLIT (round 3): the r tiles of an XCD's chunk that do not fill a last round of its nslot workgroups are split
//     along K into sp equal ranges each (sp * r <= nslot), when the caller brings scratch (pf_gemm_desc.workspace,
//     64 MiB): a workgroup's last segment is then a K range of a tail tile, whose raw fp32 sums it parks in its 256-KiB
//     slot, and a second small launch (gemm8p_tail_kernel) adds the parts in part order and applies the epilogue.  The
//     parts of all tail tiles cover the SAME K ranges at the same time: the workgroups of an XCD keep reading operand
//     panels in phase (a stream-K cut of the tail into 32 equal pieces that straddle tile boundaries was measured 5-15 %
//     slower than this on the K = 9600 GEMMs: panels are then fetched from L2 at 32 different K offsets).
//     Over one video the DiT's N = 1920 / K = 7680-9600 GEMMs ran at 0.72 of a whole number of rounds.
//   * EPILOGUE AND THE COMPILER'S WAITS (round 5).  The kernel orders its vector-memory traffic with inline-asm s_waitcnt; hipcc
//     does not see those and guards the first use of every register a global load filled with a wait of its own.  Wherever
//     such a use sits BEHIND the tile's stores that wait is an s_waitcnt vmcnt(0) for the whole store burst (rounds 2-4: the
//     next tile's bias -> every epilogue drained its stores; the residual flavour's second column half -> 16 store round
//     trips).  Rule of this file: every register a global load fills is REDEFINED (asm volatile("" : "+v"(x))) directly behind
//     the asm drain that covers it; tests/test_kernel_isa.py checks the result in the ISA (16 stores in one run, no wait behind
//     them, no scratch).  The two wave groups' epilogues run in the SAME barrier interval (group 0 takes the loop's last
//     barrier first: epi_mode bit 0); waves without epilogue loads drain the DMA queue after their conversions.
//   * COALESCED STORES (round 6).  The stamps put 8-10 k cycles of a tile boundary (13 % of a K = 1920 tile) into the ISSUE of
//     the 16 stores per wave: in the accumulator layout adjacent lanes hold different rows, every store instruction was 64
//     separate 16-byte writes (~270-530 cycles each).  The packed results are lane-permuted (ds_bpermute_b32) so that a quad
//     of lanes writes 64 contiguous bytes, pipelined against the store issue: epilogue_tile's store loop.
//   * GROUPED LAUNCH (round 5, pf_gemm_desc.A2 ...): the tile list may continue with the tiles of a second problem of the same
//     N / K / flavour (the text stream of a double block); per-problem fields are read as p.pr[g] from the kernel arguments.
// LDS: 2 K-tile buffers x (A 256 x 64 + W 256 x 64) bf16 = 128 KiB, XOR-swizzled like gemm256.hip (swizzle on the DMA
// source address and on the ds_read_b128 address).
#include "common.h"
#include "pyflow_hip.h"
#include "gemm_args.h"

using namespace pfgemm;

int pf_gemm8p_mid_split(int tiles, int nk);      // parts per tile of a launch of < one round of tiles (defined at the end)

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 32 KiB per operand per K-tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;    // 64 KiB
constexpr int SMEM = 2 * BUF_BYTES;          // 128 KiB
constexpr int GROUP_M = 4;

#define PF_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PF_BAR()                          \
    do {                                  \
        PF_FENCE();                       \
        __builtin_amdgcn_s_barrier();     \
        PF_FENCE();                       \
    } while (0)

// ---- cycle stamps (LAB BUILD ONLY: -DPF_G8_STAMP=1 | 2; nothing of this exists in the shipping library).
// A wave keeps 64 32-bit s_memtime stamps in the lanes of ONE register (v_writelane) and stores them at the end of the kernel:
// no memory traffic and no added waits inside the loop -- a stamp is issued into an SGPR pair and read behind the
// s_waitcnt lgkmcnt(0).  Mode 1: one stamp per K-tile (release from phase A's barrier = start of its first MFMA phase), 64
// K-tiles from -p.stagger on.  Mode 3: five stamps per tile inside the epilogue.  (Mode 2 -- three stamps per phase of the
// four-phase loop of rounds 2-5 -- went with that loop; its log is profiles/r06_gemm8p_stamps_four_phase.log.)  Read back with pf_lab_gemm8p_stamps.
#ifdef PF_G8_STAMP
__device__ unsigned g8_stamps[256 * 8 * 64];
#define G8_ISSUE(sreg) asm volatile("s_memtime %0" : "=s"(sreg))
#define G8_PUT(sreg, idx)                                                                                         \
    do {                                                                                                          \
        const int ix_ = (idx);                                                                                    \
        unsigned sv_, sh_;          /* both halves are read HERE: the pair stays allocated until s_memtime has written it */ \
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(sv_), "=v"(sh_) : "s"((unsigned)(sreg)), "s"((unsigned)((sreg) >> 32))); \
        stampv = lane == ix_ ? sv_ : stampv;                                                                      \
    } while (0)
#endif

struct TileCoord { int b, m0, n0, g; };           // g = problem of a grouped launch (0 / 1)

// How the `clen` tiles of one XCD's chunk are dealt to its `nslot` workgroups: n_full whole rounds, then r tail tiles, each
// split along K over sp workgroups (sp = 1: one whole tile for each of the first r workgroups, the round-2 behaviour):
// workgroup s computes part s % sp of tail tile s / sp, K-tiles [q nk / sp, (q + 1) nk / sp).
// The split pays when a part + the cost of parking and re-adding (a fixed `ov` K-tile periods: the second launch, its
// kernel boundary, the 256-KiB park; + 0.6 per part: the second launch reads r * sp * 8 parts of 256 KiB) is shorter than
// a whole tile.  Same function on the host (launch geometry of the second kernel) and in both kernels.
struct TailPlan { int n_full, r, sp; };
__host__ __device__ inline TailPlan tail_plan(int clen, int nslot, int nk, int ov) {
    TailPlan t{clen / nslot, clen % nslot, 1};
    if (ov < 0 || t.r == 0 || 2 * t.r > nslot || nk < 8) return t;
    int best = 10 * nk;
    for (int sp = 2; sp * t.r <= nslot && nk / sp >= 4; ++sp) {
        const int cost = 10 * ((nk + sp - 1) / sp) + 10 * ov + 6 * sp * t.r;
        if (cost < best) { best = cost; t.sp = sp; }
    }
    return t;
}

PF_DEVICE TileCoord tile_coord(const Args& p, int t, int tiles_m, int tiles_n) {
    const int T1 = tiles_m * p.batch * tiles_n;
    if (t >= T1) {                      // second problem of a grouped launch: few row tiles, column-major walk (shared W panels)
        const int t2 = t - T1, TM2 = p.tiles2_m * p.batch;
        const int tn = t2 / TM2, tmm = t2 - tn * TM2;
        const int b = tmm / p.tiles2_m, tm = tmm - b * p.tiles2_m;
        return TileCoord{b, tm * BM, tn * BN, 1};
    }
    const int TM = tiles_m * p.batch;
    const int GM = p.group_m > 0 ? p.group_m : GROUP_M;
    const int group_sz = GM * tiles_n;
    const int grp = t / group_sz;
    const int first_m = grp * GM;
    const int gm = min(TM - first_m, GM);
    const int r_in = t - grp * group_sz;
    const int tn = r_in / gm;
    const int tmm = first_m + (r_in - tn * gm);
    const int b = tmm / tiles_m, tm = tmm - b * tiles_m;
    return TileCoord{b, tm * BM, tn * BN, 0};
}

template <bool CONV, int EPI>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The LDS ring holds R = 8 units (two K-tiles); load slot g issues unit g + LA.  (A ring of 10 units = all 160 KiB of the
    // CU with run-time ring positions was built and measured in round 5: no gain at the tile boundaries, and the position
    // arithmetic in the load slots cost the main loop 4-9 %: profiles/r05_gemm8p_vs_r4_library_ring10.log.)
    constexpr int R = 8, LA = R - 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;          // wave tile: rows wm*128 .. +128, columns wn*64 .. +64

    // ---- this workgroup's tile list: XCD x (= blockIdx % 8) owns a contiguous chunk of the logical tile order, its
    //      workgroups take the chunk's tiles round-robin, so the tiles in flight on one XCD are neighbours (shared
    //      operand panels in that XCD's L2).
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int T = (tiles_m + (CONV ? 0 : p.tiles2_m)) * p.batch * tiles_n;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nslot = (nwg - xcd + 7) >> 3;
    const int cq = T >> 3, cr = T & 7;
    const int cs = xcd * cq + min(xcd, cr);
    const int clen = cq + (xcd < cr ? 1 : 0);
    const int nk = p.K / BK;
    // segments of this workgroup: n_full whole tiles, then at most one tail segment = K-tiles [tail_k0, tail_k1) of tile
    // `tail_tile` (a whole tile unless the tail is split: tail_plan)
    const TailPlan tp = tail_plan(clen, nslot, nk, (!CONV && p.part != nullptr && (nwg & 7) == 0) ? p.tail_ov : -1);
    // (CONV: ov = -1 is a compile-time constant, the plan folds to { n_full, r, 1 } and the split code disappears)
    int tail_tile = -1, tail_k0 = 0, tail_k1 = 0;
    if (tp.sp > 1) {
        const int j = slot / tp.sp, q = slot - j * tp.sp;
        if (j < tp.r) { tail_tile = cs + tp.n_full * nslot + j; tail_k0 = q * nk / tp.sp; tail_k1 = (q + 1) * nk / tp.sp; }
    } else if (slot < tp.r) {
        tail_tile = cs + tp.n_full * nslot + slot; tail_k1 = nk;
    }
    const bool tail_parks = tp.sp > 1;              // the tail segment ends with raw sums in scratch, not with an epilogue
    // DESYNCHRONISED START (round 4, off by default: pf_gemm_set_policy(9)).  All workgroups walk equally long tiles in lockstep,
    // so their epilogues -- 128 KiB of C each -- leave the chip in one 32-MB burst per round (~12 000 cycles per tile, 14 % of a
    // K = 1920 tile: lab/gemm4w_lab.hip measured it on a second GEMM structure).  A workgroup whose XCD chunk gives it one tile
    // less than the busiest ones (no tail split: slot >= r) can wait out a fraction of a tile time for free; its epilogues then
    // fall into the others' main loops.  Timing only: tiles, order and arithmetic are unchanged.
    if (!CONV && p.stagger > 0 && tp.sp == 1 && tp.r > 0 && slot >= tp.r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)((slot - tp.r) & 7) * (unsigned)p.stagger * (unsigned)nk;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    const int n_full = tp.n_full;
    const int n_my = n_full + (tail_tile >= 0 ? 1 : 0);
    if (n_my == 0) return;
    auto tile_of = [&](int seq) { return seq < n_full ? cs + slot + seq * nslot : tail_tile; };
    auto seg_begin = [&](int seq) { return seq < n_full ? 0 : tail_k0; };
    auto seg_end = [&](int seq) { return seq < n_full ? nk : tail_k1; };
    const int GK = n_full * nk + (tail_k1 - tail_k0);      // K-tiles of all segments
    const int U = 4 * GK;                           // 16-KiB units of the operand stream (A0, B0, B1, A1 per K-tile)

    // ---- DMA geometry.  A unit = 16 pieces of 8 LDS rows; wave w owns pieces 2w, 2w+1 of every unit.
    //   A sub s (s = 0,1): LDS rows wm'*128 + s*64 + [0,64) for wm' = 0,1   (natural row order)
    //   B sub s:           LDS rows wn'*64 + s*32 + [0,32) for wn' = 0..3;  LDS row wn'*64 + 16 j + i holds W row
    //                      n0 + wn'*64 + 32 (j>>1) + 8 (i>>2) + 4 (j&1) + (i&3)      (the epilogue permutation)
    // lane -> row-in-piece lane>>3, LDS chunk lane&7 holds source chunk (lane&7) ^ ((row>>1)&7).
    // `ln` = lane, laundered through an empty asm where the per-tile set-up uses it: otherwise the compiler hoists a
    // dozen loop-invariant lane-geometry values out of the main loop and keeps them in registers the loop needs
    auto a_lds_row = [&](int ln, int s_, int e) { const int idx = 2 * wid + e; return (idx >> 3) * 128 + s_ * 64 + (idx & 7) * 8 + (ln >> 3); };
    auto b_lds_row = [&](int ln, int s_, int e) { const int idx = 2 * wid + e; return (idx >> 2) * 64 + s_ * 32 + (idx & 3) * 8 + (ln >> 3); };
    auto b_w_row = [&](int ln, int s_, int e) {          // W row (relative to n0) stored in LDS row b_lds_row(s_, e)
        const int idx = 2 * wid + e, within = idx & 3;
        const int i = (within & 1) * 8 + (ln >> 3);
        return (idx >> 2) * 64 + 32 * s_ + 8 * (i >> 2) + 4 * (within >> 1) + (i & 3);
    };
    // wave-uniform LDS destinations (piece bases) inside a buffer
    auto a_dst = [&](int s_, int e) { const int idx = 2 * wid + e; return ((idx >> 3) * 128 + s_ * 64 + (idx & 7) * 8) * 128; };
    auto b_dst = [&](int s_, int e) { const int idx = 2 * wid + e; return TILE_BYTES + ((idx >> 2) * 64 + s_ * 32 + (idx & 3) * 8) * 128; };

    // ---- issue-side state: (tile, K-tile) of the unit group being issued, per-lane byte offsets relative to the
    //      tile's scalar base pointers
    int is_tile = 0, is_kt = 0, is_kt_end = nk;
    const char* is_abase = nullptr;                 // A + batch/tile offset (bytes), wave-uniform
    const char* is_wbase = nullptr;
    unsigned a_off[2][2], w_off[2][2];
    // conv tap walk of the issue stream (CONV only)
    int is_c0 = 0, is_dw = 0, is_dh = 0, is_dt = 0;

    auto setup_issue_tile = [&](int seq) {
        const TileCoord tc = tile_coord(p, tile_of(seq), tiles_m, tiles_n);
        const Prob& q = p.pr[CONV ? 0 : tc.g];
        const bf16_t* A = q.A + (long long)tc.b * q.sA;
        long long base_el;                          // element offset of the tile's first row
        if (CONV) {
            const int hw = p.cg.H * p.cg.W;
            const int tt = tc.m0 / hw, rem = tc.m0 - tt * hw;
            const int hh = rem / p.cg.W, ww = rem - hh * p.cg.W;
            base_el = p.cg.base_off + (((long long)tt * p.cg.st * p.cg.Hp + hh * p.cg.sh) * p.cg.Wp + ww * p.cg.sw) * p.cg.Cin;
        } else {
            base_el = (long long)tc.m0 * p.lda;
        }
        is_abase = (const char*)(A + base_el);
        is_wbase = (const char*)(q.W + (long long)tc.n0 * p.ldw);
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ra = a_lds_row(ln, s, e), rb = b_lds_row(ln, s, e);
                int m = tc.m0 + ra;
                m = m < q.M ? m : q.M - 1;
                long long el;
                if (CONV) {
                    const int hw = p.cg.H * p.cg.W;
                    const int tt = m / hw, rem = m - tt * hw;
                    const int hh = rem / p.cg.W, ww = rem - hh * p.cg.W;
                    el = p.cg.base_off + (((long long)tt * p.cg.st * p.cg.Hp + hh * p.cg.sh) * p.cg.Wp + ww * p.cg.sw) * p.cg.Cin;
                } else {
                    el = (long long)m * p.lda;
                }
                a_off[s][e] = (unsigned)((el - base_el) * 2 + (((ln & 7) ^ ((ra >> 1) & 7)) << 4));
                int n = tc.n0 + b_w_row(ln, s, e);
                n = n < p.N ? n : p.N - 1;
                w_off[s][e] = (unsigned)((long long)(n - tc.n0) * p.ldw * 2 + (((ln & 7) ^ ((rb >> 1) & 7)) << 4));
            }
        is_kt = seg_begin(seq);                     // (a start inside K only happens without CONV: tail split)
        is_kt_end = seg_end(seq);
        is_c0 = is_dw = is_dh = is_dt = 0;
    };

    // byte offset of the issue stream's current K-tile inside a row of A
    auto a_koff = [&]() -> long long {
        if (CONV) return ((((long long)is_dt * p.cg.Hp + is_dh) * p.cg.Wp + is_dw) * p.cg.Cin + is_c0) * 2;
        return (long long)is_kt * (BK * 2);
    };

    // issue unit j of the stream into buffer (j >> 2) & 1 (kind = j & 3: 0 = A sub 0, 1 = B sub 0, 2 = B sub 1, 3 = A sub 1 --
    // passed by the caller, where it is a compile-time constant: the per-lane offset arrays must not be indexed at run time)
    auto issue_unit = [&](int j, const int kind) {
        char* buf = smem + ((j >> 2) & 1) * BUF_BYTES;
        if (kind == 0 || kind == 3) {
            const int s = kind == 0 ? 0 : 1;
            const char* src = is_abase + a_koff();
#pragma unroll
            for (int e = 0; e < 2; ++e) glds16(src + a_off[s][e], buf + a_dst(s, e));
        } else {
            const int s = kind - 1;
            const char* src = is_wbase + (long long)is_kt * (BK * 2);
#pragma unroll
            for (int e = 0; e < 2; ++e) glds16(src + w_off[s][e], buf + b_dst(s, e));
        }
        if (kind == 3) {                            // K-tile complete: advance the issue stream
            ++is_kt;
            if (CONV) {
                is_c0 += BK;
                if (is_c0 == p.cg.Cin) {
                    is_c0 = 0;
                    if (++is_dw == p.cg.kw) { is_dw = 0; if (++is_dh == p.cg.kh) { is_dh = 0; ++is_dt; } }
                }
            }
            if (is_kt == is_kt_end) {
                ++is_tile;
                if (is_tile < n_my) setup_issue_tile(is_tile);
            }
        }
    };

    // ---- fragment reads.  16x16x32 operand: lane -> row (lane&15) of the 16-row fragment, k chunk 4 kk + (lane>>4).
    const int frow = lane & 15, fq = lane >> 4, fswz = (lane >> 1) & 7;
    const int a_rd = (wm * 128 + frow) * 128;                       // + s*64*128 + f*16*128 + chunk
    const int b_rd = TILE_BYTES + (wn * 64 + frow) * 128;           // + s*32*128 + jj*16*128 + chunk
    int ch[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) ch[kk] = ((4 * kk + fq) ^ fswz) << 4;

    bf16x8_t fa[4][2];           // [m-fragment][kk] of the CURRENT A sub-tile (sub 0 in phases 0-1, sub 1 in 2-3)
    bf16x8_t fb[2][2][2];        // [sub][n-fragment][kk]   W rows (the MFMA "A" operand)
    f32x4v acc[8][4];            // [m-fragment 0..7][n-fragment 0..3]: C^T tile, lane: row lane&15, cols 4 fq + r
    // The accumulators of a tile START at the bias of their columns (zero without a bias): the epilogue has no bias
    // loads to wait for and no adds; the next tile's bias is requested at the top of the epilogue and lands under it.
    f32x4_t nb0, nb1, nb2, nb3;  // bias of the wave's columns in accumulator order (n-fragment 0..3); named scalars:
                                 // an array of vectors written under a condition ends up in scratch memory
    // (the residual flavour has no registers to spare for this: it starts from zero and adds the bias in its epilogue)
    constexpr bool FOLD_BIAS = (EPI & 1) == 0;
    auto load_bias = [&](int seq) {
        const TileCoord tc = tile_coord(p, tile_of(seq), tiles_m, tiles_n);
        const float* const bias_ = p.pr[CONV ? 0 : tc.g].bias;
        const f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        nb0 = nb1 = nb2 = nb3 = z;
        // (the parts of a split tail tile start from zero: gemm8p_tail_kernel adds the bias)
        if (FOLD_BIAS && bias_ && !(tail_parks && seq >= n_full)) {
            const int c0 = tc.n0 + wn * 64 + 8 * fq, c1 = c0 + 32;
            const float* b0 = bias_ + (c0 < p.n_valid ? c0 : 0);
            const float* b1 = bias_ + (c1 < p.n_valid ? c1 : 0);
            nb0 = *(const f32x4_t*)b0;
            nb1 = *(const f32x4_t*)(b0 + 4);
            nb2 = *(const f32x4_t*)b1;
            nb3 = *(const f32x4_t*)(b1 + 4);
        }
    };
    auto acc_from_bias = [&]() {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            acc[f][0] = (f32x4v){nb0[0], nb0[1], nb0[2], nb0[3]};
            acc[f][1] = (f32x4v){nb1[0], nb1[1], nb1[2], nb1[3]};
            acc[f][2] = (f32x4v){nb2[0], nb2[1], nb2[2], nb2[3]};
            acc[f][3] = (f32x4v){nb3[0], nb3[1], nb3[2], nb3[3]};
        }
    };
    load_bias(0);
    acc_from_bias();
#ifdef PF_G8_STAMP
    unsigned stampv = 0;
    unsigned long long sA = 0, sC = 0;
    const int st0 = p.stagger < 0 ? -p.stagger : 0;            // window start (K-tile index of this workgroup's stream)
#endif
#if PF_G8_STAMP == 3          // mode 3: five stamps per tile inside the epilogue (entry, conversions done, queue drained, stores
    int e_tile = 0;             // issued, accumulators re-initialised) for this wave's first 12 tiles
#define G8_EPI(k)                                                         \
    do {                                                                  \
        PF_FENCE();                                                       \
        G8_ISSUE(sA);                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                \
        G8_PUT(sA, e_tile < 12 ? e_tile * 5 + (k) : 64);                  \
        PF_FENCE();                                                       \
    } while (0)
#else
#define G8_EPI(k)
#endif

    auto read_a = [&](int s, int bufsel) {
        const char* base = smem + bufsel * BUF_BYTES + a_rd + s * (64 * 128);
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[f][kk] = *(const bf16x8_t*)(base + f * (16 * 128) + ch[kk]);
    };
    auto read_b = [&](int s, int bufsel) {
        const char* base = smem + bufsel * BUF_BYTES + b_rd + s * (32 * 128);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[s][jj][kk] = *(const bf16x8_t*)(base + jj * (16 * 128) + ch[kk]);
    };
    auto mfma_quadrant = [&](int sa, int sb) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[sa * 4 + f][sb * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        fb[sb][jj][kk], fa[f][kk], acc[sa * 4 + f][sb * 2 + jj], 0, 0, 0);
    };

    // ---- epilogue of the compute tile, straight from the accumulators.  The flavour (residual, fp32 output, GELU) is a
    //      template parameter: a branch-free epilogue is a few hundred instructions; the all-runtime form was ~60 KB of
    //      code that missed the instruction cache once per tile (measured: 13 % of a K = 1920 tile).
    //      Order: request bias / gate / the residual pieces, drain the vector-memory queue ONCE (this also retires
    //      every DMA issued so far, see the main loop), then fp32 math and one 16-byte store per (row fragment, half).
    // Flavour 8 (QK): the wave tiles that lie in the K or the Q column block (64 columns = one head: wave-uniform) apply
    // QK-RMSNorm + RoPE to the bf16-rounded result before storing it -- the arithmetic of pf_qk_norm_rope (common.h:
    // qk_sumsq8 / qk_rstd / qk_rope8), the same bits as the separate pass.  A row's 64 channels of a head sit in 4 lanes
    // (fq = 0..3) x 2 column halves x 8: sum of squares = per half in channel order, lanes xor 16, xor 32, then the two
    // halves -- the pairwise tree of the separate kernel's 8 sub-lanes.  The (cos, sin) rows and the gains are loaded in
    // batches of QFB row fragments per column half (all loads of a batch before its one drain, as for the residual).
    constexpr bool E_RES = (EPI & 1) != 0, E_F32 = (EPI & 2) != 0, E_ACT = (EPI & 4) != 0, E_QK = (EPI & 8) != 0;
    static_assert(!E_QK || (!E_RES && !E_F32 && !CONV), "the QK epilogue is a form of the plain / GELU flavour");
    auto epilogue_tile = [&](int seq) {
        const TileCoord tc = tile_coord(p, tile_of(seq), tiles_m, tiles_n);
        const Prob& q = p.pr[CONV ? 0 : tc.g];
        const int wave_m0 = tc.m0 + wm * 128, wave_n0 = tc.n0 + wn * 64;
        const bool mapped = CONV && p.om.mode == 1;
        // the memory-layout lane geometry (row ln >> 2, piece ln & 3, the two permutation sources) is derived from a LAUNDERED
        // lane id: loop-invariant otherwise, the compiler computes it once in the prologue and keeps it alive -- or spills it
        // -- across the main loop (setup_issue_tile does the same)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // 0 = not a QK wave tile, 1 = K block, 2 = Q block (scalar)
        int qk_reg = 0;
        if (E_QK) {
            if (p.qk_hs > 0) {               // head-major: [head][k | v | q ...]: the wave tile is one 64-wide block of one head
                const int hd_ = wave_n0 / p.qk_hs, c_ = wave_n0 - hd_ * p.qk_hs;
                if (hd_ * 64 < p.qk_d) qk_reg = c_ == p.qk_k0 ? 1 : (c_ == p.qk_q0 ? 2 : 0);
            } else if (p.qk_k0 >= 0 && wave_n0 >= p.qk_k0 && wave_n0 < p.qk_k0 + p.qk_d) qk_reg = 1;
            else if (p.qk_q0 >= 0 && wave_n0 >= p.qk_q0 && wave_n0 < p.qk_q0 + p.qk_d) qk_reg = 2;
        }
        // output element offset of (row m, column n); false = nothing to store for this row
        auto out_off = [&](int m, int n, long long& coff) -> bool {
            if (mapped) {
                const int hw = p.om.H * p.om.W;
                const int tt = m / hw, rem = m - tt * hw;
                const int hh = rem / p.om.W, ww = rem - hh * p.om.W;
                const int gg = n / p.om.Cg, cc = n - gg * p.om.Cg;
                const int shw = p.om.sh * p.om.sw;
                const int pt = gg / shw, g2 = gg - pt * shw;
                const int ph = g2 / p.om.sw, pw = g2 - ph * p.om.sw;
                const int tf = tt * p.om.st + pt + p.om.t_shift;
                coff = p.om.base_off +
                       (((long long)(tf < 0 ? 0 : tf) * p.om.Hop + (hh * p.om.sh + ph)) * p.om.Wop + (ww * p.om.sw + pw)) *
                           p.om.Cout_pitch + cc;
                return tf >= 0;
            }
            coff = (long long)tc.b * q.sC + (long long)m * p.ldc + n;
            return true;
        };
        const bf16_t* res_row = E_RES ? q.res + (long long)tc.b * q.sR + (long long)(wave_m0 + frow) * p.ldr : nullptr;
        char* const c_row = (char*)q.C + ((long long)tc.b * q.sC + (long long)(wave_m0 + frow) * p.ldc) * (E_F32 ? 4 : 2);
        int ncol[2];
        bool ncol_ok[2];
        f32x4_t gate4[2][2];
        f32x4_t rb0, rb1;                                       // residual flavour: bias of the current column half
        const bool more_tiles = seq + 1 < n_my;
        // the next tile's bias: consumed after the stores are issued (acc_from_bias)
        // `early`: this wave's conversions wait for loads of their own (residual pieces, rope rows) or its stores are interleaved
        // with them (fp32 output): the vector-memory queue is drained before the first conversion.  Otherwise (plain / GELU, and
        // the V / MLP column blocks of the QK flavour) the drain comes after the conversions, directly before the first store:
        // the units requested above land under the register work.
        const bool early = E_RES || E_F32;          // (the QK wave tiles of flavour 8 have left through their own path above)
        G8_EPI(0);
        if (FOLD_BIAS && more_tiles) load_bias(seq + 1);
        auto load_col_params = [&](int hsel) {
            const int n_raw = wave_n0 + 32 * hsel + 8 * fq;
            ncol_ok[hsel] = n_raw < p.n_valid;
            const int n = ncol_ok[hsel] ? n_raw : 0;           // clamped: loads stay in bounds, stores are masked
            ncol[hsel] = n;
            if (E_RES) {
                rb0 = rb1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (q.bias) {
                    rb0 = *(const f32x4_t*)(q.bias + n);
                    rb1 = *(const f32x4_t*)(q.bias + n + 4);
                }
                if (q.gate) {
                    const float* gp = q.gate + (long long)tc.b * p.gate_stride + n;
                    gate4[hsel][0] = *(const f32x4_t*)gp;
                    gate4[hsel][1] = *(const f32x4_t*)(gp + 4);
                } else {
                    gate4[hsel][0] = gate4[hsel][1] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
                }
            }
        };
        if (!E_RES) {          // one wait for both halves; the residual flavour waits per half and loads them there
            load_col_params(0);
            load_col_params(1);
        }
        // vmcnt counts loads and stores in issue order: a load queued behind the tile's stores would wait for them, so
        // every load of the epilogue is issued before the first store and the bf16 results are packed in registers
        // (they take the place of the accumulators they were computed from) until both column halves are done.
        u32x4_t outp[2][8];
        // row fragments per batch: the residual flavour loads a column half's 8 residual pieces, waits once, converts them
        // (two drains per tile; 4-fragment batches = four drains measured 0.5-2.5 % slower)
        constexpr int FB = 8;
#pragma unroll
        for (int hsel = 0; hsel < 2; ++hsel) {
            if (E_RES) load_col_params(hsel);
            const int n = ncol[hsel];
            // wave-uniform (gelu_from is a multiple of 32, pf_gemm8p_supports): the column halves left of gelu_from skip
            // the activation's VALU work instead of computing and discarding it
            const bool do_act = E_ACT && (wave_n0 + 32 * hsel) >= p.gelu_from;
#pragma unroll
            for (int f0 = 0; f0 < 8; f0 += FB) {
                u32x4_t rbuf[FB];
                if (E_RES) {
#pragma unroll
                    for (int f = 0; f < FB; ++f) {
                        const int m = wave_m0 + 16 * (f0 + f) + frow;
                        if (mapped) {
                            long long roff;
                            out_off(m < q.M ? m : q.M - 1, n, roff);
                            rbuf[f] = *(const u32x4_t*)(q.res + roff);
                        } else {
                            // one per-lane row pointer + a wave-uniform fragment stride: no per-fragment 64-bit
                            // offsets kept alive across the epilogue; rows past M are not read at all.  (Round 6 tried these
                            // loads in the coalesced memory layout twice: with an inverse lane permutation of the pieces the
                            // flavour has no register left -- 12-92 bytes of scratch in every form --; with the fp32 VALUES
                            // permuted to meet them it fits (253 registers, same bits) and is no faster: what the flavour pays
                            // is the two exposed load -> wait round trips, not the shape of the loads.
                            // profiles/r06_gemm8p_small_variants_ab.log)
                            rbuf[f] = (u32x4_t){0u, 0u, 0u, 0u};
                            if (m < q.M) rbuf[f] = *(const u32x4_t*)(res_row + (long long)(16 * (f0 + f)) * p.ldr + n);
                        }
                    }
                }
                if ((hsel == 0 && f0 == 0 && early) || E_RES) {
                    // drains the DMA queue (see the main loop) and, for the residual / QK flavours, this batch's pieces
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    PF_FENCE();
                    // EVERY register a global load of this epilogue has filled is redefined here, right behind the drain that
                    // covers it: the compiler does not see the asm drains and would otherwise guard the first use of each with
                    // a wait of its own -- and where that use has sunk into a store's predicated block (the residual math of
                    // the second column half does), with an s_waitcnt vmcnt(0) BEHIND the previous store: rounds 2-4 ran the
                    // residual flavour's 16 stores as 16 store round trips.  After the redefinition nothing is pending in the
                    // compiler's model and the stores go out back to back.
                    if (E_RES) {
#pragma unroll
                        for (int f = 0; f < FB; ++f) asm volatile("" : "+v"(rbuf[f]));
                        asm volatile("" : "+v"(rb0), "+v"(rb1), "+v"(gate4[hsel][0]), "+v"(gate4[hsel][1]));
                    }
                    PF_FENCE();
                    if (hsel == 0 && f0 == 0 && FOLD_BIAS && more_tiles && E_F32)
                        asm volatile("" : "+v"(nb0), "+v"(nb1), "+v"(nb2), "+v"(nb3));   // (this flavour's stores interleave below)
                }
#pragma unroll
                for (int fi = 0; fi < FB; ++fi) {
                    const int f = f0 + fi;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[f][2 * hsel][r];
                        v[4 + r] = acc[f][2 * hsel + 1][r];
                        if (!FOLD_BIAS) {
                            v[r] += rb0[r];
                            v[4 + r] += rb1[r];
                        }
                    }
                    if (E_ACT) {
                        if (do_act) gelu_tanh8(v);
                    }
                    if (E_RES) {
                        float rv[8];
                        unpack8(rbuf[fi], rv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = rv[e] + gate4[hsel][0][e] * v[e];
                            v[4 + e] = rv[4 + e] + gate4[hsel][1][e] * v[4 + e];
                        }
                    }
                    if (CONV) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
                    }
                    if (E_F32) {
                        const int m = wave_m0 + 16 * f + frow;
                        if (m < q.M && ncol_ok[hsel]) {
                            float* c = (float*)(c_row + ((long long)(16 * f) * p.ldc + n) * 4);
                            *(f32x4_t*)c = (f32x4_t){v[0], v[1], v[2], v[3]};
                            *(f32x4_t*)(c + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
                        }
                    } else {
                        outp[hsel][f] = pack8(v);
                    }
                    PF_FENCE();                    // keep the scheduler from overlapping fragments (register pressure)
                }
            }
        }
        if (!E_F32) {
            G8_EPI(1);
            // ---- flavour 8 (QK-RMSNorm + RoPE on the K / Q column blocks), round 6: everything behind the conversions happens in
            // the MEMORY layout -- lane L = (row L >> 2, 16-byte piece L & 3), see COALESCED STORES below.  The packed bf16
            // results (what the separate pass would read back) are lane-permuted in place; then, on the QK wave tiles, (i) the
            // (cos, sin) rows are read as 128 contiguous bytes per quad of lanes (in the accumulator layout the 32 loads of a
            // wave tile were 64 separate 16-byte reads each: most of a 23 k-cycle tile boundary), (ii) the four lanes that share
            // a row's head are ADJACENT: the sum of squares' cross-lane adds are two DPP quad permutes instead of two trips
            // through the LDS crossbar.  Same pairwise tree (piece ^ 1, piece ^ 2, then the two column halves) and the same
            // per-lane arithmetic (common.h) as the separate pass: same bits.
            constexpr int QB = 2;                                 // (4 items per batch: 56 bytes of scratch)
            f32x4_t cs[2][QB][2], gw[2][2];                       // (cos, sin) rows: batches of QB items, double-buffered; gains
            float rs[8];                                          // 1 / rms of row (fragment f, row ln >> 2)
            const int frow2 = ln >> 2, pc = ln & 3;
            auto load_cs = [&](int bt) {                          // item i = 2 f + hsel
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int i = QB * bt + k;
                    const int m2 = wave_m0 + 16 * (i >> 1) + frow2;
                    const float* cs_ = p.qk_rope + ((long long)(q.qk_row0 + (m2 < q.M ? m2 : q.M - 1)) * 64 + 32 * (i & 1) + 8 * pc);
                    cs[bt & 1][k][0] = *(const f32x4_t*)cs_;
                    cs[bt & 1][k][1] = *(const f32x4_t*)(cs_ + 4);
                }
            };
            if (E_QK) {
                if (qk_reg) {
                    load_cs(0);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float* w_ = (qk_reg == 1 ? q.qk_wk : q.qk_wq) + 32 * h + 8 * pc;
                        gw[h][0] = *(const f32x4_t*)w_;
                        gw[h][1] = *(const f32x4_t*)(w_ + 4);
                    }
                }
                const int psrc = (((ln & 3) << 4) | (ln >> 2)) << 2;
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int d_ = 0; d_ < 4; ++d_)
                        outp[i & 1][i >> 1][d_] = (unsigned)__builtin_amdgcn_ds_bpermute(psrc, (int)outp[i & 1][i >> 1][d_]);
                if (qk_reg) {
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        float ssh[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float vb[8];
                            unpack8(outp[h][f], vb);
                            float ss = qk_sumsq8(vb);
                            ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0xB1, 0xf, 0xf, false));   // piece ^ 1
                            ss += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ss), 0x4E, 0xf, 0xf, false));   // piece ^ 2
                            ssh[h] = ss;
                        }
                        rs[f] = qk_rstd(ssh[0] + ssh[1], p.qk_eps);
                    }
                }
            }
            if (!early) {       // nothing of this wave's DMA is in flight across its stores (see the main loop's skip_wait)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PF_FENCE();
            }
            G8_EPI(2);
            if (E_QK) {
                if (qk_reg) {
                    // (every register a load of this path filled is redefined behind the drain: the file's rule.  The packed items
                    //  too: otherwise the compiler keeps the 128 UNPACKED floats of the sum-of-squares pass alive for the rotation)
#pragma unroll
                    for (int k = 0; k < QB; ++k) asm volatile("" : "+v"(cs[0][k][0]), "+v"(cs[0][k][1]));
                    asm volatile("" : "+v"(gw[0][0]), "+v"(gw[0][1]), "+v"(gw[1][0]), "+v"(gw[1][1]));
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(outp[i & 1][i >> 1]));
                    PF_FENCE();
                    const float osc = qk_reg == 2 ? p.qk_qs : 1.f;
#pragma unroll
                    for (int bt = 0; bt < 16 / QB; ++bt) {
                        if (bt + 1 < 16 / QB) load_cs(bt + 1);         // (ordinary loads with no store behind them: the compiler's own counted waits)
                        PF_FENCE();
#pragma unroll
                        for (int k = 0; k < QB; ++k) {
                            const int i = QB * bt + k;
                            float vb[8], w8[8], cs8[8], v[8];
                            unpack8(outp[i & 1][i >> 1], vb);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                w8[e] = gw[i & 1][0][e]; w8[4 + e] = gw[i & 1][1][e];
                                cs8[e] = cs[bt & 1][k][0][e]; cs8[4 + e] = cs[bt & 1][k][1][e];
                            }
                            qk_rope8(vb, rs[i >> 1], w8, cs8, osc, v);
                            outp[i & 1][i >> 1] = pack8(v);
                        }
                        PF_FENCE();
                    }
                }
            }
            // THE NEXT TILE'S BIAS HAS LANDED (every path above drained the queue after requesting it) -- but the compiler does
            // not see the asm drains: it would guard the first use of nb0..3 (acc_from_bias, BEHIND the stores) with its own
            // s_waitcnt vmcnt(0), i.e. every epilogue would wait for its whole store burst to retire (rounds 2-4 did: the
            // "7 % for the stores" of the epilogue diagnosis).  Redefining the four registers here, in front of the stores,
            // moves that wait to where the queue is empty anyway.
            if (FOLD_BIAS) {
                asm volatile("" : "+v"(nb0), "+v"(nb1), "+v"(nb2), "+v"(nb3));
                PF_FENCE();
            }
            // COALESCED STORES.  In the accumulator layout a row's 64-byte piece sits in lanes {r, r+16, r+32, r+48}: ADJACENT lanes
            // hold different rows (addresses ldc apart), so the memory pipeline sees 64 separate 16-byte writes per store
            // instruction -- measured: ~270-530 cycles per store instruction and wave, 8-10 k cycles of a tile boundary
            // (profiles/r06_gemm8p_stamps.log).  One lane permutation per packed register (ds_bpermute_b32: the LDS crossbar, no
            // LDS memory, no barrier) turns lane L into (row L >> 2, piece L & 3): every quad of lanes then writes 64 contiguous
            // bytes, the two halves of a 128-byte line back to back.  Software-pipelined by one item: the permutation of item
            // i + 1 is in flight on the crossbar while item i's store issues.  Same values, same addresses, other lanes.
            if (!mapped) {
                const int psrc = (((ln & 3) << 4) | (ln >> 2)) << 2;
                char* const crow2 = (char*)q.C + ((long long)tc.b * q.sC + (long long)(wave_m0 + (ln >> 2)) * p.ldc) * 2;
                auto perm = [&](const u32x4_t v) {
                    u32x4_t o;
#pragma unroll
                    for (int d_ = 0; d_ < 4; ++d_) o[d_] = (unsigned)__builtin_amdgcn_ds_bpermute(psrc, (int)v[d_]);
                    return o;
                };
                u32x4_t cur = E_QK ? outp[0][0] : perm(outp[0][0]);          // (flavour 8: permuted above, all of them)
#pragma unroll
                for (int i = 0; i < 16; ++i) {              // i = 2 f + hsel
                    u32x4_t nxt = cur;
                    if (i + 1 < 16) nxt = E_QK ? outp[(i + 1) & 1][(i + 1) >> 1] : perm(outp[(i + 1) & 1][(i + 1) >> 1]);
                    const int f = i >> 1, hsel = i & 1;
                    const int m2 = wave_m0 + 16 * f + (ln >> 2), n2 = wave_n0 + 32 * hsel + 8 * (ln & 3);
                    if (m2 < q.M && n2 < p.n_valid) *(u32x4_t*)(crow2 + ((long long)(16 * f) * p.ldc + n2) * 2) = cur;
                    cur = nxt;
                }
            } else {
#pragma unroll
                for (int hsel = 0; hsel < 2; ++hsel)
#pragma unroll
                    for (int f = 0; f < 8; ++f) {
                        const int m = wave_m0 + 16 * f + frow;
                        if (mapped) {
                            long long coff;
                            const bool ok = out_off(m < q.M ? m : q.M - 1, ncol[hsel], coff) && m < q.M && ncol_ok[hsel];
                            if (ok) *(u32x4_t*)((bf16_t*)q.C + coff) = outp[hsel][f];
                        } else if (m < q.M && ncol_ok[hsel]) {
                            *(u32x4_t*)(c_row + ((long long)(16 * f) * p.ldc + ncol[hsel]) * 2) = outp[hsel][f];
                        }
                        PF_FENCE();
                    }
            }
        }
    };
    // part of a split tail tile (always this workgroup's last segment): park the raw sums, lane-linear 16-byte pieces,
    // piece = accumulator index (row fragment f, column group c): slot `bid`, 32 KiB per wave
    auto park = [&]() {
        int lnp = lane;
        asm volatile("" : "+v"(lnp));
        char* const pw = (char*)p.part + ((long long)bid * 8 + wid) * 32768 + (unsigned)lnp * 16u;
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int c = 0; c < 4; ++c) *(f32x4v*)(pw + (f * 4 + c) * 1024) = acc[f][c];
    };
    auto epilogue = [&](int seq) {
        if (!CONV && tail_parks && seq >= n_full) { park(); return; }
        epilogue_tile(seq);
        G8_EPI(3);
        // the accumulators restart (from the next tile's bias; zero after / before a parked segment) only now:
        // re-initialising them while the packed results are still waiting for their stores would keep 128 + 64 registers
        // alive at once
        PF_FENCE();
        acc_from_bias();
        G8_EPI(4);
#if PF_G8_STAMP == 3
        ++e_tile;
#endif
    };

    // ---- prologue: units 0 .. LA - 1; units 0 .. 2 (A sub 0 and both B subs of the first K-tile: what slot 0 reads) must have
    //      landed before the first barrier, units 3 .. 5 (6 pieces of this wave) may be in flight
    setup_issue_tile(0);
    {
#pragma unroll
        for (int j = 0; j < LA; ++j)
            if (j < U) issue_unit(j, j & 3);
        if (U >= LA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PF_BAR();
    if (wm == 1) PF_BAR();                 // group 1 runs one barrier behind group 0
    // static priority for the second-dispatched wave group, no per-phase flips (MI355X_MICROARCH.md, "two waves per SIMD", item 4;
    // measured +1-2 % on every DiT shape: profiles/r06_gemm8p_variants_ab.log)
    if (wm == 1) __builtin_amdgcn_s_setprio(1);

    // ---- main loop over the K-tiles of all tiles of this workgroup (two phases per K-tile: file header)
    // skip_wait: load slots after an epilogue whose wait is already covered (the epilogue drained every unit issued before
    // its stores): one -- slot 2gk + 2 needs units <= 4gk + 7, all issued before the stores; the one after it does not.
    // (Requesting more units in front of the stores, so that more slots run under the store burst, was measured in round 5
    // on the four-phase form: neutral.)
    int skip_wait = 0;
    int c_tile = 0, c_kt = seg_begin(0), c_end = seg_end(0);
    auto end_slot2 = [&](int S, const int h) {
        if (2 * S + LA < U) {
            issue_unit(2 * S + LA, h == 0 ? 2 : 0);
            issue_unit(2 * S + LA + 1, h == 0 ? 3 : 1);
            if (skip_wait > 0) --skip_wait;
            else if (h == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PF_BAR();
#if PF_G8_STAMP == 1
        if (h == 0) {
            G8_ISSUE(sC);
            PF_FENCE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PF_FENCE();
            G8_PUT(sC, (S >> 1) - st0);
            PF_FENCE();
        }
#endif
    };
    for (int gk = 0; gk < GK; ++gk) {
        const int bs = gk & 1;
        read_b(0, bs);
        read_b(1, bs);
        PF_FENCE();
        read_a(0, bs);
        end_slot2(2 * gk, 0);
        mfma_quadrant(0, 0);
        mfma_quadrant(0, 1);
        PF_BAR();
        read_a(1, bs);
        end_slot2(2 * gk + 1, 1);
        mfma_quadrant(1, 1);
        mfma_quadrant(1, 0);
        // TILE BOUNDARY.  Group 1 runs one barrier behind group 0: with both epilogues in front of the loop's last barrier,
        // group 0's epilogue runs beside group 1's (short) load slot and group 1's beside group 0's next load slot -- one
        // after the other, the matrix pipe idle through both.  epi_mode bit 0: group 0 takes the barrier FIRST, so that its
        // epilogue falls into the same barrier interval as group 1's MFMA slot + epilogue (register work, drains and store
        // issue of the two groups overlap; nothing else moves: the epilogue touches no LDS and issues no DMA).
        const bool tile_done = ++c_kt == c_end;
        const bool late = tile_done && wm == 0 && (p.epi_mode & 1);
        if (late) PF_BAR();
        if (tile_done) {
            // every DMA issued so far has had >= one MFMA phase; draining here makes the wait of the next load slot
            // unnecessary (its units were all issued before this point) and keeps the store traffic of the epilogue out
            // of the counted waits
            PF_FENCE();
            epilogue(c_tile);
            PF_FENCE();
            skip_wait = 1;
            ++c_tile;
            c_kt = seg_begin(c_tile);
            c_end = seg_end(c_tile);
        }
        if (!late) PF_BAR();
    }
    if (wm == 0) PF_BAR();                 // matches group 1's extra barrier
#ifdef PF_G8_STAMP
    g8_stamps[(bid * 8 + wid) * 64 + lane] = stampv;
#endif
}

// Second launch of a GEMM whose tail tiles were split along K: C = epi(sum over the parts, in part order, of the parked
// fp32 sums) for those tiles only.  32 blocks per tail tile; a thread owns what a lane of the main kernel owns for one
// (row fragment, column half): 8 consecutive columns of one row.  p.ksplit = the main launch's workgroup count.
__global__ __launch_bounds__(256) void gemm8p_tail_kernel(const Args p) {
    const int chunk = blockIdx.x & 31, tj = blockIdx.x >> 5;
    const int xcd = tj & 7, j = tj >> 3;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int T = (tiles_m + p.tiles2_m) * p.batch * tiles_n;
    const int nwg = p.ksplit;
    const int nslot = (nwg - xcd + 7) >> 3;
    const int cq = T >> 3, cr = T & 7;
    const int cs = xcd * cq + min(xcd, cr);
    const int clen = cq + (xcd < cr ? 1 : 0);
    const int nk = p.K / BK;
    const TailPlan tp = tail_plan(clen, nslot, nk, p.tail_ov);
    if (tp.sp <= 1 || j >= tp.r) return;
    const TileCoord tc = tile_coord(p, cs + tp.n_full * nslot + j, tiles_m, tiles_n);
    const Prob& pq = p.pr[tc.g];
    const int lane = threadIdx.x & 63;
    const int item = chunk * 4 + (threadIdx.x >> 6);            // (wave 0..7, row fragment 0..7, column half 0..1)
    const int w = item >> 4, f = (item >> 1) & 7, hsel = item & 1;
    const int m = tc.m0 + (w >> 2) * 128 + 16 * f + (lane & 15);
    const int n = tc.n0 + (w & 3) * 64 + 32 * hsel + 8 * (lane >> 4);
    if (m >= pq.M || n >= p.n_valid) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    // part q of tile j was parked by workgroup (slot j sp + q, this XCD); four parts' loads are in flight at a time and
    // the sums are added in part order
    const char* const src0 = (const char*)p.part + ((long long)((j * tp.sp) * 8 + xcd) * 8 + w) * 32768 +
                             (f * 4 + 2 * hsel) * 1024 + lane * 16;
    constexpr long long PART_STRIDE = 8ll * 8 * 32768;          // next workgroup slot of the same XCD
    int q = 0;
    for (; q + 4 <= tp.sp; q += 4) {
        f32x4_t a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0[u] = *(const f32x4_t*)(src0 + (q + u) * PART_STRIDE);
            a1[u] = *(const f32x4_t*)(src0 + (q + u) * PART_STRIDE + 1024);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a0[u][e]; v[4 + e] += a1[u][e]; }
    }
    for (; q < tp.sp; ++q) {
        const f32x4_t v0 = *(const f32x4_t*)(src0 + q * PART_STRIDE), v1 = *(const f32x4_t*)(src0 + q * PART_STRIDE + 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += v0[e]; v[4 + e] += v1[e]; }
    }
    if (pq.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += pq.bias[n + e];
    }
    if (n >= p.gelu_from) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
    }
    const long long coff = (long long)tc.b * pq.sC + (long long)m * p.ldc + n;
    if (p.flags & PF_GEMM_GATE_RES) {
        float rv[8];
        unpack8(*(const u32x4_t*)(pq.res + (long long)tc.b * pq.sR + (long long)m * p.ldr + n), rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rv[e] + (pq.gate ? pq.gate[(long long)tc.b * p.gate_stride + n + e] : 1.f) * v[e];
    }
    if (p.flags & PF_GEMM_OUT_F32) {
        float* c = (float*)pq.C + coff;
        *(f32x4_t*)c = (f32x4_t){v[0], v[1], v[2], v[3]};
        *(f32x4_t*)(c + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
    } else {
        *(u32x4_t*)((bf16_t*)pq.C + coff) = pack8(v);
    }
}

int g_num_cu = 0;
// CUs the persistent launches leave free (pf_gemm_set_policy(2000 + R), R a multiple of 8 = R / 8 per XCD; default 0).  A workgroup
// of this kernel takes a CU whole (128 KiB of LDS, the whole register file): no other kernel's workgroup fits beside it, and
// the tile assignment is static, so a communication kernel (RCCL send / recv of a sequence-parallel exchange, communicate.py:
// 7-26) that is in flight when the launch starts either delays the workgroups of the CUs it holds -- the launch then ends
// with THEIR tiles -- or, queued behind it, does not start before the launch ends.  With R CUs reserved both run side by
// side at (256 - R) / 256 of the GEMM rate (tools/comm_overlap_bench.py measures the three cases).
int g_reserve_cu = 0;
static int cu_count() {
    if (!g_num_cu) {
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&g_num_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu - g_reserve_cu;          // pf_gemm8p_set_reserved_cus refuses reservations that leave < 64 CUs
}
bool g_tail_split = true;                  // pf_gemm_set_policy(-4) / (4): never / again split the tail tiles along K
int g_tail_ov = 4;                         // fixed cost of a split in K-tile periods (tail_plan; pf_gemm_set_policy(400 + ov))
int g_stagger = 0;                         // pf_gemm_set_policy(9) / (-9): desynchronised start on (290 cycles per K-tile and 1/8 step) / off
int g_epi_mode = 1;                        // Args::epi_mode (pf_gemm_set_policy(1000 + m): measurement hook)

template <bool CONV, int EPI>
int launch(const Args& a_in, hipStream_t stream, void* ws, long long ws_bytes) {
    PF_SET_MAX_LDS_ONCE((gemm8p_kernel<CONV, EPI>), SMEM);
    const int ncu = cu_count();
    Args a = a_in;
    if (CONV) a.tiles2_m = 0;
    a.pr[0] = Prob{a.A, a.W, a.C, a.bias, a.res, a.gate, a.sA, a.sC, a.sR, a.qk_wq, a.qk_wk, a.M, a.qk_row0};
    const int tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM + a.tiles2_m) * a.batch;
    // fewer tiles than CUs: with scratch the whole chip is launched anyway and the spare workgroups take K ranges of the
    // tiles (tail_plan with no full round: every tile is a tail tile), when that plan splits at all
    const bool can_split = !CONV && g_tail_split && ws && ws_bytes >= (long long)ncu * (256 << 10) && (ncu & 7) == 0;
    const bool mid = can_split && tiles < ncu && pf_gemm8p_mid_split(tiles, a.K / BK) > 1;
    const int grid = (tiles < ncu && !mid) ? tiles : ncu;
    // tail split: one 256-KiB slot of caller scratch per workgroup; the second launch covers the XCD with the most tail tiles
    a.part = nullptr;
    a.ksplit = grid;
    a.tail_ov = g_tail_ov;
    a.stagger = g_stagger;
    a.epi_mode = g_epi_mode;
    int rmax = 0;
    if (!CONV && g_tail_split && ws && (grid & 7) == 0 && ws_bytes >= (long long)grid * (256 << 10)) {
        const int nk = a.K / BK, nslot = grid >> 3;
        for (int xcd = 0; xcd < 8; ++xcd) {
            const int clen = (tiles >> 3) + (xcd < (tiles & 7) ? 1 : 0);
            const TailPlan tp = tail_plan(clen, nslot, nk, g_tail_ov);
            if (tp.sp > 1 && tp.r > rmax) rmax = tp.r;
        }
        if (rmax > 0) a.part = (float*)ws;
    }
    hipLaunchKernelGGL((gemm8p_kernel<CONV, EPI>), dim3(grid), dim3(512), SMEM, stream, a);
    if (rmax > 0) hipLaunchKernelGGL(gemm8p_tail_kernel, dim3(rmax * 8 * 32), dim3(256), 0, stream, a);
    return 0;
}

}  // namespace


// Epilogue flavours that exist as instantiations: bias (+ out_scale for conv) always; ONE of {residual (+gate),
// fp32 output, GELU-tanh from a column}.  Anything else (CLIP activations, combinations) is served by the older kernels.
bool pf_gemm8p_supports(const Args& a, bool conv) {
    if (a.flags & (PF_GEMM_ACT_QUICK_GELU | PF_GEMM_ACT_GELU_ERF)) return false;
    const bool res = (a.flags & PF_GEMM_GATE_RES) != 0, f32 = (a.flags & PF_GEMM_OUT_F32) != 0, act = a.gelu_from < a.N;
    if ((int)res + (int)f32 + (int)act > 1) return false;
    if (conv && (f32 || act || res)) return false;      // conv + shortcut add: the instantiation spills (kept on gemm256)
    if (act && (a.gelu_from & 31)) return false;        // the activation is decided per 32-column half of a wave tile
    return true;
}

// parts per tile when a launch of `tiles` (< one round) tiles is spread over the whole chip by splitting K (the plan of the
// XCD with the most tiles); 1 = the split does not pay / is not possible
int pf_gemm8p_mid_split(int tiles, int nk) {
    if (!g_tail_split) return 1;
    const int ncu = cu_count();
    if (tiles >= ncu || (ncu & 7)) return 1;
    const int clen = (tiles + 7) >> 3;
    return tail_plan(clen, ncu >> 3, nk, g_tail_ov).sp;
}

#ifdef PF_G8_STAMP
extern "C" int pf_lab_gemm8p_stamps(unsigned* dst_host) {          // [256 workgroups][8 waves][64 stamps]
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g8_stamps), sizeof(unsigned) * 256 * 8 * 64);
}
#endif
void pf_gemm8p_set_tail_split(bool on) { g_tail_split = on; }
void pf_gemm8p_set_tail_overhead(int k_tiles) { g_tail_ov = k_tiles; }
void pf_gemm8p_set_stagger(int cycles) { g_stagger = cycles; }
void pf_gemm8p_set_epi_mode(int mode) { g_epi_mode = mode; }
int pf_gemm8p_set_reserved_cus(int n) {
    const int r = n > 0 ? (n + 7) / 8 * 8 : 0;
    g_reserve_cu = 0;
    if (cu_count() - r < 64) return 1;          // cannot be honoured: an error, not a silently ignored request
    g_reserve_cu = r;
    return 0;
}
int pf_gemm8p_workgroups() { return cu_count(); }

// Scratch (bytes) with which pf_gemm8p_launch may split the tail tiles of a problem along K (one slot per workgroup).
long long pf_gemm8p_workspace_bytes() { return 256ll * (256 << 10); }

int pf_gemm8p_launch(const Args& a_in, bool conv, hipStream_t stream, void* ws, long long ws_bytes) {
    const Args& a = a_in;
    const bool res = (a.flags & PF_GEMM_GATE_RES) != 0, f32 = (a.flags & PF_GEMM_OUT_F32) != 0, act = a.gelu_from < a.N;
    if (conv) return launch<true, 0>(a, stream, nullptr, 0);   // conv + shortcut add stays on gemm256 (pf_gemm8p_supports)
    if (a.qk_d > 0) return launch<false, 12>(a, stream, nullptr, 0);      // validated by pf_gemm_bf16: no residual / fp32 output
    if (res) return launch<false, 1>(a, stream, ws, ws_bytes);
    if (f32) return launch<false, 2>(a, stream, ws, ws_bytes);
    if (act) return launch<false, 4>(a, stream, ws, ws_bytes);
    return launch<false, 0>(a, stream, ws, ws_bytes);
}
