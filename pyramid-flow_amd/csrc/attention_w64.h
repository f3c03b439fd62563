// attention_w64.h -- the 64-rows-per-wave form of the masked flash attention (included by attention.hip).
//
// Why (profiles/r03_attention_ablations_and_phase_stamps.log, the shipped 32-rows-per-wave kernel at L = 15 488): of the
// ~3 060 cycles a wave spends per 64-key tile only 512 are matrix-pipe time; 14 % go to ISSUING its four LDS-DMA pieces,
// 13 % to the barrier, 9 % to the exposed latency of the K fragment reads, and the kernel without its barrier / DMA, or
// without its softmax, is only 16-19 % faster: the per-tile fixed costs are paid once per 32 query rows.  Here a wave owns
// TWO 32-row query blocks (A, B): the same tile of K / V^T fragments, the same four DMA pieces and the same barrier serve
// twice the MFMA work, and the two blocks give the wave independent work to interleave:
//
//     QK^T(A) | QK^T(B) + softmax(A) | PV(A) + softmax(B) | barrier, DMA of tile j+2, K(j+1) fragment reads, PV(B)
//
// K(j+1)'s fragments are requested before PV(B) so their latency hides under its eight MFMAs, V(j)'s under the row maxima of
// block A.  256 query rows per 4-wave workgroup, 2 workgroups per CU (<= 256 registers per lane), LDS as before (two
// 16-KiB K | V^T tile buffers).  Same arguments, mask form, deferred rescale and numerics as attn_kernel<true, ...>
// (q pre-scaled, scores are base-2 exponents); the running reference -m of a block enters through the C operand of the
// first S^T MFMA, rebuilt from one register per tile instead of being held in sixteen.  The Q fragments (32 registers for the
// two blocks) live in LDS instead of registers (each wave stages its own 64 rows once, 8 KiB) and are re-read per tile
// next to the K fragments: 8 more ds_read_b128 per tile on an LDS pipe that is < 20 % busy, for a kernel that fits 256
// registers without spilling (the register-resident form spilled 33-38 registers into the tile loop).
#pragma once

// MODE bit 0 FAST: no running row maximum at all -- P = exp2(s) in the unshifted domain (m = 0), valid while every row's
//                  largest score lies within about +-100 of zero (fp32 / bf16 share the exponent range; with QK-RMSNorm in
//                  front the scores are bounded by 11.5 x the norm gains).  Every wave reports whether one of its rows ended
//                  with a non-finite or vanishing denominator (p.wgflags[4 wg + wave]); such workgroups are recomputed by
//      bit 2 FIXUP: the general kernel (running maximum, deferred rescale: the shipped arithmetic bit for bit), which
//                  returns at once unless its workgroup was flagged.  The pair is launched back to back.
//      bit 1 (lab only): s_memtime stamps per segment.
// The fast form removes ~47 of ~127 vector instructions per 32 rows x 64 keys: profiles/r03_mfma_valu_overlap_microbench.log
// shows v_exp_f32 costing 8.4 cycles and every other vector op ~4.5 next to MFMAs, i.e. this kernel is bound by vector
// issue, not by the matrix pipe.
// NW = waves per workgroup (4: 256 query rows, two workgroups per CU; 8: 512 rows, one workgroup per CU -- every K / V^T tile
// then serves twice the rows: half the LDS-DMA pieces per wave and tile, whose issue costs a wave 100-150 cycles each).
// VROW: V arrives token-major like K (no V^T image, no pf_v_transpose pass): the V tile is staged row-major and the PV
// operand is read with the hardware transpose (attention.hip: vrow_fragment).
template <int OCC, int MODE = 0, int NW = 4, bool VROW = false>
__global__ __launch_bounds__(64 * NW, OCC) void attn64_kernel(const AArgs p) {
    constexpr int ABL = MODE;
    constexpr bool FAST = (MODE & 1) != 0, FIXUP = (MODE & 4) != 0;
    // bit 3 (lab only, not instantiated by the library; written at the end of round 3, not yet measured): row sums of the
    // bf16-ROUNDED P by v_dot2c_f32_bf16 against (1, 1) -- 4 instead of 8 vector instructions per slot, and the
    // denominator then sums exactly the values the PV MFMAs multiply
    constexpr bool DOTSUM = (MODE & 8) != 0;
    // bit 4 (lab only, likewise): the fp32 row sums as TWO partial sums per block added by v_pk_add_f32 (32 instead of 64 adds)
    constexpr bool PKSUM = (MODE & 16) != 0;
    // bit 5 MMSUM (FAST only): the row sums come from the MATRIX pipe.  The bf16 P fragment of a 16-key slot (the B operand of
    //       the PV MFMAs: lane l holds P[key 8 (l / 32) + j][row l % 32]) is fed once more to v_mfma_f32_16x16x32_bf16, which
    //       reads the same registers as B'[k' = 8 (l / 16) + j][n' = l % 16]; against the constant A[m][k'] = [(k' / 8) % 2 == m % 2]
    //       it returns D[m][n'] = the slot's sum of row n' (m even) or row n' + 16 (m odd) over BOTH key halves.  One 16-cycle
    //       MFMA per slot (+12.5 % matrix-pipe time) replaces 16 v_add_f32 per slot and lane (64 of ~160 vector instructions
    //       per 32 rows x 64 keys of a kernel that is bound by vector issue); the sums accumulate in 4 registers per block and
    //       are read out once, in the epilogue; the denominator then sums exactly the rounded values the numerator multiplies.
    constexpr bool MMSUM = (MODE & 32) != 0;
    // bit 6 SPLIT / bit 7 COMBINE (both forms of FAST | MMSUM): a launch with fewer workgroups than ~1.25 rounds of the chip
    //       (a sequence-parallel rank's few heads: 4 heads x 2 x 61 query tiles = 488 workgroups whose heaviest sees 242 key
    //       tiles, the mean 143) is cut along the KEYS: workgroup (query tile, part) runs key tiles [part * kv_chunk, ...) and
    //       parks its unnormalised fp32 O and row sums (no running maximum in the fast pass: parts simply ADD); the COMBINE
    //       launch (one workgroup per query tile, no key loop) adds the parts in part order, applies the fast pass's range
    //       check to the total and stores -- or flags the wave for the FIXUP launch, exactly like the unsplit pair.
    constexpr bool SPLIT = (MODE & 64) != 0, COMBINE = (MODE & 128) != 0;
    static_assert(!(SPLIT || COMBINE) || (FAST && MMSUM && !(SPLIT && COMBINE) && NW == 4), "SPLIT / COMBINE are forms of the fast pass");
    constexpr int PART_FLOATS = 18 * 256;          // per wave: 16 KiB of O (64 registers x 64 lanes) + 2 KiB of row sums
    static_assert(!MMSUM || (FAST && !DOTSUM && !PKSUM), "MMSUM is a form of the FAST pass");
    typedef float f32x2_t_ __attribute__((ext_vector_type(2)));
    using ps_t = typename std::conditional<PKSUM, f32x2_t_, float>::type;
    auto ps_zero = [] { if constexpr (PKSUM) return (f32x2_t_){0.f, 0.f}; else return 0.f; };
    auto ps_total = [](const ps_t& v) { if constexpr (PKSUM) return v[0] + v[1]; else return v; };
    constexpr int PJ = 8 / NW;                 // DMA pieces per wave of each of the K and V^T tiles (8 pieces of 8 rows each)
    constexpr int NT = NW / 2;                 // 128-row query tiles (the granularity of tile_kv_end / q_row_begin) per workgroup
    // FIXUP: only the waves the fast pass flagged recompute and store.  The pair is ALIAS-SAFE for the in-place form the DiT
    // uses (O == Q: same rows, same columns): a wave reads only its own 64 Q rows and writes only its own 64 O rows, the
    // fast pass does not store the rows of a wave it flags, so the Q rows a fix-up wave re-reads are still the caller's.
    bool mine = true;
    if (FIXUP) {
        const int* f = p.wgflags + NW * blockIdx.x;
        int any = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) any |= f[i];
        if (any == 0) return;
        mine = f[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)] != 0;
    }
    constexpr int QB2 = 64 * NW, QWAVE = 64 * HD * 2;             // 8 KiB of Q per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * ABUF (K | V^T tile ring) + NW * QWAVE (Q rows)
    char* const sq = smem + 2 * ABUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq2 = (p.nqt + NT - 1) / NT, qt0 = p.qt0 / NT;
    const int nq_run = nq2 - qt0;
    const int S = SPLIT ? p.nsplit : 1;
    const int nwg = nq_run * S * p.H * p.B;
    int t = xcd_remap(blockIdx.x, nwg);
#ifndef PF_ATTN_NO_LIGHT_LAST
    // LIGHT QUERY TILES LAST (round 6).  The order was (batch, head)-major with every head's query tiles heaviest first: an XCD's
    // 64 workgroup slots end a launch on the LAST head's tiles, whose heaviest workgroups started late and run twice the
    // average -- a tail of ~half a heavy workgroup (9 % of the launch at L = 15 488, more at shorter sequences).  Now every XCD
    // walks the HEAVY tiles of its heads first (same order: one head's K / V at a time in its L2) and the LIGHT tiles of all
    // its heads at the end (the history frames' rows: they see less than half of the keys, i.e. only the first part of K / V):
    // the launch ends on short workgroups that fill the slots while the last heavy ones finish.  A bijection of the same
    // workgroup list: nothing else changes.  "Light" = a tile group whose key range ends below half of the longest one
    // (batch entry 0's table, so that every workgroup derives the same permutation).
    // (not in the KV-split form: its COMBINE launch flags workgroups by block index for the FIXUP launch, and both must map
    //  a block index to the same tile; those launches are at most 1.25 rounds of the chip anyway)
    if (S == 1 && !COMBINE && p.nsplit <= 1) {
        int kmax = 0;
        for (int i = lane; i < p.nqt; i += 64) kmax = max(kmax, p.tile_kv_end[i]);
#pragma unroll
        for (int o_ = 1; o_ < 64; o_ <<= 1) kmax = max(kmax, __shfl_xor(kmax, o_));
        int nheavy = 0;                                        // tile groups [qt0, nq2) that are not light
        for (int g0 = qt0 + lane; g0 < nq2; g0 += 64) {
            int kv = 0;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (NT * g0 + i < p.nqt) kv = max(kv, p.tile_kv_end[NT * g0 + i]);
#ifndef PF_ATTN_LIGHT_NUM          // light = key range shorter than NUM / DEN of the longest (lab builds sweep it)
#define PF_ATTN_LIGHT_NUM 1
#define PF_ATTN_LIGHT_DEN 2
#endif
            nheavy += (PF_ATTN_LIGHT_DEN * kv >= PF_ATTN_LIGHT_NUM * kmax) ? 1 : 0;
        }
#pragma unroll
        for (int o_ = 1; o_ < 64; o_ <<= 1) nheavy += __shfl_xor(nheavy, o_);
        nheavy = __builtin_amdgcn_readfirstlane(nheavy);
        const int nh = nheavy, nl = nq_run - nheavy;
        if (nl > 0 && nh > 0) {
            // this XCD's chunk [c0, c0 + len) of the linear order u = bh * nq_run + qrank; position i inside it
            const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
            const int c0 = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
            const int c1 = c0 + q8 + (xcd < r8 ? 1 : 0);
            auto heavy_below = [&](int c) { return (c / nq_run) * nh + min(c % nq_run, nh); };       // heavy u in [0, c)
            const int h0 = heavy_below(c0), hn = heavy_below(c1) - h0;
            if (i < hn) {
                const int k = h0 + i;
                t = (k / nh) * nq_run + (k % nh);
            } else {
                const int k = (c0 - h0) + (i - hn);                                                       // light u in [0, c0) + ...
                t = (k / nl) * nq_run + nh + (k % nl);
            }
        }
    }
#endif
    const int bh = t / (nq_run * S);
    const int rem_ = t - bh * (nq_run * S);
    const int qrank = rem_ / S, part = rem_ - qrank * S;
    const int qt = nq2 - 1 - qrank;                  // heaviest (latest) q tiles first
    const int unit0 = (bh * nq_run + qrank) * ((SPLIT || COMBINE) ? p.nsplit : 1);      // first part slot of this query tile
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * QB2;
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;

    // ---- this lane's two query rows (block A: rows +0..31 of the wave's 64, block B: +32..63): mask intervals
    int alo[2], ahi[2], bhi[2];
    int wmax = 0, wmin = 0x7fffffff;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int qrow = q0 + wid * 64 + x * 32 + frow;
        alo[x] = ahi[x] = bhi[x] = 0;
        if (qrow < p.L) {
            alo[x] = p.a_lo[(long long)b * p.L + qrow];
            ahi[x] = p.a_hi[(long long)b * p.L + qrow];
            bhi[x] = p.b_hi[(long long)b * p.L + qrow];
            wmin = min(wmin, bhi[x]);
        }
        wmax = max(wmax, bhi[x]);
    }
    // the ordinary loads above are CONSUMED here, before the first LDS-DMA is issued: hipcc does not count DMA pieces and
    // would otherwise wait vmcnt(0) for these values at a first use inside the tile loop, draining the prefetch
#pragma unroll
    for (int x = 0; x < 2; ++x) asm volatile("" ::"v"(alo[x]), "v"(ahi[x]), "v"(bhi[x]));
#pragma unroll
    for (int o_ = 1; o_ < 64; o_ <<= 1) {
        wmax = max(wmax, __shfl_xor(wmax, o_));
        wmin = min(wmin, __shfl_xor(wmin, o_));
    }
    int kv_end = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i)
        if (NT * qt + i < p.nqt) kv_end = max(kv_end, p.tile_kv_end[b * p.nqt + NT * qt + i]);
    const int ntiles = (kv_end + KB - 1) / KB;
    // the key tiles [jt0, jt1) this workgroup walks: all of them; one part of them (SPLIT); none (COMBINE adds parked parts)
    int jt0 = 0, jt1 = ntiles;
    if (SPLIT) {
        jt0 = part * p.kv_chunk;
        jt1 = min(ntiles, jt0 + p.kv_chunk);
        if (jt0 >= jt1) return;                       // this query tile has fewer parts (workgroup-uniform, before any barrier)
    }
    if (COMBINE) jt1 = 0;
    const int wmax_s = __builtin_amdgcn_readfirstlane(wmax), wmin_s = __builtin_amdgcn_readfirstlane(wmin);
    // tiles this wave computes: the text tiles and every image tile below the largest visibility bound of its 64 rows
    // (bounds are monotone in the key index, so the active tiles are a prefix; the rest only keeps the barrier / DMA going)
    // rows below the caller's q_row_begin (p.qt0 counts 128-row tiles; a 256-row workgroup may start 128 rows earlier) are
    // neither computed, flagged nor stored; in the fix-up launch the same holds for the waves the fast pass did not flag
    const int row_lo = p.qt0 * QB;
    const bool wave_runs = mine && (q0 + wid * 64 + 64 > row_lo);
    const int my_nt = wave_runs ? min(ntiles, (max(p.Lt, wmax_s) + KB - 1) / KB) : 0;

    // ---- DMA sources: wave owns pieces i = wid*2 + j (rows 8i..8i+7) of the K and V^T tiles.  Addresses are a wave-uniform
    //      tile base (scalar registers, advanced per tile) + one constant 32-bit byte offset per lane and piece
    const char* const kbase = (const char*)(p.K + (long long)b * p.sK + h * p.hs_qk);
    const char* const vbase = VROW ? (const char*)(p.V + (long long)b * p.sV + h * p.hs_v)
                                   : (const char*)(p.Vt + (long long)b * p.sVb + (long long)h * p.sVh);
    const int ldk2 = p.ldk * 2, ldv2 = p.ldv * 2;
    int prow[PJ];
    unsigned pc2[PJ], voff[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int i = wid * PJ + j;
        prow[j] = 8 * i + (lane >> 3);
        pc2[j] = (unsigned)(((lane & 7) ^ (((i & 1) << 2) + (lane >> 4))) * 16);
        // V^T image: row = feature, keys along the row.  VROW: this lane's source chunk of key row prow[j]
        voff[j] = VROW ? (unsigned)(((lane & 7) ^ (((prow[j] >> 1) & 1) << 2)) * 16) : (unsigned)(prow[j] * p.Lp * 2) + pc2[j];
    }
    const unsigned vtr0 = vrow_lane_offset(lane, 0), vtr1 = vrow_lane_offset(lane, 1);
    auto issue_k = [&](int jt, int buf, int j) {
        const int last = p.L - 1 - jt * KB;                 // rows of the tile beyond the sequence re-read its last key
        const unsigned off = (unsigned)(min(prow[j], last) * ldk2) + pc2[j];
        glds16(kbase + (long long)jt * KB * ldk2 + off, smem + buf * ABUF + (wid * PJ + j) * 1024);
    };
    auto issue_v = [&](int jt, int buf, int j) {
        if constexpr (VROW) {
            const int last = p.L - 1 - jt * KB;             // keys beyond the sequence re-read the last one (their P is 0)
            const unsigned off = (unsigned)(min(prow[j], last) * ldv2) + voff[j];
            glds16(vbase + (long long)jt * KB * ldv2 + off, smem + buf * ABUF + KTILE + (wid * PJ + j) * 1024);
        } else {
            glds16(vbase + (long long)jt * (KB * 2) + voff[j], smem + buf * ABUF + KTILE + (wid * PJ + j) * 1024);
        }
    };
    // the wave's own 64 query rows -> LDS, same 128-byte-row image and chunk swizzle as a K tile (piece k = rows 8k..8k+7:
    // lane -> row 8k + lane/8, LDS chunk lane%8 holds source chunk (lane%8) ^ ((row >> 1) & 7)); only this wave reads them
    if (!COMBINE) {
        const char* qbase = (const char*)(p.Q + (long long)b * p.sQ + h * p.hs_qk);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = 8 * k + (lane >> 3);
            const int qrow = min(q0 + wid * 64 + r, p.L - 1);
            const unsigned c = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
            glds16(qbase + (long long)qrow * (p.ldq * 2) + c, sq + wid * QWAVE + k * 1024);
        }
    }

    f32x16_t o[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][i][r] = 0.f;
    float m[2] = {0.f, 0.f}, l[2] = {0.f, 0.f};
    f32x4_t lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // MMSUM: row sums as a 16 x 16 MFMA accumulator per block
    bf16x8_t ones_a;                                                      // MMSUM: the constant A operand (see above)
    {
        const bf16_t one_or_zero = (((lane >> 4) ^ lane) & 1) == 0 ? (bf16_t)1.0f : (bf16_t)0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones_a[e] = one_or_zero;
    }
    bool fresh[2] = {true, true};        // the row has not seen a key yet (its reference is still unset)
    const float NINF = -__builtin_inff();

    bf16x8_t kf[2][4], vf[2][4];
    // fragment read addresses: row frow (+32 for the second half: an immediate) of a 128-byte-row tile, 16-byte chunk
    // (2 ks + hi) ^ swz -- four per-lane byte offsets, the tile's base is wave-uniform
    unsigned foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = (unsigned)(frow * 128 + (((2 * ks + hi) ^ swz) << 4));
    auto read_k = [&](int buf) {
        const char* sk = smem + buf * ABUF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) kf[i][ks] = *(const bf16x8_t*)(sk + i * 4096 + foff[ks]);
    };
    auto read_v = [&](int buf) {
        const char* sv = smem + buf * ABUF + KTILE;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (VROW) {
                vf[0][g] = vrow_fragment(sv, vtr0, g);
                vf[1][g] = vrow_fragment(sv, vtr1, g);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) vf[i][g] = *(const bf16x8_t*)(sv + i * 4096 + foff[g]);
            }
        }
    };
    auto read_q = [&](int x, bf16x8_t* qf) {
        const char* s_ = sq + wid * QWAVE + x * 4096;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(s_ + foff[ks]);
    };
    // ---- the pieces of a tile, written as SEGMENTS in the order they are to issue (sched_barrier(0) between segments):
    // an MFMA group is issued first and the independent vector work of the segment runs while it executes.  A wave issues
    // in order: eight MFMAs written back to back occupy it for 8 x 32 cycles before the next vector instruction, so the
    // overlap has to be in program order, it does not come from the matrix pipe being asynchronous.
#define PF_SEG() __builtin_amdgcn_sched_barrier(0)
    // S^T = K . Q^T - m for block x, k-slice ks (both 32-key halves): scores of key i*32 + (r&3) + 8*(r>>2) + 4*hi vs row frow
    auto qk_slice = [&](int ks, const bf16x8_t* qf, const f32x16_t& negm, f32x16_t* s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][ks], qf[ks], ks == 0 ? negm : s[i], 0, 0, 0);
    };
    auto bcast = [&](float v) {
        f32x16_t t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = v;
        return t;
    };
    // visible keys of a row = [alo, ahi) (text part) U [Lt, bhi) (image part): two unsigned range tests per score, branch-free
    // and in place.  The per-register key index starts from a laundered value so that the 32 constants of this rare path are
    // not hoisted into loop-invariant registers (they pushed the common path into scratch memory).
    auto apply_mask = [&](int x, f32x16_t* s, int j0) {
        int kb = j0 + 4 * hi;
        asm volatile("" : "+v"(kb));
        const unsigned wa = (unsigned)(ahi[x] - alo[x]), wb = (unsigned)(bhi[x] - p.Lt);
        const int ka = kb - alo[x], kt = kb - p.Lt;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = i * 32 + (r & 3) + 8 * (r >> 2);
                const bool ok = ((unsigned)(ka + c) < wa) | ((unsigned)(kt + c) < wb);
                s[i][r] = ok ? s[i][r] : NINF;      // exp2(-inf) == 0: no select needed after the exponential
            }
    };
    // row maximum over registers [r0, r1) of both halves, folded into `mt`
    auto max_part = [&](const f32x16_t* s, int r0, int r1, float mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (r >= r0 && r < r1) mt = fmaxf(mt, fmaxf(s[0][r], s[1][r]));
        return mt;
    };
    // cross-half maximum + the deferred rescale (rare): everything between the row maximum and the exponentials
    auto rescale_check = [&](int x, f32x16_t* s, float mt) {
        {   // the row's other 32 keys live in lane ^ 32
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const bool seen = mt > NINF;
        if (__builtin_amdgcn_ballot_w64(mt > DEFER || (fresh[x] && seen)) != 0) {
            const float delta = fresh[x] ? (seen ? mt : 0.f) : fmaxf(mt, 0.f);
            const float alpha = fresh[x] ? 1.f : __builtin_amdgcn_exp2f(-delta);
            fresh[x] = fresh[x] && !seen;
            m[x] += delta;
            l[x] *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[x][i][r] *= alpha; s[i][r] -= delta; }
        }
    };
    // exponentials + partial row sum + bf16 P fragment of contraction slot g (keys 16 g .. 16 g + 15 in C-layout order)
    auto exp_slot = [&](int g, f32x16_t* s, ps_t& ps, bf16x8_t* pf) {
        [[maybe_unused]] float pv_even = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = __builtin_amdgcn_exp2f(s[g >> 1][8 * (g & 1) + e]);
            if constexpr (PKSUM) {
                if (e & 1) ps += (f32x2_t_){pv_even, v};
                else pv_even = v;
            } else if constexpr (!DOTSUM && !MMSUM) {
                ps += v;
            }
            pf[g][e] = (bf16_t)v;
        }
        if constexpr (DOTSUM && !PKSUM) {
            const bf16x2_t ones = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x2_t pr = {pf[g][2 * q], pf[g][2 * q + 1]};
                ps = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, ps, false);
            }
        }
        // pinned HERE: the fragment is only consumed by a later segment, and LLVM otherwise sinks the whole slot down to its
        // use (across the branches between the segments), which puts all exponentials behind the MFMAs they should run under
        asm volatile("" : "+v"(pf[g]), "+v"(ps));
    };
    auto pv_slot = [&](int x, int g, const bf16x8_t* pf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) o[x][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i][g], pf[g], o[x][i], 0, 0, 0);
        if constexpr (MMSUM) lacc[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pf[g], lacc[x], 0, 0, 0);
    };
    // a block's denominator at the end: the VALU partial sums of the two key halves, or the MFMA accumulator's entry for
    // this lane's row (register 0: rows 0..15 of the block, register 1: rows 16..31; both key halves already inside)
    auto row_sum = [&](int x) {
        if constexpr (MMSUM) return (lane & 16) ? lacc[x][1] : lacc[x][0];
        else { float lx = l[x]; lx += __shfl_xor(lx, 32); return lx; }
    };

    // ---- prologue: Q rows and tiles 0, 1 in flight; tile 0 and Q landed; first K / Q fragments requested
    if (jt1 > jt0) {
#pragma unroll
        for (int j = 0; j < PJ; ++j) { issue_k(jt0, jt0 & 1, j); issue_v(jt0, jt0 & 1, j); }
    }
    if (jt1 > jt0 + 1) {
#pragma unroll
        for (int j = 0; j < PJ; ++j) { issue_k(jt0 + 1, (jt0 + 1) & 1, j); issue_v(jt0 + 1, (jt0 + 1) & 1, j); }
        if (PJ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    bf16x8_t qfa[4], qfb[4];
    if (my_nt > jt0) { read_k(jt0 & 1); read_q(0, qfa); }

    unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if (ABL & 2) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
            ph[k] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (ABL & 2) tprev = (unsigned)__builtin_amdgcn_s_memtime();
    // one tile.  MASKED (wave-uniform): the tile straddles a visibility bound of some row of the wave (text tiles, frame
    // boundaries): a scalar branch around the mask code at two segment boundaries (two instantiations of the whole tile
    // body, one per case, made the register allocator spill ~240 registers).
    auto tile = [&](const bool MASKED, int jt) {
        const int buf = jt & 1, j0 = jt * KB;
        const bool active = jt < my_nt;
        f32x16_t sb[2];
        bf16x8_t pb[4];
        ps_t psb = ps_zero();
        if (active && FAST) {
            // FAST: no maximum, no shift.  The exponentials of a block run under the OTHER block's MFMAs:
            //   QK(A) | QK(B) + exp(A) slots 0-2 | PV(A) + exp(A) slot 3, exp(B) slots 0-1 | boundary | PV(B) + exp(B) slots 2-3 + DMA
            f32x16_t sa[2];
            bf16x8_t pa[4];
            const f32x16_t zero = bcast(0.f);
            ps_t psa = ps_zero();
            qk_slice(0, qfa, zero, sa);
            PF_SEG();
            read_q(1, qfb);
            PF_SEG();
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) qk_slice(ks, qfa, zero, sa);
            PF_SEG();
            stamp(0);
            if (MASKED) apply_mask(0, sa, j0);
            PF_SEG();
            qk_slice(0, qfb, zero, sb);
            exp_slot(0, sa, psa, pa);
            PF_SEG();
            qk_slice(1, qfb, zero, sb);
            exp_slot(1, sa, psa, pa);
            PF_SEG();
            qk_slice(2, qfb, zero, sb);
            read_v(buf);
            exp_slot(2, sa, psa, pa);
            PF_SEG();
            qk_slice(3, qfb, zero, sb);
            PF_SEG();
            stamp(1);
            if (MASKED) apply_mask(1, sb, j0);
            PF_SEG();
            pv_slot(0, 0, pa);
            exp_slot(3, sa, psa, pa);
            PF_SEG();
            pv_slot(0, 1, pa);
            exp_slot(0, sb, psb, pb);
            PF_SEG();
            pv_slot(0, 2, pa);
            exp_slot(1, sb, psb, pb);
            PF_SEG();
            pv_slot(0, 3, pa);
            l[0] += ps_total(psa);
            PF_SEG();
            stamp(2);
        }
        if (active && !FAST) {
            f32x16_t sa[2];
            bf16x8_t pa[4];
            // S0: QK^T(A); the Q fragments of block B are requested behind its first MFMA pair and land under the rest
            {
                const f32x16_t na = bcast(-m[0]);
                qk_slice(0, qfa, na, sa);
                PF_SEG();
                read_q(1, qfb);
                PF_SEG();
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) qk_slice(ks, qfa, na, sa);
            }
            PF_SEG();
            stamp(0);
            // S1: QK^T(B) in four slices, the row maximum of block A between them
            if (MASKED) apply_mask(0, sa, j0);
            PF_SEG();
            float mta = fmaxf(sa[0][0], sa[1][0]);
            {
                const f32x16_t nb = bcast(-m[1]);
                qk_slice(0, qfb, nb, sb);
            }
            mta = max_part(sa, 1, 6, mta);
            PF_SEG();
            qk_slice(1, qfb, sb[0], sb);
            mta = max_part(sa, 6, 11, mta);
            PF_SEG();
            qk_slice(2, qfb, sb[0], sb);
            mta = max_part(sa, 11, 16, mta);
            PF_SEG();
            qk_slice(3, qfb, sb[0], sb);
            read_v(buf);
            rescale_check(0, sa, mta);
            PF_SEG();
            stamp(1);
            // S2: exponentials of block A one contraction slot ahead of its PV MFMAs; the maximum of block B at the end
            ps_t psa = ps_zero();
            exp_slot(0, sa, psa, pa);
            PF_SEG();
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                pv_slot(0, g, pa);
                exp_slot(g + 1, sa, psa, pa);
                PF_SEG();
            }
            l[0] += ps_total(psa);
            pv_slot(0, 3, pa);
            stamp(2);
            if (MASKED) apply_mask(1, sb, j0);
            float mtb = fmaxf(sb[0][0], sb[1][0]);
            mtb = max_part(sb, 1, 16, mtb);
            rescale_check(1, sb, mtb);
            PF_SEG();
        }
        stamp(3);
        // ---- tile boundary: tile jt+1 has landed for everyone, every wave is done reading tile jt
        if (jt + 1 < jt1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stamp(4);
            if (jt + 1 < my_nt) { read_k(buf ^ 1); read_q(0, qfa); }
        }
        PF_SEG();
        stamp(5);
        const bool more = jt + 2 < jt1;
        if (active && FAST) {
            // S3 (FAST): PV(B) with the last two exponential slots of block B and the DMA pieces of tile jt+2 between its pairs
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pv_slot(1, g, pb);
                PF_SEG();
                if (g < 2) exp_slot(g + 2, sb, psb, pb);
                if (more) {
                    if (PJ == 2) {
                        if (g < 2) issue_k(jt + 2, buf, g);
                        else issue_v(jt + 2, buf, g - 2);
                    } else {
                        if (g == 0) issue_k(jt + 2, buf, 0);
                        if (g == 1) issue_v(jt + 2, buf, 0);
                    }
                }
                PF_SEG();
            }
            l[1] += ps_total(psb);
            stamp(6);
            if (ABL & 2) ph[7] += 1;
        } else if (active) {
            // S3: exponentials of block B one slot ahead of PV(B); the DMA pieces of tile jt+2 spread between the MFMA pairs
            // (a piece costs its issuing wave ~100 cycles on its own, ~60 next to MFMAs)
            exp_slot(0, sb, psb, pb);
            PF_SEG();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pv_slot(1, g, pb);
                PF_SEG();                      // the MFMA pair first: the exponentials of the next slot run while it executes
                if (g < 3) exp_slot(g + 1, sb, psb, pb);
                if (more) {
                    if (PJ == 2) {
                        if (g < 2) issue_k(jt + 2, buf, g);
                        else issue_v(jt + 2, buf, g - 2);
                    } else {
                        if (g == 0) issue_k(jt + 2, buf, 0);
                        if (g == 1) issue_v(jt + 2, buf, 0);
                    }
                }
                PF_SEG();
            }
            l[1] += ps_total(psb);
            stamp(6);
            if (ABL & 2) ph[7] += 1;
        } else if (more) {
#pragma unroll
            for (int j = 0; j < PJ; ++j) { issue_k(jt + 2, buf, j); issue_v(jt + 2, buf, j); }
        }
    };

    for (int jt = jt0; jt < jt1; ++jt) {
        const int j0 = jt * KB;
        const bool masked = (j0 < p.Lt) || (j0 + KB > wmin_s);      // scalar (wave-uniform)
        tile(masked, jt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((ABL & 2) && lane == 0 && p.dbg) {
        unsigned* d_ = p.dbg + ((size_t)blockIdx.x * NW + wid) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) d_[i] = ph[i];
    }
#undef PF_SEG

    // ---- SPLIT: park this part's unnormalised sums (lane-linear 16-byte pieces, 1 KiB per wave-instruction) and whether
    //      they are finite; the COMBINE launch owns the range check of the total, the flags and the store
    if constexpr (SPLIT) {
        if (!wave_runs) return;
        float* dst = p.parts + ((long long)(unit0 + part) * NW + wid) * PART_FLOATS + lane * 4;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(f32x4_t*)(dst + ((x * 2 + i) * 4 + q4) * 256) =
                        (f32x4_t){o[x][i][4 * q4], o[x][i][4 * q4 + 1], o[x][i][4 * q4 + 2], o[x][i][4 * q4 + 3]};
        *(f32x4_t*)(dst + 16 * 256) = lacc[0];
        *(f32x4_t*)(dst + 17 * 256) = lacc[1];
        bool nonfinite = false;
#pragma unroll
        for (int x = 0; x < 2; ++x) nonfinite = nonfinite || !(row_sum(x) < 1e30f);
        const int any_nf = __builtin_amdgcn_ballot_w64(nonfinite) != 0 ? 1 : 0;
        if (lane == 0) p.part_flags[(unit0 + part) * NW + wid] = any_nf;
        return;
    }
    bool part_bad = false;
    if constexpr (COMBINE) {
        if (wave_runs) {
            const int nparts = min(p.nsplit, (ntiles + p.kv_chunk - 1) / p.kv_chunk);
            for (int pt = 0; pt < nparts; ++pt) {                 // in part order: bitwise repeatable
                const float* src = p.parts + ((long long)(unit0 + pt) * NW + wid) * PART_FLOATS + lane * 4;
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4_t v4 = *(const f32x4_t*)(src + ((x * 2 + i) * 4 + q4) * 256);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[x][i][4 * q4 + e] += v4[e];
                        }
                lacc[0] += *(const f32x4_t*)(src + 16 * 256);
                lacc[1] += *(const f32x4_t*)(src + 17 * 256);
                part_bad = part_bad || p.part_flags[(unit0 + pt) * NW + wid] != 0;
            }
        }
    }

    // ---- FAST: did every valid row end with a usable denominator?  (inf / NaN: a score beyond the fp32 range of exp2; ~0: every
    //      score far below zero.)  One flag per wave; the FIXUP launch recomputes flagged workgroups with the running maximum.
    bool store = wave_runs;
    if (FAST) {
        bool bad = false;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float lx = row_sum(x);
            const int qrow = q0 + wid * 64 + x * 32 + frow;
            const bool valid = qrow < p.L && qrow >= row_lo;
            bad = bad || (valid && !(lx > 1e-30f && lx < 1e30f));
        }
        bad = bad || part_bad;
        const int any_bad = (wave_runs && __builtin_amdgcn_ballot_w64(bad) != 0) ? 1 : 0;
        if (lane == 0) p.wgflags[NW * blockIdx.x + wid] = any_bad;
        store = store && !any_bad;          // a flagged wave leaves its rows (== its Q rows when O aliases Q) to the fix-up launch
    }
    if (!store) return;

    // ---- epilogue: O = o / l, bf16, 16-byte stores (a lane pair (l, l^32) holds 8 consecutive features of a row after
    //      one half-exchange per register pair)
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float lx = row_sum(x);
        const float inv = lx > 0.f ? 1.0f / lx : 0.f;
        const int qrow = q0 + wid * 64 + x * 32 + frow;
        const bool qvalid = qrow < p.L && qrow >= row_lo;
        bf16_t* op = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + h * HD + 8 * hi;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q8 = 0; q8 < 2; ++q8) {
                // registers 8 q8 .. 8 q8 + 7 of o[x][i]: features i*32 + 16 q8 + {0..3 | 8..11} + 4 hi  (two groups of 4)
                unsigned a0 = pack2(o[x][i][8 * q8 + 0] * inv, o[x][i][8 * q8 + 1] * inv);
                unsigned a1 = pack2(o[x][i][8 * q8 + 2] * inv, o[x][i][8 * q8 + 3] * inv);
                unsigned b0 = pack2(o[x][i][8 * q8 + 4] * inv, o[x][i][8 * q8 + 5] * inv);
                unsigned b1 = pack2(o[x][i][8 * q8 + 6] * inv, o[x][i][8 * q8 + 7] * inv);
                // group a = features f0 + 4 hi + {0..3}, group b = f0 + 8 + 4 hi + {0..3} (f0 = i*32 + 16 q8).  After the
                // half-exchange lanes < 32 hold [a(hi=0) | a(hi=1)] = f0 .. f0+7, lanes >= 32 hold [b(hi=0) | b(hi=1)] = f0+8 ..
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4_t w;
                w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
                if (qvalid) *(u32x4_t*)(op + i * 32 + q8 * 16) = w;
            }
    }
}
