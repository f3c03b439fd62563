// lab/attn128_pipe.h -- the fast attention pass with ONE wave per SIMD and every hot-loop statement hand-placed.
// Included by lab/attn_lab.hip after pyramid-flow_amd/csrc/attention.hip (AArgs, the plan arrays, vrow_lane_offset, glds16).
//
// 4 waves / 512 query rows per workgroup, 128 rows (four 32-row blocks) per wave, 64-key KV tiles, K and V tiles (token-major)
// through three LDS buffers by LDS-DMA, no running maximum (FAST pass: flags the 64-row units whose row sums left the safe
// range, exactly as attn64_kernel<.., FAST> does, for the FIXUP pass), row sums by 16x16x32 MFMAs against a 0 / 1 operand.
//
// Register plan.  hipcc left to itself moves accumulators between the two register files (220 v_accvgpr per tile, 44 spills
// in the first version of this kernel), so the accumulator file is allocated BY HAND and addressed by name:
//     a[0:127]    O^T accumulators, block x half i at 32 x + 16 i          a[128:143]  row-sum accumulators of block x at 128 + 4 x
//     a[144:175]  K fragments, half i k-slice ks at 144 + 16 i + 4 ks      a[176:207]  V fragments, half i slot g at 176 + 16 i + 4 g
// and the compiler owns the arch VGPRs only (two S^T buffers, two P buffers, the Q fragments, addresses); the build check
// (lab/gen_attn128_body.py --check) is that the kernel's ISA holds no v_accvgpr_*, no AGPR operand and no scratch access outside
// the asm statements.
//
// Schedule.  One wave's vector ALU issues about 2.5 v_exp_f32 in the 32 cycles of a 32x32x16 MFMA (lab/mfma_valu_overlap.hip),
// and with one wave per SIMD nothing else hides a stall, so the loop is a software pipeline over blocks n = 4 jt + x in which
// EVERY MFMA is followed by one "quarter": the v_cvt_pk of the previous quarter's two exponentials, then two v_exp_f32.
//     step n:  QK(n + 1)  |  PV(n - 1)  |  exp(n)                                  (2 x 8 MFMAs + 4 row-sum MFMAs, 16 quarters)
// The V fragments of tile jt are read slot by slot in step A behind the PV(jt - 1, 3) MFMAs that free their registers, the K
// fragments of tile jt + 1 in step C behind the QK(jt, 3) MFMAs; the one barrier per tile sits in step B after four MFMAs.
// The loop body is GENERATED (lab/gen_attn128_body.py) so that the interleave is exact and reviewable.
//
// Hazards (cdna_hip_programming.md 5.7): a PV / row-sum statement follows the v_cvt_pk of its P operand by at least two MFMAs
// except for the last word of a block, which is converted one quarter ahead of its first reader (s_nop 1 in the statement);
// S is first read by the vector ALU two MFMAs after the statement that wrote it; a fragment register is rewritten by an LDS
// read issued at least one MFMA after the last MFMA that reads it; the epilogue waits 32 states before reading accumulators.
#pragma once
#define DEV_PIPE __device__ __forceinline__

template <int KS, int I>
DEV_PIPE void pq_qk(f32x16_t& s, const bf16x8_t& q) {                 // S^T half I += K(k-slice KS) Q^T(k-slice KS)
    constexpr int K0 = 144 + 16 * I + 4 * KS;
    if constexpr (KS == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], %3, 0" : "=&v"(s) : "n"(K0), "n"(K0 + 3), "v"(q));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%1:%2], %3, %0" : "+v"(s) : "n"(K0), "n"(K0 + 3), "v"(q));
}
template <int X, int I, int G>
DEV_PIPE void pq_pv(const u32x4_t& pfrag) {                            // O^T block X half I += V^T(slot G) P^T(slot G)
    constexpr int O0 = 32 * X + 16 * I, V0 = 176 + 16 * I + 4 * G;
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], a[%2:%3], %4, a[%0:%1]" ::"n"(O0), "n"(O0 + 15), "n"(V0), "n"(V0 + 3), "v"(pfrag));
}
// the same statements with their quarter (v_cvt_pk of the previous quarter, two v_exp_f32) in ONE asm block: between the
// statements of a block the compiler adds nothing (between separate asm statements it put an s_nop behind most MFMAs, and a
// sixth issue slot per MFMA gap costs ~4 cycles: MI355X_MICROARCH.md "one wave per SIMD")
#define PQ_QUARTER "\n\tv_cvt_pk_bf16_f32 %[w], %[ca], %[cb]\n\tv_exp_f32 %[ea], %[xa]\n\tv_exp_f32 %[eb], %[xb]"
template <int X>
DEV_PIPE void pq_ls(const bf16x8_t& ones, const u32x4_t& pfrag) {      // row sums of block X += (0 / 1 operand) P^T(slot)
    constexpr int L0 = 128 + 4 * X;
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %3, a[%0:%1]" ::"n"(L0), "n"(L0 + 3), "v"(ones), "v"(pfrag));
}
template <int R>
DEV_PIPE void pq_zero16(const bf16x8_t& z) {                           // a[R .. R + 15] = 0 (an MFMA of zero operands)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %2, 0" ::"n"(R), "n"(R + 15), "v"(z));
}
template <int R>
DEV_PIPE void pq_zero4(const bf16x8_t& z) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%0:%1], %2, %2, 0" ::"n"(R), "n"(R + 3), "v"(z));
}
template <int R>
DEV_PIPE f32x4_t pq_get4() {
    f32x4_t r;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
    return r;
}
template <int KS>
DEV_PIPE void pq_read_k(unsigned a) {                                  // K fragments of k-slice KS (both 32-key halves)
    asm volatile("ds_read_b128 a[%0:%1], %4\n\tds_read_b128 a[%2:%3], %4 offset:4096"
                 ::"n"(144 + 4 * KS), "n"(147 + 4 * KS), "n"(160 + 4 * KS), "n"(163 + 4 * KS), "v"(a) : "memory");
}
template <int G>
DEV_PIPE void pq_read_v(unsigned a0, unsigned a1) {                    // V fragments of 16-key slot G (both feature halves)
    asm volatile("ds_read_b64_tr_b16 a[%0:%1], %8 offset:%10\n\tds_read_b64_tr_b16 a[%2:%3], %8 offset:%11\n\t"
                 "ds_read_b64_tr_b16 a[%4:%5], %9 offset:%10\n\tds_read_b64_tr_b16 a[%6:%7], %9 offset:%11"
                 ::"n"(176 + 4 * G), "n"(177 + 4 * G), "n"(178 + 4 * G), "n"(179 + 4 * G),
                   "n"(192 + 4 * G), "n"(193 + 4 * G), "n"(194 + 4 * G), "n"(195 + 4 * G),
                   "v"(a0), "v"(a1), "n"(2048 * G), "n"(2048 * G + 1024) : "memory");
}

// VAR (timing experiments only, results wrong): 1 no V fragment reads, 2 no barrier, 3 no DMA of further tiles, 4 no K fragment reads, 5 MFMA and quarter as separate asm statements
template <int STAMP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void attn128_pipe_kernel(const AArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::: "a207");                                        // the accumulator file is in use up to here
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq5 = (p.nqt + 3) / 4;
    const int nwg = nq5 * p.H * p.B;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int bh = t / nq5;
    const int qt = nq5 - 1 - (t - bh * nq5);
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 512 + wid * 128;
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;

    int alo[4], ahi[4], bhi[4];
    int wmax = 0, wmin = 0x7fffffff;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int qrow = q0 + 32 * x + frow;
        alo[x] = ahi[x] = bhi[x] = 0;
        if (qrow < p.L) {
            alo[x] = p.a_lo[(long long)b * p.L + qrow];
            ahi[x] = p.a_hi[(long long)b * p.L + qrow];
            bhi[x] = p.b_hi[(long long)b * p.L + qrow];
            wmin = min(wmin, bhi[x]);
        }
        wmax = max(wmax, bhi[x]);
    }
    bf16x8_t qf[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int qr = min(q0 + 32 * x + frow, p.L - 1);
        const bf16_t* qp = p.Q + (long long)b * p.sQ + (long long)qr * p.ldq + h * p.hs_qk + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[x][ks] = *(const bf16x8_t*)(qp + ks * 16);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("" ::"v"(alo[x]), "v"(ahi[x]), "v"(bhi[x]));
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[x][ks]));      // loaded: O may alias Q from here on
#pragma unroll
    for (int o_ = 1; o_ < 64; o_ <<= 1) {
        wmax = max(wmax, __shfl_xor(wmax, o_));
        wmin = min(wmin, __shfl_xor(wmin, o_));
    }
    int kv_end = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * qt + i < p.nqt) kv_end = max(kv_end, p.tile_kv_end[b * p.nqt + 4 * qt + i]);
    const int ntiles = (kv_end + KB - 1) / KB;
    const int wmax_s = __builtin_amdgcn_readfirstlane(wmax), wmin_s = __builtin_amdgcn_readfirstlane(wmin);
    const int my_nt = min(ntiles, (max(p.Lt, wmax_s) + KB - 1) / KB);

    // LDS-DMA pieces: this wave moves 16 of the 64 token rows of a K tile and of a V tile (two 8-row pieces each); the
    // per-lane source pointers advance by one tile per issue, rows past the end of the sequence read the last row
    const int ldk2 = p.ldk * 2, ldv2 = p.ldv * 2;
    int prow[2];
    unsigned pc2[2], pcv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wid * 2 + j;
        prow[j] = 8 * i + (lane >> 3);
        pc2[j] = (unsigned)(((lane & 7) ^ (((i & 1) << 2) + (lane >> 4))) * 16);
        pcv[j] = (unsigned)(((lane & 7) ^ (((prow[j] >> 1) & 1) << 2)) * 16);
    }
    const char* const kbase = (const char*)(p.K + (long long)b * p.sK + h * p.hs_qk);
    const char* const vbase = (const char*)(p.V + (long long)b * p.sV + h * p.hs_v);
    // piece c of a tile: 0, 1 = K rows, 2, 3 = V rows.  Full tiles come from per-lane pointers that advance by one tile per issue
    // (one 64-bit add each); the last, partial tile of the sequence clamps its rows to the last one (rare path)
    const char* nxt[4];                                                // the pointers of the NEXT tile to request
    const long long kstep = (long long)KB * ldk2, vstep = (long long)KB * ldv2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        nxt[j] = kbase + (long long)prow[j] * ldk2 + pc2[j];
        nxt[2 + j] = vbase + (long long)prow[j] * ldv2 + pcv[j];
    }
    int nxt_jt = 0;                                                    // (all four pieces are issued per tile, in order 0 .. 3)
    auto issue_piece = [&](int c, unsigned bufoff) {
        const char* src = nxt[c];
        if ((nxt_jt + 1) * KB > p.L) {                                 // partial tile: rows past the end read the last row
            const int j = c & 1, r = min(prow[j], p.L - 1 - nxt_jt * KB);
            src = c < 2 ? kbase + (long long)nxt_jt * kstep + (long long)r * ldk2 + pc2[j] : vbase + (long long)nxt_jt * vstep + (long long)r * ldv2 + pcv[j];
        }
        glds16(src, smem + bufoff + (c >> 1) * KTILE + (wid * 2 + (c & 1)) * 1024);
        nxt[c] += c < 2 ? kstep : vstep;
        if (c == 3) ++nxt_jt;
    };
    auto issue_tile = [&](unsigned bufoff) { issue_piece(0, bufoff); issue_piece(1, bufoff); issue_piece(2, bufoff); issue_piece(3, bufoff); };
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = lds0 + (unsigned)(frow * 128 + (((2 * ks + hi) ^ swz) << 4));
    const unsigned voff0 = lds0 + KTILE + vrow_lane_offset(lane, 0), voff1 = lds0 + KTILE + vrow_lane_offset(lane, 1);

    bf16x8_t ones_a, zero8;
    {
        const bf16_t one_or_zero = (((lane >> 4) ^ lane) & 1) == 0 ? (bf16_t)1.0f : (bf16_t)0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ones_a[e] = one_or_zero; zero8[e] = (bf16_t)0.0f; }
        asm volatile("" : "+v"(zero8));
    }
    pq_zero16<0>(zero8); pq_zero16<16>(zero8); pq_zero16<32>(zero8); pq_zero16<48>(zero8);
    pq_zero16<64>(zero8); pq_zero16<80>(zero8); pq_zero16<96>(zero8); pq_zero16<112>(zero8);
    pq_zero4<128>(zero8); pq_zero4<132>(zero8); pq_zero4<136>(zero8); pq_zero4<140>(zero8);
    pq_zero16<176>(zero8); pq_zero16<192>(zero8);          // tile 0's step A runs PV(-1, 3): zero V fragments (and zero P) add nothing

    const float NINF = -__builtin_inff();
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
                s[i][r] = ok ? s[i][r] : NINF;
            }
    };
#define EXP(D, X) asm volatile("v_exp_f32 %0, %1" : "=v"(D) : "v"(X));
#define CVT(W, A, B) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(W) : "v"(A), "v"(B));
#define QKM(X, KS, I, S) pq_qk<KS, I>(S[I], qf[X][KS]);
#define PVM(X, I, G, PF) pq_pv<X, I, G>(PF[G]);
#define LSM(X, G, PF) pq_ls<X>(ones_a, PF[G]);
#define PQ_Q_OUT(W, EA, EB) [w] "=&v"(W), [ea] "=&v"(EA), [eb] "=&v"(EB)
#define PQ_Q_IN(CA, CB, XA, XB) [ca] "v"(CA), [cb] "v"(CB), [xa] "v"(XA), [xb] "v"(XB)
#define QKF0(X, KS, I, S, W, CA, CB, EA, XA, EB, XB)                                                                     \
    asm volatile("v_mfma_f32_32x32x16_bf16 %[s], a[%[k0]:%[k1]], %[q], 0" PQ_QUARTER                                    \
                 : [s] "=&v"(S[I]), PQ_Q_OUT(W, EA, EB)                                                                   \
                 : [k0] "n"(144 + 16 * (I) + 4 * (KS)), [k1] "n"(147 + 16 * (I) + 4 * (KS)), [q] "v"(qf[X][KS]), PQ_Q_IN(CA, CB, XA, XB));
#define QKF(X, KS, I, S, W, CA, CB, EA, XA, EB, XB)                                                                      \
    asm volatile("v_mfma_f32_32x32x16_bf16 %[s], a[%[k0]:%[k1]], %[q], %[s]" PQ_QUARTER                                 \
                 : [s] "+v"(S[I]), PQ_Q_OUT(W, EA, EB)                                                                    \
                 : [k0] "n"(144 + 16 * (I) + 4 * (KS)), [k1] "n"(147 + 16 * (I) + 4 * (KS)), [q] "v"(qf[X][KS]), PQ_Q_IN(CA, CB, XA, XB));
#define PVF(X, I, G, PF, W, CA, CB, EA, XA, EB, XB)                                                                      \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%[o0]:%[o1]], a[%[v0]:%[v1]], %[p], a[%[o0]:%[o1]]" PQ_QUARTER             \
                 : PQ_Q_OUT(W, EA, EB)                                                                                    \
                 : [o0] "n"(32 * (X) + 16 * (I)), [o1] "n"(32 * (X) + 16 * (I) + 15), [v0] "n"(176 + 16 * (I) + 4 * (G)),  \
                   [v1] "n"(179 + 16 * (I) + 4 * (G)), [p] "v"(PF[G]), PQ_Q_IN(CA, CB, XA, XB));
#define RDK(KS) if (VAR != 4) pq_read_k<KS>(foff[KS] + kbuf_next);
#define RDV(G) if (VAR != 1) pq_read_v<G>(voff0 + vbuf, voff1 + vbuf);
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define MASKB(X, S) if (masked) { asm volatile("s_nop 15\n\ts_nop 7"); apply_mask(X, S, j0); asm volatile("s_nop 1"); }
    // every wave holds its V fragments of tile jt (and read its K fragments a step ago): buffer jt is free; every wave's pieces
    // of tile jt + 1 have landed
#define BOUNDARY()                                                                                                               \
    if (jt + 1 < ntiles) {                                                                                                       \
        if (jt + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      /* tile jt + 2 stays in flight */  \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                         \
        if (VAR != 2) __builtin_amdgcn_s_barrier();                                                                              \
        if (VAR != 3 && jt + 3 < ntiles) issue_tile(vbuf);                                                                       \
    }

    // three LDS buffers: the tile after the next one stays in flight across a boundary (one wave per SIMD has nothing else to
    // hide the latency of the LDS-DMA behind)
    if (ntiles > 0) issue_tile(0);
    if (ntiles > 1) issue_tile(ABUF);
    if (ntiles > 2) issue_tile(2 * ABUF);
    if (ntiles > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16_t sa[2], sb[2];
    u32x4_t pa[4], pb[4];
    float e0a = 0.f, e0b = 0.f, e1a = 0.f, e1b = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) pb[g] = (u32x4_t){0u, 0u, 0u, 0u};
    if (my_nt > 0) {
        const unsigned kbuf_next = 0;
        RDK(0) RDK(1) RDK(2) RDK(3)
        LGKM0()
        QKM(0, 0, 0, sa) QKM(0, 0, 1, sa) QKM(0, 1, 0, sa) QKM(0, 1, 1, sa)
        QKM(0, 2, 0, sa) QKM(0, 2, 1, sa) QKM(0, 3, 0, sa) QKM(0, 3, 1, sa)
    }
    unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int k) {
        if (STAMP > 1) {
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
            ph[k] += now - tprev;
            tprev = now;
        }
    };
    const unsigned t_loop0 = STAMP ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
    tprev = t_loop0;
    unsigned vbuf = 0, kbuf_next = ABUF, buf_after = 2 * ABUF;     // LDS offsets of the buffers of tiles jt, jt + 1, jt + 2
    for (int jt = 0; jt < my_nt; ++jt) {
        const int j0 = jt * KB;
        const bool masked = (j0 < p.Lt) || (j0 + KB > wmin_s);
        if constexpr (VAR == 5) {
        // GENERATED PAIR BODY BEGIN (lab/gen_attn128_body.py, separate statements)
        // ---- step A: QK(jt, 1) -> sb, PV(jt - 1, 3) from pb, exp(sa) -> pa; the V fragments of tile jt slot by slot behind
        //      the PV MFMAs that read the old ones
        QKM(1, 0, 0, sb)
        MASKB(0, sa)
        CVT(pb[3][3], e1a, e1b) EXP(e0a, sa[0][0]) EXP(e0b, sa[0][1])
        PVM(3, 0, 0, pb)
        CVT(pa[0][0], e0a, e0b) EXP(e1a, sa[0][2]) EXP(e1b, sa[0][3])
        QKM(1, 0, 1, sb)
        CVT(pa[0][1], e1a, e1b) EXP(e0a, sa[0][4]) EXP(e0b, sa[0][5])
        PVM(3, 1, 0, pb)
        CVT(pa[0][2], e0a, e0b) EXP(e1a, sa[0][6]) EXP(e1b, sa[0][7])
        LSM(3, 0, pb)
        QKM(1, 1, 0, sb)
        CVT(pa[0][3], e1a, e1b) EXP(e0a, sa[0][8]) EXP(e0b, sa[0][9])
        RDV(0)
        PVM(3, 0, 1, pb)
        CVT(pa[1][0], e0a, e0b) EXP(e1a, sa[0][10]) EXP(e1b, sa[0][11])
        QKM(1, 1, 1, sb)
        CVT(pa[1][1], e1a, e1b) EXP(e0a, sa[0][12]) EXP(e0b, sa[0][13])
        PVM(3, 1, 1, pb)
        CVT(pa[1][2], e0a, e0b) EXP(e1a, sa[0][14]) EXP(e1b, sa[0][15])
        LSM(3, 1, pb)
        QKM(1, 2, 0, sb)
        CVT(pa[1][3], e1a, e1b) EXP(e0a, sa[1][0]) EXP(e0b, sa[1][1])
        RDV(1)
        PVM(3, 0, 2, pb)
        CVT(pa[2][0], e0a, e0b) EXP(e1a, sa[1][2]) EXP(e1b, sa[1][3])
        QKM(1, 2, 1, sb)
        CVT(pa[2][1], e1a, e1b) EXP(e0a, sa[1][4]) EXP(e0b, sa[1][5])
        PVM(3, 1, 2, pb)
        CVT(pa[2][2], e0a, e0b) EXP(e1a, sa[1][6]) EXP(e1b, sa[1][7])
        LSM(3, 2, pb)
        QKM(1, 3, 0, sb)
        CVT(pa[2][3], e1a, e1b) EXP(e0a, sa[1][8]) EXP(e0b, sa[1][9])
        RDV(2)
        PVM(3, 0, 3, pb)
        CVT(pa[3][0], e0a, e0b) EXP(e1a, sa[1][10]) EXP(e1b, sa[1][11])
        QKM(1, 3, 1, sb)
        CVT(pa[3][1], e1a, e1b) EXP(e0a, sa[1][12]) EXP(e0b, sa[1][13])
        PVM(3, 1, 3, pb)
        CVT(pa[3][2], e0a, e0b) EXP(e1a, sa[1][14]) EXP(e1b, sa[1][15])
        LSM(3, 3, pb)
        stamp(0);
        // ---- step B: QK(jt, 2) -> sa, exp(sb) -> pb; the tile boundary; PV(jt, 0) from pa
        QKM(2, 0, 0, sa)
        MASKB(1, sb)
        CVT(pa[3][3], e1a, e1b) EXP(e0a, sb[0][0]) EXP(e0b, sb[0][1])
        RDV(3)
        QKM(2, 0, 1, sa)
        CVT(pb[0][0], e0a, e0b) EXP(e1a, sb[0][2]) EXP(e1b, sb[0][3])
        QKM(2, 1, 0, sa)
        CVT(pb[0][1], e1a, e1b) EXP(e0a, sb[0][4]) EXP(e0b, sb[0][5])
        QKM(2, 1, 1, sa)
        CVT(pb[0][2], e0a, e0b) EXP(e1a, sb[0][6]) EXP(e1b, sb[0][7])
        BOUNDARY()
        LGKM0()
        PVM(0, 0, 0, pa)
        CVT(pb[0][3], e1a, e1b) EXP(e0a, sb[0][8]) EXP(e0b, sb[0][9])
        PVM(0, 1, 0, pa)
        CVT(pb[1][0], e0a, e0b) EXP(e1a, sb[0][10]) EXP(e1b, sb[0][11])
        LSM(0, 0, pa)
        QKM(2, 2, 0, sa)
        CVT(pb[1][1], e1a, e1b) EXP(e0a, sb[0][12]) EXP(e0b, sb[0][13])
        PVM(0, 0, 1, pa)
        CVT(pb[1][2], e0a, e0b) EXP(e1a, sb[0][14]) EXP(e1b, sb[0][15])
        QKM(2, 2, 1, sa)
        CVT(pb[1][3], e1a, e1b) EXP(e0a, sb[1][0]) EXP(e0b, sb[1][1])
        PVM(0, 1, 1, pa)
        CVT(pb[2][0], e0a, e0b) EXP(e1a, sb[1][2]) EXP(e1b, sb[1][3])
        LSM(0, 1, pa)
        QKM(2, 3, 0, sa)
        CVT(pb[2][1], e1a, e1b) EXP(e0a, sb[1][4]) EXP(e0b, sb[1][5])
        PVM(0, 0, 2, pa)
        CVT(pb[2][2], e0a, e0b) EXP(e1a, sb[1][6]) EXP(e1b, sb[1][7])
        QKM(2, 3, 1, sa)
        CVT(pb[2][3], e1a, e1b) EXP(e0a, sb[1][8]) EXP(e0b, sb[1][9])
        PVM(0, 1, 2, pa)
        CVT(pb[3][0], e0a, e0b) EXP(e1a, sb[1][10]) EXP(e1b, sb[1][11])
        LSM(0, 2, pa)
        PVM(0, 0, 3, pa)
        CVT(pb[3][1], e1a, e1b) EXP(e0a, sb[1][12]) EXP(e0b, sb[1][13])
        PVM(0, 1, 3, pa)
        CVT(pb[3][2], e0a, e0b) EXP(e1a, sb[1][14]) EXP(e1b, sb[1][15])
        LSM(0, 3, pa)
        stamp(1);
        // ---- step C: QK(jt, 3) -> sb, PV(jt, 1) from pb, exp(sa) -> pa; the K fragments of tile jt + 1 k-slice by k-slice
        //      behind the QK MFMAs that read the old ones
        QKM(3, 0, 0, sb)
        MASKB(2, sa)
        CVT(pb[3][3], e1a, e1b) EXP(e0a, sa[0][0]) EXP(e0b, sa[0][1])
        PVM(1, 0, 0, pb)
        CVT(pa[0][0], e0a, e0b) EXP(e1a, sa[0][2]) EXP(e1b, sa[0][3])
        QKM(3, 0, 1, sb)
        CVT(pa[0][1], e1a, e1b) EXP(e0a, sa[0][4]) EXP(e0b, sa[0][5])
        PVM(1, 1, 0, pb)
        CVT(pa[0][2], e0a, e0b) EXP(e1a, sa[0][6]) EXP(e1b, sa[0][7])
        RDK(0)
        LSM(1, 0, pb)
        QKM(3, 1, 0, sb)
        CVT(pa[0][3], e1a, e1b) EXP(e0a, sa[0][8]) EXP(e0b, sa[0][9])
        PVM(1, 0, 1, pb)
        CVT(pa[1][0], e0a, e0b) EXP(e1a, sa[0][10]) EXP(e1b, sa[0][11])
        QKM(3, 1, 1, sb)
        CVT(pa[1][1], e1a, e1b) EXP(e0a, sa[0][12]) EXP(e0b, sa[0][13])
        PVM(1, 1, 1, pb)
        CVT(pa[1][2], e0a, e0b) EXP(e1a, sa[0][14]) EXP(e1b, sa[0][15])
        RDK(1)
        LSM(1, 1, pb)
        QKM(3, 2, 0, sb)
        CVT(pa[1][3], e1a, e1b) EXP(e0a, sa[1][0]) EXP(e0b, sa[1][1])
        PVM(1, 0, 2, pb)
        CVT(pa[2][0], e0a, e0b) EXP(e1a, sa[1][2]) EXP(e1b, sa[1][3])
        QKM(3, 2, 1, sb)
        CVT(pa[2][1], e1a, e1b) EXP(e0a, sa[1][4]) EXP(e0b, sa[1][5])
        PVM(1, 1, 2, pb)
        CVT(pa[2][2], e0a, e0b) EXP(e1a, sa[1][6]) EXP(e1b, sa[1][7])
        RDK(2)
        LSM(1, 2, pb)
        QKM(3, 3, 0, sb)
        CVT(pa[2][3], e1a, e1b) EXP(e0a, sa[1][8]) EXP(e0b, sa[1][9])
        PVM(1, 0, 3, pb)
        CVT(pa[3][0], e0a, e0b) EXP(e1a, sa[1][10]) EXP(e1b, sa[1][11])
        QKM(3, 3, 1, sb)
        CVT(pa[3][1], e1a, e1b) EXP(e0a, sa[1][12]) EXP(e0b, sa[1][13])
        PVM(1, 1, 3, pb)
        CVT(pa[3][2], e0a, e0b) EXP(e1a, sa[1][14]) EXP(e1b, sa[1][15])
        LSM(1, 3, pb)
        RDK(3)
        stamp(2);
        // ---- step D: PV(jt, 2) from pa, QK(jt + 1, 0) -> sa (on the last tile: of stale K fragments, never read), exp(sb) -> pb
        PVM(2, 0, 0, pa)
        MASKB(3, sb)
        CVT(pa[3][3], e1a, e1b) EXP(e0a, sb[0][0]) EXP(e0b, sb[0][1])
        PVM(2, 1, 0, pa)
        CVT(pb[0][0], e0a, e0b) EXP(e1a, sb[0][2]) EXP(e1b, sb[0][3])
        LSM(2, 0, pa)
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        QKM(0, 0, 0, sa)
        CVT(pb[0][1], e1a, e1b) EXP(e0a, sb[0][4]) EXP(e0b, sb[0][5])
        QKM(0, 0, 1, sa)
        CVT(pb[0][2], e0a, e0b) EXP(e1a, sb[0][6]) EXP(e1b, sb[0][7])
        QKM(0, 1, 0, sa)
        CVT(pb[0][3], e1a, e1b) EXP(e0a, sb[0][8]) EXP(e0b, sb[0][9])
        PVM(2, 0, 1, pa)
        CVT(pb[1][0], e0a, e0b) EXP(e1a, sb[0][10]) EXP(e1b, sb[0][11])
        QKM(0, 1, 1, sa)
        CVT(pb[1][1], e1a, e1b) EXP(e0a, sb[0][12]) EXP(e0b, sb[0][13])
        PVM(2, 1, 1, pa)
        CVT(pb[1][2], e0a, e0b) EXP(e1a, sb[0][14]) EXP(e1b, sb[0][15])
        LSM(2, 1, pa)
        QKM(0, 2, 0, sa)
        CVT(pb[1][3], e1a, e1b) EXP(e0a, sb[1][0]) EXP(e0b, sb[1][1])
        PVM(2, 0, 2, pa)
        CVT(pb[2][0], e0a, e0b) EXP(e1a, sb[1][2]) EXP(e1b, sb[1][3])
        QKM(0, 2, 1, sa)
        CVT(pb[2][1], e1a, e1b) EXP(e0a, sb[1][4]) EXP(e0b, sb[1][5])
        PVM(2, 1, 2, pa)
        CVT(pb[2][2], e0a, e0b) EXP(e1a, sb[1][6]) EXP(e1b, sb[1][7])
        LSM(2, 2, pa)
        LGKM0()
        QKM(0, 3, 0, sa)
        CVT(pb[2][3], e1a, e1b) EXP(e0a, sb[1][8]) EXP(e0b, sb[1][9])
        PVM(2, 0, 3, pa)
        CVT(pb[3][0], e0a, e0b) EXP(e1a, sb[1][10]) EXP(e1b, sb[1][11])
        QKM(0, 3, 1, sa)
        CVT(pb[3][1], e1a, e1b) EXP(e0a, sb[1][12]) EXP(e0b, sb[1][13])
        PVM(2, 1, 3, pa)
        CVT(pb[3][2], e0a, e0b) EXP(e1a, sb[1][14]) EXP(e1b, sb[1][15])
        LSM(2, 3, pa)
        stamp(3);
        // GENERATED PAIR BODY END
        } else {
        // GENERATED BODY BEGIN (lab/gen_attn128_body.py)
        // ---- step A: QK(jt, 1) -> sb, PV(jt - 1, 3) from pb, exp(sa) -> pa; the V fragments of tile jt slot by slot behind
        //      the PV MFMAs that read the old ones
        QKM(1, 0, 0, sb)
        MASKB(0, sa)
        CVT(pb[3][3], e1a, e1b) EXP(e0a, sa[0][0]) EXP(e0b, sa[0][1])
        PVF(3, 0, 0, pb, pa[0][0], e0a, e0b, e1a, sa[0][2], e1b, sa[0][3])
        QKF0(1, 0, 1, sb, pa[0][1], e1a, e1b, e0a, sa[0][4], e0b, sa[0][5])
        PVF(3, 1, 0, pb, pa[0][2], e0a, e0b, e1a, sa[0][6], e1b, sa[0][7])
        LSM(3, 0, pb)
        QKF(1, 1, 0, sb, pa[0][3], e1a, e1b, e0a, sa[0][8], e0b, sa[0][9])
        RDV(0)
        PVF(3, 0, 1, pb, pa[1][0], e0a, e0b, e1a, sa[0][10], e1b, sa[0][11])
        QKF(1, 1, 1, sb, pa[1][1], e1a, e1b, e0a, sa[0][12], e0b, sa[0][13])
        PVF(3, 1, 1, pb, pa[1][2], e0a, e0b, e1a, sa[0][14], e1b, sa[0][15])
        LSM(3, 1, pb)
        QKF(1, 2, 0, sb, pa[1][3], e1a, e1b, e0a, sa[1][0], e0b, sa[1][1])
        RDV(1)
        PVF(3, 0, 2, pb, pa[2][0], e0a, e0b, e1a, sa[1][2], e1b, sa[1][3])
        QKF(1, 2, 1, sb, pa[2][1], e1a, e1b, e0a, sa[1][4], e0b, sa[1][5])
        PVF(3, 1, 2, pb, pa[2][2], e0a, e0b, e1a, sa[1][6], e1b, sa[1][7])
        LSM(3, 2, pb)
        QKF(1, 3, 0, sb, pa[2][3], e1a, e1b, e0a, sa[1][8], e0b, sa[1][9])
        RDV(2)
        PVF(3, 0, 3, pb, pa[3][0], e0a, e0b, e1a, sa[1][10], e1b, sa[1][11])
        QKF(1, 3, 1, sb, pa[3][1], e1a, e1b, e0a, sa[1][12], e0b, sa[1][13])
        PVF(3, 1, 3, pb, pa[3][2], e0a, e0b, e1a, sa[1][14], e1b, sa[1][15])
        LSM(3, 3, pb)
        stamp(0);
        // ---- step B: QK(jt, 2) -> sa, exp(sb) -> pb; the tile boundary; PV(jt, 0) from pa
        QKM(2, 0, 0, sa)
        MASKB(1, sb)
        CVT(pa[3][3], e1a, e1b) EXP(e0a, sb[0][0]) EXP(e0b, sb[0][1])
        RDV(3)
        QKF0(2, 0, 1, sa, pb[0][0], e0a, e0b, e1a, sb[0][2], e1b, sb[0][3])
        QKF(2, 1, 0, sa, pb[0][1], e1a, e1b, e0a, sb[0][4], e0b, sb[0][5])
        QKF(2, 1, 1, sa, pb[0][2], e0a, e0b, e1a, sb[0][6], e1b, sb[0][7])
        BOUNDARY()
        LGKM0()
        PVF(0, 0, 0, pa, pb[0][3], e1a, e1b, e0a, sb[0][8], e0b, sb[0][9])
        PVF(0, 1, 0, pa, pb[1][0], e0a, e0b, e1a, sb[0][10], e1b, sb[0][11])
        LSM(0, 0, pa)
        QKF(2, 2, 0, sa, pb[1][1], e1a, e1b, e0a, sb[0][12], e0b, sb[0][13])
        PVF(0, 0, 1, pa, pb[1][2], e0a, e0b, e1a, sb[0][14], e1b, sb[0][15])
        QKF(2, 2, 1, sa, pb[1][3], e1a, e1b, e0a, sb[1][0], e0b, sb[1][1])
        PVF(0, 1, 1, pa, pb[2][0], e0a, e0b, e1a, sb[1][2], e1b, sb[1][3])
        LSM(0, 1, pa)
        QKF(2, 3, 0, sa, pb[2][1], e1a, e1b, e0a, sb[1][4], e0b, sb[1][5])
        PVF(0, 0, 2, pa, pb[2][2], e0a, e0b, e1a, sb[1][6], e1b, sb[1][7])
        QKF(2, 3, 1, sa, pb[2][3], e1a, e1b, e0a, sb[1][8], e0b, sb[1][9])
        PVF(0, 1, 2, pa, pb[3][0], e0a, e0b, e1a, sb[1][10], e1b, sb[1][11])
        LSM(0, 2, pa)
        PVF(0, 0, 3, pa, pb[3][1], e1a, e1b, e0a, sb[1][12], e0b, sb[1][13])
        PVF(0, 1, 3, pa, pb[3][2], e0a, e0b, e1a, sb[1][14], e1b, sb[1][15])
        LSM(0, 3, pa)
        stamp(1);
        // ---- step C: QK(jt, 3) -> sb, PV(jt, 1) from pb, exp(sa) -> pa; the K fragments of tile jt + 1 k-slice by k-slice
        //      behind the QK MFMAs that read the old ones
        QKM(3, 0, 0, sb)
        MASKB(2, sa)
        CVT(pb[3][3], e1a, e1b) EXP(e0a, sa[0][0]) EXP(e0b, sa[0][1])
        PVF(1, 0, 0, pb, pa[0][0], e0a, e0b, e1a, sa[0][2], e1b, sa[0][3])
        QKF0(3, 0, 1, sb, pa[0][1], e1a, e1b, e0a, sa[0][4], e0b, sa[0][5])
        PVF(1, 1, 0, pb, pa[0][2], e0a, e0b, e1a, sa[0][6], e1b, sa[0][7])
        RDK(0)
        LSM(1, 0, pb)
        QKF(3, 1, 0, sb, pa[0][3], e1a, e1b, e0a, sa[0][8], e0b, sa[0][9])
        PVF(1, 0, 1, pb, pa[1][0], e0a, e0b, e1a, sa[0][10], e1b, sa[0][11])
        QKF(3, 1, 1, sb, pa[1][1], e1a, e1b, e0a, sa[0][12], e0b, sa[0][13])
        PVF(1, 1, 1, pb, pa[1][2], e0a, e0b, e1a, sa[0][14], e1b, sa[0][15])
        RDK(1)
        LSM(1, 1, pb)
        QKF(3, 2, 0, sb, pa[1][3], e1a, e1b, e0a, sa[1][0], e0b, sa[1][1])
        PVF(1, 0, 2, pb, pa[2][0], e0a, e0b, e1a, sa[1][2], e1b, sa[1][3])
        QKF(3, 2, 1, sb, pa[2][1], e1a, e1b, e0a, sa[1][4], e0b, sa[1][5])
        PVF(1, 1, 2, pb, pa[2][2], e0a, e0b, e1a, sa[1][6], e1b, sa[1][7])
        RDK(2)
        LSM(1, 2, pb)
        QKF(3, 3, 0, sb, pa[2][3], e1a, e1b, e0a, sa[1][8], e0b, sa[1][9])
        PVF(1, 0, 3, pb, pa[3][0], e0a, e0b, e1a, sa[1][10], e1b, sa[1][11])
        QKF(3, 3, 1, sb, pa[3][1], e1a, e1b, e0a, sa[1][12], e0b, sa[1][13])
        PVF(1, 1, 3, pb, pa[3][2], e0a, e0b, e1a, sa[1][14], e1b, sa[1][15])
        LSM(1, 3, pb)
        RDK(3)
        stamp(2);
        // ---- step D: PV(jt, 2) from pa, QK(jt + 1, 0) -> sa (on the last tile: of stale K fragments, never read), exp(sb) -> pb
        PVM(2, 0, 0, pa)
        MASKB(3, sb)
        CVT(pa[3][3], e1a, e1b) EXP(e0a, sb[0][0]) EXP(e0b, sb[0][1])
        PVF(2, 1, 0, pa, pb[0][0], e0a, e0b, e1a, sb[0][2], e1b, sb[0][3])
        LSM(2, 0, pa)
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        QKF0(0, 0, 0, sa, pb[0][1], e1a, e1b, e0a, sb[0][4], e0b, sb[0][5])
        QKF0(0, 0, 1, sa, pb[0][2], e0a, e0b, e1a, sb[0][6], e1b, sb[0][7])
        QKF(0, 1, 0, sa, pb[0][3], e1a, e1b, e0a, sb[0][8], e0b, sb[0][9])
        PVF(2, 0, 1, pa, pb[1][0], e0a, e0b, e1a, sb[0][10], e1b, sb[0][11])
        QKF(0, 1, 1, sa, pb[1][1], e1a, e1b, e0a, sb[0][12], e0b, sb[0][13])
        PVF(2, 1, 1, pa, pb[1][2], e0a, e0b, e1a, sb[0][14], e1b, sb[0][15])
        LSM(2, 1, pa)
        QKF(0, 2, 0, sa, pb[1][3], e1a, e1b, e0a, sb[1][0], e0b, sb[1][1])
        PVF(2, 0, 2, pa, pb[2][0], e0a, e0b, e1a, sb[1][2], e1b, sb[1][3])
        QKF(0, 2, 1, sa, pb[2][1], e1a, e1b, e0a, sb[1][4], e0b, sb[1][5])
        PVF(2, 1, 2, pa, pb[2][2], e0a, e0b, e1a, sb[1][6], e1b, sb[1][7])
        LSM(2, 2, pa)
        LGKM0()
        QKF(0, 3, 0, sa, pb[2][3], e1a, e1b, e0a, sb[1][8], e0b, sb[1][9])
        PVF(2, 0, 3, pa, pb[3][0], e0a, e0b, e1a, sb[1][10], e1b, sb[1][11])
        QKF(0, 3, 1, sa, pb[3][1], e1a, e1b, e0a, sb[1][12], e0b, sb[1][13])
        PVF(2, 1, 3, pa, pb[3][2], e0a, e0b, e1a, sb[1][14], e1b, sb[1][15])
        LSM(2, 3, pa)
        stamp(3);
        // GENERATED BODY END
        }
        { const unsigned t_ = vbuf; vbuf = kbuf_next; kbuf_next = buf_after; buf_after = t_; }
    }
    if (STAMP && lane == 0) {
        const unsigned t_loop1 = (unsigned)__builtin_amdgcn_s_memtime();
        unsigned* d = p.dbg + ((size_t)blockIdx.x * 4 + wid) * 8;
        for (int i = 0; i < 5; ++i) d[i] = ph[i];
        d[6] = t_loop1 - t_loop0;
        d[7] = (unsigned)my_nt;
    }
    if (my_nt > 0) {
        CVT(pb[3][3], e1a, e1b)                          // the last quarter of the last block (the generator asserts the temp pair)
        PVM(3, 0, 0, pb) PVM(3, 1, 0, pb) LSM(3, 0, pb) PVM(3, 0, 1, pb) PVM(3, 1, 1, pb) LSM(3, 1, pb)
        PVM(3, 0, 2, pb) PVM(3, 1, 2, pb) LSM(3, 2, pb) PVM(3, 0, 3, pb) PVM(3, 1, 3, pb) LSM(3, 3, pb)
    }
    // waves whose rows end before the workgroup's last KV tile keep the barrier / DMA schedule of the tiles they skip
    for (int jt = my_nt; jt + 1 < ntiles; ++jt) {
        BOUNDARY()
        { const unsigned t_ = vbuf; vbuf = kbuf_next; kbuf_next = buf_after; buf_after = t_; }
    }
#undef EXP
#undef CVT
#undef QKM
#undef PVM
#undef LSM
#undef QKF0
#undef QKF
#undef PVF
#undef PQ_Q_OUT
#undef PQ_Q_IN
#undef RDK
#undef RDV
#undef LGKM0
#undef MASKB
#undef BOUNDARY
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    float lsum[4];
    {
        const f32x4_t l0 = pq_get4<128>(), l1 = pq_get4<132>(), l2 = pq_get4<136>(), l3 = pq_get4<140>();
        lsum[0] = (lane & 16) ? l0[1] : l0[0];
        lsum[1] = (lane & 16) ? l1[1] : l1[0];
        lsum[2] = (lane & 16) ? l2[1] : l2[0];
        lsum[3] = (lane & 16) ? l3[1] : l3[0];
    }
    // flags of the 64-row units (two blocks) whose row sums left the range in which the pass is exact enough: the FIXUP pass
    // of attention_w64.h recomputes them (same flag layout as attn64_kernel: [bh][256-row tile][4 units])
    const int nq2 = (p.nqt + 1) / 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        bool bad = false;
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) {
            const int x = 2 * u + x2;
            const int qrow = q0 + 32 * x + frow;
            bad = bad || (qrow < p.L && !(lsum[x] > 1e-30f && lsum[x] < 1e30f));
        }
        const int any_bad = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
        const int qt256 = qt * 2 + (wid >> 1), w64 = (wid & 1) * 2 + u;
        if (lane == 0 && qt256 < nq2) p.wgflags[((long long)bh * nq2 + qt256) * 4 + w64] = any_bad;
    }
    auto store_half = [&](int x, int i, const f32x4_t& c0, const f32x4_t& c1, const f32x4_t& c2, const f32x4_t& c3) {
        const float inv = lsum[x] > 0.f ? 1.0f / lsum[x] : 0.f;
        const int qrow = q0 + 32 * x + frow;
        bf16_t* op = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + h * HD + 8 * hi + i * 32;
        const f32x4_t* cc[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
        for (int q8 = 0; q8 < 2; ++q8) {
            const f32x4_t& lo = *cc[2 * q8];
            const f32x4_t& hi4 = *cc[2 * q8 + 1];
            unsigned a0 = pack2(lo[0] * inv, lo[1] * inv), a1 = pack2(lo[2] * inv, lo[3] * inv);
            unsigned b0 = pack2(hi4[0] * inv, hi4[1] * inv), b1 = pack2(hi4[2] * inv, hi4[3] * inv);
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            u32x4_t w;
            w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
            if (qrow < p.L) *(u32x4_t*)(op + q8 * 16) = w;
        }
    };
#define STORE_HALF(X, I) store_half(X, I, pq_get4<32 * X + 16 * I>(), pq_get4<32 * X + 16 * I + 4>(), pq_get4<32 * X + 16 * I + 8>(), pq_get4<32 * X + 16 * I + 12>());
    STORE_HALF(0, 0) STORE_HALF(0, 1) STORE_HALF(1, 0) STORE_HALF(1, 1)
    STORE_HALF(2, 0) STORE_HALF(2, 1) STORE_HALF(3, 0) STORE_HALF(3, 1)
#undef STORE_HALF
}
