// lab/attn128_lab.h -- EXPERIMENT (round 4, not part of the library): the fast pass of the attention pair with ONE wave per
// SIMD -- 4 waves per workgroup, one workgroup per CU, 512 registers per lane, each wave owning FOUR 32-row query blocks
// (128 rows; 512 rows per workgroup) instead of two.  Why: in attn64_kernel two waves share a SIMD and each spends ~1 150 of
// its ~3 450 cycles per 64-key tile issuing MFMAs; the rest is per-TILE cost paid per wave (4 LDS-DMA pieces, 32 fragment
// reads, the barrier) that only the other wave's MFMAs can cover -- the pipe cannot exceed 2 x 33 %.  Here the same per-tile
// cost serves twice the MFMA work, the Q fragments live in registers (no per-tile re-read), and the exponentials of one pair
// of blocks run under the MFMAs of the other pair inside ONE instruction stream.
// Same arguments / numerics as attn64_kernel<2, FAST | MMSUM, 4, VROW> (no running maximum, row sums on the matrix pipe, V
// token-major through ds_read_b64_tr_b16).  Flags: one int per 64-row unit, p.wgflags[((bh * nq256 + qt256) * 4 + w64)].
#pragma once

template <bool VROW_ = true>
__global__ __launch_bounds__(256, 1) void attn128_fast_kernel(const AArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * ABUF: K | V tile ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq5 = (p.nqt + 3) / 4;                                 // 512-row query tiles
    const int nwg = nq5 * p.H * p.B;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int bh = t / nq5;
    const int qt = nq5 - 1 - (t - bh * nq5);                         // heaviest first
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * 512 + wid * 128;                             // this wave's first row
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
    // Q fragments of the four blocks: straight from global memory into registers, once
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
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[x][ks]));      // consumed before the first LDS-DMA is issued
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

    // ---- DMA sources (as attn64_kernel: wave owns pieces wid*2 + j of the K and the V tile)
    const char* const kbase = (const char*)(p.K + (long long)b * p.sK + h * p.hs_qk);
    const char* const vbase = (const char*)(p.V + (long long)b * p.sV + h * p.hs_v);
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
    auto issue_kv = [&](int jt, int buf, int j) {
        const int last = p.L - 1 - jt * KB;
        const int r = min(prow[j], last);
        glds16(kbase + (long long)jt * KB * ldk2 + (unsigned)(r * ldk2) + pc2[j], smem + buf * ABUF + (wid * 2 + j) * 1024);
        glds16(vbase + (long long)jt * KB * ldv2 + (unsigned)(r * ldv2) + pcv[j], smem + buf * ABUF + KTILE + (wid * 2 + j) * 1024);
    };
    const unsigned vtr0 = vrow_lane_offset(lane, 0), vtr1 = vrow_lane_offset(lane, 1);
    unsigned foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = (unsigned)(frow * 128 + (((2 * ks + hi) ^ swz) << 4));

    f32x16_t o[4][2];
    f32x4_t lacc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        lacc[x] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][i][r] = 0.f;
    }
    bf16x8_t ones_a;
    {
        const bf16_t one_or_zero = (((lane >> 4) ^ lane) & 1) == 0 ? (bf16_t)1.0f : (bf16_t)0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones_a[e] = one_or_zero;
    }
    const float NINF = -__builtin_inff();
    bf16x8_t kf[2][4], vf[2][4];
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
            vf[0][g] = vrow_fragment(sv, vtr0, g);
            vf[1][g] = vrow_fragment(sv, vtr1, g);
        }
    };
    auto qk = [&](int x, f32x16_t* s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][ks], qf[x][ks], s[i], 0, 0, 0);
    };
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
    auto exps = [&](const f32x16_t* s, bf16x8_t* pf) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[g][e] = (bf16_t)__builtin_amdgcn_exp2f(s[g >> 1][8 * (g & 1) + e]);
    };
    auto pv = [&](int x, const bf16x8_t* pf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 2; ++i) o[x][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i][g], pf[g], o[x][i], 0, 0, 0);
            lacc[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pf[g], lacc[x], 0, 0, 0);
        }
    };

    // ---- prologue
    if (ntiles > 0) { issue_kv(0, 0, 0); issue_kv(0, 0, 1); }
    if (ntiles > 1) {
        issue_kv(1, 1, 0); issue_kv(1, 1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (my_nt > 0) read_k(0);

    for (int jt = 0; jt < ntiles; ++jt) {
        const int buf = jt & 1, j0 = jt * KB;
        const bool masked = (j0 < p.Lt) || (j0 + KB > wmin_s);
        if (jt < my_nt) {
            // blocks 0, 1: scores; blocks 2, 3: scores while the exponentials of 0, 1 run; PV(0, 1) while those of 2, 3 run; PV(2, 3)
            // issue-order segments (sched_barrier between them, as in attn64_kernel): at most two score blocks and two P
            // blocks are alive at a time
#define SEG() __builtin_amdgcn_sched_barrier(0)
            f32x16_t sa[2], sb[2];
            bf16x8_t pa[4], pb[4];
            qk(0, sa);
            SEG();
            qk(1, sb);
            read_v(buf);
            if (masked) apply_mask(0, sa, j0);
            SEG();
            exps(sa, pa);                 // P(0) under QK(2) ...
            qk(2, sa);
            if (masked) apply_mask(1, sb, j0);
            SEG();
            exps(sb, pb);                 // P(1) under QK(3)
            qk(3, sb);
            SEG();
            pv(0, pa);                    // PV(0) with the exponentials of block 2 ...
            if (masked) apply_mask(2, sa, j0);
            exps(sa, pa);
            SEG();
            pv(1, pb);
            if (masked) apply_mask(3, sb, j0);
            exps(sb, pb);
            SEG();
            pv(2, pa);
            SEG();
            pv(3, pb);
            SEG();
#undef SEG
        }
        if (jt + 1 < ntiles) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (jt + 1 < my_nt) read_k(buf ^ 1);
            if (jt + 2 < ntiles) { issue_kv(jt + 2, buf, 0); issue_kv(jt + 2, buf, 1); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- flags per 64-row unit + epilogue (as attn64_kernel)
    const int nq2 = (p.nqt + 1) / 2;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        bool bad = false;
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) {
            const int x = 2 * u + x2;
            const float lx = (lane & 16) ? lacc[x][1] : lacc[x][0];
            const int qrow = q0 + 32 * x + frow;
            bad = bad || (qrow < p.L && !(lx > 1e-30f && lx < 1e30f));
        }
        const int any_bad = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1 : 0;
        const int qt256 = qt * 2 + (wid >> 1), w64 = (wid & 1) * 2 + u;
        if (lane == 0 && qt256 < nq2) p.wgflags[((long long)bh * nq2 + qt256) * 4 + w64] = any_bad;
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float lx = (lane & 16) ? lacc[x][1] : lacc[x][0];
        const float inv = lx > 0.f ? 1.0f / lx : 0.f;
        const int qrow = q0 + 32 * x + frow;
        const bool qvalid = qrow < p.L;
        bf16_t* op = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + h * HD + 8 * hi;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q8 = 0; q8 < 2; ++q8) {
                unsigned a0 = pack2(o[x][i][8 * q8 + 0] * inv, o[x][i][8 * q8 + 1] * inv);
                unsigned a1 = pack2(o[x][i][8 * q8 + 2] * inv, o[x][i][8 * q8 + 3] * inv);
                unsigned b0 = pack2(o[x][i][8 * q8 + 4] * inv, o[x][i][8 * q8 + 5] * inv);
                unsigned b1 = pack2(o[x][i][8 * q8 + 6] * inv, o[x][i][8 * q8 + 7] * inv);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4_t w;
                w[0] = r0[0]; w[1] = r1[0]; w[2] = r0[1]; w[3] = r1[1];
                if (qvalid) *(u32x4_t*)(op + i * 32 + q8 * 16) = w;
            }
    }
}


