// Joint text-video flash attention with the reference's block-causal temporal mask, head_dim 64.
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask=[B,1,L,L] bool) of
// VarlenSelfAttentionWithT5Mask / VarlenSelfAttnSingle (flux_block.py:328-376, 568-606) and the
// mask construction of merge_input (modeling_pyramid_flux.py:318-350).  The L x L boolean mask
// is never materialised: the reference mask  (id_i == id_j) & (t_i >= t_j)  reduces, for the
// time-ordered sequence [text | history clips | current frame], to two key intervals per query
// row i:   keys in [a_lo_i, a_hi_i)  (text part, j < Lt)   U   keys in [Lt, b_hi_i)  (image part),
// which the host derives once per (unit, stage).  KV tiles beyond a q-tile's largest b_hi are
// never visited; waves whose 32 rows cannot see a tile skip its MFMAs.
//
// Structure: 128 q rows per workgroup (4 waves x 32 rows), KV tiles of 64 keys.  S^T = K.Q^T is
// computed "swapped" so one lane owns one q row (lane&31) and 16 keys per 32x32 tile: row max /
// sum are lane-local plus ONE cross-half exchange; P feeds the PV MFMA straight from registers
// because the contraction-slot -> key permutation of the C layout is mirrored in the V^T image
// (pf_v_transpose writes keys permuted within groups of 16).  K and V^T tiles arrive by LDS-DMA
// with source-side XOR swizzle, double-buffered, one barrier per KV tile.
#include <type_traits>
#include "common.h"
#include "pyflow_hip.h"

namespace {

constexpr int QB = 128, KB = 64, HD = 64;
constexpr int KTILE = KB * HD * 2;     // 8 KiB
constexpr int ABUF = 2 * KTILE;        // K + V^T
constexpr float NEG = -1.0e30f;
constexpr float DEFER = 6.0f;        // log2 units: P stays below 2^6 between rescales

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
// V fragment of the PV MFMA (32x32x16 A operand: lane -> feature lane % 32 of a 32-feature half, 8 keys) straight from a
// ROW-MAJOR V tile in LDS ([64 keys][64 features], 128-byte rows, 16-byte chunk c of key k stored at c ^ 4 ((k >> 1) & 1)):
// ds_read_b64_tr_b16 hands lane t of a 16-lane group column t of the [4 keys][16 features] block whose rows the group's
// lanes address (lane s: key s / 4, features 4 (s % 4) ..; profiles/r04_ds_read_tr_b16_probe.log), so two of them return the
// keys 16 g + 4 hi + {0..3} and 16 g + 8 + 4 hi + {0..3} of this lane's feature -- the contraction order of the P fragment
// (C-layout rows of S^T), which the V^T image of pf_v_transpose had to mirror with a key permutation.  `lane_off` =
// vrow_lane_offset(lane, i) for feature half i, `tile` = the V tile, g = 16-key slot.
PF_DEVICE unsigned vrow_lane_offset(int lane, int i) {
    const int kk = (lane & 15) >> 2, hi = lane >> 5, a = (lane >> 4) & 1, b = (lane & 3) >> 1;
    return (unsigned)((4 * hi + kk) * 128 + (((((i ^ (kk >> 1)) << 2) | (a << 1) | b)) << 4) + 8 * (lane & 1));
}
PF_DEVICE bf16x8_t vrow_fragment(const char* tile, unsigned lane_off, int g) {
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + lane_off + (16 * g) * 128));
    const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + lane_off + (16 * g + 8) * 128));
    const v8s_t v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

struct AArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* Vt; bf16_t* O;
    const bf16_t* V; int ldv, hs_v; long long sV;      // VROW kernels: V token-major like K (row stride ldv, head stride hs_v)
    int ldq, ldk, ldo;
    long long sQ, sK, sO, sVb, sVh;
    int Lp, L, H, B, Lt, nqt;
    int hs_qk;             // elements between consecutive heads in Q and K (64 = packed heads)
    int qt0;               // first 128-row query tile to compute (rows below are not needed by the caller)
    int prio;              // 0 no s_setprio, 1 around the MFMA groups (what pf_attention_bf16 sets), 2 around the softmax
    const int* a_lo; const int* a_hi; const int* b_hi;
    const int* tile_kv_end;
    float sc;   // softmax scale * log2(e)
    unsigned* dbg;   // lab builds only (cycle stamps of attn64_kernel<.., 2>); nullptr in the library
    int* wgflags;    // attn64 FAST / FIXUP pair: one int per wave (4 per workgroup), caller scratch
    // KV split of launches with too few workgroups for the chip (attention_w64.h SPLIT / COMBINE; caller scratch):
    int nsplit, kv_chunk;       // parts per query tile, 64-key tiles per part
    float* parts;               // [unit][wave][18 KiB]: unnormalised fp32 O and row sums of one part, lane-linear
    int* part_flags;            // [unit][wave]: the part produced a non-finite sum
};

template <bool PRE, int ILP, int OCC, bool VROW = false>
__global__ __launch_bounds__(256, OCC) void attn_kernel(const AArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * ABUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq_run = p.nqt - p.qt0;
    const int nwg = nq_run * p.H * p.B;
    int t = xcd_remap(blockIdx.x, nwg);
    const int bh = t / nq_run;
    const int qt = p.nqt - 1 - (t - bh * nq_run);     // heaviest (latest) q tiles first
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * QB;
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;

    // ---- this lane's query row ----
    const int qrow = q0 + wid * 32 + frow;
    const bool qvalid = qrow < p.L;
    const int qr = qvalid ? qrow : p.L - 1;
    const bf16_t* qp = p.Q + (long long)b * p.sQ + (long long)qr * p.ldq + h * p.hs_qk + hi * 8;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qp + ks * 16);
    int alo = 0, ahi = 0, bhi = 0;
    if (qvalid) {
        alo = p.a_lo[(long long)b * p.L + qrow];
        ahi = p.a_hi[(long long)b * p.L + qrow];
        bhi = p.b_hi[(long long)b * p.L + qrow];
    }
    int wmax = bhi, wmin = qvalid ? bhi : 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        wmax = max(wmax, __shfl_xor(wmax, o));
        wmin = min(wmin, __shfl_xor(wmin, o));
    }
    const int kv_end = p.tile_kv_end[b * p.nqt + qt];
    const int ntiles = (kv_end + KB - 1) / KB;

    // ---- DMA sources: wave owns pieces i = wid*2 + j (rows 8i..8i+7) of the K and V^T tiles ----
    const bf16_t* kbase = p.K + (long long)b * p.sK + h * p.hs_qk;
    const bf16_t* vbase = VROW ? p.V + (long long)b * p.sV + h * p.hs_v : p.Vt + (long long)b * p.sVb + (long long)h * p.sVh;
    int prow[2], pc[2], pcv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wid * 2 + j;
        prow[j] = 8 * i + (lane >> 3);
        pc[j] = ((lane & 7) ^ (((i & 1) << 2) + (lane >> 4))) * 8;
        pcv[j] = ((lane & 7) ^ (((prow[j] >> 1) & 1) << 2)) * 8;       // row-major V tile: chunk c of key k at c ^ 4 ((k >> 1) & 1)
    }
    const unsigned vtr0 = vrow_lane_offset(lane, 0), vtr1 = vrow_lane_offset(lane, 1);
    auto issue = [&](int jt, int buf) {
        const int j0 = jt * KB;
        char* base = smem + buf * ABUF + wid * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int key = j0 + prow[j];
            key = key < p.L ? key : p.L - 1;
            glds16(kbase + (long long)key * p.ldk + pc[j], base + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (VROW) {
                int key = j0 + prow[j];
                key = key < p.L ? key : p.L - 1;
                glds16(vbase + (long long)key * p.ldv + pcv[j], base + KTILE + j * 1024);
            } else {
                glds16(vbase + (long long)prow[j] * p.Lp + j0 + pc[j], base + KTILE + j * 1024);
            }
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = PRE ? 0.f : NEG, l = 0.f;
    bool fresh = true;                 // PRE: the row has not seen a key yet (its reference max is still unset)
    f32x16_t negm;                     // PRE: -m broadcast, the C operand of the first S^T MFMA (scores come out as s - m)
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    const int wmax_s = __builtin_amdgcn_readfirstlane(wmax), wmin_s = __builtin_amdgcn_readfirstlane(wmin);
    const float NINF = -__builtin_inff();

    if (ntiles > 0) issue(0, 0);
    for (int jt = 0; jt < ntiles; ++jt) {
        const int buf = jt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (jt + 1 < ntiles) issue(jt + 1, buf ^ 1);
        const int j0 = jt * KB;
        const bool has_text = j0 < p.Lt;
        if (!has_text && j0 >= wmax_s) continue;     // scalar: nothing visible for these 32 rows
        const char* sk = smem + buf * ABUF;
        const char* sv = sk + KTILE;
        // ---- all K fragments first, then the S^T MFMAs (the compiler counts lgkmcnt down as they arrive)
        bf16x8_t kf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + hi) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) kf[i][ks] = *(const bf16x8_t*)(sk + (i * 32 + frow) * 128 + ch);
        }
        f32x16_t s[2];
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
        if (PRE) {
#pragma unroll
            for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][0], qf[0], negm, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[i][r] = 0.f;
                s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][0], qf[0], s[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][ks], qf[ks], s[i], 0, 0, 0);
        if (p.prio == 1) __builtin_amdgcn_s_setprio(0);
        if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
        // ---- V^T fragments are requested now and land under the softmax arithmetic
        bf16x8_t vf[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (VROW) {
                vf[0][g] = vrow_fragment(sv, vtr0, g);
                vf[1][g] = vrow_fragment(sv, vtr1, g);
            } else {
                const int ch = ((2 * g + hi) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) vf[i][g] = *(const bf16x8_t*)(sv + (i * 32 + frow) * 128 + ch);
            }
        }
        if (has_text || (j0 + KB > wmin_s)) {        // scalar: tile straddles a visibility boundary of some row
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = j0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = key < p.Lt ? (key >= alo && key < ahi) : (key < bhi);
                    s[i][r] = ok ? s[i][r] : NINF;      // exp2(-inf) == 0: no select needed after the exponential
                }
        }
        float mt;
        if (ILP == 1) {
            mt = s[0][0];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[i][r]);
        } else {                      // ILP independent max chains instead of one 32-long dependent chain
            float mq[ILP];
#pragma unroll
            for (int c = 0; c < ILP; ++c) mq[c] = s[(c * (32 / ILP)) >> 4][(c * (32 / ILP)) & 15];
#pragma unroll
            for (int e = 0; e < 32 / ILP; ++e)
#pragma unroll
                for (int c = 0; c < ILP; ++c) {
                    const int idx = c * (32 / ILP) + e;
                    mq[c] = fmaxf(mq[c], s[idx >> 4][idx & 15]);
                }
            mt = mq[0];
#pragma unroll
            for (int c = 1; c < ILP; ++c) mt = fmaxf(mt, mq[c]);
        }
        {   // the row's other 32 keys live in lane ^ 32
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        float ps = 0.f;
        if (PRE) {
            // q was pre-multiplied by scale*log2(e) (pf_qk_norm_rope q_scale) and s already holds score - m:
            // the common path is exp2 / add / convert only.
            const bool seen = mt > NINF;
            if (__builtin_amdgcn_ballot_w64(mt > DEFER || (fresh && seen)) != 0) {
                const float delta = fresh ? (seen ? mt : 0.f) : fmaxf(mt, 0.f);
                const float alpha = fresh ? 1.f : __builtin_amdgcn_exp2f(-delta);
                fresh = fresh && !seen;
                m += delta;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o[i][r] *= alpha; s[i][r] -= delta; }
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m;
            }
            float pq[ILP];
#pragma unroll
            for (int c = 0; c < ILP; ++c) pq[c] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[i][r]);
                    s[i][r] = e;
                    pq[r % ILP] += e;
                }
#pragma unroll
            for (int c = 0; c < ILP; ++c) ps += pq[c];
        } else {
            // ---- deferred rescale: keep the running max while no row of the wave grew by more than 2^DEFER
            if (__builtin_amdgcn_ballot_w64((mt - m) * p.sc > DEFER) != 0) {
                const float mn = fmaxf(m, mt);
                const float alpha = __builtin_amdgcn_exp2f((m - mn) * p.sc);
                m = mn;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            const float msc = m * p.sc;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i][r], p.sc, -msc));
                    s[i][r] = e;
                    ps += e;
                }
        }
        l += ps;
        bf16x8_t pf[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[g][e] = (bf16_t)s[g >> 1][8 * (g & 1) + e];
        if (p.prio == 2) __builtin_amdgcn_s_setprio(0);
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i][g], pf[g], o[i], 0, 0, 0);
        if (p.prio == 1) __builtin_amdgcn_s_setprio(0);
    }
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (qvalid) {
        bf16_t* op = p.O + (long long)b * p.sO + (long long)qrow * p.ldo + h * HD + 4 * hi;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                u32x2_t w;
                w[0] = pack2(o[i][4 * q4] * inv, o[i][4 * q4 + 1] * inv);
                w[1] = pack2(o[i][4 * q4 + 2] * inv, o[i][4 * q4 + 3] * inv);
                *(u32x2_t*)(op + i * 32 + q4 * 8) = w;
            }
    }
}

#include "attention_w64.h"

// V [B, L, H*64] (row stride ldv) -> V^T [B, H, 64, Lp] with keys permuted inside each group of
// 16 (bits 2 and 3 of the key index swapped) so the PV A-operand is one ds_read_b128 per lane.
__global__ __launch_bounds__(256) void vtrans_kernel(const bf16_t* V, bf16_t* Vt, int ldv, long long sV,
                                                     long long sVb, long long sVh, int L, int Lp, int H, int hs) {
    __shared__ unsigned short tile[64][66];
    const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int j0 = kt * 64;
    // load 64 keys x 64 d : thread -> key = tid/4 (+0), 16 d values
    {
        const int key = tid >> 2, dq = (tid & 3) * 16;
        const int kk = j0 + key;
        unsigned short vals[16];
        if (kk < L) {
            const u32x4_t a = *(const u32x4_t*)(V + (long long)b * sV + (long long)kk * ldv + h * hs + dq);
            const u32x4_t c = *(const u32x4_t*)(V + (long long)b * sV + (long long)kk * ldv + h * hs + dq + 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vals[2 * e] = a[e] & 0xffff; vals[2 * e + 1] = a[e] >> 16;
                vals[8 + 2 * e] = c[e] & 0xffff; vals[8 + 2 * e + 1] = c[e] >> 16;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) vals[e] = 0;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[dq + e][key] = vals[e];
    }
    __syncthreads();
    // store: thread -> d = tid/4, 16 permuted key slots
    {
        const int d = tid >> 2, s0 = (tid & 3) * 16;
        unsigned short out[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int slot = e;                                  // position inside the group of 16
            const int key = (slot & 3) | ((slot & 4) << 1) | ((slot & 8) >> 1);   // swap bits 2,3
            out[e] = tile[d][s0 + key];
        }
        u32x4_t a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = out[2 * e] | ((unsigned)out[2 * e + 1] << 16);
            c[e] = out[8 + 2 * e] | ((unsigned)out[8 + 2 * e + 1] << 16);
        }
        bf16_t* dst = Vt + (long long)b * sVb + (long long)h * sVh + (long long)d * Lp + j0 + s0;
        *(u32x4_t*)dst = a;
        *(u32x4_t*)(dst + 8) = c;
    }
}

}  // namespace

int pf_set_err(const char* m);

// Scratch of the 64-rows-per-wave pair: one int per wave of its grid (4 per workgroup) + the KV-split region (launches
// with too few workgroups for the chip: attention_w64.h SPLIT / COMBINE): at most SPLIT_UNITS (query tile, part) units of
// 4 waves x (18 KiB of parked sums + one flag).
#ifndef PF_ATTN_SPLIT_NWG      // (lab builds sweep it: tools/attn_schedule_ab.py)
#define PF_ATTN_SPLIT_NWG 640
#endif
constexpr int SPLIT_UNITS = 2 * PF_ATTN_SPLIT_NWG;     // SPLIT_NWG workgroups x 2 parts, or 256 x 4
constexpr long long PART_BYTES = 18 * 1024;       // per wave
static long long w64_flag_bytes(int B, int H, int L) { return (((long long)((L + 255) / 256) * H * B * 4 * (long long)sizeof(int)) + 255) & ~255ll; }
extern "C" long long pf_attention_workspace_bytes(int B, int H, int L) {
    if (B <= 0 || H <= 0 || L <= 0) return 0;
    return w64_flag_bytes(B, H, L) + (long long)SPLIT_UNITS * 4 * (PART_BYTES + 16);
}
// parts per query tile for a launch of `nwg` 256-row workgroups over `ntiles` key tiles (1 = no split): launches of at most
// 1.25 rounds of the chip's 512 workgroup slots are cut along the keys into chunks of at least 8 tiles
static int w64_split(long long nwg, int ntiles) {
    if (nwg >= PF_ATTN_SPLIT_NWG || ntiles < 16) return 1;
    int s_ = nwg >= 256 ? 2 : 4;
    while (s_ > 1 && (ntiles + s_ - 1) / s_ < 8) s_ >>= 1;
    return s_;
}

// ... and an output that either does not touch Q at all or IS Q (the in-place form of the DiT: same address, leading
// dimension, batch stride and packed heads -- then every wave overwrites exactly the 64 Q rows it alone reads, and the
// fast pass leaves the rows of a flagged wave unwritten for the fix-up launch; attention_w64.h).  Any other overlap of
// the two ranges would let the fix-up launch re-read overwritten Q: those problems stay with the 128-row kernel.
static bool o_aliases_q_safely(const pf_attn_desc* d) {
    const int hs = d->head_stride_qk > 0 ? d->head_stride_qk : HD;
    const long long q_ext = ((long long)(d->B - 1) * d->strideQ + (long long)(d->L - 1) * d->ldq + (long long)(d->H - 1) * hs + HD) * 2;
    const long long o_ext = ((long long)(d->B - 1) * d->strideO + (long long)(d->L - 1) * d->ldo + (long long)d->H * HD) * 2;
    const uintptr_t q = (uintptr_t)d->Q, o = (uintptr_t)d->O;
    if (o + o_ext <= q || q + q_ext <= o) return true;                    // disjoint
    return o == q && d->ldo == d->ldq && d->strideO == d->strideQ && hs == HD;
}

static bool use_w64(const pf_attn_desc* d) {
    if (!d->q_prescaled || !d->workspace || d->L <= 0) return false;
    if (!o_aliases_q_safely(d)) return false;
    const int nqt = (d->L + QB - 1) / QB;
    const int qt0 = d->q_row_begin > 0 ? d->q_row_begin / QB : 0;
    const long long grid64 = (long long)((nqt + 1) / 2 - qt0 / 2) * d->H * d->B;
    // at least one round of the chip after the KV split (4 parts from 128 workgroups up)
    const int sp = w64_split(grid64, (d->L + KB - 1) / KB);
    const bool enough = grid64 * sp >= 512;
    // scratch THIS launch touches: the flag words, and the KV-split region only when the launch splits
    // (pf_attention_workspace_bytes is the bound that serves every q_row_begin of the shape)
    const long long need = w64_flag_bytes(d->B, d->H, d->L) + (sp > 1 ? (long long)SPLIT_UNITS * 4 * (PART_BYTES + 16) : 0);
    return d->workspace_bytes >= need && enough && (d->ldo % 8) == 0 &&
           (d->strideO % 8) == 0 && ((uintptr_t)d->O % 16) == 0 && ((uintptr_t)d->workspace % 4) == 0;
}

// 32 = the 128-row kernel, 64 = the 64-rows-per-wave pair, 64 + parts = the pair with its key range split into `parts`
extern "C" int pf_attention_which(const pf_attn_desc* d) {
    if (!d || !use_w64(d)) return 32;
    const int nqt = (d->L + QB - 1) / QB, qt0 = d->q_row_begin > 0 ? d->q_row_begin / QB : 0;
    const int sp = w64_split((long long)((nqt + 1) / 2 - qt0 / 2) * d->H * d->B, (d->L + KB - 1) / KB);
    return sp > 1 ? 64 + sp : 64;
}

extern "C" int pf_attention_bf16(const pf_attn_desc* d, hipStream_t stream) {
    if (!d || !d->Q || !d->K || (!d->Vt && !d->V) || !d->O) return pf_set_err("pf_attention_bf16: null operand");
    const bool vrow = d->V != nullptr;
    if (vrow && (d->ldv % 8)) return pf_set_err("pf_attention_bf16: ldv must be a multiple of 8");
    if (d->L <= 0 || d->B <= 0 || d->H <= 0) return pf_set_err("pf_attention_bf16: empty problem");
    if (d->Lp % 64 || d->Lp < d->L) return pf_set_err("pf_attention_bf16: Lp must be a multiple of 64 and >= L");
    if ((d->ldq % 8) || (d->ldk % 8) || (d->ldo % 4)) return pf_set_err("pf_attention_bf16: bad leading dims");
    AArgs a{};
    a.Q = (const bf16_t*)d->Q; a.K = (const bf16_t*)d->K; a.Vt = (const bf16_t*)d->Vt; a.O = (bf16_t*)d->O;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldo = d->ldo;
    a.sQ = d->strideQ; a.sK = d->strideK; a.sO = d->strideO; a.sVb = d->strideVt_b; a.sVh = d->strideVt_h;
    a.Lp = d->Lp; a.L = d->L; a.H = d->H; a.B = d->B; a.Lt = d->Lt;
    a.nqt = (d->L + QB - 1) / QB;
    a.a_lo = d->a_lo; a.a_hi = d->a_hi; a.b_hi = d->b_hi; a.tile_kv_end = d->tile_kv_end;
    a.sc = d->scale * 1.4426950408889634f;
    a.hs_qk = d->head_stride_qk > 0 ? d->head_stride_qk : HD;
    a.V = (const bf16_t*)d->V; a.ldv = d->ldv; a.sV = d->strideV; a.hs_v = a.hs_qk;
    if (a.hs_qk % 8) return pf_set_err("pf_attention_bf16: head_stride_qk must be a multiple of 8");
    a.prio = 1;              // s_setprio around the MFMA groups (measured best of {none, MFMA, softmax}: profiles/r01_attention_variants.log)
    a.qt0 = d->q_row_begin > 0 ? d->q_row_begin / QB : 0;
    if (a.qt0 >= a.nqt) return pf_set_err("pf_attention_bf16: q_row_begin beyond the sequence");
    // Large pre-scaled problems with caller scratch: the 64-rows-per-wave kernel as a fast pass + a fix-up pass
    // (attention_w64.h; +8...13 % over the kernel below from L = 3 008 up: profiles/r03_attention_w64_fast_fixup.log).
    if (use_w64(d)) {
        constexpr int SM64 = 2 * ABUF + 4 * 64 * HD * 2;
        const int grid64 = ((a.nqt + 1) / 2 - a.qt0 / 2) * a.H * a.B;
        a.wgflags = (int*)d->workspace;
        const int ntl = (a.L + KB - 1) / KB;
        const int sp = w64_split(grid64, ntl);
        if (sp > 1) {
            // too few workgroups for the chip: SPLIT (fast pass over `sp` key ranges per query tile, sums parked) -> COMBINE
            // (adds the parts, range check, store or flag) -> FIXUP (flagged workgroups, as always)
            char* reg = (char*)d->workspace + w64_flag_bytes(a.B, a.H, a.L);
            a.nsplit = sp;
            a.kv_chunk = (ntl + sp - 1) / sp;
            a.part_flags = (int*)reg;
            a.parts = (float*)(reg + (long long)SPLIT_UNITS * 4 * 16);
            if ((long long)grid64 * sp > SPLIT_UNITS) return pf_set_err("pf_attention_bf16: split units exceed the scratch");
            PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 33 | 128>), SM64);
            if (vrow) {
                PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 33 | 64, 4, true>), SM64);
                PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 4, 4, true>), SM64);
                hipLaunchKernelGGL((attn64_kernel<2, 33 | 64, 4, true>), dim3(grid64 * sp), dim3(256), SM64, stream, a);
                hipLaunchKernelGGL((attn64_kernel<2, 33 | 128>), dim3(grid64), dim3(256), SM64, stream, a);
                hipLaunchKernelGGL((attn64_kernel<2, 4, 4, true>), dim3(grid64), dim3(256), SM64, stream, a);
            } else {
                PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 33 | 64>), SM64);
                PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 4>), SM64);
                hipLaunchKernelGGL((attn64_kernel<2, 33 | 64>), dim3(grid64 * sp), dim3(256), SM64, stream, a);
                hipLaunchKernelGGL((attn64_kernel<2, 33 | 128>), dim3(grid64), dim3(256), SM64, stream, a);
                hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, stream, a);
            }
            hipError_t es = hipGetLastError();
            if (es != hipSuccess) return pf_set_err(hipGetErrorString(es));
            return 0;
        }
        // FAST | MMSUM (33): row sums on the matrix pipe, +1.5 % (L = 15 488) ... +3 % (L = 3 008) over the v_add_f32 sums
        // (mode 1) in the same-box A/B of profiles/r04_attention_rowsum_variants.log; the v_dot2c / v_pk_add forms lose 2-3 %
        if (vrow) {
            PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 33, 4, true>), SM64);
            PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 4, 4, true>), SM64);
            hipLaunchKernelGGL((attn64_kernel<2, 33, 4, true>), dim3(grid64), dim3(256), SM64, stream, a);
            hipLaunchKernelGGL((attn64_kernel<2, 4, 4, true>), dim3(grid64), dim3(256), SM64, stream, a);
        } else {
            PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 33>), SM64);
            PF_SET_MAX_LDS_ONCE((attn64_kernel<2, 4>), SM64);
            hipLaunchKernelGGL((attn64_kernel<2, 33>), dim3(grid64), dim3(256), SM64, stream, a);
            hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, stream, a);
        }
        hipError_t e64 = hipGetLastError();
        if (e64 != hipSuccess) return pf_set_err(hipGetErrorString(e64));
        return 0;
    }
    const int grid = (a.nqt - a.qt0) * a.H * a.B;
    // one max chain, three waves per SIMD: 2 / 4 chains and 2 waves per SIMD measured within 2 % (same log)
    if (vrow) {
        if (!d->q_prescaled) hipLaunchKernelGGL((attn_kernel<false, 1, 3, true>), dim3(grid), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn_kernel<true, 1, 3, true>), dim3(grid), dim3(256), 0, stream, a);
    } else if (!d->q_prescaled) hipLaunchKernelGGL((attn_kernel<false, 1, 3>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((attn_kernel<true, 1, 3>), dim3(grid), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pf_set_err(hipGetErrorString(e));
    return 0;
}

extern "C" int pf_v_transpose(const void* V, void* Vt, int ldv, long long strideV, long long strideVt_b,
                              long long strideVt_h, int B, int H, int L, int Lp, int head_stride, hipStream_t stream) {
    if (!V || !Vt) return pf_set_err("pf_v_transpose: null operand");
    if (head_stride <= 0) head_stride = 64;
    if (Lp % 64 || Lp < L || (ldv % 8) || (head_stride % 8)) return pf_set_err("pf_v_transpose: bad Lp/ldv/head_stride");
    hipLaunchKernelGGL(vtrans_kernel, dim3(Lp / 64, H, B), dim3(256), 0, stream, (const bf16_t*)V, (bf16_t*)Vt,
                       ldv, strideV, strideVt_b, strideVt_h, L, Lp, H, head_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return pf_set_err(hipGetErrorString(e));
    return 0;
}
