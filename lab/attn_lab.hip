// lab/attn_lab.hip -- standalone attention laboratory (NOT part of libpyflow_hip.so; nothing here ships).
//
// One self-contained HIP program (no torch, no Python: starts in milliseconds on the GPU box) that
//   * builds the masked C3 sequences exactly as pyflow_hip/plan.py does (text 128 | history clips | current frame, CFG
//     batch 2, 40 / 96 valid prompt tokens),
//   * runs the SHIPPED kernel (csrc/attention.hip is #included, so the baseline is the library's code bit for bit),
//   * runs experimental variants against it (max-abs / rel-L2 vs the shipped output, sampled rows vs an fp32 host
//     reference), interleaved timing rounds with HIP events,
//   * runs ABLATIONS of the shipped structure (no softmax / no barrier+DMA / no LDS reads / no MFMA) and an
//     s_memtime-stamped build that reports where a wave's cycles go per KV tile.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I pyramid-flow_amd/csrc lab/attn_lab.hip -o lab/attn_lab
// Run:    lab/attn_lab [seq] [rounds]      seq in {u30s2, u15s2, u30s0, u1s2, u5s1}
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <string>
#include <vector>

#include "../pyramid-flow_amd/csrc/attention.hip"

namespace {
#include "attn128_lab.h"
#include "attn128_pipe.h"
}

int pf_set_err(const char* m) {
    fprintf(stderr, "pf error: %s\n", m);
    return -1;
}

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

namespace lab {

// ------------------------------------------------------------------------------------------------------------------
// Ablations of the shipped structure (PRE path, ILP 1).  ABL bits:
//   1  no softmax arithmetic (P = bf16(S) straight from the accumulators)
//   2  no barrier / vmcnt wait / DMA after the first tile (every tile computes on tile 0's bytes)
//   4  no LDS fragment reads after the first tile
//   8  no MFMAs (accumulators passed through an empty asm so the dependent code stays)
//   16 s_memtime stamps around the phases of a tile (sums per wave written to `dbg`)
enum { A_NOSM = 1, A_NOSYNC = 2, A_NOLDS = 4, A_NOMFMA = 8, A_STAMP = 16, A_NOMAX = 32 };   // A_NOMAX: no row maximum, no reference (m = 0)
constexpr int NPH = 8;

template <int ABL, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_abl_kernel(const AArgs p, unsigned* dbg) {
    __shared__ __attribute__((aligned(16))) char smem[2 * ABUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq_run = p.nqt - p.qt0;
    const int nwg = nq_run * p.H * p.B;
    int t = xcd_remap(blockIdx.x, nwg);
    const int bh = t / nq_run;
    const int qt = p.nqt - 1 - (t - bh * nq_run);
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qt * QB;
    const int frow = lane & 31, hi = lane >> 5, swz = (lane >> 1) & 7;
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
    const bf16_t* kbase = p.K + (long long)b * p.sK + h * p.hs_qk;
    const bf16_t* vbase = p.Vt + (long long)b * p.sVb + (long long)h * p.sVh;
    int prow[2], pc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wid * 2 + j;
        prow[j] = 8 * i + (lane >> 3);
        pc[j] = ((lane & 7) ^ (((i & 1) << 2) + (lane >> 4))) * 8;
    }
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
        for (int j = 0; j < 2; ++j) glds16(vbase + (long long)prow[j] * p.Lp + j0 + pc[j], base + KTILE + j * 1024);
    };
    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = 0.f, l = 0.f;
    bool fresh = true;
    f32x16_t negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    const int wmax_s = __builtin_amdgcn_readfirstlane(wmax), wmin_s = __builtin_amdgcn_readfirstlane(wmin);
    const float NINF = -__builtin_inff();

    unsigned ph[NPH];
#pragma unroll
    for (int i = 0; i < NPH; ++i) ph[i] = 0;
    unsigned tprev = 0;
    auto stamp = [&](int k) {
        if (ABL & A_STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
            ph[k] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    bf16x8_t kf[2][4], vf[2][4];
    if (ntiles > 0) issue(0, 0);
    if (ABL & A_STAMP) tprev = (unsigned)__builtin_amdgcn_s_memtime();
    for (int jt = 0; jt < ntiles; ++jt) {
        const int buf = (ABL & A_NOSYNC) ? 0 : (jt & 1);
        if (!(ABL & A_NOSYNC) || jt == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        stamp(0);
        if (!(ABL & A_NOSYNC))
            if (jt + 1 < ntiles) issue(jt + 1, buf ^ 1);
        stamp(1);
        const int j0 = jt * KB;
        const bool has_text = j0 < p.Lt;
        if (!has_text && j0 >= wmax_s) continue;
        const char* sk = smem + buf * ABUF;
        const char* sv = sk + KTILE;
        if (!(ABL & A_NOLDS) || jt == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ((2 * ks + hi) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) kf[i][ks] = *(const bf16x8_t*)(sk + (i * 32 + frow) * 128 + ch);
            }
        }
        if (ABL & A_STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(2);
        f32x16_t s[2];
        if (ABL & A_NOMFMA) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                s[i] = negm;
                asm volatile("" : "+v"(s[i]) : "v"(kf[i][0]), "v"(kf[i][1]), "v"(kf[i][2]), "v"(kf[i][3]));
            }
        } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][0], qf[0], (ABL & A_NOMAX) ? f32x16_t{0.f} : negm, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i][ks], qf[ks], s[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        stamp(3);
        if (!(ABL & A_NOLDS) || jt == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = ((2 * g + hi) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) vf[i][g] = *(const bf16x8_t*)(sv + (i * 32 + frow) * 128 + ch);
            }
        }
        bf16x8_t pf[4];
        if (ABL & A_NOSM) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[g][e] = (bf16_t)s[g >> 1][8 * (g & 1) + e];
            stamp(4);
            stamp(5);
        } else {
            if (has_text || (j0 + KB > wmin_s)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = j0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const bool ok = key < p.Lt ? (key >= alo && key < ahi) : (key < bhi);
                        s[i][r] = ok ? s[i][r] : NINF;
                    }
            }
            float mt = s[0][0];
            if (!(ABL & A_NOMAX)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[i][r]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            }
            const bool seen = mt > NINF;
            if (!(ABL & A_NOMAX) && __builtin_amdgcn_ballot_w64(mt > DEFER || (fresh && seen)) != 0) {
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
            stamp(4);
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[i][r]);
                    s[i][r] = e;
                    ps += e;
                }
            l += ps;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[g][e] = (bf16_t)s[g >> 1][8 * (g & 1) + e];
            stamp(5);
        }
        if (ABL & A_NOMFMA) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                asm volatile("" : "+v"(o[i]) : "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(vf[i][0]), "v"(vf[i][1]),
                             "v"(vf[i][2]), "v"(vf[i][3]));
        } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i][g], pf[g], o[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        stamp(6);
        if (ABL & A_STAMP) ph[7] += 1;
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
    if ((ABL & A_STAMP) && lane == 0) {
        unsigned* d = dbg + ((size_t)blockIdx.x * 4 + wid) * NPH;
#pragma unroll
        for (int i = 0; i < NPH; ++i) d[i] = ph[i];
    }
}

}  // namespace lab

// ------------------------------------------------------------------------------------------------------------------
// host side
struct Plan {
    int B = 2, Lt = 128, L = 0, Lp = 0, nqt = 0;
    std::vector<int> a_lo, a_hi, b_hi, tile_end;
    double useful_pairs = 0;
};

static Plan make_plan(const std::vector<std::array<int, 3>>& clips) {
    Plan pl;
    const int valid[2] = {40, 96};
    std::vector<int> frame_t;
    int start = 0;
    for (auto& c : clips) {
        const int n = (c[1] / 2) * (c[2] / 2);
        for (int t = 0; t < c[0]; ++t)
            for (int i = 0; i < n; ++i) frame_t.push_back(start + t);
        start += c[0];
    }
    const int L_img = (int)frame_t.size();
    pl.L = pl.Lt + L_img;
    pl.Lp = (pl.L + 63) / 64 * 64;
    const int nf = start;
    std::vector<int> counts(nf, 0), frame_end(nf, 0);
    for (int f : frame_t) counts[f]++;
    int acc = pl.Lt;
    for (int f = 0; f < nf; ++f) { acc += counts[f]; frame_end[f] = acc; }
    pl.a_lo.assign((size_t)pl.B * pl.L, 0);
    pl.a_hi.assign((size_t)pl.B * pl.L, 0);
    pl.b_hi.assign((size_t)pl.B * pl.L, 0);
    for (int b = 0; b < pl.B; ++b) {
        const int v = valid[b];
        for (int i = 0; i < pl.L; ++i) {
            int lo, hi_, bh;
            if (i < v) { lo = 0; hi_ = v; bh = nf ? frame_end[0] : pl.Lt; }
            else if (i < pl.Lt) { lo = v; hi_ = pl.Lt; bh = pl.Lt; }
            else { lo = 0; hi_ = v; bh = frame_end[frame_t[i - pl.Lt]]; }
            pl.a_lo[(size_t)b * pl.L + i] = lo;
            pl.a_hi[(size_t)b * pl.L + i] = hi_;
            pl.b_hi[(size_t)b * pl.L + i] = bh;
            pl.useful_pairs += (hi_ - lo) + (bh - pl.Lt);
        }
    }
    pl.nqt = (pl.L + 127) / 128;
    pl.tile_end.assign((size_t)pl.B * pl.nqt, 0);
    for (int b = 0; b < pl.B; ++b)
        for (int qt = 0; qt < pl.nqt; ++qt) {
            int mx = pl.Lt;
            for (int i = qt * 128; i < std::min((qt + 1) * 128, pl.L); ++i) mx = std::max(mx, pl.b_hi[(size_t)b * pl.L + i]);
            pl.tile_end[(size_t)b * pl.nqt + qt] = mx;
        }
    return pl;
}

static inline unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short b) {
    unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Rng {
    unsigned long long s;
    explicit Rng(unsigned long long seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    inline unsigned long long next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    inline float uni() { return ((next() >> 40) + 0.5f) * (1.0f / 16777216.0f); }
    inline float normal() { return sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni()); }
};

int main(int argc, char** argv) {
    const std::string seq = argc > 1 ? argv[1] : "u30s2";
    const int rounds = argc > 2 ? atoi(argv[2]) : 4;
    std::vector<std::array<int, 3>> clips;
    if (seq == "u30s2") clips = {{28, 24, 40}, {1, 48, 80}, {1, 96, 160}, {1, 96, 160}};
    else if (seq == "u15s2") clips = {{13, 24, 40}, {1, 48, 80}, {1, 96, 160}, {1, 96, 160}};
    else if (seq == "u30s0") clips = {{29, 24, 40}, {1, 24, 40}, {1, 24, 40}};
    else if (seq == "u1s2") clips = {{1, 96, 160}, {1, 96, 160}};
    else if (seq == "u5s1") clips = {{4, 24, 40}, {1, 48, 80}, {1, 48, 80}};
    else { fprintf(stderr, "unknown sequence %s\n", seq.c_str()); return 2; }
    Plan pl = make_plan(clips);
    const int B = pl.B, H = 30, d = 1920, L = pl.L, Lp = pl.Lp, ld = 3 * d;
    printf("seq %s: L=%d Lp=%d nqt=%d useful pairs %.4g  (%.1f%% of dense)\n", seq.c_str(), L, Lp, pl.nqt, pl.useful_pairs,
           100.0 * pl.useful_pairs / ((double)B * L * L));
    const double flops = 4.0 * pl.useful_pairs * 64 * H;

    // ---- data: K | V | Q columns of one fused projection buffer, q pre-scaled by 0.125*log2(e) like pf_qk_norm_rope leaves it
    std::vector<unsigned short> hq((size_t)B * L * ld);
    {
        Rng r(3);
        const float qs = 0.125f * 1.4426950408889634f;
        for (size_t i = 0; i < hq.size(); ++i) {
            const int col = (int)(i % ld);
            float v = r.normal();
            if (col >= 2 * d) v *= qs;
            hq[i] = f2bf(v);
        }
    }
    unsigned short *dqkv, *dvt, *dout, *dref;
    int *d_alo, *d_ahi, *d_bhi, *d_te;
    unsigned* d_dbg;
    CK(hipMalloc(&dqkv, hq.size() * 2));
    CK(hipMalloc(&dvt, (size_t)B * H * 64 * Lp * 2));
    CK(hipMalloc(&dout, (size_t)B * L * d * 2));
    CK(hipMalloc(&dref, (size_t)B * L * d * 2));
    CK(hipMalloc(&d_alo, pl.a_lo.size() * 4));
    CK(hipMalloc(&d_ahi, pl.a_hi.size() * 4));
    CK(hipMalloc(&d_bhi, pl.b_hi.size() * 4));
    CK(hipMalloc(&d_te, pl.tile_end.size() * 4));
    const size_t nwg_max = (size_t)pl.nqt * H * B;
    CK(hipMalloc(&d_dbg, nwg_max * 4 * lab::NPH * 4));
    int* d_flags;
    CK(hipMalloc(&d_flags, nwg_max * 4 * 4));
    CK(hipMemset(d_flags, 0xff, nwg_max * 4 * 4));
    CK(hipMemcpy(dqkv, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dvt, 0, (size_t)B * H * 64 * Lp * 2));
    CK(hipMemcpy(d_alo, pl.a_lo.data(), pl.a_lo.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ahi, pl.a_hi.data(), pl.a_hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bhi, pl.b_hi.data(), pl.b_hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_te, pl.tile_end.data(), pl.tile_end.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    if (pf_v_transpose(dqkv + d, dvt, ld, (long long)L * ld, (long long)H * 64 * Lp, (long long)64 * Lp, B, H, L, Lp, 64, st)) return 1;

    pf_attn_desc desc{};
    desc.Q = dqkv + 2 * d; desc.K = dqkv; desc.Vt = dvt; desc.O = dref;
    desc.ldq = ld; desc.ldk = ld; desc.ldo = d;
    desc.strideQ = (long long)L * ld; desc.strideK = (long long)L * ld; desc.strideO = (long long)L * d;
    desc.strideVt_b = (long long)H * 64 * Lp; desc.strideVt_h = (long long)64 * Lp;
    desc.Lp = Lp; desc.L = L; desc.H = H; desc.B = B; desc.Lt = pl.Lt;
    desc.a_lo = d_alo; desc.a_hi = d_ahi; desc.b_hi = d_bhi; desc.tile_kv_end = d_te;
    desc.scale = 0.125f; desc.q_prescaled = 1;

    AArgs a{};
    a.Q = (const bf16_t*)desc.Q; a.K = (const bf16_t*)desc.K; a.Vt = (const bf16_t*)desc.Vt; a.O = (bf16_t*)dout;
    a.ldq = ld; a.ldk = ld; a.ldo = d;
    a.sQ = desc.strideQ; a.sK = desc.strideK; a.sO = desc.strideO; a.sVb = desc.strideVt_b; a.sVh = desc.strideVt_h;
    a.Lp = Lp; a.L = L; a.H = H; a.B = B; a.Lt = pl.Lt; a.nqt = pl.nqt;
    a.a_lo = d_alo; a.a_hi = d_ahi; a.b_hi = d_bhi; a.tile_kv_end = d_te;
    a.sc = 0.125f * 1.4426950408889634f; a.hs_qk = 64; a.prio = 1; a.qt0 = 0;
    a.V = (const bf16_t*)(dqkv + d); a.ldv = ld; a.sV = (long long)L * ld; a.hs_v = 64;       // token-major V (VROW kernels)
    const int grid = pl.nqt * H * B;

    struct Var { const char* name; std::function<void()> run; bool check; };
    std::vector<Var> vars;
    vars.push_back({"shipped attn_kernel<true,1,3>", [&] { pf_attention_bf16(&desc, st); }, false});
#define ABLV(name, abl, occ, chk) \
    vars.push_back({name, [&] { hipLaunchKernelGGL((lab::attn_abl_kernel<abl, occ>), dim3(grid), dim3(256), 0, st, a, d_dbg); }, chk})
    ABLV("copy of shipped (occ 3)", 0, 3, true);
    ABLV("copy of shipped (occ 2)", 0, 2, true);
    ABLV("32-row kernel, no row max / m = 0 (occ 3)", lab::A_NOMAX, 3, true);
    ABLV("32-row kernel, no row max / m = 0 (occ 4)", lab::A_NOMAX, 4, true);
    ABLV("abl: no softmax", lab::A_NOSM, 3, false);
    ABLV("abl: no barrier/DMA", lab::A_NOSYNC, 3, false);
    ABLV("abl: no LDS reads", lab::A_NOLDS, 3, false);
    ABLV("abl: no barrier/DMA, no LDS reads", lab::A_NOSYNC | lab::A_NOLDS, 3, false);
    ABLV("abl: no MFMA", lab::A_NOMFMA, 3, false);
    ABLV("abl: no softmax, no MFMA (LDS+DMA+barrier only)", lab::A_NOSM | lab::A_NOMFMA, 3, false);
    ABLV("abl: MFMA only (no softmax, sync, LDS)", lab::A_NOSM | lab::A_NOSYNC | lab::A_NOLDS, 3, false);
    const int grid64 = ((pl.nqt + 1) / 2) * H * B;
    constexpr int SM64 = 2 * ABUF + 4 * 64 * HD * 2, SM128 = 2 * ABUF + 8 * 64 * HD * 2;
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)attn64_kernel<2, 33>, hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)(attn64_kernel<2, 33, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SM64));
    CK(hipFuncSetAttribute((const void*)(attn64_kernel<2, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SM128));
    CK(hipFuncSetAttribute((const void*)(attn64_kernel<2, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, SM128));
    const int grid128 = ((pl.nqt + 3) / 4) * H * B;
    vars.push_back({"attn64_kernel<2> (64 rows / wave, 256 / WG)", [&] { hipLaunchKernelGGL((attn64_kernel<2>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    a.dbg = d_dbg;
    a.wgflags = d_flags;
    vars.push_back({"attn64 FAST (m = 0, no row max) + FIXUP launch", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 1>), dim3(grid64), dim3(256), SM64, st, a);
        hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    static pf_attn_desc desc2;
    desc2 = desc;
    desc2.O = dout;
    desc2.workspace = d_flags;
    desc2.workspace_bytes = (long long)nwg_max * 4 * 4;
    printf("pf_attention_which(desc + scratch) = %d\n", pf_attention_which(&desc2));
    vars.push_back({"pf_attention_bf16 with scratch (library dispatch)", [&] { pf_attention_bf16(&desc2, st); }, true});
    vars.push_back({"attn64 8 waves / 512 rows: FAST + FIXUP", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 1, 8>), dim3(grid128), dim3(512), SM128, st, a);
        hipLaunchKernelGGL((attn64_kernel<2, 4, 8>), dim3(grid128), dim3(512), SM128, st, a); }, true});
    vars.push_back({"attn64 FAST alone", [&] { hipLaunchKernelGGL((attn64_kernel<2, 1>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    // written at the end of round 3, first measured in round 4: row sums by v_dot2c_f32_bf16 on the packed P (MODE bit 3)
    vars.push_back({"attn64 FAST, v_pk_add_f32 row sums + FIXUP launch", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 17>), dim3(grid64), dim3(256), SM64, st, a);
        hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    vars.push_back({"attn64 FAST, dot2 row sums + FIXUP launch", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 9>), dim3(grid64), dim3(256), SM64, st, a);
        hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    // round 4: row sums from the matrix pipe (one 16x16x32 MFMA of P against a constant 0/1 operand per 16-key slot)
    vars.push_back({"attn64 FAST, MFMA row sums + FIXUP launch", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 33>), dim3(grid64), dim3(256), SM64, st, a);
        hipLaunchKernelGGL((attn64_kernel<2, 4>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    // round 4: one wave per SIMD, four 32-row blocks per wave (512 rows per workgroup), Q in registers, V token-major
    const int grid512 = ((pl.nqt + 3) / 4) * H * B;
    vars.push_back({"attn64 FAST | MMSUM | VROW alone (shipped fast pass)", [&] {
        hipLaunchKernelGGL((attn64_kernel<2, 33, 4, true>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    vars.push_back({"attn128 FAST (1 wave / SIMD, 4 blocks / wave) alone", [&] {
        hipLaunchKernelGGL((attn128_fast_kernel<true>), dim3(grid512), dim3(256), 2 * ABUF, st, a); }, true});
    vars.push_back({"attn128 pipe (hand-placed, 1 wave / SIMD) alone", [&] {
        hipLaunchKernelGGL((attn128_pipe_kernel<0, 0>), dim3(grid512), dim3(256), 3 * ABUF, st, a); }, true});
    vars.push_back({"attn128 pipe, MFMA and quarter as separate statements, alone", [&] {
        hipLaunchKernelGGL((attn128_pipe_kernel<0, 5>), dim3(grid512), dim3(256), 3 * ABUF, st, a); }, true});
    vars.push_back({"attn64 stamped", [&] { hipLaunchKernelGGL((attn64_kernel<2, 2>), dim3(grid64), dim3(256), SM64, st, a); }, true});
    vars.push_back({"attn64 FAST stamped", [&] { hipLaunchKernelGGL((attn64_kernel<2, 3>), dim3(grid64), dim3(256), SM64, st, a); }, false});
    ABLV("stamped (occ 3)", lab::A_STAMP, 3, true);

    // reference output of the shipped kernel
    pf_attention_bf16(&desc, st);
    CK(hipStreamSynchronize(st));
    std::vector<unsigned short> href((size_t)B * L * d), hout((size_t)B * L * d);
    CK(hipMemcpy(href.data(), dref, href.size() * 2, hipMemcpyDeviceToHost));

    // sampled rows vs fp64 on the host: reference values computed once, every checked variant is compared with them
    const std::vector<int> srows = {0, 39, 40, 127, 128, 367, 368, L / 2, L - 1};
    std::vector<double> sref;       // [b][row][h in 0,7,14,21,28][64]
    for (int b = 0; b < B; ++b)
        for (int row : srows)
            for (int h = 0; h < H; h += 7) {
                const size_t ri = (size_t)b * L + row;
                const int lo = pl.a_lo[ri], hi_ = pl.a_hi[ri], bh = pl.b_hi[ri];
                std::vector<double> sc;
                std::vector<int> keys;
                double mx = -1e300;
                for (int k = 0; k < L; ++k) {
                    const bool ok = k < pl.Lt ? (k >= lo && k < hi_) : (k < bh);
                    if (!ok) continue;
                    double s_ = 0;
                    for (int e = 0; e < 64; ++e)
                        s_ += (double)bf2f(hq[ri * ld + 2 * d + h * 64 + e]) * bf2f(hq[((size_t)b * L + k) * ld + h * 64 + e]);
                    s_ *= 0.6931471805599453;
                    sc.push_back(s_); keys.push_back(k);
                    mx = std::max(mx, s_);
                }
                double sum = 0;
                for (double& s_ : sc) { s_ = exp(s_ - mx); sum += s_; }
                for (int e = 0; e < 64; ++e) {
                    double acc = 0;
                    for (size_t i = 0; i < sc.size(); ++i) acc += sc[i] * bf2f(hq[((size_t)b * L + keys[i]) * ld + d + h * 64 + e]);
                    sref.push_back(acc / sum);
                }
            }
    auto vs_fp64 = [&](const std::vector<unsigned short>& out) {
        double num = 0, den = 0;
        size_t k = 0;
        for (int b = 0; b < B; ++b)
            for (int row : srows)
                for (int h = 0; h < H; h += 7)
                    for (int e = 0; e < 64; ++e) {
                        const double got = bf2f(out[((size_t)b * L + row) * d + h * 64 + e]), want = sref[k++];
                        num += (got - want) * (got - want);
                        den += want * want;
                    }
        return sqrt(num / den);
    };
    printf("shipped kernel, sampled rows vs fp64 host reference: rel-L2 %.3e\n", vs_fp64(href));

    std::vector<std::vector<float>> ms(vars.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < vars.size(); ++v) {
            vars[v].run();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 5; ++i) vars[v].run();
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms[v].push_back(t / 5);
        }
    for (size_t v = 0; v < vars.size(); ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
        char chk[96] = "";
        if (vars[v].check) {
            CK(hipMemset(dout, 0xff, hout.size() * 2));
            vars[v].run();
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
            double num = 0, den = 0, mxd = 0;
            for (size_t i = 0; i < hout.size(); ++i) {
                const double x = bf2f(hout[i]), y = bf2f(href[i]);
                num += (x - y) * (x - y); den += y * y; mxd = std::max(mxd, fabs(x - y));
            }
            snprintf(chk, sizeof chk, "  vs shipped: rel-L2 %.2e max-abs %.2e | vs fp64 rows: %.2e", sqrt(num / den), mxd, vs_fp64(hout));
        }
        printf("%-52s med %.3f ms  min %.3f ms  %6.0f TF useful (med)%s\n", vars[v].name, med, mn, flops / med / 1e9, chk);
    }
    for (int v64 = 2; v64 <= 3; ++v64) {
        CK(hipMemset(d_dbg, 0, nwg_max * 4 * lab::NPH * 4));
        if (v64 == 2) hipLaunchKernelGGL((attn64_kernel<2, 2>), dim3(grid64), dim3(256), SM64, st, a);
        else hipLaunchKernelGGL((attn64_kernel<2, 3>), dim3(grid64), dim3(256), SM64, st, a);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> hd((size_t)grid64 * 4 * 8);
        CK(hipMemcpy(hd.data(), d_dbg, hd.size() * 4, hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (size_t w = 0; w < (size_t)grid64 * 4; ++w)
            for (int i = 0; i < 8; ++i) sum[i] += hd[w * 8 + i];
        const char* nm[7] = {"S0: QK^T(A) (+ loop top)", "S1: QK^T(B) + [max(A) | exp(A) 0-2] + V reads", "S2: PV(A) + [exp(A) | exp(A) 3, exp(B) 0-1]", "tail: mask/max/check(B)",
                             "wait vmcnt/lgkmcnt + barrier", "K / Q fragment reads issued", "S3: exp(B) + PV(B) + DMA"};
        double tot = 0;
        for (int i = 0; i < 7; ++i) tot += sum[i];
        printf("attn64 stamped%s: cycles per processed 64-key tile and wave (2 x 32 rows; the matrix pipe needs 1024):\n", v64 == 3 ? " FAST" : "");
        for (int i = 0; i < 7; ++i) printf("   %-44s %8.1f  (%4.1f%%)\n", nm[i], sum[i] / sum[7], 100 * sum[i] / tot);
        printf("   %-44s %8.1f\n", "total", tot / sum[7]);
    }
    for (int sv = 0; sv < 7; ++sv) {
        CK(hipMemset(d_dbg, 0, nwg_max * 4 * lab::NPH * 4));
        const char* nm[7] = {"loop-stamped", "step-stamped", "step-stamped, NO V fragment reads", "step-stamped, NO barrier", "step-stamped, NO further DMA", "step-stamped, NO K fragment reads", "step-stamped, separate statements"};
        switch (sv) {
            case 0: hipLaunchKernelGGL((attn128_pipe_kernel<1, 0>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            case 1: hipLaunchKernelGGL((attn128_pipe_kernel<2, 0>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            case 2: hipLaunchKernelGGL((attn128_pipe_kernel<2, 1>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            case 3: hipLaunchKernelGGL((attn128_pipe_kernel<2, 2>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            case 4: hipLaunchKernelGGL((attn128_pipe_kernel<2, 3>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            case 5: hipLaunchKernelGGL((attn128_pipe_kernel<2, 4>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
            default: hipLaunchKernelGGL((attn128_pipe_kernel<2, 5>), dim3(grid512), dim3(256), 3 * ABUF, st, a); break;
        }
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> hd((size_t)grid512 * 4 * 8);
        CK(hipMemcpy(hd.data(), d_dbg, hd.size() * 4, hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (size_t w = 0; w < (size_t)grid512 * 4; ++w)
            for (int i = 0; i < 8; ++i) sum[i] += hd[w * 8 + i];
        printf("attn128 pipe %-36s cycles per 64-key tile and wave (4 x 32 rows; the matrix pipe needs 2304): loop %.1f", nm[sv], sum[6] / sum[7]);
        if (sv >= 1) printf("   steps A %.1f  B %.1f  C %.1f  D %.1f", sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7], sum[3] / sum[7]);
        printf("\n");
    }
    CK(hipMemset(d_dbg, 0, nwg_max * 4 * lab::NPH * 4));
    hipLaunchKernelGGL((lab::attn_abl_kernel<lab::A_STAMP, 3>), dim3(grid), dim3(256), 0, st, a, d_dbg);
    CK(hipStreamSynchronize(st));
    // phase stamps of the last (stamped) variant
    {
        std::vector<unsigned> hd(nwg_max * 4 * lab::NPH);
        CK(hipMemcpy(hd.data(), d_dbg, hd.size() * 4, hipMemcpyDeviceToHost));
        double sum[lab::NPH] = {0};
        for (size_t w = 0; w < nwg_max * 4; ++w)
            for (int i = 0; i < lab::NPH; ++i) sum[i] += hd[w * lab::NPH + i];
        const double nt = sum[7];
        const char* nm[7] = {"wait vmcnt(0)+barrier", "DMA issue", "K fragment reads (lgkmcnt 0)", "QK^T MFMAs issued",
                             "V reads + mask + max + rescale check", "exp / sum / convert", "PV MFMAs issued"};
        double tot = 0;
        for (int i = 0; i < 7; ++i) tot += sum[i];
        printf("stamped build: cycles per processed KV tile and wave (s_memtime, %.0f tile-waves; skipped tiles' barrier time is in phase 0/1):\n", nt);
        for (int i = 0; i < 7; ++i) printf("   %-40s %8.1f  (%4.1f%%)\n", nm[i], sum[i] / nt, 100 * sum[i] / tot);
        printf("   %-40s %8.1f   (MFMA pipe needs 512 of them)\n", "total", tot / nt);
    }
    return 0;
}
