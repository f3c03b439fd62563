// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Serves the DiT projections (K7/K11/K13/K14 of SURVEY section 2.2: QKV, out-proj, MLP, fused
// QKV+MLP of the single-stream blocks; reference call sites flux_block.py:756-758, 816-835,
// 868-872, 914-942) and -- in implicit-GEMM mode -- the CausalConv3d of the VAE decoder
// (video_vae/modeling_causal_conv.py:116-146).
//
// Structure: 128x128 block tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16 tiles.  Operand tiles go HBM -> LDS by direct LDS-DMA
// (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip); the LDS image is lane-linear, so the
// bank-conflict-free XOR swizzle is applied on the per-lane SOURCE address and again on the
// ds_read_b128 address (same involution).  Double-buffered, ONE barrier per K-step: the loads
// of tile k+1 are issued right after the barrier that publishes tile k and fly under its MFMAs.
// Round 5: launches of at most one workgroup per CU (the skinny problems this kernel still serves: the
// text rows of short sequences, a sequence-parallel rank's small shapes, the prompt encoders) run a
// THREE-stage ring instead (NSTAGE = 3, 96 KiB): tile k+2 is requested behind the barrier of tile k,
// so a K-step costs half a memory latency instead of a whole one (one K-tile is 512 MFMA cycles, a
// DMA round trip 1-2 k cycles: such launches are chains of latencies, not of MFMAs).  Larger grids keep
// two stages and two workgroups per CU, which hide the same latency across workgroups.
// Epilogue: accumulators are staged through LDS as fp32, then every thread streams whole
// 16-byte bf16 pieces (bias / GELU-tanh / gate*x+residual applied in fp32) -> coalesced stores.
#include "common.h"
#include "pyflow_hip.h"
#include "gemm_args.h"

using namespace pfgemm;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;        // A + W
constexpr int SMEM_BYTES = 2 * BUF_BYTES;        // double buffered = 64 KiB (= fp32 128x128 epilogue stage)
constexpr int SMEM_BYTES3 = 3 * BUF_BYTES;       // three stages = 96 KiB (one workgroup per CU)

template <bool CONV, int NSTAGE>
__global__ __launch_bounds__(256, NSTAGE == 2 ? 2 : 1) void gemm_kernel(const Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int per_batch = tiles_n * tiles_m;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int nwg = per_batch * p.batch * ksplit;
    int t = xcd_remap(blockIdx.x, nwg);
    const int ks_id = t / (per_batch * p.batch);        // which part of the K range this workgroup sums (split-K)
    t -= ks_id * per_batch * p.batch;
    const int b = t / per_batch;
    t -= b * per_batch;
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* A = p.A + (long long)b * p.sA;
    // ---- per-lane DMA source pointers: wave `wid` owns 1-KiB pieces i = wid*4 + j of each tile,
    //      piece i = rows 8i..8i+7; lane -> row 8i + lane/8, LDS chunk c' = lane%8 holds source
    //      chunk c = c' ^ ((row>>1)&7).
    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = wid * 4 + j;
        const int r = 8 * i + (lane >> 3);
        const int c = (lane & 7) ^ (((i & 1) << 2) + (lane >> 4));
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.cg.H * p.cg.W;
            const int tt = m / hw, rem = m - tt * hw;
            const int hh = rem / p.cg.W, ww = rem - hh * p.cg.W;
            asrc[j] = A + p.cg.base_off + (((long long)tt * p.cg.st * p.cg.Hp + hh * p.cg.sh) * p.cg.Wp + ww * p.cg.sw) * p.cg.Cin + c * 8;
        } else {
            asrc[j] = A + (long long)m * p.lda + c * 8;
        }
        wsrc[j] = p.W + (long long)(n0 + r) * p.ldw + c * 8;
    }
    const int nk = p.K / BK;

    auto issue = [&](int kt, int buf) {
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
        char* base = smem + buf * BUF_BYTES + wid * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(asrc[j] + aoff, base + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(wsrc[j] + (long long)kt * BK, base + TILE_BYTES + j * 1024);
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = wid >> 1, wn = wid & 1;
    // fragment read addressing: row = tile_row0 + (lane&31); chunk = 2*ks + (lane>>5), swizzled
    const int frow = lane & 31, fhi = lane >> 5, fswz = (lane >> 1) & 7;
    const int a_row_off = (wm * 64 + frow) * 128;
    const int w_row_off = (wn * 64 + frow) * 128;

    const int kt_begin = (int)((long long)nk * ks_id / ksplit), kt_end = (int)((long long)nk * (ks_id + 1) / ksplit);
    issue(kt_begin, 0);
    if (NSTAGE == 3 && kt_begin + 1 < kt_end) issue(kt_begin + 1, 1);
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (NSTAGE == 3) {
            // tile kt has landed (this wave's 8 pieces of tile kt + 1 may still be in flight); behind the barrier every wave is
            // done with tile kt - 1, whose buffer takes tile kt + 2
            if (kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < kt_end) issue(kt + 2, buf >= 1 ? buf - 1 : 2);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < kt_end) issue(kt + 1, buf ^ 1);
        }
        const char* sa = smem + buf * BUF_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ((2 * ks + fhi) ^ fswz) << 4;
            bf16x8_t af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *(const bf16x8_t*)(sa + a_row_off + i * 32 * 128 + ch);
                wf[i] = *(const bf16x8_t*)(sw + w_row_off + i * 32 * 128 + ch);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
        buf = NSTAGE == 3 ? (buf == 2 ? 0 : buf + 1) : buf ^ 1;
    }
    // ---- epilogue: stage fp32 accumulators in LDS (128x128 fp32 = 64 KiB) ----
    __builtin_amdgcn_s_barrier();
    float* st = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                const int col = wn * 64 + j * 32 + frow;
                st[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    const int col = (tid & 15) * 8;
    const int n = n0 + col;
    if (!CONV && ksplit > 1) {          // split-K: raw partial sums; splitk_epilogue_kernel adds them up in split order
        float* part = p.part + ((long long)(ks_id * p.batch + b) * p.M) * p.N;
#pragma unroll 2
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 16 + (tid >> 4);
            const int m = m0 + row;
            if (m >= p.M) continue;
            float* c = part + (long long)m * p.N + n;
            *(f32x4_t*)c = *(const f32x4_t*)(st + row * BN + col);
            *(f32x4_t*)(c + 4) = *(const f32x4_t*)(st + row * BN + col + 4);
        }
        return;
    }
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = p.bias ? p.bias[n + e] : 0.f;
    float gate[8];
    if (p.flags & PF_GEMM_GATE_RES) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gate[e] = p.gate ? p.gate[(long long)b * p.gate_stride + n + e] : 1.f;
    }
    const bool do_gelu = n >= p.gelu_from;
    if (n >= p.n_valid) return;
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 16 + (tid >> 4);
        const int m = m0 + row;
        if (m >= p.M) continue;
        float v[8];
        const f32x4_t v0 = *(const f32x4_t*)(st + row * BN + col);
        const f32x4_t v1 = *(const f32x4_t*)(st + row * BN + col + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bias[e]; v[4 + e] = v1[e] + bias[4 + e]; }
        if (do_gelu) {
            act8(v, p.flags);
        }
        long long coff;
        if (CONV && p.om.mode == 1) {
            const int hw = p.om.H * p.om.W;
            const int tt = m / hw, rem = m - tt * hw;
            const int hh = rem / p.om.W, ww = rem - hh * p.om.W;
            const int g = n / p.om.Cg, cc = n - g * p.om.Cg;
            const int shw = p.om.sh * p.om.sw;
            const int pt = g / shw, g2 = g - pt * shw;
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
            const long long roff = (CONV && p.om.mode == 1) ? coff : ((long long)b * p.sR + (long long)m * p.ldr + n);
            unpack8(*(const u32x4_t*)(p.res + roff), rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = rv[e] + gate[e] * v[e];
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
}

// second launch of a split-K GEMM: C = epi(sum over the splits, in split order, of the fp32 partial sums) -- the same
// epilogue arithmetic as gemm_kernel's.  One thread per (row, 8 columns).
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const Args p) {
    const int chunks = p.N >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.batch * p.M * chunks;
    if (idx >= total) return;
    const int ck = (int)(idx % chunks);
    const long long rowi = idx / chunks;
    const int m = (int)(rowi % p.M), b = (int)(rowi / p.M);
    const int n = ck * 8;
    if (n >= p.n_valid) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s_ = 0; s_ < p.ksplit; ++s_) {
        const float* c = p.part + (((long long)(s_ * p.batch + b) * p.M) + m) * p.N + n;
        const f32x4_t v0 = *(const f32x4_t*)c, v1 = *(const f32x4_t*)(c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += v0[e]; v[4 + e] += v1[e]; }
    }
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e];
    }
    if (n >= p.gelu_from) act8(v, p.flags);
    const long long coff = (long long)b * p.sC + (long long)m * p.ldc + n;
    if (p.flags & PF_GEMM_GATE_RES) {
        float rv[8];
        unpack8(*(const u32x4_t*)(p.res + (long long)b * p.sR + (long long)m * p.ldr + n), rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rv[e] + (p.gate ? p.gate[(long long)b * p.gate_stride + n + e] : 1.f) * v[e];
    }
    if (p.flags & PF_GEMM_OUT_F32) {
        float* c = (float*)p.C + coff;
        *(f32x4_t*)c = (f32x4_t){v[0], v[1], v[2], v[3]};
        *(f32x4_t*)(c + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
    } else {
        *(u32x4_t*)((bf16_t*)p.C + coff) = pack8(v);
    }
}

}  // namespace

int pf_set_err(const char* m);
#define set_err pf_set_err
int pf_gemm256_pick(long long M_total, int M, int batch, int N, int force);
int pf_gemm256_launch(const pfgemm::Args& a, int bn, bool conv, hipStream_t stream);
int pf_gemm8p_launch(const pfgemm::Args& a, bool conv, hipStream_t stream, void* ws, long long ws_bytes);   // gemm8p.hip: persistent 256 x 256 tiles
void pf_gemm8p_set_tail_split(bool on);
void pf_gemm8p_set_tail_overhead(int k_tiles);
void pf_gemm8p_set_stagger(int cycles);
void pf_gemm8p_set_epi_mode(int mode);
int pf_gemm8p_set_reserved_cus(int n);
int pf_gemm8p_workgroups();
long long pf_gemm8p_workspace_bytes();
bool pf_gemm8p_supports(const pfgemm::Args& a, bool conv);
bool pf_conv_narrow_supports(const pf_conv_desc* d);                       // convnarrow.hip: <= 8 output channels (conv_out)
int pf_conv_narrow_launch(const pf_conv_desc* d, hipStream_t stream);
bool pf_conv_halo_supports(const pf_conv_desc* d, bool wide);              // convhalo.hip: 3x3x3, N % 128 == 0, 128 / 256 / 512 input channels
int pf_conv_halo_launch(const pf_conv_desc* d, double* gn_stats, int gn_C, hipStream_t stream);

// tile-width policy of the 256-row ping-pong kernels: 0 (auto, default) | 128 | 192 | 256 (that width when it divides N) | -1 (never)
static int g_gemm256_force = 0;
static int gemm256_force() { return g_gemm256_force; }
// gemm8p (persistent 256 x 256 tiles): 1 = whenever legal (policy 8), 0 = automatic, -1 = never (policy -8)
static int g_gemm8p_mode = 0;
static bool g_narrow_enabled = true;         // pf_gemm_set_policy(-3) / (3): never / again the narrow-N conv kernel
static bool g_halo_enabled = true;           // pf_gemm_set_policy(-5) / (5): never / again the LDS-halo direct conv
static bool g_halo_maps = true;              // pf_gemm_set_policy(-7) / (7): the upsamplers' shuffled output maps stay with the implicit GEMM / take the halo kernel
static bool g_halo_wide = true;              // pf_gemm_set_policy(-6) / (6): only the N = 128 layers / also the 256- and 512-filter layers
static bool g_splitk_enabled = true;          // pf_gemm_set_policy(-2) / (2): never / again split K for skinny problems
static bool g_stage3_enabled = true;          // pf_gemm_set_policy(1202) / (1203): the 128 x 128 kernel always with two stages / three for grids <= 256
static bool g_split_cap = true;               // pf_gemm_set_policy(1204) / (1205): the K split capped at one workgroup per CU (so that it runs three stages) / not
static const bool g_gemm8p_auto = true;       // measured ahead of gemm256 on every large DiT shape (profiles/r02_gemm_ab*.log)
// 1 = a launch of >= 192 tiles (whole rounds + tail), 2 = a MID-SIZE launch (32 .. 128 tiles, e.g. a sequence-parallel rank's
// N = 1920 projections at P = 4 / 8: flux_block.py:868-872, 914-942 on L / P rows) that runs the persistent kernel on the
// WHOLE chip by splitting every tile's K range over 256 / T workgroups (the tail-split path with no full round in front:
// gemm8p.hip tail_plan) -- only with caller scratch, only when the split pays by that plan; 0 = not a gemm8p problem
int pf_gemm8p_mid_split(int tiles, int nk);                                // gemm8p.hip: parts per tile (1 = no gain)
static int use_gemm8p_kind(int M, int batch, int N, int K) {
    if (g_gemm8p_mode < 0 || N % 8 || K % 64) return 0;
    if (g_gemm8p_mode > 0) return 1;
    if (!g_gemm8p_auto || gemm256_force() != 0) return 0;          // an explicit tile-width policy addresses the older kernels
    // automatic: problems of at least 3/4 of a round of 256 x 256 tiles whose N tail wastes < 7 % of the columns
    // (a sequence-parallel rank at P = 8 runs the 7d-wide projections with 16 x 53 tiles and the 4d-wide with 16 x 30)
    const long long tiles = (long long)((M + 255) / 256) * batch * ((N + 255) / 256);
    const int n256 = (N + 255) / 256 * 256;
    if ((n256 - N) * 100 >= 7 * N) return 0;
    if (tiles >= 192) return 1;
    // (round 5: where the 256-row ping-pong kernel fills the chip with 256 x 128 tiles -- a P = 8 rank's N = 1920 projections: 240
    //  tiles -- it now runs 1.03-1.06 PFLOP/s as ONE launch and beats the K split's 0.93-1.02: profiles/r05_gemm_rank_shapes.log)
    if (tiles >= 32 && tiles <= 128 && M >= 192 && pf_gemm256_pick((long long)M * batch, M, batch, N, 0) == 0 &&
        pf_gemm8p_mid_split((int)tiles, K / 64) > 1) return 2;
    return 0;
}
static bool use_gemm8p(int M, int batch, int N, int K) { return use_gemm8p_kind(M, batch, N, K) != 0; }
extern "C" int pf_gemm_set_policy(int force) {
    if (force == 8 || force == -8) { g_gemm8p_mode = force > 0 ? 1 : -1; g_gemm256_force = 0; return 0; }
    if (force == 2 || force == -2) { g_splitk_enabled = force > 0; return 0; }
    if (force == 3 || force == -3) { g_narrow_enabled = force > 0; return 0; }
    if (force == 4 || force == -4) { pf_gemm8p_set_tail_split(force > 0); return 0; }
    if (force == 5 || force == -5) { g_halo_enabled = force > 0; return 0; }
    if (force == 6 || force == -6) { g_halo_wide = force > 0; return 0; }
    if (force == 7 || force == -7) { g_halo_maps = force > 0; return 0; }
#ifdef PF_LAB_HOOKS      // measurement-only switches: compiled into libpyflow_hip_lab.so (`make lab`), absent from the shipping library
    if (force == 9 || force == -9) { pf_gemm8p_set_stagger(force > 0 ? 290 : 0); return 0; }
    if (force >= 400 && force < 600) { pf_gemm8p_set_tail_overhead(force - 400); return 0; }   // tail_plan's fixed cost
    if (force == 1000 || force == 1001) { pf_gemm8p_set_epi_mode(force - 1000); return 0; }     // Args::epi_mode
    if (force == 1202 || force == 1203) { g_stage3_enabled = force == 1203; return 0; }         // stages of the 128 x 128 kernel
    if (force == 1204 || force == 1205) { g_split_cap = force == 1204; return 0; }              // K split capped at 256 workgroups
    if (force >= 100000 && force < 200000) { pf_gemm8p_set_stagger(-(force - 100000)); return 0; }   // stamp builds: window start (K-tile)
#endif
    if (force >= 2000 && force <= 2128) {    // CUs the persistent launches leave to communication kernels
        if (pf_gemm8p_set_reserved_cus(force - 2000)) return set_err("pf_gemm_set_policy: the reservation leaves fewer than 64 CUs to the persistent kernel");
        return 0;
    }
    if (force != 0 && force != -1 && force != 128 && force != 192 && force != 256)
        return set_err("pf_gemm_set_policy: force must be 0, -1, +-2 .. +-8, 128, 192, 256 or 2000 + R");
    g_gemm256_force = force;
    g_gemm8p_mode = 0;
    g_splitk_enabled = true;
    pf_gemm8p_set_tail_split(true);
    pf_gemm8p_set_tail_overhead(4);          // the measurement hook (400 + c) does not outlive a reset to automatic
    pf_gemm8p_set_stagger(0);
    pf_gemm8p_set_epi_mode(1);
    g_stage3_enabled = true;
    g_split_cap = true;
    return 0;
}
// Scratch that pays for this problem (pf_gemm_desc.workspace): 0 = none is used.  Large problems on the persistent 256 x 256
// kernel split their tail tiles along K with one 256-KiB slot per workgroup; skinny problems split K over ~256 workgroups.
extern "C" long long pf_gemm_workspace_bytes(int M, int batch, int N, int K) {
    if (M <= 0 || batch <= 0 || N <= 0 || K <= 0) return 0;
    if (use_gemm8p(M, batch, N, K)) return pf_gemm8p_workspace_bytes();
    if (pf_gemm256_pick((long long)M * batch, M, batch, N, gemm256_force())) return 0;
    const int grid = (N / BN) * ((M + BM - 1) / BM) * batch, nk = K / BK;
    if (grid >= 128 || nk < 8 || N % BN) return 0;
    int ks = (256 + grid - 1) / grid;
    ks = ks < nk / 4 ? ks : nk / 4;
    return ks > 1 ? (long long)ks * batch * M * N * 4 : 0;
}
// The kernel a problem of this size takes WHEN the caller brings the scratch pf_gemm_workspace_bytes asks for and no QK
// epilogue: 0 = 128x128 kernel, 8 = gemm8p_kernel, BN = gemm256_kernel<BN>.  (pf_gemm_which_desc: the decision for one
// descriptor as pf_gemm_bf16 takes it.)
extern "C" int pf_gemm_which(int M, int batch, int N, int K) {
    if (use_gemm8p(M, batch, N, K)) return 8;
    return pf_gemm256_pick((long long)M * batch, M, batch, N, gemm256_force());
}
// pf_gemm_bf16's routing, shared with pf_gemm_which_desc: 8 = the persistent kernel (a whole-round launch, or a mid-size one
// that brings enough scratch to split K and has no QK epilogue: the split's second launch applies none), else the 256-row
// kernel's tile width, else 0 = the 128 x 128 kernel
static int gemm_route(const pf_gemm_desc* d, const Args& a) {
    const bool qk = d->qk_d > 0;
    const int kind8 = use_gemm8p_kind(d->M, d->batch, d->N, d->K);
    const bool ws8 = !qk && d->workspace && d->workspace_bytes >= pf_gemm8p_workspace_bytes();
    if ((kind8 == 1 || (kind8 == 2 && ws8)) && pf_gemm8p_supports(a, false)) return 8;
    return pf_gemm256_pick((long long)d->M * d->batch, d->M, d->batch, d->N, gemm256_force());
}
// workgroups of a persistent launch = CUs of the device - the CUs reserved for communication kernels (policy 2000 + R)
extern "C" int pf_gemm_workgroups(void) { return pf_gemm8p_workgroups(); }
// a grouped descriptor is served as ONE launch only by the persistent kernel (decided on the first problem)
static bool grouped_ok(const pf_gemm_desc* d) {
    Args a{};
    a.N = d->N; a.gelu_from = d->gelu_from < 0 ? d->N : d->gelu_from; a.flags = d->flags;
    return d->K > 0 && d->K % BK == 0 && gemm_route(d, a) == 8;
}
extern "C" int pf_gemm_which_desc(const pf_gemm_desc* d) {
    if (!d || d->M <= 0 || d->batch <= 0 || d->N <= 0 || d->K <= 0) return -100;
    Args a{};
    a.N = d->N; a.gelu_from = d->gelu_from < 0 ? d->N : d->gelu_from; a.flags = d->flags;
    return gemm_route(d, a);
}
// the second problem of a grouped descriptor as a descriptor of its own (what runs when the grouping is not taken)
static pf_gemm_desc second_problem(const pf_gemm_desc* d) {
    pf_gemm_desc e = *d;
    e.A = d->A2; e.W = d->W2; e.C = d->C2; e.bias = d->bias2; e.res = d->res2; e.gate = d->gate2;
    e.M = d->M2; e.strideA = d->strideA2; e.strideC = d->strideC2; e.strideR = d->strideR2;
    e.qk_wq = d->qk_wq2; e.qk_wk = d->qk_wk2; e.qk_row0 = d->qk_row0_2;
    e.M2 = 0; e.A2 = e.W2 = nullptr; e.C2 = nullptr;
    return e;
}
static bool grouped_ok(const pf_gemm_desc* d);
extern "C" int pf_gemm_bf16(const pf_gemm_desc* d, hipStream_t stream) {
    if (!d || !d->A || !d->W || !d->C) return set_err("pf_gemm_bf16: null operand");
    if (d->M <= 0 || d->batch <= 0) return set_err("pf_gemm_bf16: empty problem");
    if (d->M2 < 0 || (d->M2 > 0 && (!d->A2 || !d->W2 || !d->C2))) return set_err("pf_gemm_bf16: grouped launch: bad second problem");
    if (d->M2 > 0 && !grouped_ok(d)) {      // not a launch of the persistent kernel: the two problems as two launches
        pf_gemm_desc first = *d;
        first.M2 = 0;
        const pf_gemm_desc second = second_problem(d);
        const int rc = pf_gemm_bf16(&first, stream);
        return rc ? rc : pf_gemm_bf16(&second, stream);
    }
    if (d->K % BK != 0 || d->K <= 0) return set_err("pf_gemm_bf16: K must be a positive multiple of 64");
    if ((d->lda % 8) || (d->ldw % 8) || (d->ldc % 8)) return set_err("pf_gemm_bf16: leading dims must be multiples of 8");
    if ((d->flags & PF_GEMM_GATE_RES) && !d->res) return set_err("pf_gemm_bf16: GATE_RES needs res");
    Args a{};
    a.A = (const bf16_t*)d->A; a.W = (const bf16_t*)d->W; a.C = d->C;
    a.bias = d->bias; a.res = (const bf16_t*)d->res; a.gate = d->gate;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldr = d->ldr;
    a.sA = d->strideA; a.sC = d->strideC; a.sR = d->strideR; a.gate_stride = d->gate_stride; a.batch = d->batch;
    a.gelu_from = d->gelu_from < 0 ? d->N : d->gelu_from; a.flags = d->flags;
    a.out_scale = 1.f; a.n_valid = d->N;
    if (a.gelu_from % 8) return set_err("pf_gemm_bf16: gelu_from must be a multiple of 8");
    a.group_m = 0;
    const bool qk = d->qk_d > 0;
    if (qk) {
        if (!d->qk_rope || !d->qk_wq || !d->qk_wk) return set_err("pf_gemm_bf16: qk_d > 0 needs qk_rope / qk_wq / qk_wk");
        const int hs = d->qk_head_stride;
        if (hs < 0 || hs % 64) return set_err("pf_gemm_bf16: qk_head_stride must be 0 or a multiple of 64");
        if (hs > 0) {        // head-major: qk_d / 64 heads of hs columns, the K / Q blocks at fixed columns inside every head
            if ((d->qk_d % 64) || (d->qk_q_col0 >= 0 && (d->qk_q_col0 % 64 || d->qk_q_col0 + 64 > hs)) ||
                (d->qk_k_col0 >= 0 && (d->qk_k_col0 % 64 || d->qk_k_col0 + 64 > hs)) || (long long)(d->qk_d / 64) * hs > d->N)
                return set_err("pf_gemm_bf16: head-major QK blocks must be 64-column aligned inside a head and the heads inside N");
            if (a.gelu_from < (d->qk_d / 64) * hs) return set_err("pf_gemm_bf16: the activation columns overlap the heads");
        } else {
        if ((d->qk_d % 64) || (d->qk_q_col0 >= 0 && d->qk_q_col0 % 64) || (d->qk_k_col0 >= 0 && d->qk_k_col0 % 64) ||
            (d->qk_q_col0 >= 0 && d->qk_q_col0 + d->qk_d > d->N) || (d->qk_k_col0 >= 0 && d->qk_k_col0 + d->qk_d > d->N))
            return set_err("pf_gemm_bf16: the QK blocks must be 64-column aligned and inside N");
        if (a.gelu_from < d->N && ((d->qk_q_col0 >= 0 && a.gelu_from < d->qk_q_col0 + d->qk_d) || (d->qk_k_col0 >= 0 && a.gelu_from < d->qk_k_col0 + d->qk_d)))
            return set_err("pf_gemm_bf16: the activation columns overlap a QK block");
        }
        if (d->flags & (PF_GEMM_GATE_RES | PF_GEMM_OUT_F32)) return set_err("pf_gemm_bf16: QK epilogue needs a plain bf16 output");
        a.qk_rope = d->qk_rope; a.qk_wq = d->qk_wq; a.qk_wk = d->qk_wk;
        a.qk_d = d->qk_d; a.qk_q0 = d->qk_q_col0; a.qk_k0 = d->qk_k_col0; a.qk_row0 = d->qk_row0;
        a.qk_eps = d->qk_eps; a.qk_qs = d->qk_q_scale; a.qk_hs = hs;
    }
    // the separate pass for every kernel choice whose epilogue cannot do it (same arithmetic: common.h qk_rope8)
    auto qk_pass = [&]() -> int {
        return pf_qk_norm_rope(d->C, d->ldc, d->strideC, d->qk_q_col0, d->qk_k_col0, d->qk_wq, d->qk_wk, nullptr, nullptr,
                               d->qk_rope + (long long)d->qk_row0 * 64, d->batch, d->M, 0, d->qk_d / 64, d->qk_eps,
                               d->qk_q_scale, d->qk_head_stride > 0 ? d->qk_head_stride : 64, stream);
    };
    // (a mid-size problem takes the persistent kernel only with its scratch: without it the older kernels fill the chip better)
    const int route = gemm_route(d, a);
    const bool g8 = route == 8;
    if (d->M2 > 0) {                         // (grouped_ok: route == 8)
        if ((d->flags & PF_GEMM_GATE_RES) && !d->res2) return set_err("pf_gemm_bf16: GATE_RES needs res2");
        if (qk && (!d->qk_wq2 || !d->qk_wk2)) return set_err("pf_gemm_bf16: grouped QK epilogue needs qk_wq2 / qk_wk2");
        a.pr[1] = Prob{(const bf16_t*)d->A2, (const bf16_t*)d->W2, d->C2, d->bias2, (const bf16_t*)d->res2, d->gate2,
                       d->strideA2, d->strideC2, d->strideR2, d->qk_wq2, d->qk_wk2, d->M2, d->qk_row0_2};
        a.tiles2_m = (d->M2 + 255) / 256;
    }
    const int bn256 = g8 ? 0 : route;
    if (!g8 && !bn256 && d->N % BN != 0)
        return set_err("pf_gemm_bf16: N must be a multiple of 128 (or of 192 with the 256x192 kernel)");
    if (g8) {
        // (with a QK epilogue the tail tiles are not split: gemm8p_tail_kernel applies no norm / rotation)
        pf_gemm8p_launch(a, false, stream, qk ? nullptr : d->workspace, qk ? 0 : d->workspace_bytes);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) return set_err(hipGetErrorString(e2));
        return 0;
    }
    a.qk_d = 0;                                   // every other kernel: plain epilogue, then the separate pass
    if (const int bn = bn256) {
        pf_gemm256_launch(a, bn, false, stream);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) return set_err(hipGetErrorString(e2));
        return qk ? qk_pass() : 0;
    }
    int grid = (d->N / BN) * ((d->M + BM - 1) / BM) * d->batch;
    // skinny problems (the 128-row text stream, the prompt encoders): < 128 workgroups, each a chain of K / 64 dependent
    // memory latencies.  With scratch from the caller the K range is split so that ~256 workgroups run chains of >= 4
    // K-tiles; a second launch sums the parts in split order and applies the epilogue.
    const int nk = d->K / BK;
    if (d->workspace && grid < 128 && nk >= 8 && g_splitk_enabled) {
        // ~256 workgroups -- and, since round 5, AT MOST 256: one per CU, which run the three-stage ring (a split of 90 tiles
        // into 3 x 90 = 270 two-stage workgroups ran 259 TFLOP/s where 2 x 90 three-stage ones run 290: profiles/
        // r05_gemm128_three_stage_ab.log)
        int ks = (256 + grid - 1) / grid;
        if (g_split_cap && ks > 1 && grid * ks > 256) ks = 256 / grid;
        ks = ks < nk / 4 ? ks : nk / 4;
        const long long per_split = (long long)d->batch * d->M * d->N * 4;
        if (per_split * ks > d->workspace_bytes) ks = (int)(d->workspace_bytes / per_split);
        if (ks > 1) {
            a.ksplit = ks;
            a.part = (float*)d->workspace;
            grid *= ks;
        }
    }
    if (grid <= 256 && g_stage3_enabled) {          // at most one workgroup per CU: the three-stage ring (see the file header)
        PF_SET_MAX_LDS_ONCE((gemm_kernel<false, 3>), SMEM_BYTES3);
        hipLaunchKernelGGL((gemm_kernel<false, 3>), dim3(grid), dim3(256), SMEM_BYTES3, stream, a);
    } else {
        PF_SET_MAX_LDS_ONCE((gemm_kernel<false, 2>), SMEM_BYTES);
        hipLaunchKernelGGL((gemm_kernel<false, 2>), dim3(grid), dim3(256), SMEM_BYTES, stream, a);
    }
    if (a.ksplit > 1) {
        const long long total = (long long)d->batch * d->M * (d->N / 8);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(hipGetErrorString(e));
    return qk ? qk_pass() : 0;
}

// ---- CausalConv3d: one routing decision shared by the launch and by pf_conv3d_fuses_gn_stats
static void conv_args(const pf_conv_desc* d, Args& a) {
    const int K = d->kt * d->kh * d->kw * d->Cin;
    a.A = (const bf16_t*)d->X; a.W = (const bf16_t*)d->W; a.C = d->Y;
    a.bias = d->bias; a.res = (const bf16_t*)d->res; a.gate = nullptr;
    a.M = d->T * d->H * d->W_; a.N = d->N; a.K = K; a.lda = 0; a.ldw = K; a.ldc = d->N; a.ldr = d->N;
    a.sA = 0; a.sC = 0; a.sR = 0; a.gate_stride = 0; a.batch = 1;
    a.gelu_from = d->N; a.flags = d->flags; a.out_scale = d->out_scale == 0.f ? 1.f : d->out_scale;
    a.n_valid = d->n_valid > 0 ? d->n_valid : d->N;
    a.cg = ConvGeom{d->H, d->W_, d->Hp, d->Wp, d->Cin, d->kt, d->kh, d->kw, d->in_base_off,
                    d->in_sh > 0 ? d->in_sh : 1, d->in_sw > 0 ? d->in_sw : 1, d->in_st > 0 ? d->in_st : 1};
    a.om = OutMap{1, d->H, d->W_, d->st, d->sh, d->sw, d->Cg, d->Hop, d->Wop, d->out_base_off, d->Cout_pitch, d->out_t_shift};
}
// -1 = conv_narrow_kernel, -2 = conv_halo128_kernel, 8 = gemm8p_kernel<true, 0>, 128 / 192 / 256 = gemm256_kernel<BN, true>,
// 0 = gemm_kernel<true>
static int conv_route(const pf_conv_desc* d, const Args& a) {
    if (g_narrow_enabled && pf_conv_narrow_supports(d)) return -1;
    if (g_halo_enabled && gemm256_force() == 0 && pf_conv_halo_supports(d, g_halo_wide) &&
        (g_halo_maps || (d->st == 1 && d->sh == 1 && d->sw == 1))) return -2;
    // (kind 2 = a mid-size launch that pays only with a K split through scratch: a conv never splits -- it stays with the
    //  256-row kernels, which also fuse the GroupNorm statistics)
    if (use_gemm8p_kind(a.M, 1, a.N, a.K) == 1 && a.n_valid % 8 == 0 && pf_gemm8p_supports(a, true)) return 8;
    return pf_gemm256_pick(a.M, a.M, 1, a.N, gemm256_force());
}
// GroupNorm statistics in the conv epilogue: only the 256-row ping-pong kernel at BN = 128 / 256 accumulates them, for a
// plain output map (no pixel shuffle / depth-to-time / frame drop) whose frames are whole numbers of 256-pixel tiles
static bool conv_fuses_gn_stats(const pf_conv_desc* d, const Args& a, int route) {
    if (!d->gn_stats || d->gn_C <= 0 || a.n_valid > d->gn_C) return false;
    if (d->st != 1 || d->sh != 1 || d->sw != 1 || d->out_t_shift != 0 || (d->flags & PF_GEMM_OUT_F32)) return false;
    if (route == -2) return true;                  // the halo kernel's tiles never straddle a frame
    if (((long long)d->H * d->W_) % 256 != 0) return false;
    return route == 128 || route == 256;
}
static const char* conv_check(const pf_conv_desc* d) {
    if (!d || !d->X || !d->W || !d->Y) return "pf_conv3d_bf16: null operand";
    if (d->Cin % BK != 0) return "pf_conv3d_bf16: Cin must be a multiple of 64 (pad channels)";
    if (d->N % BN != 0) return "pf_conv3d_bf16: N must be a multiple of 128 (pad filters)";
    if (d->T <= 0 || d->H <= 0 || d->W_ <= 0) return "pf_conv3d_bf16: empty problem";
    if ((long long)d->T * d->H * d->W_ > 0x7fffffffll) return "pf_conv3d_bf16: more than 2^31 output pixels";
    return nullptr;
}
extern "C" int pf_conv3d_which(const pf_conv_desc* d) {
    if (conv_check(d)) return -100;
    Args a{};
    conv_args(d, a);
    return conv_route(d, a);
}
extern "C" int pf_conv3d_fuses_gn_stats(const pf_conv_desc* d) {
    if (conv_check(d)) return 0;
    Args a{};
    conv_args(d, a);
    return conv_fuses_gn_stats(d, a, conv_route(d, a)) ? 1 : 0;
}

extern "C" int pf_conv3d_bf16(const pf_conv_desc* d, hipStream_t stream) {
    if (const char* msg = conv_check(d)) return set_err(msg);
    Args a{};
    conv_args(d, a);
    const int route = conv_route(d, a);
    if (route == -1) return pf_conv_narrow_launch(d, stream);
    if (route == -2) {
        const bool st = conv_fuses_gn_stats(d, a, route);
        pf_conv_halo_launch(d, st ? d->gn_stats : nullptr, d->gn_C, stream);
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return set_err(hipGetErrorString(eh));
        return 0;
    }
    if (d->Cg % 8 || d->Cout_pitch % 8) return set_err("pf_conv3d_bf16: Cg / Cout_pitch must be multiples of 8");
    if (conv_fuses_gn_stats(d, a, route)) { a.gn_stats = d->gn_stats; a.gn_C = d->gn_C; }
    if (route == 8) {
        pf_gemm8p_launch(a, true, stream, nullptr, 0);
    } else if (route > 0) {
        pf_gemm256_launch(a, route, true, stream);
    } else {
        const int grid = (a.N / BN) * ((a.M + BM - 1) / BM);
        if (grid <= 256 && g_stage3_enabled) {
            PF_SET_MAX_LDS_ONCE((gemm_kernel<true, 3>), SMEM_BYTES3);
            hipLaunchKernelGGL((gemm_kernel<true, 3>), dim3(grid), dim3(256), SMEM_BYTES3, stream, a);
        } else {
            PF_SET_MAX_LDS_ONCE((gemm_kernel<true, 2>), SMEM_BYTES);
            hipLaunchKernelGGL((gemm_kernel<true, 2>), dim3(grid), dim3(256), SMEM_BYTES, stream, a);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(hipGetErrorString(e));
    return 0;
}
