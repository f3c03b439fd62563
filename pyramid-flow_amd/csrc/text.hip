// Prompt-encoder kernels (the step right before the sampling path: FluxTextEncoderWithMask / SD3TextEncoderWithMask,
// pyramid_dit/flux_modules/modeling_text_encoder.py:15-134, pyramid_dit/mmdit_modules/modeling_text_encoder.py:15-139).
// The encoders themselves live in transformers (pinned ==4.39.3 by the reference's requirements.txt): T5EncoderModel
// (models/t5/modeling_t5.py: T5LayerNorm, T5Attention with the bucketed relative-position bias, T5DenseGatedActDense)
// and CLIPTextModel(WithProjection) (models/clip/modeling_clip.py).  Their GEMMs go through pf_gemm_bf16; this file
// holds what is left, all HBM/latency-bound at 77..128 tokens:
//   pf_embed_rows        token-embedding gather (+ learned position rows for CLIP)
//   pf_rmsnorm           T5LayerNorm: x * rsqrt(mean(x^2) + eps) * w, no mean subtraction, no bias
//   pf_glu_mul           gated FFN product  y = x[:, :F] * x[:, F:]  (the GELU half was applied by the GEMM epilogue)
//   pf_attention_small   softmax(Q K^T * scale + bias[h] + masks) V for L <= 256, head_dim 64, with an additive
//                        per-head fp32 bias (T5), a key-padding mask (T5) and / or a causal mask (CLIP)
#include "common.h"
#include "pyflow_hip.h"

int pf_set_err(const char* m);

namespace {

PF_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one thread per 8 channels
__global__ __launch_bounds__(256) void embed_rows_kernel(const bf16_t* table, const int* ids, const bf16_t* pos,
                                                         bf16_t* out, int D, int n, int L, int ldo, int vocab) {
    const int nch = D >> 3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * nch) return;
    const int r = (int)(i / nch), ch = (int)(i - (long long)r * nch);
    int id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    u32x4_t v = *(const u32x4_t*)(table + (long long)id * D + ch * 8);
    if (pos) {
        float a[8], b[8];
        unpack8(v, a);
        unpack8(*(const u32x4_t*)(pos + (long long)(r % L) * D + ch * 8), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
        v = pack8(a);
    }
    *(u32x4_t*)(out + (long long)r * ldo + ch * 8) = v;
}

// one wave per row; D <= 64*8*MAXC
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* x, bf16_t* y, const float* w, int D, int ldx, int ldy,
                                                      int nrows, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const bf16_t* xp = x + (long long)row * ldx;
    bf16_t* yp = y + (long long)row * ldy;
    const int nch = D >> 3;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            unpack8(*(const u32x4_t*)(xp + ch * 8), v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[c][e] * v[c][e];
        }
    }
    const float r = rsqrtf(wave_sum(s) / D + eps);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            const f32x4_t w0 = *(const f32x4_t*)(w + ch * 8), w1 = *(const f32x4_t*)(w + ch * 8 + 4);
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = v[c][e] * r * w0[e]; o[4 + e] = v[c][4 + e] * r * w1[e]; }
            *(u32x4_t*)(yp + ch * 8) = pack8(o);
        }
    }
}

__global__ __launch_bounds__(256) void glu_mul_kernel(const bf16_t* x, bf16_t* y, int rows, int F, int ldx, int ldy) {
    const int nch = F >> 3;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * nch) return;
    const int r = (int)(i / nch), ch = (int)(i - (long long)r * nch);
    float a[8], b[8];
    unpack8(*(const u32x4_t*)(x + (long long)r * ldx + ch * 8), a);
    unpack8(*(const u32x4_t*)(x + (long long)r * ldx + F + ch * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= b[e];
    *(u32x4_t*)(y + (long long)r * ldy + ch * 8) = pack8(a);
}

// grid (H, B, ceil(L/128)), 128 threads = one query row each.  K and V of the (b, h) pair sit in LDS as bf16 rows of
// 64; every lane reads the same key row at the same time (LDS broadcast), so the loop is pure VALU: 2 x 64 FMAs per key.
__global__ __launch_bounds__(128) void attn_small_kernel(pf_attn_small_desc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + (size_t)d.L * 64;
    const int h = blockIdx.x, b = blockIdx.y;
    const int i = blockIdx.z * 128 + threadIdx.x;
    const bf16_t* Kg = (const bf16_t*)d.K + (long long)b * d.strideK + h * 64;
    const bf16_t* Vg = (const bf16_t*)d.V + (long long)b * d.strideV + h * 64;
    for (int c = threadIdx.x; c < d.L * 8; c += 128) {
        const int j = c >> 3, part = c & 7;
        *(u32x4_t*)(Ks + j * 64 + part * 8) = *(const u32x4_t*)(Kg + (long long)j * d.ldk + part * 8);
        *(u32x4_t*)(Vs + j * 64 + part * 8) = *(const u32x4_t*)(Vg + (long long)j * d.ldv + part * 8);
    }
    __syncthreads();
    if (i >= d.L) return;
    float q[64];
    {
        const bf16_t* Qg = (const bf16_t*)d.Q + (long long)b * d.strideQ + (long long)i * d.ldq + h * 64;
#pragma unroll
        for (int part = 0; part < 8; ++part) {
            unpack8(*(const u32x4_t*)(Qg + part * 8), q + part * 8);
        }
#pragma unroll
        for (int e = 0; e < 64; ++e) q[e] *= d.scale;
    }
    float acc[64];
#pragma unroll
    for (int e = 0; e < 64; ++e) acc[e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* brow = d.bias ? d.bias + ((long long)h * d.L + i) * d.L : nullptr;
    const int* km = d.key_mask ? d.key_mask + (long long)b * d.L : nullptr;
    const int jend = d.causal ? (blockIdx.z * 128 + 128 < d.L ? blockIdx.z * 128 + 128 : d.L) : d.L;
    for (int j = 0; j < jend; ++j) {
        if (km && km[j] == 0) continue;                       // block-uniform
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int part = 0; part < 8; ++part) {
            float kf[8];
            unpack8(*(const u32x4_t*)(Ks + j * 64 + part * 8), kf);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                s0 = fmaf(q[part * 8 + e], kf[e], s0);
                s1 = fmaf(q[part * 8 + e + 1], kf[e + 1], s1);
            }
        }
        float s = s0 + s1;
        if (brow) s += brow[j];
        if (d.causal && j > i) s = -INFINITY;
        const float mn = fmaxf(m, s);
        if (mn == -INFINITY) continue;                         // nothing visible yet for this row
        const float c = __expf(m - mn), p = __expf(s - mn);
        m = mn;
        l = l * c + p;
#pragma unroll
        for (int part = 0; part < 8; ++part) {
            float vf[8];
            unpack8(*(const u32x4_t*)(Vs + j * 64 + part * 8), vf);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[part * 8 + e] = fmaf(acc[part * 8 + e], c, p * vf[e]);
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16_t* Og = (bf16_t*)d.O + (long long)b * d.strideO + (long long)i * d.ldo + h * 64;
#pragma unroll
    for (int part = 0; part < 8; ++part) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = acc[part * 8 + e] * inv;
        *(u32x4_t*)(Og + part * 8) = pack8(o);
    }
}

}  // namespace

#define CHECK_LAUNCH()                                              \
    do {                                                            \
        hipError_t e_ = hipGetLastError();                          \
        if (e_ != hipSuccess) return pf_set_err(hipGetErrorString(e_)); \
    } while (0)

extern "C" int pf_embed_rows(const void* table, const int* ids, const void* pos, void* out, int D, int n, int L, int ldo,
                             int vocab, hipStream_t stream) {
    if (!table || !ids || !out) return pf_set_err("pf_embed_rows: null operand");
    if (D <= 0 || D % 8 || ldo % 8 || ldo < D) return pf_set_err("pf_embed_rows: D / ldo must be multiples of 8, ldo >= D");
    if (n <= 0 || vocab <= 0 || (pos && L <= 0)) return pf_set_err("pf_embed_rows: empty problem");
    const long long work = (long long)n * (D >> 3);
    embed_rows_kernel<<<(unsigned)((work + 255) / 256), 256, 0, stream>>>((const bf16_t*)table, ids, (const bf16_t*)pos,
                                                                         (bf16_t*)out, D, n, L, ldo, vocab);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_rmsnorm(const void* x, void* y, const float* w, int D, int rows, int ldx, int ldy, float eps,
                          hipStream_t stream) {
    if (!x || !y || !w) return pf_set_err("pf_rmsnorm: null operand");
    if (D <= 0 || D % 8 || D > 64 * 8 * 8) return pf_set_err("pf_rmsnorm: D must be a multiple of 8 and <= 4096");
    if (ldx % 8 || ldy % 8) return pf_set_err("pf_rmsnorm: leading dims must be multiples of 8");
    if (rows <= 0) return pf_set_err("pf_rmsnorm: empty problem");
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (D <= 64 * 8 * 2)
        rmsnorm_kernel<2><<<grid, 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, w, D, ldx, ldy, rows, eps);
    else if (D <= 64 * 8 * 4)
        rmsnorm_kernel<4><<<grid, 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, w, D, ldx, ldy, rows, eps);
    else
        rmsnorm_kernel<8><<<grid, 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, w, D, ldx, ldy, rows, eps);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_glu_mul(const void* x, void* y, int rows, int F, int ldx, int ldy, hipStream_t stream) {
    if (!x || !y) return pf_set_err("pf_glu_mul: null operand");
    if (F <= 0 || F % 8 || ldx % 8 || ldy % 8 || ldx < 2 * F || ldy < F)
        return pf_set_err("pf_glu_mul: F / leading dims must be multiples of 8, ldx >= 2F, ldy >= F");
    if (rows <= 0) return pf_set_err("pf_glu_mul: empty problem");
    const long long work = (long long)rows * (F >> 3);
    glu_mul_kernel<<<(unsigned)((work + 255) / 256), 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)y, rows, F, ldx, ldy);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_attention_small_bf16(const pf_attn_small_desc* d, hipStream_t stream) {
    if (!d || !d->Q || !d->K || !d->V || !d->O) return pf_set_err("pf_attention_small_bf16: null operand");
    if (d->L <= 0 || d->L > 256) return pf_set_err("pf_attention_small_bf16: L must be in 1..256");
    if (d->B <= 0 || d->H <= 0) return pf_set_err("pf_attention_small_bf16: empty problem");
    if ((d->ldq % 8) || (d->ldk % 8) || (d->ldv % 8) || (d->ldo % 8))
        return pf_set_err("pf_attention_small_bf16: leading dims must be multiples of 8");
    const dim3 grid(d->H, d->B, (d->L + 127) / 128);
    attn_small_kernel<<<grid, 128, (size_t)d->L * 64 * 2 * 2, stream>>>(*d);
    CHECK_LAUNCH();
    return 0;
}
