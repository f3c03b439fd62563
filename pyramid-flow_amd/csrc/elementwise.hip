// HBM-bound token-wise kernels of the DiT path (all bf16 I/O as 16-byte pieces, fp32 math):
//   pf_ln_modulate     LayerNorm(no affine, eps 1e-6) * (1+scale) + shift      (K6/K12: AdaLayerNormZero /
//                      ZeroSingle / Continuous, modeling_normalization.py:107-249; nn.LayerNorm flux_block.py:1022,1033)
//   pf_qk_norm_rope    RMSNorm over head_dim (eps 1e-6, modeling_normalization.py:66-79) on q and k followed by
//                      the adjacent-pair RoPE rotation (flux_block.py:34-39), in place on the fused QKV buffer
//   pf_gemv_f32        y[b,:] (+)= W x[b,:] + bias for tiny batch (conditioning MLPs modeling_embedding.py:185-200
//                      and every block's AdaLN `linear(silu(temb))`), optional SiLU on the input
//   pf_timestep_embed  sinusoidal embedding [cos | sin] (modeling_embedding.py:11-62, flip_sin_to_cos, shift 0)
//   pf_patchify        'b c t (h p1) (w p2) -> b (t h w) (p1 p2 c)'  (modeling_pyramid_flux.py:286-287)
//   pf_cfg_euler_step  unpatchify (flux:383-388) + CFG combine (pipeline.py:771-776) + Euler step
//                      (scheduling_flow_matching.py:278-286) fused, fp32 latent state
//   pf_copy_rows / pf_renoise_upsample / pf_avgpool2 / pf_cast: small host-loop helpers (pipeline.py:555-570, 729-743)
#include "common.h"
#include "pyflow_hip.h"

int pf_set_err(const char* m);

namespace {

PF_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per row; D <= 64*8*MAXC elements
template <int MAXC>
__global__ __launch_bounds__(256) void ln_mod_kernel(const bf16_t* x, bf16_t* y, const float* shift, const float* scale,
                                                     int D, int rows_per_batch, long long x_bstride, long long y_bstride,
                                                     int ldx, int ldy, int mod_bstride, int nrows, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int b = row / rows_per_batch, r = row - b * rows_per_batch;
    const bf16_t* xp = x + (long long)b * x_bstride + (long long)r * ldx;
    bf16_t* yp = y + (long long)b * y_bstride + (long long)r * ldy;
    const int nch = D >> 3;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            unpack8(*(const u32x4_t*)(xp + ch * 8), v[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[c][e];
        }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    const float* sh = shift + (long long)b * mod_bstride;
    const float* sc = scale + (long long)b * mod_bstride;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            float o[8];
            const f32x4_t s0 = *(const f32x4_t*)(sc + ch * 8), s1 = *(const f32x4_t*)(sc + ch * 8 + 4);
            const f32x4_t h0 = *(const f32x4_t*)(sh + ch * 8), h1 = *(const f32x4_t*)(sh + ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[c][e] - mean) * rstd * (1.f + s0[e]) + h0[e];
                o[4 + e] = (v[c][4 + e] - mean) * rstd * (1.f + s1[e]) + h1[e];
            }
            *(u32x4_t*)(yp + ch * 8) = pack8(o);
        }
    }
}

// 8 lanes per (token, head, q|k) unit of 64 elements.  q_off / k_off < 0: that block is absent (the last-block form of
// the engines produces K and Q by separate GEMMs).  Arithmetic: qk_sumsq8 / qk_rstd / qk_rope8 (common.h), shared with the
// QKV GEMM's fused epilogue.
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* qkv, int ld, long long bstride, int q_off, int k_off,
                                                           const float* wq_img, const float* wk_img,
                                                           const float* wq_txt, const float* wk_txt,
                                                           const float* rope, int L, int Lt, int H, int B, float eps, float q_scale, int hstride) {
    const long long unit = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;   // over B*L*H*2
    const int sub = threadIdx.x & 7;
    const long long total = (long long)B * L * H * 2;
    if (unit >= total) return;
    const int which = unit & 1;
    if ((which ? k_off : q_off) < 0) return;
    const long long u2 = unit >> 1;
    const int h = u2 % H;
    const long long tokb = u2 / H;
    const int tok = tokb % L;
    const int b = tokb / L;
    bf16_t* ptr = qkv + (long long)b * bstride + (long long)tok * ld + (which ? k_off : q_off) + h * hstride + sub * 8;
    float v[8];
    unpack8(*(const u32x4_t*)ptr, v);
    float ss = qk_sumsq8(v);
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    const float r = qk_rstd(ss, eps);
    const float* w = (tok < Lt) ? (which ? wk_txt : wq_txt) : (which ? wk_img : wq_img);
    const float* cs = rope + ((long long)tok * 32 + sub * 4) * 2;   // [L][32][cos,sin]
    float w8[8], cs8[8], o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { w8[e] = w[sub * 8 + e]; cs8[e] = cs[e]; }
    qk_rope8(v, r, w8, cs8, which ? 1.f : q_scale, o);
    *(u32x4_t*)ptr = pack8(o);
}

// one wave per output row j; up to 4 batch rows of x. W bf16 [N][K] (row stride ldw), x fp32 [B][K].
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* W, int ldw, const float* bias, const float* x, int ldx,
                                                   float* y, int ldy, int N, int K, int B, int silu_in, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= N) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* wr = W + (long long)j * ldw;
    for (int ch = lane; ch < (K >> 3); ch += 64) {
        float w[8];
        unpack8(*(const u32x4_t*)(wr + ch * 8), w);
        for (int b = 0; b < B; ++b) {
            const f32x4_t x0 = *(const f32x4_t*)(x + (long long)b * ldx + ch * 8);
            const f32x4_t x1 = *(const f32x4_t*)(x + (long long)b * ldx + ch * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = x0[e], a1 = x1[e];
                if (silu_in) { a0 = silu_f(a0); a1 = silu_f(a1); }
                acc[b] += w[e] * a0 + w[4 + e] * a1;
            }
        }
    }
    for (int b = 0; b < B; ++b) {
        const float s = wave_sum(acc[b]);
        if (lane == 0) {
            float r = s + (bias ? bias[j] : 0.f);
            if (accumulate) r += y[(long long)b * ldy + j];
            y[(long long)b * ldy + j] = r;
        }
    }
}

__global__ void timestep_embed_kernel(float* out, int ld, int B, float t0, float t1, float t2, float t3, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= B * half) return;
    const int b = i / half, k = i - b * half;
    const float t = b == 0 ? t0 : (b == 1 ? t1 : (b == 2 ? t2 : t3));
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t * freq;
    out[(long long)b * ld + k] = cosf(a);
    out[(long long)b * ld + half + k] = sinf(a);
}

// latent clip [C, T, H, W] (fp32 or bf16) -> tokens [B copies][(t h w)][(p1 p2 c)] bf16, row stride ld
template <typename TIN>
__global__ void patchify_kernel(const TIN* x, bf16_t* tok, int C, int T, int H, int W, int ld, long long bstride, int ncopies) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int h2 = H / 2, w2 = W / 2;
    const long long total = (long long)T * h2 * w2 * 4 * C;
    if (i >= total) return;
    const int c = i % C;
    long long r = i / C;
    const int p2 = r % 2; r /= 2;
    const int p1 = r % 2; r /= 2;
    const int ww = r % w2; r /= w2;
    const int hh = r % h2;
    const int t = r / h2;
    const float v = (float)x[(((long long)c * T + t) * H + (2 * hh + p1)) * W + (2 * ww + p2)];
    const long long row = ((long long)t * h2 + hh) * w2 + ww;
    const int colr = (p1 * 2 + p2) * C + c;
    for (int b = 0; b < ncopies; ++b) tok[(long long)b * bstride + row * ld + colr] = (bf16_t)v;
}

// v tokens fp32 [2 or 1][n][4C] (row stride ld) -> x[C,1,H,W] fp32 updated in place:
//   v = vu + g (vt - vu);  x += bf16round(dsigma * v) (bf16 rounding of the product as the reference's bf16 path)
__global__ void cfg_euler_kernel(const float* v, long long vb_stride, int ld, float* x, int C, int H, int W,
                                 float guidance, int use_cfg, float dsigma, int round_bf16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)C * H * W;
    if (i >= total) return;
    const int w = i % W;
    const int h = (i / W) % H;
    const int c = i / ((long long)W * H);
    const int h2 = H / 2, w2 = W / 2;
    (void)h2;
    const long long row = (long long)(h / 2) * w2 + (w / 2);
    const int col = ((h & 1) * 2 + (w & 1)) * C + c;
    float val = v[row * ld + col];
    if (use_cfg) {
        const float vt = v[vb_stride + row * ld + col];
        if (round_bf16) {
            // reference bf16 path: model output is bf16, combine evaluated in bf16 op by op
            const float vu = (float)(bf16_t)val, vc = (float)(bf16_t)vt;
            const float d = (float)(bf16_t)(vc - vu);
            const float gd = (float)(bf16_t)(guidance * d);
            val = (float)(bf16_t)(vu + gd);
        } else {
            val = val + guidance * (vt - val);
        }
    } else if (round_bf16) {
        val = (float)(bf16_t)val;
    }
    float dv = dsigma * val;
    if (round_bf16) dv = (float)(bf16_t)dv;
    float xn = x[i] + dv;
    if (round_bf16) xn = (float)(bf16_t)xn;
    x[i] = xn;
}

__global__ void copy_rows_kernel(const bf16_t* src, bf16_t* dst, int rows, int D, int lds_, int ldd, long long sbs,
                                 long long dbs, int B) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = D >> 3;
    const long long total = (long long)B * rows * nch;
    if (i >= total) return;
    const int ch = i % nch;
    const long long rr = i / nch;
    const int r = rr % rows;
    const int b = rr / rows;
    *(u32x4_t*)(dst + (long long)b * dbs + (long long)r * ldd + ch * 8) =
        *(const u32x4_t*)(src + (long long)b * sbs + (long long)r * lds_ + ch * 8);
}

// column-block re-layout between a token-major matrix M[b][r][ld] and the per-peer chunks of a sequence-parallel
// all-to-all:  chunk p = X[off_p + (r*B + b)*cols_p + c]  <->  M[b][r][col0_p + c],  c < cols_p.
struct SpParts { int n; int col0[16]; int cols[16]; long long off[16]; };
__global__ void sp_relayout_kernel(bf16_t* mat, bf16_t* chunks, int rows, int B, int ld, long long mat_bstride,
                                   int total_chunks8, SpParts parts, int to_chunks) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (r, b, 8-column piece of all parts)
    const long long total = (long long)rows * B * total_chunks8;
    if (i >= total) return;
    int ch = i % total_chunks8;
    const long long rb = i / total_chunks8;
    const int b = rb % B;
    const int r = rb / B;
    int p = 0;
    while (ch >= (parts.cols[p] >> 3)) { ch -= parts.cols[p] >> 3; ++p; }
    bf16_t* m = mat + (long long)b * mat_bstride + (long long)r * ld + parts.col0[p] + ch * 8;
    bf16_t* x = chunks + parts.off[p] + ((long long)r * B + b) * parts.cols[p] + ch * 8;
    if (to_chunks) *(u32x4_t*)x = *(const u32x4_t*)m;
    else *(u32x4_t*)m = *(const u32x4_t*)x;
}

// x_out[C,H,W] = alpha * nearest_up2(x_in[C,H/2,W/2]) + beta * noise[C,H,W]   (pipeline.py:729-743)
__global__ void renoise_kernel(const float* xin, const float* noise, float* xout, int C, int H, int W, float alpha,
                               float beta, int round_bf16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)C * H * W;
    if (i >= total) return;
    const int w = i % W;
    const int h = (i / W) % H;
    const int c = i / ((long long)W * H);
    float n = noise[i];
    float xv = xin[((long long)c * (H / 2) + h / 2) * (W / 2) + w / 2];
    float r;
    if (round_bf16) {
        n = (float)(bf16_t)n;
        const float a = (float)(bf16_t)(alpha * xv), bb = (float)(bf16_t)(beta * n);
        r = (float)(bf16_t)(a + bb);
    } else {
        r = alpha * xv + beta * n;
    }
    xout[i] = r;
}

// 2x2 mean over the last two dims (== F.interpolate(bilinear, scale 1/2), pipeline.py:565, 1116), times `mul`
__global__ void avgpool2_kernel(const float* xin, float* xout, long long planes, int H, int W, float mul, int round_bf16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int h2 = H / 2, w2 = W / 2;
    const long long total = planes * h2 * w2;
    if (i >= total) return;
    const int w = i % w2;
    const int h = (i / w2) % h2;
    const long long pl = i / ((long long)w2 * h2);
    const float* s = xin + (pl * H + 2 * h) * W + 2 * w;
    float r = 0.25f * (s[0] + s[1] + s[W] + s[W + 1]) * mul;
    if (round_bf16) r = (float)(bf16_t)r;
    xout[i] = r;
}

}  // namespace

#define CHECK_LAUNCH()                                              \
    do {                                                            \
        hipError_t e_ = hipGetLastError();                          \
        if (e_ != hipSuccess) return pf_set_err(hipGetErrorString(e_)); \
    } while (0)

extern "C" int pf_ln_modulate(const void* x, void* y, const float* shift, const float* scale, int D, int B,
                              int rows_per_batch, long long x_bstride, long long y_bstride, int ldx, int ldy,
                              int mod_bstride, float eps, hipStream_t stream) {
    if (!x || !y || !shift || !scale) return pf_set_err("pf_ln_modulate: null operand");
    if (D % 8 || D > 64 * 8 * 4) return pf_set_err("pf_ln_modulate: D must be a multiple of 8 and <= 2048");
    const int nrows = B * rows_per_batch;
    if (nrows <= 0) return pf_set_err("pf_ln_modulate: empty problem");
    hipLaunchKernelGGL(ln_mod_kernel<4>, dim3((nrows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y,
                       shift, scale, D, rows_per_batch, x_bstride, y_bstride, ldx, ldy, mod_bstride, nrows, eps);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_qk_norm_rope(void* qkv, int ld, long long bstride, int q_off, int k_off, const float* wq_img,
                               const float* wk_img, const float* wq_txt, const float* wk_txt, const float* rope,
                               int B, int L, int Lt, int H, float eps, float q_scale, int head_stride,
                               hipStream_t stream) {
    if (!qkv || !wq_img || !wk_img || !rope) return pf_set_err("pf_qk_norm_rope: null operand");
    if (head_stride <= 0) head_stride = 64;
    if (ld % 8 || (q_off >= 0 && q_off % 8) || (k_off >= 0 && k_off % 8) || head_stride % 8)
        return pf_set_err("pf_qk_norm_rope: misaligned layout");
    const long long threads = (long long)B * L * H * 2 * 8;
    hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       (bf16_t*)qkv, ld, bstride, q_off, k_off, wq_img, wk_img, wq_txt ? wq_txt : wq_img,
                       wk_txt ? wk_txt : wk_img, rope, L, Lt, H, B, eps, q_scale, head_stride);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_gemv_f32(const void* W, int ldw, const float* bias, const float* x, int ldx, float* y, int ldy,
                           int N, int K, int B, int silu_in, int accumulate, hipStream_t stream) {
    if (!W || !x || !y) return pf_set_err("pf_gemv_f32: null operand");
    if (K % 8 || ldw % 8 || ldx % 4 || B < 1 || B > 4) return pf_set_err("pf_gemv_f32: K%8, ldw%8, ldx%4, 1<=B<=4 required");
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const bf16_t*)W, ldw, bias, x, ldx, y,
                       ldy, N, K, B, silu_in, accumulate);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_timestep_embed(float* out, int ld, int B, const float* t_host, int dim, hipStream_t stream) {
    if (!out || !t_host || B < 1 || B > 4 || dim % 2) return pf_set_err("pf_timestep_embed: bad arguments");
    float t[4] = {0, 0, 0, 0};
    for (int i = 0; i < B; ++i) t[i] = t_host[i];
    const int n = B * dim / 2;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, out, ld, B, t[0], t[1], t[2],
                       t[3], dim);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_patchify(const void* x, int x_is_f32, void* tok, int C, int T, int H, int W, int ld, long long bstride,
                           int ncopies, hipStream_t stream) {
    if (!x || !tok || (H & 1) || (W & 1)) return pf_set_err("pf_patchify: bad arguments");
    const long long total = (long long)T * (H / 2) * (W / 2) * 4 * C;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (x_is_f32)
        hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (bf16_t*)tok, C, T,
                           H, W, ld, bstride, ncopies);
    else
        hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)tok, C,
                           T, H, W, ld, bstride, ncopies);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_cfg_euler_step(const float* v, long long vb_stride, int ld, float* x, int C, int H, int W,
                                 float guidance, int use_cfg, float dsigma, int round_bf16, hipStream_t stream) {
    if (!v || !x) return pf_set_err("pf_cfg_euler_step: null operand");
    const long long total = (long long)C * H * W;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, v, vb_stride, ld, x,
                       C, H, W, guidance, use_cfg, dsigma, round_bf16);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_copy_rows(const void* src, void* dst, int rows, int D, int ld_src, int ld_dst, long long src_bstride,
                            long long dst_bstride, int B, hipStream_t stream) {
    if (!src || !dst || D % 8) return pf_set_err("pf_copy_rows: bad arguments");
    const long long total = (long long)B * rows * (D / 8);
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)src,
                       (bf16_t*)dst, rows, D, ld_src, ld_dst, src_bstride, dst_bstride, B);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_sp_relayout(void* mat, void* chunks, int rows, int B, int ld, long long mat_bstride, int n_parts,
                              const int* col0, const int* cols, const long long* off, int to_chunks, hipStream_t stream) {
    if (!mat || !chunks || !col0 || !cols || !off) return pf_set_err("pf_sp_relayout: null operand");
    if (n_parts < 1 || n_parts > 16) return pf_set_err("pf_sp_relayout: 1..16 parts");
    SpParts parts{};
    parts.n = n_parts;
    int total8 = 0;
    for (int p = 0; p < n_parts; ++p) {
        if ((col0[p] % 8) || (cols[p] % 8) || (off[p] % 8)) return pf_set_err("pf_sp_relayout: columns / offsets must be multiples of 8");
        parts.col0[p] = col0[p]; parts.cols[p] = cols[p]; parts.off[p] = off[p];
        total8 += cols[p] / 8;
    }
    if (ld % 8) return pf_set_err("pf_sp_relayout: ld must be a multiple of 8");
    const long long total = (long long)rows * B * total8;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sp_relayout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (bf16_t*)mat,
                       (bf16_t*)chunks, rows, B, ld, mat_bstride, total8, parts, to_chunks);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_renoise_upsample(const float* xin, const float* noise, float* xout, int C, int H, int W, float alpha,
                                   float beta, int round_bf16, hipStream_t stream) {
    if (!xin || !noise || !xout || (H & 1) || (W & 1)) return pf_set_err("pf_renoise_upsample: bad arguments");
    const long long total = (long long)C * H * W;
    hipLaunchKernelGGL(renoise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xin, noise, xout, C, H,
                       W, alpha, beta, round_bf16);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_avgpool2(const float* xin, float* xout, long long planes, int H, int W, float mul, int round_bf16,
                           hipStream_t stream) {
    if (!xin || !xout || (H & 1) || (W & 1)) return pf_set_err("pf_avgpool2: bad arguments");
    const long long total = planes * (H / 2) * (W / 2);
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, xin, xout, planes, H,
                       W, mul, round_bf16);
    CHECK_LAUNCH();
    return 0;
}
