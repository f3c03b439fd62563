// HBM-bound kernels of the CausalVideoVAE decode path (channels-last bf16 activations, spatially
// zero-padded frames with two leading temporal cache slots -- see DESIGN.md "VAE data layout"):
//   pf_gn_stats / pf_gn_apply   CausalGroupNorm (per-frame GroupNorm, modeling_causal_conv.py:36-43) + SiLU
//                                (modeling_resnet.py:127-141), two passes: per-(frame,channel) sums in fp64
//                                atomics, then normalise+affine+SiLU into the padded input image of the next conv
//   pf_softmax_rows              row softmax of the mid-block attention scores (diffusers Attention, upcast softmax)
//   pf_latent_to_nhwc            latent [C,T,h,w] fp32 -> padded channels-last bf16 (decode_latent's un-normalise is
//                                folded in by the host)
//   pf_blend_v / pf_blend_h      tile cross-fade (modeling_causal_vae.py:397-407), in place, sequential tile order
//   pf_to_uint8                  x*127.5+127.5, clamp, byte, crop+place into the final [T,H,W,3] frame store
//                                (pipeline.py:1238-1239, causal_vae.py:511-515)
#include "common.h"
#include "pyflow_hip.h"

int pf_set_err(const char* m);

namespace {

// x: frames of [Hp][Wp][Cp] bf16 (interior H x W at offset (1,1) if padded, else Hp=H, Wp=W, off 0)
// stats: double [T][C][2] (sum, sumsq), pre-zeroed by the caller.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* x, double* stats, int C, int Cp, int H, int W, int Hp,
                                                       int Wp, long long frame_stride, long long base_off, int pix_per_block) {
    __shared__ float red[256][17];
    const int f = blockIdx.y;
    const int nch = C >> 3;                       // 8-channel chunks per pixel
    const int lanes = 256 / nch;                  // pixel lanes per block (nch is a power-of-two divisor of 64 or <= 64)
    const int chunk = threadIdx.x % nch, pl = threadIdx.x / nch;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, H * W);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (pl < lanes) {
        const bf16_t* xf = x + base_off + (long long)f * frame_stride;
        int p = p0 + pl;
        // four independent 16-byte loads in flight per thread
        for (; p + 3 * lanes < p1; p += 4 * lanes) {
            u32x4_t raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pp = p + u * lanes;
                const int h = pp / W, w = pp - h * W;
                raw[u] = *(const u32x4_t*)(xf + ((long long)h * Wp + w) * Cp + chunk * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v[8];
                unpack8(raw[u], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
            }
        }
        for (; p < p1; p += lanes) {
            const int h = p / W, w = p - h * W;
            float v[8];
            unpack8(*(const u32x4_t*)(xf + ((long long)h * Wp + w) * Cp + chunk * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][8 + e] = q[e]; }
    __syncthreads();
    // thread t < nch*16 reduces (chunk = t/16, slot = t%16) over the pixel lanes
    for (int t = threadIdx.x; t < nch * 16; t += 256) {
        const int ck = t >> 4, slot = t & 15;
        float acc = 0.f;
        for (int l = 0; l < lanes; ++l) acc += red[l * nch + ck][slot];
        const int c = ck * 8 + (slot & 7);
        atomicAdd(&stats[((long long)f * C + c) * 2 + (slot >> 3)], (double)acc);
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x, bf16_t* y, const double* stats, const float* gamma,
                                                       const float* beta, int C, int Cp_in, int Cp_out, int G, int H, int W,
                                                       int Hp_in, int Wp_in, long long fs_in, long long off_in, int Hp_out,
                                                       int Wp_out, long long fs_out, long long off_out, float eps, int silu,
                                                       int pix_per_block) {
    __shared__ float gmean[64], grstd[64];
    const int f = blockIdx.y;
    const int cpg = C / G;
    if (threadIdx.x < G) {
        double su = 0, sq = 0;
        for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
            su += stats[((long long)f * C + c) * 2];
            sq += stats[((long long)f * C + c) * 2 + 1];
        }
        const double n = (double)cpg * H * W;
        const double mean = su / n;
        double var = sq / n - mean * mean;
        var = var > 0 ? var : 0;
        gmean[threadIdx.x] = (float)mean;
        grstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int nch = C >> 3;
    const int lanes = 256 / nch;
    const int chunk = threadIdx.x % nch, pl = threadIdx.x / nch;
    if (pl >= lanes) return;
    float ga[8], be[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 8 + e;
        ga[e] = gamma[c]; be[e] = beta[c];
        mu[e] = gmean[c / cpg]; rs[e] = grstd[c / cpg];
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, H * W);
    const bf16_t* xf = x + off_in + (long long)f * fs_in;
    bf16_t* yf = y + off_out + (long long)f * fs_out;
    int p = p0 + pl;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {
        u32x4_t raw[4];
        int hh[4], ww[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pp = p + u * lanes;
            hh[u] = pp / W; ww[u] = pp - hh[u] * W;
            raw[u] = *(const u32x4_t*)(xf + ((long long)hh[u] * Wp_in + ww[u]) * Cp_in + chunk * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v[8];
            unpack8(raw[u], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
                v[e] = silu ? silu_f(t) : t;
            }
            *(u32x4_t*)(yf + ((long long)hh[u] * Wp_out + ww[u]) * Cp_out + chunk * 8) = pack8(v);
        }
    }
    for (; p < p1; p += lanes) {
        const int h = p / W, w = p - h * W;
        float v[8];
        unpack8(*(const u32x4_t*)(xf + ((long long)h * Wp_in + w) * Cp_in + chunk * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
            v[e] = silu ? silu_f(t) : t;
        }
        *(u32x4_t*)(yf + ((long long)h * Wp_out + w) * Cp_out + chunk * 8) = pack8(v);
    }
}

// one wave per row: S[row][0:n_valid] bf16 * scale -> softmax (fp32) -> bf16, columns >= n_valid get 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* S, int ld, int n_valid, int n_cols, int rows, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* r = S + (long long)row * ld;
    float m = -1e30f;
    for (int c = lane * 8; c < n_valid; c += 512) {
        float v[8];
        unpack8(*(const u32x4_t*)(r + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < n_valid) m = fmaxf(m, v[e] * scale);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int c = lane * 8; c < n_valid; c += 512) {
        float v[8];
        unpack8(*(const u32x4_t*)(r + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) if (c + e < n_valid) sum += __expf(v[e] * scale - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int c = lane * 8; c < n_cols; c += 512) {
        float v[8];
        unpack8(*(const u32x4_t*)(r + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c + e < n_valid) ? __expf(v[e] * scale - m) * inv : 0.f;
        *(u32x4_t*)(r + c) = pack8(v);
    }
}

// z [C][T][H][W] fp32 (affine a*z+b per frame class folded by caller) -> y frames [Hp][Wp][Cp] bf16 interior
__global__ void latent_to_nhwc_kernel(const float* z, bf16_t* y, int C, int T, int H, int W, int t0, int nt, int h0, int w0,
                                      int th, int tw, int Cp, int Hp, int Wp, long long fs_out, long long off_out,
                                      float a0, float b0, float a1, float b1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)nt * th * tw * C;
    if (i >= total) return;
    const int c = i % C;
    long long r = i / C;
    const int w = r % tw; r /= tw;
    const int h = r % th;
    const int t = r / th;
    const int tg = t0 + t;
    float v = z[(((long long)c * T + tg) * H + (h0 + h)) * W + (w0 + w)];
    v = tg == 0 ? v * a0 + b0 : v * a1 + b1;
    y[off_out + (long long)t * fs_out + ((long long)h * Wp + w) * Cp + c] = (bf16_t)v;
}

// b[:, y, :] = a[:, Ha - e + y, :] * (1 - y/e) + b[:, y, :] * (y/e)  for y < e   (tiles: [T][H][W][Cp] bf16)
__global__ void blend_kernel(const bf16_t* a, bf16_t* b, int T, int Ha, int Wa, int Hb, int Wb, int Cp, int e, int vertical) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int span = vertical ? min(Wa, Wb) : min(Ha, Hb);
    const long long total = (long long)T * e * span * Cp;
    if (i >= total) return;
    const int c = i % Cp;
    long long r = i / Cp;
    const int s = r % span; r /= span;
    const int k = r % e;
    const int t = r / e;
    long long ia, ib;
    if (vertical) {
        ia = (((long long)t * Ha + (Ha - e + k)) * Wa + s) * Cp + c;
        ib = (((long long)t * Hb + k) * Wb + s) * Cp + c;
    } else {
        ia = (((long long)t * Ha + s) * Wa + (Wa - e + k)) * Cp + c;
        ib = (((long long)t * Hb + s) * Wb + k) * Cp + c;
    }
    const float wgt = (float)k / (float)e;
    // reference evaluates in the tensor dtype (bf16) op by op
    const float av = (float)(bf16_t)((float)a[ia] * (float)(bf16_t)(1.f - wgt));
    const float bv = (float)(bf16_t)((float)b[ib] * (float)(bf16_t)wgt);
    b[ib] = (bf16_t)(av + bv);
}

__global__ void nhwc_to_planar_kernel(const bf16_t* tile, float* out, int T, int Ht, int Wt, int Cp, int C, int ch, int cw,
                                      int H, int W, int y0, int x0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)T * ch * cw * C;
    if (i >= total) return;
    const int c = i % C;
    long long r = i / C;
    const int x = r % cw; r /= cw;
    const int y = r % ch;
    const int t = r / ch;
    out[(((long long)c * T + t) * H + (y0 + y)) * W + (x0 + x)] = (float)tile[(((long long)t * Ht + y) * Wt + x) * Cp + c];
}

__global__ void to_uint8_kernel(const bf16_t* tile, unsigned char* out, int T, int Ht, int Wt, int Cp, int ch, int cw, int H,
                                int W, int y0, int x0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)T * ch * cw;
    if (i >= total) return;
    const int x = i % cw;
    const int y = (i / cw) % ch;
    const int t = i / ((long long)cw * ch);
    const bf16_t* p = tile + (((long long)t * Ht + y) * Wt + x) * Cp;
    unsigned char* o = out + (((long long)t * H + (y0 + y)) * W + (x0 + x)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // image.mul(127.5).add(127.5).clamp(0,255).byte() on a bf16 tensor
        float v = (float)(bf16_t)((float)p[c] * 127.5f);
        v = (float)(bf16_t)(v + 127.5f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        o[c] = (unsigned char)v;
    }
}

// frames [T][H][W][3] uint8 RGB -> planar Y [T][H][W], Cb / Cr [T][H/2][W/2] (JFIF full-range BT.601, 16-bit fixed
// point, chroma from the 2x2 block sum): one thread per 2x2 block.  Byte work, HBM-bound (4.5 B moved per pixel).
__global__ __launch_bounds__(256) void rgb_to_yuv420_kernel(const unsigned char* rgb, unsigned char* yp, unsigned char* up,
                                                            unsigned char* vp, int T, int H, int W, long long y_fs,
                                                            long long c_fs) {
    const int hw2 = (H >> 1) * (W >> 1);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * hw2) return;
    const int t = (int)(i / hw2), r = (int)(i - (long long)t * hw2);
    const int by = r / (W >> 1), bx = r - by * (W >> 1);
    const unsigned char* f = rgb + (long long)t * H * W * 3;
    unsigned char* yf = yp + (long long)t * y_fs;
    int sr = 0, sg = 0, sb = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int y = 2 * by + dy;
        // 6 contiguous bytes = two pixels
        const unsigned char* px = f + ((long long)y * W + 2 * bx) * 3;
        unsigned char yo[2];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int R = px[3 * dx], G = px[3 * dx + 1], B = px[3 * dx + 2];
            sr += R; sg += G; sb += B;
            yo[dx] = (unsigned char)((19595 * R + 38470 * G + 7471 * B + 32768) >> 16);
        }
        *(unsigned short*)(yf + (long long)y * W + 2 * bx) = (unsigned short)(yo[0] | (yo[1] << 8));
    }
    const int cb = (-11059 * sr - 21709 * sg + 32768 * sb + (128 << 18) + (1 << 17)) >> 18;
    const int cr = (32768 * sr - 27439 * sg - 5329 * sb + (128 << 18) + (1 << 17)) >> 18;
    up[(long long)t * c_fs + r] = (unsigned char)min(max(cb, 0), 255);
    vp[(long long)t * c_fs + r] = (unsigned char)min(max(cr, 0), 255);
}

// All cache-slot updates of one chunk in ONE launch (cache_front_feat update of every CausalConv3d,
// modeling_causal_conv.py:132,143): buffer b: slots[0:2] <- frames [n, n+2) of slots[0 : 2+n)  (n >= 2), or
// slot0 <- slot1, slot1 <- slot2 (n == 1), or both slots <- 0 (n == 0: a new clip starts, causal_conv.py:128-131).  blockIdx.y = buffer, 16-byte pieces over the 2 frames.
struct ShiftList { int count; unsigned long long ptr[64]; long long fs[64]; int n[64]; };
__global__ __launch_bounds__(256) void shift_caches_kernel(const ShiftList L) {
    const int b = blockIdx.y;
    if (b >= L.count) return;
    u32x4_t* base = (u32x4_t*)L.ptr[b];
    const long long fs8 = L.fs[b] >> 3;            // 16-byte pieces per frame
    const int n = L.n[b];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < fs8; i += (long long)gridDim.x * blockDim.x) {
        if (n == 0) {                              // reset: the causal zero padding of a clip's first chunk
            base[i] = (u32x4_t){0u, 0u, 0u, 0u};
            base[fs8 + i] = (u32x4_t){0u, 0u, 0u, 0u};
        } else if (n >= 2) {
            base[i] = base[(long long)n * fs8 + i];
            base[fs8 + i] = base[(long long)(n + 1) * fs8 + i];
        } else {
            const u32x4_t v1 = base[fs8 + i], v2 = base[2 * fs8 + i];
            base[i] = v1;
            base[fs8 + i] = v2;
        }
    }
}

}  // namespace

#define CHECK_LAUNCH()                                                  \
    do {                                                                \
        hipError_t e_ = hipGetLastError();                              \
        if (e_ != hipSuccess) return pf_set_err(hipGetErrorString(e_)); \
    } while (0)

// pixels per block: aim at >= ~2048 workgroups per launch (256 CUs x 8), at least 4 pixels per thread-lane
static int pix_per_block_for(int HW, int T, int C) {
    const int lanes = 256 / (C / 8);
    long long want = ((long long)HW * T + 2047) / 2048;
    const int unit = 4 * lanes;
    long long ppb = (want + unit - 1) / unit * unit;
    if (ppb < unit) ppb = unit;
    if (ppb > HW) ppb = (HW + unit - 1) / unit * unit;
    return (int)ppb;
}

extern "C" int pf_gn_stats(const void* x, double* stats, int T, int C, int Cp, int H, int W, int Hp, int Wp,
                           long long frame_stride, long long base_off, hipStream_t stream) {
    if (!x || !stats) return pf_set_err("pf_gn_stats: null operand");
    const int nch = C / 8;
    if (C % 8 || nch > 64 || (256 % nch)) return pf_set_err("pf_gn_stats: C/8 must divide 256 and C <= 512");
    const int ppb = pix_per_block_for(H * W, T, C);
    hipLaunchKernelGGL(gn_stats_kernel, dim3((H * W + ppb - 1) / ppb, T), dim3(256), 0, stream, (const bf16_t*)x, stats, C, Cp,
                       H, W, Hp, Wp, frame_stride, base_off, ppb);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_gn_apply(const void* x, void* y, const double* stats, const float* gamma, const float* beta, int T, int C,
                           int G, int H, int W, int Cp_in, int Hp_in, int Wp_in, long long fs_in, long long off_in,
                           int Cp_out, int Hp_out, int Wp_out, long long fs_out, long long off_out, float eps, int silu,
                           hipStream_t stream) {
    if (!x || !y || !stats || !gamma || !beta) return pf_set_err("pf_gn_apply: null operand");
    const int nch = C / 8;
    if (C % 8 || nch > 64 || (256 % nch) || G > 64 || C % G) return pf_set_err("pf_gn_apply: unsupported C/G");
    const int ppb = pix_per_block_for(H * W, T, C);
    hipLaunchKernelGGL(gn_apply_kernel, dim3((H * W + ppb - 1) / ppb, T), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y,
                       stats, gamma, beta, C, Cp_in, Cp_out, G, H, W, Hp_in, Wp_in, fs_in, off_in, Hp_out, Wp_out, fs_out,
                       off_out, eps, silu, ppb);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_softmax_rows(void* S, int ld, int n_valid, int n_cols, int rows, float scale, hipStream_t stream) {
    if (!S || ld % 8 || n_cols % 8 || n_valid > n_cols) return pf_set_err("pf_softmax_rows: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (bf16_t*)S, ld, n_valid, n_cols, rows,
                       scale);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_latent_to_nhwc(const float* z, void* y, int C, int T, int H, int W, int t0, int nt, int h0, int w0, int th,
                                 int tw, int Cp, int Hp, int Wp, long long fs_out, long long off_out, float a0, float b0,
                                 float a1, float b1, hipStream_t stream) {
    if (!z || !y) return pf_set_err("pf_latent_to_nhwc: null operand");
    const long long total = (long long)nt * th * tw * C;
    hipLaunchKernelGGL(latent_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, (bf16_t*)y, C, T,
                       H, W, t0, nt, h0, w0, th, tw, Cp, Hp, Wp, fs_out, off_out, a0, b0, a1, b1);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_blend_tiles(const void* a, void* b, int T, int Ha, int Wa, int Hb, int Wb, int Cp, int extent, int vertical,
                              hipStream_t stream) {
    if (!a || !b) return pf_set_err("pf_blend_tiles: null operand");
    const int e = vertical ? (extent < Ha ? (extent < Hb ? extent : Hb) : (Ha < Hb ? Ha : Hb))
                           : (extent < Wa ? (extent < Wb ? extent : Wb) : (Wa < Wb ? Wa : Wb));
    const int span = vertical ? (Wa < Wb ? Wa : Wb) : (Ha < Hb ? Ha : Hb);
    const long long total = (long long)T * e * span * Cp;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)a, (bf16_t*)b,
                       T, Ha, Wa, Hb, Wb, Cp, e, vertical);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_nhwc_to_planar_f32(const void* tile, float* out, int T, int Ht, int Wt, int Cp, int C, int crop_h,
                                     int crop_w, int H, int W, int y0, int x0, hipStream_t stream) {
    if (!tile || !out) return pf_set_err("pf_nhwc_to_planar_f32: null operand");
    const long long total = (long long)T * crop_h * crop_w * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(nhwc_to_planar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t*)tile, out, T, Ht, Wt, Cp, C, crop_h, crop_w, H, W, y0, x0);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_to_uint8(const void* tile, void* out, int T, int Ht, int Wt, int Cp, int crop_h, int crop_w, int H, int W,
                           int y0, int x0, hipStream_t stream) {
    if (!tile || !out) return pf_set_err("pf_to_uint8: null operand");
    const long long total = (long long)T * crop_h * crop_w;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)tile,
                       (unsigned char*)out, T, Ht, Wt, Cp, crop_h, crop_w, H, W, y0, x0);
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_rgb_to_yuv420(const void* rgb, void* y, void* u, void* v, int T, int H, int W, long long y_frame_stride,
                                long long c_frame_stride, hipStream_t stream) {
    if (!rgb || !y || !u || !v) return pf_set_err("pf_rgb_to_yuv420: null operand");
    if (T <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return pf_set_err("pf_rgb_to_yuv420: H and W must be even");
    const long long total = (long long)T * (H / 2) * (W / 2);
    hipLaunchKernelGGL(rgb_to_yuv420_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const unsigned char*)rgb, (unsigned char*)y, (unsigned char*)u, (unsigned char*)v, T, H, W,
                       y_frame_stride ? y_frame_stride : (long long)H * W,
                       c_frame_stride ? c_frame_stride : (long long)(H / 2) * (W / 2));
    CHECK_LAUNCH();
    return 0;
}

extern "C" int pf_shift_caches(int count, const void* const* bufs, const long long* frame_elems, const int* n_frames,
                               hipStream_t stream) {
    if (count <= 0) return 0;
    if (count > 64 || !bufs || !frame_elems || !n_frames) return pf_set_err("pf_shift_caches: 1..64 buffers");
    ShiftList L;
    L.count = count;
    long long fs_max = 0;
    for (int i = 0; i < count; ++i) {
        if (!bufs[i] || frame_elems[i] % 8 || n_frames[i] < 0) return pf_set_err("pf_shift_caches: bad entry");
        L.ptr[i] = (unsigned long long)bufs[i];
        L.fs[i] = frame_elems[i];
        L.n[i] = n_frames[i];
        fs_max = frame_elems[i] > fs_max ? frame_elems[i] : fs_max;
    }
    long long blocks = (fs_max / 8 + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(shift_caches_kernel, dim3((unsigned)blocks, count), dim3(256), 0, stream, L);
    CHECK_LAUNCH();
    return 0;
}
