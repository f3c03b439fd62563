/* pyflow_hip.h -- C ABI of libpyflow_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary of the MI355X-native pyramidal flow-matching sampler.  The reference
 * (jy0205/Pyramid-Flow) has NO native/FFI layer: its operators are torch calls inside Python
 * classes.  Each entry point below therefore replaces a *call site* of the reference, cited as
 * file:line relative to the reference root.  The Python host (pyramid-flow_amd/) binds these with
 * ctypes and keeps the reference's class/method signatures on top.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless said otherwise;
 *   - every call is asynchronous on the hipStream_t passed in, performs no allocation and no
 *     synchronisation; the caller owns all memory;
 *   - return 0 on success, negative on error; pf_last_error() returns a thread-local message;
 *   - bf16 tensors are raw uint16 storage; "f32" pointers are float.
 */
#ifndef PYFLOW_HIP_H
#define PYFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* pf_stream_t; /* == hipStream_t */

const char* pf_last_error(void);
/* ABI version of THIS header.  pf_version() returns the version the library was built with; a caller compares the two
 * before its first launch (a descriptor struct that grew -- 2 -> 3: pf_attn_desc.workspace / workspace_bytes,
 * pf_conv_desc.gn_stats / gn_C; 3 -> 4: pf_gemm_desc.qk_*; 4 -> 5: pf_gemm_desc.qk_head_stride; 5 -> 6: the second
 * problem of a grouped GEMM launch -- would otherwise be read past its end; 6 -> 7: no struct grew, pf_shift_caches accepts
 * n_frames = 0). */
#define PF_ABI_VERSION 7
int pf_version(void);
/* sizeof() of the descriptor structs as this library was compiled: 0 pf_gemm_desc, 1 pf_conv_desc, 2 pf_attn_desc,
 * 3 pf_attn_small_desc (-1 otherwise) -- lets a foreign-language binding (ctypes / cgo / JNI struct mirrors) verify its
 * layout at load time instead of corrupting a launch. */
int pf_struct_size(int which);

/* ------------------------------------------------------------------ GEMM (nn.Linear) ------------
 * C[b] = epi( A[b] (M x K, row stride lda) . W^T (W is N x K, row stride ldw, nn.Linear layout) )
 * Replaces: attn.to_q/to_k/to_v/add_*_proj (flux_block.py:756-758, 816-835), to_out/to_add_out
 * (:868-872), FeedForward (:42-100, GELU-tanh), proj_mlp/proj_out of the single blocks (:921-937),
 * x_embedder / context_embedder / proj_out (modeling_pyramid_flux.py:401, 290, 539).
 * flags: PF_GEMM_GATE_RES   C = res + gate[b, n] * (acc + bias[n])   (gate NULL -> 1)
 *        PF_GEMM_OUT_F32    C is float instead of bf16
 * gelu_from: columns n >= gelu_from get GELU(tanh) after the bias (-1 = none).
 * Constraints: N % 128 == 0, K % 64 == 0, lda/ldw/ldc % 8 == 0.  M arbitrary. */
#define PF_GEMM_GATE_RES 1
#define PF_GEMM_OUT_F32 2
#define PF_GEMM_ACT_QUICK_GELU 4 /* the gelu_from columns get x*sigmoid(1.702x) instead (CLIP-L MLP, transformers activations.py) */
#define PF_GEMM_ACT_GELU_ERF 8   /* ... or the exact erf GELU (CLIP-G MLP) */
typedef struct {
    const void* A; const void* W; void* C;
    const float* bias;      /* [N] or NULL */
    const void* res;        /* bf16, same indexing as C with ldr / strideR */
    const float* gate;      /* [batch, gate_stride] */
    int M, N, K, lda, ldw, ldc, ldr;
    long long strideA, strideC, strideR;
    int gate_stride, batch, gelu_from, flags;
    /* optional scratch for SKINNY problems (fewer than 128 tiles of 128 x 128, e.g. the 128-token text stream of the
     * double blocks, flux_block.py:816-835 / 42-100, and the prompt encoders): with it the K range is split over
     * ceil(256 / tiles) workgroups per tile (fp32 partial sums [split][batch][M][N] in the scratch, summed in split order
     * by a second launch that applies the epilogue), which turns one workgroup's chain of K / 64 dependent memory
     * latencies into several short ones.  NULL / 0 = never split.  Must not be shared by launches that may overlap.
     * LARGE problems (the persistent 256 x 256 kernel, pf_gemm_which == 8) use the same scratch, when it holds 64 MiB
     * (pf_gemm_workspace_bytes), to split the tiles that do not fill a last round of the 256 workgroups along K: the
     * parts park raw fp32 sums (one 256-KiB slot per workgroup), a second small launch adds them in part order and
     * applies the epilogue -- the DiT's N = 1920, K = 7680 / 9600 projections otherwise run e.g. 1.25 "rounds" as 2.
     * Bitwise repeatable; differs from the scratch-less result in fp32 summation order only. */
    void* workspace;
    long long workspace_bytes;
    /* ABI 4 -- optional QK-RMSNorm + RoPE of the projection's K and Q column blocks, applied to the bf16 result exactly as
     * pf_qk_norm_rope would in a separate pass (RMSNorm over each 64-wide head with gains qk_wk / qk_wq, eps qk_eps:
     * modeling_normalization.py:66-79; adjacent-pair rotation by qk_rope, flux_block.py:34-39; Q additionally * qk_q_scale):
     * the K block is columns [qk_k_col0, qk_k_col0 + qk_d), the Q block [qk_q_col0, qk_q_col0 + qk_d) of C (a start < 0 =
     * that block is not part of this GEMM); row r of every batch entry uses table row qk_row0 + r of qk_rope
     * ([rows][32][cos, sin] fp32).  qk_d = 0: none.  Where the persistent 256 x 256 kernel runs, this happens in its
     * epilogue from the accumulators (no second pass over [rows][3 d]: flux_block.py:846-858 costs two extra HBM passes
     * per block otherwise); for any other kernel choice pf_gemm_bf16 launches the separate pass itself.  Same bits either
     * way.  Requires bf16 output, no residual, qk_d / column starts multiples of 64. */
    const float* qk_rope; const float* qk_wq; const float* qk_wk;
    int qk_d, qk_q_col0, qk_k_col0, qk_row0;
    float qk_eps, qk_q_scale;
    /* ABI 5 -- HEAD-MAJOR column layout of the fused projection (the sequence-parallel engine's [head][k | v | q][64], so that
     * "the heads of rank p" is one contiguous column block of the exchange, modeling_flux_block.py:266-325): qk_head_stride > 0
     * = columns per head (a multiple of 64); head h = columns [h qk_head_stride, (h + 1) qk_head_stride) for h < qk_d / 64,
     * its K block at column qk_k_col0 and its Q block at qk_q_col0 INSIDE the head (multiples of 64; < 0 = absent).  With it
     * the rank's K / Q leave the projection normed and rotated BEFORE the exchange (RMSNorm and RoPE are per row and per head:
     * row r uses table row qk_row0 + r = the row's global index), and the received matrix needs no pass of its own.
     * 0 = the two contiguous column blocks described above. */
    int qk_head_stride;
    /* ABI 6 -- GROUPED launch: a SECOND problem C2 = epi(A2 . W2^T) with the same N, K, lda / ldw / ldc / ldr, batch count,
     * flags, gelu_from, gate_stride and QK layout, its own operands, row count M2 (> 0 = present) and batch strides, and -- with
     * a QK epilogue -- its own gains and first RoPE row.  The double-stream blocks run every projection twice, once per
     * stream, with different weights on different rows (flux_block.py:816-835 to_q/k/v vs add_q/k/v_proj, :868-872 to_out vs
     * to_add_out, :1022-1036 ff vs ff_context; text = 128 rows per prompt): where the persistent 256 x 256 kernel serves the
     * first problem, the second one's tiles are appended to the SAME launch's tile list (no second launch, no side stream,
     * no K-split scratch for the skinny problem); any other kernel choice runs the two problems as two launches.  On whole
     * tiles the result of either problem does not depend on the grouping (a tile is computed by one workgroup in K order
     * either way); with the tail split (workspace) the launch's tile count decides WHICH tiles are split, so grouped and
     * separate results differ in fp32 summation order there, as split and unsplit results do. */
    const void* A2; const void* W2; void* C2;
    const float* bias2; const void* res2; const float* gate2;
    int M2;
    long long strideA2, strideC2, strideR2;
    const float* qk_wq2; const float* qk_wk2;
    int qk_row0_2;
} pf_gemm_desc;
int pf_gemm_bf16(const pf_gemm_desc* d, pf_stream_t stream);
/* bytes of pf_gemm_desc.workspace this problem can use (0 = it never splits) */
long long pf_gemm_workspace_bytes(int M, int batch, int N, int K);
/* Kernel selection for pf_gemm_bf16 / pf_conv3d_bf16 (test hook; results are identical up to fp32 summation order).
 * 0 = automatic: the persistent 256 x 256 kernel (gemm8p) for problems of >= 192 such tiles whose N tail wastes < 7 %,
 * else the 256 x BN ping-pong kernel (BN = 256 / 192 / 128) when it yields >= 192 tiles, else the 128 x 128 kernel;
 * -1 = always the 128 x 128 kernel; 128 / 192 / 256 = force the 256 x BN kernel whenever BN divides N;
 * 8 = force gemm8p whenever its epilogue flavour exists (bias + ONE of residual / fp32 output / GELU-tanh); -8 = never
 * gemm8p; -2 / 2 = never / again split K for skinny problems that bring a workspace; -3 / 3 = never / again
 * the narrow-N conv kernel (3x3x3 convs with <= 8 output channels: the decoder's conv_out); -4 / 4 = never / again
 * split the tail tiles of gemm8p problems that bring a workspace (also: the whole-launch K split of 32 .. 128-tile problems);
 * -5 / 5 = never / again the LDS-halo direct conv; -6 / 6 = halo conv only for N = 128 / also for wider layers; -7 / 7 = the
 * upsamplers' output maps stay with the implicit GEMM / take the halo kernel;
 * 2000 + R (R = 0 .. 128, rounded up to a multiple of 8) = the persistent kernel launches CUs - R workgroups and leaves R CUs
 * (R / 8 per XCD) to kernels that must run BESIDE it -- the RCCL send / recv kernels of a sequence-parallel exchange
 * (trainer_misc/communicate.py:7-26): a gemm8p workgroup owns its CU whole, so without a reservation an exchange in flight
 * delays the launch by its own duration and an exchange queued behind the launch waits for its end
 * (profiles/r05_comm_overlap_bench.log).  NOT reset by force = 0: the owner of the communicator sets and clears it.  Default 0.
 * A reservation that would leave the persistent kernel fewer than 64 CUs is an error (nothing changes).
 * Measurement-only switches (desynchronised start, the tail split's cost constant, the epilogue schedule, the 128 x 128
 * kernel's stage count / split cap) are NOT part of this library: they exist only in the lab build
 * (`make -C pyramid-flow_amd/csrc lab` -> libpyflow_hip_lab.so, -DPF_LAB_HOOKS), which tools/ and lab/ load explicitly. */
int pf_gemm_set_policy(int force);
/* workgroups of a persistent-kernel launch under the current reservation (CUs - R) */
int pf_gemm_workgroups(void);
/* which kernel pf_gemm_bf16 runs for (M rows per batch entry, batch, N, K) when the caller brings the scratch
 * pf_gemm_workspace_bytes asks for and no QK epilogue: 0 = gemm_kernel (128x128), 8 = gemm8p_kernel, BN > 0 =
 * gemm256_kernel<BN> */
int pf_gemm_which(int M, int batch, int N, int K);
/* ... and for ONE descriptor exactly as pf_gemm_bf16 decides (a mid-size problem takes the persistent kernel only with
 * enough scratch and without a QK epilogue; epilogue flavours gemm8p has no instantiation of go to the older kernels)
 * -- lets a profiler attribute launches to the kernel names rocprofv3 reports.  -100 = invalid descriptor. */
int pf_gemm_which_desc(const pf_gemm_desc* d);

/* ------------------------------------------------------------------ CausalConv3d ----------------
 * Implicit-GEMM convolution over a channels-last, zero-padded input (replaces CausalConv3d.forward,
 * video_vae/modeling_causal_conv.py:116-146, incl. the chunk cache: the caller keeps the two
 * previous frames in the first two temporal slots of X).  Rows = output pixels (t,h,w), K =
 * kt*kh*kw*Cin, N = filters (padded to 128).  Input element for pixel (t,h,w), tap (dt,dh,dw),
 * channel c:  X[in_base_off + (((t+dt)*Hp + h+dh)*Wp + w+dw)*Cin + c].
 * Output column n = g*Cg + c (g = (pt*sh + ph)*sw + pw: depth-to-time / pixel-shuffle group, the
 * host permutes filter rows accordingly; modeling_resnet.py:609-617, 716-729) is stored at
 * Y[out_base_off + (((t*st+pt)*Hop + h*sh+ph)*Wop + w*sw+pw)*Cout_pitch + c]; columns >= n_valid
 * are dropped.  flags PF_GEMM_GATE_RES adds res at the same output offset (resnet shortcut add,
 * modeling_resnet.py:148); out_scale multiplies the result (1/output_scale_factor). */
typedef struct {
    const void* X; const void* W; void* Y; const float* bias; const void* res;
    int T, H, W_;            /* output pixel grid */
    int Hp, Wp, Cin, kt, kh, kw;
    long long in_base_off;
    int N, n_valid;
    int st, sh, sw, Cg, Hop, Wop, Cout_pitch;
    long long out_base_off;
    int flags; float out_scale;
    int out_t_shift;         /* added to the output frame index, frames < 0 dropped (is_init_image drop, modeling_resnet.py:726) */
    int in_sh, in_sw;        /* input stride per output pixel in h / w (0 = 1): CausalDownsample2x of the encoder
                                (modeling_resnet.py:291-336) reads X[... (h*in_sh+dh) ... (w*in_sw+dw) ...] */
    int in_st;               /* input frame stride per output frame (0 = 1): CausalTemporalDownsample2x (:458-502) */
    /* optional: GroupNorm statistics of the OUTPUT accumulated by the conv's epilogue (CausalGroupNorm of the layer that
     * reads Y: modeling_causal_conv.py:36-43), double [T][gn_C][2] = (sum, sum of squares) per output frame and channel,
     * zeroed by the caller -- the layout pf_gn_stats writes, so pf_gn_apply can follow without a pf_gn_stats pass.
     * Honoured only where pf_conv3d_fuses_gn_stats(desc) returns 1 (plain output map; the LDS-halo direct convolution, or
     * the 256-row implicit-GEMM kernel on frames of a multiple of 256 pixels); otherwise ignored and the caller runs
     * pf_gn_stats.  NULL = none. */
    double* gn_stats;
    int gn_C;
} pf_conv_desc;
int pf_conv3d_bf16(const pf_conv_desc* d, pf_stream_t stream);
int pf_conv3d_fuses_gn_stats(const pf_conv_desc* d);   /* 1 = pf_conv3d_bf16(d) accumulates d->gn_stats */
/* which kernel pf_conv3d_bf16(d) launches: -1 conv_narrow_kernel (<= 8 filters: conv_out), -2 conv_halo128_kernel (LDS-halo
 * direct conv: 3 x 3 x 3 taps, N = 128 k filters, 128 / 256 / 512 input channels, frames of whole 16 x 32 patches, plain or
 * pixel-shuffle / depth-to-time output map -- the decoder's resnet convs and upsamplers, modeling_resnet.py:115-150, 609-617,
 * 716-729), 8 gemm8p_kernel, 128 / 192 / 256 gemm256_kernel<BN>, 0 the
 * 128 x 128 implicit GEMM; -100 = the descriptor is rejected */
int pf_conv3d_which(const pf_conv_desc* d);


/* ------------------------------------------------------------------ attention --------------------
 * O = softmax(Q K^T * scale + mask) V, head_dim 64, over the joint [text | image] sequence of one
 * pyramid stage.  Replaces F.scaled_dot_product_attention + the [B,1,L,L] bool mask
 * (flux_block.py:361-365, 597-599; modeling_pyramid_flux.py:318-350).  The mask is implicit: query
 * row i may see keys j in [a_lo[i], a_hi[i]) for j < Lt (text part) and keys Lt <= j < b_hi[i]
 * (image part).  tile_kv_end[b][qt] = max b_hi over the 128-row q tile qt (and >= Lt if any text key
 * is visible).  Q/K are token-major (row stride ldq/ldk, head h at column h*64).  V: either token-major like K (ABI 4:
 * `V`, `ldv`, `strideV`, head stride = head_stride_qk; the kernels transpose it on the way out of LDS with
 * ds_read_b64_tr_b16 -- no extra pass, what the engines use), or, with V = NULL, `Vt` = the image written by
 * pf_v_transpose: [B][H][64][Lp], keys permuted inside groups of 16.  O may alias Q. */
typedef struct {
    const void* Q; const void* K; const void* Vt; void* O;
    int ldq, ldk, ldo;
    long long strideQ, strideK, strideO, strideVt_b, strideVt_h;
    int B, H, L, Lp, Lt;
    const int* a_lo; const int* a_hi; const int* b_hi; /* [B][L] */
    const int* tile_kv_end;                            /* [B][ceil(L/128)] */
    float scale;
    int head_stride_qk; /* elements between consecutive heads inside a Q / K row (0 or 64 = packed heads; 192 in the
                          head-major [k|v|q] layout of the sequence-parallel exchange buffers) */
    int q_row_begin;   /* query rows below this index are not needed (their O rows are left untouched; rounded down to a
                          multiple of 128): the last block only feeds the current frame's rows to the output */
    int q_prescaled;   /* 1: Q was already multiplied by scale*log2(e) (pf_qk_norm_rope q_scale): `scale` is ignored and
                          the scores are used as base-2 exponents directly (saves one FMA per score) */
    /* optional scratch (pf_attention_workspace_bytes; NULL / 0 = none).  With it, large pre-scaled problems run as a PAIR of
     * launches of the 64-rows-per-wave kernel (csrc/attention_w64.h): a pass without any running row maximum (exact while
     * every row's largest base-2 score stays within about +-100 of zero: q.k after QK-RMSNorm is bounded by 11.5 x the
     * norm gains) that flags, per wave, rows whose softmax denominator came out non-finite or vanishing, and a pass with
     * the running maximum that recomputes flagged workgroups only (it returns at once otherwise).  Must not be shared by
     * attention launches that may overlap. */
    void* workspace;
    long long workspace_bytes;
    /* ABI 4: V token-major (NULL = use Vt).  Row stride ldv (elements, % 8 == 0), batch stride strideV, heads at the same
     * stride as in Q / K (head_stride_qk). */
    const void* V;
    int ldv;
    long long strideV;
} pf_attn_desc;
int pf_attention_bf16(const pf_attn_desc* d, pf_stream_t stream);
long long pf_attention_workspace_bytes(int B, int H, int L);
/* which kernel pf_attention_bf16 would run for this descriptor: 64 = the 64-rows-per-wave fast + fix-up pair, 32 = the
 * 32-rows-per-wave kernel (tests assert that the shapes of the benchmark take the pair) */
int pf_attention_which(const pf_attn_desc* d);
int pf_v_transpose(const void* V, void* Vt, int ldv, long long strideV, long long strideVt_b, long long strideVt_h,
                   int B, int H, int L, int Lp, int head_stride /* 0 = 64 */, pf_stream_t stream);

/* ------------------------------------------------------------------ token-wise ops ---------------
 * pf_ln_modulate: y = LN(x; no affine, eps) * (1 + scale[b]) + shift[b]   (AdaLayerNormZero/Single/
 *   Continuous modeling_normalization.py:107-249 and norm2/norm2_context flux_block.py:1022-1036).
 *   x,y bf16 [B][rows_per_batch][D] with batch strides / leading dims in elements; shift/scale fp32
 *   [B][mod_bstride]. */
int pf_ln_modulate(const void* x, void* y, const float* shift, const float* scale, int D, int B, int rows_per_batch,
                   long long x_bstride, long long y_bstride, int ldx, int ldy, int mod_bstride, float eps,
                   pf_stream_t stream);
/* pf_qk_norm_rope: in place on a fused projection buffer [B][L][ld]: q at column q_off, k at k_off
 *   (H heads x 64).  RMSNorm(eps) with fp32 weights (text rows < Lt use *_txt, NULL = same as image:
 *   norm_added_q/k vs norm_q/k, flux_block.py:846-850, 772-775), then RoPE with the per-token table
 *   rope[L][32][cos,sin] (flux_block.py:34-39).  q (not k) is multiplied by q_scale in fp32 before its single
 *   rounding to bf16 (1.0 = reference values; softmax_scale*log2(e) feeds pf_attention_bf16's q_prescaled path). */
int pf_qk_norm_rope(void* qkv, int ld, long long bstride, int q_off, int k_off, const float* wq_img,
                    const float* wk_img, const float* wq_txt, const float* wk_txt, const float* rope, int B, int L,
                    int Lt, int H, float eps, float q_scale, int head_stride /* 0 = 64 */, pf_stream_t stream);
/* pf_gemv_f32: y[b][0:N] (+)= W[N][K](bf16) . act(x[b][0:K]) + bias, 1 <= B <= 4, act = SiLU if silu_in
 *   (time_text_embed and every AdaLN linear: modeling_embedding.py:185-200, modeling_normalization.py:160,227,111) */
int pf_gemv_f32(const void* W, int ldw, const float* bias, const float* x, int ldx, float* y, int ldy, int N, int K,
                int B, int silu_in, int accumulate, pf_stream_t stream);
/* pf_timestep_embed: out[b][0:dim] = [cos(t_b f_k) | sin(t_b f_k)], f_k = exp(-ln(1e4) k/(dim/2));
 *   t_host is a HOST array of B floats (passed by value into the launch). modeling_embedding.py:11-62 */
int pf_timestep_embed(float* out, int ld, int B, const float* t_host, int dim, pf_stream_t stream);
/* pf_patchify: latent clip [C][T][H][W] (fp32 or bf16) -> ncopies x tokens[(t h w)][(p1 p2 c)] bf16
 *   (modeling_pyramid_flux.py:286-287; CFG duplicates the clip, pipeline.py:747) */
int pf_patchify(const void* x, int x_is_f32, void* tok, int C, int T, int H, int W, int ld, long long bstride,
                int ncopies, pf_stream_t stream);
/* pf_cfg_euler_step: v fp32 tokens [2][n][ld] of the current frame -> unpatchify, CFG combine with
 *   `guidance`, x += dsigma * v on the fp32 latent x[C][H][W] (pipeline.py:771-784,
 *   scheduling_flow_matching.py:278-286).  round_bf16 reproduces the reference bf16 rounding points. */
int pf_cfg_euler_step(const float* v, long long vb_stride, int ld, float* x, int C, int H, int W, float guidance,
                      int use_cfg, float dsigma, int round_bf16, pf_stream_t stream);
int pf_copy_rows(const void* src, void* dst, int rows, int D, int ld_src, int ld_dst, long long src_bstride,
                 long long dst_bstride, int B, pf_stream_t stream);
/* pf_sp_relayout: pack / unpack of the sequence-parallel all-to-all chunks (replaces the tensor_split + contiguous +
 *   cat copies of trainer_misc/communicate.py:17-21).  For part p < n_parts (host arrays col0/cols/off, elements):
 *   chunks[off[p] + (r*B + b)*cols[p] + c]  <->  mat[b*mat_bstride + r*ld + col0[p] + c],  c < cols[p], r < rows.
 *   to_chunks = 1 packs (mat -> chunks), 0 unpacks. */
int pf_sp_relayout(void* mat, void* chunks, int rows, int B, int ld, long long mat_bstride, int n_parts,
                   const int* col0, const int* cols, const long long* off, int to_chunks, pf_stream_t stream);
/* pf_renoise_upsample: xout = alpha * nearest_up2(xin) + beta * noise   (pipeline.py:729-743) */
int pf_renoise_upsample(const float* xin, const float* noise, float* xout, int C, int H, int W, float alpha,
                        float beta, int round_bf16, pf_stream_t stream);
/* pf_avgpool2: 2x2 mean * mul == F.interpolate(bilinear, 1/2) (pipeline.py:565, 1116) */
int pf_avgpool2(const float* xin, float* xout, long long planes, int H, int W, float mul, int round_bf16,
                pf_stream_t stream);


/* ------------------------------------------------------------------ VAE decode helpers -----------
 * Activations are channels-last bf16 frames [Hp][Wp][Cp]; `off`/`fs` give the element offset of the
 * first interior pixel and the frame stride.  CausalGroupNorm (per-frame GroupNorm, eps 1e-6) + SiLU:
 * modeling_causal_conv.py:36-43, modeling_resnet.py:127-141.  stats = double [T][C][2], zeroed by caller. */
int pf_gn_stats(const void* x, double* stats, int T, int C, int Cp, int H, int W, int Hp, int Wp,
                long long frame_stride, long long base_off, pf_stream_t stream);
int pf_gn_apply(const void* x, void* y, const double* stats, const float* gamma, const float* beta, int T, int C,
                int G, int H, int W, int Cp_in, int Hp_in, int Wp_in, long long fs_in, long long off_in, int Cp_out,
                int Hp_out, int Wp_out, long long fs_out, long long off_out, float eps, int silu, pf_stream_t stream);
/* row softmax of S (bf16, in place) * scale over the first n_valid columns, zero beyond (mid-block attention,
 * diffusers Attention with upcast_softmax; modeling_block.py:413-427) */
int pf_softmax_rows(void* S, int ld, int n_valid, int n_cols, int rows, float scale, pf_stream_t stream);
/* every conv-cache update of one decode / encode chunk in ONE launch (the `cache_front_feat` bookkeeping of
 * CausalConv3d.forward, modeling_causal_conv.py:128-143): for buffer i (bf16 [2 + T][frame_elems[i]], HOST arrays of
 * `count` <= 64 entries) the two leading cache slots receive the last two of the 2 + n_frames[i] frames; n_frames[i] = 0
 * (ABI 7) zeroes them instead: the causal zero padding in front of a clip's first chunk (:128-131). */
int pf_shift_caches(int count, const void* const* bufs, const long long* frame_elems, const int* n_frames, pf_stream_t stream);
/* latent z [C][T][H][W] fp32, frames t0..t0+nt, window (h0,w0,th,tw) -> channels-last bf16 with per-frame-class
 * affine (frame 0: a0 z + b0, others a1 z + b1: decode_latent un-normalisation, pipeline.py:1226-1230) */
int pf_latent_to_nhwc(const float* z, void* y, int C, int T, int H, int W, int t0, int nt, int h0, int w0, int th,
                      int tw, int Cp, int Hp, int Wp, long long fs_out, long long off_out, float a0, float b0,
                      float a1, float b1, pf_stream_t stream);
/* crop [0:crop_h, 0:crop_w] of a channels-last bf16 tile [T][Ht][Wt][Cp] -> fp32 planar out[c][t][y0+y][x0+x], c < C,
 * out dims [C][T][H][W] (assembly of the tiled-encode moments, modeling_causal_vae.py:452-466) */
int pf_nhwc_to_planar_f32(const void* tile, float* out, int T, int Ht, int Wt, int Cp, int C, int crop_h, int crop_w,
                          int H, int W, int y0, int x0, pf_stream_t stream);
/* blend_v / blend_h of tiled decode (modeling_causal_vae.py:397-407): tiles [T][H][W][Cp] bf16, b updated in place */
int pf_blend_tiles(const void* a, void* b, int T, int Ha, int Wa, int Hb, int Wb, int Cp, int extent, int vertical,
                   pf_stream_t stream);
/* crop [0:crop_h, 0:crop_w] of a decoded tile -> uint8 RGB frames at (y0, x0) of out [T][H][W][3]
 * (pipeline.py:1238-1239) */
int pf_to_uint8(const void* tile, void* out, int T, int Ht, int Wt, int Cp, int crop_h, int crop_w, int H, int W,
                int y0, int x0, pf_stream_t stream);

/* video egress (the step after the path: diffusers.utils.export_to_video(frames, path, fps=24), inference_multigpu.py:92):
 * uint8 RGB frames [T][H][W][3] -> planar Y [T][H][W] and Cb / Cr [T][H/2][W/2], JFIF full-range BT.601 in 16-bit fixed
 * point (Y = (19595 R + 38470 G + 7471 B + 2^15) >> 16; chroma from the 2x2 block sums), the planes a YUV4MPEG2 stream or
 * an encoder takes.  H and W even.  y_frame_stride / c_frame_stride: bytes between consecutive frames of the Y plane and
 * of each chroma plane (0 = dense planes; H*W*3/2 for both packs whole I420 frames [Y | Cb | Cr] back to back). */
int pf_rgb_to_yuv420(const void* rgb, void* y, void* u, void* v, int T, int H, int W, long long y_frame_stride,
                     long long c_frame_stride, pf_stream_t stream);

/* ------------------------------------------------------------------ prompt encoders --------------
 * The step before the sampling path: FluxTextEncoderWithMask (pyramid_dit/flux_modules/modeling_text_encoder.py:15-134)
 * and SD3TextEncoderWithMask (pyramid_dit/mmdit_modules/modeling_text_encoder.py:15-139) call transformers'
 * T5EncoderModel and CLIPTextModel(WithProjection) (requirements.txt pins transformers==4.39.3).  Linear layers use
 * pf_gemm_bf16 (CLIP MLP activation via PF_GEMM_ACT_*), CLIP's affine LayerNorm is pf_ln_modulate with
 * shift = beta, scale = gamma - 1.
 * pf_embed_rows: out[r][0:D] = table[ids[r]] (+ pos[r % L])   (T5 `shared` / CLIP token+position embedding);
 *   ids are clamped to [0, vocab). */
int pf_embed_rows(const void* table, const int* ids, const void* pos /* NULL = none */, void* out, int D, int n, int L,
                  int ldo, int vocab, pf_stream_t stream);
/* pf_rmsnorm: T5LayerNorm  y = x * rsqrt(mean(x^2) + eps) * w  (fp32 math, one rounding), D <= 4096 */
int pf_rmsnorm(const void* x, void* y, const float* w, int D, int rows, int ldx, int ldy, float eps, pf_stream_t stream);
/* pf_glu_mul: y[r][c] = x[r][c] * x[r][F + c], c < F   (T5DenseGatedActDense: hidden_gelu * hidden_linear) */
int pf_glu_mul(const void* x, void* y, int rows, int F, int ldx, int ldy, pf_stream_t stream);
/* pf_attention_small_bf16: O = softmax(Q K^T * scale + bias[h] + masks) V, head_dim 64, L <= 256, token-major Q/K/V/O
 *   (head h at column h*64).  bias: fp32 [H][L][L] additive (T5 relative-position bias) or NULL; key_mask: int [B][L],
 *   0 = padded key excluded (T5 attention_mask) or NULL; causal: key j > query i excluded (CLIP).  A row with no visible
 *   key gets zeros. */
typedef struct {
    const void* Q; const void* K; const void* V; void* O;
    int ldq, ldk, ldv, ldo;
    long long strideQ, strideK, strideV, strideO; /* per batch entry */
    int B, H, L;
    const float* bias; const int* key_mask;
    int causal; float scale;
} pf_attn_small_desc;
int pf_attention_small_bf16(const pf_attn_small_desc* d, pf_stream_t stream);

/* ------------------------------------------------------------------ multi-GPU communicator --------
 * RCCL over xGMI, one process per GPU.  Replaces the torch.distributed calls of the reference's multi-GPU paths:
 * dist.all_to_all in trainer_misc/communicate.py:7-26 (sequence parallelism, pf_all_to_all_v), the isend/irecv halo
 * pass and the list all_gather of video_vae/context_parallel_ops.py:41-114 (context parallelism: pf_halo_send_recv,
 * pf_all_gather_v), the velocity-token sum / input broadcast that stand in for
 * pyramid_dit_for_video_gen_pipeline.py:752-756 (pf_all_reduce_sum_f32, pf_broadcast_bytes).
 * Bootstrap like ncclCommInitRank: ONE rank calls pf_comm_unique_id and ships the 128 bytes to the others by any
 * channel; every rank then calls pf_comm_init(rank, world, id).  Collectives run on the communicator's own HIP stream,
 * ordered after the work already queued on `compute` (event); pf_comm_wait(c, s) makes stream s wait for them on the
 * device.  Counts / offsets are in BYTES and indexed by peer rank; zero counts are allowed (uneven head maps). */
typedef struct pf_comm pf_comm;
int pf_comm_unique_id(void* out128);
int pf_comm_init(pf_comm** out, int rank, int world, const void* unique_id_128);
int pf_comm_destroy(pf_comm* c);
/* COPY-ENGINE TRANSPORT (ABI 7).  The v-collectives below move chunks of at most `slot_bytes` WITHOUT kernels once windows are
 * attached: device-to-device copies between IPC-mapped exchange windows (SDMA over xGMI) ordered by stream memory operations
 * (hipStreamWriteValue32 / hipStreamWaitValue32) -- a persistent GEMM that owns every CU does not delay them and they take no
 * CU from it (csrc/comm.hip has the protocol; measured motivation: profiles/r05_comm_overlap_bench.log).  Bootstrap: every
 * rank calls pf_comm_create_window (allocates its window, returns a 64-byte IPC handle), the host ships all handles to all
 * ranks, every rank calls pf_comm_attach_windows(handles[world][64]).  Larger chunks keep the RCCL path.
 * pf_comm_init_local: a communicator without RCCL (own stream + events): usable with windows only.
 * pf_comm_transport: 0 = RCCL kernels, 1 = windows attached. */
int pf_comm_init_local(pf_comm** out, int rank, int world);
int pf_comm_create_window(pf_comm* c, long long slot_bytes, void* handle_out64);
int pf_comm_attach_windows(pf_comm* c, const void* handles);
int pf_comm_transport(const pf_comm* c);
int pf_comm_rank(const pf_comm* c);
int pf_comm_world(const pf_comm* c);
int pf_all_to_all_v(pf_comm* c, const void* send, const long long* send_bytes, const long long* send_offs, void* recv,
                    const long long* recv_bytes, const long long* recv_offs, pf_stream_t compute);
int pf_halo_send_recv(pf_comm* c, const void* send, void* recv, long long bytes, pf_stream_t compute);
int pf_all_gather_v(pf_comm* c, const void* send, void* recv, const long long* bytes, const long long* offs,
                    pf_stream_t compute);
int pf_all_reduce_sum_f32(pf_comm* c, float* buf, long long count, pf_stream_t compute);
int pf_broadcast_bytes(pf_comm* c, void* buf, long long bytes, int root, pf_stream_t compute);
int pf_comm_wait(pf_comm* c, pf_stream_t stream);

/* ------------------------------------------------------------------ launch lists ------------------
 * The kernel sequence of one transformer forward, recorded once per (unit, stage) and re-issued from C: the reference
 * calls the transformer 10-20 times per pyramid stage with identical shapes, buffers and weights
 * (pyramid_dit_for_video_gen_pipeline.py:611-660); the host cost of those ~300 launches per forward (and, sequence
 * parallel, of ~50 communicator calls: trainer_misc/communicate.py:7-26 inside flux_block.py:266-325) moves from the
 * interpreter into one call.  pf_cmdlist_<op> takes the arguments of pf_<op> with the list in front and, instead of the
 * stream, a stream SLOT: 0 = the compute stream, 1 = the side stream given to pf_cmdlist_run; pf_cmdlist_join(l, from, to)
 * makes slot `to` wait for everything recorded so far on slot `from`.  pf_cmdlist_run issues the entries in order through
 * the same entry points (bit-identical to the eager sequence).  pf_cmdlist_instantiate captures that replay into a
 * hipGraph (lists without communicator entries; every side-stream entry must have been joined back to slot 0), after
 * which pf_cmdlist_run is one hipGraphLaunch; the pointers recorded in a list must stay valid while it is in use. */
typedef struct pf_cmdlist pf_cmdlist;
pf_cmdlist* pf_cmdlist_create(void);
int pf_cmdlist_destroy(pf_cmdlist* l);
int pf_cmdlist_clear(pf_cmdlist* l);
int pf_cmdlist_size(const pf_cmdlist* l);
int pf_cmdlist_is_graph(const pf_cmdlist* l);
int pf_cmdlist_gemm(pf_cmdlist* l, const pf_gemm_desc* d, int slot);
int pf_cmdlist_attention(pf_cmdlist* l, const pf_attn_desc* d, int slot);
int pf_cmdlist_ln_modulate(pf_cmdlist* l, const void* x, void* y, const float* shift, const float* scale, int D, int B,
                           int rows_per_batch, long long x_bstride, long long y_bstride, int ldx, int ldy, int mod_bstride,
                           float eps, int slot);
int pf_cmdlist_qk_norm_rope(pf_cmdlist* l, void* qkv, int ld, long long bstride, int q_off, int k_off, const float* wq_img,
                            const float* wk_img, const float* wq_txt, const float* wk_txt, const float* rope, int B, int L,
                            int Lt, int H, float eps, float q_scale, int head_stride, int slot);
int pf_cmdlist_v_transpose(pf_cmdlist* l, const void* V, void* Vt, int ldv, long long strideV, long long strideVt_b,
                           long long strideVt_h, int B, int H, int L, int Lp, int head_stride, int slot);
int pf_cmdlist_sp_relayout(pf_cmdlist* l, void* mat, void* chunks, int rows, int B, int ld, long long mat_bstride, int n_parts,
                           const int* col0, const int* cols, const long long* off, int to_chunks, int slot);
int pf_cmdlist_copy_rows(pf_cmdlist* l, const void* src, void* dst, int rows, int D, int ld_src, int ld_dst,
                         long long src_bstride, long long dst_bstride, int B, int slot);
int pf_cmdlist_all_to_all_v(pf_cmdlist* l, pf_comm* c, const void* send, const long long* send_bytes,
                            const long long* send_offs, void* recv, const long long* recv_bytes, const long long* recv_offs,
                            int slot);
int pf_cmdlist_comm_wait(pf_cmdlist* l, pf_comm* c, int slot);
int pf_cmdlist_join(pf_cmdlist* l, int from_slot, int to_slot);
int pf_cmdlist_run(pf_cmdlist* l, pf_stream_t compute, pf_stream_t side);
int pf_cmdlist_instantiate(pf_cmdlist* l, pf_stream_t compute, pf_stream_t side);

#ifdef __cplusplus
}
#endif
#endif
