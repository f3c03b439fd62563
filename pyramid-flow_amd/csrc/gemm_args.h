// Kernel-argument structures shared by the two GEMM kernels (gemm.hip: 128x128 tile, gemm256.hip: 256xBN ping-pong).
#pragma once
#include "common.h"

namespace pfgemm {

struct ConvGeom {            // implicit-GEMM addressing of a channels-last, spatially padded input
    int H, W;                // output spatial size (rows of the GEMM are (t,h,w) pixels)
    int Hp, Wp;              // padded input spatial pitch (H+2pad, W+2pad)
    int Cin;                 // input channels (K = ntaps*Cin)
    int kt, kh, kw;          // taps
    long long base_off;      // element offset of tap (0,0,0) for output pixel (0,0,0)
    int sh, sw;              // input stride per output pixel (strided down-sampling convs of the VAE encoder)
    int st;                  // input FRAME stride per output frame (CausalTemporalDownsample2x)
};

struct OutMap {              // where output row m / column-group g lands
    int mode;                // 0 = plain [M, ldc];  1 = pixel map into [To,Hop,Wop,C] with shuffles
    int H, W;                // GEMM-row pixel grid
    int st, sh, sw;          // upsample factors (depth-to-time, pixel shuffle)
    int Cg;                  // channels per group (N = st*sh*sw*Cg for shuffles)
    int Hop, Wop;            // padded output pitches
    long long base_off;      // element offset of output pixel (0,0,0) channel 0
    int Cout_pitch;          // channel pitch of the output buffer
    int t_shift;             // added to the output frame index; negative frames are dropped
};

// per-problem fields of a gemm8p launch.  A GROUPED launch (pf_gemm_desc.M2 > 0) walks the tiles of two problems that share N, K,
// the leading dimensions, the batch count and the epilogue flavour -- the image and the text stream of a double block
// (flux_block.py:816-835, 868-872: same shapes, different weights and rows) -- in ONE persistent launch: pr[1]'s tiles follow
// pr[0]'s in the tile order.  The kernel reads them as p.pr[g] with a run-time g (one scalar load from the kernel arguments
// per field and tile, no select).
struct Prob {
    const bf16_t* A; const bf16_t* W; void* C; const float* bias; const bf16_t* res; const float* gate;
    long long sA, sC, sR;
    const float* qk_wq; const float* qk_wk;
    int M, qk_row0;
};

struct Args {
    const bf16_t* A; const bf16_t* W; void* C;
    const float* bias; const bf16_t* res; const float* gate;
    int M, N, K, lda, ldw, ldc, ldr;
    long long sA, sC, sR;    // batch strides (elements)
    int gate_stride, batch;
    int gelu_from, flags, n_valid;
    float out_scale;
    int group_m;             // M-tiles per L2 super-tile of the 256-row kernels (tile order; 0 = default)
    int ksplit;              // 128 x 128 kernel: workgroups per output tile along K (0 / 1 = none); > 1 writes raw fp32
    float* part;             //   partial sums part[((s * batch + b) * M + m) * N + n] instead of running the epilogue
                             // gemm8p: ksplit = workgroups of the main launch, part = scratch of the tail split (parked sums),
    int tail_ov;             //   tail_ov = the split's fixed cost in K-tile periods (gemm8p.hip: tail_plan)
    int stagger;             // gemm8p: > 0 = workgroups with a tile less than the busiest of their XCD start late, by
                             //   ((slot - r) % 8) * stagger * (K / 64) cycles (1/8 steps of a tile time); 0 = all start together
    int epi_mode;            // gemm8p: bit 0 = the two wave groups run their epilogues CONCURRENTLY (group 0's after the tile's last
                             //   barrier instead of before it; the default -- 0 = one after the other: A/B switch)
    // QK-RMSNorm + RoPE of the K / Q column blocks in the epilogue (gemm8p flavour 8; pf_gemm_desc.qk_*): qk_d = 0 -> none
    const float* qk_rope; const float* qk_wq; const float* qk_wk;
    int qk_d, qk_q0, qk_k0, qk_row0;
    int qk_hs;               // > 0: head-major columns, qk_hs per head; the K / Q blocks start at qk_k0 / qk_q0 inside every head
    float qk_eps, qk_qs;
    double* gn_stats;        // conv kernels of gemm256.hip: [frame][gn_C][2] (sum, sum of squares) of the OUTPUT, accumulated in
    int gn_C;                //   the epilogue for the GroupNorm that reads it (nullptr = none)
    ConvGeom cg; OutMap om;
    Prob pr[2];              // gemm8p: [0] = the fields above (filled by its launcher), [1] = the second problem of a grouped launch
    int tiles2_m;            //   256-row tiles per batch entry of pr[1] (0 = not grouped)
};


}  // namespace pfgemm
