// Shared device helpers for the pyflow HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

#define PF_DEVICE __device__ __forceinline__

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: set it once per (call site, device),
// thread-safe, so that a C-ABI host driving several GPUs from one process gets its > 64 KiB LDS launches on every device.
// `kernel` may be a parenthesised template-id.
#include <atomic>
#define PF_SET_MAX_LDS_ONCE(kernel, bytes)                                                                          \
    do {                                                                                                            \
        static std::atomic<unsigned long long> pf_done_{0ull};                                                      \
        int pf_dev_ = 0;                                                                                            \
        (void)hipGetDevice(&pf_dev_);                                                                               \
        const unsigned long long pf_bit_ = 1ull << (pf_dev_ & 63);                                                  \
        if (!(pf_done_.load(std::memory_order_acquire) & pf_bit_)) {                                                \
            (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));  \
            pf_done_.fetch_or(pf_bit_, std::memory_order_release);                                                  \
        }                                                                                                           \
    } while (0)

// Direct global -> LDS DMA, 16 bytes per lane.  LDS destination = wave-uniform `lds_base` + lane*16
// (the hardware adds the lane offset); the global source is per lane.
PF_DEVICE void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

PF_DEVICE float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

PF_DEVICE void unpack8(const u32x4_t v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i]     = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}

PF_DEVICE unsigned pack2(float a, float b) {
    bf16x2_t t;
    t[0] = (bf16_t)a;
    t[1] = (bf16_t)b;
    return __builtin_bit_cast(unsigned, t);
}

PF_DEVICE u32x4_t pack8(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
    return v;
}

PF_DEVICE float gelu_tanh(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  ==  x * sigmoid(2u)  ==  x / (1 + 2^(-2 log2(e) u))
    // 3 FMA-class ops + v_exp_f32 + v_rcp_f32 (1 ulp; the result is rounded to bf16): the IEEE division this replaces
    // expanded to ~10 VALU instructions per element in the GEMM epilogues
    const float x2 = x * x;
    const float t = x * __builtin_fmaf(x2, -2.0f * 1.4426950408889634f * 0.7978845608028654f * 0.044715f,
                                       -2.0f * 1.4426950408889634f * 0.7978845608028654f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

// the same function on 8 values, two per instruction: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 do the five non-transcendental
// operations of a pair at once (3 + 1 + 1 packed instructions instead of 10; exp2 / rcp stay per element) -- per element the
// same IEEE operations in the same order as gelu_tanh(): the same bits.  The GELU flavour of the persistent GEMM spends
// 10 k cycles of a tile boundary in this arithmetic (profiles/r06_gemm8p_stamps.log).
typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
PF_DEVICE void gelu_tanh8(float* v) {
    const float k1 = -2.0f * 1.4426950408889634f * 0.7978845608028654f * 0.044715f, k0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const pf_f32x2 x = {v[2 * i], v[2 * i + 1]};
        const pf_f32x2 x2 = x * x;
        const pf_f32x2 t = x * __builtin_elementwise_fma(x2, (pf_f32x2){k1, k1}, (pf_f32x2){k0, k0});
        pf_f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        e = (pf_f32x2){1.0f, 1.0f} + e;
        const pf_f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
        const pf_f32x2 o = x * r;
        v[2 * i] = o[0];
        v[2 * i + 1] = o[1];
    }
}

// CLIP text towers: quick_gelu = x * sigmoid(1.702 x) (CLIP-L), exact erf GELU (CLIP-G); transformers activations.py
PF_DEVICE float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
PF_DEVICE float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// activation of the GEMM epilogue: GELU-tanh unless PF_GEMM_ACT_QUICK_GELU (4) / PF_GEMM_ACT_GELU_ERF (8) is set
PF_DEVICE void act8(float* v, int flags) {
    if (flags & 12) {
        if (flags & 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = quick_gelu(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(v[e]);
    }
}

// QK-RMSNorm + RoPE on 8 consecutive channels of one (token, head) unit (modeling_normalization.py:66-79,
// flux_block.py:34-39) -- ONE definition shared by the standalone pass (elementwise.hip: qk_norm_rope_kernel) and the
// QKV GEMM's epilogue (gemm8p.hip, flavour 8), every multiply-add spelled out, so that the two produce the same bits:
// sum of squares of the bf16 values (in channel order), then per pair (x0, x1) = v * r * w, rotated by (cos, sin), * osc.
PF_DEVICE float qk_sumsq8(const float* v) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[e], v[e], ss);
    return ss;
}
PF_DEVICE float qk_rstd(float ss64, float eps) { return rsqrtf(__builtin_fmaf(ss64, 1.f / 64.f, eps)); }
// w: the 8 gains of these channels; cs: (cos, sin) of their 4 pairs
PF_DEVICE void qk_rope8(const float* v, float r, const float* w, const float* cs, float osc, float* o) {
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const float x0 = (v[2 * pr] * r) * w[2 * pr];
        const float x1 = (v[2 * pr + 1] * r) * w[2 * pr + 1];
        const float c = cs[2 * pr], s = cs[2 * pr + 1];
        o[2 * pr] = __builtin_fmaf(c, x0, -(s * x1)) * osc;
        o[2 * pr + 1] = __builtin_fmaf(s, x0, c * x1) * osc;
    }
}

PF_DEVICE float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// XCD-aware bijective block remap (8 XCDs, block b runs on XCD b%8): gives each XCD a
// contiguous run of logical tile ids so neighbouring tiles share operand panels in one L2.
PF_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
