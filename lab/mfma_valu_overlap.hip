// lab/mfma_valu_overlap.hip -- microbenchmark (NOT part of the library): do MFMA and vector-ALU work overlap on one SIMD of
// gfx950, (a) between the two waves of a SIMD, (b) inside one wave's instruction stream, and does it matter whether the
// MFMA accumulator lives in the ArchVGPR or the AccVGPR half of the register file?
// Motivation: profiles/r03_attention_ablations_and_phase_stamps.log -- the attention kernel's MFMA-only time (1.29 ms) and
// its everything-but-MFMA time (1.36 ms) ADD UP to the full kernel (2.30 ms) for every wave arrangement tried.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/mfma_valu_overlap.hip -o lab/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// one MFMA on accumulator `acc` (ACC = 1: AccVGPR, 0: ArchVGPR)
template <int ACC>
__device__ __forceinline__ void mfma(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    if (ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// KIND 0: v_add_f32, 1: v_exp_f32, 2: v_max3, 3: v_cvt_pk_bf16 ; eight independent registers round-robin
template <int KIND>
__device__ __forceinline__ void valu(float& x, float y) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
}

// one loop iteration of a wave: MF MFMAs (4 accumulators round-robin), VA vector ops spread evenly behind them (everything
// unrolled: register indices are compile-time constants)
template <int ACC, int KIND, int MF, int VA>
__device__ __forceinline__ void body(f32x16_t (&acc)[4], const bf16x8_t& a, const bf16x8_t& b, float (&x)[8], float y) {
    if constexpr (MF == 0) {
#pragma unroll
        for (int k = 0; k < VA; ++k) valu<KIND>(x[k & 7], y);
    } else {
        constexpr int PER = VA / MF;
#pragma unroll
        for (int q = 0; q < MF; ++q) {
            mfma<ACC>(acc[q & 3], a, b);
#pragma unroll
            for (int k = 0; k < PER; ++k) valu<KIND>(x[(q * PER + k) & 7], y);
        }
    }
}

// waves 0-3 (the older half of the workgroup) run MF0 MFMAs + VA0 vector ops per iteration, waves 4-7 MF1 / VA1
template <int ACC, int KIND, int MF0, int VA0, int MF1, int VA1>
__global__ __launch_bounds__(512, 2) void bench(int iters, unsigned long long* out, float seed) {
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
    bf16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e * 0.37f + threadIdx.x * 0.01f); b[e] = (__bf16)(seed * 0.5f - e * 0.21f + threadIdx.x * 0.02f); }
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = seed * 0.001f * (e + 1) + threadIdx.x * 1e-6f;
    const float y = seed * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wid < 4) {
        for (int it = 0; it < iters; ++it) body<ACC, KIND, MF0, VA0>(acc, a, b, x, y);
    } else {
        for (int it = 0; it < iters; ++it) body<ACC, KIND, MF1, VA1>(acc, a, b, x, y);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sink += acc[i][r];
#pragma unroll
    for (int e = 0; e < 8; ++e) sink += x[e];
    if ((threadIdx.x & 63) == 0) {
        out[(size_t)blockIdx.x * 8 + wid] = t1 - t0;
        if (sink == 12345.678f) out[0] = 0;
    }
}

template <int ACC, int KIND, int MF0, int VA0, int MF1, int VA1>
static void run(const char* name, unsigned long long* d_out, int nblk) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((bench<ACC, KIND, MF0, VA0, MF1, VA1>), dim3(nblk), dim3(512), 0, 0, iters, d_out, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((bench<ACC, KIND, MF0, VA0, MF1, VA1>), dim3(nblk), dim3(512), 0, 0, iters, d_out, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)nblk * 8);
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double c0 = 0, c1 = 0;
    for (int b = 0; b < nblk; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? c0 : c1) += (double)h[(size_t)b * 8 + w];
    c0 /= nblk * 4.0 * iters; c1 /= nblk * 4.0 * iters;
    printf("%-52s acc=%s  %7.1f | %7.1f cycles/iter (older | younger half)   %.3f ms\n", name, ACC ? "AGPR" : "VGPR", c0, c1, ms);
}

int main() {
    int dev = 0, ncu = 256;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned long long* d_out;
    CK(hipMalloc(&d_out, (size_t)ncu * 8 * 8));
    printf("one 512-thread workgroup per CU (2 waves per SIMD); cycles per loop iteration from s_memtime; 32x32x16 bf16 MFMA = 32 cycles of the matrix pipe\n");
#define BOTH(KIND, name, a0, b0, a1, b1) run<0, KIND, a0, b0, a1, b1>(name, d_out, ncu); run<1, KIND, a0, b0, a1, b1>(name, d_out, ncu);
    printf("--- A. both waves of a SIMD do the same thing\n");
    BOTH(0, "8 MFMA / iter, no vector ops", 8, 0, 8, 0);
    BOTH(0, "48 v_add / iter, no MFMA", 0, 48, 0, 48);
    BOTH(1, "48 v_exp / iter, no MFMA", 0, 48, 0, 48);
    printf("--- B. one wave of each SIMD runs MFMAs, its partner vector ops (do the two pipes overlap ACROSS waves?)\n");
    BOTH(0, "older: 8 MFMA | younger: 48 v_add", 8, 0, 0, 48);
    BOTH(1, "older: 8 MFMA | younger: 48 v_exp", 8, 0, 0, 48);
    BOTH(0, "older: 48 v_add | younger: 8 MFMA", 0, 48, 8, 0);
    BOTH(0, "older: 8 MFMA | younger: 96 v_add", 8, 0, 0, 96);
    printf("--- C. every wave interleaves: MFMA, k vector ops, MFMA, ... (how many vector ops hide behind one MFMA INSIDE a wave?)\n");
    BOTH(0, "8 x (MFMA + 2 v_add)", 8, 16, 8, 16);
    BOTH(0, "8 x (MFMA + 4 v_add)", 8, 32, 8, 32);
    BOTH(0, "8 x (MFMA + 6 v_add)", 8, 48, 8, 48);
    BOTH(0, "8 x (MFMA + 8 v_add)", 8, 64, 8, 64);
    BOTH(1, "8 x (MFMA + 2 v_exp)", 8, 16, 8, 16);
    BOTH(1, "8 x (MFMA + 4 v_exp)", 8, 32, 8, 32);
    BOTH(1, "8 x (MFMA + 6 v_exp)", 8, 48, 8, 48);
    BOTH(2, "8 x (MFMA + 4 v_max3)", 8, 32, 8, 32);
    BOTH(3, "8 x (MFMA + 4 v_cvt_pk_bf16)", 8, 32, 8, 32);
    printf("--- D. one wave per SIMD active only (waves 4-7 idle): same interleaves\n");
    BOTH(0, "8 x (MFMA + 6 v_add), partner idle", 8, 48, 0, 0);
    BOTH(1, "8 x (MFMA + 6 v_exp), partner idle", 8, 48, 0, 0);
    return 0;
}
