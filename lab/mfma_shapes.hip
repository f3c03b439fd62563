// lab/mfma_shapes.hip -- microbenchmark (NOT part of the library): sustained bf16 MFMA rate of the two gfx950 shapes on
// pseudo-random operands with the whole chip busy (the regime the GEMMs run in: power-limited clocks), operands re-read from
// registers only.  Question: does v_mfma_f32_32x32x16_bf16 (half the operand-register reads per FLOP) sustain a higher rate
// than v_mfma_f32_16x16x32_bf16 at the chip's power limit?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ bf16x8_t rnd8(unsigned seed, int zero) {
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned h = hash(seed * 8 + e);
        const float f = ((h & 0xffff) / 32768.0f - 1.0f);          // uniform [-1, 1)
        v[e] = (__bf16)(zero ? 0.f : f);
    }
    return v;
}

template <int SHAPE>        // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512, 2) void k(int iters, int zero, float* out) {
    const unsigned tid = blockIdx.x * 512 + threadIdx.x;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(tid * 8 + i, zero); b[i] = rnd8(tid * 8 + 4 + i, zero); }
    float sink = 0.f;
    if (SHAPE == 0) {
        f32x16_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q >> 2) & 3], b[q & 3], acc[q & 3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sink += acc[i][r];
    } else {
        f32x4_t acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 32; ++q)
                acc[q & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(q >> 2) & 3], b[q & 3], acc[q & 15], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sink += acc[i][r];
    }
    if (sink == 1234.5f) out[0] = sink;
}

template <int SHAPE>
static void run(const char* name, int zero, float* d) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(512), 0, 0, 2000, zero, d);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(512), 0, 0, iters, zero, d);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    // per iteration and wave: 16 x 32x32x16 (32768 FLOP each) or 32 x 16x16x32 (16384 each) = 524288 FLOP
    const double flop = (double)blocks * 8 * iters * 524288.0;
    printf("%-34s %s operands: %.3f ms  %.0f TFLOP/s\n", name, zero ? "zero  " : "random", best, flop / best / 1e9);
}
int main() {
    float* d; CK(hipMalloc(&d, 64));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_mfma_f32_32x32x16_bf16", 0, d);
        run<1>("v_mfma_f32_16x16x32_bf16", 0, d);
    }
    run<0>("v_mfma_f32_32x32x16_bf16", 1, d);
    run<1>("v_mfma_f32_16x16x32_bf16", 1, d);
    return 0;
}
