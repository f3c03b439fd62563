// Interleaved A/B of pf_gemm_bf16's persistent kernel under different pf_gemm_set_policy hooks, on the DiT's GEMM shapes at
// the headline sequence (M = 2 x 15 488), WITHOUT Python: the binary starts in a second on the GPU box.
//   build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/gemm_epi_ab.cpp -Lpyramid-flow_amd -lpyflow_hip \
//                 -Wl,-rpath,'$ORIGIN/../pyramid-flow_amd' -o tools/gemm_epi_ab
//   run:    tools/gemm_epi_ab [rounds] [arm policy lists, e.g. "1000" "1003"]   (an arm = comma-separated policy values
//           applied in order before the launch; default arms: 1000 (the wave groups' epilogues one after the other, as in
//           round 4) and 1001 (concurrent: round 5's default))
// Prints per shape: median / min / max TFLOP/s per arm and whether the arms' outputs are bit-identical (64-bit checksums).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pyflow_hip.h"
#ifdef R4LIB          /* A/B against the round-4 build of the library (lab/r4lib): it has no pf_gemm_which_desc */
#define WHICH(d) pf_gemm_which((d).M, (d).batch, (d).N, (d).K)
#else
#define WHICH(d) pf_gemm_which_desc(&(d))
#endif

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

__global__ void fill_bf16(unsigned short* p, long long n, unsigned seed, float scale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // sum of two uniforms, roughly bell-shaped, in [-scale, scale]
        const float u = ((h & 0xffff) + (h >> 16)) * (1.0f / 65535.0f) - 1.0f;
        const float v = u * scale;
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}
__global__ void fill_f32(float* p, long long n, unsigned seed, float scale, float offset) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = offset + scale * (((h & 0xffff) + (h >> 16)) * (1.0f / 65535.0f) - 1.0f);
    }
}
__global__ void fill_rope(float* p, long long rows) {          // [rows][32][cos, sin] of some angle
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 32) return;
    const float ang = (float)(i % 977) * 0.013f;
    p[2 * i] = cosf(ang);
    p[2 * i + 1] = sinf(ang);
}
__global__ void checksum_u16(const unsigned short* p, long long n, unsigned long long* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned long long s = 0;
    for (; i < n; i += stride) s += (unsigned long long)p[i] * (unsigned long long)((i % 1021) + 1);
    atomicAdd(out, s);
}

struct Shape { int M, B, N, K, gelu_from; bool res, qk; const char* what; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    std::vector<std::vector<int>> arms;
    std::vector<std::string> arm_names;
    for (int i = 2; i < argc; ++i) {
        std::vector<int> a;
        char* dup = strdup(argv[i]);
        for (char* t = strtok(dup, ","); t; t = strtok(nullptr, ",")) a.push_back(atoi(t));
        arms.push_back(a);
        arm_names.push_back(argv[i]);
        free(dup);
    }
    if (arms.empty()) { arms = {{1000}, {1001}}; arm_names = {"1000", "1001"}; }
    const int D = 1920, L = 15488;
    const Shape shapes[] = {
        {L, 2, 3 * D, D, -1, false, true, "double K|V|Q (QK epilogue)"},
        {L, 2, 7 * D, D, 3 * D, false, true, "single K|V|Q|MLP (QK + GELU)"},
        {L, 2, 4 * D, D, 0, false, false, "double MLP up (GELU)"},
        {L, 2, D, D, -1, true, false, "double attn out (residual)"},
        {L, 2, D, 4 * D, -1, true, false, "double MLP down (residual)"},
        {L, 2, D, 5 * D, -1, true, false, "single proj_out (residual)"},
        {3008, 2, 7 * D, D, 3 * D, false, true, "single K|V|Q|MLP at L = 3008"},
        {3008, 2, D, 5 * D, -1, true, false, "single proj_out at L = 3008"},
        // a sequence-parallel rank's rows at P = 8 (L / 8 = 1 936): 128 tiles of 256 x 256 for the d-wide projections
        {1936, 2, D, D, -1, true, false, "P = 8 rank: attn out (residual)"},
        {1936, 2, D, 4 * D, -1, true, false, "P = 8 rank: MLP down (residual)"},
        {1936, 2, D, 5 * D, -1, true, false, "P = 8 rank: proj_out (residual)"},
        {1936, 2, 3 * D, D, -1, false, true, "P = 8 rank: K|V|Q (QK epilogue)"},
        {1936, 2, 4 * D, D, 0, false, false, "P = 8 rank: MLP up (GELU)"},
        // the 128 text rows of a double block where they are not grouped (short sequences): the 128 x 128 kernel, K split with scratch
        {128, 2, 3 * D, D, -1, false, false, "text K|V|Q (128 rows)"},
        {128, 2, 4 * D, D, 0, false, false, "text MLP up (128 rows, GELU)"},
        {128, 2, D, 4 * D, -1, true, false, "text MLP down (128 rows, residual)"},
        {368, 2, D, 5 * D, -1, true, false, "proj_out at L = 368 (unit 0, stage 0)"},
        {608, 2, 7 * D, D, 3 * D, false, false, "K|V|Q|MLP at L = 608"},
        // the d-wide projections WITHOUT their residual (what the residual loads + gate arithmetic of flavour 1 cost: compare 3 / 5)
        {L, 2, D, D, -1, false, false, "attn out shape, plain (no residual)"},
        {L, 2, D, 5 * D, -1, false, false, "proj_out shape, plain (no residual)"},
    };
    const char* only = getenv("GEMM_AB_SHAPES");          // e.g. "0,1,3"
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const long long maxA = 2ll * L * 5 * D, maxC = 2ll * L * 7 * D, maxW = 7ll * D * D;
    unsigned short *A, *W, *C, *R;
    float *bias, *gate, *rope, *wq, *wk;
    unsigned long long* cs;
    void* ws;
    const long long ws_bytes = pf_gemm_workspace_bytes(L, 2, D, 5 * D);
    CK(hipMalloc(&A, maxA * 2)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&C, maxC * 2)); CK(hipMalloc(&R, 2ll * L * D * 2));
    CK(hipMalloc(&bias, 7 * D * 4)); CK(hipMalloc(&gate, 2 * D * 4)); CK(hipMalloc(&rope, (long long)L * 64 * 4));
    CK(hipMalloc(&wq, 64 * 4)); CK(hipMalloc(&wk, 64 * 4)); CK(hipMalloc(&cs, 8));
    CK(hipMalloc(&ws, ws_bytes > 0 ? ws_bytes : 64 << 20));
    fill_bf16<<<2048, 256, 0, st>>>(A, maxA, 1u, 1.5f);
    fill_bf16<<<2048, 256, 0, st>>>(W, maxW, 2u, 0.04f);
    fill_bf16<<<2048, 256, 0, st>>>(R, 2ll * L * D, 3u, 1.0f);
    fill_f32<<<64, 256, 0, st>>>(bias, 7 * D, 4u, 0.5f, 0.f);
    fill_f32<<<64, 256, 0, st>>>(gate, 2 * D, 5u, 0.5f, 0.f);
    fill_f32<<<1, 64, 0, st>>>(wq, 64, 6u, 0.2f, 1.f);
    fill_f32<<<1, 64, 0, st>>>(wk, 64, 7u, 0.2f, 1.f);
    fill_rope<<<(L * 32 + 255) / 256, 256, 0, st>>>(rope, L);
    CK(hipStreamSynchronize(st));
    printf("# rounds %d, arms:", rounds);
    for (auto& n : arm_names) printf(" [%s]", n.c_str());
    printf("   (TFLOP/s median (min .. max); workspace %lld MiB)\n", ws_bytes >> 20);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int si = -1;
    for (const Shape& s : shapes) {
        ++si;
        if (only) {
            char key[8]; snprintf(key, sizeof key, "%d", si);
            std::string o = std::string(",") + only + ",";
            if (o.find(std::string(",") + key + ",") == std::string::npos) continue;
        }
        pf_gemm_desc d;
        memset(&d, 0, sizeof d);
        d.A = A; d.W = W; d.C = C; d.bias = bias;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.K; d.ldw = s.K; d.ldc = s.N; d.ldr = s.N;
        d.strideA = (long long)s.M * s.K; d.strideC = (long long)s.M * s.N; d.strideR = (long long)s.M * s.N;
        d.batch = s.B; d.gelu_from = s.gelu_from; d.gate_stride = s.N;
        if (s.res) { d.res = R; d.gate = gate; d.flags = PF_GEMM_GATE_RES; }
        if (s.qk) {
            d.qk_rope = rope; d.qk_wq = wq; d.qk_wk = wk; d.qk_d = D; d.qk_k_col0 = 0; d.qk_q_col0 = 2 * D; d.qk_row0 = 0;
            d.qk_eps = 1e-6f; d.qk_q_scale = 0.18f;
        } else {
            d.workspace = ws; d.workspace_bytes = ws_bytes;
        }
        const double flop = 2.0 * s.M * s.B * s.N * (double)s.K;
        const int iters = std::max(3, (int)(1.5e12 / flop));
        std::vector<std::vector<double>> tf(arms.size());
        std::vector<unsigned long long> sums(arms.size());
        auto apply = [&](const std::vector<int>& a) { for (int p : a) if (pf_gemm_set_policy(p)) { printf("policy %d: %s\n", p, pf_last_error()); exit(1); } };
        for (size_t a = 0; a < arms.size(); ++a) {       // warm-up + checksum
            apply(arms[a]);
            CK(hipMemsetAsync(C, 0, (size_t)s.B * s.M * s.N * 2, st));
            if (pf_gemm_bf16(&d, st)) { printf("pf_gemm_bf16: %s\n", pf_last_error()); return 1; }
            CK(hipMemsetAsync(cs, 0, 8, st));
            checksum_u16<<<1024, 256, 0, st>>>(C, (long long)s.B * s.M * s.N, cs);
            CK(hipMemcpyAsync(&sums[a], cs, 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
        }
        for (int r = 0; r < rounds; ++r) {
            for (size_t k = 0; k < arms.size(); ++k) {
                const size_t a = (r & 1) ? arms.size() - 1 - k : k;          // ABBA
                apply(arms[a]);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) pf_gemm_bf16(&d, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                tf[a].push_back(flop * iters / (ms * 1e-3) / 1e12);
            }
        }
        pf_gemm_set_policy(0);
        printf("M=%dx%d N=%d K=%d %-34s which=%d:", s.M, s.B, s.N, s.K, s.what, WHICH(d));
        for (size_t a = 0; a < arms.size(); ++a) {
            std::sort(tf[a].begin(), tf[a].end());
            printf("  [%s] %.0f (%.0f .. %.0f)", arm_names[a].c_str(), tf[a][tf[a].size() / 2], tf[a].front(), tf[a].back());
        }
        bool same = true;
        for (size_t a = 1; a < arms.size(); ++a) same = same && sums[a] == sums[0];
        printf("  bits %s  checksum %016llx\n", same ? "identical" : "DIFFER", sums[0]);          // (compare across library builds too)
        fflush(stdout);
    }
    return 0;
}
