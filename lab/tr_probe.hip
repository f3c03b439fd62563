// lab/tr_probe.hip -- what does ds_read_b64_tr_b16 deliver?  Each lane supplies the address of 4 consecutive 16-bit
// elements; LDS holds element i at index i, lane l reads at element 4 l.  Printed: for every lane the four element indices it
// received = (source lane, source position) pairs -- the transpose pattern inside a 16-lane group.
// build: hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int el = 4 * l;
    if (mode == 1) el = (l & 15) * 64 + (l >> 4) * 4;          // 16 rows of 64 elements, lane group g reads columns 4g..4g+3
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + el));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int main() {
    short* d;
    hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d (lane l reads 4 elements at %s):\n", mode, mode ? "(l%16)*64 + (l/16)*4" : "4*l");
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) {
                const int v = h[l * 4 + e];
                if (mode == 0) printf("  [lane %2d pos %d]", v / 4, v % 4);
                else printf("  [row %2d col %2d]", v / 64, v % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
