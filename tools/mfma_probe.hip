// Probe the operand / accumulator layout of v_mfma_f32_32x32x2f32 on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float* A /*[32][2]*/, const float* B /*[2][32]*/, float* out /*[64][16]*/) {
    const int l = threadIdx.x;
    f32x16 c = {0};
    // assumption under test: lane l supplies A[l%32][l/32] and B[l/32][l%32]
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * 2 + l / 32], B[(l / 32) * 32 + l % 32], c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[l * 16 + v] = c[v];
}
int main() {
    float hA[64], hB[64], hO[64 * 16];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) hA[i * 2 + k] = (float)(i + 1) * (k == 0 ? 1.0f : 100.0f);
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (k == 0 ? 1.0f : 0.001f) * (float)(j + 1);
    float *dA, *dB, *dO;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dO, sizeof hO);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dO);
    hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 16; ++v) {
        const int i = 8 * (v / 4) + 4 * (l / 32) + (v % 4), j = l % 32;
        const float want = hA[i * 2] * hB[j] + hA[i * 2 + 1] * hB[32 + j];
        if (fabsf(hO[l * 16 + v] - want) > 1e-3f * fabsf(want)) { if (bad < 5) printf("lane %d v %d got %g want %g\n", l, v, hO[l*16+v], want); ++bad; }
    }
    printf("layout check: %d mismatches (C[i][j]: i = 8*(v/4) + 4*(lane/32) + v%%4, j = lane%%32)\n", bad);
    return bad != 0;
}
