// Probe the operand / accumulator layout of v_mfma_f32_16x16x4f32 on gfx950 (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* A /*[16][4]*/, const float* B /*[4][16]*/, float* out /*[64][4]*/) {
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float hA[64], hB[64], hO[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (float)(i + 1) * (k + 1) * (k % 2 ? 0.5f : 2.0f);
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)(j + 1) + 0.125f * k;
    float *dA, *dB, *dO;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dO, sizeof hO);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dO);
    hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        const int i = 4 * (l / 16) + v, j = l % 16;
        float want = 0; for (int k = 0; k < 4; ++k) want += hA[i * 4 + k] * hB[k * 16 + j];
        if (fabsf(hO[l * 4 + v] - want) > 1e-3f * fabsf(want)) { if (bad < 5) printf("lane %d v %d got %g want %g\n", l, v, hO[l*4+v], want); ++bad; }
    }
    printf("16x16x4 layout check: %d mismatches (C[i][j]: i = 4*(lane/16) + v, j = lane%%16; A[i=lane%%16][k=lane/16])\n", bad);
    return bad != 0;
}
