// Layout and issue-rate probe of v_mfma_f32_32x32x16_bf16 against v_mfma_f32_32x32x2_f32 (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_bf16_probe.hip && /tmp/probe
// measured on MI355X: layout A[i = lane & 31][k = 8 (lane >> 5) + e], B[k][n = lane & 31] confirmed; 2099 vs 144 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// layout probe: A[i][k] = i + 100*k (as bf16-exact small ints?), B[k][j]: identity-like selections
__global__ void probe(const float* A, const float* B, float* D) {   // A [32][16], B [16][32], D [32][32]
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        int k = 8 * (l >> 5) + j;
        a[j] = (__bf16)A[(l & 31) * 16 + k];
        b[j] = (__bf16)B[k * 32 + (l & 31)];
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = c[r];
    }
}
// throughput: N back-to-back MFMAs on 4 independent accumulators per wave
template <int KIND>
__global__ void rate(float* out, int iters) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    float fa = threadIdx.x, fb = 1.5f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 17 - 8);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 11) % 13 - 6);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    float md = 0; for (int i = 0; i < 1024; ++i) md = fmaxf(md, fabsf(D[i] - R[i]));
    printf("layout probe: max |D - ref| = %g\n", md);
    float* o; hipMalloc(&o, 1024 * 256 * 4);
    for (int kind = 0; kind < 2; ++kind) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        int iters = 4000;
        if (kind == 0) rate<0><<<1024, 256>>>(o, 10); else rate<1><<<1024, 256>>>(o, 10);
        hipEventRecord(e0);
        if (kind == 0) rate<0><<<1024, 256>>>(o, iters); else rate<1><<<1024, 256>>>(o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 1024.0 * 4 * iters * 4 * (kind == 0 ? 32768.0 : 4096.0);
        printf("%s: %.3f ms, %.1f TFLOP/s\n", kind == 0 ? "mfma_f32_32x32x16_bf16" : "mfma_f32_32x32x2_f32", ms, flop / ms / 1e9);
    }
    return 0;
}
