// Hardware probe (run on the GPU box): operand/accumulator lane mapping of
// v_mfma_f32_32x32x16_f16 and whether fp16 subnormal operands are honoured.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const _Float16* A, const _Float16* B, float* D) {  // A[32][16], B[16][32], D[32][32]
    const int l = threadIdx.x, lr = l & 31, lk = l >> 5;
    half8 a, b;
    for (int t = 0; t < 8; ++t) {
        a[t] = A[lr * 16 + lk * 8 + t];
        b[t] = B[(lk * 8 + t) * 32 + lr];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + lr] = c[r];
}

int main() {
    _Float16 hA[512], hB[512];
    float hD[1024], ref[1024];
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    // 1) layout: asymmetric small integers
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (_Float16)(float)((i * 3 + k * 5) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (_Float16)(float)((k * 2 + j * 11) % 9 - 4);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += (float)hA[i * 16 + k] * (float)hB[k * 32 + j]; ref[i * 32 + j] = s; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    float e = 0; for (int i = 0; i < 1024; ++i) e = fmaxf(e, fabsf(hD[i] - ref[i]));
    printf("LAYOUT max_err=%g (0 => A[i=l&31][k=(l>>5)*8+t], B[k=(l>>5)*8+t][j=l&31], C row=(r&3)+8*(r>>2)+4*(l>>5) col=l&31)\n", e);
    // 2) subnormal fp16 operands: A = 2^-20 (fp16 subnormal), B = 2^10 -> sum over 16 k = 2^-6 if honoured
    for (int i = 0; i < 512; ++i) { hA[i] = (_Float16)ldexpf(1.f, -20); hB[i] = (_Float16)1024.f; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("DENORM_A result=%g expected_if_honoured=%g\n", hD[5], ldexpf(1.f, -6));
    for (int i = 0; i < 512; ++i) { hB[i] = (_Float16)ldexpf(1.f, -20); hA[i] = (_Float16)1024.f; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    printf("DENORM_B result=%g expected_if_honoured=%g\n", hD[5], ldexpf(1.f, -6));
    return 0;
}
