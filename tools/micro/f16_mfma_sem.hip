// Micro-test: numerics of v_mfma_f32_16x16x32_f16 on gfx950, the questions the split-operand (f16x4) mode rests on.
//   1. are f16 SUBNORMAL A/B inputs multiplied (not flushed)?
//   2. are the f16 x f16 products exact and how is the 32-term sum rounded (one rounding? a chain?)
//   3. does a three-plane split x = x0 + x1/2^11 + x2/2^22 reproduce an fp32 dot product to fp32 rounding?
//   hipcc --offload-arch=gfx950 -O3 -o build/f16_mfma_sem tools/micro/f16_mfma_sem.hip && build/f16_mfma_sem
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// D[16][16] = A[16][32] . B[16][32]^T + C; A row i = lane&15, k = (lane>>4)*8 + e; same for B (column j = lane&15)
__global__ void mm(const _Float16* A, const _Float16* B, float* D, int reps) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < reps; ++r) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = A[(size_t)r * 512 + (lane & 15) * 32 + (lane >> 4) * 8 + e];
      b[e] = B[(size_t)r * 512 + (lane & 15) * 32 + (lane >> 4) * 8 + e];
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  }
  // D row (lane>>4)*4 + e, column lane&15  (A supplies rows)
  for (int e = 0; e < 4; ++e) D[((lane >> 4) * 4 + e) * 16 + (lane & 15)] = acc[e];
}

static void run(const std::vector<_Float16>& A, const std::vector<_Float16>& B, std::vector<float>& D, int reps) {
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps);
  D.resize(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  hipFree(dA); hipFree(dB); hipFree(dD);
}

int main() {
  // ---- 1. subnormals: A[0][0] = 2^-20 (f16 subnormal), B[0][0] = 2^10 -> 2^-10 if multiplied, 0 if flushed
  {
    std::vector<_Float16> A(512, (_Float16)0.f), B(512, (_Float16)0.f);
    std::vector<float> D;
    A[0] = (_Float16)ldexpf(1.f, -20); B[0] = (_Float16)1024.f;
    A[32 + 1] = (_Float16)3.0f; B[32 + 1] = (_Float16)ldexpf(1.f, -24);   // row 1 / col 1: B subnormal (smallest)
    run(A, B, D, 1);
    printf("subnormal A: D[0][0] = %g (expect %g)\n", D[0], ldexpf(1.f, -10));
    printf("subnormal B: D[1][1] = %g (expect %g)\n", D[17], 3.0f * ldexpf(1.f, -24));
  }
  // ---- 2. summation: row 0: 1 + 31 x 2^-24 (each below half an ulp of 1): exact sum = 1 + 31*2^-24; a chain of fp32 adds gives 1
  {
    std::vector<_Float16> A(512, (_Float16)0.f), B(512, (_Float16)0.f);
    std::vector<float> D;
    for (int k = 0; k < 32; ++k) { A[k] = (_Float16)(k == 0 ? 1.0f : ldexpf(1.f, -12)); B[k] = (_Float16)(k == 0 ? 1.0f : ldexpf(1.f, -12)); }
    // row 1: big cancellation: 2048*2048 - 2048*2048 + 2^-20
    A[32 + 0] = (_Float16)2048.f; B[32 + 0] = (_Float16)2048.f; A[32 + 1] = (_Float16)-2048.f; B[32 + 1] = (_Float16)2048.f;
    A[32 + 2] = (_Float16)ldexpf(1.f, -10); B[32 + 2] = (_Float16)ldexpf(1.f, -10);
    run(A, B, D, 1);
    printf("sum 1 + 31*2^-24: D = %.10g  (exact %.10g, fp32 chain 1)\n", D[0], 1.0 + 31 * ldexp(1.0, -24));
    printf("cancellation: D[1][1] = %g (exact %g)\n", D[17], ldexpf(1.f, -20));
  }
  // ---- 3. split dot products: x, w fp32 random; planes as the engine's f16x4 mode builds them
  {
    const int reps = 12;   // K = 384 per plane
    const int K = 32 * reps;
    srand(1);
    std::vector<float> x(16 * K), w(16 * K);
    for (auto& v : x) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * (rand() % 4 == 0 ? 8.0 : 0.5));
    for (auto& v : w) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 0.05);
    const float S = 16384.f * 4;   // weight scale: max |w| * S < 2^15
    auto rtz = [](float v) { _Float16 h = (_Float16)v; if (fabsf((float)h) > fabsf(v)) { unsigned short u; memcpy(&u, &h, 2); u -= 1; memcpy(&h, &u, 2); } return h; };
    // planes: [x0 | x1*2^11 | x2*2^22 | x0] . [w0 | w0/2^11 | w0/2^22 | w1]
    std::vector<_Float16> A(4 * 16 * K), B(4 * 16 * K);
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < K; ++k) {
        const float xv = x[i * K + k];
        const _Float16 x0 = rtz(xv); const float r1 = xv - (float)x0;
        const _Float16 x1 = rtz(r1 * 2048.f); const float r2 = r1 * 2048.f - (float)x1;
        const _Float16 x2 = (_Float16)(r2 * 2048.f);
        const float wv = w[i * K + k] * S;
        const _Float16 w0 = (_Float16)wv; const _Float16 w1 = (_Float16)(wv - (float)w0);
        const _Float16 w0b = (_Float16)((float)w0 / 2048.f), w0c = (_Float16)((float)w0 / 2048.f / 2048.f);
        auto at = [&](int plane, int r, int kk) { const int kg = plane * K + kk; return (size_t)(kg / 32) * 512 + r * 32 + kg % 32; };
        A[at(0, i, k)] = x0; A[at(1, i, k)] = x1; A[at(2, i, k)] = x2; A[at(3, i, k)] = x0;
        B[at(0, i, k)] = w0; B[at(1, i, k)] = w0b; B[at(2, i, k)] = w0c; B[at(3, i, k)] = w1;
      }
    std::vector<float> D;
    run(A, B, D, 4 * reps);
    double e_split = 0, e_f32 = 0, mag = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0; float c = 0.f;
        for (int k = 0; k < K; ++k) { ref += (double)x[i * K + k] * (double)w[j * K + k]; c = fmaf(x[i * K + k], w[j * K + k], c); }
        e_split = fmax(e_split, fabs(D[i * 16 + j] / S - ref)); e_f32 = fmax(e_f32, fabs(c - ref)); mag = fmax(mag, fabs(ref));
      }
    printf("split f16x4 dot products, K = %d: max |err| %.3e   fp32 fma chain: %.3e   (max |value| %.3f, fp32 ulp there %.3e)\n", K, e_split, e_f32, mag, mag * ldexp(1.0, -23));
  }
  return 0;
}
