// Microbenchmark: what of an mlp_fused chunk step overlaps with its 96 MFMAs when ONE wave per SIMD runs it?
// Per iteration and wave: 48 MFMAs into 48 distinct accumulators ("GEMM2") + 48 into 4 ("GEMM1"), optionally with
//   L: their 48 A fragments read from LDS (ds_read_b128, groups of 4, two groups ahead, counted waits)
//   V: the GELU's vector-ALU work on 16 values (as in the kernel, without the table)
//   T: the 16 table reads (ds_read_b64 at data-dependent addresses)
//   R: the GELU input taken from the GEMM1 accumulators (v_accvgpr_read when they live in AGPRs)
//   B: a workgroup barrier per iteration
// hipcc --offload-arch=gfx950 -O3 -o build/mlp_loop tools/micro/mlp_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define WAITF(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))

template <int L, int V, int T, int R, int B>
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long* out, float* sink, float seed) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 56 * 1024 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = seed * (float)((i * 2654435761u) >> 20);
  bf16x8 xb[2], hb[2];
  for (int e = 0; e < 8; ++e) { xb[0][e] = (__bf16)(0.01f * ((lane * 7 + e) % 13)); xb[1][e] = (__bf16)(0.02f * ((lane + 3 * e) % 11)); hb[0][e] = xb[1][e]; hb[1][e] = xb[0][e]; }
  f32x4 acc2[48], accn[4];
  for (int i = 0; i < 48; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) accn[i] = f32x4{seed, 0.1f, -0.2f, 0.3f};
  float gv[16];
  for (int e = 0; e < 16; ++e) gv[e] = seed * (lane + e);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const unsigned fa = lds0 + (unsigned)lane * 16u;           // conflict-free 1-KiB fragments
  const unsigned lut = lds0 + 48 * 1024;
  bf16x8 f0[4], f1[4], f2[4];
  for (int e = 0; e < 4; ++e) { f0[e] = xb[0]; f1[e] = xb[1]; f2[e] = xb[0]; }
  float2 t4[4];
  for (int e = 0; e < 4; ++e) t4[e] = make_float2(0.5f, 0.001f);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#define G(dst, n) if (L) { RD128(dst[0], fa, (4 * (n) + 0) * 1024); RD128(dst[1], fa, (4 * (n) + 1) * 1024); RD128(dst[2], fa, (4 * (n) + 2) * 1024); RD128(dst[3], fa, (4 * (n) + 3) * 1024); }
#define W(n, f) if (L) WAITF(n, f);
#define M1(src, n) _Pragma("unroll") for (int i = 0; i < 8; ++i) accn[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[i >> 1], xb[i & 1], accn[i & 3], 0, 0, 0);
#define M2(src, n) _Pragma("unroll") for (int i = 0; i < 8; ++i) acc2[(8 * (n) + i) % 48] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[i >> 1], hb[i & 1], acc2[(8 * (n) + i) % 48], 0, 0, 0);
#define GEL(q)                                                                                                  \
  if (V) {                                                                                                     \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                             \
      const float x = R ? accn[q][e] * 0.001f : gv[4 * (q) + e];                                                \
      const float u = fmaf(__builtin_amdgcn_fmed3f(x, -8.0f, 7.984375f), 64.0f, 512.0f);                        \
      if (T) { const unsigned ad = lut + ((unsigned)(int)u << 3); asm volatile("ds_read_b64 %0, %1" : "=v"(t4[e]) : "v"(ad)); } \
      if (T) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t4[e]));                                                \
      gv[4 * (q) + e] = x * fmaf(__builtin_amdgcn_fractf(u), t4[e].y, t4[e].x) + 0.37f;                         \
    }                                                                                                          \
  }
  for (int it = 0; it < iters; ++it) {
    if (B) __builtin_amdgcn_s_barrier();
    G(f0, 0) G(f1, 1)
    if (L) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]));
    G(f2, 2)  GEL(0)  M1(f0, 0)
    G(f0, 3)  W(8, f1)  M2(f1, 0)
    G(f1, 4)  W(8, f2)  M1(f2, 1)
    G(f2, 5)  GEL(1)  W(8, f0)  M2(f0, 1)
    G(f0, 6)  W(8, f1)  M1(f1, 2)
    G(f1, 7)  W(8, f2)  M2(f2, 2)
    G(f2, 8)  GEL(2)  W(8, f0)  M1(f0, 3)
    G(f0, 9)  W(8, f1)  M2(f1, 3)
    G(f1, 10) W(8, f2)  M1(f2, 4)
    G(f2, 11) GEL(3)  W(8, f0)  M2(f0, 4)
    W(4, f1)  M1(f1, 5)
    W(0, f2)  M2(f2, 5)
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 48; ++i) s += acc2[i][0];
  for (int i = 0; i < 4; ++i) s += accn[i][1];
  for (int e = 0; e < 16; ++e) s += gv[e];
  if (lane == 0) { out[blockIdx.x * 4 + wave] = t1 - t0; sink[blockIdx.x * 4 + wave] = s; }
}

template <int L, int V, int T, int R, int B> void run(unsigned long long* dout, float* sink, const char* name) {
  const int iters = 400, blocks = 256, lds = 57 * 1024;
  hipFuncSetAttribute((const void*)k<L, V, T, R, B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k<L, V, T, R, B><<<blocks, 256, lds>>>(20, dout, sink, 0.37f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<L, V, T, R, B><<<blocks, 256, lds>>>(iters, dout, sink, 0.37f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
  printf("%-58s %6.0f cycles per step, %7.1f ns\n", name, cyc / iters, ms * 1e6 / 20 / iters);
}

int main() {
  unsigned long long* dout; float* sink;
  hipMalloc(&dout, 8 * 1024); hipMalloc(&sink, 4 * 1024);
  run<0, 0, 0, 0, 0>(dout, sink, "96 MFMAs (48 into 48 accumulators + 48 into 4)");
  run<1, 0, 0, 0, 0>(dout, sink, "+ 48 fragment reads from LDS");
  run<0, 1, 0, 0, 0>(dout, sink, "+ GELU VALU on 16 values");
  run<0, 1, 0, 1, 0>(dout, sink, "+ GELU VALU, input from the GEMM1 accumulators");
  run<1, 1, 0, 0, 0>(dout, sink, "+ fragment reads + GELU VALU");
  run<1, 1, 1, 0, 0>(dout, sink, "+ fragment reads + GELU VALU + table reads");
  run<1, 1, 1, 1, 0>(dout, sink, "+ fragment reads + GELU + table + accumulator input");
  run<1, 1, 1, 1, 1>(dout, sink, "+ all of it + barrier");
  return 0;
}
