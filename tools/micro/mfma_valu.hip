// Microbenchmark: how much vector-ALU work hides behind the MFMAs of ONE wave per SIMD, by MFMA shape?
// Per iteration: the matrix work of an mlp_fused chunk per wave (96 x v_mfma_f32_16x16x32_bf16 or 48 x v_mfma_f32_32x32x16_bf16,
// same FLOPs, 1536 matrix-pipe cycles) with V GELU-like VALU instructions (clamp, fma, convert, fract, fma, mul on independent values)
// written between the MFMA groups.   hipcc --offload-arch=gfx950 -O3 -o build/mfma_valu tools/micro/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int VPG, int SGB = 0>   // SGB 1: sched_group_barrier asks for 1 MFMA : n VALU interleaving.  SHAPE 0: 16x16x32 (groups of 8 MFMAs), 1: 32x32x16 (groups of 4); VPG: GELU-like values per group
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long* out, float* sink, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  f32x4 acc4[12];
  f32x16 acc16[6];
  for (int i = 0; i < 12; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  float gv[16];
  for (int e = 0; e < 16; ++e) gv[e] = seed * (lane + e);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int grp = 0; grp < 12; ++grp) {
      if (SHAPE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc4[(grp * 8 + i) % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[(grp * 8 + i) % 12], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc16[(grp * 4 + i) % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[(grp * 4 + i) % 6], 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < VPG; ++v) {      // ~7 VALU per value, as GELU by table without the table read
        float x = gv[(grp * VPG + v) & 15];
        const float u = fmaf(__builtin_amdgcn_fmed3f(x, -8.0f, 7.984375f), 64.0f, 512.0f);
        const float fr = __builtin_amdgcn_fractf(u);
        const float t = (float)(int)u;
        x = x * fmaf(fr, 0.001f, t * 0.002f);
        gv[(grp * VPG + v) & 15] = x + 1.0f;
      }
      if (SGB) {
        constexpr int NM = SHAPE == 0 ? 8 : 4, NV = (VPG * 7 + NM - 1) / NM;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += acc4[i][0];
  for (int i = 0; i < 6; ++i) s += acc16[i][0];
  for (int e = 0; e < 16; ++e) s += gv[e];
  if (lane == 0) { out[blockIdx.x * 4 + wave] = t1 - t0; sink[blockIdx.x * 4 + wave] = s; }
}

template <int SHAPE, int VPG, int SGB = 0> void run(unsigned long long* dout, float* sink, const char* name) {
  const int iters = 400, blocks = 256;
  k<SHAPE, VPG, SGB><<<blocks, 256>>>(20, dout, sink, 0.37f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<SHAPE, VPG, SGB><<<blocks, 256>>>(iters, dout, sink, 0.37f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
  printf("%-64s %8.0f cycles per iteration, %7.1f ns (matrix pipe alone: 1536 cycles)\n", name, cyc / iters, ms * 1e6 / 20 / iters);
}

int main() {
  unsigned long long* dout; float* sink;
  hipMalloc(&dout, 8 * 1024); hipMalloc(&sink, 4 * 1024);
  run<0, 0>(dout, sink, "96 x 16x16x32, no VALU");
  run<0, 1>(dout, sink, "96 x 16x16x32 + 12 values (~84 VALU)");
  run<0, 2>(dout, sink, "96 x 16x16x32 + 24 values (~168 VALU)");
  run<0, 4>(dout, sink, "96 x 16x16x32 + 48 values (~336 VALU)");
  run<1, 0>(dout, sink, "48 x 32x32x16, no VALU");
  run<1, 1>(dout, sink, "48 x 32x32x16 + 12 values (~84 VALU)");
  run<1, 2>(dout, sink, "48 x 32x32x16 + 24 values (~168 VALU)");
  run<1, 4>(dout, sink, "48 x 32x32x16 + 48 values (~336 VALU)");
  run<0, 2, 1>(dout, sink, "96 x 16x16x32 + 24 values, sched_group_barrier 1:2");
  run<1, 2, 1>(dout, sink, "48 x 32x32x16 + 24 values, sched_group_barrier 1:4");
  run<1, 4, 1>(dout, sink, "48 x 32x32x16 + 48 values, sched_group_barrier 1:7");
  return 0;
}
