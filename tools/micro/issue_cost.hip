// Microbenchmark: what does it cost a wave that also issues MFMAs to bring 12 KiB per iteration into LDS, by LDS-DMA or by plain
// loads + ds_write_b128?  One wave per SIMD (256-thread workgroups, one per CU), 96 independent MFMAs per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o build/issue_cost tools/micro/issue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int MODE>   // 0 MFMA only, 1 LDS-DMA burst, 2 plain loads burst + ds_write one iteration later, 3 LDS-DMA one piece per 8 MFMAs
__global__ __launch_bounds__(256, 1) void k(const unsigned char* src, unsigned bytes, int iters, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)bytes, 0x00020000);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u4 r[12];
  for (int j = 0; j < 12; ++j) r[j] = u4{0u, 0u, 0u, 0u};
  unsigned chunk = blockIdx.x % 48;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned base = chunk * 49152u + wave * 12288u;
    chunk = chunk + 1 == 48 ? 0 : chunk + 1;
    if (MODE == 4 || MODE == 5 || MODE == 6) {   // the shape of mlp_fused's iteration head: barrier, 10 fragment reads, the burst, wait for the reads
      __builtin_amdgcn_s_barrier();
      u4 f[10];
      const unsigned la = (unsigned)(size_t)(lds_ptr)(smem + ((it + 1) & 1) * 49152 + lane * 16);
#pragma unroll
      for (int j = 0; j < 10; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[j]) : "v"(la), "n"(j * 2048));
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 12; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + j) * 1024), 16, lane * 16, base + j * 1024, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
      a[0] = (__bf16)(float)(f[0][0] + f[5][1] + f[9][2]);
      if (MODE == 6) {   // burst AFTER the wait for the reads
#pragma unroll
        for (int j = 0; j < 12; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + j) * 1024), 16, lane * 16, base + j * 1024, 0, 0);
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 12; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + j) * 1024), 16, lane * 16, base + j * 1024, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {   // write last iteration's data, then request this iteration's
        const unsigned ad = (unsigned)(size_t)(lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + j) * 1024 + lane * 16);
        asm volatile("s_waitcnt vmcnt(11)\n\tds_write_b128 %0, %1" ::"v"(ad), "v"(r[j]) : "memory");
        r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, base + j * 1024, 0);
      }
    }
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      if (MODE == 3) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + rep) * 1024), 16, lane * 16, base + rep * 1024, 0, 0);
        if (rep < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((it & 1) * 48 + wave * 12 + 8 + rep) * 1024), 16, lane * 16, base + (8 + rep) * 1024, 0, 0);
      }
    }
    if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += acc[i][0];
  if (MODE == 2) for (int j = 0; j < 12; ++j) s += (float)r[j][0];
  if (lane == 0) { out[blockIdx.x * 4 + wave] = t1 - t0; sink[blockIdx.x * 4 + wave] = s + smem[tid]; }
}

template <int MODE> void run(const unsigned char* d, unsigned bytes, unsigned long long* dout, float* sink, const char* name) {
  const int iters = 400, blocks = 256;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  k<MODE><<<blocks, 256, 96 * 1024>>>(d, bytes, 20, dout, sink);
  k<MODE><<<blocks, 256, 96 * 1024>>>(d, bytes, iters, dout, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
  printf("%-44s %8.0f cycles per iteration (96 MFMAs = 1536)\n", name, cyc / iters);
}

int main() {
  const unsigned bytes = 48 * 49152;
  unsigned char* d; unsigned long long* dout; float* sink;
  hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&dout, 8 * 1024); hipMalloc(&sink, 4 * 1024);
  run<0>(d, bytes, dout, sink, "MFMA only");
  run<1>(d, bytes, dout, sink, "+ 12 LDS-DMA pieces, burst");
  run<3>(d, bytes, dout, sink, "+ 12 LDS-DMA pieces, 1-2 per 12 MFMAs");
  run<2>(d, bytes, dout, sink, "+ 12 plain loads + 12 ds_write_b128");
  run<5>(d, bytes, dout, sink, "barrier + 10 ds_read_b128 + wait, no DMA");
  run<4>(d, bytes, dout, sink, "barrier + 10 reads + 12 DMA + wait");
  run<6>(d, bytes, dout, sink, "barrier + 10 reads + wait + 12 DMA");
  return 0;
}
