// Microbenchmark: per-CU throughput of LDS-DMA (buffer_load_dwordx4 ... lds) from an L2-resident buffer for different
// source address patterns and wave counts.  hipcc --offload-arch=gfx950 -O3 -o build/dma_bench tools/micro/dma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int PAT>
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned bytes, int iters, unsigned long long* out, int depth_mode) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)bytes, 0x00020000);
  const unsigned npieces = bytes / 1024;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned pc = wave;   // piece counter
  if (PAT >= 5) {   // plain loads to registers, then ds_write_b128 (PAT 5: linear 1 KiB; 6: 8 rows x 128 B)
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    for (int it = 0; it < iters; ++it) {
      u4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned piece = pc % npieces;
        const unsigned vo = PAT == 5 ? piece * 1024 + lane * 16 : ((piece * 8 + (lane >> 3)) * 768 % (bytes - 1024)) / 16 * 16 + (lane & 7) * 16;
        r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
        pc += nw;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const unsigned a = (unsigned)(size_t)(lds_ptr)(smem + (wave * 8 + j) * 1024 + lane * 16); asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(r[j])); }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
    return;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned piece = pc % npieces;
      unsigned vo;
      if (PAT == 0) vo = piece * 1024 + lane * 16;
      else if (PAT == 1) vo = piece * 1024 + (((lane & ~15) | ((lane & 15) ^ (piece & 15))) * 16);
      else if (PAT == 2) vo = ((piece * 8 + (lane >> 3)) * 768 % (bytes - 1024)) / 16 * 16 + (lane & 7) * 16;
      else if (PAT == 3) vo = ((piece * 8 + (lane >> 3)) * 768 % (bytes - 1024)) / 16 * 16 + (((lane & 7) ^ ((lane >> 4) & 7)) * 16);
      else vo = piece * 1024 + (((lane & ~3) | ((lane & 3) ^ ((lane >> 3) & 3))) * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * 8 + j) * 1024), 16, vo, 0, 0, 0);
      pc += nw;
    }
    if (depth_mode == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
}

template <int PAT> void run(const unsigned char* d, unsigned bytes, int threads, int depth, unsigned long long* dout, const char* name) {
  const int iters = 200, blocks = 256;
  const size_t lds = (threads / 64) * 8 * 1024;
  hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PAT><<<blocks, threads, lds>>>(d, bytes, 20, dout, depth);
  hipEventRecord(a);
  k<PAT><<<blocks, threads, lds>>>(d, bytes, iters, dout, depth);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : h) cyc += v; cyc /= h.size();
  const double per_cu_bytes = (double)iters * 8 * 1024 * (threads / 64);
  printf("%-34s waves/CU %d depth %s : %7.1f B/clk/CU  (%6.1f cycles per 1-KiB piece per CU)  %6.1f GB/s/CU  chip %5.2f TB/s\n", name, threads / 64,
         depth ? "16" : "8 ", per_cu_bytes / cyc, cyc / (iters * 8.0 * (threads / 64)), per_cu_bytes / (ms * 1e6), per_cu_bytes * blocks / (ms * 1e9));
}

int main(int argc, char** argv) {
  const unsigned bytes = argc > 1 ? atoi(argv[1]) * 1024 : 2400 * 1024;
  unsigned char* d; unsigned long long* dout;
  hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&dout, 8 * 4096);
  printf("source buffer %u KiB (shared by all 256 workgroups)\n", bytes / 1024);
  for (int threads : {256, 512})
    for (int depth : {0, 1}) {
      run<0>(d, bytes, threads, depth, dout, "linear 1 KiB");
      run<1>(d, bytes, threads, depth, dout, "xor-16 inside 256-B windows");
      run<2>(d, bytes, threads, depth, dout, "8 rows x 128 B (stride 768)");
      run<3>(d, bytes, threads, depth, dout, "8 rows x 128 B, chunk xor");
      run<4>(d, bytes, threads, depth, dout, "xor-4 inside 64-B rows");
      if (depth == 0) { run<5>(d, bytes, threads, depth, dout, "plain loads + ds_write, linear"); run<6>(d, bytes, threads, depth, dout, "plain loads + ds_write, 8x128"); }
    }
  return 0;
}
