// Microbenchmark: LDS-DMA (buffer_load ... lds) rate of ONE workgroup per CU against the bytes it keeps in flight - the question behind the
// single-page (latency) regime of gemm_sp.hip / conv3p.hip: does a lone workgroup's K loop go at bytes-in-flight / round-trip time?
// Pieces of 8 rows x 128 B (the kernels' loader pattern) or linear 1 KiB; source buffer of <argv[1]> KiB (2400: L2-resident; 65536: beyond the L2s).
//   hipcc --offload-arch=gfx950 -O3 -o build/dma_depth tools/micro/dma_depth.hip && build/dma_depth 2400
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int PAT, int BATCH>   // BATCH pieces per wave in flight, then a full wait
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned bytes, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, (int)bytes, 0x00020000);
  const unsigned npieces = bytes / 1024;
  unsigned pc = (blockIdx.x * 977u + wave) % npieces;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const unsigned piece = pc % npieces;
      const unsigned vo = PAT == 0 ? piece * 1024 + lane * 16 : (unsigned)(((unsigned long long)(piece * 8 + (lane >> 3)) * 2304ull) % (bytes - 1024)) / 16 * 16 + (lane & 7) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + ((wave * BATCH + j) % 152) * 1024), 16, vo, 0, 0, 0);
      pc += nw * 13;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
}

template <int PAT, int BATCH> void run(const unsigned char* d, unsigned bytes, int threads, unsigned long long* dout) {
  const int iters = 400 / BATCH + 20, blocks = 256;
  hipFuncSetAttribute((const void*)k<PAT, BATCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<PAT, BATCH><<<blocks, threads, 152 * 1024>>>(d, bytes, 5, dout);
  hipEventRecord(a);
  k<PAT, BATCH><<<blocks, threads, 152 * 1024>>>(d, bytes, iters, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double per_cu = (double)iters * BATCH * 1024 * (threads / 64);
  printf("%-12s waves %d  in flight %3d KiB/CU : %6.1f GB/s per CU   round %6.2f us\n", PAT ? "8 x 128 B" : "linear KiB", threads / 64, BATCH * (threads / 64),
         per_cu / (ms * 1e6), ms * 1e3 / iters);
}

int main(int argc, char** argv) {
  const unsigned bytes = (argc > 1 ? atoi(argv[1]) : 2400) * 1024u;
  unsigned char* d; unsigned long long* dout;
  hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&dout, 8 * 4096);
  printf("source buffer %u KiB, one workgroup per CU\n", bytes / 1024);
  for (int threads : {256, 512}) {
    run<1, 1>(d, bytes, threads, dout); run<1, 2>(d, bytes, threads, dout); run<1, 4>(d, bytes, threads, dout); run<1, 8>(d, bytes, threads, dout);
    run<1, 16>(d, bytes, threads, dout); run<1, 32>(d, bytes, threads, dout);
    run<0, 4>(d, bytes, threads, dout); run<0, 16>(d, bytes, threads, dout);
  }
  return 0;
}
