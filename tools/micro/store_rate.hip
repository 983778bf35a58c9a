// Microbenchmark: what the STORE path of MI355X sustains for the store shapes of the detector's HBM-bound kernels:
//   plain     every wave instruction writes one contiguous KiB (64 lanes x 16 B): the guide's float4-copy shape
//   c1split   conv1_split_kernel's: a lane holds 8 consecutive channels (16 B) of one pixel and plane; a wave instruction writes 16 pixels x 64 B
//             (four lanes side by side), the pixels 256 B apart (pixel row = [x0 (128 B) | x1 (128 B)]); the four instructions (t, plane) of a
//             16-pixel block fill its 4 KiB
//   head      the packed-pairs head (conv3p<32,NP=2>): pixel row [x0 (64 B) | x1 (64 B)] = 128 B; a wave instruction writes 16 pixels x 64 B,
//             128 B apart; two instructions fill a 2-KiB block
//   lines     the c1split bytes re-ordered so that an instruction writes WHOLE 128-byte lines (8 lanes side by side, 8 pixel-planes per
//             instruction): what an LDS transpose in front of the stores would buy
// each with default-policy and nontemporal stores, 1 - 4 workgroups of 256 threads per CU, and on 8 / 32 / 64 / 128 / 256 CUs (one workgroup each) to
// separate a per-CU limit from a chip-wide one.  Nothing is read; a launch writes ~1.6 GB (the full-chip cases), larger than the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o build/store_rate tools/micro/store_rate.hip && build/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PAT, bool NT>
__global__ __launch_bounds__(256) void k(unsigned char* out, long long blocks4k, f32x4 v, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
  const long long nw = (long long)gridDim.x * 4, w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (long long b = w0; b < blocks4k; b += nw) {   // one 4-KiB block (16 pixels of conv1_split's tensor) per wave and step, four store instructions
    unsigned char* base = out + b * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned char* p;
      if (PAT == 0) p = base + j * 1024 + lane * 16;                                  // plain
      else if (PAT == 1) p = base + fr * 256 + (j >> 1) * 128 + (j & 1) * 64 + fg * 16;   // c1split: j = (plane, t)
      else if (PAT == 2) p = base + (j >> 1) * 2048 + fr * 128 + (j & 1) * 64 + fg * 16;  // head: two 2-KiB blocks, j & 1 = plane
      else p = base + j * 1024 + (lane >> 3) * 128 + (lane & 7) * 16;                 // lines (= plain's bytes, lanes grouped by line)
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
      else *reinterpret_cast<f32x4*>(p) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[w0] = t1 - t0;
}

// a matrix-core burn (random-ish f16 operands, every SIMD busy) of ~`iters` x 64 MFMAs per wave: the neighbour a detector kernel has in the engine (conv3p at the power wall)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__global__ __launch_bounds__(512) void burn(float* sink, int iters) {
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.37f * ((threadIdx.x * 7 + e * 13) % 61) - 9.f); b[e] = (_Float16)(0.21f * ((threadIdx.x * 11 + e * 5) % 53) - 5.f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    a[it & 7] = (_Float16)((float)a[it & 7] * -1.0009765625f);
  }
  float sum = 0.f;
  for (int j = 0; j < 8; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (sum == 1.2345e30f) sink[0] = sum;
}

template <int PAT, bool NT> void run_after_burn(const char* name, unsigned char* d, size_t bytes, int grid, unsigned long long* dcyc, float* sink, int burn_iters) {
  const long long blocks = (long long)(bytes / 4096);
  hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  float best = 1e30f, burn_ms = 0.f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(c);
    burn<<<512, 512>>>(sink, burn_iters);
    hipEventRecord(a);
    k<PAT, NT><<<grid, 256>>>(d, blocks, v, dcyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
    hipEventElapsedTime(&burn_ms, c, a);
  }
  printf("%-8s %-3s grid %4d behind a %6.2f ms matrix-core burn: %7.1f us  %6.2f TB/s\n", name, NT ? "nt" : "", grid, burn_ms, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

template <int PAT, bool NT> void run(const char* name, unsigned char* d, size_t bytes, int grid, unsigned long long* dcyc) {
  const long long blocks = (long long)(bytes / 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  k<PAT, NT><<<grid, 256>>>(d, blocks, v, dcyc);
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(a);
    k<PAT, NT><<<grid, 256>>>(d, blocks, v, dcyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long mx = 0; for (auto c : h) if (c > mx) mx = c;
  const int cus = grid < 256 ? grid : 256;
  printf("%-8s %-3s grid %4d (%d per CU on %3d CUs)  %8.1f MB  %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (in-kernel clock)\n", name, NT ? "nt" : "", grid, (grid + 255) / 256, cus,
         bytes / 1e6, best * 1e3, bytes / (best * 1e-3) / 1e12, (double)bytes / cus / (double)mx);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const size_t full = (size_t)1610612736;   // 8 pages x 1024 x 768 x 256 B: conv1_split's output of one group
  unsigned char* d; unsigned long long* dcyc;
  hipMalloc(&d, full); hipMalloc(&dcyc, 8 * 4096 * 4);
  printf("# store shapes at full chip, 1 - 4 workgroups of 256 threads per CU\n");
  for (int per : {1, 2, 3, 4, 8, 16}) {
    run<0, false>("plain", d, full, 256 * per, dcyc); run<0, true>("plain", d, full, 256 * per, dcyc);
    run<1, false>("c1split", d, full, 256 * per, dcyc); run<1, true>("c1split", d, full, 256 * per, dcyc);
    run<2, false>("head", d, full, 256 * per, dcyc); run<2, true>("head", d, full, 256 * per, dcyc);
    run<3, false>("lines", d, full, 256 * per, dcyc); run<3, true>("lines", d, full, 256 * per, dcyc);
  }
  printf("# fewer CUs, one workgroup each (bytes scaled with the CU count): a per-CU limit shows as constant B/clk/CU\n");
  for (int cus : {8, 32, 64, 128, 256}) {
    const size_t bytes = full / 256 * cus;
    run<0, false>("plain", d, bytes, cus, dcyc); run<1, false>("c1split", d, bytes, cus, dcyc); run<0, true>("plain", d, bytes, cus, dcyc);
  }
  printf("# a burst per CU: 128 KiB per workgroup (fc1's tile: 131 KB of hidden pairs), every CU at once, one workgroup of 256 threads per CU\n");
  for (int cus : {8, 64, 256}) {
    const size_t bytes = (size_t)131072 * cus;
    run<0, false>("plain", d, bytes, cus, dcyc); run<1, false>("c1split", d, bytes, cus, dcyc);
  }
  printf("# the same store kernels right behind a matrix-core burn on every CU (the chip's clocks as a power-bound neighbour leaves them)\n");
  float* sink; hipMalloc(&sink, 64);
  for (int iters : {2000, 20000, 100000}) {
    run_after_burn<0, false>("plain", d, full, 4096, dcyc, sink, iters);
    run_after_burn<1, false>("c1split", d, full, 4096, dcyc, sink, iters);
    run_after_burn<2, false>("head", d, full, 4096, dcyc, sink, iters);
  }
  return 0;
}
