// CRAFT's upconv4.0, skip half (inside the TorchScript module run at tuatara.cpp:376): ReLU(W_skip . skip + b + upsample2x(z)) - a 1x1 convolution over the 128-channel skip
// tensor (activation pairs, 512 bytes per pixel) to 64 channels with the half-resolution addend's bilinear upsample in its epilogue - persistent, weights resident
// (gfx950 / MI355X).
//
// On gemm2.hip's loop this layer is 12 288 tiles of 128 x 64 with two k steps each: per tile the whole weight matrix (48 KB) streams through L2 -> LDS again beside 64 KB of
// activations, the four-tap gather of z waits out an L2 round trip behind the K loop, and the next tile's first wait also waits for this tile's stores (vmcnt counts loads and
// stores together): 366 us per 8-page launch where the layer's bytes take 211 (profiles/r06_pmc_craft_x4.json: 3.5 TB/s).  Here the weights (w0, w0 / 2^11, w1: 48 KB as ready
// MFMA A fragments, the staged planes gemm2.hip multiplies) are fetched once per workgroup; a tile is 64 consecutive pixels: ONE burst of 32 LDS-DMA pieces, the z gather of the tile issued
// in FRONT of the wait for that burst (inline asm loads, counted waits: the gather's latency runs under the burst and the MFMAs), 48 MFMAs per wave, and the next tile's burst
// requested in front of the epilogue.  Two workgroups share a CU (80 KB of LDS each).
//
// Same arithmetic in the same order as gemm2.hip's pairs loop and its up_z epilogue, so that heat maps stay bit-identical (tuning key "up_resident", test): per accumulator and
// 64-deep k step (x0 w0) (x0 w1) (x1 w0b), each over its two 32-deep halves in order; epilogue acc * out_scale + bias + [ly0 (lx0 a + lx1 b) + ly1 (lx0 c + lx1 d)] with the
// same fused multiply-adds, ReLU, pair split.
#include <algorithm>
#include <stdexcept>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int UBM = 64;                         // pixels per tile
constexpr int UK = 128, UN = 64;                // input channels (per plane), output channels
constexpr int UXBYTES = 4 * UBM * 128;          // four sub-tiles (plane, k step) of 64 rows x 128 B: gemm2.hip's LDS image per sub-tile
constexpr int UWFRAGS = 2 * 3 * 2 * 4;          // [k step][w0 | w0b | w1][32-deep half][channel tile of 16]: one KiB each
constexpr int ULDS = UXBYTES + UWFRAGS * 1024;  // 81920: two workgroups per CU
}  // namespace

__global__ __launch_bounds__(256, 2) void conv1u_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const xs = smem;               // sub-tile s = 2 * plane + k step at s * 8192: row r at r * 128, 16-byte chunk c at position c ^ ((r >> 1) & 7)
  unsigned char* const wl = smem + UXBYTES;     // fragment f = ((k0 * 3 + plane) * 2 + kk) * 4 + j at f * 1024 + lane * 16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ntiles = p.M / UBM, per_xcd = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, t_first = xcd * per_xcd + (int)(blockIdx.x >> 3), t_end = min(ntiles, (xcd + 1) * per_xcd), t_step = (int)(gridDim.x >> 3);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)(unsigned)((size_t)p.M * 512), 0x00020000);

  // ---- the weights, once per workgroup.  A tile j, row q = channel 32 (j >> 1) + (q >> 2) * 8 + (j & 1) * 4 + (q & 3) (gemm2.hip's row permutation: a lane ends with channels
  // 32 t + 8 fg .. + 7 of its pixel in acc[2 t], acc[2 t + 1]); weight rows are [w0 | w0 / 2^11 | w1] of 128 halves each
  {
    const f16* wbase = reinterpret_cast<const f16*>(p.wgt);
    for (int f = wave; f < UWFRAGS; f += 4) {
      const int j = f & 3, kk = (f >> 2) & 1, pl = (f >> 3) % 3, k0 = f / 24;
      const int n = 32 * (j >> 1) + (fr >> 2) * 8 + (j & 1) * 4 + (fr & 3);
      const f16x8 w = *reinterpret_cast<const f16x8*>(wbase + (size_t)n * (3 * UK) + pl * UK + k0 * 64 + kk * 32 + fg * 8);
      *reinterpret_cast<f16x8*>(wl + f * 1024 + lane * 16) = w;
    }
  }
  float bv[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[t][e] = p.bias ? p.bias[t * 32 + fg * 8 + e] : 0.f;
  RangeWatch rw;

  // a lane's share of a tile's burst: piece q = 8 s + 2 i' ... : sub-tile s = piece >> 3, rows 8 (piece & 7) .. + 7; this lane row r = 8 (piece & 7) + (lane >> 3),
  // source chunk (lane & 7) ^ ((r >> 1) & 7) of the sub-tile's 128 bytes of pixel r: byte r * 512 + plane * 256 + k0 * 128 + chunk * 16
  int rel[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = i * 4 + wave, s = piece >> 3, r = (piece & 7) * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    rel[i] = r * 512 + (s >> 1) * 256 + (s & 1) * 128 + g * 16;
  }
  auto request_tile = [&](int tile, bool live) {   // (not live: eight out-of-range requests - zero fill, no traffic - so that every tile issues the same eight)
    const unsigned base = (unsigned)tile * (unsigned)(UBM * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(xs + (i * 4 + wave) * 1024), 16, live ? base + (unsigned)rel[i] : 0x80000000u, 0, 0, 0);
  };
  // fragment addressing (gemm2.hip): row = 16 wave + fr, (row >> 1) & 7 == (lane >> 1) & 7
  const int frag = (wave * 16 + fr) * 128 + ((fg ^ ((lane >> 1) & 7)) << 4);
  const int Hl = p.H >> 1, Wl = p.W >> 1;
  typedef __attribute__((ext_vector_type(4))) float f4;

  __syncthreads();                               // the weight fragments are written with ds_write
  if (t_first < t_end) request_tile(t_first, true);
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const bool has_next = tile + t_step < t_end;
    // ---- this tile's z gather, in front of the wait for its burst: row m = 64 tile + 16 wave + fr; taps and weights as gemm2.hip's up_z pass
    const int m = tile * UBM + wave * 16 + fr;
    const int xo = m % p.W, tq = m / p.W, yo = tq % p.H, bq = tq / p.H;
    const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0), x1 = x0 + (x0 < Wl - 1 ? 1 : 0);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
    const int64_t pb = (int64_t)bq * Hl * Wl;
    const float* z00 = p.up_z + (pb + (int64_t)y0 * Wl + x0) * p.up_ld + fg * 8;
    const float* z01 = p.up_z + (pb + (int64_t)y0 * Wl + x1) * p.up_ld + fg * 8;
    const float* z10 = p.up_z + (pb + (int64_t)y1 * Wl + x0) * p.up_ld + fg * 8;
    const float* z11 = p.up_z + (pb + (int64_t)y1 * Wl + x1) * p.up_ld + fg * 8;
    f4 ga[2][2], gb[2][2], gc[2][2], gd[2][2];   // [t][h]: channels 32 t + 8 fg + 4 h .. + 3 of the four taps
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ga[t][h]) : "v"(z00 + t * 32 + 4 * h) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gb[t][h]) : "v"(z01 + t * 32 + 4 * h) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gc[t][h]) : "v"(z10 + t * 32 + 4 * h) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gd[t][h]) : "v"(z11 + t * 32 + 4 * h) : "memory");
      }
    // the 16 gather loads are the youngest operations: everything older - this tile's burst, the previous tile's stores - is done at vmcnt(16)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k0 = 0; k0 < 2; ++k0) {
      f16x8 x0f[2], x1f[2], w0[2][4], w0b[2][4], w1[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        x0f[kk] = *reinterpret_cast<const f16x8*>(xs + (0 * 2 + k0) * 8192 + (frag ^ (kk * 64)));
        x1f[kk] = *reinterpret_cast<const f16x8*>(xs + (1 * 2 + k0) * 8192 + (frag ^ (kk * 64)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          w0[kk][j] = *reinterpret_cast<const f16x8*>(wl + (((k0 * 3 + 0) * 2 + kk) * 4 + j) * 1024 + lane * 16);
          w0b[kk][j] = *reinterpret_cast<const f16x8*>(wl + (((k0 * 3 + 1) * 2 + kk) * 4 + j) * 1024 + lane * 16);
          w1[kk][j] = *reinterpret_cast<const f16x8*>(wl + (((k0 * 3 + 2) * 2 + kk) * 4 + j) * 1024 + lane * 16);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[kk][j], x0f[kk], acc[j], 0, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[kk][j], x0f[kk], acc[j], 0, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b[kk][j], x1f[kk], acc[j], 0, 0, 0);
    }
    __syncthreads();                             // every wave has read its fragments of this tile
    request_tile(tile + t_step, has_next);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the gather has landed once all but the 8 requests just issued are done.  ONE wait on one code path: with a branch around it the compiler may copy the gather's
    // registers on the way into a branch - before the data is there (the last tile of every workgroup came out wrong that way)
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(ga[0][0]), "+v"(ga[0][1]), "+v"(ga[1][0]), "+v"(ga[1][1]), "+v"(gb[0][0]), "+v"(gb[0][1]), "+v"(gb[1][0]), "+v"(gb[1][1]),
                 "+v"(gc[0][0]), "+v"(gc[0][1]), "+v"(gc[1][0]), "+v"(gc[1][1]), "+v"(gd[0][0]), "+v"(gd[0][1]), "+v"(gd[1][0]), "+v"(gd[1][1]) : : "memory");
    // ---- epilogue (gemm2.hip's up_z pass and store, NP = 3)
    const float lx1 = wx, lx0 = 1.f - lx1, ly1 = wy, ly0 = 1.f - ly1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float top = fmaf(lx1, gb[t][h][e], lx0 * ga[t][h][e]), bot = fmaf(lx1, gd[t][h][e], lx0 * gc[t][h][e]);
          v[4 * h + e] = fmaf(acc[2 * t + h][e], p.out_scale, bv[t][4 * h + e]) + fmaf(ly1, bot, ly0 * top);
        }
      if (p.act == kActRelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      st_split_n(p.out, (int64_t)m, p.out_ld, t * 32 + fg * 8, v, p.out_planes, rw);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile's dead requests target this workgroup's LDS
  rw.flush(p.range_flag, p.range_tag);
}

// gemm2.hip's split case this kernel replaces: pairs, 1x1, one source of 128 channels, 64 outputs, the half-resolution addend, planes out
bool conv1u_eligible(const ConvParams& p) {
  if (p.split != 3 || p.ks != 1 || p.C0 != UK || p.C1 != 0 || p.Cout != UN || !p.up_z || p.up_ld < UN || p.up_ld % 4) return false;
  if (p.M % UBM || p.M != p.B * p.H * p.W || (p.H & 1) || (p.W & 1)) return false;
  if (!p.out || p.out_planes != 2 || p.out_ld != UN || p.out_relu || p.out_pool || p.out_f32 || p.resid || p.relu0 || p.x_tiled || p.out_tiled) return false;
  if (p.act != kActRelu && p.act != kActNone) return false;
  if (((uintptr_t)p.in0 | (uintptr_t)p.wgt | (uintptr_t)p.up_z | (uintptr_t)p.out) & 15) return false;
  return (size_t)p.M * 512 < ((size_t)1 << 31);
}

void launch_conv1u(const ConvParams& p_in, hipStream_t s) {
  if (!conv1u_eligible(p_in)) throw std::runtime_error("conv1u: shape not supported");
  const ConvParams p = with_range_ctx(p_in);
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)conv1u_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ULDS)); });
  const int tiles = p.M / UBM;
  const int cus = device_cu_count(256);
  const int grid = std::max(8, std::min((tiles + 7) & ~7, cus * 2 / 8 * 8));
  hipLaunchKernelGGL(conv1u_kernel, dim3(grid), dim3(256), ULDS, s, p);
}

}  // namespace ttr
