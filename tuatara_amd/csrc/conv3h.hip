// CRAFT's 32-channel head layers (conv_cls.0 / .2 / .4: 3x3, 32 -> 32 at half resolution; inside the TorchScript module run at tuatara.cpp:376) on PACKED pairs,
// persistent (gfx950 / MI355X).
//
// conv3p.hip's packed-pairs tile (NP = 2) is one workgroup per 8 x 32 patch: per patch it pulls 44 KB of halo patch AND 72 KB of weights (18 tap steps of 4 KB) through
// L2 -> LDS for 32 KB of output, and waits one L2 round trip per tap - 148 us per 8-page launch where the layer's bytes take 65 (profiles/r05_pmc_craft_x4.json:
// 2.6 - 2.7 TB/s).  Here the workgroup is persistent (conv3s.hip's structure, the bf16 engine's head): the weights of all nine taps live in LDS as ready MFMA A fragments
// (w0 and w1: 36 KB; w0 / 2^11 is formed in registers, as in the fused head tail), fetched once per workgroup; per patch there is ONE burst of LDS-DMA loads, one barrier,
// 216 MFMAs per wave with every address a lane constant plus an immediate, and the epilogue.  Two workgroups share a CU (80 KB each): one's patch load runs under the
// other's MFMAs.
//
// Same arithmetic in the same order as conv3p.hip's NP = 2 loop, so that heat maps stay bit-identical (tests/test_gpu_split_gemm.py, tuning key "head_persistent"): per
// accumulator chunk 0 = taps 0 .. 8 x (x0 w0, then x1 w0b), chunk 1 = taps 0 .. 8 x (x0 w1); the MFMAs of chunk 1's second K half multiply x1 by the zero half of the
// [w1 | 0] rows there and are skipped here (they add +0).  Same channel-to-lane map (a lane ends with channels 8 fg .. 8 fg + 7 of pixel fr), same epilogue: bias, ReLU,
// pair split to the packed row [x0 (32) | x1 (32)], or conv_cls.6 + conv_cls.8 on the pixel (ConvParams::tail_heat).
#include <algorithm>
#include <stdexcept>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int HPH = 8, HPW = 32, HW2 = HPW + 2, NHALO = (HPH + 2) * HW2;   // 340 halo pixels of 128 bytes
constexpr int XPIECES = (NHALO + 7) / 8;                                    // 1-KiB pieces of 8 slots
constexpr int XBYTES = XPIECES * 1024;                                      // 44032
constexpr int WFRAGS = 2 * 9 * 2;                                           // [w0 | w1][tap][channel tile]: one KiB each (64 lanes x 16 bytes)
constexpr int HLDS = XBYTES + WFRAGS * 1024;                                // 80896: two workgroups per CU
}  // namespace

__global__ __launch_bounds__(256, 2) void conv3h_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const xs = smem;                 // [XPIECES * 8 slots][128 B]: slot pi = pr * 34 + pc, 16-byte chunk c of the pixel row at position c ^ (pi & 7) (conv3p.hip's image)
  unsigned char* const wl = smem + XBYTES;        // [WFRAGS][64 lanes][16 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ptx = p.W / HPW, pty = p.H / HPH;
  const int ntiles = p.B * pty * ptx, per_xcd = (ntiles + 7) >> 3;
  // XCD x (blockIdx % 8) walks a contiguous eighth of the patches, its workgroups side by side: neighbouring patches share their halo rows in that XCD's L2 (conv3s.hip)
  const int xcd = blockIdx.x & 7, t_first = xcd * per_xcd + (int)(blockIdx.x >> 3), t_end = min(ntiles, (xcd + 1) * per_xcd), t_step = (int)(gridDim.x >> 3);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)(unsigned)((size_t)p.M * 128), 0x00020000);

  // ---- the weights, once per workgroup: fragment (c, tap, jj) = A rows of channel tile jj (row q = channel (q >> 2) * 8 + jj * 4 + (q & 3)), k = 8 fg .. 8 fg + 7 of the
  // tap's 32 input channels, from plane 0 (c = 0: w0) or plane 2 (c = 1: w1) of the packed rows [cout][3][9 * 64] (engine.h: Linear::wsp)
  {
    const int n0 = (fr >> 2) * 8 + (fr & 3);
    const f16* wbase = reinterpret_cast<const f16*>(p.wgt);
    for (int f = wave; f < WFRAGS; f += 4) {
      const int c = f / 18, tap = (f - c * 18) >> 1, jj = f & 1;
      const int n = n0 + jj * 4;
      f16x8 w = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
      if (n < p.Cout) w = *reinterpret_cast<const f16x8*>(wbase + (size_t)n * (3 * 576) + (c ? 2 * 576 : 0) + tap * 64 + fg * 8);
      *reinterpret_cast<f16x8*>(wl + f * 1024 + lane * 16) = w;
    }
  }
  const int nch = fg * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (p.bias && nch + e < p.Cout) ? p.bias[nch + e] : 0.f;
  const f16 dn = (f16)(1.f / 2048.f);
  const f16x8 dnv = {dn, dn, dn, dn, dn, dn, dn, dn};
  RangeWatch rw;
  // the fused tail's operands, once per workgroup (inside the patch loop their loads would sit behind the next patch's requests and wait for them)
  f16x8 ta0 = dnv, ta1 = dnv, tc0 = dnv, tc1 = dnv;
  float4 tb6 = make_float4(0.f, 0.f, 0.f, 0.f);
  float tb8[2] = {0.f, 0.f};
  if (p.tail_heat) {
    const f16* w6 = reinterpret_cast<const f16*>(p.tail_w6) + fr * 64 + fg * 8;
    const f16* w8 = reinterpret_cast<const f16*>(p.tail_w8) + fr * 64 + fg * 8;
    ta0 = *reinterpret_cast<const f16x8*>(w6); ta1 = *reinterpret_cast<const f16x8*>(w6 + 32);
    tc0 = *reinterpret_cast<const f16x8*>(w8); tc1 = *reinterpret_cast<const f16x8*>(w8 + 32);
    tb6 = *reinterpret_cast<const float4*>(p.tail_b6 + 4 * fg);
    tb8[0] = p.tail_b8[0]; tb8[1] = p.tail_b8[1];
  }
  __syncthreads();                                 // the weight fragments are written with ds_write

  // The patch of tile t + 1 is requested as soon as every wave has read tile t's last fragment - in FRONT of tile t's epilogue - and awaited behind it with a counted
  // wait (vmcnt counts loads and stores together in issue order: the epilogue's stores are the only younger operations), so that the epilogue's arithmetic and the
  // stores' round trip run under the next patch's load.
  // a lane's share of the burst: piece q = 4 i + wave covers halo slots 8 q .. 8 q + 7, this lane slot 8 q + (lane >> 3) = halo pixel (pr, pc), chunk (lane & 7) ^ (slot & 7);
  // relative to the patch's pixel (y0, x0) that is a constant byte offset, and inside the image unless the patch touches its border (four compares per piece)
  constexpr int PPW = (XPIECES + 3) / 4;             // pieces per wave
  int rel[PPW], prc[PPW];                            // byte offset from pixel (y0 - 1, x0 - 1); (pr << 8) | pc, or -1 past the halo
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = i * 4 + wave, pi = piece * 8 + (lane >> 3);
    const int pr = pi / HW2, pc = pi - pr * HW2;
    const int g = (lane & 7) ^ (pi & 7);
    rel[i] = (pr * p.W + pc) * 128 + g * 16;
    prc[i] = (piece < XPIECES && pi < NHALO) ? ((pr << 8) | pc) : -1;
  }
  auto request_patch = [&](int tile) {
    const int b = tile / (pty * ptx), trem = tile - b * pty * ptx, ty = trem / ptx, tx = trem - ty * ptx;
    const int y0 = ty * HPH, x0 = tx * HPW;
    const unsigned base = (unsigned)(((b * p.H + y0 - 1) * p.W + x0 - 1) * 128);   // (may wrap below zero at the image's first pixel: only used where the lane is inside)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = i * 4 + wave;
      if (piece < XPIECES) {
        const int y = y0 - 1 + (prc[i] >> 8), x = x0 - 1 + (prc[i] & 255);
        const bool ok = prc[i] >= 0 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned vo = ok ? base + (unsigned)rel[i] : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(xs + piece * 1024), 16, vo, 0, 0, 0);
      }
    }
  };
  // store instructions of a tile's epilogue per wave (all of them younger than the next patch's requests): 4 blocks x 2 (pairs: two planes; fp32: two 16-byte halves), or 4 x 1 (the fused tail's heat map)
  const bool st8 = !p.tail_heat && p.out != nullptr;
  int stamp_n = 0;   // (diagnostics, tuning key dec_stamps = 6: s_memtime stamps of wave 0 of workgroup 0 over its first 24 patches, 8 per patch)
#define TTR_C3H_STAMP(k) do { if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp_n < 24) p.dbg[stamp_n * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
  if (t_first < t_end) request_patch(t_first);
  bool first = true;
  for (int tile = t_first; tile < t_end; tile += t_step) {
    const int b = tile / (pty * ptx), trem = tile - b * pty * ptx, ty = trem / ptx, tx = trem - ty * ptx;
    const int y0 = ty * HPH, x0 = tx * HPW;
    TTR_C3H_STAMP(0);
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (st8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (half the epilogue's store instructions: a count below the real one only waits for more)
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    first = false;
    TTR_C3H_STAMP(1);                              // the patch has landed (this wave's share)
    __syncthreads();
    TTR_C3H_STAMP(2);

    f32x4 acc[2][4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tile row r = 64 wave + 16 i + fr is patch pixel (r >> 5, r & 31); tap (ky, kx) reads halo slot pi = (py + ky) * 34 + px + kx
    int opq = 0;
    asm volatile("" : "+v"(opq));                  // (per patch: keeps the 36 (tap, tile) addresses from being hoisted out of the patch loop as 36 registers)
    const int pib = ((wave * 64) >> 5) * HW2 + fr + opq;
    int xb0[8], xb1[8];                            // x0 / x1 fragment address for swizzle variant sg: pi & 7 = (pib + sg) & 7 with sg = (offset) & 7
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) {
      const int sw = (pib + sg) & 7;
      xb0[sg] = pib * 128 + ((fg ^ sw) << 4);
      xb1[sg] = pib * 128 + (((4 + fg) ^ sw) << 4);
    }
    // ---- chunk 0: x0 w0, then x1 w0b, per tap
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tapoff = (tap / 3) * HW2 + (tap % 3);
      f16x8 w0[2], fx0[4], fx1[4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) w0[jj] = *reinterpret_cast<const f16x8*>(wl + (tap * 2 + jj) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = ((i * 16) >> 5) * HW2 + ((i * 16) & 31) + tapoff;
        fx0[i] = *reinterpret_cast<const f16x8*>(xs + xb0[off & 7] + off * 128);
        fx1[i] = *reinterpret_cast<const f16x8*>(xs + xb1[off & 7] + off * 128);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[jj], fx0[i], acc[jj][i], 0, 0, 0);
      const f16x8 w0b[2] = {w0[0] * dnv, w0[1] * dnv};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b[jj], fx1[i], acc[jj][i], 0, 0, 0);
    }
    // ---- chunk 1: x0 w1 (its second K half multiplies x1 by zeros in conv3p.hip's form: skipped)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tapoff = (tap / 3) * HW2 + (tap % 3);
      f16x8 w1[2], fx0[4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) w1[jj] = *reinterpret_cast<const f16x8*>(wl + ((9 + tap) * 2 + jj) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = ((i * 16) >> 5) * HW2 + ((i * 16) & 31) + tapoff;
        fx0[i] = *reinterpret_cast<const f16x8*>(xs + xb0[off & 7] + off * 128);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[jj], fx0[i], acc[jj][i], 0, 0, 0);
    }

    TTR_C3H_STAMP(3);                              // MFMAs issued
    __syncthreads();                               // every wave has read its last fragment of this patch
    TTR_C3H_STAMP(4);
    if (tile + t_step < t_end) request_patch(tile + t_step);
    asm volatile("" ::: "memory");                 // the epilogue's stores stay BEHIND the requests (the counted wait at the top relies on the order)
    __builtin_amdgcn_sched_barrier(0);
    TTR_C3H_STAMP(5);                              // next patch requested
    // ---- epilogue (conv3p.hip's, NP = 2)
    if (nch < p.Cout) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wave * 64 + i * 16 + fr;
        const int64_t m = ((int64_t)b * p.H + y0 + (r >> 5)) * p.W + x0 + (r & 31);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[0][i][e], p.out_scale, bv[e]); v[4 + e] = fmaf(acc[1][i][e], p.out_scale, bv[4 + e]); }
        if (p.act == kActRelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (p.tail_heat) {   // conv_cls.6 + ReLU + conv_cls.8 on this pixel: as in conv3p.hip (pairs x pairs, three MFMAs per product)
          const f16x8 a0 = ta0, a1 = ta1, c0 = tc0, c1 = tc1;
          f16x8 x0v, x1v;
          split2_x8(v, x0v, x1v, rw);
          f32x4 t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0 * dnv, x1v, t6, 0, 0, 0);
          t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x0v, t6, 0, 0, 0);
          const float4 b6 = tb6;
          float yv[8] = {fmaxf(fmaf(t6[0], p.tail_s6, b6.x), 0.f), fmaxf(fmaf(t6[1], p.tail_s6, b6.y), 0.f), fmaxf(fmaf(t6[2], p.tail_s6, b6.z), 0.f),
                         fmaxf(fmaf(t6[3], p.tail_s6, b6.w), 0.f), 0.f, 0.f, 0.f, 0.f};
          f16x8 y0v, y1v;
          split2_x8(yv, y0v, y1v, rw);
          f32x4 t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0, y0v, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0 * dnv, y1v, t8, 0, 0, 0);
          t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, y0v, t8, 0, 0, 0);
          if (fg == 0) *reinterpret_cast<float2*>(p.tail_heat + m * 2) = make_float2(fmaf(t8[0], p.tail_s8, tb8[0]), fmaf(t8[1], p.tail_s8, tb8[1]));
          continue;
        }
        if (p.out) st_split_n(p.out, m, p.out_ld, nch, v, p.out_planes, rw);
      }
    }
    TTR_C3H_STAMP(6);                              // epilogue issued
    ++stamp_n;
  }
#undef TTR_C3H_STAMP
  rw.flush(p.range_flag, p.range_tag);
}

// the shapes this kernel takes: conv3p.hip's packed-pairs case on 8 x 32 patches, one output (or the fused tail)
bool conv3h_eligible(const ConvParams& p) {
  if (p.split != 2 || p.ks != 3 || p.dil != 1 || p.C0 != 64 || p.C1 != 0 || p.Cout > 32 || p.Cout % 8) return false;
  if (p.H % HPH || p.W % HPW || p.B <= 0 || p.M != p.B * p.H * p.W) return false;
  if (p.out_relu || p.out_pool || p.resid || p.up_z || p.pre_wgt || p.relu0) return false;
  if (!p.tail_heat && (!p.out || (p.out_planes != 2 && p.out_planes != 0))) return false;
  if (p.act != kActRelu && p.act != kActNone) return false;
  if (((uintptr_t)p.in0 | (uintptr_t)p.wgt) & 15) return false;
  return (size_t)p.M * 128 < ((size_t)1 << 31);
}

static unsigned long long* g_c3h_stamps = nullptr;
void set_conv3h_stamps(unsigned long long* d) { g_c3h_stamps = d; }
static int g_c3h_wgs_per_cu = 2;
void set_conv3h_wgs_per_cu(int v) { g_c3h_wgs_per_cu = v < 1 ? 1 : (v > 2 ? 2 : v); }

void launch_conv3h(const ConvParams& p_in, hipStream_t s) {
  if (!conv3h_eligible(p_in)) throw std::runtime_error("conv3h: shape not supported");
  ConvParams p = with_range_ctx(p_in);
  p.dbg = g_c3h_stamps;
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)conv3h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HLDS)); });
  const int tiles = p.B * (p.H / HPH) * (p.W / HPW);
  const int cus = device_cu_count(256);
  const int grid = std::max(8, std::min((tiles + 7) & ~7, cus * g_c3h_wgs_per_cu / 8 * 8));   // a multiple of 8: an equal number of workgroups per XCD
  hipLaunchKernelGGL(conv3h_kernel, dim3(grid), dim3(256), HLDS, s, p);
}

}  // namespace ttr
