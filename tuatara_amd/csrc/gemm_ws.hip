// bf16 linear layer with the WEIGHTS held in registers for the whole launch ("weight-stationary"), K <= 384:
//   out[m][n] = act( sum_k X[m][k] Wt[n][k] + bias[n] (+ resid) ),   same ConvParams contract as gemm2.hip (ks = 1).
//
// The ViT encoder linears of PARSeq with K = 384 (qkv, proj, fc1; run inside the TorchScript module called at
// tuatara.cpp:307) are bound in gemm2 by the L2 -> LDS fill rate: with K this short both operand tiles are re-staged
// for every output tile.  Here a workgroup (8 waves) owns 256 output columns for the whole launch: wave w keeps the
// MFMA A fragments of its 32 columns x 384 k (24 fragments = 96 VGPRs) in registers, loaded once, and only the
// activation panels [64 rows x K] stream through a 3-slot LDS ring (LDS-DMA, two panels ahead, ONE barrier per panel = per
// 96 MFMAs of a wave); all waves read the same X fragments.  L2 -> LDS bytes per MFMA are 2x below the 256x256 gemm2 tile and 4x
// below the 128x128 one, and the weights are fetched exactly once per workgroup.
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int WS_BM = 64, WS_BN = 256, WS_SLOTS = 3;   // a ring slot holds a whole [64 rows x K] activation panel
}  // namespace

template <int NK>   // K / 64 (1..6)
__global__ __launch_bounds__(512) void gemm_ws_kernel(ConvParams p, int nslices, int mgroups) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int K = NK * 64, SLOT = WS_BM * K * 2;       // K-step sub-tile ks of a slot: [64 rows][128 B] at ks * 8192
  float2* const glut = reinterpret_cast<float2*>(smem + WS_SLOTS * SLOT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;

  // workgroups bid and bid + 8 share an XCD (speed only): inside an XCD consecutive workgroups are the N slices of one
  // M group, so the slices that read the same activation panels run side by side on one L2
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int per_xcd = gridDim.x >> 3;                 // workgroups per XCD (grid is a multiple of 8)
  const int slice = j % nslices, mg_local = j / nslices, mg_per_xcd = per_xcd / nslices;
  if (mg_local >= mg_per_xcd) return;                  // leftover workgroups of an XCD that do not fill a slice set
  const int mgroup = xcd * mg_per_xcd + mg_local;      // 0 .. mgroups-1
  const int tilesM = (p.M + WS_BM - 1) / WS_BM;
  if (mgroup >= tilesM) return;
  const int n0 = slice * WS_BN;
  if (p.act == kActGelu) {
    for (int i = tid; i < 512; i += 512) reinterpret_cast<uint4*>(glut)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  }

  // ---- resident weight fragments: tile jj (0/1) row q of this wave is column n0 + 32*wave + (q>>2)*8 + jj*4 + (q&3), so the
  // lane ends up with 8 consecutive output columns (16-byte stores), as in gemm2.hip
  bf16x8 fw[NK * 2][2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int n = n0 + wave * 32 + (fr >> 2) * 8 + jj * 4 + (fr & 3);
    const bf16* wp = reinterpret_cast<const bf16*>(p.wgt) + (size_t)min(n, p.Cout - 1) * K + fg * 8;
#pragma unroll
    for (int u = 0; u < NK * 2; ++u) {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(wp + u * 32);
      if (n >= p.Cout) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      fw[u][jj] = v;
    }
  }

  // ---- X stream: this workgroup's M tiles are mgroup, mgroup + mgroups, ...; wave w loads rows 8w..8w+7 of every K-step
  // sub-tile (NK 1-KiB pieces per panel); LDS chunk lane&7 of row r holds global chunk (lane&7) ^ ((r>>1)&7)
  const int nt = (tilesM - mgroup + mgroups - 1) / mgroups;
  const __amdgpu_buffer_rsrc_t rsx = ws_rsrc(p.in0, (unsigned)((size_t)p.M * K * 2));
  const int xrow = wave * 8 + (lane >> 3);
  const unsigned xchunk = (unsigned)(((lane & 7) ^ ((xrow >> 1) & 7)) * 16);
  int it = 0;                                          // next tile to issue
  auto issue_x = [&]() {
    const int m = (mgroup + it * mgroups) * WS_BM + xrow;
    unsigned char* sb = smem + (it % WS_SLOTS) * SLOT + wave * 1024;
    const unsigned base = m < p.M ? (unsigned)m * (unsigned)(K * 2) + xchunk : 0x80000000u;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const unsigned vo = base == 0x80000000u ? base : base + (unsigned)(ks * 128);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sb + ks * 8192), 16, vo, 0, 0, 0);
    }
    ++it;
  };
  const int frag_lane = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);

  f32x4 acc[4][2];
  auto mfma_phase = [&](int t) {
    const unsigned char* xb = smem + (t % WS_SLOTS) * SLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 fx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(xb + ks * 8192 + (frag_lane ^ (kk * 64)) + i * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks * 2 + kk][jj], fx[i], acc[i][jj], 0, 0, 0);
      }
    }
  };
  // epilogue of M tile t: lane holds columns n..n+7 of row m for every i
  const int n = n0 + wave * 32 + fg * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  if (p.bias && n < p.Cout) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  auto epilogue = [&](int t) {
    const int m0 = (mgroup + t * mgroups) * WS_BM;
    if (n >= p.Cout) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + i * 16 + fr;
      if (m >= p.M) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[i][0][e] + bv[e]; v[4 + e] = acc[i][1][e] + bv[4 + e]; }
      if (p.resid) {
        const float* rp = p.resid + (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld + n;
        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      if (p.act == kActRelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (p.act == kActGelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_lut(v[e], glut);
      }
      if (p.out) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + (int64_t)m * p.out_ld + n) = o;
      }
      if (p.out_f32) {
        float* op = p.out_f32 + (int64_t)m * p.out_f32_ld + n;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  };

  // Waves w and w + 4 share a SIMD.  Waves 0-3 run [MFMA(t), epilogue(t)] after barrier t; waves 4-7 run [epilogue(t-1),
  // MFMA(t)]: the SIMD's matrix pipe works on one wave's panel while its vector ALU finishes the other wave's previous panel
  // (bias / GELU / stores), instead of both waves queueing for the same unit.  The epilogue's stores and residual loads
  // retire in order with the X stream, so the counted wait below only ever waits for older operations too.
  const bool late = wave >= 4;
  for (int a = 0; a < 2 && it < nt; ++a) issue_x();       // two panels ahead
  for (int t = 0; t < nt; ++t) {
    // panel t has landed once at most (issued - t - 1) younger panels (NK loads each, in order) remain in flight
    if (it - t - 1 >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // everyone's pieces landed; slot (t+2)%3 = (t-1)%3 is free again
    if (it < nt) issue_x();
    if (!late) { mfma_phase(t); epilogue(t); }
    else { if (t > 0) epilogue(t - 1); mfma_phase(t); }
  }
  if (late && nt > 0) epilogue(nt - 1);
}

const char* gemm_ws_check(const ConvParams& p) {
  if (p.ks != 1 || p.C1 || p.relu0 || p.relu1 || p.out_relu || p.out_pool) return "gemm_ws: plain linear layers only";
  if (p.C0 % 64 || p.C0 < 64 || p.C0 > 384) return "gemm_ws: K must be a multiple of 64, at most 384";
  if (p.Cout % 8) return "gemm_ws: Cout % 8";
  if (p.out && (p.out_ld % 8 || ((uintptr_t)p.out & 15))) return "gemm_ws: bf16 output alignment";
  if (p.out_f32 && (p.out_f32_ld % 4 || ((uintptr_t)p.out_f32 & 15))) return "gemm_ws: f32 output alignment";
  if (p.resid && (p.resid_ld % 4 || ((uintptr_t)p.resid & 15))) return "gemm_ws: residual alignment";
  if (p.bias && ((uintptr_t)p.bias & 15)) return "gemm_ws: bias alignment";
  if (((uintptr_t)p.in0 & 15) || ((uintptr_t)p.wgt & 15)) return "gemm_ws: operand alignment";
  if ((size_t)p.M * p.C0 * 2 >= ((size_t)1 << 31)) return "gemm_ws: tensor too large";
  if (p.M <= 0 || p.Cout <= 0) return "gemm_ws: bad shape";
  return nullptr;
}

template <int NK>
static void launch_ws(const ConvParams& p, int nslices, int mgroups, int grid, size_t lds, hipStream_t s) {
  static bool once = false;
  if (!once) {
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SLOTS * WS_BM * NK * 128 + 8192));
    once = true;
  }
  hipLaunchKernelGGL(gemm_ws_kernel<NK>, dim3(grid), dim3(512), lds, s, p, nslices, mgroups);
}

void launch_gemm_ws(const ConvParams& p_in, hipStream_t s) {
  if (const char* e = gemm_ws_check(p_in)) throw std::runtime_error(e);
  ConvParams p = p_in;
  p.gelu_lut = p.act == kActGelu ? gelu_lut_for_current_device() : nullptr;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8) cus = prop.multiProcessorCount;
  const int nslices = (p.Cout + WS_BN - 1) / WS_BN;
  const int per_xcd = std::max(nslices, cus / 8);                 // one workgroup per CU; at least one slice set per XCD
  const int mg_per_xcd = per_xcd / nslices;
  const int mgroups = 8 * mg_per_xcd;
  const size_t lds = (size_t)WS_SLOTS * WS_BM * p.C0 * 2 + (p.act == kActGelu ? 8192 : 0);
  const int grid = 8 * per_xcd;
  switch (p.C0 / 64) {
    case 1: return launch_ws<1>(p, nslices, mgroups, grid, lds, s);
    case 2: return launch_ws<2>(p, nslices, mgroups, grid, lds, s);
    case 3: return launch_ws<3>(p, nslices, mgroups, grid, lds, s);
    case 4: return launch_ws<4>(p, nslices, mgroups, grid, lds, s);
    case 5: return launch_ws<5>(p, nslices, mgroups, grid, lds, s);
    default: return launch_ws<6>(p, nslices, mgroups, grid, lds, s);
  }
}

}  // namespace ttr
