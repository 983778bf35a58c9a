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
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int WS_BM = 64, WS_BN = 256, WS_SLOTS = 3;   // a ring slot holds a whole [64 rows x K] activation panel
}  // namespace

// ACT >= 0: the activation is a template constant and the output is bf16 only (the ViT's qkv / fc1 / cross k,v): the epilogue
// is straight-line code — a vector instruction of one wave does not issue while its SIMD partner's MFMAs do, so every
// epilogue instruction is time taken from the matrix pipe.  ACT < 0: any ConvParams epilogue (activation read at run time).
template <int NK, int ACT>   // NK = K / 64 (1..6)
__global__ __launch_bounds__(512) void gemm_ws_kernel(ConvParams p, int nslices, int mgroups) {
  constexpr bool LEAN = ACT >= 0;
  const int act = LEAN ? ACT : p.act;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int K = NK * 64, SLOT = WS_BM * K * 2;       // K-step sub-tile ks of a slot: [64 rows][128 B] at ks * 8192
  float2* const glut = reinterpret_cast<float2*>(smem + WS_SLOTS * SLOT);
  const unsigned glut_lds = (unsigned)(size_t)(lds_ptr)glut;
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
  if (act == kActGelu) {
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
    const unsigned base = (unsigned)m * (unsigned)(K * 2) + xchunk;   // rows >= M lie past the descriptor's end: zero-filled
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sb + ks * 8192), 16, base + (unsigned)(ks * 128), 0, 0, 0);
    ++it;
  };
  const int frag_lane = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);

  // lane holds columns n..n+7 of rows m0 + 16 i + fr; bvv[jj] = the bias of columns n + 4 jj .. + 3 (zero past Cout / without bias)
  const int n = n0 + wave * 32 + fg * 8;
  f32x4 bvv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (p.bias && n < p.Cout) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bvv[0] = f32x4{b0.x, b0.y, b0.z, b0.w}; bvv[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
  }
  f32x4 acc[4][2];
  auto mfma_phase = [&](int t) {
    const unsigned char* xb = smem + (t % WS_SLOTS) * SLOT;
    // fragment loads run one K step (32 k) ahead of the MFMAs that use them: two register sets, the loads of step s + 1 are
    // issued (sched_barrier keeps them there) before the 8 MFMAs of step s, so their LDS latency hides behind those MFMAs
    bf16x8 fx[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fx[0][i] = *reinterpret_cast<const bf16x8*>(xb + frag_lane + i * 2048);
#pragma unroll
    for (int st = 0; st < NK * 2; ++st) {
      if (st + 1 < NK * 2) {
        const int ks = (st + 1) >> 1, kk = (st + 1) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) fx[(st + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(xb + ks * 8192 + (frag_lane ^ (kk * 64)) + i * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)   // the first K step starts from the bias (the MFMA's C operand), not from zero + a later add
          acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[st][jj], fx[st & 1][i], st == 0 ? bvv[jj] : acc[i][jj], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // `counted`: bf16 output only, through buffer stores that are ISSUED for every (i) even when the row / column is out of
  // range (offsets past the descriptor's end are dropped by the buffer unit), so every epilogue puts exactly 4 stores on the
  // wave's in-order vector-memory queue and the wait for an activation panel can count past them
  const bool counted = LEAN || (p.out && !p.out_f32 && !p.resid && (size_t)p.M * p.out_ld * 2 < ((size_t)1 << 31));
  const __amdgpu_buffer_rsrc_t rso = ws_rsrc(p.out ? p.out : p.in0, counted ? (unsigned)((size_t)p.M * p.out_ld * 2) : 0u);
  // byte offset of (row fr + 16 i, column n) inside an M tile; columns past Cout get an offset no row offset brings back in range
  unsigned obase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) obase[i] = n < p.Cout ? ((unsigned)(i * 16 + fr) * (unsigned)p.out_ld + (unsigned)n) * 2u : 0x80000000u;
  auto epilogue = [&](int t) {
    const int m0 = (mgroup + t * mgroups) * WS_BM;
    if (counted) {
      const unsigned mo = (unsigned)m0 * (unsigned)p.out_ld * 2u;     // rows >= M: past the descriptor's end, dropped
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[i][0][e]; v[4 + e] = acc[i][1][e]; }
        if (act == kActRelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (act == kActGelu) {
          gelu_lut8_lds(v, glut_lds);
        }
        union { bf16x8 h; u32x4 u; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (bf16)v[e];
        const unsigned off = obase[i] + mo;
        // streaming policy (nt): these outputs are written once, 64 bytes per row per instruction, and read by the next kernel
        // long after they left the L2 — measured 1.25-1.35x on the whole launch against default-policy stores
        if (p.dbg_flags & 1) continue;
        if (p.store_policy == 1) __builtin_amdgcn_raw_buffer_store_b128(o.u, rso, off, 0, 2);
        else if (p.store_policy == 2) __builtin_amdgcn_raw_buffer_store_b128(o.u, rso, off, 0, 19);   // sc0 sc1 nt
        else __builtin_amdgcn_raw_buffer_store_b128(o.u, rso, off, 0, 0);
      }
      return;
    }
    if (n >= p.Cout) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + i * 16 + fr;
      if (m >= p.M) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[i][0][e]; v[4 + e] = acc[i][1][e]; }
      if (p.resid) {
        const float* rp = p.resid + (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld + n;
        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      if (act == kActRelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (act == kActGelu) {
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
#define WS_STAMP(ph) do { if (p.dbg && blockIdx.x == 0 && (tid & 255) == 0 && t < 24) p.dbg[((tid >> 8) * 24 + t) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
  for (int t = 0; t < nt; ++t) {
    WS_STAMP(0);
    // panel t has landed once at most (issued - t - 1) younger panels (NK loads each, in order) remain in flight
    // (and, from t = 3 on in counted mode, the 2 x 4 output stores issued since that panel's loads)
    if (it - t - 1 >= 1) {
      if (counted && t >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK + 8) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NK) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // everyone's pieces landed; slot (t+2)%3 = (t-1)%3 is free again
    WS_STAMP(1);
    if (it < nt) { if (p.dbg_flags & 2) ++it; else issue_x(); }
    WS_STAMP(2);
    if (!late) { mfma_phase(t); WS_STAMP(3); epilogue(t); }
    else { if (t > 0) epilogue(t - 1); WS_STAMP(3); mfma_phase(t); }
    WS_STAMP(4);
  }
#undef WS_STAMP
  if (late && nt > 0) epilogue(nt - 1);
}

unsigned long long* g_ws_dbg = nullptr;
int g_ws_dbg_flags = 0;
int g_ws_lean = 1;
void set_gemm_ws_lean(int v) { g_ws_lean = v; }
void set_gemm_ws_dbg_flags(int f) { g_ws_dbg_flags = f; }
void set_gemm_ws_stamps(unsigned long long* d) { g_ws_dbg = d; }

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

template <int NK, int ACT = -1>
static void launch_ws(const ConvParams& p, int nslices, int mgroups, int grid, size_t lds, hipStream_t s) {
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ws_kernel<NK, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_SLOTS * WS_BM * NK * 128 + 8192)); });
  hipLaunchKernelGGL((gemm_ws_kernel<NK, ACT>), dim3(grid), dim3(512), lds, s, p, nslices, mgroups);
}

void launch_gemm_ws(const ConvParams& p_in, hipStream_t s) {
  if (const char* e = gemm_ws_check(p_in)) throw std::runtime_error(e);
  ConvParams p = p_in;
  p.gelu_lut = p.act == kActGelu ? gelu_lut_for_current_device() : nullptr;
  p.dbg = g_ws_dbg; p.dbg_flags = g_ws_dbg_flags; p.store_policy = g_store_policy;
  const int cus = device_cu_count(256);
  const int nslices = (p.Cout + WS_BN - 1) / WS_BN;
  const int per_xcd = std::max(nslices, cus / 8);                 // one workgroup per CU; at least one slice set per XCD
  const int mg_per_xcd = per_xcd / nslices;
  const int mgroups = 8 * mg_per_xcd;
  const size_t lds = (size_t)WS_SLOTS * WS_BM * p.C0 * 2 + (p.act == kActGelu ? 8192 : 0);
  const int grid = 8 * per_xcd;
  // the ViT shapes (K = 384, bf16 output, no residual) take the straight-line epilogues
  const bool lean = p.C0 == 384 && p.out && !p.out_f32 && !p.resid && (size_t)p.M * p.out_ld * 2 < ((size_t)1 << 31) && g_ws_lean;
  if (lean && p.act == kActNone) return launch_ws<6, kActNone>(p, nslices, mgroups, grid, lds, s);
  if (lean && p.act == kActGelu) return launch_ws<6, kActGelu>(p, nslices, mgroups, grid, lds, s);
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
