// NOT part of the build: the pair-split form of the bf16 fused MLP block (two waves per SIMD), a measured negative of round 2 (no faster than
// mlp_fused.hip; DESIGN_APPENDIX.md).  Kept as a record; it compiled against csrc/kernels.h of that round.
// The MLP half of a PARSeq ViT encoder block as one kernel, second generation: TWO waves per SIMD.
//
//   x_out = x + fc2( GELU( fc1( LayerNorm_2(x) ) ) )            and, optionally,  y = LayerNorm_next(x_out)  (bf16)
//
// (timm Block.forward second half, run inside the TorchScript module the reference calls at tuatara.cpp:307.)
// mlp_fused.hip runs one wave per SIMD with a [32 rows x 384] f32 tile in 192 accumulators: nothing overlaps a wave's own
// barrier, LDS-DMA issue, GELU or epilogue, and rocprof reports 0.26 MFMA utilisation (profiles/r02_pmc_mfma_parseq.json).
// The register file is unified - 512 per lane per SIMD, one allocation for all waves of a kernel - so a second wave per
// SIMD needs the wave's state in 256 registers.  Here the two waves of a SIMD (w and w + 4 of an 8-wave workgroup) share
// the same 32 rows and split the work of every hidden chunk:
//
//   * both keep LayerNorm_2 of the 32 rows as MFMA B fragments (96 registers, k = all 384 channels);
//   * GEMM1: wave half hh multiplies 16 of the chunk's 32 hidden units (one 16-row weight tile, 24 MFMAs), adds the bias
//     (C operand), applies GELU and rounds to bf16: 4 of the 8 hidden units a lane needs as GEMM2's B fragment.  The other 4
//     come from the partner through LDS (16 bytes per lane and chunk, double buffered across the chunk barrier);
//   * GEMM2: wave half hh accumulates 192 of the 384 output channels (12 of the 24 weight tiles, 24 MFMAs) into 96
//     accumulators, one chunk behind GEMM1;
//   * the two halves run the two phases of an iteration in opposite order, so one wave's GELU / fragment waits sit beside
//     the other's MFMAs.
//
// Weight images, their LDS-DMA ring (here 2 slots: an item is fetched while the previous one is multiplied), rounding
// points and the epilogue (bias + residual, f32 store, next LayerNorm - its row statistics cross the pair through LDS) are
// those of mlp_fused.hip; results differ from it by fp32 summation order only.  LDS-read and L2->LDS traffic per panel are
// unchanged (every weight fragment still feeds two MFMAs).
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int E = 384, HID = 1536, CH = 32, NCH = HID / CH;   // 48 chunks of 32 hidden units
constexpr int W1B = CH * E * 2;                               // W1 chunk image [3 k-segments][32 rows][256 B]
constexpr int W2B = E * CH * 2;                               // W2 chunk image [384 rows][64 B]
constexpr int SLOT = W1B + W2B, NSLOT = 2;
constexpr int LUT_OFF = NSLOT * SLOT;                         // GELU table, 8 KiB
constexpr int B1_OFF = LUT_OFF + 8192;                        // fc1 bias, f32 [1536]
constexpr int EX_OFF = B1_OFF + HID * 4;                      // hidden halves: [2 buffers][8 waves][64 lanes] 16 B
constexpr int RED_OFF = EX_OFF + 2 * 8 * 1024;                // row statistics of the epilogue: [8 waves][2 row tiles][16] f32
constexpr int PAIR_LDS = RED_OFF + 8 * 2 * 16 * 4;            // 130,048 B
constexpr int BM = 128, ITEMS = NCH + 1;
static_assert(PAIR_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// LDS accesses next to the LDS-DMA stream are inline asm (hipcc drains the vector-memory queue before any it can see)
#define PR_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// the lane id, re-read from the hardware wherever it is needed: volatile, so that neither it nor anything derived from it is
// hoisted out of the chunk loop and kept (= spilled: a spill reload is a vector-memory load whose wait stalls the weight DMA)
__device__ __forceinline__ unsigned pr_lane() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
#define PR_LANE() pr_lane()
#define PR_WAITF(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
}  // namespace

// HH = the wave's half, a compile-time constant per code path: half 0 runs an iteration as [GEMM2 of the previous chunk, GEMM1 + GELU
// of this one], half 1 the other way round, so the SIMD's two waves are half an iteration apart from barrier to barrier (one's
// fragment waits and GELU beside the other's MFMAs) - a run-time `if` around the two orders made hipcc spill inside the loop
template <int HH>
__device__ __forceinline__ void mlp_pair_body(const MlpParams& p, unsigned char* smem) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int hh = HH;
  const int pr = wave & 3;                     // pair (rows 32 pr ..) and half: waves w and w + 4 share a SIMD
  int q, g;                                                    // lane & 15, lane >> 4: re-derived per phase (see pr_lane)
  const int npanels = (p.M + BM - 1) / BM;
  if ((int)blockIdx.x >= npanels) return;
  const int my_n = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_n * ITEMS;                              // ring items this workgroup walks

  for (int i = tid; i < 512; i += 512) reinterpret_cast<uint4*>(smem + LUT_OFF)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  for (int i = tid; i < HID / 4; i += 512) reinterpret_cast<float4*>(smem + B1_OFF)[i] = reinterpret_cast<const float4*>(p.b1)[i];

  const __amdgpu_buffer_rsrc_t rs1 = p_rsrc(p.w1p, (unsigned)(HID * E * 2));
  const __amdgpu_buffer_rsrc_t rs2 = p_rsrc(p.w2p, (unsigned)(HID * E * 2));
  // item Gi of the launch = {W1 chunk ii (ii < 48), W2 chunk ii - 1 (ii >= 1)}: 24 + 24 one-KiB pieces, 3 + 3 per wave; the
  // images are stored in memory as they sit in LDS (pack_mlp_w1 / pack_mlp_w2), so a lane's source offset is 16 * lane
  auto issue = [&](int Gi) {
    const int ii = Gi % ITEMS;
    unsigned char* sb = smem + (Gi & 1) * SLOT + wave * 1024;
    const unsigned sl = PR_LANE() * 16u;
    if (ii < NCH) {
#pragma unroll
      for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sb + j * 8192), 16, sl, ii * W1B + (wave + 8 * j) * 1024, 0, 0);
    }
    if (ii >= 1) {
#pragma unroll
      for (int j = 0; j < 3; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(sb + W1B + j * 8192), 16, sl, (ii - 1) * W2B + (wave + 8 * j) * 1024, 0, 0);
    }
  };

  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const unsigned lut_lds = lds0 + LUT_OFF;
  float* const red = reinterpret_cast<float*>(smem + RED_OFF);

  issue(0);
  __syncthreads();                                             // table and bias staged (this also waits for item 0: once per launch)

  int G = 0;
  for (int pi = 0; pi < my_n; ++pi) {
    const int panel = (int)blockIdx.x + pi * (int)gridDim.x;
    const int row0 = panel * BM + pr * 32;
#define PR_PSTAMP(k) do { if (p.dbg && blockIdx.x == 0 && tid == 0 && pi < 4) p.dbg[384 + pi * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
    PR_PSTAMP(0);

    bf16x8 xf[2][12];                                           // LayerNorm_2 of the pair's 32 rows as MFMA B fragments
    { const unsigned lf = PR_LANE(); q = (int)(lf & 15u); g = (int)(lf >> 4); }
    f32x4 acc2[2][12];                                          // this half's [32 rows x 192 channels] of the output tile
    // ---- LayerNorm of the 32 rows (both waves of the pair, redundantly).  Lane (q, g): row 16 rt + q, channels 32 ks + 8 g + e.
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = min(row0 + rt * 16 + q, p.M - 1);
      const float* xr = p.x + (size_t)row * E + g * 8;
      float v[12][8];
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(xr + ks * 32), b = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        v[ks][0] = a.x; v[ks][1] = a.y; v[ks][2] = a.z; v[ks][3] = a.w; v[ks][4] = b.x; v[ks][5] = b.y; v[ks][6] = b.z; v[ks][7] = b.w;
      }
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < 12; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[ks][e];
      s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      const float mean = s * (1.f / E);
      float s2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < 12; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; s2 += d * d; }
      s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
      const float rstd = rsqrtf(s2 * (1.f / E) + p.ln_eps);
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        const float4 g0 = *reinterpret_cast<const float4*>(p.ln_g + ks * 32 + g * 8), g1 = *reinterpret_cast<const float4*>(p.ln_g + ks * 32 + g * 8 + 4);
        const float4 t0 = *reinterpret_cast<const float4*>(p.ln_b + ks * 32 + g * 8), t1 = *reinterpret_cast<const float4*>(p.ln_b + ks * 32 + g * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((v[ks][e] - mean) * rstd * gg[e] + bb[e]);
        xf[rt][ks] = o;
      }
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int t = 0; t < 12; ++t) acc2[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    PR_PSTAMP(1);
    // ---- 49 iterations: iteration it = GEMM1 + GELU of chunk it (it < 48) and GEMM2 of chunk it - 1 (it >= 1)
    u32x4 hprev = u32x4{0u, 0u, 0u, 0u};                         // this half's GELU output of the previous chunk: {rt 0: 4 bf16, rt 1: 4 bf16}
    u32x4 hnew = hprev;                                          // (half 1 only: its phase 1 runs before the phase 2 that still needs the previous half)
    for (int it = 0; it <= NCH; ++it, ++G) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // this wave's pieces of item G have landed, its hidden half is in LDS
      __builtin_amdgcn_s_barrier();                              // everyone's have; iteration it - 1 is over: slot (G + 1) & 1 and the older exchange buffer are free
      const unsigned ll = PR_LANE();
      const bool more = G + 1 < total && !((p.ablate & 1) && G >= 2);
      const int nii = (G + 1) % ITEMS;                           // the next item: W1 chunk nii (nii < 48), W2 chunk nii - 1 (nii >= 1)
      // its 3 + 3 pieces are issued one behind each MFMA group of this iteration: the address unit takes ~20 cycles per 1-KiB piece
      // and holds the issuing wave meanwhile - time the SIMD's other wave spends in its MFMAs (a burst of all 48 pieces behind the
      // barrier stopped all eight waves for ~900 cycles)
      auto piece = [&](int k) {
        if (!more) return;
        unsigned char* sb = smem + ((G + 1) & 1) * SLOT + wave * 1024;
        const unsigned sl = ll * 16u;
        if (k < 3) { if (nii < NCH) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sb + k * 8192), 16, sl, nii * W1B + (wave + 8 * k) * 1024, 0, 0); }
        else if (nii >= 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(sb + W1B + (k - 3) * 8192), 16, sl, (nii - 1) * W2B + (wave + 8 * (k - 3)) * 1024, 0, 0);
      };
      const unsigned sbase = lds0 + (unsigned)((G & 1) * SLOT);
      const unsigned qq = ll & 15u, gq = ll >> 4;

      // phase 1: GEMM1 (hidden units 8 g + 4 hh + e of the chunk, rows q of both row tiles), GELU, hand the half to the partner
      auto phase1 = [&]() {
        const unsigned rowb = sbase + qq * 256u + (unsigned)hh * 4096u;
        unsigned a1v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a1v[t] = rowb + ((((unsigned)(4 * t) + gq) ^ qq) << 4);
        f32x4 accn[2];
        bf16x8 fa[4], fb[4];
        const unsigned ba = lds0 + B1_OFF + (unsigned)(it * CH * 4) + gq * 32u + (unsigned)hh * 16u;
        PR_RD128(accn[0], ba, 0);
#define PR_G1(dst, n)                                                                                   \
        PR_RD128(dst[0], a1v[0], (n) * 8192); PR_RD128(dst[1], a1v[1], (n) * 8192);                     \
        PR_RD128(dst[2], a1v[2], (n) * 8192); PR_RD128(dst[3], a1v[3], (n) * 8192);
#define PR_M1(src, n)                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                   \
          _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                              \
            accn[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[t], xf[rt][4 * (n) + t], accn[rt], 0, 0, 0);
        PR_G1(fa, 0)
        PR_G1(fb, 1)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(accn[0]), "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]));
        accn[1] = accn[0];
        PR_M1(fa, 0)
        PR_G1(fa, 2)
        piece(0);
        PR_WAITF(4, fb);
        PR_M1(fb, 1)
        piece(1);
        PR_WAITF(0, fa);
        PR_M1(fa, 2)
        piece(2);
        // GELU(x) = x Phi(x), Phi by the interpolated table (common.h: gelu_lut); 8 values per lane, one row tile at a time
        // (four table reads in flight: the temporaries of all eight at once push the loop over the 256-register budget)
        bf16x8 o;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          float u4[4];
          float2 t4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            u4[e] = fmaf(__builtin_amdgcn_fmed3f(accn[rt][e], -8.0f, 7.984375f), 64.0f, 512.0f);
            const unsigned ad = lut_lds + ((unsigned)(int)u4[e] << 3);
            asm volatile("ds_read_b64 %0, %1" : "=v"(t4[e]) : "v"(ad));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t4[0]), "+v"(t4[1]), "+v"(t4[2]), "+v"(t4[3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[rt * 4 + e] = (bf16)(accn[rt][e] * fmaf(__builtin_amdgcn_fractf(u4[e]), t4[e].y, t4[e].x));
        }
        if (HH == 0) hprev = *reinterpret_cast<u32x4*>(&o); else hnew = *reinterpret_cast<u32x4*>(&o);
        const unsigned ex = lds0 + EX_OFF + (unsigned)((it & 1) * 8192 + wave * 1024) + ll * 16u;
        if (HH == 0) asm volatile("ds_write_b128 %0, %1" :: "v"(ex), "v"(hprev) : "memory"); else asm volatile("ds_write_b128 %0, %1" :: "v"(ex), "v"(hnew) : "memory");
      };
      // phase 2: GEMM2 of the previous chunk: output channels 192 hh .. 192 hh + 191 (weight tiles 12 hh .. 12 hh + 11)
      auto phase2 = [&]() {
        const unsigned a2 = sbase + (unsigned)W1B + qq * 64u + ((gq ^ ((qq >> 1) & 3u)) << 4) + (unsigned)hh * 12288u;
        const unsigned ex = lds0 + EX_OFF + (unsigned)(((it - 1) & 1) * 8192 + (wave ^ 4) * 1024) + ll * 16u;
        u32x4 part;
        bf16x8 fa[4], fb[4];
        PR_RD128(part, ex, 0);
#define PR_G2(dst, n)                                                                                   \
        PR_RD128(dst[0], a2, (4 * (n) + 0) * 1024); PR_RD128(dst[1], a2, (4 * (n) + 1) * 1024);         \
        PR_RD128(dst[2], a2, (4 * (n) + 2) * 1024); PR_RD128(dst[3], a2, (4 * (n) + 3) * 1024);
#define PR_M2(src, n)                                                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
          acc2[0][4 * (n) + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[t], h0, acc2[0][4 * (n) + t], 0, 0, 0); \
          acc2[1][4 * (n) + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[t], h1, acc2[1][4 * (n) + t], 0, 0, 0); \
        }
        PR_G2(fa, 0)
        PR_G2(fb, 1)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(part), "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]));
        // B fragment of row tile rt: hidden 8 g + e, e = 0..3 from half 0, 4..7 from half 1
        u32x4 w0, w1;
        if (hh == 0) { w0 = u32x4{hprev[0], hprev[1], part[0], part[1]}; w1 = u32x4{hprev[2], hprev[3], part[2], part[3]}; }
        else         { w0 = u32x4{part[0], part[1], hprev[0], hprev[1]}; w1 = u32x4{part[2], part[3], hprev[2], hprev[3]}; }
        const bf16x8 h0 = *reinterpret_cast<bf16x8*>(&w0), h1 = *reinterpret_cast<bf16x8*>(&w1);
        PR_M2(fa, 0)
        PR_G2(fa, 2)
        piece(3);
        PR_WAITF(4, fb);
        PR_M2(fb, 1)
        piece(4);
        PR_WAITF(0, fa);
        PR_M2(fa, 2)
        piece(5);
      };
      if (HH == 0) {
        if (it >= 1 && !(p.ablate & 4)) phase2(); else { piece(3); piece(4); piece(5); }
        if (it < NCH && !(p.ablate & 8)) phase1(); else { piece(0); piece(1); piece(2); }
      } else {
        if (it < NCH && !(p.ablate & 8)) phase1(); else { piece(0); piece(1); piece(2); }
        if (it >= 1 && !(p.ablate & 4)) phase2(); else { piece(3); piece(4); piece(5); }
        hprev = hnew;
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    PR_PSTAMP(2);
    { const unsigned lf = PR_LANE(); q = (int)(lf & 15u); g = (int)(lf >> 4); }
    // ---- epilogue: + bias2 + residual -> f32; this half holds channels 192 hh + 32 pl + 8 g + e (pl = 0..5) of row 16 rt + q
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rt * 16 + q;
      const bool live = row < p.M;
      const size_t ro = (size_t)min(row, p.M - 1) * E + (size_t)hh * 192 + g * 8;
      float v[6][8];
      {
        float4 c0[6], c1[6], r0[6], r1[6];
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
          c0[pl] = *reinterpret_cast<const float4*>(p.b2 + hh * 192 + pl * 32 + g * 8); c1[pl] = *reinterpret_cast<const float4*>(p.b2 + hh * 192 + pl * 32 + g * 8 + 4);
          r0[pl] = *reinterpret_cast<const float4*>(p.x + ro + pl * 32); r1[pl] = *reinterpret_cast<const float4*>(p.x + ro + pl * 32 + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
          v[pl][0] = acc2[rt][2 * pl][0] + c0[pl].x + r0[pl].x; v[pl][1] = acc2[rt][2 * pl][1] + c0[pl].y + r0[pl].y;
          v[pl][2] = acc2[rt][2 * pl][2] + c0[pl].z + r0[pl].z; v[pl][3] = acc2[rt][2 * pl][3] + c0[pl].w + r0[pl].w;
          v[pl][4] = acc2[rt][2 * pl + 1][0] + c1[pl].x + r1[pl].x; v[pl][5] = acc2[rt][2 * pl + 1][1] + c1[pl].y + r1[pl].y;
          v[pl][6] = acc2[rt][2 * pl + 1][2] + c1[pl].z + r1[pl].z; v[pl][7] = acc2[rt][2 * pl + 1][3] + c1[pl].w + r1[pl].w;
        }
      }
      // f32 stores: v_permlane32_swap (g <-> g + 2) of the lower half's second quad with the upper half's first quad makes each
      // instruction write 64 contiguous bytes per row (as in mlp_fused.hip)
      {
        float* const ob = p.x_out + ro - g * 8 + ((g & 1) * 8 + (g >> 1) * 4);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
          typedef __attribute__((ext_vector_type(4))) float f4;
          f4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a_ = v[pl][e], b_ = v[pl][4 + e];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
            lo[e] = a_; hi[e] = b_;
          }
          if (live && !(p.ablate & 16)) {
            float* op = ob + pl * 32;
            if (p.store_nt) { __builtin_nontemporal_store(lo, reinterpret_cast<f4*>(op)); __builtin_nontemporal_store(hi, reinterpret_cast<f4*>(op + 16)); }
            else { *reinterpret_cast<f4*>(op) = lo; *reinterpret_cast<f4*>(op + 16) = hi; }
          }
        }
      }
      if (p.nln_out) {                                          // uniform.  Row statistics over all 384 channels: the two halves add up through LDS
        float s = 0.f;
#pragma unroll
        for (int pl = 0; pl < 6; ++pl)
#pragma unroll
          for (int e = 0; e < 8; ++e) s += v[pl][e];
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        __syncthreads();                                        // (the previous use of `red` is over)
        if (g == 0) red[(wave * 2 + rt) * 16 + q] = s;
        __syncthreads();
        const float mean = (red[((pr) * 2 + rt) * 16 + q] + red[((pr + 4) * 2 + rt) * 16 + q]) * (1.f / E);   // half 0 + half 1: the same order in both waves
        float s2 = 0.f;
#pragma unroll
        for (int pl = 0; pl < 6; ++pl)
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[pl][e] - mean; s2 += d * d; }
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        __syncthreads();
        if (g == 0) red[(wave * 2 + rt) * 16 + q] = s2;
        __syncthreads();
        const float rstd = rsqrtf((red[((pr) * 2 + rt) * 16 + q] + red[((pr + 4) * 2 + rt) * 16 + q]) * (1.f / E) + p.nln_eps);
#pragma unroll
        for (int pl = 0; pl < 6; ++pl) {
          const float4 g0 = *reinterpret_cast<const float4*>(p.nln_g + hh * 192 + pl * 32 + g * 8), g1 = *reinterpret_cast<const float4*>(p.nln_g + hh * 192 + pl * 32 + g * 8 + 4);
          const float4 t0 = *reinterpret_cast<const float4*>(p.nln_b + hh * 192 + pl * 32 + g * 8), t1 = *reinterpret_cast<const float4*>(p.nln_b + hh * 192 + pl * 32 + g * 8 + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)((v[pl][e] - mean) * rstd * gg[e] + bb[e]);
          if (live) { if (p.store_nt) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p.nln_out + ro + pl * 32)); else *reinterpret_cast<bf16x8*>(p.nln_out + ro + pl * 32) = o; }
        }
      }
    }
    PR_PSTAMP(3);
  }
}

__global__ __launch_bounds__(512, 2) void mlp_pair_kernel(MlpParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) mlp_pair_body<0>(p, smem); else mlp_pair_body<1>(p, smem);   // waves 0..3 / 4..7
}

static int g_pair_store_nt = 1;
static int g_pair_ablate = 0;
static unsigned long long* g_pair_dbg = nullptr;
void set_mlp_pair_stamps(unsigned long long* d) { g_pair_dbg = d; }
void set_mlp_pair_ablate(int v) { g_pair_ablate = v; }

const char* mlp_pair_check(const MlpParams& p) {
  if (p.M <= 0) return "mlp_pair: bad row count";
  if (!p.x || !p.x_out || !p.ln_g || !p.ln_b || !p.w1p || !p.b1 || !p.w2p || !p.b2 || !p.gelu_lut) return "mlp_pair: null operand";
  if (p.nln_out && (!p.nln_g || !p.nln_b)) return "mlp_pair: next LayerNorm parameters";
  if (p.att) return "mlp_pair: the attention projection is not fused in this kernel";
  const uintptr_t a = (uintptr_t)p.x | (uintptr_t)p.x_out | (uintptr_t)p.ln_g | (uintptr_t)p.ln_b | (uintptr_t)p.w1p | (uintptr_t)p.b1 | (uintptr_t)p.w2p |
                      (uintptr_t)p.b2 | (uintptr_t)p.nln_out | (uintptr_t)p.nln_g | (uintptr_t)p.nln_b | (uintptr_t)p.gelu_lut;
  if (a & 15) return "mlp_pair: operands must be 16-byte aligned";
  return nullptr;
}

void launch_mlp_pair(const MlpParams& p_in, hipStream_t s) {
  MlpParams p = p_in;
  p.gelu_lut = gelu_lut_for_current_device();
  p.store_nt = g_pair_store_nt;
  p.ablate = g_pair_ablate;
  p.dbg = g_pair_dbg;
  if (const char* e = mlp_pair_check(p)) throw std::runtime_error(e);
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS)); });
  const int cus = device_cu_count(256);
  const int npanels = (p.M + BM - 1) / BM;
  hipLaunchKernelGGL(mlp_pair_kernel, dim3(std::min(cus, npanels)), dim3(512), PAIR_LDS, s, p);
}

}  // namespace ttr
