// The MLP half of a PARSeq ViT encoder block as ONE kernel (bf16 MFMA, f32 residual stream):
//
//   x_out = x + fc2( GELU( fc1( LayerNorm_2(x) ) ) )            and, optionally,  y = LayerNorm_next(x_out)  (bf16)
//
// (timm Block.forward second half, run inside the TorchScript module called at tuatara.cpp:307; 12 blocks, E = 384,
// hidden = 1536).  As separate launches (layernorm_kernel, gemm_ws fc1, gemm2 fc2) this is 2.0 GB of HBM traffic per
// block at 1280 crops — the [M][1536] hidden activation alone is written and read back (1.0 GB) — and fc2 / the
// LayerNorm run at the HBM roofline.  Fused, a row panel's hidden activation never leaves the registers:
//
//   * a workgroup = 4 waves (one per SIMD, so each wave may use the whole 512-entry register file) owns a panel of
//     128 rows, wave w rows 32w .. 32w+31 as two MFMA column tiles.  The wave normalises its rows straight from the f32
//     residual stream into MFMA B fragments (96 VGPRs, k = all 384 channels) — LayerNorm costs no extra pass.
//   * the hidden dimension is walked in 48 chunks of 32 units.  Per chunk GEMM1 (A = 32 rows of W1 x 384 k, 48 MFMAs)
//     leaves, per lane, 8 consecutive hidden units of one row in the accumulators (weight rows are permuted while
//     staging, as in gemm2.hip); bias is the MFMA's C operand, GELU and the bf16 rounding happen in registers and the
//     result IS the B fragment (k = 8 (lane>>4) + e) of GEMM2 (A = 384 rows of W2 x 32 k, 48 MFMAs) — no LDS round trip.
//     The [32 rows x 384] f32 output tile of the wave lives in 192 accumulator registers for the whole panel.
//   * only the weights stream: chunk c = W1[32c..32c+31][:] + W2[:, 32c..32c+31] (24 KiB each, stored in memory as the LDS
//     images themselves, packed at load time) go global -> LDS by LDS-DMA into a 3-slot ring, two chunks ahead, ONE barrier
//     per chunk (= per 96 MFMAs of a wave).  Every workgroup streams the same 2.4 MB in the same order: L2 hits.
//   * epilogue: + bias2 + residual (f32, re-read), f32 store (128 bytes per row per instruction), and the next
//     LayerNorm (next block's norm1, or the encoder's final norm) on the rows still in registers.
//
// Rounding points are those of the separate kernels: LayerNorm output and the hidden activation are rounded to bf16,
// sums are fp32, the residual stream is fp32.  Results differ from the unfused path by fp32 summation order only.
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int E = 384, HID = 1536, CH = 32, NCH = HID / CH;   // 48 chunks of 32 hidden units
constexpr int W1B = CH * E * 2;                               // W1 chunk image [3 k-segments][32 rows][256 B]
constexpr int W2B = E * CH * 2;                               // W2 chunk image [384 rows][64 B]
constexpr int SLOT = W1B + W2B, NSLOT = 3;
constexpr int LUT_OFF = NSLOT * SLOT;                         // GELU table, 8 KiB
constexpr int B1_OFF = LUT_OFF + 8192;                        // fc1 bias, f32 [1536]
constexpr int MLP_LDS = B1_OFF + HID * 4;                     // 161,792 B of the 160 KiB
constexpr int BM = 128;
static_assert(MLP_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t m_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// LDS reads as inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it can see once LDS-DMA loads are in
// flight (it cannot prove they do not alias), which would serialise the weight stream.  Waits are placed by hand below.
#define MLP_RDV(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define MLP_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// the lane id straight from the hardware (2 VALU): anything derived from threadIdx that lives across the chunk loop is a spill candidate
__device__ __forceinline__ unsigned mlp_lane_v() {   // volatile: hipcc cannot hoist what is derived from it out of the chunk loop (and then spill it:
  unsigned l;                                          // a spill reload inside the loop waits with vmcnt(0), i.e. for the whole weight prefetch)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
#define MLP_LANE() mlp_lane_v()
#define MLP_WAIT8(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
// timing experiments of the STAMPS build (MlpParams::ablate; results are wrong): 1 = per-channel vectors (biases, LayerNorm gains)
// not loaded, 2 = no epilogue stores, 4 = attention output / residual rows of the front not loaded
#define MLP_VEC4(ptr) (((STAMPS >> 1) & 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(ptr))
__device__ __forceinline__ float4 mlp_fake4(unsigned k) {   // pseudo-random values in [-2, 2): the no-load experiment must keep the MFMA operands toggling
  const unsigned h = k * 2654435761u + 12345u;
  return make_float4((float)((int)(h & 1023u) - 512) * (1.f / 256), (float)((int)((h >> 10) & 1023u) - 512) * (1.f / 256),
                     (float)((int)((h >> 20) & 1023u) - 512) * (1.f / 256), (float)((int)((h >> 5) & 1023u) - 512) * (1.f / 256));
}
#define MLP_ROW4(ptr) (((STAMPS >> 1) & 4) ? (((STAMPS >> 1) & 8) ? mlp_fake4((unsigned)(size_t)(ptr)) : make_float4(.5f, .25f, 1.f, 2.f)) : *reinterpret_cast<const float4*>(ptr))
}  // namespace

// PROJ: the block's attention output projection runs in front, in the same launch: x' = x + att . Wp^T + bp is accumulated in the
// registers that then hold the MLP's output tile, so the residual stream is read once and written once per block (12 more ring
// items per panel: the [384 x 32] k-step slabs of a chunk-major copy of Wp, in the W2 half of a slot)
// STAMPS: a build with the phase stamps of tools/mlp_stamps.py (kept out of the production instantiations: the stamp address is one
// more value for the register allocator to spill inside the chunk loop)
template <bool PROJ, int STAMPS>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(MlpParams p) {
  constexpr int NPJ = PROJ ? E / CH : 0, ITEMS = NPJ + NCH + 1;   // ring items per panel
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const int npanels = (p.M + BM - 1) / BM;
  if ((int)blockIdx.x >= npanels) return;
  const int my_n = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_n * ITEMS;                              // ring items this workgroup walks (see the chunk loop)

  if (p.stagger) {                                             // start late by group (see MlpParams::stagger)
    const int gi = ((int)blockIdx.x >> 3) % (p.stagger >> 16);
    const long long wait = (long long)gi * (p.stagger & 0xffff) * 100;   // wall clock: 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(64);
  }
  for (int i = tid; i < 512; i += 256) reinterpret_cast<uint4*>(smem + LUT_OFF)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  for (int i = tid; i < HID / 4; i += 256) reinterpret_cast<float4*>(smem + B1_OFF)[i] = reinterpret_cast<const float4*>(p.b1)[i];

  // ---- weight stream: this wave's 6 + 6 one-KiB pieces of a chunk; source offsets are relative to the chunk's base
  const __amdgpu_buffer_rsrc_t rs1 = m_rsrc(p.w1p, (unsigned)(HID * E * 2));
  const __amdgpu_buffer_rsrc_t rs2 = m_rsrc(p.w2p, (unsigned)(HID * E * 2));
  // Both weight operands are stored in memory as the LDS images themselves (packed once at load time: pack_mlp_w1 / pack_mlp_w2
  // in engine.cpp), so a piece is 1 KiB of contiguous memory — the address unit serves those ~1.5x faster than 8 x 128-byte
  // gathers (tools/micro/dma_bench.hip) — and a lane's source offset is just 16 * lane.
  //   W1 image of a chunk, segment-major: [3 k-segments of 256 B][32 rows][256 B]; LDS row R = 16 jj + q' holds hidden unit
  //   (q'>>2)*8 + jj*4 + (q'&3) of the chunk, and 16-byte chunk c (0..15) of a segment sits at position c ^ (R & 15) (every
  //   row of a segment starts on bank 0).  Piece pp = 8 s + r = rows 4r .. 4r+3 of segment s.
  //   W2 image of a chunk: LDS row R' = 16 ot + q' holds output channel (ot>>1)*32 + (q'>>2)*8 + (ot&1)*4 + (q'&3); chunk gch of
  //   the 64-byte row sits at position gch ^ ((R'>>1) & 3).  Piece ot = rows 16 ot .. 16 ot + 15.
  const __amdgpu_buffer_rsrc_t rsp = m_rsrc(PROJ ? (const void*)p.wpp : (const void*)p.w2p, (unsigned)((PROJ ? E * E : HID * E) * 2));
  // Per-channel vectors (biases, LayerNorm gains and offsets: 1536 B each) ride in the W1 half of the two ring items that do not use
  // it — the last slab of Wp (bp | ln_g | ln_b for the front) and the panel's last item (b2 | nln_g | nln_b for the epilogue) —
  // 2 KiB apart.  Read per lane from global memory they cost 288 wave-wide 16-byte loads per wave and panel, 17 % of the launch
  // (tools/mlp_ablate_sweep.sh); from LDS they are broadcast reads.  Wave w fetches both 1-KiB halves of vector w % 3 (wave 3
  // repeats wave 0's: every wave issues the same number of pieces); lanes behind byte 1536 are out of range and write zeros.
  auto issue_vec = [&](int slot, const float* v0, const float* v1, const float* v2) {
    const int v = wave == 3 ? 0 : wave;
    const float* src = v == 0 ? v0 : (v == 1 ? v1 : v2);
    const __amdgpu_buffer_rsrc_t rv = m_rsrc(src, (unsigned)(E * 4));
    const unsigned src_lane = MLP_LANE() * 16u;
    unsigned char* d = smem + slot * SLOT + v * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)d, 16, src_lane, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr)(d + 1024), 16, src_lane + 1024u, 0, 0, 0);   // (in the vector offset: that is what the range check sees)
  };
  auto issue = [&](int Gi) {                                   // item Gi of the launch
    int ii = Gi % ITEMS;
    const int slot = Gi % NSLOT;
    const unsigned src_lane = MLP_LANE() * 16u;                 // (re-read per call: see mlp_lane_v)
    unsigned char* sb = smem + slot * SLOT + wave * 1024;
    if (PROJ && ii < NPJ) {                                     // k-step slab ii of Wp
#pragma unroll
      for (int j = 0; j < 6; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lds_ptr)(sb + W1B + j * 4096), 16, src_lane, ii * W2B + (wave + 4 * j) * 1024, 0, 0);
      if (ii == NPJ - 1) issue_vec(slot, p.bp, p.ln_g, p.ln_b);
      return;
    }
    ii -= NPJ;                                                  // MLP item: W1 chunk ii (ii < 48) and W2 chunk ii - 1 (ii >= 1)
    if (ii < NCH) {
      const int cb = ii * W1B;
#pragma unroll
      for (int j = 0; j < 6; ++j)   // piece 8 (j >> 1) + wave + 4 (j & 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sb + (j >> 1) * 8192 + (j & 1) * 4096), 16, src_lane, cb + (j >> 1) * 8192 + (wave + 4 * (j & 1)) * 1024, 0, 0);
    }
    if (ii >= 1) {
      const int cb = (ii - 1) * W2B;
#pragma unroll
      for (int j = 0; j < 6; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_ptr)(sb + W1B + j * 4096), 16, src_lane, cb + (wave + 4 * j) * 1024, 0, 0);
    }
    if (ii == NCH) issue_vec(slot, p.b2, p.nln_out ? p.nln_g : p.b2, p.nln_out ? p.nln_b : p.b2);
  };
  auto pieces = [&](int Gi) -> int {                            // pieces per wave of item Gi: 12; 6 for the one-operand items, 8 for those that carry vectors
    const int ii = Gi % ITEMS;
    return (ii == NPJ - 1 || ii == ITEMS - 1) ? 8 : ((ii < NPJ || ii == NPJ) ? 6 : 12);
  };

  // ---- fragment read addresses relative to a slot
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const unsigned lut_lds = lds0 + LUT_OFF;
  const unsigned b1_lds = lds0 + B1_OFF + (unsigned)(g * 8 * 4);

  issue(0);
  if (total > 1) issue(1);
  __syncthreads();                                             // table and bias staged

  int G = 0;
  for (int pi = 0; pi < my_n; ++pi) {
    const int panel = (int)blockIdx.x + pi * (int)gridDim.x;
    const int row0 = panel * BM + wave * 32;
#define MLP_PSTAMP(k) do { if ((STAMPS & 1) && p.dbg && blockIdx.x == 0 && tid == 0 && pi < 4) { p.dbg[384 + pi * 4 + (k)] = __builtin_readcyclecounter(); p.dbg[400 + pi * 4 + (k)] = wall_clock64(); } } while (0)   // shader clock | 100 MHz wall clock
    MLP_PSTAMP(0);   // panel start | chunk loop start | chunk loop end | panel end

    bf16x8 xf[2][12];                                           // LayerNorm_2 of this wave's 32 rows as MFMA B fragments
    f32x4 acc2[2][24];                                          // the wave's [32 rows x 384] f32 output tile
    int i = 0;
#define MLP_STAMP(ph) do { if ((STAMPS & 1) && p.dbg && blockIdx.x == 0 && tid == 0 && pi == 0 && i < NCH) p.dbg[i * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
// Fragment groups of 4 (8 MFMAs each), three register buffers, reads two groups ahead.  GEMM1 group n = k-steps 2n, 2n+1 x the
// two 16-row weight tiles; GEMM2 group n = output-channel tiles 4n .. 4n+3.
#define MLP_G1(dst, n)                                                                                                   \
  MLP_RD128(dst[0], a1v[(2 * (n)) & 3], ((2 * (n)) >> 2) * 8192);     MLP_RD128(dst[1], a1v[(2 * (n)) & 3], 4096 + ((2 * (n)) >> 2) * 8192); \
  MLP_RD128(dst[2], a1v[(2 * (n) + 1) & 3], ((2 * (n) + 1) >> 2) * 8192); MLP_RD128(dst[3], a1v[(2 * (n) + 1) & 3], 4096 + ((2 * (n) + 1) >> 2) * 8192);
#define MLP_G2(dst, n)                                                                                                   \
  MLP_RD128(dst[0], a2, (4 * (n) + 0) * 1024); MLP_RD128(dst[1], a2, (4 * (n) + 1) * 1024);                              \
  MLP_RD128(dst[2], a2, (4 * (n) + 2) * 1024); MLP_RD128(dst[3], a2, (4 * (n) + 3) * 1024);
#define MLP_M1(src, n)                                                                                                   \
  _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                                                                       \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                     \
      _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                                   \
        accn[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[k2 * 2 + jj], xf[rt][2 * (n) + k2], accn[rt][jj], 0, 0, 0);
#define MLP_M2B(src, n, b0, b1)                                                                                          \
  _Pragma("unroll") for (int o4 = 0; o4 < 4; ++o4) {                                                                     \
    acc2[0][4 * (n) + o4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[o4], b0, acc2[0][4 * (n) + o4], 0, 0, 0);        \
    acc2[1][4 * (n) + o4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(src[o4], b1, acc2[1][4 * (n) + o4], 0, 0, 0);        \
  }
#define MLP_M2(src, n) MLP_M2B(src, n, hf[0], hf[1])
#define MLP_WAITF(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#define MLP_WAITFT(n, f, t) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]))
    // GELU of one quarter (4 hidden units of one row tile), rounded to bf16 into half of the row tile's B fragment.
    // Table form (default): A = index + read request, B = interpolate (1024 knots of Phi on [-8, 8), |dPhi| < 7e-6), multiply.
    // MLP_GELU_POLY 1 (measured, slower: 618 vs 555 us per launch): gelu(x) = 0.5 x + |x| S(min(|x|, 5)), S(u) = Phi(u) - 0.5 as a
    // degree-12 polynomial in t = 0.4 u - 1 (Chebyshev fit on [0, 5], fp32 Horner: |gelu - exact| < 2.1e-6 over [-10, 10]) - no LDS
    // read, but 96 packed FMAs + ~50 more vector-ALU instructions per chunk than the table form's ~110, and every one of them is
    // issue time of the wave's serial stream here (the isolated loop of tools/micro/mlp_loop.hip hides such work; this kernel does not).
#ifndef MLP_GELU_POLY
#define MLP_GELU_POLY 0
#endif
    float u4[4];
    float2 t4[4];
#if MLP_GELU_POLY
    for (int e = 0; e < 4; ++e) { u4[e] = 0.f; t4[e] = make_float2(0.f, 0.f); }   // (only named by the waits' register lists)
    auto gelu_a = [&](const f32x4&) {};
    auto gelu_b = [&](const f32x4& acc, bf16x8& o, int half) {
      constexpr float kC[13] = {0.49378976225852966f, 0.043821267783641815f, -0.13688436150550842f, 0.23961056768894196f, -0.23270182311534882f,
                                0.06559479981660843f, 0.13090801239013672f, -0.16348905861377716f, 0.025033878162503242f, 0.07832171767950058f,
                                -0.04101016744971275f, -0.013859635218977928f, 0.01086505502462387f};
      float t[4], s[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { t[e] = fmaf(fminf(fabsf(acc[e]), 5.0f), 0.4f, -1.0f); s[e] = kC[12]; }
#pragma unroll
      for (int k = 11; k >= 0; --k)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = fmaf(s[e], t[e], kC[k]);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[half * 4 + e] = (bf16)fmaf(fabsf(acc[e]), s[e], 0.5f * acc[e]);
    };
#else
    auto gelu_a = [&](const f32x4& acc) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u4[e] = fmaf(__builtin_amdgcn_fmed3f(acc[e], -8.0f, 7.984375f), 64.0f, 512.0f);
        const unsigned ad = ((STAMPS >> 1) & 16) ? lut_lds + (((unsigned)(int)u4[e] << 3) & 8u) : lut_lds + ((unsigned)(int)u4[e] << 3);   // (16: timing experiment, two table entries only)
        asm volatile("ds_read_b64 %0, %1" : "=v"(t4[e]) : "v"(ad));
      }
    };
    auto gelu_b = [&](const f32x4& acc, bf16x8& o, int half) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[half * 4 + e] = (bf16)(acc[e] * fmaf(__builtin_amdgcn_fractf(u4[e]), t4[e].y, t4[e].x));
    };
#endif

    // Software pipeline over the 48 hidden chunks: iteration i runs GEMM1 of chunk i with the GELU of chunk i-1 in the
    // shadow of its MFMAs, then GEMM2 of chunk i-1.  Ring item G (49 per panel, G counts over the whole launch) =
    // {W1 chunk i if i < 48, W2 chunk i-1 if i >= 1} in slot G % 3.
    f32x4 accp[2][2];                                           // GEMM1 result of the previous iteration, before the activation
    f32x4 accn[2][2];
    bf16x8 f0[4], f1[4], f2[4];
    bf16x8 hf[2];
    unsigned a1v[4], a2;
    // top of an iteration: item G has landed once at most the pieces of item G + 1 (12, or 6 at a panel edge) are still in
    // flight — loads retire in order, and whatever the epilogue / prologue put on the queue since is younger still
#define MLP_TOP()                                                                                          \
  {                                                                                                        \
    MLP_STAMP(0);                                                                                          \
    if (G + 1 < total) {                                                                                   \
      const int pc_ = pieces(G + 1);                                                                       \
      if (pc_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                       \
      else if (pc_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                  \
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                               \
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_s_barrier(); /* everyone's pieces landed; slot (G+2)%3 = (G-1)%3 is free again */     \
    MLP_STAMP(1);                                                                                          \
    const unsigned sbase = lds0 + (unsigned)((G % NSLOT) * SLOT);                                          \
    /* fragment addresses are recomputed from the lane id every iteration (the empty asm stops the compiler from hoisting  \
       them out of the loop into registers it then spills: a spill reload is a vector-memory load, and the wait for it drains  \
       the LDS-DMA prefetch) */                                                                            \
    const unsigned ll = MLP_LANE();   /* read from the hardware, not kept in a register across the loop */ \
    const unsigned qq = ll & 15u, gq = ll >> 4;                                                            \
    const unsigned rowb = sbase + qq * 256u;                                                               \
    a1v[0] = rowb + (((0u + gq) ^ qq) << 4); a1v[1] = rowb + (((4u + gq) ^ qq) << 4);                      \
    a1v[2] = rowb + (((8u + gq) ^ qq) << 4); a1v[3] = rowb + (((12u + gq) ^ qq) << 4);                     \
    a2 = sbase + (unsigned)W1B + qq * 64u + ((gq ^ ((qq >> 1) & 3u)) << 4);                                \
  }
    // the first fragments are requested before the DMA issue, whose ~80 cycles per piece hide their latency.  (The CU's address
    // unit takes 16-20 cycles per 1-KiB piece and the four waves queue behind one another; issuing a third of the item after
    // every fourth MFMA group instead, one wave at a time, cost more in spills than it saved.)
#define MLP_HEAD1()                                                                                        \
  {                                                                                                        \
    const unsigned ba = b1_lds + (unsigned)(i * CH * 4);                                                   \
    MLP_RD128(accn[0][0], ba, 0);                                                                          \
    MLP_RD128(accn[0][1], ba, 16);                                                                         \
    MLP_G1(f0, 0)                                                                                          \
    MLP_G1(f1, 1)                                                                                          \
    if (G + 2 < total) issue(G + 2);                                                                       \
    MLP_STAMP(2);                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(accn[0][0]), "+v"(accn[0][1]), "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3])); \
    accn[1][0] = accn[0][0]; accn[1][1] = accn[0][1];                                                      \
  }
    if (!PROJ) {
    // ---- LayerNorm of this wave's 32 rows -> B fragments.  Lane (q, g) holds, of row 16 rt + q, channels 32 ks + 8 g + e.
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
        const float4 g0 = MLP_VEC4(p.ln_g + ks * 32 + g * 8), g1 = MLP_VEC4(p.ln_g + ks * 32 + g * 8 + 4);
        const float4 t0 = MLP_VEC4(p.ln_b + ks * 32 + g * 8), t1 = MLP_VEC4(p.ln_b + ks * 32 + g * 8 + 4);
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
      for (int ot = 0; ot < 24; ++ot) acc2[rt][ot] = f32x4{0.f, 0.f, 0.f, 0.f};

    } else {
      // ---- attention output projection: acc2 = att . Wp^T over 12 ring items (k-step slabs), then x' = acc2 + bp + x stays in
      // acc2 as the MLP's starting value and LayerNorm_2(x') becomes the GEMM1 operand.  Lane (q, g) loads, of row 16 rt + q,
      // channels 32 ks + 8 g + e of the attention output: the same fragment layout as xf.
      bf16x8 af[2][12];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = min(row0 + rt * 16 + q, p.M - 1);
        const bf16* ar = p.att + (size_t)row * E + g * 8;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
          if ((STAMPS >> 1) & 8) { const float4 a = mlp_fake4((unsigned)(size_t)(ar + ks * 32)), b = mlp_fake4((unsigned)(size_t)(ar + ks * 32) + 7u);
            af[rt][ks] = bf16x8{(bf16)a.x, (bf16)a.y, (bf16)a.z, (bf16)a.w, (bf16)b.x, (bf16)b.y, (bf16)b.z, (bf16)b.w}; }
          else af[rt][ks] = ((STAMPS >> 1) & 4) ? bf16x8{(bf16)1.f, (bf16).5f, (bf16)1.f, (bf16).5f, (bf16)1.f, (bf16).5f, (bf16)1.f, (bf16).5f} : *reinterpret_cast<const bf16x8*>(ar + ks * 32);
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ot = 0; ot < 24; ++ot) acc2[rt][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
      i = NCH;                                                  // (no stamps in this phase)
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        MLP_TOP()
        MLP_G2(f0, 0)
        MLP_G2(f1, 1)
        if (ks < 10) {                                          // item G + 2 is another slab of Wp (known at compile time)
          unsigned char* sb = smem + ((G + 2) % NSLOT) * SLOT + wave * 1024;
          const unsigned src_lane = MLP_LANE() * 16u;
#pragma unroll
          for (int j = 0; j < 6; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsp, (lds_ptr)(sb + W1B + j * 4096), 16, src_lane, (ks + 2) * W2B + (wave + 4 * j) * 1024, 0, 0);
          if (ks + 2 == NPJ - 1) issue_vec((G + 2) % NSLOT, p.bp, p.ln_g, p.ln_b);
        } else if (G + 2 < total) issue(G + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]));
        MLP_G2(f2, 2)  MLP_M2B(f0, 0, af[0][ks], af[1][ks])
        MLP_G2(f0, 3)  MLP_M2B(f1, 1, af[0][ks], af[1][ks])
        MLP_G2(f1, 4)  MLP_WAITF(8, f2);  MLP_M2B(f2, 2, af[0][ks], af[1][ks])
        MLP_G2(f2, 5)  MLP_WAITF(8, f0);  MLP_M2B(f0, 3, af[0][ks], af[1][ks])
        MLP_WAITF(4, f1);  MLP_M2B(f1, 4, af[0][ks], af[1][ks])
        MLP_WAITF(0, f2);  MLP_M2B(f2, 5, af[0][ks], af[1][ks])
        __builtin_amdgcn_sched_barrier(0);
        ++G;
      }
      i = 0;
      // bp | ln_g | ln_b came in with the last slab (item G - 1); lane (q, g) reads channels 32 pp + 8 g .. + 7 of each.
      // Pass 1 (both row tiles): x' = acc + bp + x back into the accumulators, with the row statistics gathered on the way as
      // sums shifted by the lane's first value (then merged over the row's four lanes as mean / M2 pairs) — a separate
      // variance pass would keep the 96 values of a row tile in vector registers next to xf, which hipcc answers with spills
      // (and a spill reload in front of the chunk loop waits for the weight prefetch).  Pass 2 reads x' back from the
      // accumulators, both row tiles per gain / offset read.
      const unsigned vfb = lds0 + (unsigned)(((G - 1) % NSLOT) * SLOT) + (MLP_LANE() >> 4) * 32u;
      float rstd2[2], nmr2[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const size_t ro = (size_t)min(row0 + rt * 16 + q, p.M - 1) * E + g * 8;
        float K = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int pb = 0; pb < 6; ++pb) {                        // batches of 2 channel groups
          f32x4 c[4];
          float4 r0[2], r1[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const unsigned a = vfb + (unsigned)((2 * pb + k) * 128);
            MLP_RDV(c[2 * k], a);  MLP_RDV(c[2 * k + 1], a + 16u);
            r0[k] = MLP_ROW4(p.x + ro + (2 * pb + k) * 32); r1[k] = MLP_ROW4(p.x + ro + (2 * pb + k) * 32 + 4);
          }
          MLP_WAITF(0, c);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int pp = 2 * pb + k;
            f32x4 lo = acc2[rt][2 * pp], hi = acc2[rt][2 * pp + 1];
            lo[0] += c[2 * k][0] + r0[k].x; lo[1] += c[2 * k][1] + r0[k].y; lo[2] += c[2 * k][2] + r0[k].z; lo[3] += c[2 * k][3] + r0[k].w;
            hi[0] += c[2 * k + 1][0] + r1[k].x; hi[1] += c[2 * k + 1][1] + r1[k].y; hi[2] += c[2 * k + 1][2] + r1[k].z; hi[3] += c[2 * k + 1][3] + r1[k].w;
            acc2[rt][2 * pp] = lo; acc2[rt][2 * pp + 1] = hi;
            if (pp == 0) K = lo[0];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d0 = lo[e] - K, d1 = hi[e] - K; S1 += d0 + d1; S2 = fmaf(d0, d0, S2); S2 = fmaf(d1, d1, S2); }
          }
          __builtin_amdgcn_sched_barrier(0);                    // (a batch's values are used up before the next batch is read)
        }
        // the lane's 96 values: mean_l = K + S1 / 96, M2_l = S2 - S1^2 / 96; the row: mean = average of the four mean_l,
        // M2 = sum of M2_l + 96 (mean_l - mean)^2
        const float ml = S1 * (1.f / 96), mean_l = K + ml;
        float ms = mean_l;
        ms += __shfl_xor(ms, 16); ms += __shfl_xor(ms, 32);
        const float mean = ms * 0.25f, dm = mean_l - mean;
        float m2 = fmaf(96.f * dm, dm, S2 - S1 * ml);
        m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
        rstd2[rt] = rsqrtf(m2 * (1.f / E) + p.ln_eps);
        nmr2[rt] = -mean * rstd2[rt];
      }
      // y = x' * (rstd * gain) + (offset - mean * rstd * gain); the empty asm ties the batch reads to the statistics, or they move
      // above pass 1 and their values get spilled
      unsigned vg = vfb + 2048u;
      asm volatile("" : "+v"(vg) : "v"(rstd2[1]));
#pragma unroll
      for (int pb = 0; pb < 6; ++pb) {                          // batches of 2 channel groups: gain lo | hi, offset lo | hi
        f32x4 gb[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const unsigned a = vg + (unsigned)((2 * pb + k) * 128);
          MLP_RDV(gb[4 * k], a);  MLP_RDV(gb[4 * k + 1], a + 16u);  MLP_RDV(gb[4 * k + 2], a + 2048u);  MLP_RDV(gb[4 * k + 3], a + 2064u);
        }
        MLP_WAIT8(0, gb);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = 2 * pb + k;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gn = e < 4 ? gb[4 * k][e] : gb[4 * k + 1][e - 4], of = e < 4 ? gb[4 * k + 2][e] : gb[4 * k + 3][e - 4];
              o[e] = (bf16)fmaf(e < 4 ? acc2[rt][2 * pp][e] : acc2[rt][2 * pp + 1][e - 4], rstd2[rt] * gn, fmaf(nmr2[rt], gn, of));
            }
            xf[rt][pp] = o;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MLP_PSTAMP(1);
    {                                                           // chunk 0: GEMM1 only
      MLP_TOP()
      MLP_HEAD1()
      MLP_G1(f2, 2)  MLP_M1(f0, 0)
      MLP_G1(f0, 3)  MLP_M1(f1, 1)
      MLP_G1(f1, 4)  MLP_WAITF(8, f2);  MLP_M1(f2, 2)
      MLP_G1(f2, 5)  MLP_WAITF(8, f0);  MLP_M1(f0, 3)
      MLP_WAITF(4, f1);  MLP_M1(f1, 4)
      MLP_WAITF(0, f2);  MLP_M1(f2, 5)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) { accp[rt][0] = accn[rt][0]; accp[rt][1] = accn[rt][1]; }
      MLP_STAMP(5);
      ++G;
    }
    // In the steady loop wave w issues its 12 pieces of item G + 2 after MFMA group w - 1 (wave 0 right behind the barrier): four
    // bursts hitting the address unit together behind the barrier cost 461 cycles per wave, free-running ones 233
    // (tools/micro/issue_cost.hip).  All pieces are on the queue long before the next top-of-iteration wait.
#define MLP_DMA_AT(t) if (more && wave == (t)) issue(G + 2);
    for (i = 1; i < NCH; ++i, ++G) {                             // GEMM1(i) with GELU(i-1) in its shadow, then GEMM2(i-1)
      MLP_TOP()
      const bool more = G + 2 < total;
      {
        const unsigned ba = b1_lds + (unsigned)(i * CH * 4);
        MLP_RD128(accn[0][0], ba, 0);
        MLP_RD128(accn[0][1], ba, 16);
        MLP_G1(f0, 0)
        MLP_G1(f1, 1)
        MLP_DMA_AT(0)
        MLP_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(accn[0][0]), "+v"(accn[0][1]), "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3]));
        accn[1][0] = accn[0][0]; accn[1][1] = accn[0][1];
      }
      // waits below: the group about to be multiplied and the table reads of the quarter about to be finished were issued
      // before the newest 4 fragment reads, and LDS returns in order
      MLP_G1(f2, 2)  gelu_a(accp[0][0]);                                   MLP_M1(f0, 0)  MLP_DMA_AT(1)
      MLP_G1(f0, 3)  MLP_WAITFT(4, f2, t4);  gelu_b(accp[0][0], hf[0], 0);  gelu_a(accp[0][1]);  MLP_M1(f1, 1)  MLP_DMA_AT(2)
      MLP_G1(f1, 4)  MLP_WAITFT(4, f0, t4);  gelu_b(accp[0][1], hf[0], 1);  gelu_a(accp[1][0]);  MLP_M1(f2, 2)  MLP_DMA_AT(3)
      MLP_G1(f2, 5)  MLP_WAITFT(4, f1, t4);  gelu_b(accp[1][0], hf[1], 0);  gelu_a(accp[1][1]);  MLP_M1(f0, 3)
      MLP_G2(f0, 0)  MLP_WAITFT(4, f2, t4);  gelu_b(accp[1][1], hf[1], 1);                       MLP_M1(f1, 4)
      MLP_G2(f1, 1)  MLP_WAITF(8, f2);  MLP_M1(f2, 5)
      MLP_STAMP(3);
      MLP_G2(f2, 2)  MLP_WAITF(8, f0);  MLP_M2(f0, 0)
      MLP_G2(f0, 3)  MLP_WAITF(8, f1);  MLP_M2(f1, 1)
      MLP_G2(f1, 4)  MLP_WAITF(8, f2);  MLP_M2(f2, 2)
      MLP_G2(f2, 5)  MLP_WAITF(8, f0);  MLP_M2(f0, 3)
      MLP_WAITF(4, f1);  MLP_M2(f1, 4)
      MLP_WAITF(0, f2);  MLP_M2(f2, 5)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) { accp[rt][0] = accn[rt][0]; accp[rt][1] = accn[rt][1]; }
      MLP_STAMP(5);
    }
    {                                                           // after the last chunk: GELU + GEMM2 of chunk 47
      MLP_TOP()
      MLP_G2(f0, 0)
      MLP_G2(f1, 1)
      if (G + 2 < total) issue(G + 2);
      gelu_a(accp[0][0]);  MLP_WAITFT(0, f0, t4);  gelu_b(accp[0][0], hf[0], 0);
      gelu_a(accp[0][1]);  MLP_WAITFT(0, f1, t4);  gelu_b(accp[0][1], hf[0], 1);
      gelu_a(accp[1][0]);  MLP_WAITFT(0, f0, t4);  gelu_b(accp[1][0], hf[1], 0);
      gelu_a(accp[1][1]);  MLP_WAITFT(0, f1, t4);  gelu_b(accp[1][1], hf[1], 1);
      MLP_G2(f2, 2)  MLP_M2(f0, 0)
      MLP_G2(f0, 3)  MLP_M2(f1, 1)
      MLP_G2(f1, 4)  MLP_WAITF(8, f2);  MLP_M2(f2, 2)
      MLP_G2(f2, 5)  MLP_WAITF(8, f0);  MLP_M2(f0, 3)
      MLP_WAITF(4, f1);  MLP_M2(f1, 4)
      MLP_WAITF(0, f2);  MLP_M2(f2, 5)
      __builtin_amdgcn_sched_barrier(0);
      ++G;
    }

    MLP_PSTAMP(2);
    // ---- epilogue: + bias2 (+ residual) -> f32 store, then the next LayerNorm on the values still in registers.  Lane (q, g)
    // holds, of row 16 rt + q, channels 32 pp + 8 g + e (pp = 0..11).  b2 | nln_g | nln_b came in with the panel's last item
    // (G - 1), in the W1 half of its slot.  Pass 1 per row tile, in batches of 2 channel groups: the sum goes back into the
    // accumulators and out to x_out, the row statistics are gathered on the way (shifted sums, as in the front).
    const unsigned veb = lds0 + (unsigned)(((G - 1) % NSLOT) * SLOT) + (MLP_LANE() >> 4) * 32u;
    float rstd2[2], nmr2[2];
    size_t ro2[2];
    bool live2[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + rt * 16 + q;
      const bool live = row < p.M;
      const size_t ro = (size_t)min(row, p.M - 1) * E + g * 8;
      ro2[rt] = ro; live2[rt] = live;
      // f32 stores: a lane holds 8 consecutive channels = 2 x 16 B, so storing them as they are writes four 16-byte pieces with
      // 16-byte holes per row per instruction.  v_permlane32_swap (lanes i <-> i + 32, i.e. g <-> g + 2) of the lower half's second
      // quad with the upper half's first quad makes each instruction write 64 contiguous bytes per row:
      //   first instruction:  g = 0, 1, 2, 3 -> bytes 0-15, 32-47, 16-31, 48-63 of the row's 128;   second: the same + 64
      float* const ob = p.x_out + ro - g * 8 + ((g & 1) * 8 + (g >> 1) * 4);     // row base + this lane's 16-byte slot of the first half
      float K = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
      for (int pb = 0; pb < 6; ++pb) {
        f32x4 c[4];
        float4 r0[2], r1[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = 2 * pb + k;
          const unsigned a = veb + (unsigned)(pp * 128);
          MLP_RDV(c[2 * k], a);  MLP_RDV(c[2 * k + 1], a + 16u);
          if (!PROJ) { r0[k] = MLP_ROW4(p.x + ro + pp * 32); r1[k] = MLP_ROW4(p.x + ro + pp * 32 + 4); }   // PROJ: the residual is already inside acc2
        }
        MLP_WAITF(0, c);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = 2 * pb + k;
          f32x4 lo = acc2[rt][2 * pp], hi = acc2[rt][2 * pp + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[e] += c[2 * k][e]; hi[e] += c[2 * k + 1][e]; }
          if (!PROJ) { lo[0] += r0[k].x; lo[1] += r0[k].y; lo[2] += r0[k].z; lo[3] += r0[k].w; hi[0] += r1[k].x; hi[1] += r1[k].y; hi[2] += r1[k].z; hi[3] += r1[k].w; }
          acc2[rt][2 * pp] = lo; acc2[rt][2 * pp + 1] = hi;
          if (pp == 0) K = lo[0];
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d0 = lo[e] - K, d1 = hi[e] - K; S1 += d0 + d1; S2 = fmaf(d0, d0, S2); S2 = fmaf(d1, d1, S2); }
          typedef __attribute__((ext_vector_type(4))) float f4;
          f4 slo, shi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a_ = lo[e], b_ = hi[e];                    // (the pair-returning builtin gave hi == lo here)
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
            slo[e] = a_; shi[e] = b_;
          }
          if (live && !p.no_x_store && !((STAMPS >> 1) & 2)) {
            float* op = ob + pp * 32;
            if (p.store_nt) { __builtin_nontemporal_store(slo, reinterpret_cast<f4*>(op)); __builtin_nontemporal_store(shi, reinterpret_cast<f4*>(op + 16)); }
            else { *reinterpret_cast<f4*>(op) = slo; *reinterpret_cast<f4*>(op + 16) = shi; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      const float ml = S1 * (1.f / 96), mean_l = K + ml;
      float ms = mean_l;
      ms += __shfl_xor(ms, 16); ms += __shfl_xor(ms, 32);
      const float mean = ms * 0.25f, dm = mean_l - mean;
      float m2 = fmaf(96.f * dm, dm, S2 - S1 * ml);
      m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
      rstd2[rt] = rsqrtf(m2 * (1.f / E) + p.nln_eps);
      nmr2[rt] = -mean * rstd2[rt];
    }
    if (p.nln_out) {                                            // uniform
      unsigned vg = veb + 2048u;
      asm volatile("" : "+v"(vg) : "v"(rstd2[1]));
#pragma unroll
      for (int pb = 0; pb < 6; ++pb) {                          // batches of 2 channel groups: gain lo | hi, offset lo | hi
        f32x4 gb[8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const unsigned a = vg + (unsigned)((2 * pb + k) * 128);
          MLP_RDV(gb[4 * k], a);  MLP_RDV(gb[4 * k + 1], a + 16u);  MLP_RDV(gb[4 * k + 2], a + 2048u);  MLP_RDV(gb[4 * k + 3], a + 2064u);
        }
        MLP_WAIT8(0, gb);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int pp = 2 * pb + k;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float gn = e < 4 ? gb[4 * k][e] : gb[4 * k + 1][e - 4], of = e < 4 ? gb[4 * k + 2][e] : gb[4 * k + 3][e - 4];
              o[e] = (bf16)fmaf(e < 4 ? acc2[rt][2 * pp][e] : acc2[rt][2 * pp + 1][e - 4], rstd2[rt] * gn, fmaf(nmr2[rt], gn, of));
            }
            if (live2[rt] && !((STAMPS >> 1) & 2)) { if (p.store_nt) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p.nln_out + ro2[rt] + pp * 32)); else *reinterpret_cast<bf16x8*>(p.nln_out + ro2[rt] + pp * 32) = o; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MLP_PSTAMP(3);
  }
}

static unsigned long long* g_mlp_dbg = nullptr;
static int g_mlp_store_nt = 1;   // streaming policy on the epilogue's residual / LayerNorm stores: -0.25 ms per 32-page step
void set_mlp_store_nt(int v) { g_mlp_store_nt = v; }
void set_mlp_stamps(unsigned long long* d) { g_mlp_dbg = d; }
static int g_mlp_stagger = 0, g_mlp_ablate = 0;
void set_mlp_ablate(int v) { g_mlp_ablate = v; }
void set_mlp_stagger(int v) { g_mlp_stagger = (v >> 16) > 1 ? v : 0; }

const char* mlp_fused_check(const MlpParams& p) {
  if (p.M <= 0) return "mlp_fused: bad row count";
  if (!p.x || !p.x_out || !p.ln_g || !p.ln_b || !p.w1p || !p.b1 || !p.w2p || !p.b2 || !p.gelu_lut) return "mlp_fused: null operand";
  if (p.nln_out && (!p.nln_g || !p.nln_b)) return "mlp_fused: next LayerNorm parameters";
  if (p.no_x_store && !p.nln_out) return "mlp_fused: no output";
  if (p.att && (!p.wpp || !p.bp || (((uintptr_t)p.att | (uintptr_t)p.wpp | (uintptr_t)p.bp) & 15))) return "mlp_fused: projection operands";
  const uintptr_t a = (uintptr_t)p.x | (uintptr_t)p.x_out | (uintptr_t)p.ln_g | (uintptr_t)p.ln_b | (uintptr_t)p.w1p | (uintptr_t)p.b1 | (uintptr_t)p.w2p |
                      (uintptr_t)p.b2 | (uintptr_t)p.nln_out | (uintptr_t)p.nln_g | (uintptr_t)p.nln_b | (uintptr_t)p.gelu_lut;
  if (a & 15) return "mlp_fused: operands must be 16-byte aligned";
  return nullptr;
}

void launch_mlp_fused(const MlpParams& p_in, hipStream_t s) {
  MlpParams p = p_in;
  p.gelu_lut = gelu_lut_for_current_device();
  p.dbg = g_mlp_dbg; p.store_nt = g_mlp_store_nt; p.stagger = g_mlp_stagger; p.ablate = g_mlp_ablate;
  if (const char* e = mlp_fused_check(p)) throw std::runtime_error(e);
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS));
    TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS)); });
  const int cus = device_cu_count(256);
  const int npanels = (p.M + BM - 1) / BM;
  const dim3 grid(std::min(cus, npanels));
#ifdef MLP_ABLATE_BUILDS
  if (p.att && p.ablate && !p.dbg) {   // production code + one ablation, timed from outside (rocprofv3)
#define ABL(a) if (p.ablate == a) { static PerDeviceOnce o; o.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<true, 2 * a>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS)); }); hipLaunchKernelGGL((mlp_fused_kernel<true, 2 * a>), grid, dim3(256), MLP_LDS, s, p); return; }
    ABL(2) ABL(12) ABL(14)
  }
#endif
  if (p.dbg) {
    if (p.att) hipLaunchKernelGGL((mlp_fused_kernel<true, 1>), grid, dim3(256), MLP_LDS, s, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<false, 1>), grid, dim3(256), MLP_LDS, s, p);
  } else if (p.att) hipLaunchKernelGGL((mlp_fused_kernel<true, 0>), grid, dim3(256), MLP_LDS, s, p);
  else hipLaunchKernelGGL((mlp_fused_kernel<false, 0>), grid, dim3(256), MLP_LDS, s, p);
}

}  // namespace ttr
