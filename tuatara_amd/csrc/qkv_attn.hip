// PARSeq ViT encoder: qkv projection + self-attention of one head in ONE kernel, bf16 (timm Attention.forward inside the
// TorchScript module called at tuatara.cpp:307):
//
//   out[n][t][64h + d] = softmax_k( Q_h[t] . K_h[k] / 8 ) V_h[k][d],   [Q_h | K_h | V_h] = LN1(x)[n] . W_h^T + b_h
//
// As separate launches (gemm_ws qkv + attn_enc2) the [M][1152] qkv tensor is written and read back: 0.75 GB of the
// 1.26 GB the pair moves per block at 1280 crops.  Here a persistent workgroup of 4 waves (one per SIMD, 512 registers each)
// owns ONE head for a share of the crops:
//   * its 192 weight rows (Wq_h, Wk_h, Wv_h: 147 KB) live in registers for the whole launch — wave w the MFMA A fragments of
//     output columns 48w .. 48w+47 (36 fragments = 144 VGPRs), bias as the MFMA's C operand;
//   * a crop's LayerNorm output [128 x 384] streams through a 2-slot LDS ring in two 64-row halves (LDS-DMA, k-step-major
//     image as in gemm_ws.hip); per half every wave runs 144 MFMAs (each activation fragment feeds its 3 column tiles) and drops
//     the rounded bf16 results straight into the Q / K / V tiles in LDS in the layout attn_enc2.hip reads (Q, K chunk-swizzled,
//     K rows permuted so a lane of the score MFMA ends up with 8 consecutive keys);
//   * the attention itself is attn_enc2.hip's: S^T = K Q^T, in-register softmax -> P fragments, V^T by transposed LDS reads,
//     128-byte output rows.  The next crop's first half is already streaming in meanwhile.
// The six head workgroups of a crop group sit on one XCD (blockIdx % 8) and read the same activation rows: one HBM fetch
// (40 groups on 30 of an XCD's 32 CUs; two more groups on the left-over CUs).
// Rounding points are those of the separate kernels (q, k, v rounded to bf16; P rounded to bf16; output bf16).
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
constexpr int S = 128, DH = 64, E = 384, NH = 6;
constexpr int TILE = S * DH * 2;            // 16 KiB per Q / K / V tile
constexpr int XHALF = 64 * E * 2;           // 48 KiB: one 64-row half of a crop's activations, [6 k-steps of 64][64 rows][128 B]
constexpr int QA_LDS = 3 * TILE + 2 * XHALF;   // 147,456 B
constexpr int GROUPS_PER_XCD = 5;           // 5 crop groups x 6 heads = 30 of an XCD's 32 CUs
#define QA_RD128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
}  // namespace

__global__ __launch_bounds__(256, 1) void qkv_attn_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w, const float* __restrict__ bias,
                                                         bf16* __restrict__ out, int N) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const sQ = smem;
  unsigned char* const sK = smem + TILE;
  unsigned char* const sV = smem + 2 * TILE;
  unsigned char* const sX = smem + 3 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;

  // workgroups b and b + 8 share an XCD (speed only): XCD x runs crop groups 5x .. 5x+4, six head workgroups each
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  // ... and the two left-over CUs of every XCD (16 workgroups) run the twelve head workgroups of crop groups 40 and 41, spread over
  // the XCDs (their activations are fetched once per head instead of once per group): 42 groups, 31 crops each at 1280 instead of 32
  constexpr int NGROUPS = 8 * GROUPS_PER_XCD + 2;
  int h, group;
  if (slot < GROUPS_PER_XCD * NH) { h = slot % NH; group = xcd * GROUPS_PER_XCD + slot / NH; }
  else {
    const int sp = xcd * 2 + (slot - GROUPS_PER_XCD * NH);
    if (sp >= 2 * NH) return;
    h = sp % NH; group = 8 * GROUPS_PER_XCD + sp / NH;
  }
  if (group >= N) return;
  const int ncrops = (N - group + NGROUPS - 1) / NGROUPS;     // crops group, group + 42, ...

  // ---- resident weights: output column c = 48 wave + 16 ct + row of the head's [Q | K | V] block (64 each)
  bf16x8 fw[3][12];
  f32x4 fb[3];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int c = 48 * wave + 16 * ct + q, part = c >> 6, within = c & 63;
    const bf16* wp = w + (size_t)(part * E + DH * h + within) * E + g * 8;
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) fw[ct][ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
    const int cb = 48 * wave + 16 * ct + 4 * g, pb = cb >> 6, wb = cb & 63;
    const float4 bv = *reinterpret_cast<const float4*>(bias + pb * E + DH * h + wb);
    fb[ct] = f32x4{bv.x, bv.y, bv.z, bv.w};
  }

  // ---- activation stream: half Hh = 2 * (crop index) + (0 | 1) goes to ring slot Hh & 1; wave w loads rows 8 (w + 4 j) .. + 7 of
  // every k-step sub-tile (j = 0, 1): LDS chunk lane&7 of row r holds global chunk (lane&7) ^ ((r>>1)&7)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(x), 0, (int)((size_t)N * S * E * 2), 0x00020000);
  auto issue_half = [&](int Hh) {
    const int crop = group + (Hh >> 1) * NGROUPS;
    unsigned char* sb = sX + (Hh & 1) * XHALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 8 * (wave + 4 * j) + (lane >> 3);
      const unsigned base = (unsigned)(((crop * S + (Hh & 1) * 64 + r) * E) * 2) + (unsigned)((((lane & 7) ^ ((r >> 1) & 7))) * 16);
#pragma unroll
      for (int ks = 0; ks < 6; ++ks)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sb + ks * 8192 + (wave + 4 * j) * 1024), 16, base + (unsigned)(ks * 128), 0, 0, 0);
    }
  };
  const int nhalves = 2 * ncrops;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  const unsigned frag_lane = (unsigned)(q * 128 + ((g ^ ((q >> 1) & 7)) << 4));   // fragment row q of a 16-row tile, chunk g (k-step even)
  const int swz = (q >> 1) & 7;

  issue_half(0);
  for (int ci = 0; ci < ncrops; ++ci) {
    const int crop = group + ci * NGROUPS;
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
      const int Hh = 2 * ci + hf;
      // this wave's 12 pieces of half Hh have landed; at the first half of a later crop the previous crop's 4 output stores are
      // younger than they are (the half was requested before the attention ran) and may stay in flight
      if (hf == 0 && ci > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                               // everyone's landed; the other slot and (hf == 0) the Q/K/V tiles are free
      // wave w requests its 12 pieces of the next half after k-step 2w - 1 (wave 0 right here): four bursts arriving at the address
      // unit together behind the barrier cost each wave twice what free-running ones do (tools/micro/issue_cost.hip)
      const bool moreh = Hh + 1 < nhalves;
#define QA_DMA_AT(wv) if (moreh && wave == (wv)) issue_half(Hh + 1);
      QA_DMA_AT(0)
      // ---- [64 rows x 192 columns] = X_half . W_h^T: 12 k-steps of 32, 4 row tiles x 3 column tiles
      const unsigned xb = lds0 + (unsigned)(3 * TILE + (Hh & 1) * XHALF) + frag_lane;
      f32x4 acc[4][3];
      bf16x8 fx[2][4];
#define QA_LOADX(buf, st)                                                                                   \
  QA_RD128(fx[buf][0], xb ^ (((st) & 1) * 64), ((st) >> 1) * 8192 + 0 * 2048);                                \
  QA_RD128(fx[buf][1], xb ^ (((st) & 1) * 64), ((st) >> 1) * 8192 + 1 * 2048);                                \
  QA_RD128(fx[buf][2], xb ^ (((st) & 1) * 64), ((st) >> 1) * 8192 + 2 * 2048);                                \
  QA_RD128(fx[buf][3], xb ^ (((st) & 1) * 64), ((st) >> 1) * 8192 + 3 * 2048);
#define QA_WAITX(n, buf) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(fx[buf][0]), "+v"(fx[buf][1]), "+v"(fx[buf][2]), "+v"(fx[buf][3]))
#define QA_MM(buf, st)                                                                                      \
  _Pragma("unroll") for (int rt = 0; rt < 4; ++rt)                                                          \
    _Pragma("unroll") for (int ct = 0; ct < 3; ++ct)                                                        \
      acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ct][st], fx[buf][rt], (st) == 0 ? fb[ct] : acc[rt][ct], 0, 0, 0);
      QA_LOADX(0, 0)
      QA_LOADX(1, 1)   QA_WAITX(4, 0);  QA_MM(0, 0)
      QA_LOADX(0, 2)   QA_WAITX(4, 1);  QA_MM(1, 1)  QA_DMA_AT(1)
      QA_LOADX(1, 3)   QA_WAITX(4, 0);  QA_MM(0, 2)
      QA_LOADX(0, 4)   QA_WAITX(4, 1);  QA_MM(1, 3)  QA_DMA_AT(2)
      QA_LOADX(1, 5)   QA_WAITX(4, 0);  QA_MM(0, 4)
      QA_LOADX(0, 6)   QA_WAITX(4, 1);  QA_MM(1, 5)  QA_DMA_AT(3)
      QA_LOADX(1, 7)   QA_WAITX(4, 0);  QA_MM(0, 6)
      QA_LOADX(0, 8)   QA_WAITX(4, 1);  QA_MM(1, 7)
      QA_LOADX(1, 9)   QA_WAITX(4, 0);  QA_MM(0, 8)
      QA_LOADX(0, 10)  QA_WAITX(4, 1);  QA_MM(1, 9)
      QA_LOADX(1, 11)  QA_WAITX(4, 0);  QA_MM(0, 10)
      QA_WAITX(0, 1);  QA_MM(1, 11)
#undef QA_DMA_AT
#undef QA_LOADX
#undef QA_WAITX
#undef QA_MM
      // ---- rounded results -> Q / K / V tiles.  Lane holds, of token t = 64 hf + 16 rt + q, columns cb .. cb + 3 of the head block.
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int cb = 48 * wave + 16 * ct + 4 * g, part = cb >> 6, col = cb & 63;   // part is uniform per (wave, ct)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int t = 64 * hf + 16 * rt + q;
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[rt][ct][r];
          unsigned char* dst;
          if (part == 2) dst = sV + t * 128 + col * 2;
          else {
            // K: LDS row R holds key (R & ~31) + ((R&15)>>2)*8 + ((R>>4)&1)*4 + (R&3); inverse for key t
            const int R = part == 0 ? t : (t & ~31) + ((t >> 2) & 1) * 16 + ((t >> 3) & 3) * 4 + (t & 3);
            dst = (part == 0 ? sQ : sK) + R * 128 + ((((col >> 3) ^ ((R >> 1) & 7))) << 4) + ((col >> 2) & 1) * 8;
          }
          *reinterpret_cast<bf16x4*>(dst) = o;
        }
      }
    }
    __syncthreads();                                              // the crop's Q / K / V tiles are complete

    // ---- attention (attn_enc2.hip): S^T = K Q^T for this wave's 32 queries
    bf16x8 fq[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        fq[qt][ks] = *reinterpret_cast<const bf16x8*>(sQ + (wave * 32 + qt * 16 + q) * 128 + (((ks * 4 + g) ^ swz) << 4));
    f32x4 sacc[2][8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      bf16x8 fk[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fk[ks] = *reinterpret_cast<const bf16x8*>(sK + (kt * 16 + q) * 128 + (((ks * 4 + g) ^ swz) << 4));
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[0], fq[qt][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[1], fq[qt][1], a, 0, 0, 0);
      }
    }
    bf16x8 fp[2][4];
    float rinv[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ev = __expf((sacc[qt][2 * s + (e >> 2)][e & 3] - mx) * 0.125f);
          const bf16 et = (bf16)ev;
          sum += (float)et;
          o[e] = et;
        }
        fp[qt][s] = o;
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      rinv[qt] = 1.0f / sum;
    }
    f32x4 oacc[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned vbase = lds0 + (unsigned)(2 * TILE + (8 * g + (q >> 2)) * 128 + (q & 3) * 8);
#define QA_TR(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vbase), "n"(off))
#define QA_STEP(s)                                                                                          \
  {                                                                                                         \
    bf16x4 lo[4], hi[4];                                                                                    \
    QA_TR(lo[0], (s) * 4096 + 0);  QA_TR(hi[0], (s) * 4096 + 512 + 0);                                      \
    QA_TR(lo[1], (s) * 4096 + 32); QA_TR(hi[1], (s) * 4096 + 512 + 32);                                     \
    QA_TR(lo[2], (s) * 4096 + 64); QA_TR(hi[2], (s) * 4096 + 512 + 64);                                     \
    QA_TR(lo[3], (s) * 4096 + 96); QA_TR(hi[3], (s) * 4096 + 512 + 96);                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])); \
    _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                      \
      const bf16x8 fv = __builtin_shufflevector(lo[dt], hi[dt], 0, 1, 2, 3, 4, 5, 6, 7);                    \
      oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp[0][s], oacc[0][dt], 0, 0, 0);            \
      oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp[1][s], oacc[1][dt], 0, 0, 0);            \
    }                                                                                                       \
  }
    QA_STEP(0)
    QA_STEP(1)
    QA_STEP(2)
    QA_STEP(3)
#undef QA_STEP
#undef QA_TR
    // ---- out through this wave's own Q rows, then 128-byte rows
    unsigned char* const so = sQ + wave * 32 * 128;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(oacc[qt][dt][r] * rinv[qt]);
        *reinterpret_cast<bf16x4*>(so + (qt * 16 + q) * 128 + (dt * 16 + 4 * g) * 2) = o;
      }
    __builtin_amdgcn_wave_barrier();
    bf16* const op = out + ((size_t)crop * S + wave * 32) * E + h * DH;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + (lane >> 3), c = lane & 7;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(so + row * 128 + c * 16);
      *reinterpret_cast<bf16x8*>(op + (size_t)row * E + c * 8) = v;
    }
  }
}

void launch_qkv_attn(const bf16* x, const bf16* w, const float* bias, bf16* out, int N, hipStream_t s) {
  if (N <= 0) return;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out) & 15) throw std::runtime_error("qkv_attn: operands must be 16-byte aligned");
  if ((size_t)N * S * E * 2 >= ((size_t)1 << 31)) throw std::runtime_error("qkv_attn: too many crops for 32-bit buffer offsets");
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QA_LDS)); });
  hipLaunchKernelGGL(qkv_attn_kernel, dim3(8 * 32), dim3(256), QA_LDS, s, x, w, bias, out, N);
}

}  // namespace ttr
