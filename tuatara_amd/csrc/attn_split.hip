// PARSeq ViT encoder self-attention in split-operand arithmetic (split.h): one (crop, head) per workgroup, fp32-equivalent
// products on the f16 matrix cores.  Counterpart of attn_enc2.hip (same tiling, same transposed forms) for the engine's
// default precision; timm Attention.forward inside the TorchScript module called at tuatara.cpp:307.
//
//   out[n][q][64h + d] = sum_k softmax_k( Q[q] . K[k] / 8 ) V[k][d]
//
// qkv arrives as planes [N*128][3][1152] (x0 | x1 | x2 of every value, written by the qkv GEMM's epilogue); the result leaves as
// planes [N*128][3][384] for the projection GEMM.  Which operand is the exact triple and which the pair (split.h):
//   * S^T = K Q^T: Q is the triple (q0, q1, q2) - a wave's 32 query rows go global -> registers, they are read once -, K the
//     pair: its planes x0, x1 come global -> LDS by LDS-DMA and the pair's members k0b = k0 / 2^11 and k1 = x1 / 2^11 are formed from
//     the fragments by packed f16 multiplies (exact);
//   * O^T = V^T P^T: P = exp(s - max) is split in registers into (p0, p1, p2) and IS the B fragment (K rows permuted as in
//     attn_enc2.hip); V is the pair, V^T fragments by `ds_read_b64_tr_b16` from the row-major planes.
// Four MFMAs per product pair: (w0, x0), (w0b, x1), (w0b, x2), (w1, x0).  Absolute error of a score: <= |q| 64 2^-25 from the
// pair's subnormal range - below fp32's own rounding of the sum.  64 KB of LDS: two workgroups per CU, one's loads under the
// other's MFMAs (the kernel is HBM-bound: 7 of the 9 qkv planes = 96 KB per (crop, head)).
#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
constexpr int S = 128, DH = 64, E3 = 1152, EO = 384;
constexpr int RS = 3 * E3;           // halves per qkv row (three planes)
constexpr int TILE = S * DH * 2;     // 16 KiB per tile

__device__ __forceinline__ f16x8 scale_down(f16x8 v) {   // v / 2^11 (exact unless subnormal)
  const f16 s = (f16)(1.f / 2048.f);
  return v * f16x8{s, s, s, s, s, s, s, s};
}
}  // namespace

__global__ __launch_bounds__(256, 2) void attn_enc_split_kernel(const f16* __restrict__ qkv, f16* __restrict__ out, int N, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE];
  unsigned char* const sK = smem;               // [2 planes][128 keys (permuted)][128 B]
  unsigned char* const sV = smem + 2 * TILE;    // [2 planes][128 keys][128 B]
  const int n = blockIdx.x / 6, h = blockIdx.x - n * 6;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  const f16* base = qkv + (size_t)n * S * RS + h * DH;

  // ---- K, V planes 0 and 1: one LDS-DMA burst (piece p = LDS rows 8p .. 8p+7, this lane row 8p + (lane>>3), chunk position lane&7)
  // K: LDS row R holds key (R & ~31) + ((R&15)>>2)*8 + ((R>>4)&1)*4 + (R&3), position c holds chunk c ^ ((R>>1)&7).
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)(((S - 1) * RS + 2 * E3) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = wave + 4 * j, R = p * 8 + (lane >> 3), c = lane & 7;
      const int cs = c ^ ((R >> 1) & 7);
      const int key = (R & ~31) + ((R & 15) >> 2) * 8 + ((R >> 4) & 1) * 4 + (R & 3);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sK + pl * TILE + p * 1024), 16, (unsigned)((key * RS + pl * E3 + EO + cs * 8) * 2), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(sV + pl * TILE + p * 1024), 16, (unsigned)((R * RS + pl * E3 + 2 * EO + c * 8) * 2), 0, 0, 0);
      }
    }
  }
  // ---- Q: the wave's 32 rows, three planes, straight into fragments (B operand: n = query q, k = 8 g + e)
  f16x8 fq[3][2][2];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        fq[pl][qt][ks] = *reinterpret_cast<const f16x8*>(base + (size_t)(wave * 32 + qt * 16 + q) * RS + pl * E3 + (ks * 4 + g) * 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- S^T = K Q^T for this wave's 32 queries: sacc[qt][kt], lane = query 16 qt + q, LDS key rows 16 kt + 4 g + r
  const int swz = (q >> 1) & 7;
  f32x4 sacc[2][8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    f32x4 a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = (kt * 16 + q) * 128 + (((ks * 4 + g) ^ swz) << 4);
      const f16x8 k0 = *reinterpret_cast<const f16x8*>(sK + off), x1 = *reinterpret_cast<const f16x8*>(sK + TILE + off);
      const f16x8 k0b = scale_down(k0), k1 = scale_down(x1);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, fq[0][qt][ks], a[qt], 0, 0, 0);
        a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[1][qt][ks], a[qt], 0, 0, 0);
        a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[2][qt][ks], a[qt], 0, 0, 0);
        a[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, fq[0][qt][ks], a[qt], 0, 0, 0);
      }
    }
    sacc[0][kt] = a[0]; sacc[1][kt] = a[1];
  }

  // ---- softmax over the 128 keys of a query (32 values in this lane, the rest in lanes q + 16 g'); P split into its planes
  f16x8 fp[3][2][4];                                         // [plane][qt][32-key step]: keys 32 s + 8 g + e
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
      float ev[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ev[e] = __expf((sacc[qt][2 * s + (e >> 2)][e & 3] - mx) * 0.125f);
        sum += ev[e];
      }
      split3_x8(ev, fp[0][qt][s], fp[1][qt][s], fp[2][qt][s], rw);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    rinv[qt] = 1.0f / sum;
  }

  // ---- O^T = V^T P^T: A = V^T fragment (16 d x 32 keys) by two transposed reads per plane of the row-major V tile
  f32x4 oacc[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned vbase = (unsigned)(size_t)(lds_ptr)sV + (unsigned)((8 * g + (q >> 2)) * 128 + (q & 3) * 8);
#define ATS_TR(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vbase), "n"(off))
#define ATS_STEP(s)                                                                                         \
  {                                                                                                         \
    f16x4 lo[2][4], hi[2][4];                                                                               \
    ATS_TR(lo[0][0], (s) * 4096 + 0);  ATS_TR(hi[0][0], (s) * 4096 + 512 + 0);                              \
    ATS_TR(lo[0][1], (s) * 4096 + 32); ATS_TR(hi[0][1], (s) * 4096 + 512 + 32);                             \
    ATS_TR(lo[0][2], (s) * 4096 + 64); ATS_TR(hi[0][2], (s) * 4096 + 512 + 64);                             \
    ATS_TR(lo[0][3], (s) * 4096 + 96); ATS_TR(hi[0][3], (s) * 4096 + 512 + 96);                             \
    ATS_TR(lo[1][0], TILE + (s) * 4096 + 0);  ATS_TR(hi[1][0], TILE + (s) * 4096 + 512 + 0);                \
    ATS_TR(lo[1][1], TILE + (s) * 4096 + 32); ATS_TR(hi[1][1], TILE + (s) * 4096 + 512 + 32);               \
    ATS_TR(lo[1][2], TILE + (s) * 4096 + 64); ATS_TR(hi[1][2], TILE + (s) * 4096 + 512 + 64);               \
    ATS_TR(lo[1][3], TILE + (s) * 4096 + 96); ATS_TR(hi[1][3], TILE + (s) * 4096 + 512 + 96);               \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[0][2]), "+v"(lo[0][3]), "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[0][2]), "+v"(hi[0][3]), \
                 "+v"(lo[1][0]), "+v"(lo[1][1]), "+v"(lo[1][2]), "+v"(lo[1][3]), "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(hi[1][2]), "+v"(hi[1][3])); \
    _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                      \
      const f16x8 v0 = __builtin_shufflevector(lo[0][dt], hi[0][dt], 0, 1, 2, 3, 4, 5, 6, 7);               \
      const f16x8 x1 = __builtin_shufflevector(lo[1][dt], hi[1][dt], 0, 1, 2, 3, 4, 5, 6, 7);               \
      const f16x8 v0b = scale_down(v0), v1 = scale_down(x1);                                                \
      _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                    \
        f32x4 a = oacc[qt][dt];                                                                             \
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, fp[0][qt][s], a, 0, 0, 0);                           \
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0b, fp[1][qt][s], a, 0, 0, 0);                          \
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0b, fp[2][qt][s], a, 0, 0, 0);                          \
        oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, fp[0][qt][s], a, 0, 0, 0);                \
      }                                                                                                     \
    }                                                                                                       \
  }
  ATS_STEP(0)
  ATS_STEP(1)
  ATS_STEP(2)
  ATS_STEP(3)
#undef ATS_STEP
#undef ATS_TR

  // ---- out: lane holds d = 16 dt + 4 g + r of query 16 qt + q.  The three planes of the wave's [32 queries x 64 d] result are
  // staged through LDS (K / V are dead once every wave is here), then leave as whole 128-byte rows, 16 bytes per lane
  __syncthreads();
  unsigned char* const so = smem + wave * (3 * 32 * 128);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f16x2 a0, b0, c0, a1, b1, c1;
      split3_pair(oacc[qt][dt][0] * rinv[qt], oacc[qt][dt][1] * rinv[qt], a0, b0, c0, rw);
      split3_pair(oacc[qt][dt][2] * rinv[qt], oacc[qt][dt][3] * rinv[qt], a1, b1, c1, rw);
      unsigned char* d = so + (qt * 16 + q) * 128 + (dt * 16 + 4 * g) * 2;
      *reinterpret_cast<f16x4*>(d) = f16x4{a0[0], a0[1], a1[0], a1[1]};
      *reinterpret_cast<f16x4*>(d + 32 * 128) = f16x4{b0[0], b0[1], b1[0], b1[1]};
      *reinterpret_cast<f16x4*>(d + 64 * 128) = f16x4{c0[0], c0[1], c1[0], c1[1]};
    }
  __builtin_amdgcn_wave_barrier();
  f16* const op = out + ((size_t)n * S + wave * 32) * (3 * EO) + h * DH;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + (lane >> 3), c = lane & 7;
      const f16x8 v = *reinterpret_cast<const f16x8*>(so + pl * 32 * 128 + row * 128 + c * 16);
      *reinterpret_cast<f16x8*>(op + (size_t)row * (3 * EO) + pl * EO + c * 8) = v;
    }
  rw.flush(range_flag, range_tag);
}

void launch_attn_enc_split(const void* qkv_planes, void* out_planes, int N, hipStream_t s) {
  if (N <= 0) return;
  if (((uintptr_t)qkv_planes | (uintptr_t)out_planes) & 15) throw std::runtime_error("attn_enc_split: operands must be 16-byte aligned");
  hipLaunchKernelGGL(attn_enc_split_kernel, dim3(N * 6), dim3(256), 0, s, (const f16*)qkv_planes, (f16*)out_planes, N, range_ctx().flag, range_ctx().tag);
}

}  // namespace ttr
