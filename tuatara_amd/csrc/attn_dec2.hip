// PARSeq decoder cross-attention of the refinement pass (26 query rows per crop against the crop's 128 memory tokens, 12 heads of
// 32; nn.MultiheadAttention inside the TorchScript module called at tuatara.cpp:307), bf16, on the matrix cores.
//
//   out[n][q][32h + d] = sum_k softmax_k( Q[q][32h..] . K[k][32h..] / sqrt(32) ) V[k][32h + d]
//   Q = cross_q output [N*26][384], K | V = kvmem [N][128][768] (K columns 0..383, V columns 384..767)
//
// dec_cross_attn_rows_kernel (parseq_ops.hip, still used for the AR steps' single row) runs one workgroup per query row: at 26
// rows per crop that is 26 passes over the crop's 196 KB of K/V and ~165 us of vector-ALU dot products at 1220 crops (382 us).
// Here one wave owns a crop's PAIR of heads (2p, 2p+1) = 64 contiguous columns, which makes the tiles exactly those of the
// encoder's attn_enc2_kernel (128-byte rows, same swizzles, K rows permuted so that exp(S) is the P.V operand, V^T by transposed
// LDS reads), with two changes: the two 32-dim K steps are two separate heads (scores, softmax and P per head), and the query
// tile has 26 valid rows of 32 (the rest read as zeros and are not stored).  Numerics as the encoder attention: scores fp32, P
// rounded to bf16, the normaliser sums the ROUNDED P (the per-row kernel keeps P in fp32).
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
constexpr int S = 128, EQ = 384, EKV = 768;
constexpr int QT = 32 * 128, KT = S * 128;   // Q tile 4 KiB (32 rows), K / V tiles 16 KiB
}  // namespace

__global__ __launch_bounds__(64) void dec_cross_attn_mfma_kernel(const bf16* __restrict__ qin, const bf16* __restrict__ kvmem, bf16* __restrict__ out, int N, int RQ) {   // RQ <= 32 query rows per crop
  __shared__ __attribute__((aligned(1024))) unsigned char smem[QT + 2 * KT];
  unsigned char* const sQ = smem;
  unsigned char* const sK = smem + QT;
  unsigned char* const sV = smem + QT + KT;
  const int n = blockIdx.x / 6, pr = blockIdx.x - n * 6;       // crop, head pair
  const int lane = threadIdx.x;
  const int q = lane & 15, g = lane >> 4;

  // ---- one burst: piece p = LDS rows 8p .. 8p+7, this lane row 8p + (lane>>3), chunk position lane&7.
  // K: LDS row R holds key (R & ~31) + ((R&15)>>2)*8 + ((R>>4)&1)*4 + (R&3).  Q, K: position c holds chunk c ^ ((R>>1)&7).
  {
    const bf16* qb = qin + (size_t)n * RQ * EQ + pr * 64;
    const bf16* kb = kvmem + (size_t)n * S * EKV + pr * 64;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(qb), 0, (int)((RQ - 1) * EQ * 2 + 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(kb), 0, (int)((S - 1) * EKV * 2 + (EQ + 64) * 2), 0x00020000);
#pragma unroll
    for (int p = 0; p < 4; ++p) {                               // query rows RQ..31: out of range -> zeros
      const int R = p * 8 + (lane >> 3), c = lane & 7, cs = c ^ ((R >> 1) & 7);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr)(sQ + p * 1024), 16, R < RQ ? (unsigned)((R * EQ + cs * 8) * 2) : 0x80000000u, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int R = p * 8 + (lane >> 3), c = lane & 7, cs = c ^ ((R >> 1) & 7);
      const int key = (R & ~31) + ((R & 15) >> 2) * 8 + ((R >> 4) & 1) * 4 + (R & 3);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(sK + p * 1024), 16, (unsigned)((key * EKV + cs * 8) * 2), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr)(sV + p * 1024), 16, (unsigned)((R * EKV + EQ + c * 8) * 2), 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                               // (one wave: orders the LDS-DMA writes before the reads)

  const int swz = (q >> 1) & 7;                                // (row >> 1) & 7 of every fragment row (tile-aligned base + q)
  const unsigned vbase = (unsigned)(size_t)(lds_ptr)sV + (unsigned)((8 * g + (q >> 2)) * 128 + (q & 3) * 8);
#define ATT_TR(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vbase), "n"(off))
  f32x4 oacc[2][4];                                            // [query tile][16-dim tile of the pair's 64 columns]
  float rinv[2][2];                                            // [head of the pair][query tile]
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {                              // head 2 pr + hh = K step hh of the 64-column rows
    bf16x8 fq[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) fq[qt] = *reinterpret_cast<const bf16x8*>(sQ + (qt * 16 + q) * 128 + (((hh * 4 + g) ^ swz) << 4));
    // S^T = K Q^T: sacc[qt][kt], lane = query 16 qt + q, LDS key rows 16 kt + 4 g + r
    f32x4 sacc[2][8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const bf16x8 fk = *reinterpret_cast<const bf16x8*>(sK + (kt * 16 + q) * 128 + (((hh * 4 + g) ^ swz) << 4));
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[qt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // softmax over the 128 keys of a query: 32 values in this lane, the rest in lanes q + 16 g'
    bf16x8 fp[2][4];                                           // P^T fragments: [qt][32-key step]: keys 32 s + 8 g + e
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
          const float ev = __expf((sacc[qt][2 * s + (e >> 2)][e & 3] - mx) * 0.17677669529663687f);   // 1 / sqrt(32)
          const bf16 et = (bf16)ev;
          sum += (float)et;                                    // normalise by what P.V will actually sum
          o[e] = et;
        }
        fp[qt][s] = o;
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      rinv[hh][qt] = 1.0f / sum;
    }
    // O^T = V^T P^T for this head's two 16-dim tiles: A = V^T fragment (16 d x 32 keys) by two transposed reads of the row-major V tile
#pragma unroll
    for (int d2 = 0; d2 < 2; ++d2) {
      oacc[0][2 * hh + d2] = f32x4{0.f, 0.f, 0.f, 0.f};
      oacc[1][2 * hh + d2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#define ATT_STEP(s)                                                                                                   \
    {                                                                                                                 \
      bf16x4 lo[2], hi[2];                                                                                            \
      if (hh == 0) { ATT_TR(lo[0], (s) * 4096 + 0);  ATT_TR(hi[0], (s) * 4096 + 512 + 0);                             \
                     ATT_TR(lo[1], (s) * 4096 + 32); ATT_TR(hi[1], (s) * 4096 + 512 + 32); }                          \
      else         { ATT_TR(lo[0], (s) * 4096 + 64); ATT_TR(hi[0], (s) * 4096 + 512 + 64);                            \
                     ATT_TR(lo[1], (s) * 4096 + 96); ATT_TR(hi[1], (s) * 4096 + 512 + 96); }                          \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));                      \
      _Pragma("unroll") for (int d2 = 0; d2 < 2; ++d2) {                                                              \
        const bf16x8 fv = __builtin_shufflevector(lo[d2], hi[d2], 0, 1, 2, 3, 4, 5, 6, 7);                            \
        oacc[0][2 * hh + d2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp[0][s], oacc[0][2 * hh + d2], 0, 0, 0);  \
        oacc[1][2 * hh + d2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp[1][s], oacc[1][2 * hh + d2], 0, 0, 0);  \
      }                                                                                                               \
    }
    ATT_STEP(0)
    ATT_STEP(1)
    ATT_STEP(2)
    ATT_STEP(3)
#undef ATT_STEP
  }
#undef ATT_TR

  // ---- out: lane holds d = 16 dt + 4 g + r of query 16 qt + q; staged through the Q rows in LDS (its fragments are consumed),
  // then whole 128-byte rows (the pair's 64 columns), 16 bytes per lane, rows < RQ only
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16)(oacc[qt][dt][r] * rinv[dt >> 1][qt]);
      *reinterpret_cast<bf16x4*>(sQ + (qt * 16 + q) * 128 + (dt * 16 + 4 * g) * 2) = o;
    }
  __builtin_amdgcn_wave_barrier();
  bf16* const op = out + (size_t)n * RQ * EQ + pr * 64;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 8 + (lane >> 3), c = lane & 7;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(sQ + row * 128 + c * 16);
    if (row < RQ) *reinterpret_cast<bf16x8*>(op + (size_t)row * EQ + c * 8) = v;
  }
}

void launch_dec_cross_attn_mfma(const bf16* q, const bf16* kvmem, bf16* out, int N, int R, hipStream_t s) {
  if (N <= 0) return;
  if (R < 1 || R > 32) throw std::runtime_error("dec_cross_attn_mfma: 1..32 query rows per crop");
  if (((uintptr_t)q | (uintptr_t)kvmem | (uintptr_t)out) & 15) throw std::runtime_error("dec_cross_attn_mfma: operands must be 16-byte aligned");
  hipLaunchKernelGGL(dec_cross_attn_mfma_kernel, dim3(N * 6), dim3(64), 0, s, q, kvmem, out, N, R);
}

}  // namespace ttr
