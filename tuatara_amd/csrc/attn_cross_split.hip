// PARSeq decoder cross-attention of the REFINEMENT pass (26 query rows per crop against the crop's 128 memory tokens, 12 heads of 32) on the matrix cores, in
// split-operand arithmetic (split.h): one WAVE per (crop, head).  Counterpart of attn_split.hip (the encoder's stand-alone attention: same fragment forms, same
// transposed reads, DH = 32 instead of 64, 32 query rows - 26 real - instead of 128); nn.MultiheadAttention inside the decoder layer of the TorchScript module
// the reference runs at tuatara.cpp:307.
//
//   out[n][r][32 h + d] = sum_j softmax_j( Q[r] . K[j] / sqrt(32) ) V[j][d]
//
// q: fp32 [N * 26][384] (the cross_q linear), kvmem: fp32 [N * 128][768] (K | V of the crop's memory, the cross_kv linear), out: exact triples [N * 26][3][384]
// for the cross_out linear.  dec_cross_attn_crop_kernel (parseq_ops.hip) does the same on the vector ALU - 26 x 128 x 32 multiply-adds per head, two LDS reads
// each - and is bound by the LDS instruction rate (550 us at 1280 crops whatever its occupancy and head split); here
//   * K_h and V_h (128 x 32 fp32 each) are read once with 16-byte loads, written to the wave's own LDS as f16 PAIRS (x0 | x1 2^11): K with its rows permuted so
//     that S^T = K Q^T leaves 8 consecutive keys per lane (the P fragment of O^T = V^T P^T), 16-byte chunks swizzled against the 64-byte row stride;
//   * Q is the exact triple in registers (a score goes through an exponential), P a pair (p in [0, 1] on 22+ bits is fp32's resolution of it): S^T takes four
//     MFMAs per product pair, O^T three, 112 MFMAs per head;
//   * V^T fragments come by `ds_read_b64_tr_b16` from the row-major V planes; the 26 result rows leave as 8-byte pieces of the three planes.
// Two waves (two heads) per workgroup, 16 KB of LDS per wave (V takes K's place once S^T is formed; ten waves per CU hide each other's loads): no workgroup
// barrier anywhere - a wave reads only what it wrote.
#include <stdexcept>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
constexpr int XS_KEYS = 128, XS_DH = 32, XS_ROWB = 64;          // bytes per LDS row (32 halves)
constexpr int XS_PLANE = XS_KEYS * XS_ROWB;                       // 8 KiB per plane
constexpr int XS_WAVE_LDS = 2 * XS_PLANE;                         // x0 | x1 of K, then of V

__device__ __forceinline__ f16x8 xs_scale_down(f16x8 v) {         // v / 2^11 (exact unless subnormal)
  const f16 s = (f16)(1.f / 2048.f);
  return v * f16x8{s, s, s, s, s, s, s, s};
}
}  // namespace

__global__ __launch_bounds__(128) void dec_cross_attn_split_kernel(const float* __restrict__ q, const float* __restrict__ kvmem, f16* __restrict__ out, int N, int R,
                                                                   unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * XS_WAVE_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x, h = blockIdx.y * 2 + wave;
  const int qq = lane & 15, g = lane >> 4;
  unsigned char* const sK = smem + wave * XS_WAVE_LDS;            // [2 planes][128 rows (keys permuted)][64 B]
  unsigned char* const sV = sK;                                   // [2 planes][128 keys][64 B], once S^T has read K

  // ---- K_h: fp32 -> pairs -> LDS.  A wave instruction fetches 8 key rows x 128 bytes; this lane: key 8 it + (lane >> 3), d = 4 (lane & 7) .. + 3
  const float* const kv = kvmem + (int64_t)n * XS_KEYS * 768 + h * XS_DH + (lane & 7) * 4;
  const int c4 = lane & 7;
  // (K, V and Q arrive as fp32 - the cross_kv / cross_q linears' rows - and become f16 planes HERE: they are watched with the output)
#pragma unroll 8
  for (int it = 0; it < 16; ++it) {
    const int key = it * 8 + (lane >> 3);
    const float4 kf = *reinterpret_cast<const float4*>(kv + (int64_t)key * 768);
    f16x2 a0, b0, a1, b1;
    split2_pair(kf.x, kf.y, a0, b0, rw); split2_pair(kf.z, kf.w, a1, b1, rw);
    // LDS row of key k = 32 s + 8 g' + 4 a + b:  R = 32 s + 16 a + 4 g' + b   (so that S^T's accumulators hold 8 consecutive keys per lane)
    const int Rk = (key & ~31) + ((key >> 2) & 1) * 16 + ((key >> 3) & 3) * 4 + (key & 3);
    const int posk = (((c4 >> 1) ^ ((Rk >> 2) & 3)) << 4) + (c4 & 1) * 8;
    *reinterpret_cast<f16x4*>(sK + Rk * XS_ROWB + posk) = f16x4{a0[0], a0[1], a1[0], a1[1]};
    *reinterpret_cast<f16x4*>(sK + XS_PLANE + Rk * XS_ROWB + posk) = f16x4{b0[0], b0[1], b1[0], b1[1]};
  }
  // ---- Q: rows 16 qt + qq (zero behind the 26th), d = 8 g .. 8 g + 7: the B operand of S^T as an exact triple
  f16x8 fq[3][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = qt * 16 + qq;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      const float* qp = q + ((int64_t)n * R + r) * 384 + h * XS_DH + g * 8;
      const float4 a = *reinterpret_cast<const float4*>(qp), b = *reinterpret_cast<const float4*>(qp + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    split3_x8(v, fq[0][qt], fq[1][qt], fq[2][qt], rw);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's LDS writes are done (a wave's LDS operations complete in order; the compiler must not move reads above)
  __builtin_amdgcn_wave_barrier();

  // ---- S^T = K Q^T: sacc[qt][kt], lane = query 16 qt + qq, LDS key rows 16 kt + 4 g + r
  const int swz = (qq >> 2) & 3;
  f32x4 sacc[2][8];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    const int off = (kt * 16 + qq) * XS_ROWB + ((g ^ swz) << 4);
    const f16x8 k0 = *reinterpret_cast<const f16x8*>(sK + off), x1 = *reinterpret_cast<const f16x8*>(sK + XS_PLANE + off);
    const f16x8 k0b = xs_scale_down(k0), k1 = xs_scale_down(x1);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, fq[0][qt], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[1][qt], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[2][qt], a, 0, 0, 0);
      sacc[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, fq[0][qt], a, 0, 0, 0);
    }
  }

  // ---- V_h over K's place (the wave's LDS reads above are complete: their values went into the MFMAs the compiler waits for... made explicit here)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#pragma unroll 8
  for (int it = 0; it < 16; ++it) {
    const int key = it * 8 + (lane >> 3);
    const float4 vf = *reinterpret_cast<const float4*>(kv + (int64_t)key * 768 + 384);
    f16x2 a0, b0, a1, b1;
    split2_pair(vf.x, vf.y, a0, b0, rw); split2_pair(vf.z, vf.w, a1, b1, rw);
    *reinterpret_cast<f16x4*>(sV + key * XS_ROWB + c4 * 8) = f16x4{a0[0], a0[1], a1[0], a1[1]};
    *reinterpret_cast<f16x4*>(sV + XS_PLANE + key * XS_ROWB + c4 * 8) = f16x4{b0[0], b0[1], b1[0], b1[1]};
  }

  // ---- softmax over the 128 keys of a query (32 values in this lane, the rest in lanes qq + 16 g'); P as pairs
  f16x8 fp[2][2][4];                                              // [plane][qt][32-key step]: keys 32 s + 8 g + e
  float rinv[2];
  constexpr float kScale = 0.17677669529663687f;                  // 1 / sqrt(32)
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
    RangeWatch rp;                                                // (dead: the probabilities are <= 1)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float ev[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ev[e] = __expf((sacc[qt][2 * s + (e >> 2)][e & 3] - mx) * kScale);
        sum += ev[e];
      }
      split2_x8(ev, fp[0][qt][s], fp[1][qt][s], rp);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    rinv[qt] = 1.0f / sum;
  }

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // V is in LDS
  __builtin_amdgcn_wave_barrier();
  // ---- O^T = V^T P^T: A = V^T fragment (16 d x 32 keys) by two transposed reads per plane of the row-major V tile (attn_split.hip, 64-byte rows here)
  f32x4 oacc[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned vbase = (unsigned)(size_t)(lds_ptr)sV + (unsigned)((8 * g + (qq >> 2)) * XS_ROWB + (qq & 3) * 8);
#define XS_TR(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vbase), "n"(off))
#define XS_STEP(s)                                                                                           \
  {                                                                                                          \
    f16x4 lo[2][2], hi[2][2];                                                                                \
    XS_TR(lo[0][0], (s) * 2048 + 0);  XS_TR(hi[0][0], (s) * 2048 + 256 + 0);                                 \
    XS_TR(lo[0][1], (s) * 2048 + 32); XS_TR(hi[0][1], (s) * 2048 + 256 + 32);                                \
    XS_TR(lo[1][0], XS_PLANE + (s) * 2048 + 0);  XS_TR(hi[1][0], XS_PLANE + (s) * 2048 + 256 + 0);           \
    XS_TR(lo[1][1], XS_PLANE + (s) * 2048 + 32); XS_TR(hi[1][1], XS_PLANE + (s) * 2048 + 256 + 32);          \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(hi[0][0]), "+v"(hi[0][1]),    \
                 "+v"(lo[1][0]), "+v"(lo[1][1]), "+v"(hi[1][0]), "+v"(hi[1][1]));                            \
    _Pragma("unroll") for (int dt = 0; dt < 2; ++dt) {                                                       \
      const f16x8 v0 = __builtin_shufflevector(lo[0][dt], hi[0][dt], 0, 1, 2, 3, 4, 5, 6, 7);                \
      const f16x8 x1 = __builtin_shufflevector(lo[1][dt], hi[1][dt], 0, 1, 2, 3, 4, 5, 6, 7);                \
      const f16x8 v0b = xs_scale_down(v0), v1 = xs_scale_down(x1);                                           \
      _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                                     \
        f32x4 a = oacc[qt][dt];                                                                              \
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, fp[0][qt][s], a, 0, 0, 0);                            \
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0b, fp[1][qt][s], a, 0, 0, 0);                           \
        oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, fp[0][qt][s], a, 0, 0, 0);                 \
      }                                                                                                      \
    }                                                                                                        \
  }
  XS_STEP(0)
  XS_STEP(1)
  XS_STEP(2)
  XS_STEP(3)
#undef XS_STEP
#undef XS_TR

  // ---- out: lane holds d = 16 dt + 4 g + r of query 16 qt + qq: 8 bytes of each plane of a row
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int r = qt * 16 + qq;
    if (r >= R) continue;
    f16* const op = out + ((int64_t)n * R + r) * (3 * 384) + h * XS_DH + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f16x2 a0, b0, c0, a1, b1, c1;
      split3_pair(oacc[qt][dt][0] * rinv[qt], oacc[qt][dt][1] * rinv[qt], a0, b0, c0, rw);
      split3_pair(oacc[qt][dt][2] * rinv[qt], oacc[qt][dt][3] * rinv[qt], a1, b1, c1, rw);
      *reinterpret_cast<f16x4*>(op + dt * 16) = f16x4{a0[0], a0[1], a1[0], a1[1]};
      *reinterpret_cast<f16x4*>(op + 384 + dt * 16) = f16x4{b0[0], b0[1], b1[0], b1[1]};
      *reinterpret_cast<f16x4*>(op + 768 + dt * 16) = f16x4{c0[0], c0[1], c1[0], c1[1]};
    }
  }
  rw.flush(range_flag, range_tag);
}

// q fp32 [N * R][384], kvmem fp32 [N * 128][768] -> out triples [N * R][3][384]; R <= 32 query rows per crop (the refinement pass: 26)
void launch_dec_cross_attn_split(const float* q, const float* kvmem, void* out_planes, int N, int R, hipStream_t s) {
  if (N <= 0) return;
  if (R < 1 || R > 32) throw std::runtime_error("dec_cross_attn_split: 1 .. 32 query rows per crop");
  if (((uintptr_t)q | (uintptr_t)kvmem | (uintptr_t)out_planes) & 15) throw std::runtime_error("dec_cross_attn_split: operands must be 16-byte aligned");
  hipLaunchKernelGGL(dec_cross_attn_split_kernel, dim3(N, 6), dim3(128), 0, s, q, kvmem, (f16*)out_planes, N, R, range_ctx().flag, range_ctx().tag);
}

}  // namespace ttr
