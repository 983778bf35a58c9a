// Skinny bf16 GEMM for the PARSeq decoder's per-step linears (M = crops in flight, a few hundred rows):
//   out[m][n] = act( sum_k X[m][k] Wt[n][k] + bias[n] (+ resid) ),  same ConvParams contract as gemm2.hip (ks = 1).
//
// These problems are latency-bound, not throughput-bound: gemm2's 128x128 tiles give 15-60 workgroups and walk K in
// 6-24 dependent, barely pipelined steps.  Here a workgroup owns a 32x32 output tile (hundreds of workgroups), pulls
// its whole [32 x KC] X and W panels into LDS in ONE burst of LDS-DMA loads (KC = K up to 768; K = 1536 takes two
// bursts), waits once and runs the MFMAs: one memory round trip per launch.  LDS rows are K-major; the 16-byte
// chunks of a row are XOR-swizzled on the source side (chunk ^= row & 15) so ds_read_b128 is conflict-free.
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int SK_BM = 32, SK_BN = 32, SK_KC = 768;
}  // namespace

__global__ __launch_bounds__(256) void gemm_sk_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  if (p.skip && __builtin_nontemporal_load(p.skip) >= p.skip_n) return;   // every crop has emitted EOS: nothing left to decode
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int K = p.C0;
  const int tilesN = (p.Cout + SK_BN - 1) / SK_BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
  const int m0 = tm * SK_BM, n0 = tn * SK_BN;
  const int KC = K < SK_KC ? K : SK_KC;            // K is a multiple of 128; KC of 128
  const int CPR = KC >> 3;                         // 16-byte chunks per panel row
  const int npieces = (SK_BM * CPR) >> 6;          // 1-KiB pieces per panel
  unsigned char* const xs = smem;                  // [32][KC] bf16
  unsigned char* const ws = smem + SK_BM * KC * 2; // [32][KC] bf16
  const __amdgpu_buffer_rsrc_t rsx = sk_rsrc(p.ln_in ? p.wgt : p.in0, p.ln_in ? 16u : (unsigned)((size_t)p.M * K * 2));
  const __amdgpu_buffer_rsrc_t rsw = sk_rsrc(p.wgt, (unsigned)((size_t)p.Cout * K * 2));
  constexpr unsigned OOB = 0x80000000u;

  const int mi = wave >> 1, nj = wave & 1;         // this wave's 16x16 output tile
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    if (kc0) __syncthreads();                      // everyone is done reading the previous panels
    for (int pc = wave; pc < npieces; pc += 4) {
      const int q = pc * 64 + lane;
      const int row = q / CPR, c = q - row * CPR;
      const int g = (c & ~15) | ((c & 15) ^ (row & 15));
      const int m = m0 + row, n = n0 + row;
      const unsigned xo = m < p.M ? (unsigned)((m * K + kc0 + g * 8) * 2) : OOB;
      const unsigned wo = n < p.Cout ? (unsigned)((n * K + kc0 + g * 8) * 2) : OOB;
      if (!p.ln_in) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(xs + pc * 1024), 16, xo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(ws + pc * 1024), 16, wo, 0, 0, 0);
    }
    if (p.ln_in) {
      // fused LayerNorm prologue (K == 384 == one panel): wave w normalises tile rows 8w..8w+7 in one pass, 8 lanes per row,
      // lane j of a row owning the 16-byte chunks j, j+8, .. j+40; each chunk lands at the swizzled position the fragment
      // reads expect
      const int row = wave * 8 + (lane >> 3), m = m0 + row, j = lane & 7;
      const bool live = m < p.M;
      const float* lrow = p.ln_in + (int64_t)m * p.ln_ld;
      if (p.tok) {   // PARSeq AR step: the row is the embedding of this crop's previous token (see ConvParams::tok)
        int tokv = 0;
        if (p.tok_logits) {
          float best = -INFINITY; int bi = 0x7fffffff;
          if (live) {
            const float* lg = p.tok_logits + (int64_t)m * p.tok_logits_ld;
            float x[16];                              // all loads in flight before the first compare (tok_C <= 128)
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = j + 8 * u < p.tok_C ? lg[j + 8 * u] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 16; ++u) if (x[u] > best) { best = x[u]; bi = j + 8 * u; }
          }
#pragma unroll
          for (int o = 4; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
          }
          if (live && tn == 0 && j == 0) {
            int* trow = p.tok + (int64_t)m * p.tok_ld;
            trow[p.tok_col] = bi;
            if (p.done_count && bi == p.tok_eos) {       // this crop's first EOS?  (column 0 is BOS)
              bool first = true;
              for (int c = 1; c < p.tok_col; ++c) first = first && trow[c] != p.tok_eos;
              if (first) atomicAdd(p.done_count, 1);
            }
          }
          tokv = bi;
        } else if (live) {
          tokv = p.tok[(int64_t)m * p.tok_ld + p.tok_col];
        }
        tokv = tokv < 0 ? 0 : (tokv > p.tok_max ? p.tok_max : tokv);
        lrow = p.tok_emb + (int64_t)tokv * 384;
      }
      float v[6][8];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (live) {
          const float* src = lrow + (c * 8 + j) * 8;
          a = *reinterpret_cast<const float4*>(src); b = *reinterpret_cast<const float4*>(src + 4);
          if (p.tok && p.tok_pos) {
            const float4 pa = *reinterpret_cast<const float4*>(p.tok_pos + (c * 8 + j) * 8), pb = *reinterpret_cast<const float4*>(p.tok_pos + (c * 8 + j) * 8 + 4);
            a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w; b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
          }
        }
        v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
      }
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[c][e];
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float mean = sum * (1.f / 384.f);
      float q2 = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q2 += d * d; }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
      const float rstd = rsqrtf(q2 * (1.f / 384.f) + p.ln_eps);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int ch = c * 8 + j;
        bf16x8 ob;
        if (live) {
          const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(p.ln_gamma + ch * 8 + 4);
          const float4 t0 = *reinterpret_cast<const float4*>(p.ln_beta + ch * 8), t1 = *reinterpret_cast<const float4*>(p.ln_beta + ch * 8 + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) ob[e] = (bf16)((v[c][e] - mean) * rstd * gg[e] + bb[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) ob[e] = (bf16)0.f;
        }
        const int cidx = (ch & ~15) | ((ch & 15) ^ (row & 15));
        *reinterpret_cast<bf16x8*>(xs + row * KC * 2 + cidx * 16) = ob;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* xr = xs + (mi * 16 + fr) * KC * 2;
    const unsigned char* wr = ws + (nj * 16 + fr) * KC * 2;
    const int sw = fr;                             // (row & 15) of both fragment rows
    for (int u = 0; u < (KC >> 5); ++u) {
      const int G = u * 4 + fg;
      const int c = (G & ~15) | ((G & 15) ^ sw);
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(wr + c * 16);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + c * 16);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
  }

  // lane holds out[m = m0 + 16mi + fr][n = n0 + 16nj + 4fg .. +3]
  const int m = m0 + mi * 16 + fr, n = n0 + nj * 16 + fg * 4;
  if (m >= p.M || n >= p.Cout) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const bool vec = ((p.Cout | p.out_ld | p.out_f32_ld | p.resid_ld) & 3) == 0;   // uniform; unused strides are 0
  const int64_t rrow = p.resid ? (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld : 0;
  if (vec) {
    if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    if (p.resid) { const float4 r = *reinterpret_cast<const float4*>(p.resid + rrow + n); v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.Cout) { if (p.bias) v[e] += p.bias[n + e]; if (p.resid) v[e] += p.resid[rrow + n + e]; }
  }
  if (p.act == kActRelu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (p.act == kActGelu) {
    const float2* lut = reinterpret_cast<const float2*>(p.gelu_lut);   // straight from L2: 4 values per lane
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_lut(v[e], lut);
  }
  if (vec) {
    if (p.out) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.out) + (int64_t)m * p.out_ld + n) = o;
    }
    if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + (int64_t)m * p.out_f32_ld + n) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < p.Cout) {
        if (p.out) reinterpret_cast<bf16*>(p.out)[(int64_t)m * p.out_ld + n + e] = (bf16)v[e];
        if (p.out_f32) p.out_f32[(int64_t)m * p.out_f32_ld + n + e] = v[e];
      }
  }
}

const char* gemm_sk_check(const ConvParams& p) {
  if (p.ks != 1 || p.C1 || p.relu0 || p.relu1 || p.out_relu || p.out_pool) return "gemm_sk: plain linear layers only";
  if (p.C0 % 128 || p.C0 < 128 || (p.C0 > SK_KC && p.C0 % SK_KC)) return "gemm_sk: K must be a multiple of 128 (and of 768 above 768)";
  const bool vec = ((p.Cout | p.out_ld | p.out_f32_ld | p.resid_ld) & 3) == 0;   // the kernel's vector epilogue (else scalar)
  if (vec && p.out && ((uintptr_t)p.out & 7)) return "gemm_sk: bf16 output alignment";
  if (vec && p.out_f32 && ((uintptr_t)p.out_f32 & 15)) return "gemm_sk: f32 output alignment";
  if (vec && p.resid && ((uintptr_t)p.resid & 15)) return "gemm_sk: residual alignment";
  if (p.bias && ((uintptr_t)p.bias & 15)) return "gemm_sk: bias alignment";
  if ((!p.ln_in && ((uintptr_t)p.in0 & 15)) || ((uintptr_t)p.wgt & 15)) return "gemm_sk: operand alignment";
  if (p.ln_in && (p.C0 != 384 || p.ln_ld % 4 || ((uintptr_t)p.ln_in & 15) || !p.ln_gamma || !p.ln_beta || ((uintptr_t)p.ln_gamma & 15) || ((uintptr_t)p.ln_beta & 15)))
    return "gemm_sk: fused LayerNorm needs K == 384 and 16-byte aligned f32 rows / parameters";
  if (p.tok && (!p.ln_in || !p.tok_emb || ((uintptr_t)p.tok_emb & 15) || ((uintptr_t)p.tok_pos & 15) || (p.tok_logits && (p.tok_C <= 0 || p.tok_C > 128)) || p.tok_max < 0))
    return "gemm_sk: token prologue needs the LayerNorm prologue, a 16-byte aligned embedding table and position row";
  const size_t lim = (size_t)1 << 31;
  if ((!p.ln_in && (size_t)p.M * p.C0 * 2 >= lim) || (size_t)p.Cout * p.C0 * 2 >= lim) return "gemm_sk: tensor too large";
  if (p.M <= 0 || p.Cout <= 0) return "gemm_sk: bad shape";
  return nullptr;
}

void launch_gemm_sk(const ConvParams& p_in, hipStream_t s) {
  if (const char* e = gemm_sk_check(p_in)) throw std::runtime_error(e);
  ConvParams p = p_in;
  p.gelu_lut = p.act == kActGelu ? gelu_lut_for_current_device() : nullptr;
  const int KC = p.C0 < SK_KC ? p.C0 : SK_KC;
  const size_t lds = (size_t)(SK_BM + SK_BN) * KC * 2;
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_sk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (SK_BM + SK_BN) * SK_KC * 2)); });
  const int tilesM = (p.M + SK_BM - 1) / SK_BM, tilesN = (p.Cout + SK_BN - 1) / SK_BN;
  hipLaunchKernelGGL(gemm_sk_kernel, dim3(tilesM * tilesN), dim3(256), lds, s, p);
}

}  // namespace ttr
