// PARSeq autoregressive decoder as ONE persistent kernel (bf16 mode).
//
// The reference runs the 26-step AR loop inside the TorchScript module it calls at
// tuatara.cpp:307 (upstream PARSeq.forward, decode_ar=True).  Step i needs the argmax of
// step i-1, so a kernel-per-op schedule is ~14 tiny launches x 25 steps, each far too small
// to fill the chip.  Crops are independent, so here one workgroup owns G crops for the
// whole loop: every intermediate (content row, query stream, FFN hidden, logits, tokens)
// lives in LDS, each linear is an MFMA product out^T[N][16] = W[N][K] . X^T[K][16] whose
// weight fragments stream straight from L2 into registers (36 KiB in flight per wave) and
// whose activation operand is read from LDS; LayerNorm, the two attentions (48 lanes x 16
// bytes = one K/V row per wave instruction), bias/GELU/residual epilogues and the argmax
// are all in-kernel.  The only global traffic is weights (L2 resident), the crop's
// cross-attention K/V (kvmem) and the self-attention K/V cache, which the refinement
// pass (ordinary GEMM kernels) reads afterwards together with the tokens.
//
// Arithmetic and rounding points are those of the kernel-per-op path in engine.cpp
// (parseq_forward): bf16 GEMM inputs, fp32 accumulation, fp32 residual stream, bf16
// attention I/O — only fp32 summation orders differ.
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {

constexpr int E = 384, FF = 1536;
constexpr int LDX = E + 8;      // bf16 row stride of the [16][384] activation buffers (16-byte skew)
constexpr int LDH = FF + 8;     // bf16 row stride of the FFN hidden buffer
constexpr int LDT = E + 4;      // f32 row stride of the query stream
constexpr int NTHREADS = 512, NWAVES = 8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct Smem {
  bf16 xa[16 * LDX];
  bf16 xb[16 * LDX];
  bf16 h[16 * LDH];
  float tgt[16 * LDT];
  float logits[16 * 96];
  int tok[16 * 26];
  float2 glut[1024];
};

// out^T[N][16 crops] = W[N][K] . X^T : tile t (16 output features) belongs to wave t % 8; a wave
// works on 3 tiles x 12 k-steps at a time (36 independent 16-byte weight loads in flight).
// epi(n, crop, acc) is called by the lane holding features n..n+3 of crop `crop`.
template <int K, int LD, class Epi>
__device__ __forceinline__ void dec_gemm(const bf16* __restrict__ W, int N, const bf16* X, int wave, int lane, Epi epi) {
  const int fr = lane & 15, fg = lane >> 4;
  const int ntiles = (N + 15) >> 4;
  for (int t0 = wave; t0 < ntiles; t0 += 3 * NWAVES) {
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kc = 0; kc < K; kc += 384) {
      bf16x8 a[3][12];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int row = min((t0 + g * NWAVES) * 16 + fr, N - 1);   // tiles / rows past N are computed on a valid row and dropped
        const bf16* wp = W + (size_t)row * K + kc + fg * 8;
#pragma unroll
        for (int u = 0; u < 12; ++u) a[g][u] = *reinterpret_cast<const bf16x8*>(wp + u * 32);
      }
      __builtin_amdgcn_sched_barrier(0);   // all 36 loads in flight before the first MFMA (hipcc otherwise sinks each load to its use)
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(X + fr * LD + kc + u * 32 + fg * 8);
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][u], b, acc[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int tile = t0 + g * NWAVES;
      if (tile < ntiles) epi(tile * 16 + fg * 4, fr, acc[g]);
    }
  }
}

// LayerNorm of rows `wave` and `wave + 8` of the f32 stream -> bf16 (same lane/element map as layernorm_kernel)
__device__ __forceinline__ void dec_ln(const float* src, int ld, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       bf16* dst, int rows, int wave, int lane) {
  for (int r = wave; r < rows; r += NWAVES) {
    float v[6], s = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { v[k] = src[r * ld + lane + 64 * k]; s += v[k]; }
    const float mean = wsum(s) * (1.0f / 384);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float d = v[k] - mean; q += d * d; }
    const float rstd = rsqrtf(wsum(q) * (1.0f / 384) + eps);
#pragma unroll
    for (int k = 0; k < 6; ++k) { const int c = lane + 64 * k; dst[r * LDX + c] = (bf16)((v[k] - mean) * rstd * gamma[c] + beta[c]); }
  }
}

__device__ __forceinline__ void unpack8(const u32x4& u, float* f) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(&u);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
}

}  // namespace

__global__ __launch_bounds__(NTHREADS) void dec_ar_kernel(DecArParams p, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (p.skip && __builtin_nontemporal_load(p.skip) >= p.skip_n) return;   // tail form: every crop of the batch has emitted EOS
  Smem& S = *reinterpret_cast<Smem*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * G;
  const int rows = min(G, p.N - n0);           // real crops of this workgroup (LDS rows 0..rows-1)
  const int fg = lane >> 4;
  (void)fg;

  // zero the activation buffers once: rows >= `rows` feed the MFMA B operand and must stay finite
  for (int i = tid; i < (int)(sizeof(Smem) / 4); i += NTHREADS) reinterpret_cast<uint32_t*>(smem_raw)[i] = 0u;
  __syncthreads();
  for (int i = tid; i < 16 * 26; i += NTHREADS) S.tok[i] = (i % 26 == 0) ? 95 : 96;   // BOS, then PAD
  for (int i = tid; i < 512; i += NTHREADS) reinterpret_cast<uint4*>(S.glut)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  __syncthreads();
  if (p.first_step == 0) {
    for (int i = tid; i < rows * 26; i += NTHREADS) p.tokens[(size_t)n0 * 26 + i] = S.tok[i];
  } else {
    // tail form: take over the tokens the kernel-per-op steps produced, form token first_step (first maximal index of the
    // previous step's logits, as argmax_kernel / the skinny GEMM's prologue do), and leave if this workgroup's crops are all done
    for (int i = tid; i < rows * 26; i += NTHREADS) if (i % 26 < p.first_step) S.tok[i] = p.tokens[(size_t)n0 * 26 + i];
    __syncthreads();
    for (int r = wave; r < rows; r += NWAVES) {
      const float* x = p.prev_logits + (size_t)(n0 + r) * p.prev_ld;
      float best = -INFINITY; int bi = 0x7fffffff;
      for (int c = lane; c < 95; c += 64) { const float v = x[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0) { S.tok[r * 26 + p.first_step] = bi; p.tokens[(size_t)(n0 + r) * 26 + p.first_step] = bi; }
    }
    __syncthreads();
    int open_rows = 0;
    for (int r = 0; r < rows; ++r) {
      bool eos = false;
      for (int c = 1; c <= p.first_step; ++c) eos = eos || S.tok[r * 26 + c] == 0;
      open_rows += eos ? 0 : 1;
    }
    if (open_rows == 0) return;                   // (uniform: every thread read the same LDS words)
  }

  const float kScale = 0.17677669529663687f;   // 1/sqrt(32)
  // optional phase stamps (diagnostic builds of the caller only): dbg[step*16 + phase] = shader clock, workgroup 0, thread 0
#define DEC_STAMP(ph) do { if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[i * 16 + (ph)] = __builtin_readcyclecounter(); } while (0)

  for (int i = p.first_step; i < 26; ++i) {
    DEC_STAMP(0);
    // ---- content row i: emb[tok] (+ pos_q[i-1]) -> norm_c -> xa
    for (int r = wave; r < rows; r += NWAVES) {
      int tok = S.tok[r * 26 + i];
      tok = tok < 0 ? 0 : (tok > 96 ? 96 : tok);
      float v[6], s = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int c = lane + 64 * k;
        v[k] = p.emb[tok * E + c];
        if (i > 0) v[k] = p.posq[(i - 1) * E + c] + v[k];
        s += v[k];
      }
      const float mean = wsum(s) * (1.0f / 384);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) { const float d = v[k] - mean; q += d * d; }
      const float rstd = rsqrtf(wsum(q) * (1.0f / 384) + 1e-5f);
#pragma unroll
      for (int k = 0; k < 6; ++k) { const int c = lane + 64 * k; S.xa[r * LDX + c] = (bf16)((v[k] - mean) * rstd * p.g_c[c] + p.b_c[c]); }
    }
    __syncthreads();
    DEC_STAMP(1);
    // ---- K/V of content row i -> cache row i (global; the refinement pass needs all 26 rows)
    dec_gemm<E, LDX>(p.w_selfkv, 2 * E, S.xa, wave, lane, [&](int n, int crop, const f32x4& a) {
      if (crop < rows) {
        const float4 b = *reinterpret_cast<const float4*>(p.b_selfkv + n);
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 o;
        o[0] = (bf16)(a[0] + b.x); o[1] = (bf16)(a[1] + b.y); o[2] = (bf16)(a[2] + b.z); o[3] = (bf16)(a[3] + b.w);
        *reinterpret_cast<bf16x4*>(p.kvcache + ((size_t)(n0 + crop) * 26 + i) * 768 + n) = o;
      }
    });
    if (i >= p.nsteps) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's cache stores have reached L2
    __syncthreads();

    DEC_STAMP(2);
    // ---- self attention: query qself[i] against cache rows 0..i -> xb.  Lanes 0..47 hold 8 dims each (head = lane/4).
    for (int r = wave; r < rows; r += NWAVES) {
      const int ln = lane < 48 ? lane : 47;
      float q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = p.qself[i * E + ln * 8 + e];
      const bf16* kv = p.kvcache + (size_t)(n0 + r) * 26 * 768 + ln * 8;
      float s[26];
      {
        u32x4 kq[26];
#pragma unroll
        for (int j = 0; j < 26; ++j) kq[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kv + (size_t)min(j, i) * 768));
#pragma unroll
        for (int j = 0; j < 26; ++j) {
          float kf[8];
          unpack8(kq[j], kf);
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) d += q[e] * kf[e];
          d += __shfl_xor(d, 1);
          d += __shfl_xor(d, 2);
          s[j] = j <= i ? d * kScale : -INFINITY;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the V loads behind the K rows' last use (register budget)
      u32x4 vq[26];
#pragma unroll
      for (int j = 0; j < 26; ++j) vq[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kv + (size_t)min(j, i) * 768 + E));
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 26; ++j) mx = fmaxf(mx, s[j]);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 26; ++j) { s[j] = j <= i ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
      const float inv = 1.0f / sum;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 26; ++j) {
        float vf[8];
        unpack8(vq[j], vf);
        const float pj = s[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pj * vf[e];
      }
      if (lane < 48) {
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = (bf16)o[e];
        *reinterpret_cast<bf16x8*>(S.xb + r * LDX + lane * 8) = ob;
      }
    }
    __syncthreads();
    DEC_STAMP(3);
    // ---- tgt = pos_q[i] + self_out(sa)
    dec_gemm<E, LDX>(p.w_selfout, E, S.xb, wave, lane, [&](int n, int crop, const f32x4& a) {
      const float4 b = *reinterpret_cast<const float4*>(p.b_selfout + n);
      const float4 q = *reinterpret_cast<const float4*>(p.posq + i * E + n);
      *reinterpret_cast<float4*>(S.tgt + crop * LDT + n) = make_float4(a[0] + b.x + q.x, a[1] + b.y + q.y, a[2] + b.z + q.z, a[3] + b.w + q.w);
    });
    __syncthreads();
    dec_ln(S.tgt, LDT, p.g_1, p.b_1, 1e-5f, S.xa, rows, wave, lane);
    __syncthreads();
    DEC_STAMP(4);
    // ---- cross attention query
    dec_gemm<E, LDX>(p.w_crossq, E, S.xa, wave, lane, [&](int n, int crop, const f32x4& a) {
      const float4 b = *reinterpret_cast<const float4*>(p.b_crossq + n);
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o;
      o[0] = (bf16)(a[0] + b.x); o[1] = (bf16)(a[1] + b.y); o[2] = (bf16)(a[2] + b.z); o[3] = (bf16)(a[3] + b.w);
      *reinterpret_cast<bf16x4*>(S.xb + crop * LDX + n) = o;
    });
    __syncthreads();
    DEC_STAMP(5);
    // ---- cross attention over the crop's 128 memory tokens (online softmax, 8 keys per chunk, double buffered) -> xa
    for (int r = wave; r < rows; r += NWAVES) {
      const int ln = lane < 48 ? lane : 47;
      float q[8];
      {
        const bf16x8 qb = *reinterpret_cast<const bf16x8*>(S.xb + r * LDX + ln * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (float)qb[e];
      }
      const bf16* kv = p.kvmem + (size_t)(n0 + r) * 128 * 768 + ln * 8;
      u32x4 kb[2][8], vb[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kb[0][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)j * 768);
        vb[0][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)j * 768 + E);
      }
      float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
      auto consume = [&](const u32x4* kq, const u32x4* vq) {
        float s[8], cm = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float kf[8];
          unpack8(kq[j], kf);
          float d = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) d += q[e] * kf[e];
          d += __shfl_xor(d, 1);
          d += __shfl_xor(d, 2);
          s[j] = d * kScale;
          cm = fmaxf(cm, s[j]);
        }
        const float mn = fmaxf(m, cm);
        const float sc = __expf(m - mn);   // first chunk: exp(-inf) = 0
        l *= sc;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= sc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pj = __expf(s[j] - mn);
          l += pj;
          float vf[8];
          unpack8(vq[j], vf);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += pj * vf[e];
        }
        m = mn;
      };
#pragma unroll 1
      for (int c = 0; c < 16; c += 2) {   // chunk c is in buffer 0; prefetch c+1 into 1, then c+2 into 0
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kb[1][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)((c + 1) * 8 + j) * 768);
          vb[1][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)((c + 1) * 8 + j) * 768 + E);
        }
        consume(kb[0], vb[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < 16) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            kb[0][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)((c + 2) * 8 + j) * 768);
            vb[0][j] = *reinterpret_cast<const u32x4*>(kv + (size_t)((c + 2) * 8 + j) * 768 + E);
          }
        }
        consume(kb[1], vb[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < 48) {
        const float inv = 1.0f / l;
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = (bf16)(o[e] * inv);
        *reinterpret_cast<bf16x8*>(S.xa + r * LDX + lane * 8) = ob;
      }
    }
    __syncthreads();
    DEC_STAMP(6);
    // ---- tgt += cross_out(ca)
    dec_gemm<E, LDX>(p.w_crossout, E, S.xa, wave, lane, [&](int n, int crop, const f32x4& a) {
      const float4 b = *reinterpret_cast<const float4*>(p.b_crossout + n);
      float4* t = reinterpret_cast<float4*>(S.tgt + crop * LDT + n);
      const float4 o = *t;
      *t = make_float4(o.x + (a[0] + b.x), o.y + (a[1] + b.y), o.z + (a[2] + b.z), o.w + (a[3] + b.w));
    });
    __syncthreads();
    dec_ln(S.tgt, LDT, p.g_2, p.b_2, 1e-5f, S.xb, rows, wave, lane);
    __syncthreads();
    DEC_STAMP(7);
    // ---- FFN
    dec_gemm<E, LDX>(p.w_ffn1, FF, S.xb, wave, lane, [&](int n, int crop, const f32x4& a) {
      const float4 b = *reinterpret_cast<const float4*>(p.b_ffn1 + n);
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o;
      o[0] = (bf16)gelu_lut(a[0] + b.x, S.glut); o[1] = (bf16)gelu_lut(a[1] + b.y, S.glut); o[2] = (bf16)gelu_lut(a[2] + b.z, S.glut); o[3] = (bf16)gelu_lut(a[3] + b.w, S.glut);
      *reinterpret_cast<bf16x4*>(S.h + crop * LDH + n) = o;
    });
    __syncthreads();
    DEC_STAMP(8);
    dec_gemm<FF, LDH>(p.w_ffn2, E, S.h, wave, lane, [&](int n, int crop, const f32x4& a) {
      const float4 b = *reinterpret_cast<const float4*>(p.b_ffn2 + n);
      float4* t = reinterpret_cast<float4*>(S.tgt + crop * LDT + n);
      const float4 o = *t;
      *t = make_float4(o.x + (a[0] + b.x), o.y + (a[1] + b.y), o.z + (a[2] + b.z), o.w + (a[3] + b.w));
    });
    __syncthreads();
    dec_ln(S.tgt, LDT, p.g_f, p.b_f, 1e-5f, S.xa, rows, wave, lane);
    __syncthreads();
    DEC_STAMP(9);
    // ---- head -> logits (LDS, optionally global), argmax -> next token
    dec_gemm<E, LDX>(p.w_head, 95, S.xa, wave, lane, [&](int n, int crop, const f32x4& a) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < 95) S.logits[crop * 96 + n + e] = a[e] + p.b_head[n + e];
    });
    __syncthreads();
    for (int r = wave; r < rows; r += NWAVES) {
      const float* x = S.logits + r * 96;
      float best = -INFINITY; int bi = 0x7fffffff;
      for (int c = lane; c < 95; c += 64) {
        const float v = x[c];
        if (p.ar_logits) p.ar_logits[((size_t)(n0 + r) * 26 + i) * 95 + c] = v;
        if (v > best) { best = v; bi = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (lane == 0 && i + 1 < 26) { S.tok[r * 26 + i + 1] = bi; p.tokens[(size_t)(n0 + r) * 26 + i + 1] = bi; }
    }
    __syncthreads();
    DEC_STAMP(10);
  }
}

void launch_dec_ar(const DecArParams& p, int G, hipStream_t s) {
  if (p.N <= 0) return;
  if (G != 4 && G != 8 && G != 16) throw std::runtime_error("dec_ar: crops per workgroup must be 4, 8 or 16");
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)dec_ar_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem))); });
  hipLaunchKernelGGL(dec_ar_kernel, dim3((p.N + G - 1) / G), dim3(NTHREADS), sizeof(Smem), s, p, G);
}

}  // namespace ttr
