// PARSeq-side kernels other than the GEMMs (igemm.hip): crop patchify with the /255
// normalisation (tuatara.cpp:443-446), wavefront-level LayerNorm, the fused encoder
// attention (QK^T -> softmax -> PV on MFMA tiles, one workgroup per crop x head), and
// the small decoder kernels of the 26-step autoregressive loop + refinement pass that
// run inside the TorchScript module the reference calls at tuatara.cpp:307.
#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

// ------------------------------------------------------------------ patchify
// crops u8 [N][32][128][3] -> A [N*128][96], token = py*16+px (4x8 patches), k = (dy*8+dx)*3+c
template <typename T>
__global__ void patchify_kernel(const uint8_t* __restrict__ crops, T* __restrict__ out, int N, int ld) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (token row, dy): 24 contiguous bytes
  int total = N * 128 * 4;
  if (idx >= total) return;
  int dy = idx & 3, row = idx >> 2;
  int n = row >> 7, tok = row & 127, py = tok >> 4, px = tok & 15;
  const uint8_t* src = crops + (((size_t)n * 32 + py * 4 + dy) * 128 + px * 8) * 3;
  T* dst = out + (size_t)row * ld + dy * 24;
#pragma unroll
  for (int i = 0; i < 24; ++i) dst[i] = (T)((float)src[i] / 255.0f);
  for (int i = 96 + dy; i < ld; i += 4) out[(size_t)row * ld + i] = (T)0.f;   // K padded for the GEMM kernel (ld = 128)
}

void launch_patchify(Precision prec, const uint8_t* crops, void* out, int N, int ld, hipStream_t s) {
  if (N <= 0) return;
  if (ld < 96) throw std::runtime_error("patchify: row stride < 96");
  dim3 grid((N * 128 * 4 + 255) / 256);
  if (prec == kBF16) hipLaunchKernelGGL(patchify_kernel<bf16>, grid, dim3(256), 0, s, crops, (bf16*)out, N, ld);
  else hipLaunchKernelGGL(patchify_kernel<float>, grid, dim3(256), 0, s, crops, (float*)out, N, ld);
}

// ------------------------------------------------------------------ LayerNorm: one wave64 per row
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int PER_LANE>
__global__ void layernorm_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float eps, T* __restrict__ out, int out_ld, int M, const int* skip, int skip_n) {
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  constexpr int D = PER_LANE * 64;
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* x = in + (int64_t)row * in_ld;
  float v[PER_LANE], s = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { v[i] = x[lane + 64 * i]; s += v[i]; }
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { float d = v[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
  T* o = out + (int64_t)row * out_ld;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) { int c = lane + 64 * i; o[c] = (T)((v[i] - mean) * rstd * gamma[c] + beta[c]); }
}

void launch_layernorm(Precision prec, const float* in, int in_ld, const float* gamma, const float* beta, float eps, void* out, int out_ld, int M, int D, hipStream_t s,
                      const int* skip, int skip_n) {
  if (D != 384) throw std::runtime_error("layernorm: D must be 384");
  if (M <= 0) return;
  dim3 grid((M + 3) / 4);
  if (prec == kBF16) hipLaunchKernelGGL((layernorm_kernel<bf16, 6>), grid, dim3(256), 0, s, in, in_ld, gamma, beta, eps, (bf16*)out, out_ld, M, skip, skip_n);
  else hipLaunchKernelGGL((layernorm_kernel<float, 6>), grid, dim3(256), 0, s, in, in_ld, gamma, beta, eps, (float*)out, out_ld, M, skip, skip_n);
}

// ------------------------------------------------------------------ encoder attention
// One workgroup (4 waves) per (crop, head): S = 128 tokens, d = 64.  Wave w owns query
// rows 32w..32w+31.  Q/K tiles and V^T are staged in LDS; S = QK^T accumulates in MFMA
// tiles, softmax runs in registers with 16-lane shuffles, P goes through LDS (re-using
// the Q/K space) to become the A operand of P.V.
template <typename T> struct FragT;
template <> struct FragT<bf16> { bf16x8 v; };
template <> struct FragT<float> { float v[8]; };
template <typename T> __device__ __forceinline__ f32x4 mma16x(const FragT<T>& a, const FragT<T>& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16x<bf16>(const FragT<bf16>& a, const FragT<bf16>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16x<float>(const FragT<float>& a, const FragT<float>& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}
// 8 consecutive elements starting at p (16-byte aligned for bf16, 32-byte for f32)
template <typename T> __device__ __forceinline__ FragT<T> ld_frag(const T* p) {
  FragT<T> f;
  if constexpr (sizeof(T) == 2) { uint4 v = *reinterpret_cast<const uint4*>(p); f.v = *reinterpret_cast<bf16x8*>(&v); }
  else { *reinterpret_cast<uint4*>(&f.v[0]) = *reinterpret_cast<const uint4*>(p); *reinterpret_cast<uint4*>(&f.v[4]) = *reinterpret_cast<const uint4*>(p + 4); }
  return f;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_enc_kernel(const T* __restrict__ qkv, T* __restrict__ out) {
  constexpr int S = 128, DH = 64, E = 384, LDQ = DH + 8, LDV = S + 8, LDP = S + 8;  // +8 elements: 16-byte row skew
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* sQ = reinterpret_cast<T*>(smem);            // [128][LDQ]
  T* sK = sQ + S * LDQ;                          // [128][LDQ]
  T* sVt = sK + S * LDQ;                         // [64][LDV]   V transposed: [d][key]
  T* sP = sQ;                                    // [4 waves][32][LDP] aliases Q/K after S is done
  static_assert(4 * 32 * LDP <= 2 * S * LDQ, "P must fit in the Q/K space");

  const int n = blockIdx.x / 6, h = blockIdx.x % 6;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* base = qkv + (int64_t)n * S * (3 * E) + h * DH;
  constexpr int EPC = 16 / sizeof(T), CPR = DH / EPC;  // chunks per row
  for (int q = tid; q < S * CPR; q += 256) {
    int row = q / CPR, c = q % CPR;
    const T* g = base + (int64_t)row * (3 * E) + c * EPC;
    *reinterpret_cast<uint4*>(sQ + row * LDQ + c * EPC) = *reinterpret_cast<const uint4*>(g);
    *reinterpret_cast<uint4*>(sK + row * LDQ + c * EPC) = *reinterpret_cast<const uint4*>(g + E);
    T v[EPC];
    *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(g + 2 * E);
#pragma unroll
    for (int i = 0; i < EPC; ++i) sVt[(c * EPC + i) * LDV + row] = v[i];
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  f32x4 sacc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DH / 32; ++ks) {
    FragT<T> a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = ld_frag<T>(sQ + (wave * 32 + i * 16 + fr) * LDQ + ks * 32 + fg * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      FragT<T> b = ld_frag<T>(sK + (j * 16 + fr) * LDQ + ks * 32 + fg * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i) sacc[i][j] = mma16x<T>(a[i], b, sacc[i][j]);
    }
  }
  __syncthreads();  // every wave is done reading Q/K before P overwrites them

  // softmax over keys; element (i, j, r): row = i*16 + fg*4 + r, key = j*16 + fr.  scale = 1/sqrt(64)
  T* myP = sP + wave * 32 * LDP;
  float rinv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, sacc[i][j][r]);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float e = __expf((sacc[i][j][r] - mx) * 0.125f);
        T et = (T)e;
        sum += (float)et;  // normalise by what P.V will actually sum
        myP[(i * 16 + fg * 4 + r) * LDP + j * 16 + fr] = et;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      rinv[i][r] = 1.0f / sum;
    }
  __syncthreads();

  f32x4 oacc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < S / 32; ++ks) {
    FragT<T> a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = ld_frag<T>(myP + (i * 16 + fr) * LDP + ks * 32 + fg * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      FragT<T> b = ld_frag<T>(sVt + (j * 16 + fr) * LDV + ks * 32 + fg * 8);
#pragma unroll
      for (int i = 0; i < 2; ++i) oacc[i][j] = mma16x<T>(a[i], b, oacc[i][j]);
    }
  }
  T* o = out + (int64_t)n * S * E + h * DH;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = wave * 32 + i * 16 + fg * 4 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[(int64_t)row * E + j * 16 + fr] = (T)(oacc[i][j][r] * rinv[i][r]);
    }
}

static int g_attn_impl = 1;
void set_attn_impl(int v) { g_attn_impl = v; }

void launch_attn_enc(Precision prec, const void* qkv, void* out, int N, hipStream_t s) {
  if (N <= 0) return;
  if (prec == kBF16 && g_attn_impl) return launch_attn_enc2((const bf16*)qkv, (bf16*)out, N, s);
  auto lds_bytes = [](size_t es) { return (2 * 128 * (64 + 8) + 64 * (128 + 8)) * es; };
  if (prec == kBF16) {
    hipLaunchKernelGGL(attn_enc_kernel<bf16>, dim3(N * 6), dim3(256), lds_bytes(2), s, (const bf16*)qkv, (bf16*)out);
  } else {
    static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)attn_enc_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(4))); });
    hipLaunchKernelGGL(attn_enc_kernel<float>, dim3(N * 6), dim3(256), lds_bytes(4), s, (const float*)qkv, (float*)out);
  }
}

// ------------------------------------------------------------------ decoder: content embedding + norm_c
// token i of crop n: i == 0 -> emb[tok] (null context, no position), else pos_q[i-1] + emb[tok].
// emb is pre-scaled by sqrt(384) at export.  One wave per row.
// prev_logits (an AR step, i1 == i0 + 1 >= 2): the token of column i0 is not there yet - it is the first maximal index of the previous step's
// logits row (argmax_kernel's rule), which the row's wave finds first, writes to tokens[n][i0] and counts (done_count: crops whose FIRST EOS
// this is): the argmax launch between two AR steps folded into the next step's first kernel.
template <typename T>
__global__ void dec_embed_ln_kernel(int* __restrict__ tokens, const float* __restrict__ emb, const float* __restrict__ pos_q,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, T* __restrict__ out, int N, int i0, int i1,
                                    const int* skip, int skip_n, int planes, const float* __restrict__ prev_logits, int prev_ld, int C, int* done_count, int eos, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  const int R = i1 - i0;
  int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N * R) return;
  int n = row / R, i = i0 + row % R;
  int tok;
  if (prev_logits) {
    const float* x = prev_logits + (int64_t)n * prev_ld;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { float v = x[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    tok = bi;
    if (lane == 0) {
      tokens[n * 26 + i] = bi;
      if (done_count && bi == eos) {
        bool first = true;
        for (int c = 1; c < i; ++c) first = first && tokens[n * 26 + c] != eos;
        if (first) atomicAdd(done_count, 1);
      }
    }
  } else tok = tokens[n * 26 + i];
  tok = tok < 0 ? 0 : (tok > 96 ? 96 : tok);
  float v[6], s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int c = lane + 64 * k;
    v[k] = emb[tok * 384 + c];
    if (i > 0) v[k] = pos_q[(i - 1) * 384 + c] + v[k];
    s += v[k];
  }
  const float mean = wave_sum(s) * (1.0f / 384);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) { float d = v[k] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / 384) + eps);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    int c = lane + 64 * k;
    const float y = (v[k] - mean) * rstd * gamma[c] + beta[c];
    if (planes) st_split_one(out, row, 384, c, y, planes, rw); else out[(int64_t)row * 384 + c] = (T)y;
  }
  rw.flush(range_flag, range_tag);
}

// split engines (planes out): the same rows in the 48 lanes x 8 values form, the LayerNorm through ln384_row8 - what gemm_skx.hip's token prologue
// does inside the self_kv linear of an AR step of few rows, operation for operation (the two forms give identical planes)
template <int NPL>
__global__ __launch_bounds__(256) void dec_embed_ln_planes_kernel(int* __restrict__ tokens, const float* __restrict__ emb, const float* __restrict__ pos_q,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps, f16* __restrict__ out, int N,
                                                                  int i0, int i1, const int* skip, int skip_n, const float* __restrict__ prev_logits, int prev_ld, int C,
                                                                  int* done_count, int eos, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;
  const int R = i1 - i0;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N * R) return;
  const int n = row / R, i = i0 + row % R;
  int tok;
  if (prev_logits) {
    const float* x = prev_logits + (int64_t)n * prev_ld;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { float v = x[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    tok = bi;
    if (lane == 0) {
      tokens[n * 26 + i] = bi;
      if (done_count && bi == eos) {
        bool first = true;
        for (int c = 1; c < i; ++c) first = first && tokens[n * 26 + c] != eos;
        if (first) atomicAdd(done_count, 1);
      }
    }
  } else tok = tokens[n * 26 + i];
  tok = tok < 0 ? 0 : (tok > 96 ? 96 : tok);
  const bool act = lane < 48;
  const int c = (act ? lane : 0) * 8;
  float v[8];
  {
    const float* x = emb + (int64_t)tok * 384 + c;
    const float4 a = *reinterpret_cast<const float4*>(x), b = *reinterpret_cast<const float4*>(x + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  if (i > 0) {
    const float* pp = pos_q + (int64_t)(i - 1) * 384 + c;
    const float4 pa = *reinterpret_cast<const float4*>(pp), pb = *reinterpret_cast<const float4*>(pp + 4);
    v[0] = pa.x + v[0]; v[1] = pa.y + v[1]; v[2] = pa.z + v[2]; v[3] = pa.w + v[3]; v[4] = pb.x + v[4]; v[5] = pb.y + v[5]; v[6] = pb.z + v[6]; v[7] = pb.w + v[7];
  }
  float y[8];
  ln384_row8(v, act, gamma + c, beta + c, eps, y);
  if (!act) return;
  f16* d = out + (int64_t)row * (NPL * 384) + c;
  f16x8 o0, o1, o2;
  if (NPL == 3) { split3_x8(y, o0, o1, o2, rw); *reinterpret_cast<f16x8*>(d + 768) = o2; }
  else split2_x8(y, o0, o1, rw);
  *reinterpret_cast<f16x8*>(d) = o0; *reinterpret_cast<f16x8*>(d + 384) = o1;
  rw.flush(range_flag, range_tag);
}

void launch_dec_embed_ln(Precision prec, int* tokens, const float* emb, const float* pos_q, const float* gamma, const float* beta, float eps,
                         void* out, int N, int i0, int i1, hipStream_t s, const int* skip, int skip_n, int planes,
                         const float* prev_logits, int prev_ld, int C, int* done_count, int eos) {
  int rows = N * (i1 - i0);
  if (rows <= 0) return;
  if (prev_logits && (i1 != i0 + 1 || i0 < 1)) throw std::runtime_error("dec_embed_ln: the folded argmax belongs to one AR step's column");
  dim3 grid((rows + 3) / 4);
  if (prec != kBF16 && planes == 3 && !(((uintptr_t)emb | (uintptr_t)pos_q | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15))
    hipLaunchKernelGGL(dec_embed_ln_planes_kernel<3>, grid, dim3(256), 0, s, tokens, emb, pos_q, gamma, beta, eps, (f16*)out, N, i0, i1, skip, skip_n, prev_logits, prev_ld, C, done_count, eos, range_ctx().flag, range_ctx().tag);
  else if (prec == kBF16) hipLaunchKernelGGL(dec_embed_ln_kernel<bf16>, grid, dim3(256), 0, s, tokens, emb, pos_q, gamma, beta, eps, (bf16*)out, N, i0, i1, skip, skip_n, 0, prev_logits, prev_ld, C, done_count, eos, range_ctx().flag, range_ctx().tag);
  else hipLaunchKernelGGL(dec_embed_ln_kernel<float>, grid, dim3(256), 0, s, tokens, emb, pos_q, gamma, beta, eps, (float*)out, N, i0, i1, skip, skip_n, planes, prev_logits, prev_ld, C, done_count, eos, range_ctx().flag, range_ctx().tag);
}

// ------------------------------------------------------------------ decoder self attention
// 12 heads x 32 dims, <= 26 keys.  One 384-thread workgroup per (crop, query row); thread = (head, dim).
// q: f32 [26][384] = Wq.norm_q(pos_queries)+bq (crop independent, precomputed at engine creation).
// kvcache: T [N][26][768] (K | V).  Masks (PARSeq.forward):
//   mode 0 (AR step qi): keys 0..qi visible.
//   mode 1 (refine): key j hidden iff j == qi+1 (cloze) or tokens[n][0..j] contains EOS (key padding).
template <typename T>
__global__ __launch_bounds__(384) void dec_self_attn_kernel(const float* __restrict__ q, const T* __restrict__ kv, const int* __restrict__ tokens,
                                                            T* __restrict__ out, int R, int qi0, int mode, const int* skip, int skip_n, int planes, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  __shared__ float sq[384];
  __shared__ float sp[12][28];
  __shared__ int visible[26];
  const int row = blockIdx.x, n = row / R, qi = (mode == 0) ? qi0 : row % R;
  if (skip && mode == 0) {   // ... and per crop: this crop has emitted EOS (token columns 1 .. qi0), nothing behind it is ever read
    for (int c = 1; c <= qi0; ++c) if (tokens[n * 26 + c] == 0) return;
  }
  const int nkeys = (mode == 0) ? qi + 1 : 26;
  const int t = threadIdx.x;
  sq[t] = q[qi * 384 + t];
  if (t < 26) {
    int vis = t < nkeys;
    if (mode == 1) {
      if (t == qi + 1) vis = 0;
      for (int j = 0; j <= t; ++j) if (tokens[n * 26 + j] == 0) vis = 0;
    }
    visible[t] = vis;
  }
  __syncthreads();
  const T* kvn = kv + (int64_t)n * 26 * 768;
  if (t < 12 * 26) {
    int h = t / 26, j = t % 26;
    float s = -INFINITY;
    if (visible[j]) {
      s = 0.f;
      const T* k = kvn + j * 768 + h * 32;
#pragma unroll
      for (int d = 0; d < 32; ++d) s += sq[h * 32 + d] * (float)k[d];
      s *= 0.17677669529663687f;  // 1/sqrt(32)
    }
    sp[h][j] = s;
  }
  __syncthreads();
  if (t < 12) {
    float mx = -INFINITY;
    for (int j = 0; j < 26; ++j) mx = fmaxf(mx, sp[t][j]);
    float sum = 0.f;
    for (int j = 0; j < 26; ++j) { float e = visible[j] ? __expf(sp[t][j] - mx) : 0.f; sp[t][j] = e; sum += e; }
    float inv = 1.0f / sum;
    for (int j = 0; j < 26; ++j) sp[t][j] *= inv;
  }
  __syncthreads();
  {
    int h = t >> 5;
    float acc = 0.f;
    for (int j = 0; j < nkeys; ++j)
      if (visible[j]) acc += sp[h][j] * (float)kvn[j * 768 + 384 + t];      // (a masked key's V row is not read: its weight is 0, but 0 x NaN would not be)
    if (planes) st_split_one(out, row, 384, t, acc, planes, rw); else out[(int64_t)row * 384 + t] = (T)acc;
  }
  rw.flush(range_flag, range_tag);
}

// Refinement pass (mode 1, R = 26 query rows per crop), bf16: ONE workgroup per crop does all 26 rows — the crop's K/V cache
// (26 x 768, 40 KB) goes to LDS once instead of being pulled through L2 by 26 workgroups (1.3 GB of L2 traffic at 1220 crops,
// 190 us).  Same operations in the same order as dec_self_attn_kernel, so the result is bit-identical.
namespace {
constexpr int SR_KS = 776;                                   // padded K/V row (elements): 1552 B, 4 banks apart from row to row
constexpr int SR_LDS = 26 * SR_KS * 2 + 26 * 12 * 26 * 4 + 256;   // K/V rows, probabilities, pad flags [32] + tokens [32]
}
__global__ __launch_bounds__(384) void dec_self_attn_refine_kernel(const float* __restrict__ q, const bf16* __restrict__ kv, const int* __restrict__ tokens,
                                                                   bf16* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sr_smem[];
  bf16* const skv = reinterpret_cast<bf16*>(sr_smem);                                   // [26][SR_KS]
  float* const sp = reinterpret_cast<float*>(sr_smem + 26 * SR_KS * 2);                 // [26 qi][12 h][26 j]
  int* const pad = reinterpret_cast<int*>(sr_smem + 26 * SR_KS * 2 + 26 * 12 * 26 * 4); // key j hidden by an EOS at or before it
  const int n = blockIdx.x, t = threadIdx.x;
  const bf16* kvn = kv + (int64_t)n * 26 * 768;
  for (int v = t; v < 26 * 96; v += 384) {
    const int r = v / 96, c = v - r * 96;
    *reinterpret_cast<bf16x8*>(skv + r * SR_KS + c * 8) = *reinterpret_cast<const bf16x8*>(kvn + r * 768 + c * 8);
  }
  if (t < 26) pad[32 + t] = tokens[n * 26 + t];              // one load per thread (a serial loop of dependent global loads costs ~20 us)
  __syncthreads();
  if (t < 26) {
    int hid = 0;
    for (int j = 0; j <= t; ++j) if (pad[32 + j] == 0) hid = 1;
    pad[t] = hid;
  }
  __syncthreads();
  if (t < 312) {                                             // thread = one (query row, head): its 26 scores stay in registers
    const int qi = t / 12, h = t - qi * 12;
    float4 qv[8];
    const float4* q4 = reinterpret_cast<const float4*>(q + qi * 384 + h * 32);
#pragma unroll
    for (int c = 0; c < 8; ++c) qv[c] = q4[c];
    float sc[26], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 26; ++j) {
      float s = -INFINITY;
      if (!pad[j] && j != qi + 1) {                          // cloze mask + key padding (PARSeq.forward)
        s = 0.f;
        const bf16x8* k8 = reinterpret_cast<const bf16x8*>(skv + j * SR_KS + h * 32);   // 16-byte aligned: rows are 1552 B apart
#pragma unroll
        for (int c = 0; c < 4; ++c) {                        // d = 0 .. 31 in order, as the per-row kernel
          const bf16x8 kk = k8[c];
          const float4 qa = qv[2 * c], qb = qv[2 * c + 1];
          s += qa.x * (float)kk[0]; s += qa.y * (float)kk[1]; s += qa.z * (float)kk[2]; s += qa.w * (float)kk[3];
          s += qb.x * (float)kk[4]; s += qb.y * (float)kk[5]; s += qb.z * (float)kk[6]; s += qb.w * (float)kk[7];
        }
        s *= 0.17677669529663687f;  // 1/sqrt(32)
      }
      sc[j] = s;
      mx = fmaxf(mx, s);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 26; ++j) { const float e = sc[j] == -INFINITY ? 0.f : __expf(sc[j] - mx); sc[j] = e; sum += e; }
    const float inv = 1.0f / sum;
    float* row = sp + t * 26;                                // (qi, h) = (t / 12, t % 12)
#pragma unroll
    for (int j = 0; j < 26; ++j) row[j] = sc[j] * inv;
  }
  __syncthreads();
  {
    const int h = t >> 5;
    float vj[26];                                            // this thread's V column, once
#pragma unroll
    for (int j = 0; j < 26; ++j) vj[j] = (float)skv[j * SR_KS + 384 + t];
    for (int qi = 0; qi < 26; ++qi) {
      const float2* row = reinterpret_cast<const float2*>(sp + (qi * 12 + h) * 26);   // 104-byte rows: 8-byte aligned
      float acc = 0.f;
#pragma unroll
      for (int j2 = 0; j2 < 13; ++j2) { const float2 p2 = row[j2]; acc += p2.x * vj[2 * j2]; acc += p2.y * vj[2 * j2 + 1]; }
      out[((int64_t)n * 26 + qi) * 384 + t] = (bf16)acc;
    }
  }
}
static int g_self_refine = 1;
void set_dec_self_refine(int v) { g_self_refine = v; }

void launch_dec_self_attn(Precision prec, const float* q, const void* kvcache, const int* tokens, void* out, int N, int R, int qi0, int mode, hipStream_t s,
                          const int* skip, int skip_n, int planes) {
  if (N <= 0) return;
  if (prec == kBF16 && mode == 1 && R == 26 && g_self_refine) {
    static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)dec_self_attn_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SR_LDS)); });
    hipLaunchKernelGGL(dec_self_attn_refine_kernel, dim3(N), dim3(384), SR_LDS, s, q, (const bf16*)kvcache, tokens, (bf16*)out);
    return;
  }
  dim3 grid(N * R);
  if (prec == kBF16) hipLaunchKernelGGL(dec_self_attn_kernel<bf16>, grid, dim3(384), 0, s, q, (const bf16*)kvcache, tokens, (bf16*)out, R, qi0, mode, skip, skip_n, 0, range_ctx().flag, range_ctx().tag);
  else hipLaunchKernelGGL(dec_self_attn_kernel<float>, grid, dim3(384), 0, s, q, (const float*)kvcache, tokens, (float*)out, R, qi0, mode, skip, skip_n, planes, range_ctx().flag, range_ctx().tag);
}

// ------------------------------------------------------------------ decoder cross attention
// one workgroup per query row and head group (NH heads in 32 NH threads; gridDim.y = 12 / NH groups: 1 = all 12 heads in 384 threads); 32 dims per head against the
// crop's 128 memory tokens.  A page's few rows spread their heads over workgroups (launch_dec_cross_attn): per head the sums are the same in the same order.
template <typename T, int NH = 12>   // NH: heads per workgroup (12 / gridDim.y)
__global__ __launch_bounds__(NH * 32) void dec_cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ kvmem, T* __restrict__ out, int R,
                                                             const int* skip, int skip_n, const int* done_tok, int done_col, int planes, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  __shared__ float sq[NH * 32];
  __shared__ float sp[NH][128];
  const int row = blockIdx.x, n = row / R, t = threadIdx.x;
  constexpr int NT = NH * 32, nh = NH;
  const int c0 = blockIdx.y * NT;   // this workgroup's columns c0 .. c0 + NT - 1 = heads c0 / 32 .. + NH - 1
  if (done_tok) {   // ... and per crop: it has emitted EOS in token columns 1 .. done_col; what this row would produce is never read
    for (int c = 1; c <= done_col; ++c) if (done_tok[n * 26 + c] == 0) return;
  }
  sq[t] = (float)q[(int64_t)row * 384 + c0 + t];
  __syncthreads();
  const T* kvn = kvmem + (int64_t)n * 128 * 768 + c0;
  for (int idx = t; idx < nh * 128; idx += NT) {
    int h = idx >> 7, j = idx & 127;
    const T* k = kvn + j * 768 + h * 32;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) s += sq[h * 32 + d] * (float)k[d];
    sp[h][j] = s * 0.17677669529663687f;
  }
  __syncthreads();
  {  // softmax: 32 lanes per head row
    int h = t >> 5, l = t & 31;
    float v[4], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = sp[h][l + 32 * i]; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = __expf(v[i] - mx); sum += v[i]; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 4; ++i) sp[h][l + 32 * i] = v[i] * inv;
  }
  __syncthreads();
  {
    int h = t >> 5;
    float acc = 0.f;
    for (int j = 0; j < 128; ++j) acc += sp[h][j] * (float)kvn[j * 768 + 384 + t];
    if (planes) st_split_one(out, row, 384, c0 + t, acc, planes, rw); else out[(int64_t)row * 384 + c0 + t] = (T)acc;
  }
  rw.flush(range_flag, range_tag);
}

// fp32 / f16x4 engines, refinement pass (R = 26 query rows per crop): ONE workgroup per crop instead of one per row, so the crop's 393 KB of K / V
// are read once instead of 26 times (1.16 ms -> the read time at 1280 crops).  Head by head: K_h and V_h (128 x 32 floats each) go to LDS, then
// the 26 x 128 scores, the row softmaxes and P V, every sum in the order dec_cross_attn_kernel uses (d, lanes and j ascending, the same
// shuffles); the two agree like two fp32 evaluations (tests/test_gpu_x4_parity.py).
// blockIdx.y: head group (12 / gridDim.y heads each) - a page's few crops spread over the chip (40 crops: 480 workgroups of one head instead of 40 of
// twelve, 94 -> ~15 us); heads are independent, the sums inside a head keep their order.
__global__ __launch_bounds__(384) void dec_cross_attn_crop_kernel(const float* __restrict__ q, const float* __restrict__ kvmem, float* __restrict__ out,
                                                                  int R, const int* skip, int skip_n, int planes, unsigned* range_flag, unsigned range_tag) {
  RangeWatch rw;   // (split.h)
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;
  constexpr int RMAX = 26;
  // (the queries per head: 26 x 32 floats instead of the crop's 26 x 384 - 49 KB of LDS per workgroup instead of 86, three workgroups per CU instead of one;
  // the kernel is a chain of barriers, and at 1280 crops it ran at 0.9 TB/s of its 0.5 GB)
  __shared__ float sq[RMAX][32];
  __shared__ float sk[128][33];          // (+1: lanes of a wave read 64 different rows at one d)
  __shared__ float sv[128][32];
  __shared__ float sp[RMAX][128];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* kvn = kvmem + (int64_t)n * 128 * 768;
  const int hper = 12 / (int)gridDim.y, h0 = (int)blockIdx.y * hper;
  for (int h = h0; h < h0 + hper; ++h) {
    __syncthreads();                       // the previous head's sq / sk / sv / sp no longer read
    for (int i = t; i < R * 8; i += 384) {
      const int r = i >> 3, c = i & 7;
      *reinterpret_cast<float4*>(&sq[r][c * 4]) = *reinterpret_cast<const float4*>(q + ((int64_t)n * R + r) * 384 + h * 32 + c * 4);
    }
    for (int i = t; i < 128 * 8; i += 384) {
      const int j = i >> 3, c = i & 7;
      const float4 kk = *reinterpret_cast<const float4*>(kvn + j * 768 + h * 32 + c * 4);
      const float4 vv = *reinterpret_cast<const float4*>(kvn + j * 768 + 384 + h * 32 + c * 4);
      sk[j][c * 4] = kk.x; sk[j][c * 4 + 1] = kk.y; sk[j][c * 4 + 2] = kk.z; sk[j][c * 4 + 3] = kk.w;
      *reinterpret_cast<float4*>(&sv[j][c * 4]) = vv;
    }
    __syncthreads();
    {   // scores: thread = key j, rows rg, rg + 3, ...
      const int j = t & 127, rg = t >> 7;
      float kr[32];
#pragma unroll
      for (int d = 0; d < 32; ++d) kr[d] = sk[j][d];
      for (int r = rg; r < R; r += 3) {
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) sc += sq[r][d] * kr[d];
        sp[r][j] = sc * 0.17677669529663687f;
      }
    }
    __syncthreads();
    for (int r = t >> 5; r < R; r += 12) {   // softmax: 32 lanes per row, as dec_cross_attn_kernel
      const int l = t & 31;
      float v[4], mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = sp[r][l + 32 * i]; mx = fmaxf(mx, v[i]); }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = __expf(v[i] - mx); sum += v[i]; }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int i = 0; i < 4; ++i) sp[r][l + 32 * i] = v[i] * inv;
    }
    __syncthreads();
    {   // P V: thread = dim d of rows rg, rg + 12, rg + 24
      const int d = t & 31, rg = t >> 5;
      for (int r = rg; r < R; r += 12) {
        float acc = 0.f;
        for (int j = 0; j < 128; ++j) acc += sp[r][j] * sv[j][d];
        const int64_t row = (int64_t)n * R + r;
        if (planes) st_split_one(out, row, 384, h * 32 + d, acc, planes, rw); else out[row * 384 + h * 32 + d] = acc;
      }
    }
  }
  rw.flush(range_flag, range_tag);
}

// bf16 fast path: a K (or V) row of the crop's memory is 768 bytes = 48 lanes x 16 bytes, so one wave instruction fetches one
// whole row (the kernel above reads 2 bytes per lane at a 1.5 KB stride).  4 waves x 32 keys each, online softmax per wave,
// partial (max, sum, out) merged through LDS.  Lane c < 48 holds dims 8c..8c+7, head = c / 4.
__global__ __launch_bounds__(256) void dec_cross_attn_rows_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kvmem, bf16* __restrict__ out, int R,
                                                                  const int* skip, int skip_n, const int* done_tok, int done_col) {
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  if (done_tok) {   // per crop (AR steps, R = 1): a crop that has emitted EOS in token columns 1 .. done_col reads no more of its 196 KB of K / V -
    const int* tr = done_tok + (int64_t)(blockIdx.x / R) * 26;   // what the rest of the step computes for its row is never read (refinement masks it)
    for (int c = 1; c <= done_col; ++c) if (tr[c] == 0) return;
  }
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  __shared__ float sm[4][48], sl[4][48], so[4][48][8];
  const int row = blockIdx.x, n = row / R, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ln = lane < 48 ? lane : 47;
  float qf[8];
  {
    const bf16x8 qb = *reinterpret_cast<const bf16x8*>(q + (int64_t)row * 384 + ln * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qb[e];
  }
  const bf16* kv = kvmem + ((int64_t)n * 128 + wave * 32) * 768 + ln * 8;
  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {          // 2 chunks of 16 keys: 32 row loads in flight per wave
    u32x4 kb[16], vb[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      kb[j] = *reinterpret_cast<const u32x4*>(kv + (int64_t)(c * 16 + j) * 768);
      vb[j] = *reinterpret_cast<const u32x4*>(kv + (int64_t)(c * 16 + j) * 768 + 384);
    }
    float sc[16], cm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const bf16x8 kk = *reinterpret_cast<const bf16x8*>(&kb[j]);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qf[e] * (float)kk[e];
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      sc[j] = d * 0.17677669529663687f;
      cm = fmaxf(cm, sc[j]);
    }
    const float mn = fmaxf(m, cm), scale = __expf(m - mn);
    l *= scale;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= scale;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float pj = __expf(sc[j] - mn);
      l += pj;
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(&vb[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += pj * (float)vv[e];
    }
    m = mn;
  }
  if (lane < 48) {
    sm[wave][lane] = m; sl[wave][lane] = l;
#pragma unroll
    for (int e = 0; e < 8; ++e) so[wave][lane][e] = o[e];
  }
  __syncthreads();
  if (threadIdx.x < 48) {
    const int c = threadIdx.x;
    float mm = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c]));
    float ll = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __expf(sm[w][c] - mm);
      ll += sl[w][c] * f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += so[w][c][e] * f;
    }
    const float inv = 1.0f / ll;
    bf16x8 ob;
#pragma unroll
    for (int e = 0; e < 8; ++e) ob[e] = (bf16)(acc[e] * inv);
    *reinterpret_cast<bf16x8*>(out + (int64_t)row * 384 + c * 8) = ob;
  }
}

static int g_cross_mfma = 1;
void set_dec_cross_mfma(int v) { g_cross_mfma = v; }
static int g_cross_crop = 1;    // fp32 / f16x4 engines, refinement pass: one workgroup per crop (dec_cross_attn_crop_kernel); 0 = one per row
void set_dec_cross_crop(int v) { g_cross_crop = v; }
static int g_cross_rows_hsplit = 4;   // the per-row kernel at <= 128 rows: head groups per row (a divisor of 12; 0 / 1: one workgroup per row)
void set_dec_cross_rows_hsplit(int v) { g_cross_rows_hsplit = (v >= 1 && v <= 12 && 12 % v == 0) ? v : 1; }
static int g_cross_split = 1;   // f16x4 engine, refinement pass: the matrix-core kernel of attn_cross_split.hip (one wave per crop and head); 0 = the per-crop vector kernel
void set_dec_cross_split(int v) { g_cross_split = v; }

void launch_dec_cross_attn(Precision prec, const void* q, const void* kvmem, void* out, int N, int R, hipStream_t s, const int* skip, int skip_n,
                           const int* done_tok, int done_col, int planes) {
  if (N <= 0) return;
  if (prec == kBF16 && g_cross_mfma && (R == 26 || g_cross_mfma == 2)) return launch_dec_cross_attn_mfma((const bf16*)q, (const bf16*)kvmem, (bf16*)out, N, R, s);   // refinement pass (attn_dec2.hip); 2: the AR steps' single row too
  if (prec != kBF16 && planes == 3 && g_cross_split && g_cross_crop && R > 1 && R <= 32 && !skip)
    return launch_dec_cross_attn_split((const float*)q, (const float*)kvmem, out, N, R, s);
  if (prec != kBF16 && g_cross_crop && R > 1 && R <= 26) {
    const int hsplit = N <= 64 ? 12 : N <= 256 ? 4 : N <= 512 ? 2 : 1;   // head groups: enough workgroups for the chip when the crops are few
    hipLaunchKernelGGL(dec_cross_attn_crop_kernel, dim3(N, hsplit), dim3(384), 0, s, (const float*)q, (const float*)kvmem, (float*)out, R, skip, skip_n, planes, range_ctx().flag, range_ctx().tag);
    return;
  }
  dim3 grid(N * R);
  if (prec == kBF16) { hipLaunchKernelGGL(dec_cross_attn_rows_kernel, grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)kvmem, (bf16*)out, R, skip, skip_n, R == 1 ? done_tok : nullptr, done_col); return; }
  // a page's few rows (an AR step of <= 128 crops): three heads per workgroup, four times the workgroups - each pulls 98 KB of K / V through its CU instead of 393
  const int hgroups = N * R <= 128 ? g_cross_rows_hsplit : 1;
  grid.y = hgroups;
#define TTR_CROSS_ROWS(NH) hipLaunchKernelGGL((dec_cross_attn_kernel<float, NH>), grid, dim3(NH * 32), 0, s, (const float*)q, (const float*)kvmem, (float*)out, R, skip, skip_n, R == 1 ? done_tok : nullptr, done_col, planes, range_ctx().flag, range_ctx().tag)
  switch (hgroups) {
    case 2: TTR_CROSS_ROWS(6); break;
    case 3: TTR_CROSS_ROWS(4); break;
    case 4: TTR_CROSS_ROWS(3); break;
    case 6: TTR_CROSS_ROWS(2); break;
    case 12: TTR_CROSS_ROWS(1); break;
    default: TTR_CROSS_ROWS(12);
  }
#undef TTR_CROSS_ROWS
}

// ------------------------------------------------------------------ argmax (first maximal index, like torch.argmax on CPU)
__global__ void argmax_kernel(const float* __restrict__ logits, int ld, int C, int* __restrict__ tokens, int tok_ld, int col, int N,
                              const int* skip, int skip_n, int* done_count, int eos) {
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const float* x = logits + (int64_t)n * ld;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) { float v = x[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    tokens[n * tok_ld + col] = bi;
    if (done_count && bi == eos) {   // upstream PARSeq's break (system.py): count the crops whose FIRST EOS is this token
      bool first = true;
      for (int c = 1; c < col; ++c) first = first && tokens[n * tok_ld + c] != eos;
      if (first) atomicAdd(done_count, 1);
    }
  }
}

void launch_argmax(const float* logits, int ld, int C, int* tokens, int tok_ld, int col, int N, hipStream_t s, const int* skip, int skip_n, int* done_count, int eos) {
  if (N <= 0) return;
  hipLaunchKernelGGL(argmax_kernel, dim3((N + 3) / 4), dim3(256), 0, s, logits, ld, C, tokens, tok_ld, col, N, skip, skip_n, done_count, eos);
}

__global__ void fill_i32_kernel(int* p, int value, int n, int stride) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[(int64_t)i * stride] = value;
}
void launch_fill_i32(int* p, int value, int n, int stride, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, value, n, stride);
}

}  // namespace ttr
