// Shared device/host definitions for the tuatara MI355X (gfx950) engine.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <stdint.h>

namespace ttr {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum Precision { kBF16 = 0, kF32 = 1, kSplit = 2 };   // kSplit: fp32 tensors, matrix products as four f16 MFMAs (split.h)
enum Act { kActNone = 0, kActRelu = 1, kActGelu = 2 };

// Implicit-GEMM convolution / linear layer.  Activations are NHWC, weights are
// [Cout][taps][Cin] (K contiguous), so both MFMA operands are read K-major.
struct ConvParams {
  const void* in0; int C0;   // source 0 (T, NHWC) and its channel count
  const void* in1; int C1;   // optional source 1: virtual channel concat [in0 | in1]
  int relu0, relu1;          // apply ReLU while loading that source
  int B, H, W;               // spatial size (stride 1, "same" padding)
  int ks, dil;               // kernel size 1 or 3; dilation
  const void* wgt;           // T [Cout][ks*ks*(C0+C1)]
  // gemm_sp.hip only: the activation planes in0 (x_tiled) / the planes output `out` (out_tiled) are laid out as the loader's 1-KiB pieces,
  // [rows / 8][plane][channels / 64][8 rows][64 halves], instead of row-major [row][plane][channels] (the recogniser's encoder, rows a multiple of 8)
  int x_tiled, out_tiled;
  const void* wgt_tiled;     // gemm_sp.hip only, optional: the split weight planes as contiguous 1-KiB loader pieces (Engine::tile_planes); `wgt` stays valid
  const float* bias;         // f32 [Cout] or null
  void* out; int out_ld;     // T output, row stride in elements (may be null)
  float* out_f32; int out_f32_ld;  // optional f32 output
  void* out_relu;            // optional second T output (row stride out_ld): max(value, 0) of what `out` receives
  void* out_pool; int pool_relu;   // gemm2 only: optional 2x2/stride-2 max-pooled T output [B][H/2][W/2][Cout] (row stride out_ld), ReLU first if pool_relu
  const float* resid; int resid_ld; int resid_mod;  // f32 residual added before act; row = m % resid_mod if resid_mod
  int Cout, M, act;
  const void* pre_wgt; const float* pre_bias;   // conv3p only: fuse CRAFT's conv1_1 in front (in0 = u8 canvas [B][H][W][3], pre_wgt = T [64][32]; split: the layer's planes [64][3][32])
  float pre_scale;                              // ... split: conv1_1's output scale (Linear::inv_scale)
  unsigned pre_range_tag;                       // ... split: the range guard's tag for the fused layer's own planes (split.h: RangeWatch; the kernel's range_tag names the layer behind it)
  // gemm_sk only: take the activation rows from LayerNorm(ln_in) instead of in0 (f32 [M][384] rows, stride ln_ld)
  const float* ln_in; int ln_ld; const float* ln_gamma; const float* ln_beta; float ln_eps;
  // gemm_sk + LayerNorm prologue only, PARSeq AR step (tok != null): the row to normalise is emb[token] (+ tok_pos), token =
  // tok[m * tok_ld + tok_col], or, when tok_logits is given, the first maximal index of the f32 row tok_logits[m * tok_logits_ld ..+ tok_C),
  // which column-tile 0 also writes to tok[m * tok_ld + tok_col] (argmax + dec_embed_ln folded into the self_kv GEMM)
  int* tok; int tok_ld, tok_col; const float* tok_logits; int tok_logits_ld, tok_C; const float* tok_emb; const float* tok_pos; int tok_max;
  unsigned long long* dbg;   // gemm_ws diagnostics: shader-clock stamps of workgroup 0 (null = off)
  int store_policy;          // set by the launchers: 0 default, 1 nt, 2 sc0 sc1 nt on the big streaming output stores
  int dbg_flags;             // gemm_ws diagnostics (timing experiments only, results are wrong): 1 = no output stores, 2 = no activation loads
  // PARSeq AR early exit (upstream system.py: the loop breaks once every crop of the batch has emitted EOS): a kernel returns at
  // once when *skip >= skip_n; the token prologue counts crops whose FIRST EOS (id tok_eos) is the token it has just formed
  const int* skip; int skip_n;
  int* done_count; int tok_eos;
  const void* gelu_lut;      // set by launch_gemm2: float2 [1024] = {Phi(x_i), Phi(x_i+1) - Phi(x_i)}, x_i = -8 + i/64
  // split-operand mode (split.h; gemm2 / conv3p): in0 / in1 are f16 planes [M][3 C], wgt f16 [Cout][3 K] = w0 | w0/2^11 | w1; out_scale = 1 / S of
  // the weight tensor; out / out_relu / out_pool are fp32 [M][out_ld] or, with out_planes, f16 planes [M][3 out_ld]
  int split; float out_scale; int out_planes;
  // out_planes == 3: columns from out_full_cols on may leave their third plane unwritten (0 = write every plane everywhere).  PARSeq's qkv output:
  // the attention kernel reads K and V as pairs, so only the 384 Q columns need the triple - a fifth of the layer's store traffic
  int out_full_cols;
  // conv3p.hip, packed pairs (split = 2), CRAFT's conv_cls.4 only: the two 1x1 layers behind it (16 -> 16 ReLU -> 2) as the epilogue of its tile;
  // nothing but the heat map is written.  tail_w6 / tail_w8: f16 [16 rows][w0 | w1][32 k] (fragment rows; conv_cls.8's k slot 8 g + e holds
  // channel 4 g + e for e < 4), tail_b6 [16], tail_b8 [2], tail_s6 / tail_s8 = 1 / S of the two tensors, tail_heat f32 [M][2]
  const void* tail_w6; const void* tail_w8; const float* tail_b6; const float* tail_b8; float tail_s6, tail_s8; float* tail_heat;
  // gemm2.hip, split mode, ks = 1: up_z = fp32 [B][H / 2][W / 2][up_ld] - a tensor at half the resolution whose 2x bilinear upsample (align_corners = false,
  // upsample2x_planes_kernel's arithmetic) at the row's pixel is added before the activation.  CRAFT's upconvN.0 layers are 1x1 convolutions over
  // cat(upsample(y), skip): a 1x1 convolution commutes with the upsample, so W_up . y runs at the low resolution (a quarter of the rows) and arrives here,
  // and the upsampled tensor is never written (engine_craft.cpp: upconv_commuted)
  const float* up_z; int up_ld;
  // ... with up_2d (set by launch_gemm2): a tile's BM rows are a 2-D block of (BM / 16) x 16 pixels instead of BM consecutive ones, so that the four-tap gather of z
  // re-uses its rows inside the workgroup (a 16 x 16 block touches 9 x 9 low-resolution pixels; 256 pixels of an image row touch 2 x 129)
  int up_2d;
  // split-operand mode: the engine's sticky range word and this layer's tag (split.h: RangeWatch); null = not watched.  The launchers fill them from
  // range_ctx() (kernels.h) when the caller left them empty
  unsigned* range_flag; unsigned range_tag;
  // gemm_sp.hip: workgroup (blockIdx.x >> 3) % cu_stagger_groups waits that many cu_stagger_groups-ths of cu_stagger ticks (s_memrealtime, 10 ns) before its first
  // tile, so that the CUs do not reach their epilogues - every CU's store burst - together (0 = off; set by the launcher from the tuning keys)
  int cu_stagger, cu_stagger_groups;
  // persistent kernels with two workgroups per CU (qkv_attn4.hip): tiles handed out by counter instead of by stride - the older workgroup of a CU wins the
  // issue arbitration and runs ahead, so equal shares end unequally.  tile_ctr[xcd] counts the tiles of that XCD's range handed out beyond each workgroup's
  // first, tile_ctr[8] the workgroups that are through; the last one zeroes all nine for the next launch on the stream (tile_counters(), launch.h).
  unsigned* tile_ctr;
};

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7): plenty for a bf16 result
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

// GELU(x) = x * Phi(x), Phi by linear interpolation in the float2[1024] table {Phi(x_i), Phi(x_i+1) - Phi(x_i)}, x_i = -8 + i/64
// (|error| <= 7.4e-6 |x|: two orders below a bf16 ulp)
__device__ __forceinline__ float gelu_lut(float x, const float2* lut) {
  const float u = fmaf(__builtin_amdgcn_fmed3f(x, -8.0f, 7.984375f), 64.0f, 512.0f);   // 0 <= u < 1024
  const float2 t = lut[(int)u];                                                          // u >= 0: truncation is floor
  return x * fmaf(__builtin_amdgcn_fractf(u), t.y, t.x);                                 // fract(u) == u - floor(u) exactly
}

// GELU(x) = x * Phi(x) to fp32 accuracy at a third of the instructions of erff: Phi as one cubic per interval of 1/32 over [-8, 8), its four
// coefficients one 16-byte table entry (float4[512] {a0, a1, a2, a3}: Phi(x_i + t / 32) ~ a0 + t (a1 + t (a2 + t a3)), the Hermite cubic through
// Phi and its derivative at both ends: interpolation error <= h^4 max|d4 Phi| / 384 = 1.4e-9; beyond +-8 Phi is 0 / 1 to fp32).  One LDS read
// and three fmas per value - the form with {Phi, dPhi} pairs (two reads, the coefficients formed per value) cost 224 us of fc1's 835 at 1280 crops.
__device__ __forceinline__ float gelu_hermite(float x, const float2* lut) {
  const float u = fmaf(__builtin_amdgcn_fmed3f(x, -8.0f, 7.96875f), 32.0f, 256.0f);   // 0 <= u < 512
  const float4 c = reinterpret_cast<const float4*>(lut)[(int)u];                       // u >= 0: truncation is floor
  const float t = __builtin_amdgcn_fractf(u);
  return x * fmaf(fmaf(fmaf(c.w, t, c.z), t, c.y), t, c.x);
}

// The same for 8 values: every table read issued before the first polynomial (one LDS round trip per 8 values instead of one per value);
// bit-identical to eight gelu_hermite calls.
__device__ __forceinline__ void gelu_hermite8(float (&v)[8], const float2* lut) {
  float u[8];
  float4 c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    u[e] = fmaf(__builtin_amdgcn_fmed3f(v[e], -8.0f, 7.96875f), 32.0f, 256.0f);
    c[e] = reinterpret_cast<const float4*>(lut)[(int)u[e]];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = __builtin_amdgcn_fractf(u[e]);
    v[e] = v[e] * fmaf(fmaf(fmaf(c[e].w, t, c[e].z), t, c[e].y), t, c[e].x);
  }
}

// GELU of 8 values with the table in LDS at byte address lut_lds.  The table reads are inline asm: hipcc orders every LDS
// read it can see behind all in-flight LDS-DMA loads AND stores of the wave (s_waitcnt vmcnt(0) — it cannot prove that the
// DMA does not write the table), which drains the activation prefetch and waits for write acknowledgements 4x per epilogue.
// Same arithmetic as gelu_lut().
__device__ __forceinline__ void gelu_lut8_lds(float (&v)[8], unsigned lut_lds) {
  float u[8];
  float2 t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    u[e] = fmaf(__builtin_amdgcn_fmed3f(v[e], -8.0f, 7.984375f), 64.0f, 512.0f);
    const unsigned a = lut_lds + ((unsigned)(int)u[e] << 3);
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[e]) : "v"(a));
  }
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] * fmaf(__builtin_amdgcn_fractf(u[e]), t[e].y, t[e].x);
}

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

}  // namespace ttr

#define TTR_HIP_CHECK(expr)                                                                       \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) ttr::hip_fail(#expr, _e, __FILE__, __LINE__);                           \
  } while (0)

namespace ttr {
void hip_fail(const char* what, hipError_t e, const char* file, int line);  // throws std::runtime_error

// hipFuncSetAttribute acts on the CURRENT device: the launchers set a kernel's dynamic-LDS limit once per (kernel, device),
// from whatever thread gets there first (an engine per GPU in one process, engines driven from several threads).
// Compute units of the current device, asked of the runtime once per device (hipGetDeviceProperties fills a ~1.5 KB struct and takes
// tens of microseconds: too slow for a launch path that runs 60 times per batch)
inline int device_cu_count(int fallback = 256) {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fallback;
  std::atomic<int>& c = cache[dev & 63];
  int n = c.load(std::memory_order_relaxed);
  if (n > 0) return n;
  hipDeviceProp_t prop;
  n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : fallback;
  c.store(n, std::memory_order_relaxed);
  return n;
}

struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  std::mutex m;
  template <typename F> void run(F&& f) {
    int dev = 0;
    TTR_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    std::lock_guard<std::mutex> lk(m);
    if (done.load(std::memory_order_relaxed) & bit) return;
    f();
    done.fetch_or(bit, std::memory_order_release);
  }
};
}
