// Split-operand ("f16x4") arithmetic: fp32-equivalent products on the f16 matrix cores.
//
// The reference computes both models in fp32 (tuatara.cpp:363-376, :443-446, :307).  The bf16 MFMA rounds its operands to 8
// bits; the f32 MFMA runs at 1/16 of the 16-bit rate.  This mode writes every fp32 ACTIVATION x exactly as three f16 planes
//
//     x = x0 + (x1 + x2) / 2^11,     x0 = rtz_f16(x),  x1 = rtz_f16((x - x0) 2^11),  x2 = f16((x - x0) 2^11 - x1)
//
// (11 + 11 + 2 significand bits; the lower planes are stored scaled by 2^11 so that x1 never reaches the f16 subnormal range and
// x2's subnormal spacing is 2^-35 in units of x; the round-toward-zero conversions never produce an infinity: |x| < 65504 is the
// only range condition) and every WEIGHT as a pair w S = w0 + w1 (S a power of two per tensor that puts max |w| S in
// [2^13, 2^14): 22+ bits, static), plus the copy w0b = w0 / 2^11.  A product is
//
//     x w S  =  x0 w0  +  x1 w0b  +  x2 w0b  +  x0 w1          (dropped: x1 w1 ~ 2^-22, x2 w1 ~ 2^-33)
//
// i.e. FOUR f16 MFMAs into ONE fp32 accumulator.  Laid out along K this is a plain GEMM with K' = 4 K:
//     activation row [x0 | x1 | x2]       (3 C halves per pixel; the fourth K quarter re-reads plane 0)
//     weight row     [w0 | w0b | w1]      (each plane [taps][Cin]; K quarters 1 and 2 both read w0b)
// so the LDS-DMA kernels keep their loops; the epilogue multiplies by 1 / S (exact).  Measured against the fp32 oracle this is
// at the level of fp32's own rounding noise ("f16x4"); pairs (x0, x1) alone lose the last bit of half the
// activations and miss north_star's 1e-3 on the logits by 2x, bf16 pairs by 10x (oracle/splitsim.py).
#pragma once
#include <hip/hip_runtime.h>

namespace ttr {

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// ---- the mode's one range condition, watched: |x| < 65504 for every value that is written as planes.  The conversions below saturate silently (a triple's
// round-toward-zero planes stop at the largest finite f16, a pair is clamped), so a network whose activations leave the f16 range would come out finite and
// WRONG.  Every kernel that writes planes keeps the running maximum of |x| over the values it splits - one v_max3_f32 per two values, no branch - and, if
// that maximum is not below 65504 (an infinity included), leaves its layer's tag in the engine's sticky flag word: one predicated atomic per thread, at
// the end of the kernel.  The engine reads the word with a batch's results and fails the call naming the layer (ttr_config / tuning key "range_guard").
// A NaN alone does not move a maximum; one can only come from an infinity earlier on (caught) or from the weight file (checked at load).
struct RangeWatch {
  float m = 0.f;
  // m = max(m, |a|, |b|) as the ONE instruction it is (v_max3_f32 with |.| source modifiers; a NaN operand is passed over, as by fmaxf).  Spelled in C the compiler
  // canonicalises each operand first (IEEE mode): 14 instructions per eight values instead of 4, in every epilogue that writes planes.
  __device__ __forceinline__ void note(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
#else
    m = fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
#endif
  }
  __device__ __forceinline__ void note8(const float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) note(v[e], v[e + 1]);
  }
  __device__ __forceinline__ void flush(unsigned* flag, unsigned tag) const {
    if (flag && !(m < 65504.f)) atomicCAS(flag, 0u, tag);       // the first layer that trips keeps the word (launches are stream-ordered)
  }
};

// x - (float)h for h one half of a packed f16 pair: ONE v_fma_mix_f32 reading the half in place (the compiler spells it as a conversion of each half + a packed
// fma: three instructions per pair instead of two, in every epilogue that writes planes).  Exact wherever the C form was: h is x rounded to f16.
__device__ __forceinline__ float less_f16_lo(float x, f16x2 h) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
#else
  return x - (float)h[0];
#endif
}
__device__ __forceinline__ float less_f16_hi(float x, f16x2 h) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
#else
  return x - (float)h[1];
#endif
}
// (float)h * c + x, likewise (c in a scalar register)
__device__ __forceinline__ float fma_f16_lo(f16x2 h, float c, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(c), "v"(x));
  return r;
#else
  return fmaf((float)h[0], c, x);
#endif
}
__device__ __forceinline__ float fma_f16_hi(f16x2 h, float c, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "s"(c), "v"(x));
  return r;
#else
  return fmaf((float)h[1], c, x);
#endif
}

// ---- the exact TRIPLE: two values -> their three planes (packed pairs)
__device__ __forceinline__ void split3_pair(float a, float b, f16x2& p0, f16x2& p1, f16x2& p2, RangeWatch& rw) {
  rw.note(a, b);
  p0 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
  const f32x2 ab = f32x2{a, b} * 2048.f;                                                                     // (one packed multiply)
  const float ra = fma_f16_lo(p0, -2048.f, ab[0]), rb = fma_f16_hi(p0, -2048.f, ab[1]);                      // exact
  p1 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(ra, rb));
  const float sa = less_f16_lo(ra, p1), sb = less_f16_hi(rb, p1);                                            // exact: the last <= 2 bits
  p2 = f16x2{(f16)sa, (f16)sb};
}
// eight values -> three 16-byte plane vectors
__device__ __forceinline__ void split3_x8(const float (&v)[8], f16x8& o0, f16x8& o1, f16x8& o2, RangeWatch& rw) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f16x2 a, b, c;
    split3_pair(v[e], v[e + 1], a, b, c, rw);
    o0[e] = a[0]; o0[e + 1] = a[1]; o1[e] = b[0]; o1[e + 1] = b[1]; o2[e] = c[0]; o2[e + 1] = c[1];
  }
}
__device__ __forceinline__ float join3(f16 a, f16 b, f16 c) { return (float)a + ((float)b + (float)c) * (1.f / 2048.f); }   // (every step exact)

// ---- the PAIR (x0, x1), for layers whose result tolerates ~23.5-bit activations (CRAFT: its heat map stays at fp32 noise level
// with it): x0 = rn_f16(x), x1 = rn_f16((x - x0) 2^11) - round to nearest both times (unbiased; 3 of 4 values are exact,
// the rest off by one fp32 ulp), |x| clamped to the f16 range first.  THREE MFMAs per product: x0 w0 + x0 w1 + x1 w0b.
__device__ __forceinline__ void split2_pair(float a, float b, f16x2& p0, f16x2& p1, RangeWatch& rw) {
  rw.note(a, b);
  a = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f); b = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
  p0 = f16x2{(f16)a, (f16)b};
  const f32x2 ab = f32x2{a, b} * 2048.f;                                                                     // (one packed multiply)
  const float ra = fma_f16_lo(p0, -2048.f, ab[0]), rb = fma_f16_hi(p0, -2048.f, ab[1]);                      // exact
  p1 = f16x2{(f16)ra, (f16)rb};
}
__device__ __forceinline__ void split2_x8(const float (&v)[8], f16x8& o0, f16x8& o1, RangeWatch& rw) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f16x2 a, b;
    split2_pair(v[e], v[e + 1], a, b, rw);
    o0[e] = a[0]; o0[e + 1] = a[1]; o1[e] = b[0]; o1[e + 1] = b[1];
  }
}
__device__ __forceinline__ float join2(f16 a, f16 b) { return (float)a + (float)b * (1.f / 2048.f); }

// 8 consecutive channels n.. of pixel m -> a tensor of `planes` f16 planes (row = planes * ld halves: 3 = triple, 2 = pair), or plain
// fp32 [m][ld] when planes == 0
__device__ __forceinline__ void st_split_n(void* out, int64_t m, int ld, int n, const float (&v)[8], int planes, RangeWatch& rw) {
  if (planes == 3) {
    f16x8 a, b, c;
    split3_x8(v, a, b, c, rw);
    f16* o = reinterpret_cast<f16*>(out) + m * (3 * (int64_t)ld) + n;
    *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + ld) = b; *reinterpret_cast<f16x8*>(o + 2 * ld) = c;
  } else if (planes == 2) {
    f16x8 a, b;
    split2_x8(v, a, b, rw);
    f16* o = reinterpret_cast<f16*>(out) + m * (2 * (int64_t)ld) + n;
    *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + ld) = b;
  } else {
    float* o = reinterpret_cast<float*>(out) + m * ld + n;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// one value -> element c of row `row` of a tensor with `planes` planes (3 = triple; 0 = plain fp32 [row][ld]): for the thread-per-element kernels
__device__ __forceinline__ void st_split_one(void* out, int64_t row, int ld, int c, float v, int planes, RangeWatch& rw) {
  if (planes == 3) {
    f16x2 a, b, d;
    split3_pair(v, 0.f, a, b, d, rw);
    f16* o = reinterpret_cast<f16*>(out) + row * (3 * (int64_t)ld) + c;
    o[0] = a[0]; o[ld] = b[0]; o[2 * ld] = d[0];
  } else {
    reinterpret_cast<float*>(out)[row * ld + c] = v;
  }
}

// K quarter q of the four products (triple) -> activation plane / weight plane; the pair form uses quarters 0, 3, 1
__host__ __device__ __forceinline__ constexpr int split_xplane(int q) { return q == 3 ? 0 : q; }
__host__ __device__ __forceinline__ constexpr int split_wplane(int q) { return q == 0 ? 0 : q == 3 ? 2 : 1; }

// LayerNorm of one 384-wide row held by a wavefront as 48 lanes x 8 consecutive values (lanes 48 - 63 idle, `act` false): y = (v - mean) rstd gamma + beta,
// two passes over the registers, every multiply-add spelled out so that the kernels which share this (layernorm_planes_kernel, gemm_skx.hip's
// LayerNorm prologue) round identically whatever the compiler would contract on its own.  All 64 lanes must call it (wave reductions).
__device__ __forceinline__ void ln384_row8(const float (&v)[8], bool act, const float* gamma8, const float* beta8, float eps, float (&y)[8]) {
  constexpr int D = 384;
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += act ? v[e] : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float d = act ? v[e] - mean : 0.f; q = fmaf(d, d, q); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(fmaf(q, 1.0f / D, eps));
  const float4 g0 = *reinterpret_cast<const float4*>(gamma8), g1 = *reinterpret_cast<const float4*>(gamma8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta8), b1 = *reinterpret_cast<const float4*>(beta8 + 4);
  const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = fmaf(__fmul_rn(v[e] - mean, rstd), gg[e], bb[e]);
}

}  // namespace ttr
