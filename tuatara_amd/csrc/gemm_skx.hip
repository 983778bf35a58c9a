// Skinny split-operand linear for the recogniser's autoregressive steps of a single page (gfx950 / MI355X):
//   out[m][n] = act( (sum_k X[m][k] W[n][k]) / S + bias[n] (+ resid[m][n]) ),   a workgroup = 64 rows x 32 channels x the whole K.
//
// The AR loop of PARSeq (the 26 sequential decoder steps inside the module run at /root/reference/tuatara.cpp:307) issues six linears per step
// on one row per crop: 40 rows for a page.  gemm_sp.hip's 128-row tiles give such a problem 3 - 12 workgroups that walk K in 6 - 24 dependent
// ring steps: ~12 us per launch, 170 launches = a quarter of a page's latency (profiles/r03_single_page_kernel_trace.txt).  These problems are
// bound by one memory round trip, so here
//   * a workgroup owns 32 output channels of a block of <= 64 rows: Cout / 32 workgroups for a page (12 - 48; every weight byte read once),
//     20 times that for the 1280 rows of a 32-page batch, whose AR steps fill 30 - 120 of gemm_sp.hip's 128 x 128 tiles (15 us per launch
//     whatever the layer): hundreds of small workgroups finish such a linear in a third of that;
//   * its four waves split K (wave w takes the 32-deep k steps w, w + 4, ...), fetch their operand fragments straight from global memory into
//     registers - the activation planes are a few tens of KB and live in L2, the weight rows are contiguous - three steps ahead of the MFMAs,
//     and meet once, in LDS, to add the four partial tiles;
//   * exact triples x weight pairs as everywhere in the decoder (split.h): (w0, x0) (w0 / 2^11, x1) (w0 / 2^11, x2) (w1, x0).
// Same ConvParams contract as gemm2.hip's split mode (ks = 1, one source, split = 4): fp32 and / or planes outputs, bias, residual
// (with resid_mod), ReLU / GELU, the AR early exit (`skip`).
#include <stdexcept>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
__device__ __forceinline__ __amdgpu_buffer_rsrc_t skx_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f16x8 skx_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
constexpr int SKX_BN = 32, SKX_DEPTH = 3;
}  // namespace

// RB: 16-row blocks per workgroup (blockIdx.y walks the rows in steps of 16 RB).
// LNP (RB = 1, K = 384): the activation rows are LayerNorm(ln_in rows) - fp32 [M][384], ConvParams::ln_* - normalised, split into exact triples and
// parked in LDS by the workgroup itself (its 16 rows: 24 KB of reads, the arithmetic of layernorm_planes_kernel to the operation), while its first
// weight fragments are on their way.  An AR step's three LayerNorms (norm1 -> cross-attention query, norm2 -> ffn1, the final norm -> head) lose
// their launches - three of fourteen dependent launches per step, each a kernel of ~5 us and a hand-over of ~4.
template <int RB, bool LNP = false>
__global__ __launch_bounds__(256) void gemm_skx_kernel(ConvParams p) {
  static_assert(!LNP || RB == 1, "the LayerNorm prologue is the 16-row form's");
  __shared__ __attribute__((aligned(16))) float part[4][2][RB][64][4];    // the four waves' partial tiles
  __shared__ int skip_now;
  if (p.skip) {   // AR early exit, decided once per WORKGROUP, before the barriers
    if (LNP && p.done_count) {
      // the token prologue's launch bumps the very counter it is asked to honour (its blockIdx.x == 0 workgroups, below): waves that each read it
      // could disagree mid-launch - one returns, the others wait for rows it never parks.  One read per workgroup, handed on through LDS
      if (threadIdx.x == 0) skip_now = __builtin_nontemporal_load(p.skip) >= p.skip_n;
      __syncthreads();
      if (skip_now) return;
    } else if (__builtin_nontemporal_load(p.skip) >= p.skip_n) return;      // (nobody writes the counter during this launch: every wave reads the same value)
  }
  constexpr int XROW = 3 * 384 + 8;                                       // halves per parked row: three planes + 16 bytes (rows 16 bytes apart modulo the banks)
  __shared__ __attribute__((aligned(16))) f16 xs[LNP ? 16 * XROW : 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15, g = lane >> 4;
  RangeWatch rw;                                                            // (split.h: maximum of |x| over the values this lane writes as planes)
  const int K = p.C0, nsteps = K >> 5;                                     // 32-deep k steps
  const int n0 = blockIdx.x * SKX_BN, m0 = blockIdx.y * (16 * RB);
  const __amdgpu_buffer_rsrc_t rsx = skx_rsrc(p.in0, (unsigned)((size_t)p.M * K * 6));      // rows [x0 | x1 | x2]
  const __amdgpu_buffer_rsrc_t rsw = skx_rsrc(p.wgt, (unsigned)((size_t)p.Cout * K * 6));   // rows [w0 | w0b | w1]
  constexpr unsigned OOB = 0x80000000u;
  // lane-constant byte offsets of the fragments' rows (k = 8 g .. 8 g + 7 of the step); the step and the plane ride in the scalar offset
  unsigned xo[RB], wo[2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) { const int m = m0 + rb * 16 + q; xo[rb] = m < p.M ? ((unsigned)m * (unsigned)(3 * K) + g * 8) * 2u : OOB; }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) { const int n = n0 + cb * 16 + q; wo[cb] = n < p.Cout ? ((unsigned)n * (unsigned)(3 * K) + g * 8) * 2u : OOB; }

  struct Frags { f16x8 x[RB][3], w[2][2]; };
  // (always the same number of loads, so that the compiler's s_waitcnt counts stay exact: a step past the end fetches out of range - zero fill, no traffic)
  auto fetch = [&](Frags& f, int step) {
    const unsigned ks = (unsigned)step * 64u;                               // 32 halves
    const unsigned dead = step < nsteps ? 0u : OOB;
    if constexpr (!LNP) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f.x[rb][pl] = skx_load(rsx, xo[rb] | dead, ks + (unsigned)(pl * K) * 2u);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f.w[cb][0] = skx_load(rsw, wo[cb] | dead, ks);                        // w0
      f.w[cb][1] = skx_load(rsw, wo[cb] | dead, ks + (unsigned)(2 * K) * 2u);   // w1
    }
  };
  f32x4 acc[2][RB];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[cb][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f16 dn = (f16)(1.f / 2048.f);
  const f16x8 dnv = {dn, dn, dn, dn, dn, dn, dn, dn};
  auto multiply = [&](const Frags& f) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const f16x8 w0b = f.w[cb][0] * dnv;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        f32x4 a = acc[cb][rb];
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[cb][0], f.x[rb][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b, f.x[rb][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0b, f.x[rb][2], a, 0, 0, 0);
        acc[cb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[cb][1], f.x[rb][0], a, 0, 0, 0);
      }
    }
  };
  // this wave's steps: wave, wave + 4, ...; SKX_DEPTH of them in flight.  Rounds of SKX_DEPTH steps, every fetch and every multiply
  // unconditional (steps past the end multiply zeros): straight-line code with static load counts
  Frags fb[SKX_DEPTH];
  const int mine = (nsteps - wave + 3) >> 2;                                // steps this wave owns
  const int rounds = (mine + SKX_DEPTH - 1) / SKX_DEPTH;
#pragma unroll
  for (int d = 0; d < SKX_DEPTH; ++d) { fetch(fb[d], wave + 4 * d); __builtin_amdgcn_sched_barrier(0); }   // (in stage order: the loop's first wait must not cover stage 2)
  if constexpr (LNP) {
    // wave w normalises rows w, w + 4, w + 8, w + 12 of the block: 48 lanes x 8 values, as layernorm_planes_kernel
    constexpr int D = 384;
    const bool act = lane < 48;
    const int c = (act ? lane : 0) * 8;
    // The wave's four rows TOGETHER, stage by stage - every row's chain is two or three dependent memory round trips (its logits or token, the embedding row,
    // then the arithmetic): one row after the other that was up to twelve round trips in front of the first MFMA, the bulk of this launch's ~11 us at a page's
    // 40 rows.  Same operations per row in the same order: the planes are bit-identical to the row-by-row form (tests).
    int mrow[4]; bool live[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { mrow[rr] = m0 + wave + 4 * rr; live[rr] = mrow[rr] < p.M; }              // (wave-uniform)
    float v[4][8];
    if (p.tok) {
      // an AR step's content rows (ConvParams::tok*, as gemm_sk.hip's token prologue; dec_embed_ln_planes_kernel is the stand-alone form): the row to
      // normalise is emb[token] (+ the position query), token = tok[m][tok_col] or - tok_logits - the first maximal index of the previous step's
      // logits row, which the first column tile also writes back and counts (first EOS of the crop: done_count).  embed + LayerNorm + self_kv: one launch.
      int token[4] = {0, 0, 0, 0};
      if (p.tok_logits) {
        float best[4]; int bi[4];
        if (p.tok_C <= 128) {                                                // (PARSeq: 95 classes) two values per lane, the four rows' loads in flight together
          float t0[4], t1[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float* lg = p.tok_logits + (int64_t)(live[rr] ? mrow[rr] : m0) * p.tok_logits_ld;
            t0[rr] = lane < p.tok_C ? lg[lane] : -INFINITY;
            t1[rr] = lane + 64 < p.tok_C ? lg[lane + 64] : -INFINITY;
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {                                   // (the loop's order: cc = lane, then lane + 64; strict > keeps the first maximum)
            best[rr] = -INFINITY; bi[rr] = 0x7fffffff;
            if (lane < p.tok_C && t0[rr] > best[rr]) { best[rr] = t0[rr]; bi[rr] = lane; }
            if (lane + 64 < p.tok_C && t1[rr] > best[rr]) { best[rr] = t1[rr]; bi[rr] = lane + 64; }
          }
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float* lg = p.tok_logits + (int64_t)(live[rr] ? mrow[rr] : m0) * p.tok_logits_ld;
            best[rr] = -INFINITY; bi[rr] = 0x7fffffff;
            for (int cc = lane; cc < p.tok_C; cc += 64) { const float t = lg[cc]; if (t > best[rr]) { best[rr] = t; bi[rr] = cc; } }
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best[rr], o); const int oi = __shfl_xor(bi[rr], o);
            if (ov > best[rr] || (ov == best[rr] && oi < bi[rr])) { best[rr] = ov; bi[rr] = oi; }
          }
          token[rr] = bi[rr];
        }
        if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (!live[rr]) continue;
            const int m = mrow[rr];
            p.tok[m * p.tok_ld + p.tok_col] = bi[rr];
            if (p.done_count && bi[rr] == p.tok_eos) {
              bool first = true;
              for (int cc = 1; cc < p.tok_col; ++cc) first = first && p.tok[m * p.tok_ld + cc] != p.tok_eos;
              if (first) atomicAdd(p.done_count, 1);
            }
          }
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) token[rr] = live[rr] ? p.tok[mrow[rr] * p.tok_ld + p.tok_col] : 0;
      }
      float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
      if (p.tok_pos) { pa = *reinterpret_cast<const float4*>(p.tok_pos + c); pb = *reinterpret_cast<const float4*>(p.tok_pos + c + 4); }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        int tk = token[rr];
        tk = tk < 0 ? 0 : (tk > p.tok_max ? p.tok_max : tk);
        const float* x = p.tok_emb + (int64_t)tk * D + c;
        const float4 a = *reinterpret_cast<const float4*>(x), b = *reinterpret_cast<const float4*>(x + 4);
        v[rr][0] = a.x; v[rr][1] = a.y; v[rr][2] = a.z; v[rr][3] = a.w; v[rr][4] = b.x; v[rr][5] = b.y; v[rr][6] = b.z; v[rr][7] = b.w;
        if (p.tok_pos) {
          v[rr][0] = pa.x + v[rr][0]; v[rr][1] = pa.y + v[rr][1]; v[rr][2] = pa.z + v[rr][2]; v[rr][3] = pa.w + v[rr][3];
          v[rr][4] = pb.x + v[rr][4]; v[rr][5] = pb.y + v[rr][5]; v[rr][6] = pb.z + v[rr][6]; v[rr][7] = pb.w + v[rr][7];
        }
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float* x = p.ln_in + (int64_t)(live[rr] ? mrow[rr] : m0) * p.ln_ld + c;
        const float4 a = *reinterpret_cast<const float4*>(x), b = *reinterpret_cast<const float4*>(x + 4);
        v[rr][0] = a.x; v[rr][1] = a.y; v[rr][2] = a.z; v[rr][3] = a.w; v[rr][4] = b.x; v[rr][5] = b.y; v[rr][6] = b.z; v[rr][7] = b.w;
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = wave + 4 * rr;
      f16x8 o0 = {0, 0, 0, 0, 0, 0, 0, 0}, o1 = o0, o2 = o0;
      if (live[rr]) {                                                       // (wave-uniform)
        float y[8];
        ln384_row8(v[rr], act, p.ln_gamma + c, p.ln_beta + c, p.ln_eps, y);
        split3_x8(y, o0, o1, o2, rw);
      }
      if (act) {
        f16* d = xs + rl * XROW + c;
        *reinterpret_cast<f16x8*>(d) = o0; *reinterpret_cast<f16x8*>(d + D) = o1; *reinterpret_cast<f16x8*>(d + 2 * D) = o2;
      }
    }
    __syncthreads();
    // K = 384: twelve steps, this wave's are wave, wave + 4, wave + 8 = stages 0, 1, 2 of the one round
#pragma unroll
    for (int d = 0; d < SKX_DEPTH; ++d)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fb[d].x[0][pl] = *reinterpret_cast<const f16x8*>(xs + q * XROW + pl * D + (wave + 4 * d) * 32 + g * 8);
  }
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int d = 0; d < SKX_DEPTH; ++d) {
      multiply(fb[d]);
      __builtin_amdgcn_sched_barrier(0);      // (keeps the refill of a stage behind its MFMAs and in front of the next stage's: the wait counts stay two stages deep)
      fetch(fb[d], wave + 4 * ((r + 1) * SKX_DEPTH + d));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- meet: partial tiles -> LDS, then pair (cb, rb) number pr goes to wave pr % 4
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<f32x4*>(&part[wave][cb][rb][lane][0]) = acc[cb][rb];
  __syncthreads();
#pragma unroll
  for (int pr = 0; pr < 2 * RB; ++pr) {
    if ((pr & 3) != wave) continue;                                         // (wave-uniform)
    const int cb = pr / RB, rb = pr % RB;
    f32x4 v = *reinterpret_cast<const f32x4*>(&part[0][cb][rb][lane][0]);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(&part[w][cb][rb][lane][0]);
    // the lane holds channels n .. n + 3 of row m (C[channel 4 g + r][row q])
    const int n = n0 + cb * 16 + 4 * g, m = m0 + rb * 16 + q;
    if (n >= p.Cout || m >= p.M) continue;
    float o[4];
    // whole groups of four channels with 16-byte aligned rows take vector accesses; a ragged channel count (PARSeq's head: 95 classes) or an odd
    // row stride (its logits rows: 95 floats) goes element by element
    const bool vec = n + 3 < p.Cout;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      if (vec) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w; }
      else { for (int e = 0; e < 4; ++e) if (n + e < p.Cout) bv[e] = p.bias[n + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], p.out_scale, bv[e]);
    if (p.resid) {
      const float* rp = p.resid + (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld + n;
      if (vec && ((p.resid_ld & 3) | ((uintptr_t)p.resid & 15)) == 0) { const float4 r = *reinterpret_cast<const float4*>(rp); o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
      else { for (int e = 0; e < 4; ++e) if (n + e < p.Cout) o[e] += rp[e]; }
    }
    if (p.act == kActRelu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    } else if (p.act == kActGelu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = gelu_exact(o[e]);
    }
    if (p.out) {   // (eligibility: planes outputs have whole groups and aligned rows)
      if (p.out_planes == 3) {
        f16x2 a0, b0, c0, a1, b1, c1;
        split3_pair(o[0], o[1], a0, b0, c0, rw); split3_pair(o[2], o[3], a1, b1, c1, rw);
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        f16* d = reinterpret_cast<f16*>(p.out) + (int64_t)m * (3 * (int64_t)p.out_ld) + n;
        *reinterpret_cast<f16x4*>(d) = f16x4{a0[0], a0[1], a1[0], a1[1]};
        *reinterpret_cast<f16x4*>(d + p.out_ld) = f16x4{b0[0], b0[1], b1[0], b1[1]};
        *reinterpret_cast<f16x4*>(d + 2 * p.out_ld) = f16x4{c0[0], c0[1], c1[0], c1[1]};
      } else if (p.out_planes == 2) {
        f16x2 a0, b0, a1, b1;
        split2_pair(o[0], o[1], a0, b0, rw); split2_pair(o[2], o[3], a1, b1, rw);
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        f16* d = reinterpret_cast<f16*>(p.out) + (int64_t)m * (2 * (int64_t)p.out_ld) + n;
        *reinterpret_cast<f16x4*>(d) = f16x4{a0[0], a0[1], a1[0], a1[1]};
        *reinterpret_cast<f16x4*>(d + p.out_ld) = f16x4{b0[0], b0[1], b1[0], b1[1]};
      } else {
        float* d = reinterpret_cast<float*>(p.out) + (int64_t)m * p.out_ld + n;
        if (vec && ((p.out_ld & 3) | ((uintptr_t)p.out & 15)) == 0) *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
        else { for (int e = 0; e < 4; ++e) if (n + e < p.Cout) d[e] = o[e]; }
      }
    }
    if (p.out_f32) {
      float* d = p.out_f32 + (int64_t)m * p.out_f32_ld + n;
      if (vec && ((p.out_f32_ld & 3) | ((uintptr_t)p.out_f32 & 15)) == 0) *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
      else { for (int e = 0; e < 4; ++e) if (n + e < p.Cout) d[e] = o[e]; }
    }
  }
  rw.flush(p.range_flag, p.range_tag);
}

// shapes: exact triples (split = 4), ks = 1, one source, K a multiple of 32, no pooled / ReLU-copy outputs; planes outputs want whole groups of four
// channels and aligned rows, fp32 outputs take any channel count and row stride (whether the kernel is the faster one for a shape is the
// caller's call: it is for a few thousand rows at most)
// the LayerNorm-prologue form: fp32 rows of 384 in, a page's worth of them (the 16-row workgroups)
// rows up to which the LayerNorm-prologue form is offered (16-row workgroups: every one of them re-reads its 32 weight rows).  At a batch's 1280 rows the three LayerNorm + linear
// pairs of an AR step still win as one launch each: 25 steps 2.39 -> 2.17 ms of kernel time, 75 launches fewer (the token prologue keeps its 256: one crop per row there)
static int g_skx_ln_max_rows = 2048;
void set_gemm_skx_ln_max_rows(int v) { g_skx_ln_max_rows = v; }
bool gemm_skx_ln_eligible(const ConvParams& p) {
  if ((!p.ln_in && !p.tok) || !p.ln_gamma || !p.ln_beta || p.C0 != 384 || p.M > (p.tok ? 256 : g_skx_ln_max_rows)) return false;
  if (p.tok ? (!p.tok_emb || (((uintptr_t)p.tok_emb | (uintptr_t)p.tok_pos) & 15)) : (p.ln_ld % 4 != 0 || ((uintptr_t)p.ln_in & 15))) return false;
  if (((uintptr_t)p.ln_gamma | (uintptr_t)p.ln_beta) & 15) return false;
  ConvParams q = p;
  q.in0 = p.wgt;   // (any aligned pointer: the planes operand is not read)
  return gemm_skx_eligible(q);
}

bool gemm_skx_eligible(const ConvParams& p) {
  if (p.split != 4 || p.ks != 1 || p.C1 != 0 || p.out_pool || p.out_relu || p.M <= 0 || p.Cout <= 0 || p.C0 % 32 != 0 || p.out_full_cols) return false;
  if ((size_t)p.M * p.C0 * 6 >= ((size_t)1 << 31) || (p.M + 63) / 64 > 65535) return false;
  if ((size_t)((p.Cout + 7) / 8 * 8) * p.C0 * 6 >= ((size_t)1 << 31)) return false;
  if (p.out && p.out_planes && (p.Cout % 4 || p.out_ld % 4 || ((uintptr_t)p.out & 15))) return false;
  if (((uintptr_t)p.out | (uintptr_t)p.out_f32 | (uintptr_t)p.resid) & 3) return false;   // (fp32 rows: vector accesses where base and stride allow, else scalar)
  return !(((uintptr_t)p.in0 | (uintptr_t)p.wgt | (uintptr_t)p.bias) & 15);
}

void launch_gemm_skx(const ConvParams& p_in, hipStream_t s) {
  const ConvParams p = with_range_ctx(p_in);
  if (p.ln_in || p.tok) {
    if (!gemm_skx_ln_eligible(p)) throw std::runtime_error("gemm_skx: LayerNorm prologue: shape not supported");
    hipLaunchKernelGGL((gemm_skx_kernel<1, true>), dim3((p.Cout + SKX_BN - 1) / SKX_BN, (p.M + 15) / 16), dim3(256), 0, s, p);
    return;
  }
  if (!gemm_skx_eligible(p)) throw std::runtime_error("gemm_skx: shape not supported");
  // Rows per workgroup: a workgroup streams its rows' activation planes (6 K bytes per row) and its 32 weight rows (4 K bytes each) at what
  // one CU takes in (~64 B / clk), so for a page's worth of rows 16-row workgroups (four times as many of them, each weight row read four
  // times from L2) finish sooner than one 64-row workgroup per 32 channels: K = 1536 at 50 rows 343 KB instead of 786 KB per workgroup;
  // with a batch's 1280 rows there are workgroups enough and 64-row blocks keep the weight re-reads down.
  const int rbsel = p.M <= 256 ? 1 : 4;
  const dim3 grid((p.Cout + SKX_BN - 1) / SKX_BN, (p.M + 16 * rbsel - 1) / (16 * rbsel)), block(256);
  if (rbsel == 1) hipLaunchKernelGGL(gemm_skx_kernel<1>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(gemm_skx_kernel<4>, grid, block, 0, s, p);
}

}  // namespace ttr
