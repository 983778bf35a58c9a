// 3x3 (stride 1, pad 1, dilation 1) NHWC bf16 convolution with a *patch-stationary* input tile (gfx950).
//
// gemm2.hip treats a 3x3 conv as a GEMM over K = 9*Cin and re-stages the activation tile for every
// tap, i.e. it pulls every input pixel through L2 -> LDS nine times.  For the layers with few output
// channels (CRAFT's slice1.*: Cout 64/128) that traffic, not the MFMA pipe, is the limit.  Here the
// workgroup owns an 8 x 32 pixel patch of one image: for each 64-channel chunk the (8+2) x (32+2)
// halo patch is brought into LDS ONCE (LDS-DMA, out-of-image pixels zero-filled by the buffer
// resource's range rule) and the nine taps read their shifted MFMA fragments from it; only the
// 64-channel x tap weight tile streams per K step (double buffered, one barrier per step).
// L2 -> LDS bytes per MFMA drop by 1.7x (Cout 256) to 4x (Cout 64).
//
// Same ConvParams contract, same transposed-MFMA / 16-byte-store epilogue (bias, ReLU, optional
// ReLU copy, optional fused 2x2 max-pool) as gemm2.hip; replaces LibTorch's conv2d inside the CRAFT
// TorchScript module run at tuatara.cpp:376.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

#ifndef TTR_ST_OUT
#define TTR_ST_OUT
// bf16 activation store: streaming (nt) policy unless "store_policy" is 0 (measured: CRAFT -0.2 ms per 32-page step)
__device__ __forceinline__ void st_out(bf16* dst, bf16x8 v, int policy) {
  if (policy == 1 || policy == 2) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
  else *reinterpret_cast<bf16x8*>(dst) = v;
}
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
// Patch geometry: 256 output pixels as 8 x 32 (LPW = 5) or, for maps whose width is not a multiple of 32 (CRAFT's 64 x 48
// level), 16 x 16 (LPW = 4).  Halo patch (PH+2) x (PW+2) pixel slots of 128 B, padded to whole 1-KiB pieces of 8 slots.
template <int LPW> struct Geo {
  static constexpr int PW = 1 << LPW, PH = 256 / PW, HW2 = PW + 2, NHALO = (PH + 2) * HW2;
  static constexpr int XSLOTS = (NHALO + 7) / 8 * 8, XPIECES = XSLOTS / 8, XSTAGE = XSLOTS * 128;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int BN, int WM, int WN, int LPW = 5>
struct C3 {
  using G = Geo<LPW>;
  static constexpr int XPIECES = G::XPIECES, XSTAGE = G::XSTAGE;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = 256 / WM, TN = BN / WN, MI = TM / 16, NJ = TN / 16;
  static constexpr int XPW = (XPIECES + NW - 1) / NW;   // X pieces per wave (last ones may be past the end)
  static constexpr int WP = BN / 8, WPW = WP / NW;
  static constexpr int WSTAGE = BN * 128;
  static constexpr int LDS = 2 * XSTAGE + 2 * WSTAGE;
  static_assert((NW == 8 || NW == 4) && TM % 64 == 0 && TN % 32 == 0 && WP % NW == 0, "conv3p tiling");
};

}  // namespace

// FIRST (BN = 64, Cin = 64 only): the input tensor does not exist.  p.in0 is the u8 canvas [B][H][W][3] and the 64-channel
// halo patch is computed in the prologue: conv1_1 (3 -> 64, 3x3, ReLU; weights p.pre_wgt [64][32] with k = (ky*3+kx)*3+c as
// in conv1_direct_kernel, bias p.pre_bias) evaluated on the 10 x 34 halo pixels and written straight into the LDS patch.
// That removes CRAFT's largest tensor (100 MB per page written and read back) and the conv1_1 launch.
//
// SP (split-operand mode, split.h): f16 planes.  The input holds [x0 | x1 | x2] per pixel (3 Cin halves), the weight rows
// [w0 | w0/2^11 | w1] (3 x 9 Cin).  A 64-channel chunk becomes four virtual chunks: x0 with w0, x0 again with w1
// (the patch stays), x1 with w0/2^11, x2 with w0/2^11 - 36 tap steps into the one accumulator, three patch loads instead of one.
// NP = 2, PACKED pairs, for layers of 32 input channels (CRAFT's head: upconv4.3's output, conv_cls.0 / .2 / .4's inputs): the pixel's row is
// [x0 (32 channels) | x1 (32 channels)] - a pairs tensor of 32 channels IS that row - and one patch load serves both virtual chunks: the first
// multiplies it by weight rows [w0 | w0 / 2^11] (x0 w0 + x1 w0b in ONE chunk), the second by [w1 | 0] (x0 w1).  Two chunks over 128 bytes
// per pixel instead of three over 256 (32 real channels zero-padded to 64): two thirds of the MFMAs, half the bytes.  The weight planes
// keep the [plane 0 | plane 1 | plane 2] row layout with plane 1 unused, so the loaders below need no third case.
// NWS: weight stages (2, 4 or 8; the static-address loop only).  A tap's weights are requested NWS - 1 taps ahead.  Two stages - one tap ahead - is what the
// batch regime wants (the CU's other workgroups cover the wait, the LDS buys residency); a workgroup ALONE on its CU (the deep layers of a single page: 48 - 192
// workgroups on 256 CUs) waits out one L2 / Infinity-Cache round trip per tap with them, ~0.65 us x 216 taps for 512 input channels, and takes four (eight: no further gain).
template <int BN, int WM, int WN, bool FIRST, int XS, int LPW, int NP = 0, int NWS = 2>   // NP: 0 = bf16, 4 = triples, 3 = pairs, 2 = packed pairs (split.h).  XS: patch stages (1 when Cin = 64: a single chunk, and two workgroups fit a CU)
__global__ __launch_bounds__(WM * WN * 64, (XS == 1 && BN <= 128 && !(BN == 128 && WN == 1) ? 4 : 2)) void conv3p_kernel(ConvParams p) {   // (second number: waves per SIMD the register budget allows)
  static_assert(NWS == 2 || ((NWS == 4 || NWS == 8) && XS == 1 && BN <= 64 && (!FIRST || NP == 3)), "more than two weight stages: the static-address loop's");
  using C = C3<BN, WM, WN, LPW>;
  using G = Geo<LPW>;
  constexpr bool SP = NP != 0;
  static_assert(!SP || (XS == 1 && (!FIRST || (NP == 3 && BN == 64 && WM == 4 && WN == 2))), "split mode: single patch stage; the fused first layer on pairs and the 64-wide eight-wave tile only");
  using frag_t = typename std::conditional<SP, f16x8, bf16x8>::type;
  constexpr int PL = NP == 4 ? 3 : NP == 3 ? 2 : 1;      // activation planes per pixel
  constexpr int VC = SP ? NP : 1;                         // virtual chunks per 64-channel chunk: (x0, w0) (x0, w1) (x1, w0b) [(x2, w0b)]
  constexpr int PH = G::PH, PW = G::PW, HW2 = G::HW2, XSLOTS = G::XSLOTS, XPIECES = G::XPIECES, XSTAGE = G::XSTAGE, NHALO = G::NHALO;
  static_assert(!FIRST || LPW == 5, "the fused first layer uses 8 x 32 patches");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const xs = smem;                 // [2][XSLOTS][128 B]  slot pi = pr*34 + pc, chunk c holds channels 8*(c ^ (pi&7))..
  unsigned char* const ws = smem + XS * XSTAGE;   // [2][BN][128 B]      as in gemm2.hip

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- XCD-aware tile order (as gemm2.hip): consecutive tiles = N tiles of one patch, then the next patch
  const int ptx = p.W / PW, pty = p.H / PH;
  const int tilesM = p.B * pty * ptx, tilesN = (p.Cout + BN - 1) / BN;
  const int T = tilesM * tilesN;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile / tilesN, tn = tile - tm * tilesN;
  const int n0 = tn * BN;
  const int b = tm / (pty * ptx), trem = tm - b * pty * ptx, ty = trem / ptx, tx = trem - ty * ptx;
  const int y0 = ty * PH, x0 = tx * PW;

  const int Cin = p.C0, K = 9 * Cin, KP = SP ? 3 * K : K;   // KP: weight row length
  const int nchunks = VC * (Cin >> 6), nsteps = nchunks * 9;   // (virtual chunks when SP)
  const __amdgpu_buffer_rsrc_t rsx = mk_rsrc(p.in0, FIRST ? 16u : (unsigned)((size_t)p.M * Cin * 2 * PL));
  const __amdgpu_buffer_rsrc_t rsw = mk_rsrc(p.wgt, (unsigned)((size_t)p.Cout * KP * 2));
  constexpr unsigned OOB = 0x80000000u;

  // ---- loader state: X piece q = i*8 + wave covers halo slots 8q..8q+7; this lane owns slot 8q + (lane>>3), LDS chunk lane&7
  unsigned xo[C::XPW];
#pragma unroll
  for (int i = 0; i < C::XPW; ++i) {
    const int pi = (i * C::NW + wave) * 8 + (lane >> 3);
    const int pr = pi / HW2, pc = pi - pr * HW2;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const int g = (lane & 7) ^ (pi & 7);
    const bool ok = pi < NHALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
    xo[i] = ok ? (unsigned)((((b * p.H + y) * p.W + x) * (Cin * PL) + g * 8) * 2) : OOB;
  }
  unsigned wb[C::WPW];
#pragma unroll
  for (int j = 0; j < C::WPW; ++j) {
    const int row = (j * C::NW + wave) * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    const int q16 = row & 15;
    const int n = n0 + (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);
    wb[j] = n < p.Cout ? (unsigned)((n * KP + g * 8) * 2) : OOB;
  }
  auto stage_x = [&](int chunk) {
    unsigned char* sb = xs + (chunk & (XS - 1)) * XSTAGE;
    unsigned co = (unsigned)(chunk * 64 * 2);
    if constexpr (SP) { const int q = chunk % VC, pl = q < 2 ? 0 : q - 1; co = (unsigned)((pl * Cin + (chunk / VC) * 64) * 2); }
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const int piece = i * C::NW + wave;
      if (piece < XPIECES) {
        const unsigned vo = xo[i] == OOB ? OOB : xo[i] + co;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sb + piece * 1024), 16, vo, 0, 0, 0);
      }
    }
  };
  // weights of K step (chunk, tap) -> W stage `parity`; [Cout][tap][Cin].  The K offset rides in the instruction's scalar offset (no
  // per-lane arithmetic; the range check sees the lane offset only, so rows past Cout stay out of range)
  auto stage_w = [&](int chunk, int tap, int parity, bool dead = false) {   // dead: a step past the end - out-of-range loads (zero fill, no traffic) keep the wait counts static
    unsigned char* sb = ws + parity * C::WSTAGE;
    unsigned ko = (unsigned)((tap * Cin + chunk * 64) * 2);
    if constexpr (SP) { const int q = chunk % VC, pl = q == 0 ? 0 : q == 1 ? 2 : 1; ko = (unsigned)((pl * K + tap * Cin + (chunk / VC) * 64) * 2); }
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const unsigned vo = dead ? OOB : wb[j];   // (a local copy: with the array element as the builtin's argument hipcc's host pass drops the kernel's stub without a word)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + (j * C::NW + wave) * 1024), 16, vo, dead ? 0u : ko, 0, 0);
    }
  };

  // ---- fragment addressing.  Tile row r = wm*TM + 16i + fr is patch pixel (r>>5, r&31); tap (ky,kx) reads halo slot
  // pi = (py+ky)*34 + px+kx, 16-byte chunk (4kk + fg) ^ (pi & 7): conflict-free for any 16 consecutive slots.
  int pi0[C::MI];
#pragma unroll
  for (int i = 0; i < C::MI; ++i) {
    const int r = wm * C::TM + i * 16 + fr;
    pi0[i] = (r >> LPW) * HW2 + (r & (PW - 1));
  }
  const int wfl = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4) + wn * C::TN * 128;

  f32x4 acc[C::NJ][C::MI];
#pragma unroll
  for (int j = 0; j < C::NJ; ++j)
#pragma unroll
    for (int i = 0; i < C::MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_w(0, 0, 0);
  // (FIRST on pairs: the x1 plane of the halo patch, kept in registers until the third virtual chunk wants it in the one patch stage: [16-slot block of this wave][channel half])
  constexpr int X1R = (FIRST && SP) ? (XSLOTS / 16 + C::NW - 1) / C::NW : 1;
  f16x8 x1k[X1R][2];
  if constexpr (!FIRST) {
    stage_x(0);
    if constexpr (NWS > 2) {   // taps 1 .. NWS - 2 of chunk 0 (NWS - 1 <= 7 < 9 taps)
#pragma unroll
      for (int t = 1; t < NWS - 1; ++t) stage_w(0, t, t);
    }
  } else if constexpr (SP) {
    // ---- conv1_1 on the halo patch in split arithmetic, exactly as conv1_split_kernel (split_ops.hip) evaluates it - the same tables, the same three MFMAs per
    // accumulator in the same order, the same epilogue - so the patch holds the planes that kernel would have written and the layer's result is bit-identical to the
    // two-launch form.  x0 goes to the patch stage, x1 stays in registers (x1k).  CRAFT's largest tensor (64 channels at full resolution: 1.6 GB per eight pages,
    // written once and read once) does not exist.
    if constexpr (NWS > 2) {
#pragma unroll
      for (int t = 1; t < NWS - 1; ++t) stage_w(0, t, t);
    }
    // the canvas rows around the halo as aligned dwords, one per thread: row bytes 3 x0 - 8 .. 3 x0 + 107 (3 x0 and 3 W are multiples of 4: a dword lies inside the row
    // or outside it), of which bytes + 2 .. + 109 are the 36 pixels x0 - 2 .. x0 + 33; zero outside the image
    unsigned char* cv = smem + XS * XSTAGE + NWS * C::WSTAGE;   // [12][29 dwords]
    f16x2* lut = reinterpret_cast<f16x2*>(cv + 1536);           // [256]: the two planes of v / 255, side by side (one read per input)
    const uint8_t* canvas = reinterpret_cast<const uint8_t*>(p.in0);
    if (tid < 12 * 29) {
      const int rr = tid / 29, j = tid - rr * 29;
      const int y = y0 - 2 + rr, byte0 = 3 * x0 - 8 + 4 * j;
      const bool ok = y >= 0 && y < p.H && byte0 >= 0 && byte0 < 3 * p.W;
      reinterpret_cast<unsigned*>(cv)[tid] = ok ? *reinterpret_cast<const unsigned*>(canvas + ((int64_t)b * p.H + y) * p.W * 3 + byte0) : 0u;
    }
    RangeWatch rw1;
    if (tid < 256) {
      f16x2 a, bq;
      split2_pair((float)tid / 255.0f, 0.f, a, bq, rw1);
      lut[tid] = f16x2{a[0], bq[0]};
    }
    const f16* w1p = reinterpret_cast<const f16*>(p.pre_wgt);   // [64][3][32]: w0 | w0 / 2^11 | w1
    f16x8 f1[3][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) f1[pl][jj] = *reinterpret_cast<const f16x8*>(w1p + n * 96 + pl * 32 + fg * 8);
    }
    float b1[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) b1[t][e] = p.pre_bias[32 * t + fg * 8 + e];
    int koff[8];                                       // byte offset of this lane's 8 inputs k = (ky * 3 + kx) * 3 + c inside the canvas patch
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = fg * 8 + e;
      koff[e] = k < 27 ? (k / 9) * 116 + (k % 9) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < X1R; ++r) {
      const int mt = wave + r * C::NW;
      const int pi = mt * 16 + fr;
      const int pr = pi / HW2, pc = pi - pr * HW2;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool inside = pi < NHALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const unsigned char* base = cv + 2 + (pi < NHALO ? pr * 116 + pc * 3 : 0);   // canvas pixel (y - 1, x - 1)
      f16x8 fx0, fx1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int byte = koff[e] >= 0 ? base[koff[e]] : 0;   // table entry 0 = (0, 0)
        const f16x2 q2 = lut[byte];
        fx0[e] = q2[0]; fx1[e] = q2[1];
      }
      f32x4 a1[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[0][jj], fx0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[1][jj], fx1, a, 0, 0, 0);
        a1[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1[2][jj], fx0, a, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = fmaxf(fmaf(a1[2 * t][e], p.pre_scale, b1[t][e]), 0.f);
          v[4 + e] = fmaxf(fmaf(a1[2 * t + 1][e], p.pre_scale, b1[t][4 + e]), 0.f);
        }
        f16x8 o0, o1;
        split2_x8(v, o0, o1, rw1);
        const f16x8 zero = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (!inside) { o0 = zero; o1 = zero; }         // outside the image: conv1_2's zero padding (and the patch's padding slots)
        x1k[r][t] = o1;
        if (pi < XSLOTS) *reinterpret_cast<f16x8*>(xs + pi * 128 + (((4 * t + fg) ^ (pi & 7)) << 4)) = o0;   // (the last block reaches past the patch: the weight stages begin there)
      }
    }
    rw1.flush(p.range_flag, p.pre_range_tag);
    __syncthreads();   // the patch is written with ds_write: the K loop's raw s_barrier would not wait for it
  } else {
    // ---- conv1_1 on the halo patch -> xs stage 0 (the LDS behind the operand stages holds the u8 canvas patch and a u8/255 table)
    unsigned char* cv = smem + XS * XSTAGE + NWS * C::WSTAGE;  // [12][36*3] canvas bytes around the halo (zero outside the image)
    bf16* lut = reinterpret_cast<bf16*>(cv + 1536);    // bf16(v / 255.0f), v = 0..255
    const uint8_t* canvas = reinterpret_cast<const uint8_t*>(p.in0);
    for (int q = tid; q < 12 * 108; q += C::NT) {
      const int rr = q / 108, cc = q - rr * 108, px = cc / 3;
      const int y = y0 - 2 + rr, x = x0 - 2 + px;
      cv[q] = (y >= 0 && y < p.H && x >= 0 && x < p.W) ? canvas[(((int64_t)b * p.H + y) * p.W + x) * 3 + (cc - px * 3)] : (uint8_t)0;
    }
    if (tid < 256) lut[tid] = (bf16)((float)tid / 255.0f);
    bf16x8 f1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
      f1[jj] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.pre_wgt) + n * 32 + fg * 8);
    }
    float b1[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) b1[t][e] = p.pre_bias[32 * t + fg * 8 + e];
    int koff[8];                                       // byte offset of this lane's 8 (tap, channel) pairs inside the canvas patch
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = fg * 8 + e, tp = k / 3;
      koff[e] = k < 27 ? (tp / 3) * 108 + (tp % 3) * 3 + (k - tp * 3) : -1;
    }
    __syncthreads();
    for (int mt = wave; mt * 16 < XSLOTS; mt += C::NW) {
      const int pi = mt * 16 + fr;
      const int pr = pi / HW2, pc = pi - pr * HW2;
      const int y = y0 - 1 + pr, x = x0 - 1 + pc;
      const bool inside = pi < NHALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const unsigned char* base = cv + pr * 108 + pc * 3;   // canvas pixel (y-1, x-1)
      bf16x8 fx;
#pragma unroll
      for (int e = 0; e < 8; ++e) fx[e] = (koff[e] >= 0 && pi < NHALO) ? (bf16)((float)base[koff[e]] * 0.00392156862745098f)   /* == bf16(v / 255.0f) for all 256 byte values */ : (bf16)0.f;
      f32x4 a1[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) a1[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[jj], fx, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = inside ? (bf16)fmaxf(a1[2 * t][e] + b1[t][e], 0.f) : (bf16)0.f;       // outside the image: conv1_2's zero padding
          o[4 + e] = inside ? (bf16)fmaxf(a1[2 * t + 1][e] + b1[t][4 + e], 0.f) : (bf16)0.f;
        }
        if (pi < XSLOTS) *reinterpret_cast<bf16x8*>(xs + pi * 128 + (((4 * t + fg) ^ (pi & 7)) << 4)) = o;
      }
    }
    __syncthreads();   // the patch is written with ds_write: the K loop's raw s_barrier would not wait for it
  }
  // Production variants (one patch stage, 64-pixel wave tiles): the K loop with the nine taps unrolled and every fragment address a
  // lane constant plus a compile-time offset.  Tile i, tap t reads halo slot pi = pib + off(i) + tapoff(t), chunk (4 kk + fg) ^ (pi & 7),
  // and pi & 7 = (pib + sg) & 7 with sg = (off + tapoff) & 7: eight swizzle variants of ONE base address cover all 36 positions, the
  // rest is an immediate.  (MFMA and VALU instructions of a SIMD do not overlap — conv3p_first2s_kernel's stamps — and the runtime-tap
  // loop spends ~30 VALU instructions per K step on these addresses; this one ~8.)
  // Measured per layer (16 pages): the 64-wide tiles gain 6-17 % (upconv2.3 98 -> 91, upconv3.3 136 -> 112, upconv4.3 258 -> 220 us); the
  // 128-wide ones do not (3.27 -3.5 %, 1.7 +11 %: they sit at the 128-register limit and the extra live addresses spill), so those keep
  // the runtime-tap loop below.
  // BN = 128 on FOUR waves (WN = 1: wave tiles of 64 pixels x all 128 channels, 64 MFMAs per tap and wave, 24 fragment reads instead of 32 per 64 MFMAs; two
  // workgroups per CU = two waves per SIMD with 256 registers each) takes this loop too: the experiment behind the tuning key c3_c128_waves.
  constexpr bool STATIC_ADDR = XS == 1 && (BN <= 64 || (BN == 128 && WN == 1)) && C::TM == 64 && (!FIRST || SP);
  if constexpr (STATIC_ADDR) {
    const int pib = ((wm * C::TM) >> LPW) * HW2 + fr;
    int xbase[8];                                          // byte offsets from smem (32-bit LDS arithmetic)
#pragma unroll
    for (int sg = 0; sg < 8; ++sg) xbase[sg] = pib * 128 + ((fg ^ ((pib + sg) & 7)) << 4);
    const int wb0 = XS * XSTAGE + wfl;                     // stage 0; stage 1 = + WSTAGE
    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int par = (chunk + tap) & (NWS - 1);         // (chunk * 9 + tap) % NWS: 9 = 1 modulo 2, 4 and 8
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NWS - 2) * C::WPW) : "memory");   // the taps requested after this one may still be on their way
        __builtin_amdgcn_s_barrier();
        if (tap == 0 && chunk > 0 && (!SP || chunk % VC != 1)) {   // single patch stage: the next chunk's patch can only be fetched now
          if constexpr (FIRST && SP) {                              // ... or, fused first layer, written from the registers that hold its x1 plane
#pragma unroll
            for (int r = 0; r < X1R; ++r) {
              const int mt = wave + r * C::NW, pi = mt * 16 + fr;
              if (pi < XSLOTS) {
#pragma unroll
                for (int t = 0; t < 2; ++t) *reinterpret_cast<f16x8*>(xs + pi * 128 + (((4 * t + fg) ^ (pi & 7)) << 4)) = x1k[r][t];
              }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          } else {
          stage_x(chunk);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          }
        }
        const int wbuf = wb0 + par * C::WSTAGE;
        const int tapoff = (tap / 3) * HW2 + (tap % 3);
        int k64 = 64;                                      // opaque per step: the second K half's addresses (base ^ 64) are formed at their use,
        asm volatile("" : "+v"(k64));                      // not kept in eight more registers across the loop (128-VGPR budget)
        frag_t fx[C::MI], fw[C::NJ];
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) fw[j] = *reinterpret_cast<const frag_t*>(smem + wbuf + j * 2048);
#pragma unroll
        for (int i = 0; i < C::MI; ++i) {
          const int off = ((i * 16) >> LPW) * HW2 + ((i * 16) & (PW - 1)) + tapoff;
          fx[i] = *reinterpret_cast<const frag_t*>(smem + xbase[off & 7] + off * 128);
        }
        if constexpr (NWS == 2) {
          if (tap < 8) stage_w(chunk, tap + 1, par ^ 1);
          else if (chunk + 1 < nchunks) stage_w(chunk + 1, 0, par ^ 1);
        } else {   // tap + NWS - 1 of this chunk or the next, into the stage the previous tap has left; every step issues (past the end: dead loads)
          constexpr int AH = NWS - 1;
          const int t2 = tap + AH >= 9 ? tap + AH - 9 : tap + AH, c2 = tap + AH >= 9 ? chunk + 1 : chunk;
          stage_w(c2, t2, (par + AH) & (NWS - 1), c2 >= nchunks);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          if (kk) {
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) fw[j] = *reinterpret_cast<const frag_t*>(smem + (wbuf ^ k64) + j * 2048);
#pragma unroll
            for (int i = 0; i < C::MI; ++i) {
              const int off = ((i * 16) >> LPW) * HW2 + ((i * 16) & (PW - 1)) + tapoff;
              fx[i] = *reinterpret_cast<const frag_t*>(smem + (xbase[off & 7] ^ k64) + off * 128);
            }
          }
#pragma unroll
          for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) {
              if constexpr (SP) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fx[i], acc[j][i], 0, 0, 0);
              else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[i], acc[j][i], 0, 0, 0);
            }
        }
      }
    }
    if constexpr (NWS > 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing dead requests target this workgroup's LDS
  } else {
  int chunk = 0, tap = 0;
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (XS == 1 && tap == 0 && chunk > 0 && (!SP || chunk % VC != 1)) {   // single patch stage: the next chunk's patch can only be fetched now (its latency is
      stage_x(chunk);                         // covered by the CU's other workgroup, which is what the single stage buys)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const unsigned char* xb = xs + (chunk & (XS - 1)) * XSTAGE;
    const unsigned char* wbuf = ws + (s & 1) * C::WSTAGE;
    const int tapoff = (tap / 3) * HW2 + (tap % 3);
    constexpr bool EARLY = !(XS == 1 && BN <= 128);   // the two-workgroups-per-CU variants have 128 VGPRs: one fragment set at a time
    frag_t fx[EARLY ? 2 : 1][C::MI], fw[EARLY ? 2 : 1][C::NJ];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j) fw[0][j] = *reinterpret_cast<const frag_t*>(wbuf + wfl + j * 2048);
    int xa[C::MI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
      const int pi = pi0[i] + tapoff;
      xa[i] = pi * 128 + ((fg ^ (pi & 7)) << 4);
      fx[0][i] = *reinterpret_cast<const frag_t*>(xb + xa[i]);
    }
    if (s + 1 < nsteps) stage_w(tap == 8 ? chunk + 1 : chunk, tap == 8 ? 0 : tap + 1, (s + 1) & 1);
    if (XS == 2 && tap == 0 && chunk + 1 < nchunks) stage_x(chunk + 1);
    if constexpr (EARLY) {
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) fw[1][j] = *reinterpret_cast<const frag_t*>(wbuf + (wfl ^ 64) + j * 2048);
#pragma unroll
      for (int i = 0; i < C::MI; ++i) fx[1][i] = *reinterpret_cast<const frag_t*>(xb + (xa[i] ^ 64));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) {
            if constexpr (SP) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
            else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk) {
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) fw[0][j] = *reinterpret_cast<const frag_t*>(wbuf + (wfl ^ 64) + j * 2048);
#pragma unroll
          for (int i = 0; i < C::MI; ++i) fx[0][i] = *reinterpret_cast<const frag_t*>(xb + (xa[i] ^ 64));
        }
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) {
            if constexpr (SP) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][j], fx[0][i], acc[j][i], 0, 0, 0);
            else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[0][j], fx[0][i], acc[j][i], 0, 0, 0);
          }
      }
    }
    if (++tap == 9) { tap = 0; ++chunk; }
  }
  }

  // ---- epilogue: lane holds channels n..n+7 of patch pixel (py, px) for every i; i^2 is the pixel below / above
  RangeWatch rw;   // (split.h: the running maximum of |x| over the values this lane writes as planes)
#pragma unroll
  for (int t = 0; t < C::NJ / 2; ++t) {
    const int n = n0 + wn * C::TN + t * 32 + fg * 8;
    if (n >= p.Cout) continue;
    float bv[8];
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    }
    float pooled[C::MI][8];   // only the even-row entries are used (and only when pooling)
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
      const int r = wm * C::TM + i * 16 + fr;
      const int y = y0 + (r >> LPW), x = x0 + (r & (PW - 1));
      const int64_t m = ((int64_t)b * p.H + y) * p.W + x;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (SP) { v[e] = fmaf(acc[2 * t][i][e], p.out_scale, bv[e]); v[4 + e] = fmaf(acc[2 * t + 1][i][e], p.out_scale, bv[4 + e]); }
        else { v[e] = acc[2 * t][i][e] + bv[e]; v[4 + e] = acc[2 * t + 1][i][e] + bv[4 + e]; }
      }
      if (p.act == kActRelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (NP == 2) {
        if (p.tail_heat) {
          // conv_cls.6 + ReLU + conv_cls.8 on this pixel (the reference's last two convolutions, inside CRAFT's module run at tuatara.cpp:376):
          // the lane's 8 values are channels 8 g + e of pixel fr - an MFMA B fragment (k = channel) as they stand.  conv_cls.6 = W6 (16 x 32 k) times
          // them: C[channel 4 g + r][pixel]; its 4 values per lane are, with zeros behind them, the B fragment of conv_cls.8, whose weight rows
          // were laid out for exactly that k order.  Pairs x pairs, three MFMAs per product, as everywhere in CRAFT.
          const f16 dn = (f16)(1.f / 2048.f);
          const f16x8 dnv = {dn, dn, dn, dn, dn, dn, dn, dn};
          const f16* w6 = reinterpret_cast<const f16*>(p.tail_w6) + fr * 64 + fg * 8;
          const f16* w8 = reinterpret_cast<const f16*>(p.tail_w8) + fr * 64 + fg * 8;
          const f16x8 a0 = *reinterpret_cast<const f16x8*>(w6), a1 = *reinterpret_cast<const f16x8*>(w6 + 32);
          const f16x8 c0 = *reinterpret_cast<const f16x8*>(w8), c1 = *reinterpret_cast<const f16x8*>(w8 + 32);
          f16x8 x0, x1;
          split2_x8(v, x0, x1, rw);
          f32x4 t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0 * dnv, x1, t6, 0, 0, 0);
          t6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x0, t6, 0, 0, 0);
          const float4 b6 = *reinterpret_cast<const float4*>(p.tail_b6 + 4 * fg);
          float y[8] = {fmaxf(fmaf(t6[0], p.tail_s6, b6.x), 0.f), fmaxf(fmaf(t6[1], p.tail_s6, b6.y), 0.f), fmaxf(fmaf(t6[2], p.tail_s6, b6.z), 0.f),
                        fmaxf(fmaf(t6[3], p.tail_s6, b6.w), 0.f), 0.f, 0.f, 0.f, 0.f};
          f16x8 y0, y1;
          split2_x8(y, y0, y1, rw);
          f32x4 t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0, y0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0 * dnv, y1, t8, 0, 0, 0);
          t8 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, y0, t8, 0, 0, 0);
          if (fg == 0) *reinterpret_cast<float2*>(p.tail_heat + m * 2) = make_float2(fmaf(t8[0], p.tail_s8, p.tail_b8[0]), fmaf(t8[1], p.tail_s8, p.tail_b8[1]));
          continue;
        }
      }
      if constexpr (SP) {
        if (p.out) st_split_n(p.out, m, p.out_ld, n, v, p.out_planes, rw);
        if (p.out_relu) {
          float w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = fmaxf(v[e], 0.f);
          st_split_n(p.out_relu, m, p.out_ld, n, w, p.out_planes, rw);
        }
      } else {
      if (p.out) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        st_out(reinterpret_cast<bf16*>(p.out) + m * p.out_ld + n, o, p.store_policy);
      }
      if (p.out_relu) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(v[e], 0.f);
        st_out(reinterpret_cast<bf16*>(p.out_relu) + m * p.out_ld + n, o, p.store_policy);
      }
      }
      if (p.out_pool) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xv = p.pool_relu ? fmaxf(v[e], 0.f) : v[e];
          xv = fmaxf(xv, __shfl_xor(xv, 1));           // horizontal partner: px ^ 1
          pooled[i][e] = xv;
        }
      }
    }
    if (p.out_pool) {
#pragma unroll
      for (int i = 0; i < C::MI; ++i) {
        constexpr int PD = PW / 16;                    // m-tiles per patch row: the pixel below is PD tiles further
        if ((i & PD) == 0) {                           // even patch row: partner row is tile i + PD
          const int r = wm * C::TM + i * 16 + fr;
          const int yo = (y0 + (r >> LPW)) >> 1, xo2 = (x0 + (r & (PW - 1))) >> 1;
          if constexpr (SP) {
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = fmaxf(pooled[i][e], pooled[i + PD][e]);
            if ((fr & 1) == 0) st_split_n(p.out_pool, ((int64_t)b * (p.H >> 1) + yo) * (p.W >> 1) + xo2, p.out_ld, n, w, p.out_planes, rw);
          } else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(pooled[i][e], pooled[i + PD][e]);
          if ((fr & 1) == 0)
            st_out(reinterpret_cast<bf16*>(p.out_pool) + (((int64_t)b * (p.H >> 1) + yo) * (p.W >> 1) + xo2) * p.out_ld + n, o, p.store_policy);
          }
        }
      }
    }
  }
  if constexpr (SP) rw.flush(p.range_flag, p.range_tag);
}

// ------------------------------------------------------------------------------------------------------------------
// CRAFT's first two convolutions as one PERSISTENT kernel (conv1_1 3->64 + ReLU, conv1_2 64->64 + ReLU, optional 2x2 max
// pool): the FIRST variant above spends 9 barriers and 72 KB of weight staging on every 8x32 patch.  Here all nine taps of
// conv1_2's weights (72 KB) are staged into LDS once per workgroup and stay; a workgroup walks patches with stride gridDim.
// Per patch a wave has two independent jobs: the 144 MFMAs of conv1_2 on patch p (LDS -> matrix pipe) and conv1_1 on the
// 10x34 halo of patch p+1 into the other patch buffer (byte gathers, VALU, LDS writes).  Waves 0-3 run them in that order,
// waves 4-7 in the opposite order: the two waves that share a SIMD (w and w+4) then occupy complementary pipes instead of
// queueing for the same one (measured per patch before: 6.0k cycles of prologue + 4.2k of MFMA, all waves in lock step).
__global__ __launch_bounds__(512) void conv3p_first2_kernel(ConvParams p) {
  using G = Geo<5>;
  constexpr int PH = G::PH, PW = G::PW, HW2 = G::HW2, NHALO = G::NHALO;
  constexpr int XB = NHALO * 128;                       // one patch buffer (340 slots, no padding: 2 buffers + weights = 160 KB)
  constexpr int NW = 8, TM = 64, MI = 4;                // 8 waves as 4 (pixels) x 2 (channels): wave tile 64 px x 32 ch
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const wsm = smem;                      // [9 taps][64 rows][128 B]   conv1_2 weights, gemm2 row/chunk permutation
  unsigned char* const xs = smem + 9 * 8192;            // [2][340][128 B]            conv1_1 output on the halo patch
  unsigned char* const cv = xs + 2 * XB;                // [12][108] canvas bytes of the patch whose conv1_1 runs next
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int ptx = p.W / PW, pty = p.H / PH, npatch = p.B * pty * ptx;

  // ---- resident conv1_2 weights: tap t tile = rows n (permuted as in gemm2) x 64 k; 8 pieces of 1 KiB per tap, one per wave
  {
    const __amdgpu_buffer_rsrc_t rsw = mk_rsrc(p.wgt, (unsigned)((size_t)p.Cout * 576 * 2));
    const int row = wave * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    const int q16 = row & 15;
    const int n = (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned vo = n < p.Cout ? (unsigned)((n * 576 + t * 64 + g * 8) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(wsm + t * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  }
  // conv1_1 operands (registers, for the whole launch)
  bf16x8 f1[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
    f1[jj] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.pre_wgt) + n * 32 + fg * 8);
  }
  float b1[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) b1[t][e] = p.pre_bias[32 * t + fg * 8 + e];
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fg * 8 + e, tp = k / 3;
    koff[e] = k < 27 ? (tp / 3) * 108 + (tp % 3) * 3 + (k - tp * 3) : 0;   // k >= 27 multiplies a zero weight column: any byte will do
  }
  // the (up to) three halo m-tiles of this wave: slot, canvas offset and halo coordinates do not depend on the patch
  int h_pi[3], h_off[3], h_pr[3], h_pc[3];
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    const int pi = (wave + NW * t3) * 16 + fr;
    h_pi[t3] = pi; h_pr[t3] = pi / HW2; h_pc[t3] = pi - h_pr[t3] * HW2;
    h_off[t3] = pi < NHALO ? h_pr[t3] * 108 + h_pc[t3] * 3 : 0;
  }
  float bv[8];                                           // conv1_2 bias of this lane's 8 channels
  {
    const int n = wn * 32 + fg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (p.bias && n + e < p.Cout) ? p.bias[n + e] : 0.f;
  }
  int pi0[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) { const int r = wm * TM + i * 16 + fr; pi0[i] = (r >> 5) * HW2 + (r & 31); }
  const int wfl = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4) + wn * 32 * 128;

  const uint8_t* canvas = reinterpret_cast<const uint8_t*>(p.in0);
  auto patch_origin = [&](int patch, int& b, int& y0, int& x0) {
    b = patch / (pty * ptx);
    const int trem = patch - b * pty * ptx, ty = trem / ptx;
    y0 = ty * PH; x0 = (trem - ty * ptx) * PW;
  };
  // canvas bytes: thread tid owns bytes tid, tid+512, tid+1024 of the 12 x 108 byte patch around the halo
  auto canvas_byte = [&](int patch, int q) -> uint8_t {
    int b, y0, x0;
    patch_origin(patch, b, y0, x0);
    const int rr = q / 108, cc = q - rr * 108, px = cc / 3;
    const int y = y0 - 2 + rr, x = x0 - 2 + px;
    return (q < 12 * 108 && y >= 0 && y < p.H && x >= 0 && x < p.W) ? canvas[(((int64_t)b * p.H + y) * p.W + x) * 3 + (cc - px * 3)] : (uint8_t)0;
  };
  // conv1_1 (+ bias, ReLU; zero outside the image = conv1_2's padding) on the halo of `patch` from cv into patch buffer xb
  auto prologue = [&](int patch, unsigned char* xb) {
    int b, y0, x0;
    patch_origin(patch, b, y0, x0);
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
      if ((wave + NW * t3) * 16 < NHALO) {               // wave-uniform
        const int pi = h_pi[t3];
        const int y = y0 - 1 + h_pr[t3], x = x0 - 1 + h_pc[t3];
        const bool inside = pi < NHALO && y >= 0 && y < p.H && x >= 0 && x < p.W;
        const unsigned char* base = cv + h_off[t3];
        bf16x8 fx;
#pragma unroll
        for (int e = 0; e < 8; ++e) fx[e] = (bf16)((float)base[koff[e]] * 0.00392156862745098f);   // == bf16(v / 255.0f) for every byte value
        f32x4 a1[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) a1[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[jj], fx, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = inside ? (bf16)fmaxf(a1[2 * t][e] + b1[t][e], 0.f) : (bf16)0.f;
            o[4 + e] = inside ? (bf16)fmaxf(a1[2 * t + 1][e] + b1[t][4 + e], 0.f) : (bf16)0.f;
          }
          if (pi < NHALO) *reinterpret_cast<bf16x8*>(xb + pi * 128 + (((4 * t + fg) ^ (pi & 7)) << 4)) = o;
        }
      }
    }
  };

  int patch = blockIdx.x;
  uint8_t cb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) cb[k] = patch < npatch ? canvas_byte(patch, tid + 512 * k) : (uint8_t)0;
#pragma unroll
  for (int k = 0; k < 3; ++k) if (tid + 512 * k < 12 * 108) cv[tid + 512 * k] = cb[k];
  __syncthreads();
  if (patch < npatch) prologue(patch, xs);
  if (patch + (int)gridDim.x < npatch) {
#pragma unroll
    for (int k = 0; k < 3; ++k) cb[k] = canvas_byte(patch + gridDim.x, tid + 512 * k);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the resident weights have landed

#define F2_STAMP(ph) do { if (p.dbg && blockIdx.x == 0 && (tid & 255) == 0 && it < 24) p.dbg[((tid >> 8) * 24 + it) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)
  for (int it = 0; patch < npatch; patch += gridDim.x, ++it) {
    int b, y0, x0;
    patch_origin(patch, b, y0, x0);
    const int nextp = patch + gridDim.x;
    const unsigned char* xcur = xs + (it & 1) * XB;
    unsigned char* xnext = xs + ((it + 1) & 1) * XB;
    F2_STAMP(0);
    __syncthreads();                                     // A: everyone finished the previous iteration (its cv and patch-buffer reads)
    F2_STAMP(1);
    if (nextp < npatch) {
#pragma unroll
      for (int k = 0; k < 3; ++k) if (tid + 512 * k < 12 * 108) cv[tid + 512 * k] = cb[k];
      if (nextp + (int)gridDim.x < npatch) {
#pragma unroll
        for (int k = 0; k < 3; ++k) cb[k] = canvas_byte(nextp + gridDim.x, tid + 512 * k);   // in flight for a whole iteration
      }
    }
    F2_STAMP(2);
    __syncthreads();                                     // B: cv (patch p+1) and the patch buffer of p are complete
    F2_STAMP(3);

    f32x4 acc[2][MI];
    auto mfma_phase = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tapoff = (t / 3) * HW2 + (t % 3);
        int xa[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) { const int pi = pi0[i] + tapoff; xa[i] = pi * 128 + ((fg ^ (pi & 7)) << 4); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          bf16x8 fw[2], fx[MI];
#pragma unroll
          for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(wsm + t * 8192 + (wfl ^ (kk * 64)) + j * 2048);
#pragma unroll
          for (int i = 0; i < MI; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(xcur + (xa[i] ^ (kk * 64)));
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[i], acc[j][i], 0, 0, 0);
        }
      }
    };
    if (wave < 4) {
      mfma_phase();
      F2_STAMP(4);
      if (nextp < npatch) prologue(nextp, xnext);
    } else {
      if (nextp < npatch) prologue(nextp, xnext);
      F2_STAMP(4);
      mfma_phase();
    }
    F2_STAMP(5);
    // ---- epilogue: lane holds channels n..n+7 of patch pixel (py, px) for every i; tile i + 2 is the pixel below
    const int n = wn * 32 + fg * 8;
    if (n < p.Cout) {
      float pooled[MI][8];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * TM + i * 16 + fr;
        const int64_t m = ((int64_t)b * p.H + y0 + (r >> 5)) * p.W + x0 + (r & 31);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaxf(acc[0][i][e] + bv[e], 0.f); v[4 + e] = fmaxf(acc[1][i][e] + bv[4 + e], 0.f); }
        if (p.out) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          st_out(reinterpret_cast<bf16*>(p.out) + m * p.out_ld + n, o, p.store_policy);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pooled[i][e] = fmaxf(v[e], __shfl_xor(v[e], 1));
      }
      if (p.out_pool) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if ((i & 2) == 0) {
            const int r = wm * TM + i * 16 + fr;
            const int yo = (y0 + (r >> 5)) >> 1, xo2 = (x0 + (r & 31)) >> 1;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(pooled[i][e], pooled[i + 2][e]);
            if ((fr & 1) == 0)
              st_out(reinterpret_cast<bf16*>(p.out_pool) + (((int64_t)b * (p.H >> 1) + yo) * (p.W >> 1) + xo2) * p.out_ld + n, o, p.store_policy);
          }
        }
      }
    }
    F2_STAMP(6);
  }
#undef F2_STAMP
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the kernel above (stamps of that one, tools/first2_stamps.py: 12.9k cycles per patch of which the
// matrix pipe needs 4.6k — every wave ran canvas addressing (1.3k), its share of conv1_1 (3-4k), 144 MFMAs at half rate
// (4.3-5.7k) and the pooling epilogue (1.5-1.9k) one after the other).  Here the two waves of a SIMD have different jobs:
//   waves 0-3 (consumers): conv1_2 on patch p, 64 pixels (two patch rows) x all 64 channels each: 288 MFMAs back to back with
//     the fragments of the next K step already in registers, then the pooled epilogue (vertical max in registers, horizontal
//     max by DPP, bias + ReLU after the max: x -> relu(x + b) is monotone, so the result is bit-identical);
//   waves 4-7 (producers): conv1_1 on the halo of patch p+1 into the other patch buffer.  A producer owns 6 consecutive
//     16-pixel halo tiles and the <= 6 canvas rows under them: it fetches those rows itself as dwords (one patch ahead, three
//     per lane, out-of-image dwords zero by the buffer range rule) into a private LDS strip — no canvas hand-over between
//     waves, ONE workgroup barrier per patch.
// What the stamps of THIS kernel showed (8.1k cycles per patch, 1.6x the first form): while a wave streams independent MFMAs its
// SIMD partner's vector-ALU instructions do not issue at all — the producer got through its scalar, LDS and memory instructions
// and then sat in front of its first VALU instruction until the consumer's 288 MFMAs were done, with or without s_setprio (which
// only decides who waits) and with or without MFMAs of its own.  MFMA time and VALU time of a SIMD simply add up, so the
// kernel is written to need few VALU instructions: every LDS address is a lane constant plus a compile-time offset (eight
// swizzle variants of one base address cover all 36 (tile, tap) fragment positions; the patch-buffer flip is an add on those
// constants, not a second code copy), conversions and bias adds are packed (v_pk_mul/add_f32, one v_cvt_pk_bf16_f32 per pair),
// ReLU is v_pk_max_i16 on the rounded pair, border tests run on border patches only, pooled stores are buffer stores with a
// scalar row offset.  (d16 LDS loads cannot pack two bytes per register here: with SRAM ECC a d16 load clears the other half.)
__global__ __launch_bounds__(512) void conv3p_first2s_kernel(ConvParams p) {
  using G = Geo<5>;
  constexpr int PH = G::PH, PW = G::PW, HW2 = G::HW2, NHALO = G::NHALO;
  constexpr int XB = NHALO * 128;                       // one patch buffer
  constexpr int CVW = 6 * 112;                          // a producer's canvas strip: 6 rows x 28 dwords (byte 2 of a row = halo column 0, channel 0 of its left neighbour)
  constexpr int MI = 4;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) short i16x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const wsm = smem;                      // [9 taps][64 rows][128 B]   conv1_2 weights, gemm2 row/chunk permutation
  unsigned char* const xs = smem + 9 * 8192;            // [2][340][128 B]            conv1_1 output on the halo patch
  unsigned char* const cv = xs + 2 * XB;                // [4][6][112]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ptx = p.W / PW, pty = p.H / PH, npatch = p.B * pty * ptx;
  const bool producer = wave >= 4;

  {   // resident conv1_2 weights: as conv3p_first2_kernel
    const __amdgpu_buffer_rsrc_t rsw = mk_rsrc(p.wgt, (unsigned)((size_t)p.Cout * 576 * 2));
    const int row = wave * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    const int q16 = row & 15;
    const int n = (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned vo = n < p.Cout ? (unsigned)((n * 576 + t * 64 + g * 8) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(wsm + t * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  }
  auto patch_origin = [&](int patch, int& b, int& y0, int& x0) {
    b = patch / (pty * ptx);
    const int trem = patch - b * pty * ptx, ty = trem / ptx;
    y0 = ty * PH; x0 = (trem - ty * ptx) * PW;
  };
  auto on_border = [&](int y0, int x0) { return y0 == 0 || y0 + PH == p.H || x0 == 0 || x0 + PW == p.W; };
#define F2S_STAMP(ph) do { if (p.dbg && blockIdx.x == 0 && (tid & 255) == 0 && it < 24) p.dbg[((tid >> 8) * 24 + it) * 8 + (ph)] = __builtin_readcyclecounter(); } while (0)

  // Everything a lane needs per patch is a constant of the lane (LDS addresses of its fragments / canvas bytes / halo slots)
  // plus a compile-time offset: the two waves of a SIMD share its issue port, so every VALU instruction of either role is
  // matrix-pipe time (stamps: 288 MFMAs + ~1000 VALU instructions of the partner took 9.1k cycles per patch, not 4.6k).
  if (producer) {
    // ================================================================ producers
    __builtin_amdgcn_s_setprio(3);                       // else their few MFMAs wait behind the partner's stream of 288 until it ends
    const int pw = wave & 3;
    const int cvrow0 = (96 * pw) / HW2;                  // first canvas row (of the 12 around the halo) this wave's tiles touch: 0, 2, 5, 8
    const int ntile = pw < 3 ? 6 : 4;                    // 22 halo tiles of 16 pixels
    unsigned char* const cvw = cv + pw * CVW;
    const __amdgpu_buffer_rsrc_t rsc = mk_rsrc(p.in0, (unsigned)((size_t)p.B * p.H * p.W * 3));
    bf16x8 f1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
      f1[jj] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.pre_wgt) + n * 32 + fg * 8);
    }
    f32x2 b1[2][4];                                      // conv1_1 bias of channels 32 t + 8 fg + 2 e2 (+1)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) b1[t][e2] = f32x2{p.pre_bias[32 * t + fg * 8 + 2 * e2], p.pre_bias[32 * t + fg * 8 + 2 * e2 + 1]};
    // gather addresses: tile t3, k = 8 fg + e -> canvas byte (tap row, tap column, channel) of this lane's halo pixel
    unsigned ga[6][8], wo[6][2];
    int h_pr[6], h_pc[6];
#pragma unroll
    for (int t3 = 0; t3 < 6; ++t3) {
      const int pi = (6 * pw + t3) * 16 + fr;
      h_pr[t3] = pi / HW2; h_pc[t3] = pi - h_pr[t3] * HW2;
      const int hoff = pi < NHALO ? (h_pr[t3] - cvrow0) * 112 + h_pc[t3] * 3 + 2 : 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = fg * 8 + e, tp = k / 3;
        const int koff = k < 27 ? (tp / 3) * 112 + (tp % 3) * 3 + (k - tp * 3) : 0;   // k >= 27 multiplies a zero weight column: any byte will do
        ga[t3][e] = (unsigned)(size_t)(lds_ptr)(cvw + hoff + koff);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) wo[t3][t] = (unsigned)(pi * 128 + (((4 * t + fg) ^ (pi & 7)) << 4));
    }
    // canvas strip: lane owns dwords q = lane + 64 k (k < 3, q < 168) = row q / 28, dword q % 28 of the strip; a row starts 8 bytes
    // left of the patch's first pixel (dword aligned: 3 * x0 is a multiple of 96)
    int c_row[3], c_col[3], c_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = lane + 64 * k < 168 ? lane + 64 * k : lane;   // lanes past the strip re-read a dword they own and do not store it
      c_row[k] = q / 28; c_col[k] = q - c_row[k] * 28;
      c_off[k] = (cvrow0 - 2 + c_row[k]) * p.W * 3 - 8 + 4 * c_col[k];
    }
    auto canvas_load = [&](int patch, unsigned (&cb)[3]) {
      int b, y0, x0;
      patch_origin(patch, b, y0, x0);
      if (!on_border(y0, x0)) {                            // every strip dword is inside the image
        const int sbase = ((b * p.H + y0) * p.W + x0) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) cb[k] = __builtin_amdgcn_raw_buffer_load_b32(rsc, (unsigned)(sbase + c_off[k]), 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int y = y0 - 2 + cvrow0 + c_row[k], xb = x0 * 3 - 8 + 4 * c_col[k];
          const bool ok = y >= 0 && y < p.H && xb >= 0 && xb + 4 <= p.W * 3;
          cb[k] = __builtin_amdgcn_raw_buffer_load_b32(rsc, ok ? (unsigned)((b * p.H + y) * p.W * 3 + xb) : 0x80000000u, 0, 0);
        }
      }
    };
    auto canvas_put = [&](const unsigned (&cb)[3]) {
#pragma unroll
      for (int k = 0; k < 3; ++k) if (lane + 64 * k < 168) *reinterpret_cast<unsigned*>(cvw + (lane + 64 * k) * 4) = cb[k];
      asm volatile("" ::: "memory");
    };
    // conv1_1 (+ bias, ReLU; zero outside the image = conv1_2's padding) on this wave's halo tiles of `patch`, strip -> patch buffer
    auto prologue_t = [&](int b, int y0, int x0, int it, auto brd) {   // writes the patch buffer wo[][] points into
      constexpr bool border = decltype(brd)::value;
      unsigned char* const xb = xs;
      unsigned d[2][8];
      // (no d16 pair loads: with SRAM ECC on, a d16 LDS load clears the other register half)
#define F2S_GATHER(t3, dd)                                                                        \
  _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                 \
    const unsigned ad = ga[t3][e];                                                                \
    asm volatile("ds_read_u8 %0, %1" : "=v"(dd[e]) : "v"(ad));                                    \
  }
      F2S_GATHER(0, d[0])
#pragma unroll
      for (int t3 = 0; t3 < 6; ++t3) {
        if (t3 < ntile) {                                  // wave-uniform
          unsigned (&dc)[8] = d[t3 & 1];
          if (t3 + 1 < 6 && t3 + 1 < ntile) {
            F2S_GATHER(t3 + 1, d[(t3 + 1) & 1])
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(dc[0]), "+v"(dc[1]), "+v"(dc[2]), "+v"(dc[3]), "+v"(dc[4]), "+v"(dc[5]), "+v"(dc[6]), "+v"(dc[7]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dc[0]), "+v"(dc[1]), "+v"(dc[2]), "+v"(dc[3]), "+v"(dc[4]), "+v"(dc[5]), "+v"(dc[6]), "+v"(dc[7]));
          }
          if (t3 == 0) F2S_STAMP(3);
          bf16x8 fx;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {                 // == bf16(v / 255.0f) for every byte value
            const f32x2 v = f32x2{(float)dc[2 * e2], (float)dc[2 * e2 + 1]} * f32x2{0.00392156862745098f, 0.00392156862745098f};
            const bf16x2 r = __builtin_convertvector(v, bf16x2);
            fx[2 * e2] = r[0]; fx[2 * e2 + 1] = r[1];
          }
          f32x4 a1[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) a1[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[jj], fx, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          const int pi = (6 * pw + t3) * 16 + fr;
          bool inside = true;
          if (border) {
            const int y = y0 - 1 + h_pr[t3], x = x0 - 1 + h_pc[t3];
            inside = y >= 0 && y < p.H && x >= 0 && x < p.W;
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            union { i16x2 h[4]; u32x4v u; } o;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {               // + bias, round, ReLU on the rounded pair (sign test: bf16 >= 0 iff int16 >= 0)
              const f32x4& av = a1[2 * t + (e2 >> 1)];
              const f32x2 v = f32x2{av[2 * (e2 & 1)], av[2 * (e2 & 1) + 1]} + b1[t][e2];
              const bf16x2 r = __builtin_convertvector(v, bf16x2);
              o.h[e2] = __builtin_elementwise_max(__builtin_bit_cast(i16x2, r), i16x2{0, 0});
            }
            if (border && !inside) o.u = u32x4v{0u, 0u, 0u, 0u};
            if (pi < NHALO) *reinterpret_cast<u32x4v*>(xb + wo[t3][t]) = o.u;
          }
          if (t3 == 0) F2S_STAMP(4);
          if (t3 == 2) F2S_STAMP(5);
        }
      }
    };
    auto prologue = [&](int patch, int it) {
      int b, y0, x0;
      patch_origin(patch, b, y0, x0);
      if (on_border(y0, x0)) prologue_t(b, y0, x0, it, std::true_type{});
      else prologue_t(b, y0, x0, it, std::false_type{});
    };
    auto flip = [&](unsigned delta) {                      // the other patch buffer (one code copy for both: the kernel's hot code must stay in the 64 KB I-cache)
#pragma unroll
      for (int t3 = 0; t3 < 6; ++t3) { wo[t3][0] += delta; wo[t3][1] += delta; }
    };

    int patch = blockIdx.x;
    unsigned cb[3];
    canvas_load(patch, cb);
    canvas_put(cb);
    prologue(patch, 1 << 20);
    if (patch + (int)gridDim.x < npatch) canvas_load(patch + gridDim.x, cb);
    for (int it = 0; patch < npatch; patch += gridDim.x, ++it) {
      const int nextp = patch + gridDim.x;
      F2S_STAMP(0);
      __syncthreads();                                   // patch buffer it & 1 complete; the other one no longer read
      F2S_STAMP(1);
      if (nextp < npatch) {
        canvas_put(cb);
        if (nextp + (int)gridDim.x < npatch) canvas_load(nextp + gridDim.x, cb);   // in flight for a whole patch
        F2S_STAMP(2);
        flip((it & 1) ? (unsigned)-XB : (unsigned)XB);
        prologue(nextp, it);
      }
      F2S_STAMP(6);
    }
  } else {
    // ================================================================ consumers
    const int wm = wave;
    f32x2 bv[2][4];                                        // conv1_2 bias: channels 32 h + 8 fg + 2 e2 (+1)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const int n = 32 * h + fg * 8 + 2 * e2;
        bv[h][e2] = f32x2{(p.bias && n < p.Cout) ? p.bias[n] : 0.f, (p.bias && n + 1 < p.Cout) ? p.bias[n + 1] : 0.f};
      }
    // pixel fragments: tile i = patch row 2 wm + (i >> 1), columns 16 (i & 1) + fr; tap (dy, dx) reads halo slot
    // pi = base + 16 (i & 1) + 34 ((i >> 1) + dy) + dx, chunk (4 kk + fg) ^ (pi & 7), and pi & 7 = (base + 2 ((i >> 1) + dy) + dx) & 7:
    // eight lane constants (one per value of that sum mod 8) + compile-time offsets address every fragment of the patch
    const int pib = 2 * wm * HW2 + fr;
    const unsigned char* xbase[2][8];
#pragma unroll
    for (int sg = 0; sg < 8; ++sg)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xbase[kk][sg] = xs + pib * 128 + ((((4 * kk + fg) ^ ((pib + sg) & 7))) << 4);
    const int wfl = (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4);
    const unsigned char* wbase[2][2];                      // [kk][taps 0-4 | 5-8] (the offset field holds 16 bits)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { wbase[kk][0] = wsm + (wfl ^ (kk * 64)); wbase[kk][1] = wsm + 5 * 8192 + (wfl ^ (kk * 64)); }
    // pooled output: lane stores (even fr only) pixel column (16 i + fr) / 2 of pooled row y0 / 2 + wm, channels 32 h + 8 fg ..
    const __amdgpu_buffer_rsrc_t rso = mk_rsrc(p.out_pool, p.out_pool ? (unsigned)((size_t)p.B * (p.H >> 1) * (p.W >> 1) * p.out_ld * 2) : 0u);
    const unsigned so_lane = ((fr & 1) == 0 && fg * 8 < p.Cout) ? (unsigned)(((fr >> 1) * p.out_ld + fg * 8) * 2) : 0x80000000u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the resident weights has landed

    auto body = [&](int patch, int it) {
      int b, y0, x0;
      patch_origin(patch, b, y0, x0);
      F2S_STAMP(0);
      __syncthreads();
      F2S_STAMP(1);
      f32x4 acc[4][MI];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 fw[2][4], fx[2][MI];
      auto ldfrag = [&](auto sc, bf16x8 (&w)[4], bf16x8 (&x)[MI]) {   // K step s = 2 * tap + half
        constexpr int s = decltype(sc)::value, t = s >> 1, kk = s & 1, dy = t / 3, dx = t % 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const bf16x8*>(wbase[kk][t >= 5] + (t >= 5 ? t - 5 : t) * 8192 + j * 2048);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          constexpr int dummy = 0; (void)dummy;
          const int rho = (i >> 1) + dy;
          x[i] = *reinterpret_cast<const bf16x8*>(xbase[kk][(2 * rho + dx) & 7] + (16 * (i & 1) + HW2 * rho + dx) * 128);
        }
      };
      ldfrag(std::integral_constant<int, 0>{}, fw[0], fx[0]);
      auto kstep = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if constexpr (s + 1 < 18) ldfrag(std::integral_constant<int, s + 1>{}, fw[(s + 1) & 1], fx[(s + 1) & 1]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s & 1][j], fx[s & 1][i], acc[j][i], 0, 0, 0);
        // issue order of the step: one fragment read of step s + 1 in front of every two MFMAs of step s (the compiler's own
        // order puts the reads a few MFMAs ahead of their use and waits on them)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      kstep(std::integral_constant<int, 0>{}); kstep(std::integral_constant<int, 1>{}); kstep(std::integral_constant<int, 2>{});
      kstep(std::integral_constant<int, 3>{}); kstep(std::integral_constant<int, 4>{}); kstep(std::integral_constant<int, 5>{});
      kstep(std::integral_constant<int, 6>{}); kstep(std::integral_constant<int, 7>{}); kstep(std::integral_constant<int, 8>{});
      kstep(std::integral_constant<int, 9>{}); kstep(std::integral_constant<int, 10>{}); kstep(std::integral_constant<int, 11>{});
      kstep(std::integral_constant<int, 12>{}); kstep(std::integral_constant<int, 13>{}); kstep(std::integral_constant<int, 14>{});
      kstep(std::integral_constant<int, 15>{}); kstep(std::integral_constant<int, 16>{}); kstep(std::integral_constant<int, 17>{});
      F2S_STAMP(2);
      // ---- epilogue: lane holds, of patch pixel (2 wm + (i >> 1), 16 (i & 1) + fr), channels 32 h + 8 fg + {0..3} (j = 2 h) and + {4..7} (j = 2 h + 1)
      if (p.out) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int r = wm * 64 + i * 16 + fr;
          const int64_t m = ((int64_t)b * p.H + y0 + (r >> 5)) * p.W + x0 + (r & 31);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = 32 * h + fg * 8;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(acc[2 * h + (e >> 2)][i][e & 3] + bv[h][e >> 1][e & 1], 0.f);
            if (n < p.Cout) st_out(reinterpret_cast<bf16*>(p.out) + m * p.out_ld + n, o, p.store_policy);
          }
        }
      }
      if (p.out_pool) {
        // max over the 2 x 2 window first (vertical partner = tile i + 2 in this lane, horizontal partner = lane ^ 1 by DPP), then
        // + bias, round to bf16, ReLU on the rounded value: each step is monotone, so this equals pooling relu(x + b) bit for bit
        const unsigned srow = (unsigned)((((b * (p.H >> 1) + (y0 >> 1) + wm) * (p.W >> 1) + (x0 >> 1)) * p.out_ld) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            union { i16x2 hh[4]; u32x4v u; } o;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
              float m2[2];
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                const int e = 2 * e2 + c, j = 2 * h + (e >> 2);
                const float v = fmaxf(acc[j][i][e & 3], acc[j][i + 2][e & 3]);
                asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(m2[c]) : "v"(v));
              }
              const f32x2 s2 = f32x2{m2[0], m2[1]} + bv[h][e2];
              const bf16x2 r = __builtin_convertvector(s2, bf16x2);
              o.hh[e2] = __builtin_elementwise_max(__builtin_bit_cast(i16x2, r), i16x2{0, 0});
            }
            const unsigned soff = srow + (unsigned)((i * 8 * p.out_ld + 32 * h) * 2);
            if (32 * h < p.Cout) {
              if (p.store_policy == 1 || p.store_policy == 2) __builtin_amdgcn_raw_buffer_store_b128(o.u, rso, so_lane, soff, 2);
              else __builtin_amdgcn_raw_buffer_store_b128(o.u, rso, so_lane, soff, 0);
            }
          }
      }
      F2S_STAMP(3);
    };
    int patch = blockIdx.x;
    for (int it = 0; patch < npatch; patch += gridDim.x, ++it) {
      body(patch, it);
      const int delta = (it & 1) ? -XB : XB;               // the other patch buffer next time
#pragma unroll
      for (int sg = 0; sg < 8; ++sg) { xbase[0][sg] += delta; xbase[1][sg] += delta; }
    }
  }
#undef F2S_STAMP
#undef F2S_GATHER
}

static int g_first_persistent = 2;   // fused conv1_1 + conv1_2: 2 = wave-specialised persistent kernel, 1 = first persistent form, 0 = per-patch FIRST variant
static unsigned long long* g_c3_dbg = nullptr;
void set_conv3p_stamps(unsigned long long* d) { g_c3_dbg = d; }

static void launch_first2(const ConvParams& p_in, hipStream_t s) {
  ConvParams p = p_in;
  p.dbg = g_c3_dbg;
  using G = Geo<5>;
  constexpr int lds = 9 * 8192 + 2 * G::NHALO * 128 + 1408;   // 162,176 B of the 163,840
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)conv3p_first2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); });
  const int cus = device_cu_count(256);
  const int npatch = p.B * (p.H / G::PH) * (p.W / G::PW);
  if (g_first_persistent == 2 && ((uintptr_t)p.in0 & 3) == 0 && (size_t)p.B * p.H * p.W * 3 < ((size_t)1 << 31)) {
    constexpr int lds2 = 9 * 8192 + 2 * G::NHALO * 128 + 4 * 6 * 112;   // 163,456 B of the 163,840
    static PerDeviceOnce once2;
  once2.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)conv3p_first2s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds2)); });
    hipLaunchKernelGGL(conv3p_first2s_kernel, dim3(std::min(npatch, cus)), dim3(512), lds2, s, p);
    return;
  }
  hipLaunchKernelGGL(conv3p_first2_kernel, dim3(std::min(npatch, cus)), dim3(512), lds, s, p);
}

template <int BN, int WM, int WN, bool FIRST = false, int XS = 2, int LPW = 5, int NP = 0, int NWS = 2>
static void launch_c3(const ConvParams& p_in, hipStream_t s) {
  const ConvParams p = with_range_ctx(p_in);
  using C = C3<BN, WM, WN, LPW>;
  using G = Geo<LPW>;
  const int tilesM = p.B * (p.H / G::PH) * (p.W / G::PW), tilesN = (p.Cout + BN - 1) / BN;
  constexpr int lds = XS * G::XSTAGE + NWS * C::WSTAGE + (FIRST ? (NP ? 2560 : 2048) : 0);   // (the canvas patch and the u8 tables of the fused first layer)
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)conv3p_kernel<BN, WM, WN, FIRST, XS, LPW, NP, NWS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); });
  hipLaunchKernelGGL((conv3p_kernel<BN, WM, WN, FIRST, XS, LPW, NP, NWS>), dim3(tilesM * tilesN), dim3(C::NT), lds, s, p);
}

void set_conv3p_first_persistent(int v) { g_first_persistent = v; }
static int g_c32_tile = 1;          // Cout <= 32: 32-wide tiles, 4 waves of 64 pixels x 32 channels (0: the 64-wide tile)
void set_conv3p_c32_tile(int v) { g_c32_tile = v; }
static int g_narrow_bn64 = 2;       // 16 x 16-patch maps with Cout > 64 on 64-wide tiles: 0 never, 1 always, 2 when the 128-wide tiles do not fill the chip once
void set_conv3p_narrow_bn64(int v) { g_narrow_bn64 = v; }
// split pairs, Cout > 64 tiles on 8 x 32 patches: 4 waves (wave tile 64 x 128, static addresses, 256 registers: the default) or 8 (wave tile 64 x 64, runtime-tap loop, 128
// registers).  Same K order, same sums: bit-identical heat maps.  Measured per 8-page group (the eight layers on these tiles, same box, twice): 7.94 / 8.00 -> 7.47 / 7.54 ms.
// The eight-wave loop spends ~33 vector and ~40 scalar instructions per tap on fragment addresses, in all four waves of a SIMD (vector instructions and MFMAs of a
// SIMD share an issue port: conv3p_first2s_kernel's stamps); the four-wave loop has 47 vector instructions per NINE taps (576 MFMAs).  Counters per MFMA, whole
// kernel (profiles/r04_pmc_stall_craft.txt): vector 1.23 -> 0.49, scalar 2.03 -> 0.38, LDS 0.51 -> 0.385; MFMA busy 0.65 at 1.68 GHz -> 0.69 at 1.63.
// Requesting each pixel fragment one block of eight MFMAs ahead changes nothing on top (7.47 / 7.54): not a latency.
static int g_c128_waves = 4;
void set_conv3p_c128_waves(int w) { g_c128_waves = w; }
static int g_c64_waves = 8;        // Cout <= 64 tiles: 8 waves (wave tile 64x32) or 4 waves (wave tile 64x64, fewer LDS fragment reads per MFMA)
void set_conv3p_c64_waves(int w) { g_c64_waves = w; }
static int g_force_bn128 = 1;     // BN = 128 single-stage tiles, two workgroups per CU, for every Cout > 64 (0: BN = 256, one per CU, for Cout % 256 == 0):
                                  // measured +5..17 % on CRAFT's 256/512-channel layers (1.19-1.40 PFLOP/s)
void set_conv3p_force_bn128(int v) { g_force_bn128 = v; }
static int g_xs1_max_cin = 1 << 20;   // layers with Cout <= 128 and Cin up to this use one patch stage and two workgroups per CU
void set_conv3p_single_stage_max_cin(int c) { g_xs1_max_cin = c; }

const char* conv3p_check(const ConvParams& p) {
  if (p.split) {   // f16 planes in; fp32 or planes out
    if (p.ks != 3 || p.dil != 1 || p.C1 || p.relu0 || p.relu1) return "conv3p: 3x3, dilation 1, single source";
    if (p.pre_wgt && (p.split != 3 || p.C0 != 64 || p.Cout != 64 || !p.pre_bias || !(p.pre_scale > 0.f) || p.H % 8 || p.W % 32 || ((uintptr_t)p.pre_wgt & 15) || (size_t)p.M * 3 >= ((size_t)1 << 31)))
      return "conv3p: the fused first layer (split) is conv1_1 in front of conv1_2 on pairs, 8 x 32 patches";
    if (p.C0 % 64 || p.Cout % 8) return "conv3p: Cin % 64, Cout % 8";
    if (!((p.H % 8 == 0 && p.W % 32 == 0) || (p.H % 16 == 0 && p.W % 16 == 0))) return "conv3p: the map must tile into 8x32 or 16x16 patches";
    if (p.resid || p.out_f32 || p.act == kActGelu) return "conv3p: conv epilogues only";
    const int ov = p.out_planes ? 8 : 4;
    if (p.out && (p.out_ld % ov || ((uintptr_t)p.out & 15))) return "conv3p: output alignment";
    if (p.out_relu && (!p.out || ((uintptr_t)p.out_relu & 15))) return "conv3p: out_relu alignment";
    if (p.out_pool && (p.out_ld % ov || ((uintptr_t)p.out_pool & 15) || (p.H | p.W) & 1)) return "conv3p: out_pool alignment";
    if (!p.out && !p.out_pool && !(p.split == 2 && p.tail_heat)) return "conv3p: no output";
    if (p.tail_heat && (p.split != 2 || p.Cout != 32 || !p.tail_w6 || !p.tail_w8 || !p.tail_b6 || !p.tail_b8 || (((uintptr_t)p.tail_w6 | (uintptr_t)p.tail_w8 | (uintptr_t)p.tail_b6) & 15)))
      return "conv3p: the fused head tail belongs to conv_cls.4 on packed pairs (32 output channels)";
    if ((p.bias && ((uintptr_t)p.bias & 15)) || (!p.pre_wgt && ((uintptr_t)p.in0 & 15)) || ((uintptr_t)p.wgt & 15)) return "conv3p: operand alignment";
    const size_t lim = (size_t)1 << 31;
    if (p.split != 2 && p.split != 3 && p.split != 4) return "conv3p: split must be 2 (packed pairs), 3 or 4";
    if (p.split == 2 && (p.C0 != 64 || p.Cout > 32)) return "conv3p: packed pairs are the 32-channel layers' form (64 halves per pixel, Cout <= 32)";
    if (p.out_planes != 0 && p.out_planes != 2 && p.out_planes != 3) return "conv3p: out_planes must be 0, 2 or 3";
    if ((size_t)p.M * p.C0 * (p.split == 4 ? 6 : p.split == 3 ? 4 : 2) >= lim || (size_t)p.Cout * 27 * p.C0 * 2 >= lim) return "conv3p: tensor too large for 32-bit buffer offsets";
    if (p.M != p.B * p.H * p.W || p.M <= 0 || !(p.out_scale > 0.f)) return "conv3p: bad shape";
    return nullptr;
  }
  if (p.ks != 3 || p.dil != 1) return "conv3p: 3x3, dilation 1 only";
  if (p.C1 || p.relu0 || p.relu1) return "conv3p: single source, no ReLU on load";
  if (p.C0 % 64 || p.Cout % 8) return "conv3p: Cin % 64, Cout % 8";
  if (!((p.H % 8 == 0 && p.W % 32 == 0) || (p.H % 16 == 0 && p.W % 16 == 0))) return "conv3p: the map must tile into 8x32 or 16x16 patches";
  if (p.resid || p.out_f32 || p.act == kActGelu) return "conv3p: conv epilogues only";
  if (p.out && (p.out_ld % 8 || ((uintptr_t)p.out & 15))) return "conv3p: output alignment";
  if (p.out_relu && (!p.out || ((uintptr_t)p.out_relu & 15))) return "conv3p: out_relu alignment";
  if (p.out_pool && (p.out_ld % 8 || ((uintptr_t)p.out_pool & 15))) return "conv3p: out_pool alignment";
  if (!p.out && !p.out_pool) return "conv3p: no output";
  if (p.bias && ((uintptr_t)p.bias & 15)) return "conv3p: bias alignment";
  if ((!p.pre_wgt && ((uintptr_t)p.in0 & 15)) || ((uintptr_t)p.wgt & 15) || ((uintptr_t)p.pre_wgt & 15)) return "conv3p: operand alignment";
  const size_t lim = (size_t)1 << 31;
  // (the fused first pair reads the u8 canvas - 3 bytes per pixel - and never forms its 64-channel input)
  const size_t in_bytes = p.pre_wgt ? (size_t)p.M * 3 : (size_t)p.M * p.C0 * 2;
  const size_t out_bytes = (size_t)p.M * p.Cout * 2 / (p.out ? 1 : 4);
  if (in_bytes >= lim || out_bytes >= lim || (size_t)p.Cout * 9 * p.C0 * 2 >= lim) return "conv3p: tensor too large for 32-bit buffer offsets";
  if (p.M != p.B * p.H * p.W || p.M <= 0) return "conv3p: bad shape";
  return nullptr;
}

// tile width (output channels per workgroup) launch_conv3p picks for a split-operand layer: 128, 64 or 32 (the profile's kernel kinds)
static int g_narrow_wide = 1;    // 8 x 32-patch layers too take 64-wide tiles when 128-wide ones would leave workgroup slots empty (0: round 3's rule, 16 x 16-patch layers only)
static int g_narrow_frac = 8;    // ... "empty" = fewer tiles than g_narrow_frac / 4 per CU (8: two per CU, the kernels' residency)
void set_conv3p_narrow_wide(int v) { g_narrow_wide = v; }
void set_conv3p_narrow_frac(int v) { g_narrow_frac = v < 1 ? 1 : v; }
static int g_deep_w = 1;         // the 32-wide tiles of a single page's deep layers on four weight stages (conv3p_kernel: NWS)
void set_conv3p_deep_w(int v) { g_deep_w = v; }
static int g_deep_w64 = 1 << 20; // ... and the 64-wide tiles while they number fewer than this many per CU (0 = never; default: always - the same two workgroups per CU,
                                 // the deep layers of an 8-page group 262 -> 235 us, a single page's 64-wide layers -7 %, the full-resolution ones unchanged)
void set_conv3p_deep_w64(int v) { g_deep_w64 = v; }
static int g_narrowest_frac = 4; // 32-wide tiles (four waves) when even the 64-wide ones number fewer than g_narrowest_frac / 4 per CU: the deep layers of a single page
void set_conv3p_narrowest_frac(int v) { g_narrowest_frac = v < 0 ? 0 : v; }
// (the K loop, and with it every sum's order, is the same on every tile width: a page's result does not depend on the batch it came in)
static bool c3_narrowest(const ConvParams& p, bool wide) {
  const int tiles64 = p.B * (wide ? (p.H / 8) * (p.W / 32) : (p.H / 16) * (p.W / 16)) * ((p.Cout + 63) / 64);
  return p.Cout > 32 && tiles64 < g_narrowest_frac * device_cu_count(256) / 4;
}
int conv3p_split_bn(const ConvParams& p) {
  const bool wide = p.H % 8 == 0 && p.W % 32 == 0;
  if (c3_narrowest(p, wide)) return 32;
  if (p.Cout <= 64) return (wide && p.Cout <= 32) ? 32 : 64;
  const int tiles128 = p.B * (wide ? (p.H / 8) * (p.W / 32) : (p.H / 16) * (p.W / 16)) * ((p.Cout + 127) / 128);
  const bool narrow = tiles128 < g_narrow_frac * device_cu_count(256) / 4 && (!wide || g_narrow_wide);
  return narrow ? 64 : 128;
}

static int g_head_persistent = 1;   // packed-pairs head layers on conv3h.hip's persistent kernel (tuning key head_persistent; 0: this file's per-patch tile)
void set_conv3p_head_persistent(int v) { g_head_persistent = v; }
void launch_conv3p(const ConvParams& p, hipStream_t s) {
  if (const char* e = conv3p_check(p)) throw std::runtime_error(e);
  if (p.split) {   // the one-patch-stage tiles (two workgroups per CU)
    const bool wide = p.H % 8 == 0 && p.W % 32 == 0;
    if (p.pre_wgt) return launch_c3<64, 4, 2, true, 1, 5, 3, 4>(p, s);   // conv1_1 fused in front of conv1_2 (conv3p_check: pairs, 64 -> 64 channels, 8 x 32 patches)
    const int tiles128 = p.B * (wide ? (p.H / 8) * (p.W / 32) : (p.H / 16) * (p.W / 16)) * ((p.Cout + 127) / 128);
    const bool narrow = tiles128 < g_narrow_frac * device_cu_count(256) / 4 && (!wide || g_narrow_wide);   // fewer 128-wide tiles than workgroup slots: 64-wide tiles fill the chip (a single page)
    const bool narrowest = c3_narrowest(p, wide);
#define TTR_C3_SPLIT(NPV)                                                                               \
    if (narrowest) return wide ? launch_c3<32, 4, 1, false, 1, 5, NPV>(p, s) : launch_c3<32, 4, 1, false, 1, 4, NPV>(p, s);   \
    if (p.Cout <= 64) {                                                                                 \
      if (!wide) return launch_c3<64, 4, 2, false, 1, 4, NPV>(p, s);                                    \
      if (p.Cout <= 32) return launch_c3<32, 4, 1, false, 1, 5, NPV>(p, s);                             \
      return launch_c3<64, 4, 2, false, 1, 5, NPV>(p, s);                                               \
    }                                                                                                   \
    if (!wide) return narrow ? launch_c3<64, 4, 2, false, 1, 4, NPV>(p, s) : launch_c3<128, 4, 2, false, 1, 4, NPV>(p, s);   \
    return narrow ? launch_c3<64, 4, 2, false, 1, 5, NPV>(p, s) : launch_c3<128, 4, 2, false, 1, 5, NPV>(p, s);
    if (p.split == 2) {   // packed pairs (the 32-channel head layers): one 128-byte row [x0 (32) | x1 (32)] per pixel, two virtual chunks
      if (!wide) return launch_c3<64, 4, 2, false, 1, 4, 2>(p, s);
      if (g_head_persistent && conv3h_eligible(p)) return launch_conv3h(p, s);   // the persistent form (conv3h.hip): weights resident, one burst per patch; bit-identical
      return launch_c3<32, 4, 1, false, 1, 5, 2>(p, s);
    }
    if (p.split == 3) {
      // pairs (the default): the tiles of a page or two - workgroups (nearly) alone on their CUs - request their weights further ahead (NWS)
      const int tiles64 = p.B * (wide ? (p.H / 8) * (p.W / 32) : (p.H / 16) * (p.W / 16)) * ((p.Cout + 63) / 64);
      if (narrowest && g_deep_w) return wide ? launch_c3<32, 4, 1, false, 1, 5, 3, 4>(p, s) : launch_c3<32, 4, 1, false, 1, 4, 3, 4>(p, s);
      // (the 64-wide tiles on four waves of 64 x 64 instead of eight of 64 x 32 - both on static addresses: 1525 against 1457 us per 8-page group: eight stay)
      if (!narrowest && g_deep_w64 && p.Cout >= 64 && (p.Cout <= 64 || narrow) && tiles64 < g_deep_w64 * device_cu_count(256))
        return wide ? launch_c3<64, 4, 2, false, 1, 5, 3, 4>(p, s) : launch_c3<64, 4, 2, false, 1, 4, 3, 4>(p, s);
      if (g_c128_waves == 4 && wide && p.Cout > 64 && !narrow) return launch_c3<128, 4, 1, false, 1, 5, 3>(p, s);
      TTR_C3_SPLIT(3)
    }
    TTR_C3_SPLIT(4)
#undef TTR_C3_SPLIT
  }
  if (p.pre_wgt) {
    if (p.C0 != 64 || p.Cout > 64 || !p.pre_bias) throw std::runtime_error("conv3p: the fused first layer needs Cin = 64, Cout <= 64");
    if (g_first_persistent && p.act == kActRelu && !p.out_relu && p.H % 8 == 0 && p.W % 32 == 0) return launch_first2(p, s);
    return launch_c3<64, 4, 2, true, 1>(p, s);
  }
  const bool wide = p.H % 8 == 0 && p.W % 32 == 0;   // 8 x 32 patches; else 16 x 16 (checked above)
  if (p.Cout <= 64) {
    if (!wide) return launch_c3<64, 4, 2, false, 1, 4>(p, s);
    if (p.Cout <= 32 && g_c32_tile && p.C0 <= g_xs1_max_cin) return launch_c3<32, 4, 1, false, 1>(p, s);   // upconv4.3: a 64-wide tile multiplies 32 rows of zero weights
    if (g_c64_waves == 4) return launch_c3<64, 4, 1, false, 1>(p, s);
    return p.C0 <= g_xs1_max_cin ? launch_c3<64, 4, 2, false, 1>(p, s) : launch_c3<64, 4, 2>(p, s);
  }
  if (!wide) {
    // fewer 128-wide tiles than the chip holds at once (two workgroups per CU): 64-wide tiles fill it better (upconv1.3 at 16 pages:
    // 384 tiles -> 768, 112 -> 98 us; the layers with a full round of 128-wide tiles lose 0-2 % on the narrower tile)
    const int tiles128 = p.B * (p.H / 16) * (p.W / 16) * ((p.Cout + 127) / 128);
    const bool narrow = g_narrow_bn64 == 1 || (g_narrow_bn64 == 2 && tiles128 < 2 * device_cu_count(256));
    return narrow ? launch_c3<64, 4, 2, false, 1, 4>(p, s) : launch_c3<128, 4, 2, false, 1, 4>(p, s);
  }
  if (p.Cout <= 128 || p.Cout % 256 || g_force_bn128) return p.C0 <= g_xs1_max_cin ? launch_c3<128, 4, 2, false, 1>(p, s) : launch_c3<128, 4, 2>(p, s);
  return launch_c3<256, 2, 4>(p, s);
}

}  // namespace ttr
