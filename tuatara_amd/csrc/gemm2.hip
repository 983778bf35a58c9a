// bf16 implicit-GEMM convolution / linear kernel, second generation (gfx950 / MI355X).
//
//   out[m][n] = act( sum_k X[m][k] * Wt[n][k] + bias[n] (+ resid) )
//
// Same contract as igemm.hip (ConvParams; replaces the LibTorch conv/linear calls inside the
// TorchScript modules run at tuatara.cpp:376 and tuatara.cpp:307) but built around the
// CDNA4 LDS-DMA path:
//   * both operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging,
//     no ds_write); the conv halo, ragged M and ragged Cout are zero-filled by the buffer
//     resource's out-of-range rule (voffset 0x80000000), so there is no branch in the loader;
//   * BK = 64 (one full 128-byte line per tile row), two LDS stages, ONE barrier per K step:
//     tile t+1 streams in while tile t is multiplied;
//   * the LDS image is lane-linear per wave instruction (8 rows x 128 B); bank conflicts are
//     removed by permuting the 16-byte chunks of each row on the *source* side
//     (chunk ^= (row>>1)&7) and applying the same XOR on the ds_read_b128 side;
//   * the MFMA is issued transposed (A = weights, B = activations) so that a lane ends up
//     holding 4 consecutive output channels of one pixel; weight rows are permuted while
//     staging so two adjacent MFMA tiles give 8 consecutive channels -> one 16-byte store;
//   * XCD-aware tile order (bijective remap): the N tiles of an M tile and neighbouring M
//     tiles run on one XCD and share its L2.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <type_traits>
#include <vector>

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

typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int BM, int BN, int WM, int WN, int XST = 2, int WST = 2>
struct G2Cfg {
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = BM / WM, TN = BN / WN;   // wave tile
  static constexpr int MI = TM / 16, NJ = TN / 16;
  static constexpr int XP = BM / 8, WP = BN / 8;     // 1-KiB pieces (8 rows x 128 B) per operand tile
  static constexpr int XPW = XP / NW, WPW = WP / NW;  // pieces per wave
  static constexpr int XBYTES = BM * 128, WBYTES = BN * 128;   // one stage of each operand
  static constexpr int LDS = XST * XBYTES + WST * WBYTES;      // X ring of XST stages, W ring of WST (2; 3: split mode, w0b formed in registers)
  static_assert(XP % NW == 0 && WP % NW == 0, "tile pieces must divide over the waves");
  static_assert(TN % 32 == 0 && TM % 16 == 0, "wave tile");
};

// XST = 3: the activation tiles run TWO K steps ahead of the MFMAs (the weight tiles one).  Activations are streamed once, so
// every X tile is a first-touch HBM miss (~2 us); one K step of MFMA work is shorter than that and the one-step-ahead form
// spent 37-51 % of its wave cycles parked on vmcnt (SQ_WAIT_ANY).  Weights stay L2 resident and need only one step.
//
// SP (split-operand mode, split.h): the operands are f16 planes - activation rows [x0 | x1 | x2] of 3 C halves per pixel, weight
// rows [w0 | w0/2^11 | w1] of 3 K - and the K loop runs over K' = 4 K (quarter q: activation plane {0,1,2,0}, weight plane {0,1,1,2}); the
// accumulator times p.out_scale is the fp32 product.  Every output (out, out_relu, out_pool, out_f32) is fp32 then, or, with
// p.out_planes, out / out_relu / out_pool are written as the three planes of the value (row stride 3 out_ld halves).
// NP = products per value: 0 = the bf16 kernel, 4 = exact activation triples (planes x0 | x1 | x2), 3 = activation pairs (x0 | x1).
// WST = 3 (split mode, big tiles): the scaled weight copy w0b = w0 / 2^11 is not staged - the phases that need it read the W0 tile and scale
// the fragments with packed f16 multiplies (exact) - so a k0 costs X0 X1 [X2] W0 W1: 4 (5) tile loads for 3 (4) products instead of 5 (6).
template <int BM, int BN, int WM, int WN, int MINB, int XST, int NP, int WST = 2>
__global__ __launch_bounds__(WM * WN * 64, MINB) void gemm2_kernel(ConvParams p) {
  using C = G2Cfg<BM, BN, WM, WN, XST, WST>;
  static_assert(WST == 2 || (WST == 3 && NP != 0 && XST == 3), "three W slots: the split mode's reuse-order loop only");
  constexpr bool WR = WST == 3;     // w0b formed in registers
  constexpr bool SP = NP != 0;
  static_assert(NP == 0 || NP == 4 || (NP == 3 && XST == 3), "pairs run in the reuse-order loop only");
  if (p.skip && __builtin_nontemporal_load(p.skip) >= p.skip_n) return;   // AR early exit (ConvParams::skip): uniform, before any barrier
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // GELU epilogue: Phi(x) by linear interpolation in an 8 KiB table kept in LDS behind the operand stages (the
  // erf polynomial + exp + rcp form made the epilogue, not the MFMAs, the longest part of the K = 384 PARSeq GEMMs)
  float2* const glut = reinterpret_cast<float2*>(smem + C::LDS);
  if (p.act == kActGelu && p.gelu_lut) {   // (split mode: the cubic table of gelu_hermite(), 512 x float4)
    for (int i = tid; i < 512; i += C::NT) reinterpret_cast<uint4*>(glut)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  }   // visible after the first K-step barrier

  // ---- persistent, XCD-aware tile schedule.  Workgroups bid and bid+8 share an XCD (round-robin placement, speed only);
  // XCD x owns a contiguous run of the tile list (N tiles of one M tile adjacent) and its J = gridDim/8 workgroups walk
  // that run with stride J.  While a tile's last K step computes, the first stage of the workgroup's NEXT tile is already
  // streaming in, so the epilogue and the next tile's load latency overlap (the K = 384 PARSeq GEMMs have only 6 K steps).
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const int T = tilesM * tilesN;
  const int xcd = blockIdx.x & 7, J = gridDim.x >> 3;
  int xcd_first, xcd_count;
  {
    const int q = T >> 3, r = T & 7;
    xcd_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xcd_count = q + (xcd < r ? 1 : 0);
  }
  int idx = blockIdx.x >> 3;
  if (idx >= xcd_count) return;

  const int Ctot = p.C0 + p.C1;
  const int K = p.ks * p.ks * Ctot;
  constexpr int PL = NP == 4 ? 3 : NP == 3 ? 2 : 1;   // activation planes per pixel
  const int KP = NP == 4 ? 4 * K : K;            // K as the plane-major loop sees it
  const int nk = KP >> 6;
  const int HW = p.H * p.W;

  const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(p.in0, (unsigned)((size_t)p.M * p.C0 * 2 * PL));
  const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.C1 ? p.in1 : p.in0, (unsigned)((size_t)p.M * (p.C1 ? p.C1 : p.C0) * 2 * PL));
  const int KW = SP ? 3 * K : K;                 // weight row length
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.wgt, (unsigned)((size_t)p.Cout * KW * 2));
  constexpr unsigned OOB = 0x80000000u;

  // Tile row -> pixel.  Normally the identity (tile rows are consecutive pixels).  With a fused 2x2 max-pool the
  // tile is BM/4 consecutive *pooled* pixels: row r is sub-pixel (dy, dx) = (r>>1 & 1, r & 1) of pooled pixel r>>2,
  // so the four partners of a pool window sit in four neighbouring lanes of one MFMA tile.
  auto row_to_pixel = [&](int grow) -> int {
    if (p.up_2d) {   // (ConvParams::up_2d: tile t = block (ty, tx) of image b, row r = pixel (r >> 4, r & 15) of the block; launcher: W % 16 == 0, H % (BM / 16) == 0)
      constexpr int BH = BM / 16;
      const int t = grow / BM, r = grow - t * BM;
      const int bw = p.W >> 4, bh = p.H / BH;
      const int tx = t % bw, tq = t / bw, ty = tq % bh, b = tq / bh;
      return (b * p.H + ty * BH + (r >> 4)) * p.W + tx * 16 + (r & 15);
    }
    if (!p.out_pool) return grow;
    const int q = grow >> 2, sub = grow & 3, Wo = p.W >> 1, Ho = p.H >> 1;
    const int xo = q % Wo, t = q / Wo, yo = t % Ho, b = t / Ho;
    return (b * p.H + 2 * yo + (sub >> 1)) * p.W + 2 * xo + (sub & 1);
  };

  // ---- two loader streams walk the workgroup's (tile, K step) sequence ahead of the MFMAs.  Piece q = i*NW + wave of a stage
  // covers tile rows 8q..8q+7; this lane owns row 8q + (lane>>3) and LDS chunk (lane&7), which holds global chunk
  // (lane&7) ^ ((row>>1)&7).
  unsigned char* const xring = smem;
  unsigned char* const wring = smem + XST * C::XBYTES;
  unsigned xb0[C::XPW], xb1[C::XPW];   // X stream: byte offset of (pixel, chunk) in source 0 / 1 for its current tile
  unsigned xmask[C::XPW];              // bit t: tap t stays inside the image (bit 0 only for 1x1)
  unsigned wb[C::WPW];                 // W stream: byte offset of (weight row, chunk), or OOB
  int x_idx = idx, x_k = 0, x_slot = 0; bool x_ok = true;
  int w_idx = idx, w_k = 0, w_slot = 0; bool w_ok = true;
  auto setup_x = [&](int li) {
    const int tile = xcd_first + li, tm = tile / tilesN;
    const int m0 = tm * BM;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const int row = (i * C::NW + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int m = row_to_pixel(m0 + row);
      unsigned mask = 0;
      if (m0 + row < p.M) {
        if (p.ks == 3) {
          const int r = m % HW, y = r / p.W, x = r - y * p.W;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int yy = y + (t / 3 - 1) * p.dil, xx = x + (t % 3 - 1) * p.dil;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1u << t;
          }
        } else mask = 1u;
      }
      xmask[i] = mask;
      xb0[i] = ((unsigned)m * (unsigned)(p.C0 * PL) + g * 8) * 2u;
      xb1[i] = ((unsigned)m * (unsigned)(p.C1 * PL) + g * 8) * 2u;
    }
  };
  auto setup_w = [&](int li) {
    const int tile = xcd_first + li, tm = tile / tilesN, n0 = (tile - tm * tilesN) * BN;
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const int row = (j * C::NW + wave) * 8 + (lane >> 3);       // LDS row of the weight tile
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int q16 = row & 15;
      const int nl = (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);   // channel held by that LDS row
      const int n = n0 + nl;
      wb[j] = (n < p.Cout) ? ((unsigned)n * (unsigned)KW + g * 8) * 2u : OOB;
    }
  };
  auto issue_x = [&]() {               // loads of (x_idx, x_k) into ring slot x_slot, then advance the stream
    unsigned char* sb = xring + x_slot * C::XBYTES;
    int k0 = x_k << 6, plane = 0;
    if (SP) { plane = k0 / K; k0 -= plane * K; plane = split_xplane(plane); }   // K chunk -> (activation plane, k inside it)
    const int tap = k0 / Ctot, cc = k0 - tap * Ctot;
    const bool s1 = cc >= p.C0;
    int dpix = 0;
    if (p.ks == 3) { const int ky = tap / 3, kx = tap - ky * 3; dpix = ((ky - 1) * p.W + (kx - 1)) * p.dil; }
    const int Cs = s1 ? p.C1 : p.C0;
    const unsigned soff = (unsigned)((dpix * Cs * PL + plane * Cs + (s1 ? cc - p.C0 : cc)) * 2);
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const unsigned vo = (xmask[i] & bit) ? (s1 ? xb1[i] : xb0[i]) + soff : OOB;
      lds_ptr dst = (lds_ptr)(sb + (i * C::NW + wave) * 1024);
      if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, vo, 0, 0, 0);
      else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, vo, 0, 0, 0);
    }
    x_slot = x_slot + 1 == XST ? 0 : x_slot + 1;
    if (++x_k == nk) { x_k = 0; x_idx += J; x_ok = x_idx < xcd_count; if (x_ok) setup_x(x_idx); }
  };
  auto issue_w = [&]() {
    unsigned char* sb = wring + w_slot * C::WBYTES;
    unsigned koff = (unsigned)(w_k << 7);                // k0 * 2 bytes: weights are [Cout][taps][Cin] = K contiguous
    if (SP) { const int k0 = w_k << 6, q = k0 / K; koff = (unsigned)((split_wplane(q) * K + (k0 - q * K)) * 2); }
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const unsigned vo = wb[j] == OOB ? OOB : wb[j] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + (j * C::NW + wave) * 1024), 16, vo, 0, 0, 0);
    }
    w_slot ^= 1;
    if (++w_k == nk) { w_k = 0; w_idx += J; w_ok = w_idx < xcd_count; if (w_ok) setup_w(w_idx); }
  };

  // fragment addressing: row = tile-aligned base + (lane&15), so (row>>1)&7 == (lane>>1)&7
  const int frag_lane = (lane & 15) * 128 + ((((lane >> 4)) ^ ((lane >> 1) & 7)) << 4);
  const unsigned char* xfrag[2];
  const unsigned char* wfrag[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    xfrag[kk] = xring + wm * C::TM * 128 + (frag_lane ^ (kk * 64));
    wfrag[kk] = wring + wn * C::TN * 128 + (frag_lane ^ (kk * 64));
  }
  const int fg = lane >> 4, fr = lane & 15;

  // ---- SP with a three-slot X ring: the K loop in REUSE order.  Per 64-wide k0 the four products run as
  //   ph0 (X0, W0)   ph1 (X0, W1)   ph2 (X1, W0b)   ph3 (X2, W0b)
  // so X0 and W0b are staged once and multiplied twice: six tile loads per k0 instead of the eight of the plane-major order
  // (the kernel is bound by the L2 -> LDS fill rate).  X slot = activation plane (0, 1, 2); the W ring's two slots alternate
  // with every load (W0, W1, W0b, W0', ...).  A tile is requested as soon as its slot is free:
  //   ph0: W1(k0), X2(k0)   ph1: W0b(k0)   ph2: W0(k0+1), X0(k0+1)   ph3: X1(k0+1)       (prologue: X0, W0, X1 of the first k0)
  // every load has at least one phase of MFMA work (256 x 128 tile: ~1 k cycles) in front of its first use, the X tiles two or three.
  constexpr bool RU = SP && XST == 3;            // (NP = 3: three phases per k0 - (X0, W0) (X0, W1) (X1, W0b) - on rotating X slots)
  const int nk0 = K >> 6;
  int rx_idx = idx, rx_k0 = 0, rx_pl = 0, rx_slot = 0; bool rx_ok = true;      // X stream position: (tile, k0, plane); slot = plane for triples, rotating for pairs
  int rw_idx = idx, rw_k0 = 0, rw_j = 0, rw_slot = 0; bool rw_ok = true;   // W stream: j = 0, 1, 2 -> weight plane 0 (w0), 2 (w1), 1 (w0b)
  auto ru_issue_x = [&]() {
    unsigned char* sb = xring + (NP == 3 ? rx_slot : rx_pl) * C::XBYTES;
    if (NP == 3) rx_slot = rx_slot == 2 ? 0 : rx_slot + 1;
    const int k0 = rx_k0 << 6, tap = k0 / Ctot, cc = k0 - tap * Ctot;
    const bool s1 = cc >= p.C0;
    int dpix = 0;
    if (p.ks == 3) { const int ky = tap / 3, kx = tap - ky * 3; dpix = ((ky - 1) * p.W + (kx - 1)) * p.dil; }
    const int Cs = s1 ? p.C1 : p.C0;
    const unsigned soff = (unsigned)((dpix * Cs * PL + rx_pl * Cs + (s1 ? cc - p.C0 : cc)) * 2);
    const unsigned bit = 1u << tap;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const unsigned vo = (xmask[i] & bit) ? (s1 ? xb1[i] : xb0[i]) + soff : OOB;
      lds_ptr dst = (lds_ptr)(sb + (i * C::NW + wave) * 1024);
      if (s1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, vo, 0, 0, 0);
      else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, vo, 0, 0, 0);
    }
    if (++rx_pl == PL) { rx_pl = 0; if (++rx_k0 == nk0) { rx_k0 = 0; rx_idx += J; rx_ok = rx_idx < xcd_count; if (rx_ok) setup_x(rx_idx); } }
  };
  auto ru_issue_w = [&]() {
    unsigned char* sb = wring + rw_slot * C::WBYTES;
    const int wpl = rw_j == 0 ? 0 : rw_j == 1 ? 2 : 1;          // (WR: j = 0, 1 only: w0, w1)
    const unsigned koff = (unsigned)((wpl * K + (rw_k0 << 6)) * 2);
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const unsigned vo = wb[j] == OOB ? OOB : wb[j] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + (j * C::NW + wave) * 1024), 16, vo, 0, 0, 0);
    }
    if (WR) rw_slot = rw_slot == 2 ? 0 : rw_slot + 1; else rw_slot ^= 1;
    if (++rw_j == (WR ? 2 : 3)) { rw_j = 0; if (++rw_k0 == nk0) { rw_k0 = 0; rw_idx += J; rw_ok = rw_idx < xcd_count; if (rw_ok) setup_w(rw_idx); } }
  };

  setup_x(idx); setup_w(idx);
  bool x_ahead = false;                                  // an X stage younger than the step about to run is in flight
  int xr = 0, wr = 0;                                    // ring slots the MFMAs read next
  if constexpr (RU) {
    ru_issue_x(); ru_issue_w(); ru_issue_x();            // X0, W0, X1 of the first k0
  } else {
    issue_x(); issue_w();                                // X(0), W(0)
    if (XST == 3 && x_ok) { issue_x(); x_ahead = true; } // X(1)
  }
  while (true) {
    f32x4 acc[C::NJ][C::MI];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j)
#pragma unroll
      for (int i = 0; i < C::MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ctile = xcd_first + idx;
    const int m0c = (ctile / tilesN) * BM, n0c = (ctile % tilesN) * BN;
    using frag_t = typename std::conditional<SP, f16x8, bf16x8>::type;
    if constexpr (RU) {
      // one phase: wait until at most `pend` of this wave's loads are outstanding, barrier, fragments of (X slot xs, W slot ws), the
      // phase's loads, MFMAs
      auto phase = [&](int xs, int ws, auto pend, bool iss_w, bool iss_x, bool strict = false, bool scale_w = false) -> bool {
        constexpr int PEND = decltype(pend)::value;
        if (strict) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the younger load the count allows for was not issued)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PEND) : "memory");
        __builtin_amdgcn_s_barrier();
        const int xo = xs * C::XBYTES, wo = ws * C::WBYTES;
        frag_t fx[2][C::MI], fw[2][C::NJ];
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) fw[0][j] = *reinterpret_cast<const frag_t*>(wfrag[0] + wo + j * 2048);
#pragma unroll
        for (int i = 0; i < C::MI; ++i) fx[0][i] = *reinterpret_cast<const frag_t*>(xfrag[0] + xo + i * 2048);
        if (iss_w && rw_ok) ru_issue_w();
        const bool x_issued = iss_x && rx_ok;
        if (x_issued) ru_issue_x();
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) fw[1][j] = *reinterpret_cast<const frag_t*>(wfrag[1] + wo + j * 2048);
#pragma unroll
        for (int i = 0; i < C::MI; ++i) fx[1][i] = *reinterpret_cast<const frag_t*>(xfrag[1] + xo + i * 2048);
        if (WR && scale_w) {   // w0b = w0 / 2^11 (exact unless subnormal: the same values the staged copy holds)
          const f16 sc = (f16)(1.f / 2048.f);
          const f16x8 scv = {sc, sc, sc, sc, sc, sc, sc, sc};
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) fw[kk][j] = fw[kk][j] * scv;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        return x_issued;
      };
      // Outstanding loads of this wave, oldest first, at the START of each phase (a stream that has run out issues nothing, which only
      // makes the counts smaller than assumed - a stricter wait is always safe):
      //   ph0: X0 W0 X1          need X0, W0  -> <= XPW left
      //   ph1: X1 W1 X2          need W1      -> <= XPW left          (ph0 issues W1 before X2)
      //   ph2: X2 W0b            need X1, W0b -> 0 left
      //   ph3: W0' X0'           need X2      -> <= WPW + XPW left    (X2 was complete at ph2's wait)
      if constexpr (WR && NP == 4) {
      // three W slots: W0(k0) in slot wr, W1(k0) in wr + 1, W0(k0+1) in wr + 2 (mod 3).  Loads in issue order:
      //   prologue X0 W0 X1 | ph0: W1 X2 | ph1: W0' | ph2: X0' | ph3: X1'
      //   ph0 needs W0, X0 (<= XPW left: X1) | ph1 needs W1 (<= XPW left: X2) | ph2 needs X1 (complete since ph1's wait) | ph3 needs X2: the
      //   loads behind it are W0', X0' (<= WPW + XPW left) - strict when the streams have ended and nothing was issued behind it
      for (int k0 = 0; k0 < nk0; ++k0) {
        const int w1s = wr == 2 ? 0 : wr + 1;
        phase(0, wr, std::integral_constant<int, C::XPW>{}, true, true);                          // (X0, W0); requests W1, X2
        const bool wn = rw_ok;
        phase(0, w1s, std::integral_constant<int, C::XPW>{}, true, false);                        // (X0, W1); requests W0'
        const bool xn = phase(1, wr, std::integral_constant<int, C::XPW + C::WPW>{}, false, true, false, true);   // (X1, W0 / 2^11); requests X0'
        phase(2, wr, std::integral_constant<int, C::XPW + C::WPW>{}, false, true, !(wn && xn), true);             // (X2, W0 / 2^11); requests X1'
        wr = w1s == 2 ? 0 : w1s + 1;
      }
      } else if constexpr (WR && NP == 3) {
      //   prologue X0 W0 X1 | ph0: W1 X0' | ph1: W0' | ph2: X1'
      //   ph0 needs X0, W0 (<= XPW left: X1) | ph1 needs W1 (<= XPW left: X0', strict when it was not issued) | ph2 needs X1 (complete)
      for (int k0 = 0; k0 < nk0; ++k0) {
        const int x1s = xr == 2 ? 0 : xr + 1, w1s = wr == 2 ? 0 : wr + 1;
        const bool xn = phase(xr, wr, std::integral_constant<int, C::XPW>{}, true, true);          // (X0, W0); requests W1, X0'
        phase(xr, w1s, std::integral_constant<int, C::XPW>{}, true, false, !xn);                   // (X0, W1); requests W0'
        phase(x1s, wr, std::integral_constant<int, C::XPW + C::WPW>{}, false, true, false, true);  // (X1, W0 / 2^11); requests X1'
        wr = w1s == 2 ? 0 : w1s + 1;
        xr = x1s == 2 ? 0 : x1s + 1;
      }
      } else if constexpr (NP == 4) {
      for (int k0 = 0; k0 < nk0; ++k0) {
        phase(0, wr, std::integral_constant<int, C::XPW>{}, true, true);          // (X0, W0); requests W1, X2
        phase(0, wr ^ 1, std::integral_constant<int, C::XPW>{}, true, false);     // (X0, W1); requests W0b
        phase(1, wr, std::integral_constant<int, 0>{}, true, true);               // (X1, W0b); requests W0', X0'
        phase(2, wr, std::integral_constant<int, C::XPW + C::WPW>{}, false, true);   // (X2, W0b); requests X1'
        wr ^= 1;
      }
      } else {
      // pairs: X0(k0) sits in slot xr, X1(k0) in xr + 1, X0(k0+1) in xr + 2 (mod 3).  Outstanding loads at the start of a phase:
      //   ph0: X0 .. W0 X1      need X0, W0  -> <= XPW left
      //   ph1: X1 W1 X0'        need W1      -> <= XPW left          (ph0 issues W1 before X0')
      //   ph2: X0' W0b          need X1, W0b -> 0 left
      for (int k0 = 0; k0 < nk0; ++k0) {
        const int x1s = xr == 2 ? 0 : xr + 1;
        const bool xn = phase(xr, wr, std::integral_constant<int, C::XPW>{}, true, true);   // (X0, W0); requests W1, X0'
        phase(xr, wr ^ 1, std::integral_constant<int, C::XPW>{}, true, false, !xn);         // (X0, W1); requests W0b   (at the very end of the X stream W1 is the youngest load)
        phase(x1s, wr, std::integral_constant<int, 0>{}, true, true);             // (X1, W0b); requests W0', X1'
        wr ^= 1;
        xr = x1s == 2 ? 0 : x1s + 1;
      }
      }
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      const int xo = xr * C::XBYTES, wo = wr * C::WBYTES;
      // my pieces of this K step have landed (loads retire in order: only the younger X stage may still be in flight)
      if (XST == 3 && x_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::XPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // everyone's have; everyone is done reading the slots refilled below
      // all fragment reads of this K step are issued up front (their latency hides behind the
      // loader's address arithmetic), then the MFMAs run back to back
      frag_t fx[2][C::MI], fw[2][C::NJ];
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) fw[0][j] = *reinterpret_cast<const frag_t*>(wfrag[0] + wo + j * 2048);
#pragma unroll
      for (int i = 0; i < C::MI; ++i) fx[0][i] = *reinterpret_cast<const frag_t*>(xfrag[0] + xo + i * 2048);
      if (w_ok) issue_w();                               // W one step ahead, issued BEFORE the younger X stage
      x_ahead = x_ok;
      if (x_ok) issue_x();                               // X two steps ahead (one when XST == 2)
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) fw[1][j] = *reinterpret_cast<const frag_t*>(wfrag[1] + wo + j * 2048);
#pragma unroll
      for (int i = 0; i < C::MI; ++i) fx[1][i] = *reinterpret_cast<const frag_t*>(xfrag[1] + xo + i * 2048);
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
      xr = xr + 1 == XST ? 0 : xr + 1;
      wr ^= 1;
    }
    }
    idx += J;
    const bool has_next = idx < xcd_count;
    RangeWatch rw;   // (split.h: the maximum of |x| over the values this lane writes as planes; per tile, so that nothing lives across the K loop)

  // ---- epilogue: lane holds channels n = nb + 32t + (lane>>4)*8 + 0..7 of pixel m = mb + 16i + (lane&15)
    // ConvParams::up_z (CRAFT's commuted up-convolutions): a pass of its own in front, loads only - bias + the bilinear 2x upsample of the half-resolution
    // tensor at each row's pixel, folded into the accumulators in place.  Inside the loop below every block's four-tap gather sat behind the previous
    // block's stores (vmcnt counts both: a write round trip and a read round trip per 16-row block; gemm_sp.hip's two-pass epilogue has the account).
    bool folded = false;
    if constexpr (SP && !(BM == 256 && BN == 256)) {   // (the 256 x 256 tile is not one these layers run on: launch_gemm2)
      if (p.up_z) {
        folded = true;
        const int Hl = p.H >> 1, Wl = p.W >> 1;
        int64_t o00[C::MI], o01[C::MI], o10[C::MI], o11[C::MI];
        float wx[C::MI], wy[C::MI];
  #pragma unroll
        for (int i = 0; i < C::MI; ++i) {
          const int grow = m0c + wm * C::TM + i * 16 + fr;
          const int m = row_to_pixel(grow < p.M ? grow : 0);
          const int xo = m % p.W, tq = m / p.W, yo = tq % p.H, bq = tq / p.H;
          const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
          const int y0 = (int)sy, x0 = (int)sx;
          const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0), x1 = x0 + (x0 < Wl - 1 ? 1 : 0);
          wy[i] = sy - (float)y0; wx[i] = sx - (float)x0;
          const int64_t pb = (int64_t)bq * Hl * Wl;
          o00[i] = (pb + (int64_t)y0 * Wl + x0) * p.up_ld; o01[i] = (pb + (int64_t)y0 * Wl + x1) * p.up_ld;
          o10[i] = (pb + (int64_t)y1 * Wl + x0) * p.up_ld; o11[i] = (pb + (int64_t)y1 * Wl + x1) * p.up_ld;
        }
  #pragma unroll
        for (int t = 0; t < C::NJ / 2; ++t) {
          const int n = n0c + wn * C::TN + t * 32 + fg * 8;
          if (n >= p.Cout) continue;
          float bv[8];
          if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
          } else {
  #pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = 0.f;
          }
  #pragma unroll
          for (int i = 0; i < C::MI; ++i) {
            const float lx1 = wx[i], lx0 = 1.f - lx1, ly1 = wy[i], ly0 = 1.f - ly1;
            const float* zb = p.up_z + n;
  #pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 a = *reinterpret_cast<const float4*>(zb + o00[i] + 4 * h), b = *reinterpret_cast<const float4*>(zb + o01[i] + 4 * h);
              const float4 c = *reinterpret_cast<const float4*>(zb + o10[i] + 4 * h), d = *reinterpret_cast<const float4*>(zb + o11[i] + 4 * h);
              const float aa[4] = {a.x, a.y, a.z, a.w}, bb[4] = {b.x, b.y, b.z, b.w}, cc[4] = {c.x, c.y, c.z, c.w}, dd[4] = {d.x, d.y, d.z, d.w};
  #pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float top = fmaf(lx1, bb[e], lx0 * aa[e]), bot = fmaf(lx1, dd[e], lx0 * cc[e]);
                acc[2 * t + h][i][e] = fmaf(acc[2 * t + h][i][e], p.out_scale, bv[4 * h + e]) + fmaf(ly1, bot, ly0 * top);
              }
            }
          }
        }
      }
    }
  #pragma unroll
    for (int t = 0; t < C::NJ / 2; ++t) {
      const int n = n0c + wn * C::TN + t * 32 + fg * 8;
      if (n >= p.Cout) continue;
      float bv[8];
      if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      } else {
  #pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      }
  #pragma unroll
      for (int i = 0; i < C::MI; ++i) {
        const int grow = m0c + wm * C::TM + i * 16 + fr;
        const bool valid = grow < p.M;                 // uniform over each group of 4 lanes when pooling (M % 4 == 0)
        const int m = row_to_pixel(valid ? grow : 0);
        float v[8];
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (SP) {
            v[e] = folded ? acc[2 * t][i][e] : fmaf(acc[2 * t][i][e], p.out_scale, bv[e]);
            v[4 + e] = folded ? acc[2 * t + 1][i][e] : fmaf(acc[2 * t + 1][i][e], p.out_scale, bv[4 + e]);
          }
          else { v[e] = acc[2 * t][i][e] + bv[e]; v[4 + e] = acc[2 * t + 1][i][e] + bv[4 + e]; }
        }
        if (p.resid && valid) {
          const float* rp = p.resid + (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld + n;
          const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
          v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        if (p.act == kActRelu) {
  #pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == kActGelu) {
  #pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = SP ? (p.gelu_lut ? gelu_hermite(v[e], glut) : gelu_exact(v[e])) : gelu_lut(v[e], glut);
        }
        if constexpr (SP) {
          if (p.dbg_flags & 1) { if (v[0] == 1.2345e30f) reinterpret_cast<float*>(p.out)[0] = v[1]; continue; }   // timing experiment: no output stores
          if (p.out && valid) st_split_n(p.out, (int64_t)m, p.out_ld, n, v, p.out_planes, rw);
          if (p.out_relu && valid) {
            float w[8];
  #pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = fmaxf(v[e], 0.f);
            st_split_n(p.out_relu, (int64_t)m, p.out_ld, n, w, p.out_planes, rw);
          }
        } else {
        if (p.out && valid) {
          bf16x8 o;
  #pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          st_out(reinterpret_cast<bf16*>(p.out) + (int64_t)m * p.out_ld + n, o, p.store_policy);
        }
        if (p.out_relu && valid) {
          bf16x8 o;
  #pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(v[e], 0.f);
          st_out(reinterpret_cast<bf16*>(p.out_relu) + (int64_t)m * p.out_ld + n, o, p.store_policy);
        }
        }
        if (p.out_f32 && valid) {
          float* op = p.out_f32 + (int64_t)m * p.out_f32_ld + n;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (p.out_pool) {   // 2x2 max over lanes fr, fr^1, fr^2, fr^3; rounding to bf16 commutes with max
          float w[8];
  #pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = p.pool_relu ? fmaxf(v[e], 0.f) : v[e];
            x = fmaxf(x, __shfl_xor(x, 1));
            x = fmaxf(x, __shfl_xor(x, 2));
            w[e] = x;
          }
          if constexpr (SP) {
            if (valid && (fr & 3) == 0) st_split_n(p.out_pool, (int64_t)(grow >> 2), p.out_ld, n, w, p.out_planes, rw);
          } else {
            bf16x8 o;
  #pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)w[e];
            if (valid && (fr & 3) == 0) st_out(reinterpret_cast<bf16*>(p.out_pool) + (int64_t)(grow >> 2) * p.out_ld + n, o, p.store_policy);
          }
        }
      }
    }
    if constexpr (SP) rw.flush(p.range_flag, p.range_tag);
    if (!has_next) break;
  }
}

static int num_cus() { return device_cu_count(256); }   // (per device: a process may drive several)

template <int BM, int BN, int WM, int WN, int MINB, int XST, int NP = 0, int WST = 2>
static void launch_g2(const ConvParams& p_in, hipStream_t s) {
  const ConvParams p = with_range_ctx(p_in);
  constexpr bool SP = NP != 0;
  using C = G2Cfg<BM, BN, WM, WN, XST, WST>;
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm2_kernel<BM, BN, WM, WN, MINB, XST, NP, WST>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + (C::LDS + 8208 <= 160 * 1024 ? 8208 : 0))); });
  const size_t lds = C::LDS + (p.act == kActGelu && p.gelu_lut ? (SP ? 8208 : 8192) : 0);
  // persistent grid: as many workgroups as fit the chip at once (a multiple of 8: one share per XCD), never more than tiles
  const int per_cu = std::max(1, std::min((int)(160 * 1024 / lds), 2048 / C::NT));
  const int cap = num_cus() * per_cu / 8 * 8;
  const int grid = std::min((tilesM * tilesN + 7) / 8 * 8, std::max(cap, 8));
  hipLaunchKernelGGL((gemm2_kernel<BM, BN, WM, WN, MINB, XST, NP, WST>), dim3(grid), dim3(C::NT), lds, s, p);
}

// Phi table of the GELU epilogue, one per device, built on first use (host erf in double)
const void* gelu_lut_for_current_device() {
  static std::mutex mu;
  static const void* lut[64] = {nullptr};
  int dev = 0;
  TTR_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64) throw std::runtime_error("gemm2: device index out of range");
  if (!lut[dev]) {
    std::vector<float> h(2048);
    for (int i = 0; i < 1024; ++i) {
      const double x0 = -8.0 + i / 64.0, x1 = x0 + 1.0 / 64.0;
      const double p0 = 0.5 * (1.0 + std::erf(x0 * 0.70710678118654752440)), p1 = 0.5 * (1.0 + std::erf(x1 * 0.70710678118654752440));
      h[2 * i] = (float)p0; h[2 * i + 1] = (float)(p1 - p0);
    }
    void* d = nullptr;
    TTR_HIP_CHECK(hipMalloc(&d, 8192));
    TTR_HIP_CHECK(hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice));
    lut[dev] = d;
  }
  return lut[dev];
}

// Cubic table of the split mode's GELU epilogue (common.h: gelu_hermite): float4[512], entry i = the Hermite cubic of Phi over
// [x_i, x_i + 1/32], x_i = -8 + i / 32, in t = (x - x_i) 32: {Phi_i, h phi_i, 3 d - h (2 phi_i + phi_i+1), -2 d + h (phi_i + phi_i+1)}, d = Phi_i+1 - Phi_i
// (host erf / exp in double)
const void* gelu_hermite_lut_for_current_device() {
  static std::mutex mu;
  static const void* lut[64] = {nullptr};
  int dev = 0;
  TTR_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64) throw std::runtime_error("gemm2: device index out of range");
  if (!lut[dev]) {
    std::vector<float> h(4 * 512 + 8, 0.f);
    auto Phi = [](double x) { return 0.5 * (1.0 + std::erf(x * 0.70710678118654752440)); };
    auto phi = [](double x) { return std::exp(-0.5 * x * x) * 0.39894228040143267794; };
    const double hh = 1.0 / 32.0;
    for (int i = 0; i < 512; ++i) {
      const double x0 = -8.0 + i * hh, x1 = x0 + hh;
      const double d = Phi(x1) - Phi(x0), p0 = hh * phi(x0), p1 = hh * phi(x1);
      h[4 * i] = (float)Phi(x0); h[4 * i + 1] = (float)p0; h[4 * i + 2] = (float)(3.0 * d - 2.0 * p0 - p1); h[4 * i + 3] = (float)(-2.0 * d + p0 + p1);
    }
    void* d = nullptr;
    TTR_HIP_CHECK(hipMalloc(&d, h.size() * 4));
    TTR_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    lut[dev] = d;
  }
  return lut[dev];
}

static int g_x_ring3 = 1;
void set_gemm2_x_ring3(int v) { g_x_ring3 = v; }
static int g_split_wreg = 1;    // split mode, 256 x 128 / 128 x 256 tiles: w0b formed in registers from the W0 tile (three W slots) instead of staged
void set_gemm2_split_wreg(int v) { g_split_wreg = v; }
static int g_split_stream = 2;  // split mode, pairs, plain GEMM shapes: the streamlined kernels of gemm_sp.hip (2: tile shape by problem, 1: always two 128 x 128 workgroups per CU, 0: gemm2's own loop)
void set_gemm2_split_stream(int v) { g_split_stream = v; }
static int g_split_stream4 = 1; // ... and for activation triples (proj, the decoder's linears)
void set_gemm2_split_stream4(int v) { g_split_stream4 = v; }
static int g_split_dbg = 0;     // split mode timing experiments (results are wrong): 1 = no output stores
void set_gemm2_split_dbg(int v) { g_split_dbg = v; }
static int g_split_few = 1;     // split mode, gemm2's own loop: 128 x 64 tiles when the 128 x 128 ones number no more than the CUs
void set_gemm2_split_few(int v) { g_split_few = v; }
static int g_split_cfg = 0;     // split mode: force a tile configuration (0 = automatic)
void set_gemm2_split_cfg(int v) { g_split_cfg = v; }
static int g_split_reuse = 1;   // split mode: 1 = reuse-order K loop (X0 and W0b staged once per k0), 0 = plane-major order with a two-slot X ring
void set_gemm2_split_reuse(int v) { g_split_reuse = v; }

const char* gemm2_check(const ConvParams& p) {
  const int Ctot = p.C0 + p.C1;
  const int es_in = p.split == 4 ? 6 : p.split == 3 ? 4 : 2;          // bytes per activation element (three / two f16 planes when split)
  if (p.split && p.split != 3 && p.split != 4) return "gemm2: split must be 3 (pairs) or 4 (triples)";
  if (p.split && p.out_planes != 0 && p.out_planes != 2 && p.out_planes != 3) return "gemm2: out_planes must be 0, 2 or 3";
  const int es_out = p.split ? (p.out_planes ? 2 : 4) : 2;            // bytes per element of out / out_relu / out_pool (per plane)
  const int ovec = 16 / es_out;                                        // elements per 16-byte store
  if (p.ks != 1 && p.ks != 3) return "gemm2: ks must be 1 or 3";
  if (Ctot % 64 || p.C0 % 64) return "gemm2: channel counts must be multiples of 64";
  if (p.Cout % 8) return "gemm2: Cout must be a multiple of 8";
  if (p.relu0 || p.relu1) return "gemm2: ReLU-on-load is not supported";
  if (p.out && (p.out_ld % ovec || ((uintptr_t)p.out & 15))) return "gemm2: output must be 16-byte aligned";
  if (p.out_relu && (!p.out || ((uintptr_t)p.out_relu & 15))) return "gemm2: out_relu needs out and 16-byte alignment";
  if (p.out_pool && ((p.H | p.W) & 1 || p.out_ld % ovec || ((uintptr_t)p.out_pool & 15))) return "gemm2: fused max-pool needs even H, W and a 16-byte aligned output";
  if (p.out_f32 && (p.out_f32_ld % 4 || ((uintptr_t)p.out_f32 & 15))) return "gemm2: f32 output must be 16-byte aligned";
  if (p.resid && (p.resid_ld % 4 || ((uintptr_t)p.resid & 15))) return "gemm2: residual must be 16-byte aligned";
  if (p.bias && ((uintptr_t)p.bias & 15)) return "gemm2: bias must be 16-byte aligned";
  if (((uintptr_t)p.in0 & 15) || ((uintptr_t)p.wgt & 15) || (p.C1 && ((uintptr_t)p.in1 & 15))) return "gemm2: operands must be 16-byte aligned";
  const size_t lim = (size_t)1 << 31;   // buffer offsets: valid lanes < 2^31, 0x80000000 is the out-of-range marker
  const int K = p.ks * p.ks * Ctot * (p.split ? 3 : 1);
  if ((size_t)p.M * p.C0 * es_in >= lim || (size_t)p.M * p.C1 * es_in >= lim || (size_t)p.Cout * K * 2 >= lim) return "gemm2: tensor too large for 32-bit buffer offsets";
  if (p.M != p.B * p.H * p.W || p.M <= 0 || p.Cout <= 0) return "gemm2: bad shape";
  if (p.split && !(p.out_scale > 0.f)) return "gemm2: split mode needs out_scale";
  if (p.up_z && (!p.split || p.ks != 1 || p.C1 || (p.H & 1) || (p.W & 1) || p.up_ld % 4 || p.up_ld < p.Cout || ((uintptr_t)p.up_z & 15) || p.out_pool || p.resid))
    return "gemm2: the half-resolution addend (up_z) is the split mode's, on a single-source 1x1 layer over even H and W";
  return nullptr;
}

// cfg: 0 auto, 1 = 256x256/8w, 2 = 256x128/8w, 3 = 128x128/4w, 4 = 256x64/4w, 5 = 128x64/4w, 6 = 128x256/8w
static int g_up_2d = 0;         // the skip halves of the commuted up-convolutions on 2-D tiles (ConvParams::up_2d; tuning key up_2d): bit-identical, measured 2 % SLOWER (the gather is not bound by its locality) - off
void set_gemm2_up_2d(int v) { g_up_2d = v; }
static int g_up_resident = 1;   // the skip half of CRAFT's upconv4.0 (128 -> 64 channels + the half-resolution addend) on conv1u.hip's persistent kernel (tuning key up_resident)
void set_gemm2_up_resident(int v) { g_up_resident = v; }
void launch_gemm2(const ConvParams& p_in, int cfg, hipStream_t s) {
  if (const char* e = gemm2_check(p_in)) throw std::runtime_error(e);
  if (g_up_resident && cfg == 0 && conv1u_eligible(p_in)) return launch_conv1u(p_in, s);   // bit-identical (test)
  ConvParams p = p_in;
  const int Ctot = p.C0 + p.C1;
  p.gelu_lut = p.act == kActGelu ? (p.split ? gelu_hermite_lut_for_current_device() : gelu_lut_for_current_device()) : nullptr;
  if (cfg == 0 && p.split && g_split_cfg) cfg = g_split_cfg;
  if (cfg == 0 && p.split && p.Cout > 64) {
    // split mode (reuse-order K loop): 256 x 128 tiles, one workgroup per CU - the 256 x 256 tile with its 160 KB of LDS loses to it on
    // every shape measured (fc1 1.79 vs ~1.0 ms); small problems keep the 128-wide tiles so that the chip fills
    const int64_t t2 = (int64_t)((p.M + 255) / 256) * ((p.Cout + 127) / 128);
    cfg = t2 >= 2 * num_cus() ? 2 : 3;
    // wide outputs of many rows: 128 x 256 tiles halve the number of passes over the activation planes (the big operand; the weights
    // stay in L2): qkv 848 -> 774 us at 1280 crops.  Not for Cout = 384 (1.5 tiles of 256)
    if (cfg == 2 && p.Cout >= 1024 && p.ks == 1 && p.M >= 65536) cfg = 6;
  }
  if (cfg == 0) {   // measured on MI355X (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt)
    if (p.Cout <= 64) cfg = 5;
    else {
      cfg = 3;
      if (p.Cout % 256 == 0) {   // 256x256 tiles pay when they fill the 256 CUs evenly
        const int64_t tiles = (int64_t)((p.M + 255) / 256) * (p.Cout / 256);
        const double eff = (double)tiles / (double)(((tiles + 255) / 256) * 256);
        if (tiles >= 1024 || (tiles <= 256 && eff >= 0.74) || eff >= 0.85) cfg = 1;
      }
      // plenty of 256x128 tiles: one per CU with the deeper X ring edges out two 128x128 per CU (fc2 +4 %, qkv +4 %)
      if (cfg == 3 && g_x_ring3 && p.act != kActGelu && (int64_t)((p.M + 255) / 256) * ((p.Cout + 127) / 128) >= 512) cfg = 2;
    }
  }
  // X ring depth: 3 (activation tiles two K steps ahead) wherever the LDS budget keeps the configuration's workgroups-per-CU;
  // the GELU table (8 KiB) pushes the 256x256 and 128x128 tiles back to 2
  const bool deep = g_x_ring3 && p.act != kActGelu;
  if (p.split) p.dbg_flags = g_split_dbg;
  if (p.up_z && cfg == 1) cfg = 2;                 // (the half-resolution addend's epilogue is not compiled into the 256 x 256 tile)
  {
    const int bm = (cfg == 1 || cfg == 2 || cfg == 4) ? 256 : 128;   // (cfg 3 may still become 5 below: 128 rows either way)
    p.up_2d = (g_up_2d && p.up_z && !p.out_pool && p.W % 16 == 0 && p.H % (bm / 16) == 0 && p.M % bm == 0 && p.M == p.B * p.H * p.W) ? 1 : 0;
  }
  if (p.split && cfg == 1) p.gelu_lut = nullptr;   // 256 x 256 tiles fill the LDS: erf instead of the table
  if (p.split == 4 && (cfg == 2 || cfg == 3 || cfg == 6) && g_split_stream && g_split_stream4 && gemm_sp_eligible(p)) return launch_gemm_sp(p, cfg, s);   // triples
  if (p.split == 3 && (cfg == 2 || cfg == 3 || cfg == 6) && g_split_stream && gemm_sp_eligible(p)) {
    // measured at 1280 crops (tools/x4_parseq_ab.sh): qkv 607 / fc1 835 us on the 128 x 256 tiles against 730 / 892 on two 128 x 128 workgroups
    // per CU; fc2 (Cout 384, K 1536: three long tiles per row block, the epilogue 1 / 24 of a tile) 630 against 586
    int sc = cfg;
    if (g_split_stream == 1 || (p.Cout < 512 && Ctot >= 1024)) sc = 3;
    return launch_gemm_sp(p, sc, s);
  }
  // a dilated 3x3 layer of a batch (CRAFT's slice5.1: 1.06 PFLOP/s executed on this file's loop, whose tap arithmetic runs per load) on gemm_sp.hip's loop
  if (p.split == 3 && p.ks == 3 && (cfg == 2 || cfg == 3) && g_split_stream && gemm_sp_ks3_eligible(p))
    return launch_gemm_sp_ks3(p, cfg, s);
  if (p.x_tiled || p.out_tiled) throw std::runtime_error("gemm2: tiled planes are gemm_sp.hip's (this shape did not qualify for it)");
  // a page's worth of pixels on this loop (CRAFT's dilated 3x3 and two-source 1x1 layers): no more 128 x 128 tiles than CUs - 128 x 64 tiles on twice
  // as many workgroups move three quarters of the bytes per K step each (slice5.1 at one page 182 -> 157 us, upconv1.0 65 -> 52, upconv2.0 41 -> 33)
  if (p.split && cfg == 3 && g_split_few && (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128) <= num_cus()) cfg = 5;
  if (p.split) {   // three-slot X ring everywhere: the reuse-order K loop
    const bool ru = g_split_reuse != 0 || p.split == 3;
    if (p.split == 3) {
      switch (cfg) {
        case 1: return launch_g2<256, 256, 2, 4, 1, 3, 3>(p, s);
        case 2: return g_split_wreg ? launch_g2<256, 128, 4, 2, 1, 3, 3, 3>(p, s) : launch_g2<256, 128, 4, 2, 1, 3, 3>(p, s);
        case 3: return launch_g2<128, 128, 2, 2, 2, 3, 3>(p, s);
        case 4: return launch_g2<256, 64, 4, 1, 1, 3, 3>(p, s);
        case 5: return launch_g2<128, 64, 2, 2, 2, 3, 3>(p, s);
        case 6: return g_split_wreg ? launch_g2<128, 256, 2, 4, 1, 3, 3, 3>(p, s) : launch_g2<128, 256, 2, 4, 1, 3, 3>(p, s);
        default: throw std::runtime_error("gemm2: unknown configuration");
      }
    }
    switch (cfg) {
      case 1: return ru ? launch_g2<256, 256, 2, 4, 1, 3, 4>(p, s) : launch_g2<256, 256, 2, 4, 1, 2, 4>(p, s);
      case 2: return ru ? (g_split_wreg ? launch_g2<256, 128, 4, 2, 1, 3, 4, 3>(p, s) : launch_g2<256, 128, 4, 2, 1, 3, 4>(p, s)) : launch_g2<256, 128, 4, 2, 1, 2, 4>(p, s);
      case 3: return ru ? launch_g2<128, 128, 2, 2, 2, 3, 4>(p, s) : launch_g2<128, 128, 2, 2, 2, 2, 4>(p, s);
      case 4: return ru ? launch_g2<256, 64, 4, 1, 1, 3, 4>(p, s) : launch_g2<256, 64, 4, 1, 2, 2, 4>(p, s);
      case 5: return ru ? launch_g2<128, 64, 2, 2, 2, 3, 4>(p, s) : launch_g2<128, 64, 2, 2, 2, 2, 4>(p, s);
      case 6: return ru ? (g_split_wreg ? launch_g2<128, 256, 2, 4, 1, 3, 4, 3>(p, s) : launch_g2<128, 256, 2, 4, 1, 3, 4>(p, s)) : launch_g2<128, 256, 2, 4, 1, 2, 4>(p, s);
      default: throw std::runtime_error("gemm2: unknown configuration");
    }
  }
  switch (cfg) {
    case 1: return deep ? launch_g2<256, 256, 2, 4, 1, 3>(p, s) : launch_g2<256, 256, 2, 4, 1, 2>(p, s);
    case 2: return deep ? launch_g2<256, 128, 4, 2, 1, 3>(p, s) : launch_g2<256, 128, 4, 2, 1, 2>(p, s);
    case 3: return deep ? launch_g2<128, 128, 2, 2, 2, 3>(p, s) : launch_g2<128, 128, 2, 2, 2, 2>(p, s);
    case 4: return launch_g2<256, 64, 4, 1, 2, 2>(p, s);
    case 5: return launch_g2<128, 64, 2, 2, 2, 2>(p, s);
    case 6: return deep ? launch_g2<128, 256, 2, 4, 1, 3>(p, s) : launch_g2<128, 256, 2, 4, 1, 2>(p, s);
    default: throw std::runtime_error("gemm2: unknown configuration");
  }
}

template <typename T>
__global__ void fill_random_kernel(T* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) ^ (seed * 0x9E3779B9u);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = (T)(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
  }
}

void launch_fill_random(Precision prec, void* p, size_t n, unsigned seed, float scale, hipStream_t s) {
  if (n == 0) return;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 65536);
  if (prec == kBF16) hipLaunchKernelGGL(fill_random_kernel<bf16>, dim3(grid), dim3(256), 0, s, (bf16*)p, n, seed, scale);
  else hipLaunchKernelGGL(fill_random_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)p, n, seed, scale);
}

}  // namespace ttr
