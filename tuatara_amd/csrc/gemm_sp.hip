// Split-operand linear layers of the recogniser, streamlined (gfx950 / MI355X):   out = act((X W^T) / S + bias (+ resid))
// with X as f16 activation PAIRS (planes x0 | x1 per row, split.h) or exact TRIPLES (x0 | x1 | x2) and W as [w0 | w0b | w1] rows.
//
// Same contract and the same LDS image as gemm2.hip's split mode (ConvParams, ks = 1, one source; replaces the nn.Linear calls inside
// the TorchScript PARSeq run at tuatara.cpp:307), for the shapes where that kernel's K loop was bound by its own bookkeeping: rocprof's
// counters on the 1280-crop encoder GEMMs showed 4 vector and 2.5 scalar instructions per MFMA, all of them - the loader's tap / plane /
// tile arithmetic with its branches - between the barrier and the MFMAs of every phase, in all eight waves at once, so the matrix pipe idled
// half of every phase (profiles/r03_pmc_stall_parseq.txt).  Here
//   * every tile is read from LDS ONCE per k0: products run (X0, W0) (X0, W1) (X1, W0 / 2^11) with the X0 and W0 fragments kept in registers
//     (32 fragment reads per k0 instead of 48) and w0b formed in registers;
//   * fragments are read one phase AHEAD of their MFMAs (the barrier of phase s certifies the tile phase s + 1 multiplies), so the LDS latency
//     sits under 32 MFMAs instead of in front of them;
//   * the loader is branch-free and costs no vector instruction per load: a lane's row offsets are fixed per tile, the (plane, k0) offset rides
//     in the instruction's scalar offset, the tile change is an add of a precomputed delta under a scalar mask, and a stream that has run out
//     keeps issuing out-of-range loads (zero fill, no traffic) so that every s_waitcnt count is exact without special cases;
//   * three X and three W ring slots: activation tiles are requested 3 - 4 phases before their barrier, weight tiles 3 - 4.
// Per k0 (phases 0, 1, 2), in steady state:
//     ph0: MFMA X0(k) W0(k)        reads W1(k)               requests X1(k+1), W1(k+1)
//     ph1: MFMA X0(k) W1(k)        reads X1(k)               requests W0(k+2)               then  w0s = W0(k) / 2^11 (registers)
//     ph2: MFMA X1(k) w0s          reads X0(k+1), W0(k+1)    requests X0(k+2)
// (k runs on across the workgroup's tiles; the epilogue sits between ph2 of a tile's last k0 and ph0 of the next tile, whose first fragments
// are already in registers and whose later tiles keep landing meanwhile.)
#include <algorithm>
#include <stdexcept>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

typedef __attribute__((address_space(3))) void* lds_ptr;

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// fp32 output stores are streaming (nt): a launch writes 0.25 GB that nothing reads before the next launch, through an L2 of 4 MB per XCD that also
// holds the weight planes every tile re-reads; a lane's 32 bytes and its three neighbours' make whole 128-byte lines (fc2 -3 %, proj -2 % at 1280
// crops).  The f16 plane outputs keep the default policy: a row block contributes 64-byte pieces that only meet their other half in L2, and
// streamed they reach the memory as partial writes (fc1 +5 %, qkv + attention +2 %; HBM fetch of fc1 848 -> 337 MB per launch all the same).
static __device__ __forceinline__ void sp_store_f32x8(float* o, const float (&v)[8]) {
  __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(o));
  __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(o + 4));
}

template <int BM, int BN, int WM, int WN, int XST, int WST>
struct SpCfg {
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int TM = BM / WM, TN = BN / WN;   // wave tile
  static constexpr int MI = TM / 16, NJ = TN / 16;
  static constexpr int XPW = BM / 8 / NW, WPW = BN / 8 / NW;   // 1-KiB pieces (8 rows x 128 B) per wave and tile
  static constexpr int XBYTES = BM * 128, WBYTES = BN * 128;
  static constexpr int LDS = XST * XBYTES + WST * WBYTES;
  static_assert((XST == 2 || XST == 3) && (WST == 2 || WST == 3) && !(XST == 2 && WST == 3), "ring depths: (3, 3), (3, 2) or (2, 2)");
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && TN % 32 == 0 && TM % 16 == 0, "tile shape");
};

// NP = 3: activation pairs, phases (X0, W0) (X0, W1) (X1, w0s) as described above.  NP = 4: exact triples, a fourth phase (X2, w0s); X2 goes into the
// registers X0 has left and the next k0's X0 into X1's, so the two activation fragment sets swap roles every k0 (the K loop runs in pairs of k0;
// K / 64 must be even); rings X 3 + W 2:
//     ph0: MFMA X0(k) W0(k)   reads W1(k)               requests X0(k+1), W0(k+1)
//     ph1: MFMA X0(k) W1(k)   reads X1(k)               requests W1(k+1)              then  w0s = W0(k) / 2^11
//     ph2: MFMA X1(k) w0s     reads X2(k)               requests X1(k+1)
//     ph3: MFMA X2(k) w0s     reads X0(k+1), W0(k+1)    requests X2(k+1)
// EPI = 1: the encoder's qkv projection with the self-attention of a (crop, head) as its epilogue (below, "attention epilogue"): tiles of
// 128 rows (one crop) x 192 channels (Q | K | V of one head: the caller passes the weight rows in head-major order), eight waves as 4 x 2.
// EM: the plain epilogue's case fixed at compile time (0 = by ConvParams' flags at run time, every case in one kernel).  1 = the encoder's fc1: GELU by the
// table, the hidden activation out as tiled pairs; 2 = the residual linears (proj, fc2, the decoder's): bias + residual, fp32 rows out.  With the flags
// tested per 8-value block hipcc serialised the GELU table reads (one LDS round trip and two branches per VALUE) and spilled scalars into lanes.
// KS3: a 3x3 convolution of any dilation (CRAFT's slice5.1: dilation 6 on the 1/16-resolution map, which conv3p.hip's patches do not take) as this GEMM with
// K = 9 taps x Cin: the weight stream is unchanged (its rows are [tap][Cin] already); the activation stream re-forms its lanes' row offsets at every tap
// change - pixel + (dy W + dx) dil, out of range (zero fill) where the tap leaves the image - from two registers per piece kept per tile (offset of the
// centre pixel, its (y, x)): ~8 vector instructions per piece and tap, every Cin / 64 k steps.  gemm2.hip's loop did that arithmetic per load.
// DBG (timing experiments, results are wrong): bit 0 = the loader streams request nothing, bit 1 = no fragment reads from LDS (the MFMAs run on what the registers hold)
// STAG (pairs loop, eight waves): the second-dispatched half of the workgroup (waves 4 - 7: the SIMD partners of waves 0 - 3) runs half a phase behind - it
// defers the second 32-deep half of every phase's MFMAs to the far side of the next barrier, so that behind a barrier one wave of a SIMD issues matrix
// instructions back to back while its partner issues the phase's LDS-DMA requests (60 - 185 cycles each, during which that wave issues nothing else) and
// fragment reads, and the other way round in the second half of the interval (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Same requests in the same
// intervals, same accumulation order per accumulator: results are bit-identical.
template <int BM, int BN, int WM, int WN, int XST, int WST, int MINB, bool SCHED, int NP, int EPI = 0, int EM = 0, bool KS3 = false, int DBG = 0, bool STAG = false>
__global__ __launch_bounds__(WM * WN * 64, MINB) void gemm_sp_kernel(ConvParams p) {
  using C = SpCfg<BM, BN, WM, WN, XST, WST>;
  static_assert(NP == 3 || (NP == 4 && XST == 3 && WST == 2), "pairs, or triples on rings 3 + 2");
  static_assert(EPI == 0 || (EPI == 1 && BM == 128 && BN == 192 && WM == 4 && WN == 2 && NP == 3), "attention epilogue: 128 x 192 tiles on 4 x 2 waves, activation pairs");
  constexpr int PLX = NP == 3 ? 2 : 3;                                    // activation planes per row
  if (p.skip && __builtin_nontemporal_load(p.skip) >= p.skip_n) return;   // AR early exit (ConvParams::skip): uniform, before any barrier
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  static_assert(EM == 0 || EPI == 0, "fixed epilogue cases are the plain epilogue's");
  static_assert(!KS3 || (NP == 3 && EPI == 0), "3x3 taps: the pairs loop");
  static_assert(!STAG || (NP == 3 && WM * WN == 8 && SCHED), "staggered halves: the pairs loop on eight waves");
  float2* const glut = reinterpret_cast<float2*>(smem + C::LDS);          // Hermite GELU table behind the rings (common.h: gelu_hermite)
  if (EM == 1 || EM == 3 || (EM == 0 && p.act == kActGelu && p.gelu_lut)) {
    for (int i = tid; i < 512; i += C::NT) reinterpret_cast<uint4*>(glut)[i] = reinterpret_cast<const uint4*>(p.gelu_lut)[i];
  }   // visible after the first barrier

  // persistent, XCD-aware tile schedule: as gemm2.hip
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
  if (p.cu_stagger > 0) {   // ConvParams::cu_stagger: a start delay by workgroup, uniform over the workgroup, before anything is requested
    const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) % p.cu_stagger_groups) * (unsigned)p.cu_stagger / (unsigned)p.cu_stagger_groups;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }

  const int K = p.C0, nkt = K >> 6, nk0 = KS3 ? 9 * nkt : nkt;   // (KS3: k steps per tap, per tile)
  const int Kw = KS3 ? 9 * K : K;                                  // halves per weight plane
  const __amdgpu_buffer_rsrc_t rsx = sp_rsrc(p.in0, (unsigned)((size_t)p.M * K * 2 * PLX));   // rows [x0 | x1 (| x2)]
  // weights: rows [w0 | w0b | w1], or (wgt_tiled) the same halves as 1-KiB pieces [Cout / 8][plane][K / 64][8 rows][64]: what one wave instruction of the
  // loader fetches is then one contiguous KiB instead of eight 128-byte runs 6 K bytes apart (LDS-DMA from L2: 55 - 65 against 36 - 40 B / clk per CU,
  // profiles/r03_pmc_stall_parseq.txt section 4), the rows of a piece in the order the LDS image wants them
  const bool wtiled = p.wgt_tiled != nullptr;
  const __amdgpu_buffer_rsrc_t rsw = sp_rsrc(wtiled ? p.wgt_tiled : p.wgt, (unsigned)((size_t)((p.Cout + 31) / 32 * 32) * Kw * 6));
  // byte offsets of one activation plane and of one k0: row-major rows; the loader's 8-row pieces (x_tiled = 1); the producer's 16-row pieces (x_tiled = 2, below)
  const unsigned x_plstep = p.x_tiled == 2 ? (unsigned)(K >> 5) * 1024u : p.x_tiled ? (unsigned)(K >> 6) * 1024u : (unsigned)K * 2u;
  const unsigned x_kstep = p.x_tiled == 2 ? 2048u : p.x_tiled ? 1024u : 128u;
  const unsigned w_pl1 = wtiled ? (unsigned)(2 * (K >> 6)) * 1024u : (unsigned)Kw * 4u, w_kstep = wtiled ? 1024u : 128u;   // byte offsets of plane w1 and of one k0
  constexpr unsigned OOB = 0x80000000u;

  unsigned char* const xring = smem;
  unsigned char* const wring = smem + XST * C::XBYTES;

  // a lane's row offsets for local tile li: piece q = i * NW + wave covers tile rows 8q .. 8q + 7, this lane owns row 8q + (lane >> 3) and the
  // LDS chunk (lane & 7), which holds global chunk (lane & 7) ^ ((row >> 1) & 7)   (gemm2.hip's image)
  auto tile_offsets = [&](int li, unsigned (&xo)[C::XPW], unsigned (&wo)[C::WPW]) {
    const bool live = li < xcd_count;
    const int tile = xcd_first + li, tm = tile / tilesN, m0 = tm * BM, n0 = (tile - tm * tilesN) * BN;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const int row = (i * C::NW + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int m = m0 + row;
      // x_tiled = 2: [rows / 16][plane][channels / 32][4 chunks][16 rows][8 halves] - a KiB is what ONE store instruction of the producing epilogue writes, its 64
      // lanes in lane order (lane = row + 16 chunk: the MFMA accumulator layout as it stands), i.e. eight whole 128-byte lines per instruction with nothing to merge
      // in L2, so the stores can stream (nt) past the weights the tiles re-read.  This loader gathers its 8 rows x 8 chunks from two such KiB (eight 128-byte runs).
      if (p.x_tiled == 2) xo[i] = (live && m < p.M) ? ((unsigned)(m >> 4) * (unsigned)(PLX * (K >> 5)) + (unsigned)(g >> 2)) * 1024u + (unsigned)(((g & 3) * 16 + (m & 15)) * 16) : OOB;
      else if (p.x_tiled) xo[i] = (live && m < p.M) ? (unsigned)(m >> 3) * (unsigned)(PLX * (K >> 6)) * 1024u + (unsigned)((m & 7) * 128 + g * 16) : OOB;
      else xo[i] = (live && m < p.M) ? ((unsigned)m * (unsigned)(PLX * K) + g * 8) * 2u : OOB;
    }
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const int row = (j * C::NW + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int q16 = row & 15;
      const int nl = (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);   // channel held by that LDS row
      const int n = n0 + nl;
      if (wtiled) wo[j] = (live && n < p.Cout) ? (unsigned)((n0 + (row & ~7)) >> 3) * (unsigned)(3 * (K >> 6)) * 1024u + (unsigned)((row & 7) * 128 + g * 16) : OOB;
      else wo[j] = (live && n < p.Cout) ? ((unsigned)n * (unsigned)(3 * Kw) + g * 8) * 2u : OOB;
    }
  };

  // ---- the two loader streams: tiles in the order X0(0) X1(0) X0(1) ... / W0(0) W1(0) W0(1) ..., k0 running on into the workgroup's next
  // tile.  Stream state is scalar; xvo / wvo are the stream's current-tile row offsets, xdl / wdl the delta to its next tile's.
  unsigned xvo[C::XPW], wvo[C::WPW], xdl[C::XPW], wdl[C::WPW];
  int xs_k = 0, xs_pl = 0, xs_slot = 0;
  int ws_k = 0, ws_pl = 0, ws_slot = 0;
  // KS3: the X stream's own tile and tap; per piece the centre pixel's row offset (OOB past M or past the workgroup's tiles) and its (y << 16 | x)
  unsigned xcen[KS3 ? C::XPW : 1];
  int xyq[KS3 ? C::XPW : 1];
  int xs_tap = 0, xs_li = idx;
  auto x_set_tile = [&](int li) {
    const bool live = li < xcd_count;
    const int tile = xcd_first + li, tm = tile / tilesN, m0 = tm * BM;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const int row = (i * C::NW + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int m = m0 + row;
      const int mm = m < p.M ? m : 0;
      const int xq = mm % p.W, yq = (mm / p.W) % p.H;
      xyq[KS3 ? i : 0] = (yq << 16) | xq;
      xcen[KS3 ? i : 0] = (live && m < p.M) ? ((unsigned)m * (unsigned)(PLX * K) + g * 8) * 2u : OOB;
    }
  };
  auto x_set_tap = [&](int tap) {
    const int ty = tap / 3, dy = (ty - 1) * p.dil, dx = (tap - 3 * ty - 1) * p.dil;
    const unsigned shift = (unsigned)((dy * p.W + dx) * (PLX * K * 2));
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const int y = (xyq[KS3 ? i : 0] >> 16) + dy, x = (xyq[KS3 ? i : 0] & 0xffff) + dx;
      const bool ok = xcen[KS3 ? i : 0] != OOB && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      xvo[i] = ok ? xcen[KS3 ? i : 0] + shift : OOB;
    }
  };
  auto issue_x = [&]() {
    const unsigned soff = (unsigned)xs_pl * x_plstep + (unsigned)xs_k * x_kstep;
    unsigned char* sb = xring + xs_slot * C::XBYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) {
      const unsigned vo = (DBG & 4) ? OOB : (DBG & 8) ? xvo[i] - (xvo[i] != OOB ? xvo[i] / (1u << 20) * (1u << 20) : 0u) : xvo[i];   // (a local copy: hipcc's host pass fails to instantiate the kernel when the array element is passed directly; DBG 4: every request out of range, 8: every request inside the first MiB)
      if constexpr (!(DBG & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sb + i * C::NW * 1024), 16, vo, soff, 0, 0);
    }
    xs_slot = xs_slot == XST - 1 ? 0 : xs_slot + 1;
    const bool lastpl = xs_pl == PLX - 1;              // the last plane -> next k0
    const int k1 = xs_k + (lastpl ? 1 : 0);
    xs_pl = lastpl ? 0 : xs_pl + 1;
    if constexpr (KS3) {                               // xs_k counts the k steps inside a tap
      const bool tapwrap = k1 == nkt;
      xs_k = tapwrap ? 0 : k1;
      if (tapwrap) {                                   // (scalar: uniform over the workgroup)
        xs_tap = xs_tap == 8 ? 0 : xs_tap + 1;
        if (xs_tap == 0) { xs_li += J; x_set_tile(xs_li); }
        x_set_tap(xs_tap);
      }
    } else {
    const bool wrap = k1 == nk0;
    xs_k = wrap ? 0 : k1;
    const unsigned mask = wrap ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < C::XPW; ++i) xvo[i] += xdl[i] & mask;
    }
  };
  auto issue_w = [&]() {
    const unsigned soff = (ws_pl ? w_pl1 : 0u) + (unsigned)ws_k * w_kstep;
    unsigned char* sb = wring + ws_slot * C::WBYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) {
      const unsigned vo = (DBG & 4) ? OOB : wvo[j];
      if constexpr (!(DBG & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * C::NW * 1024), 16, vo, soff, 0, 0);
    }
    ws_slot = ws_slot == WST - 1 ? 0 : ws_slot + 1;
    const int k1 = ws_k + ws_pl;
    ws_pl ^= 1;
    const bool wrap = k1 == nk0;
    ws_k = wrap ? 0 : k1;
    const unsigned mask = wrap ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int j = 0; j < C::WPW; ++j) wvo[j] += wdl[j] & mask;
  };
  // deltas from the tile the streams are in (local index LI) to the one after it; once per tile, outside the K loop
#define TTR_SP_NEXT_DELTAS(LI)                                                   \
  {                                                                              \
    unsigned xa_[C::XPW], wa_[C::WPW], xb_[C::XPW], wb_[C::WPW];                 \
    tile_offsets((LI), xa_, wa_);                                                \
    tile_offsets((LI) + J, xb_, wb_);                                            \
    _Pragma("unroll") for (int i = 0; i < C::XPW; ++i) xdl[i] = xb_[i] - xa_[i]; \
    _Pragma("unroll") for (int j = 0; j < C::WPW; ++j) wdl[j] = wb_[j] - wa_[j]; \
  }

  // fragment addressing (gemm2.hip): row = tile-aligned base + (lane & 15), so (row >> 1) & 7 == (lane >> 1) & 7
  const int frag_lane = (lane & 15) * 128 + ((((lane >> 4)) ^ ((lane >> 1) & 7)) << 4);
  const int xfo[2] = {wm * C::TM * 128 + frag_lane, wm * C::TM * 128 + (frag_lane ^ 64)};
  const int wfo[2] = {wn * C::TN * 128 + frag_lane, wn * C::TN * 128 + (frag_lane ^ 64)};
  const int fg = lane >> 4, fr = lane & 15;
  int xr = 0, wr = 0;                     // ring slots the next fragment reads take
  f16x8 fx0[2][C::MI], fx1[2][C::MI], fw0[2][C::NJ], fw1[2][C::NJ];
  bool dbg_first = true;   // (DBG & 2: the first call of each reader still reads, so that the registers hold something)
  auto read_x = [&](f16x8 (&f)[2][C::MI]) {
    if constexpr ((DBG & 2) != 0) { if (!dbg_first) { asm volatile("" : "+v"(f[0][0]), "+v"(f[1][0])); return; } }
    const unsigned char* b = xring + xr * C::XBYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < C::MI; ++i) f[kk][i] = *reinterpret_cast<const f16x8*>(b + xfo[kk] + i * 2048);
    xr = xr == XST - 1 ? 0 : xr + 1;
  };
  auto read_w = [&](f16x8 (&f)[2][C::NJ]) {
    if constexpr ((DBG & 2) != 0) { if (!dbg_first) { asm volatile("" : "+v"(f[0][0]), "+v"(f[1][0])); return; } }
    const unsigned char* b = wring + wr * C::WBYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < C::NJ; ++j) f[kk][j] = *reinterpret_cast<const f16x8*>(b + wfo[kk] + j * 2048);
    wr = wr == WST - 1 ? 0 : wr + 1;
  };

  // ---- prologue: the loads the steady state would have issued before ph0 of the first k0, in its order
  tile_offsets(idx, xvo, wvo);
  TTR_SP_NEXT_DELTAS(idx)
  if constexpr (KS3) { x_set_tile(idx); x_set_tap(0); }
  // Requests per phase in steady state, by ring depth (a tile is requested as soon as the slot it takes has been read):
  //   X ring 3:  ph0 X1(k+1)   ph2 X0(k+2)        X ring 2:  ph0 X0(k+1)   ph2 X1(k+1)
  //   W ring 3:  ph0 W1(k+1)   ph1 W0(k+2)        W ring 2:  ph0 W0(k+1)   ph1 W1(k+1)
  // (ph0 requests X before W).  PH0 / PH1 / PH2 = loads that may still be in flight at that phase's barrier = those requested after the
  // youngest tile the phase's reads need (ph0: W1(k); ph1: X1(k); ph2: X0(k+1), W0(k+1)):
  //   (3, 3)  ... X1(k) W1(k) | W0(k+1) | X0(k+1) | X1(k+1) W1(k+1) | W0(k+2) | X0(k+2) ...     PH0 = W + X, PH2 = X + 2 W
  //   (3, 2)  ... X1(k) W0(k) | W1(k)   | X0(k+1) | X1(k+1) W0(k+1) | W1(k+1) | X0(k+2) ...     PH0 = X,     PH2 = W
  //   (2, 2)  ... X0(k) W0(k) | W1(k)   | X1(k)   | X0(k+1) W0(k+1) | W1(k+1) | X1(k+1) ...     PH0 = X, PH1 = X + W, PH2 = W
  constexpr int XW = C::XPW + C::WPW;
  constexpr int PH0 = WST == 3 ? XW : C::XPW;
  constexpr int PH1 = XST == 2 ? XW : 63;                 // (rings of 3: X1(k) was complete at ph0's wait)
  constexpr int PH2 = WST == 3 ? XW + C::WPW : C::WPW;
  // Triples (rings 3 + 2):  ... | W1(k) | X1(k) | X2(k) | X0(k+1) W0(k+1) | W1(k+1) | X1(k+1) | X2(k+1) | ...   (ph1, ph2, ph3, ph0, ...)
  //   ph0 reads W1(k): behind it X1(k) X2(k)                      ph1 reads X1(k): behind it X2(k) X0(k+1) W0(k+1)
  //   ph2 reads X2(k): behind it X0(k+1) W0(k+1) W1(k+1)          ph3 reads X0(k+1), W0(k+1): behind them W1(k+1) X1(k+1)
  constexpr int T0 = 2 * C::XPW, T1 = 2 * C::XPW + C::WPW, T2 = C::XPW + 2 * C::WPW, T3 = XW;
  issue_x(); issue_w();                    // X0(0), W0(0)
  if constexpr (NP == 4) {
    issue_w();                             // W1(0)                 (ph1 of k = -1)
    issue_x();                             // X1(0)                 (ph2)
    issue_x();                             // X2(0)                 (ph3)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::WPW + 2 * C::XPW) : "memory");
  } else if constexpr (XST == 3 && WST == 3) {
    issue_x(); issue_w();                  // X1(0), W1(0)          (as ph0 of k = -1)
    issue_w();                             // W0(1)                 (ph1)
    issue_x();                             // X0(1)                 (ph2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XW) : "memory");
  } else if constexpr (XST == 3) {
    issue_x();                             // X1(0)                 (ph0; W0(0) is out already)
    issue_w();                             // W1(0)                 (ph1)
    issue_x();                             // X0(1)                 (ph2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XW + C::XPW) : "memory");
  } else {
    issue_w();                             // W1(0)                 (ph1)
    issue_x();                             // X1(0)                 (ph2)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XW) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_x(fx0); read_w(fw0);
  if constexpr ((DBG & 2) != 0) { read_x(fx1); read_w(fw1); dbg_first = false; }

  int stamp_tile = 0;   // (diagnostics: phase stamps of workgroup 0, wave 0 over its first 24 tiles, ConvParams::dbg)
#define TTR_SP_STAMP(ph) do { if (EPI == 1 && p.dbg && blockIdx.x == 0 && tid == 0 && stamp_tile < 24) p.dbg[stamp_tile * 16 + (ph)] = __builtin_readcyclecounter(); } while (0)
  while (true) {
    TTR_SP_STAMP(0);
    if constexpr (EPI == 1) { if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp_tile < 24) p.dbg[stamp_tile * 16 + 11] = __builtin_amdgcn_s_memrealtime(); }   // (100 MHz: the shader clock the kernel ran at)
    f32x4 acc[C::NJ][C::MI];
#pragma unroll
    for (int j = 0; j < C::NJ; ++j)
#pragma unroll
      for (int i = 0; i < C::MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ctile = xcd_first + idx;
    const int m0c = (ctile / tilesN) * BM, n0c = (ctile % tilesN) * BN;

    // One phase: [wait, barrier] - requests and fragment reads (for the NEXT phase's MFMAs) - this phase's MFMAs.  The scheduling fences keep
    // the compiler from sinking the reads down to their consumers on the far side of the next barrier (which puts the LDS latency back in front
    // of the MFMAs) and from hoisting MFMAs over a barrier; inside a phase the reads and requests are spread over the first MFMAs.
    auto mfmas = [&](const f16x8 (&fw)[2][C::NJ], const f16x8 (&fx)[2][C::MI]) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
    };
    auto interleave = [&](auto reads_c, auto loads_c) {   // per MFMA at the head of the phase: one fragment read (two when there are 16), one request
      constexpr int READS = decltype(reads_c)::value, LOADS = decltype(loads_c)::value;
#define TTR_SP_GROUP(G)                                                                      \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, READS > 8 ? 2 : 1, 0);                     \
      if constexpr (G < LOADS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      TTR_SP_GROUP(0) TTR_SP_GROUP(1) TTR_SP_GROUP(2) TTR_SP_GROUP(3) TTR_SP_GROUP(4) TTR_SP_GROUP(5) TTR_SP_GROUP(6) TTR_SP_GROUP(7)
#undef TTR_SP_GROUP
    };
    auto scale_w0 = [&]() {   // w0s = w0 / 2^11 (exact unless subnormal: the values the staged w0b plane holds), into the registers W1 has left
      const f16 sc = (f16)(1.f / 2048.f);
      const f16x8 scv = {sc, sc, sc, sc, sc, sc, sc, sc};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) fw1[kk][j] = fw0[kk][j] * scv;
    };
    if constexpr (NP == 4) {
      auto phase_head = [&](auto allow_c) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(decltype(allow_c)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      // one k0 with X0 (then X2) in `a` and X1 (then the next k0's X0) in `b`
      auto k_step = [&](f16x8 (&a)[2][C::MI], f16x8 (&b)[2][C::MI]) {
        phase_head(std::integral_constant<int, T0>{});
        issue_x(); issue_w();              // X0(k+1), W0(k+1)
        read_w(fw1);                       // W1(k)
        mfmas(fw0, a);
        if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::NJ>{}, std::integral_constant<int, XW>{});
        phase_head(std::integral_constant<int, T1>{});
        issue_w();                         // W1(k+1)
        read_x(b);                         // X1(k)
        mfmas(fw1, a);
        if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::MI>{}, std::integral_constant<int, C::WPW>{});
        __builtin_amdgcn_sched_barrier(0);
        scale_w0();
        phase_head(std::integral_constant<int, T2>{});
        issue_x();                         // X1(k+1)
        read_x(a);                         // X2(k)
        mfmas(fw1, b);
        if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::MI>{}, std::integral_constant<int, C::XPW>{});
        phase_head(std::integral_constant<int, T3>{});
        issue_x();                         // X2(k+1)
        read_x(b); read_w(fw0);            // X0(k+1), W0(k+1)
        mfmas(fw1, a);
        if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::MI + 2 * C::NJ>{}, std::integral_constant<int, C::XPW>{});
      };
      for (int k0 = 0; k0 < nk0; k0 += 2) { k_step(fx0, fx1); k_step(fx1, fx0); }
    } else if constexpr (STAG) {
      // one code path for both halves: per barrier interval the early half (waves 0 - 3) runs [head of phase x: requests, reads, first 32-deep half of its MFMAs]
      // [second half of phase x], the late half [second half of phase x - 1] [head of phase x]
      const bool late = wave >= 4;
      auto mfmas_kk = [&](const f16x8 (&fw)[2][C::NJ], const f16x8 (&fx)[2][C::MI], int kk) {
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
          for (int j = 0; j < C::NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[kk][j], fx[kk][i], acc[j][i], 0, 0, 0);
      };
      auto scale_w0_kk = [&](int kk) {
        const f16 sc = (f16)(1.f / 2048.f);
        const f16x8 scv = {sc, sc, sc, sc, sc, sc, sc, sc};
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) fw1[kk][j] = fw0[kk][j] * scv;
      };
      auto head = [&](auto allow_c) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(decltype(allow_c)::value) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int k0 = 0; k0 < nk0; ++k0) {
        // ph0 interval
        head(std::integral_constant<int, PH0>{});
        if (late && k0 > 0) mfmas_kk(fw1, fx1, 1);      // (w0s X1 of the previous k0, second half)
        __builtin_amdgcn_sched_barrier(0);
        issue_x(); issue_w();
        read_w(fw1);                         // W1(k)
        mfmas_kk(fw0, fx0, 0);
        interleave(std::integral_constant<int, 2 * C::NJ>{}, std::integral_constant<int, XW>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!late) mfmas_kk(fw0, fx0, 1);
        // ph1 interval
        head(std::integral_constant<int, PH1>{});
        if (late) mfmas_kk(fw0, fx0, 1);
        __builtin_amdgcn_sched_barrier(0);
        issue_w();
        read_x(fx1);                         // X1(k)
        mfmas_kk(fw1, fx0, 0);
        interleave(std::integral_constant<int, 2 * C::MI>{}, std::integral_constant<int, C::WPW>{});
        __builtin_amdgcn_sched_barrier(0);
        scale_w0_kk(0);
        if (!late) { mfmas_kk(fw1, fx0, 1); __builtin_amdgcn_sched_barrier(0); scale_w0_kk(1); }
        // ph2 interval
        head(std::integral_constant<int, PH2>{});
        if (late) { mfmas_kk(fw1, fx0, 1); __builtin_amdgcn_sched_barrier(0); scale_w0_kk(1); }
        __builtin_amdgcn_sched_barrier(0);
        issue_x();
        read_x(fx0); read_w(fw0);            // X0(k+1), W0(k+1)
        mfmas_kk(fw1, fx1, 0);
        interleave(std::integral_constant<int, 2 * C::MI + 2 * C::NJ>{}, std::integral_constant<int, C::XPW>{});
        __builtin_amdgcn_sched_barrier(0);
        if (!late) mfmas_kk(fw1, fx1, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (late) mfmas_kk(fw1, fx1, 1);       // the last phase's deferred half
    } else {
    for (int k0 = 0; k0 < nk0; ++k0) {
      // ph0
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PH0) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_x(); issue_w();                // X1(k+1) [X0(k+1)], W1(k+1) [W0(k+1)]
      read_w(fw1);                         // W1(k)
      mfmas(fw0, fx0);
      if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::NJ>{}, std::integral_constant<int, XW>{});
      // ph1
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PH1) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_w();                           // W0(k+2) [W1(k+1)]
      read_x(fx1);                         // X1(k)
      mfmas(fw1, fx0);
      if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::MI>{}, std::integral_constant<int, C::WPW>{});
      __builtin_amdgcn_sched_barrier(0);
      scale_w0();
      // ph2
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PH2) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_x();                           // X0(k+2) [X1(k+1)]
      read_x(fx0); read_w(fw0);            // X0(k+1), W0(k+1)
      mfmas(fw1, fx1);
      if constexpr (SCHED) interleave(std::integral_constant<int, 2 * C::MI + 2 * C::NJ>{}, std::integral_constant<int, C::XPW>{});
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    TTR_SP_STAMP(1);                        // K loop done
    idx += J;
    const bool has_next = idx < xcd_count;
    TTR_SP_NEXT_DELTAS(idx)                 // the streams are inside tile idx now (nk0 >= 3: they run at most two k0 ahead)
    RangeWatch rw;                          // (split.h: the maximum of |x| over the values this lane writes as planes; per tile, so that nothing lives across the K loop)

    if (EM == 0 && (p.dbg_flags & 2)) {   // timing experiment: no epilogue at all
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < C::NJ; ++j)
#pragma unroll
        for (int i = 0; i < C::MI; ++i) sum += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
      if (sum == 1.2345e30f) reinterpret_cast<float*>(p.out ? p.out : (void*)p.out_f32)[0] = sum;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (!has_next) break;
      continue;
    }
    if constexpr (EPI == 1) {
      if (p.dbg_flags & 4) {   // timing experiment (results are wrong): the K loop alone - no attention, one predicated store keeps the accumulators alive
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < C::NJ; ++j)
#pragma unroll
          for (int i = 0; i < C::MI; ++i) sum += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
        if (sum == 1.2345e30f) reinterpret_cast<float*>(p.out)[0] = sum;
        if (!has_next) break;
        continue;
      }
      // ---- attention epilogue (timm Attention.forward inside the TorchScript module run at tuatara.cpp:307; attn_split.hip is the stand-alone
      // form): the tile is Q | K | V [128 rows][64] of one (crop, head), its 192 channels ordered so that BOTH waves of a row block hold the
      // same kinds: tile channel 96 wn + 32 t + dd is Q (t = 0), K (t = 1) or V (t = 2), d = 32 wn + dd.  A lane holds, of row 32 wm + 16 i + q,
      // the 8 channels 96 wn + 32 t + 8 g + e (e < 4 in acc[2t][i], e >= 4 in acc[2t+1][i]), i.e. d = 32 wn + 8 g + e: an MFMA operand fragment
      // of that row (k = 8 g + e) for the d half `wn`.  So:
      //   * wave (wm, wn) takes the 16 queries 32 wm + 16 wn + q: its own d half of their Q (exact triple) stays in registers, the other
      //     half comes from the partner wave through LDS as fragment images (and the partner's rows of this wave's half go the other way);
      //   * K (pair) goes to LDS in attn_split.hip's image (rows permuted so that S^T = K Q^T leaves 8 consecutive keys per lane = the P
      //     fragment of O^T = V^T P^T), V (pair) follows into the same bytes once every wave has its scores, with its d columns permuted so
      //     that O^T's accumulators hold 8 consecutive d per lane (16-byte stores of the output planes);
      //   * four MFMAs per product as there: (k0, q0) (k0/2^11, q1) (k0/2^11, q2) (k1/2^11, q0).
      // The rings are not touched: the loader streams run on into the next tile meanwhile.  No wave-dependent control flow.
      typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
      constexpr int KV = 16384;                                   // one plane of K or V: 128 rows x 128 B
      unsigned char* const sQ = smem + C::LDS;                    // [4 wm][2 reader wn][3 planes][64 lanes][16 B]
      unsigned char* const sK = sQ + 24576;                       // [2 planes][128 rows][128 B]; V afterwards
      const float osc = p.out_scale;
      // lane-constant LDS offsets (few, so that they stay in registers across the K loop): everything else is an immediate
      const int kR0 = wm * 32 + (((fr >> 2) & 1) << 4) + ((fr >> 3) << 2) + (fr & 3);   // LDS row of key 32 wm + q (key + 16: row + 8)
      const int kwr = kR0 * 128 + (((wn * 4 + fg) ^ ((kR0 >> 1) & 7)) << 4);           // K write, i = 0: chunk 4 wn + g at position chunk ^ ((row >> 1) & 7)
      const int krd = fr * 128 + (((wn * 4 + fg) ^ ((fr >> 1) & 7)) << 4);             // K fragment read, kt = 0, this wave's d half (the other: ^ 64)
      const int qwr = (wm * 2 + (1 - wn)) * 3072 + lane * 16, qrd = (wm * 2 + wn) * 3072 + lane * 16;   // Q exchange image: written for the partner, read as reader
      const int vwr = (wm * 32 + fr) * 128 + wn * 64 + fg * 8;                          // V write, i = 0
      const float* const bp = p.bias + n0c + wn * 96 + fg * 8;
      f16x8 fq[2][3];                                             // [own d half, other d half][plane]
      f16x8 vp[2][2];                                             // V pairs [plane][i]
      const f16 dn = (f16)(1.f / 2048.f);
      const f16x8 dnv = {dn, dn, dn, dn, dn, dn, dn, dn};
      // (K's and V's second planes go to LDS already scaled by 2^-11 - the factor their MFMA operand carries: one multiply per element where it is written
      // instead of one per reading wave and product, the same f16 product either way)
      auto tile_values = [&](int t, int i, float (&v)[8]) {       // bias + scale of the lane's 8 channels of block t, row half i
        const float4 b0 = *reinterpret_cast<const float4*>(bp + t * 32), b1 = *reinterpret_cast<const float4*>(bp + t * 32 + 4);
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[2 * t][i][e], osc, bv[e]); v[4 + e] = fmaf(acc[2 * t + 1][i][e], osc, bv[4 + e]); }
      };
      {   // Q: the rows of this wave's queries (i = wn) stay, the other 16 go to the partner
        float v0[8], v1[8], keep[8], send[8];
        tile_values(0, 0, v0); tile_values(0, 1, v1);
#pragma unroll
        for (int e = 0; e < 8; ++e) { keep[e] = wn ? v1[e] : v0[e]; send[e] = wn ? v0[e] : v1[e]; }
        split3_x8(keep, fq[0][0], fq[0][1], fq[0][2], rw);
        f16x8 a, b, c;
        split3_x8(send, a, b, c, rw);
        unsigned char* d = sQ + qwr;
        *reinterpret_cast<f16x8*>(d) = a; *reinterpret_cast<f16x8*>(d + 1024) = b; *reinterpret_cast<f16x8*>(d + 2048) = c;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // K: key 16 i + q of the wave's 32: i = 1 is 1024 bytes on with position bit 2 flipped
        float v[8];
        tile_values(1, i, v);
        f16x8 a, b;
        split2_x8(v, a, b, rw);
        unsigned char* d = sK + (kwr + i * 1024 ^ i * 64);
        *reinterpret_cast<f16x8*>(d) = a; *reinterpret_cast<f16x8*>(d + KV) = b * dnv;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // V: kept until every wave is through with K
        float v[8];
        tile_values(2, i, v);
        split2_x8(v, vp[0][i], vp[1][i], rw);
        vp[1][i] = vp[1][i] * dnv;
      }
      rw.flush(p.range_flag, p.range_tag);   // (here, not behind the tile: the watch would cost a register across the attention; the output rows are convex combinations of the V rows just watched)
      TTR_SP_STAMP(2);                      // Q / K / V converted, Q and K written
      __syncthreads();
      TTR_SP_STAMP(3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fq[1][pl] = *reinterpret_cast<const f16x8*>(sQ + qrd + pl * 1024);
      // S^T = K Q^T: sacc[kt], lane = query q, LDS key rows 16 kt + 4 g + r = keys 32 (kt >> 1) + 8 g + 4 (kt & 1) + r
      f32x4 sacc[8];
      {
        const unsigned char* const kb[2] = {sK + krd, sK + (krd ^ 64)};
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f16x8 k0 = *reinterpret_cast<const f16x8*>(kb[hf] + kt * 2048), k1 = *reinterpret_cast<const f16x8*>(kb[hf] + KV + kt * 2048);
            const f16x8 k0b = k0 * dnv;
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, fq[hf][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[hf][1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b, fq[hf][2], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, fq[hf][0], a, 0, 0, 0);
          }
          sacc[kt] = a;
        }
      }
      TTR_SP_STAMP(4);                      // S = Q K^T done
      // softmax over the 128 keys of a query: 32 values here, the rest in lanes q + 16 g'
      // (the probabilities as PAIRS: p in [0, 1] on 22+ bits is fp32's own resolution of it - three MFMAs per product in P V and a split of half the
      // instructions; the queries stay exact triples: a score's error is |q||k| times the operand's, and it goes through an exponential)
      f16x8 fp[2][4];                                             // [plane][32-key step]: keys 32 s + 8 g + e
      float rinv;
      {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        RangeWatch rp;                      // (dead: the probabilities are <= 1)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          float ev[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            ev[e] = __expf((sacc[2 * s4 + (e >> 2)][e & 3] - mx) * 0.125f);
            sum += ev[e];
          }
          split2_x8(ev, fp[0][s4], fp[1][s4], rp);
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        rinv = 1.0f / sum;
      }
      TTR_SP_STAMP(5);                      // softmax + probability planes done
      __syncthreads();                                            // every wave has read K
      TTR_SP_STAMP(6);
      // V: row = key, the 8 values d = 32 wn + 8 g + e go to column positions 32 wn + 4 g + e (e < 4) and 32 wn + 16 + 4 g + e - 4
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f16x8 v = vp[pl][i];
          unsigned char* d = sK + vwr + (pl * KV + i * 2048);
          *reinterpret_cast<f16x4*>(d) = f16x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f16x4*>(d + 32) = f16x4{v[4], v[5], v[6], v[7]};
        }
      __syncthreads();
      TTR_SP_STAMP(7);                      // V written + barrier
      // O^T = V^T P^T: A = V^T fragment (16 column positions x 32 keys) by transposed reads of the row-major planes (attn_split.hip)
      f32x4 oacc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        const unsigned vbase = (unsigned)(size_t)(lds_ptr)sK + (unsigned)((8 * fg + (fr >> 2)) * 128 + (fr & 3) * 8);
#define TTR_SPA_TR(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vbase), "n"(off))
#define TTR_SPA_STEP(s)                                                                                        \
        {                                                                                                      \
          f16x4 lo[2][4], hi[2][4];                                                                            \
          TTR_SPA_TR(lo[0][0], (s) * 4096 + 0);  TTR_SPA_TR(hi[0][0], (s) * 4096 + 512 + 0);                   \
          TTR_SPA_TR(lo[0][1], (s) * 4096 + 32); TTR_SPA_TR(hi[0][1], (s) * 4096 + 512 + 32);                  \
          TTR_SPA_TR(lo[0][2], (s) * 4096 + 64); TTR_SPA_TR(hi[0][2], (s) * 4096 + 512 + 64);                  \
          TTR_SPA_TR(lo[0][3], (s) * 4096 + 96); TTR_SPA_TR(hi[0][3], (s) * 4096 + 512 + 96);                  \
          TTR_SPA_TR(lo[1][0], KV + (s) * 4096 + 0);  TTR_SPA_TR(hi[1][0], KV + (s) * 4096 + 512 + 0);         \
          TTR_SPA_TR(lo[1][1], KV + (s) * 4096 + 32); TTR_SPA_TR(hi[1][1], KV + (s) * 4096 + 512 + 32);        \
          TTR_SPA_TR(lo[1][2], KV + (s) * 4096 + 64); TTR_SPA_TR(hi[1][2], KV + (s) * 4096 + 512 + 64);        \
          TTR_SPA_TR(lo[1][3], KV + (s) * 4096 + 96); TTR_SPA_TR(hi[1][3], KV + (s) * 4096 + 512 + 96);        \
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[0][2]), "+v"(lo[0][3]), "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[0][2]), "+v"(hi[0][3]), \
                       "+v"(lo[1][0]), "+v"(lo[1][1]), "+v"(lo[1][2]), "+v"(lo[1][3]), "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(hi[1][2]), "+v"(hi[1][3])); \
          _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                   \
            const f16x8 v0 = __builtin_shufflevector(lo[0][dt], hi[0][dt], 0, 1, 2, 3, 4, 5, 6, 7);            \
            const f16x8 v1 = __builtin_shufflevector(lo[1][dt], hi[1][dt], 0, 1, 2, 3, 4, 5, 6, 7);            \
            const f16x8 v0b = v0 * dnv;                                                                        \
            f32x4 a = oacc[dt];                                                                                \
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, fp[0][s], a, 0, 0, 0);                              \
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0b, fp[1][s], a, 0, 0, 0);                             \
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, fp[0][s], a, 0, 0, 0);                       \
          }                                                                                                    \
        }
        TTR_SPA_STEP(0)
        TTR_SPA_STEP(1)
        TTR_SPA_STEP(2)
        TTR_SPA_STEP(3)
#undef TTR_SPA_STEP
#undef TTR_SPA_TR
      }
      TTR_SP_STAMP(8);                      // P V done
      // out planes [M][3][384]: oacc[2u], oacc[2u+1] hold d = 32 u + 8 g + 0..7 of query q
      __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), in front of the first store: the prefetched tiles have landed (requested an epilogue ago)
      TTR_SP_STAMP(9);
      {
        const int orow = m0c + wm * 32 + wn * 16 + fr, ohead = n0c / 192;
        // (out_tiled: the projection GEMM's loader pieces - a head's 64 channels are one 64-deep k step; planes 6 KiB apart instead of 384 halves)
        f16* const op = p.out_tiled ? reinterpret_cast<f16*>(p.out) + ((int64_t)(orow >> 3) * 18 + ohead) * 512 + (orow & 7) * 64 + fg * 8
                                    : reinterpret_cast<f16*>(p.out) + (int64_t)orow * (3 * 384) + ohead * 64 + fg * 8;
        const int opl = p.out_tiled ? 6 * 512 : 384;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = oacc[2 * u][e] * rinv; v[4 + e] = oacc[2 * u + 1][e] * rinv; }
          f16x8 a, b, c;
          RangeWatch ro;                     // (dead)
          split3_x8(v, a, b, c, ro);
          *reinterpret_cast<f16x8*>(op + u * 32) = a; *reinterpret_cast<f16x8*>(op + opl + u * 32) = b; *reinterpret_cast<f16x8*>(op + 2 * opl + u * 32) = c;
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      TTR_SP_STAMP(10);                     // output planes stored and drained
      ++stamp_tile;
      if (!has_next) break;
      // The next tile's first fragments were read in the last phase above, like every k0's; this epilogue needs their 64 registers, so they are
      // read AGAIN here (their ring slots are untouched: nothing was requested meanwhile) and the first copies die at the top of the epilogue.
      xr = xr == 0 ? XST - 1 : xr - 1; wr = wr == 0 ? WST - 1 : wr - 1;
      read_x(fx0); read_w(fw0);
      continue;
    }
    // ---- epilogue: lane holds channels n = nb + 32t + (lane>>4)*8 + 0..7 of row m = mb + 16i + (lane&15).  Two passes.  Pass 1 issues EVERY
    // vector-memory load of the epilogue (bias, residual rows: buffer loads, out-of-range lanes read zeros, no branches) and folds them into the
    // accumulators in place; pass 2 (activation, planes, stores) loads nothing.  Mixed, a load behind a store costs a full drain of the stores:
    // the counter counts both, they complete out of order with respect to each other, so the compiler can only wait for vmcnt(0) - the
    // one-pass form did that once per 16-row block, eight write round trips per tile.
    constexpr bool FC1 = EM == 1 || EM == 3, RES = EM == 2 || EM == 4;
    const int e_act = FC1 ? (int)kActGelu : RES ? (int)kActNone : p.act;
    const bool e_lut = FC1 ? true : RES ? false : p.gelu_lut != nullptr;
    const bool e_resid = FC1 ? false : RES ? true : p.resid != nullptr;
    {
      const __amdgpu_buffer_rsrc_t rsb = sp_rsrc(p.bias, p.bias ? (unsigned)p.Cout * 4u : 0u);
      const int64_t rrows = p.resid_mod ? p.resid_mod : p.M;
      const __amdgpu_buffer_rsrc_t rsr = sp_rsrc(p.resid, e_resid ? (unsigned)(rrows * p.resid_ld * 4) : 0u);
      unsigned rrow[C::MI];
#pragma unroll
      for (int i = 0; i < C::MI; ++i) {
        const int m = m0c + wm * C::TM + i * 16 + fr;
        rrow[i] = m < p.M ? (unsigned)(p.resid_mod ? m % p.resid_mod : m) * (unsigned)p.resid_ld * 4u : OOB;
      }
      const float osc = p.out_scale;
#pragma unroll
      for (int t = 0; t < C::NJ / 2; ++t) {   // per 32-channel block: its loads in flight together, then the sums
        const int n = n0c + wn * C::TN + t * 32 + fg * 8;
        const unsigned ncol = n < p.Cout ? (unsigned)n * 4u : OOB;
        const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, ncol, 0, 0));
        const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, ncol, 16, 0));
        if (e_resid) {
          f32x4 rv[C::MI][2];
#pragma unroll
          for (int i = 0; i < C::MI; ++i) {
            const unsigned ro = (ncol != OOB && rrow[i] != OOB) ? rrow[i] + ncol : OOB;
            rv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, ro, 0, 0));
            rv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, ro, 16, 0));
          }
#pragma unroll
          for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[2 * t][i][e] = fmaf(acc[2 * t][i][e], osc, b0[e]) + rv[i][0][e];
              acc[2 * t + 1][i][e] = fmaf(acc[2 * t + 1][i][e], osc, b1[e]) + rv[i][1][e];
            }
        } else {
#pragma unroll
          for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[2 * t][i][e] = fmaf(acc[2 * t][i][e], osc, b0[e]);
              acc[2 * t + 1][i][e] = fmaf(acc[2 * t + 1][i][e], osc, b1[e]);
            }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the loads above (and the K loop's prefetch, requested a phase and more ago); no store is out yet
    // EM = 3 / 4: the same values stored as WHOLE lines.  In the accumulator layout a store instruction covers 16 rows x 64 bytes (four lanes side by side) - sixteen
    // half lines - and a CU's store path takes such instructions at 14 - 16 B / clk against 37 - 53 for instructions that write whole 128-byte lines
    // (tools/micro/store_rate.hip, profiles/r06_store_rate.txt: a 128-KiB burst per CU).  Lanes fr and fr ^ 8 of a 16-lane row exchange one of their two 16-byte
    // pieces (DPP row_ror:8: three vector instructions per dword) so that an instruction carries rows 0 - 7 (then 8 - 15) of the block in full.
    auto swap8 = [&](const f16x8& lo_piece, const f16x8& hi_piece, f16x8& first, f16x8& second) {
      // lanes fr < 8 keep lo_piece for the first instruction and hand hi_piece to lane fr + 8 (second instruction); lanes fr >= 8 the other way round
      typedef __attribute__((ext_vector_type(4))) int i32x4;
      const bool hi = (lane & 8) != 0;
      const i32x4 l = __builtin_bit_cast(i32x4, lo_piece), h = __builtin_bit_cast(i32x4, hi_piece);
      i32x4 f, g;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int send = hi ? l[d] : h[d];
        const int recv = __builtin_amdgcn_update_dpp(0, send, 0x128, 0xF, 0xF, false);   // row_ror:8
        f[d] = hi ? recv : l[d];
        g[d] = hi ? h[d] : recv;
      }
      first = __builtin_bit_cast(f16x8, f); second = __builtin_bit_cast(f16x8, g);
    };
    if constexpr (EM == 3) {   // fc1: GELU, pairs, the next GEMM's 1-KiB loader pieces - every store instruction one contiguous KiB (launcher: M % 16 == 0, Cout % 64 == 0, out_tiled == 1)
      const int kb = p.out_ld >> 6;
#pragma unroll
      for (int i = 0; i < C::MI; ++i) {
        const int mrow = m0c + wm * C::TM + i * 16 + (fr & 7);          // first instruction's row; the second's is 8 further down (the next piece)
#pragma unroll
        for (int tp = 0; tp < C::NJ / 4; ++tp) {
          f16x8 a[2], b[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int t = 2 * tp + hf;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[2 * t][i][e]; v[4 + e] = acc[2 * t + 1][i][e]; }
            gelu_hermite8(v, glut);
            split2_x8(v, a[hf], b[hf], rw);
          }
          f16x8 a0, a1, b0, b1;
          swap8(a[0], a[1], a0, a1); swap8(b[0], b[1], b0, b1);
          const int n64 = n0c + wn * C::TN + tp * 64;
          f16* o = reinterpret_cast<f16*>(p.out) + ((int64_t)(mrow >> 3) * (2 * kb) + (n64 >> 6)) * 512 + (mrow & 7) * 64 + (fg + ((lane & 8) ? 4 : 0)) * 8;
          if (mrow < p.M) {
            *reinterpret_cast<f16x8*>(o) = a0; *reinterpret_cast<f16x8*>(o + kb * 512) = b0;
            *reinterpret_cast<f16x8*>(o + 2 * kb * 512) = a1; *reinterpret_cast<f16x8*>(o + 3 * kb * 512) = b1;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if constexpr (EM == 4) {   // residual linears: fp32 rows, every store instruction eight whole 128-byte lines (launcher: M % 16 == 0, Cout % 32 == 0)
#pragma unroll
      for (int i = 0; i < C::MI; ++i) {
        const int mrow = m0c + wm * C::TM + i * 16 + (fr & 7);
#pragma unroll
        for (int t = 0; t < C::NJ / 2; ++t) {
          const f16x8 lo = __builtin_bit_cast(f16x8, acc[2 * t][i]), hi = __builtin_bit_cast(f16x8, acc[2 * t + 1][i]);   // (16 bytes each: the lane's channels 0 - 3 and 4 - 7 of the block)
          f16x8 s0, s1;
          swap8(lo, hi, s0, s1);
          const int n = n0c + wn * C::TN + t * 32 + fg * 8 + ((lane & 8) ? 4 : 0);
          float* o = p.out_f32 + (int64_t)mrow * p.out_f32_ld + n;
          if (mrow < p.M) {
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, s0), reinterpret_cast<f32x4*>(o));
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, s1), reinterpret_cast<f32x4*>(o + 8 * (int64_t)p.out_f32_ld));
          }
        }
      }
    } else
    // (rows outside, channel blocks inside: the 64-byte pieces a row's blocks contribute to one 128-byte line leave back to back)
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
      const int m = m0c + wm * C::TM + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int t = 0; t < C::NJ / 2; ++t) {
        const int n = n0c + wn * C::TN + t * 32 + fg * 8;
        if (n >= p.Cout) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[2 * t][i][e]; v[4 + e] = acc[2 * t + 1][i][e]; }
        if constexpr (EM == 1) {          // GELU (the block's eight table reads in flight together), then tiled pairs: the next GEMM's loader pieces
          gelu_hermite8(v, glut);
          f16x8 a, b;
          split2_x8(v, a, b, rw);
          if (p.out_tiled == 2) {         // 16-row pieces in lane order (see the loader): this instruction's 64 lanes write one contiguous KiB
            const int kb = p.out_ld >> 5;
            f16* o = reinterpret_cast<f16*>(p.out) + ((int64_t)(m >> 4) * (2 * kb) + (n >> 5)) * 512 + ((n >> 3) & 3) * 128 + (m & 15) * 8;
            if (p.store_policy == 1) { __builtin_nontemporal_store(a, reinterpret_cast<f16x8*>(o)); __builtin_nontemporal_store(b, reinterpret_cast<f16x8*>(o + kb * 512)); }
            else { *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + kb * 512) = b; }
          } else {
          const int kb = p.out_ld >> 6;
          f16* o = reinterpret_cast<f16*>(p.out) + ((int64_t)(m >> 3) * (2 * kb) + (n >> 6)) * 512 + (m & 7) * 64 + (n & 63);
          *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + kb * 512) = b;
          }
          __builtin_amdgcn_sched_barrier(0);   // block by block: hoisting every block's table reads to the top costs 256 registers (spills)
          continue;
        } else if constexpr (EM == 2) {   // fp32 rows
          sp_store_f32x8(p.out_f32 + (int64_t)m * p.out_f32_ld + n, v);
          continue;
        }
        if (e_act == kActRelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (e_act == kActGelu) {
          if (e_lut) gelu_hermite8(v, glut);
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_exact(v[e]);
          }
        }
        if (p.dbg_flags & 1) { if (v[0] == 1.2345e30f) reinterpret_cast<float*>(p.out)[0] = v[1]; continue; }   // timing experiment: no output stores
        if (p.out) {
          if (p.out_planes == 3 && p.out_full_cols > 0 && n >= p.out_full_cols) {   // a triple whose third plane nobody reads (ConvParams::out_full_cols)
            f16x8 a, b, c;
            split3_x8(v, a, b, c, rw);
            f16* o = reinterpret_cast<f16*>(p.out) + (int64_t)m * (3 * (int64_t)p.out_ld) + n;
            *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + p.out_ld) = b;
          } else if (p.out_planes && p.out_tiled) {   // the next GEMM's loader pieces (8-row pieces; out_tiled = 2: 16-row pieces in lane order)
            const int kb = p.out_tiled == 2 ? p.out_ld >> 5 : p.out_ld >> 6;
            f16* o = p.out_tiled == 2 ? reinterpret_cast<f16*>(p.out) + ((int64_t)(m >> 4) * (p.out_planes * kb) + (n >> 5)) * 512 + ((n >> 3) & 3) * 128 + (m & 15) * 8
                                      : reinterpret_cast<f16*>(p.out) + ((int64_t)(m >> 3) * (p.out_planes * kb) + (n >> 6)) * 512 + (m & 7) * 64 + (n & 63);
            f16x8 a, b, c;
            if (p.out_planes == 3) { split3_x8(v, a, b, c, rw); *reinterpret_cast<f16x8*>(o + 2 * kb * 512) = c; }
            else split2_x8(v, a, b, rw);
            *reinterpret_cast<f16x8*>(o) = a; *reinterpret_cast<f16x8*>(o + kb * 512) = b;
          } else if (p.out_planes) st_split_n(p.out, (int64_t)m, p.out_ld, n, v, p.out_planes, rw);
          else sp_store_f32x8(reinterpret_cast<float*>(p.out) + (int64_t)m * p.out_ld + n, v);
        }
        if (p.out_f32) sp_store_f32x8(p.out_f32 + (int64_t)m * p.out_f32_ld + n, v);
      }
    }
    // The stores drain here (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait): the K loop's counted waits would take them for loads in flight.  (Leaving them
    // in flight under the next tile's first k0 - what that k0 reads was requested before the epilogue - was measured: no gain, profiles/r03_pmc_stall_parseq.txt §3.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    rw.flush(p.range_flag, p.range_tag);
    if (!has_next) break;
  }   // (the streams' trailing out-of-range loads, which target this workgroup's LDS, have landed: the wait above)
}

#undef TTR_SP_NEXT_DELTAS
#undef TTR_SP_STAMP

static int g_sp_stagger = 0, g_sp_stagger_groups = 2;   // ConvParams::cu_stagger (ticks of 10 ns) and its group count
void set_gemm_sp_stagger(int v) { g_sp_stagger = v; }
void set_gemm_sp_stagger_groups(int v) { g_sp_stagger_groups = v < 1 ? 1 : v; }

template <int BM, int BN, int WM, int WN, int XST, int WST, int MINB, bool SCHED, int NP = 3, int EPI = 0, int EM = 0, bool KS3 = false, int DBG = 0, bool STAG = false>
static void launch_sp(const ConvParams& p_in, hipStream_t s) {
  using C = SpCfg<BM, BN, WM, WN, XST, WST>;
  constexpr int TABLE = EPI == 1 ? 57344 : 8208;   // behind the rings: the GELU table, or the attention epilogue's Q / K / V images
  ConvParams p = with_range_ctx(p_in);
  p.cu_stagger = g_sp_stagger; p.cu_stagger_groups = g_sp_stagger_groups;
  if ((size_t)(C::LDS + TABLE) * MINB > 160 * 1024) p.gelu_lut = nullptr;   // no room for the table beside these rings: erf
  static PerDeviceOnce once;
  static_assert(EPI == 0 || C::LDS + TABLE <= 160 * 1024, "attention epilogue: rings + images must fit the LDS");
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_sp_kernel<BM, BN, WM, WN, XST, WST, MINB, SCHED, NP, EPI, EM, KS3, DBG, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, std::min(C::LDS + TABLE, 160 * 1024))); });
  const size_t lds = C::LDS + (EPI == 1 || (p.act == kActGelu && p.gelu_lut) ? TABLE : 0);
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const int cap = device_cu_count(256) * MINB / 8 * 8;
  const int grid = std::min((tilesM * tilesN + 7) / 8 * 8, std::max(cap, 8));
  hipLaunchKernelGGL((gemm_sp_kernel<BM, BN, WM, WN, XST, WST, MINB, SCHED, NP, EPI, EM, KS3, DBG, STAG>), dim3(grid), dim3(C::NT), lds, s, p);
}

static int g_qkv_attn_dbg = 0;   // timing experiments on the fused qkv + attention launch (results are wrong): 4 = the K loop alone
void set_qkv_attn_dbg(int v) { g_qkv_attn_dbg = v; }
static unsigned long long* g_qkv_attn_stamps = nullptr;
void set_qkv_attn_stamps(unsigned long long* d) { g_qkv_attn_stamps = d; }
static int g_sp_few = 1;   // the few-tile rules of launch_gemm_sp (a page's worth of rows)
void set_gemm_sp_few(int v) { g_sp_few = v; }
static int g_sp_epi = 3;   // (+ 16: fc1's stores as whole KiB pieces, EM = 3; + 32: the residual linears' fp32 rows as whole lines, EM = 4 - also on fc2's tile)
static int g_sp_epi_unused = 3;     // bits: 1 = fc1's case on its own kernels (EM = 1), 2 = the residual linears' (EM = 2) on the 256 x 128 triples tile (proj), 8 = on the other 128- / 256-row tiles, 4 = on the 64-row tiles; 0: the general kernel everywhere
void set_gemm_sp_epi(int v) { g_sp_epi = v; }
static int g_sp_stag = 0;    // 1: waves 4 - 7 of fc1's eight-wave tile half a phase behind (gemm_sp_kernel's STAG: measured 4 - 5 % SLOWER, profiles/r06_parseq_kloop_experiments.txt); 0: every wave in step
void set_gemm_sp_stag(int v) { g_sp_stag = v; }
static int g_sp_dbg = 0;     // gemm_sp_kernel's DBG on fc1's and fc2's instances (timing experiments, wrong results): 1 = no loader requests, 2 = no fragment reads, 3 = neither
void set_gemm_sp_dbg(int v) { g_sp_dbg = v; }
static int g_sp_sched = 1;   // 1: fragment reads and requests interleaved with the first MFMAs of a phase; 0: in front of them
void set_gemm_sp_sched(int v) { g_sp_sched = v; }

// shapes these kernels take (gemm2.hip's split mode keeps the rest): ks = 1, one source, no pooled / ReLU-copy outputs, K a multiple of 64;
// pairs: K >= 192; triples: K >= 128 and K / 64 even (the K loop runs in pairs of k0)
bool gemm_sp_eligible(const ConvParams& p) {
  if (p.up_z) return false;   // (the half-resolution addend is an epilogue of gemm2.hip's split loop)
  if (p.out_tiled && (!p.out_planes || p.out_ld % 64 != 0 || p.out_full_cols)) return false;   // (tiled planes: whole 64-channel blocks, every plane)
  if ((p.out_tiled == 2 || p.x_tiled == 2) && p.M % 16 != 0) return false;                     // (16-row pieces)
  if ((p.split != 3 && p.split != 4) || p.ks != 1 || p.C1 != 0 || p.out_pool || p.out_relu || p.C0 % 64 != 0 || p.Cout % 8 != 0) return false;
  if (p.split == 3 ? p.C0 < 192 : (p.C0 < 128 || (p.C0 >> 6) % 2 != 0)) return false;
  if (p.resid && (size_t)(p.resid_mod ? p.resid_mod : p.M) * p.resid_ld * 4 >= ((size_t)1 << 31)) return false;   // the epilogue reads the residual through a buffer descriptor
  return (size_t)p.M * p.C0 * (p.split == 3 ? 4 : 6) < ((size_t)1 << 31) && (size_t)p.Cout * p.C0 * 6 < ((size_t)1 << 31);
}

// 3x3 taps (any dilation) on the pairs loop (gemm_sp_kernel's KS3): row-major planes in, single source, no pooled / ReLU-copy / tiled outputs
static int g_sp_ks3 = 1;   // 0: gemm2.hip's loop; 1: this file's (tile by launch_gemm_sp_ks3); 3: ... never the 64-row tile
void set_gemm_sp_ks3(int v) { g_sp_ks3 = v; }
bool gemm_sp_ks3_eligible(const ConvParams& p) {
  if (!g_sp_ks3 || p.split != 3 || p.ks != 3 || p.dil < 1 || p.C1 != 0 || p.out_pool || p.out_relu || p.C0 % 64 != 0 || p.Cout % 8 != 0) return false;
  if (p.x_tiled || p.out_tiled || p.wgt_tiled || p.up_z || p.resid || p.out_full_cols || p.skip) return false;
  if (p.M != p.B * p.H * p.W || p.H >= 32768 || p.W >= 65536) return false;
  return (size_t)p.M * p.C0 * 4 < ((size_t)1 << 31) && (size_t)((p.Cout + 31) / 32 * 32) * 9 * p.C0 * 6 < ((size_t)1 << 31);
}
void launch_gemm_sp_ks3(const ConvParams& p, int cfg, hipStream_t s) {
  if (!gemm_sp_ks3_eligible(p)) throw std::runtime_error("gemm_sp (3x3 taps): shape not supported");
  const int tiles128 = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
  if (cfg == 2) launch_sp<256, 128, 4, 2, 3, 3, 1, true, 3, 0, 0, true>(p, s);
  // (a page - no more 128 x 128 tiles than CUs -: 64-row tiles on eight waves, as the 1x1 layers'; slice5.1 of one page 157 -> ~105 us.  g_sp_ks3 = 3 keeps the 128-row tile)
  else if (g_sp_few && g_sp_ks3 != 3 && tiles128 <= device_cu_count(256)) launch_sp<64, 128, 2, 4, 3, 3, 1, true, 3, 0, 0, true>(p, s);
  else launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 0, true>(p, s);
}

// cfg: 2 = 256 x 128 tiles, 6 = 128 x 256 (one workgroup of 8 waves per CU), 3 = 128 x 128 (two workgroups of 4 waves per CU: one's epilogue -
// its stores have to drain before its next tile's first s_waitcnt vmcnt - runs under the other's MFMAs).  Rings: pairs 3 + 3 on the big tiles,
// 3 + 2 (2 + 2 beside the GELU table) on the small one; triples 3 + 2 everywhere.  The caller has run gemm2_check and filled p.gelu_lut.
void launch_gemm_sp(const ConvParams& p, int cfg, hipStream_t s) {
  if (!gemm_sp_eligible(p)) throw std::runtime_error("gemm_sp: shape not supported");
  const bool sched = g_sp_sched != 0;
  const bool table = p.act == kActGelu && p.gelu_lut;
  const int cus = device_cu_count(256);
  const int tiles128 = ((p.M + 127) / 128) * ((p.Cout + 127) / 128);
  // A page's worth of rows: fewer 128 x 128 tiles than half the CUs.  A workgroup alone on its CU walks K at the rate its tiles arrive (~40 B / clk
  // per CU for these 8-row pieces: 32 KB per phase against 512 cycles of MFMAs), so 64-row tiles on twice as many CUs shorten every phase of the
  // chain: fc2 at 40 crops (K = 1536, 120 tiles) 44 -> 31 us, and eight waves (wave tiles of 32 x 32) issue the loader's pieces at 1.4 - 1.6 x the rate of
  // four (tools/micro/dma_depth.hip): a page's recogniser -70 us more.  (Deeper rings alone - 3 + 3 on the 128-row tile - changed nothing: not a latency.)
  const bool few = cfg == 3 && g_sp_few && 2 * tiles128 <= cus;
  // Beyond that, up to two rounds of 128-row tiles (a page of 43 - 128 crops: proj / fc2 on 129 - 384 tiles): a lone workgroup's time is the operand rows its CU
  // pulls in, rounds x (BM + BN) per k step, so 96-row tiles win where a quarter more workgroups need no extra round - 60 crops: 180 -> 240 tiles, one round of
  // 224 rows instead of 256 (fc2 46.8 -> 41.4 us, proj 24.5 -> 20.6); 100 crops: 300 -> 402 tiles, two rounds either way.  (g_sp_few = 2: without this rule.)
  const int tiles96 = ((p.M + 95) / 96) * ((p.Cout + 127) / 128);
  const int rounds128 = (tiles128 + cus - 1) / cus, rounds96 = (tiles96 + cus - 1) / cus;
  const bool mid = cfg == 3 && g_sp_few == 1 && !few && tiles128 <= 4 * cus && rounds96 * 224 < rounds128 * 256 && !(p.act == kActGelu && p.gelu_lut);
  // the encoder's and the decoder's recurring epilogue cases as kernels of their own (gemm_sp_kernel's EM): fc1 (GELU table, tiled pairs out) and the residual
  // linears (bias + residual, fp32 rows out).  Same arithmetic, same order: bit-identical to the general kernel (tests/test_gpu_split_gemm.py; g_sp_epi = 0 turns them off)
  const bool plain = sched && p.dbg_flags == 0;
  const bool em_fc1 = plain && (g_sp_epi & 1) && table && !p.resid && p.out && p.out_planes == 2 && p.out_tiled && !p.out_full_cols && !p.out_f32;
  // (measured at 1280 crops, same box, twice: fc1 770 -> 670 us per layer; proj on the 256 x 128 triples tile 255 -> 242; fc2 on two 128 x 128 workgroups per CU
  // 555 -> 572 - its fixed case keeps 15 registers in scratch where the general kernel parks scalars in lanes - so that one stays on the general kernel; a page's
  // 64-row tiles: no difference)
  const bool em_res = plain && (g_sp_epi & (few ? 4 : (p.split == 4 && cfg == 2) ? 2 : 8)) && p.act == kActNone && p.resid && !p.out && p.out_f32;
  const bool lines_fc1 = em_fc1 && (g_sp_epi & 16) && p.out_tiled == 1 && p.M % 16 == 0 && p.Cout % 64 == 0;
  const bool lines_res = plain && (g_sp_epi & 32) && p.act == kActNone && p.resid && !p.out && p.out_f32 && p.M % 16 == 0 && p.Cout % 32 == 0 && !few;
  if (p.split == 4) {
    if (few) { if (em_res) launch_sp<64, 128, 2, 4, 3, 2, 1, true, 4, 0, 2>(p, s); else launch_sp<64, 128, 2, 4, 3, 2, 1, true, 4>(p, s); return; }
    if (mid) { launch_sp<96, 128, 2, 2, 3, 2, 2, true, 4>(p, s); return; }
    if (cfg == 6) launch_sp<128, 256, 2, 4, 3, 2, 1, true, 4>(p, s);
    else if (cfg == 2) { if (lines_res) launch_sp<256, 128, 4, 2, 3, 2, 1, true, 4, 0, 4>(p, s); else if (em_res) launch_sp<256, 128, 4, 2, 3, 2, 1, true, 4, 0, 2>(p, s); else launch_sp<256, 128, 4, 2, 3, 2, 1, true, 4>(p, s); }
    else { if (lines_res) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 4, 0, 4>(p, s); else if (em_res) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 4, 0, 2>(p, s); else launch_sp<128, 128, 2, 2, 3, 2, 2, true, 4>(p, s); }
    return;
  }
  // (64 x 64 tiles, two per CU: the same 31 us - the CU's fill rate, not the workgroup's)
  if (few) { if (em_res) launch_sp<64, 128, 2, 4, 3, 3, 1, true, 3, 0, 2>(p, s); else launch_sp<64, 128, 2, 4, 3, 3, 1, true>(p, s); return; }
  if (mid) { launch_sp<96, 128, 2, 2, 3, 2, 2, true>(p, s); return; }
  // (a wide layer on one round of 128 x 256 tiles instead - fc1 at 40 crops: 240 - is no faster: 31 -> 32 - 37 us)
  if (cfg == 6 && em_fc1 && g_sp_dbg) {
    if (g_sp_dbg == 1) launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 1, false, 1>(p, s); else if (g_sp_dbg == 4) launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 1, false, 4>(p, s); else launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 1, false, 8>(p, s);
    return;
  }
  if (cfg != 6 && cfg != 2 && !table && !em_res && sched && g_sp_dbg) {
    if (g_sp_dbg == 1) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 0, false, 1>(p, s); else if (g_sp_dbg == 4) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 0, false, 4>(p, s); else launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 0, false, 8>(p, s);
    return;
  }
  if (cfg == 6 && em_fc1 && !lines_fc1 && g_sp_stag) { launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 1, false, 0, true>(p, s); return; }
  if (cfg == 6) { if (lines_fc1) launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 3>(p, s); else if (em_fc1) launch_sp<128, 256, 2, 4, 3, 3, 1, true, 3, 0, 1>(p, s); else if (sched) launch_sp<128, 256, 2, 4, 3, 3, 1, true>(p, s); else launch_sp<128, 256, 2, 4, 3, 3, 1, false>(p, s); }
  else if (cfg == 2) { if (em_fc1) launch_sp<256, 128, 4, 2, 3, 3, 1, true, 3, 0, 1>(p, s); else if (sched) launch_sp<256, 128, 4, 2, 3, 3, 1, true>(p, s); else launch_sp<256, 128, 4, 2, 3, 3, 1, false>(p, s); }
  else if (table) { if (em_fc1) launch_sp<128, 128, 2, 2, 2, 2, 2, true, 3, 0, 1>(p, s); else if (sched) launch_sp<128, 128, 2, 2, 2, 2, 2, true>(p, s); else launch_sp<128, 128, 2, 2, 2, 2, 2, false>(p, s); }
  else { if (lines_res && mid == false) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 4>(p, s); else if (em_res) launch_sp<128, 128, 2, 2, 3, 2, 2, true, 3, 0, 2>(p, s); else if (sched) launch_sp<128, 128, 2, 2, 3, 2, 2, true>(p, s); else launch_sp<128, 128, 2, 2, 3, 2, 2, false>(p, s); }
}

// The encoder's qkv projection + self-attention as ONE launch (EPI = 1 above).  x_pairs: LayerNorm output as f16 pairs [N * 128][2][384];
// w_planes: the qkv weight planes [1152][3][384] with the rows in HEAD-MAJOR order (row 192 h + 64 c + d = upstream row 384 c + 64 h + d,
// c = 0 / 1 / 2 for Q / K / V), bias likewise; out: attention output as exact triples [N * 128][3][384] (the projection GEMM's input).
void launch_qkv_attn_split(const void* x_pairs, const void* w_planes, const float* bias, float inv_scale, void* out_planes, int N, hipStream_t s, const void* w_tiled,
                           int x_tiled, int out_tiled) {
  if (N <= 0) return;
  if (qkv_attn4_enabled()) return launch_qkv_attn4(x_pairs, w_planes, bias, inv_scale, out_planes, N, s, w_tiled, x_tiled, out_tiled);   // (experiment, off by default)
  if (((uintptr_t)x_pairs | (uintptr_t)w_planes | (uintptr_t)bias | (uintptr_t)out_planes) & 15) throw std::runtime_error("qkv_attn_split: operands must be 16-byte aligned");
  if ((size_t)N * 128 * 384 * 6 >= ((size_t)1 << 31)) throw std::runtime_error("qkv_attn_split: too many crops for 32-bit buffer offsets (the caller groups them)");
  ConvParams p{};
  p.in0 = x_pairs; p.C0 = 384; p.B = 1; p.H = 1; p.W = N * 128; p.ks = 1; p.dil = 1;
  p.wgt = w_planes; p.wgt_tiled = w_tiled; p.bias = bias; p.split = 3; p.out_scale = inv_scale; p.out_planes = 3;
  p.x_tiled = x_tiled; p.out_tiled = out_tiled;
  p.out = out_planes; p.out_ld = 384; p.Cout = 1152; p.M = N * 128; p.act = kActNone;
  p.dbg_flags = g_qkv_attn_dbg; p.dbg = g_qkv_attn_stamps;
  launch_sp<128, 192, 4, 2, 3, 2, 1, true, 3, 1>(p, s);
}

}  // namespace ttr
