// The encoder's qkv projection + self-attention of a (crop, head) as FOUR-wave workgroups, two per CU (gfx950 / MI355X): the form VERDICT r05 asked for
// (timm Attention.forward inside the TorchScript PARSeq run at tuatara.cpp:307).
//
// gemm_sp.hip's EPI = 1 tile runs one eight-wave workgroup per CU: all eight waves go through the tile's K loop (matrix cores) and then, in lock step, through its attention
// (conversions, softmax, LDS round trips: vector work) while the matrix pipe idles - 395 + 288 us per layer at 1280 crops.  Here a tile is owned by FOUR waves (wave tiles of
// 64 rows x 96 channels) and two such workgroups share a CU, a tile apart in time: one's attention runs beside the other's K loop.  What that takes:
//   * LDS: 160 KB / 2 = 80 KB per workgroup = the four 64-deep plane tiles of ONE k0 (x0, x1: 128 rows x 128 B; w0, w1: 192 rows), in gemm_sp.hip's image.  No second copy of
//     anything: a slot is requested again as soon as every wave has its fragments of it, two phases (of three per k0) before it is read.  The attention's images (Q exchange
//     24 KB, K / V 32 KB) lie OVER the tiles: the loader stops for the attention and restarts behind it.
//   * registers: 256 per lane (two waves per SIMD, one from each workgroup): 96 accumulators + at most 104 of fragments (x0 32, w0 / 2^11 48, half of w1 24).
//   * the same bits as gemm_sp.hip's tile: every accumulator of the projection sees that kernel's MFMAs in its order - per k0 (w0, x0) (w1, x0) (w0 / 2^11, x1), each over both
//     32-deep halves - and every score its two d halves in that kernel's order (tests/test_gpu_attn.py asserts np.array_equal).  A first form that walked K in 32-deep steps of
//     [x0 | x1] x [w0 | w1] (two stages of 40 KB, one barrier per step) was 6 - 9 % faster than this one and agreed to fp32 rounding only: on 10 240 crops it moved one
//     crop's worst logit to 1.4e-3 from the oracle (profiles/r06_qkv_attn4.txt), so it was not kept.
// The attention is gemm_sp.hip's epilogue on wave tiles of 64 rows: a wave takes 32 queries (two 16-row blocks) instead of 16.
// Measured (profiles/r06_qkv_attn4.txt): 650 - 668 us per launch at 1280 crops against the eight-wave tile's 677 - 693 (- 3 %), nothing on the page rate, + 0.17 ms on a single
// page's p50 (a lone four-wave workgroup takes longer over its tile): tuning key "qkv_attn4" is OFF by default.
#include <algorithm>
#include <stdexcept>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
constexpr int QXB = 128 * 128, QWB = 192 * 128, QSTAGE = QXB + QWB;   // 16 KB + 24 KB
constexpr int QLDS = 2 * QSTAGE;                                       // 81920
constexpr int QKV = 16384;                                             // one plane of K or V: 128 rows x 128 B
}  // namespace

__global__ __launch_bounds__(256, 2) void qkv_attn4_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int K = 384, NKS = K / 32;

  // persistent, XCD-aware tile schedule (gemm_sp.hip): tile = crop * 6 + head
  const int T = (p.M / 128) * 6;
  const int xcd = blockIdx.x & 7, J = gridDim.x >> 3;
  int xcd_first, xcd_count;
  {
    const int q = T >> 3, r = T & 7;
    xcd_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xcd_count = q + (xcd < r ? 1 : 0);
  }
  // a workgroup's first tile is its own (blockIdx.x >> 3); the rest come from the XCD's counter (ConvParams::tile_ctr): of the two workgroups of a CU the older one wins
  // the issue arbitration and runs ahead - with equal shares one ended at 470 us, the other at 610 (profiles/r06_qkv_attn4.txt)
  int idx = blockIdx.x >> 3;
  auto leave = [&]() {                    // the last workgroup out zeroes the counters for the next launch on the stream
    if (tid == 0 && p.tile_ctr) {
      __threadfence();
      if (atomicAdd(p.tile_ctr + 8, 1u) == gridDim.x - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(p.tile_ctr + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (idx >= xcd_count) { leave(); return; }

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in0), 0, (int)(unsigned)((size_t)p.M * K * 4), 0x00020000);
  const bool wtiled = p.wgt_tiled != nullptr;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wtiled ? p.wgt_tiled : p.wgt), 0, (int)(unsigned)((size_t)1152 * K * 6), 0x00020000);

  // ---- loader (gemm_sp.hip's image): a plane tile is 64 deep - 128 activation rows (16 pieces of 8 rows x 128 B, 4 per wave) or 192 weight rows (24 pieces, 6 per wave).  A lane
  // owns row 8 q + (lane >> 3) of piece q and LDS position (lane & 7), which holds chunk g = position ^ ((row >> 1) & 7) of the row's 64 k.
  unsigned xo[4], wo[6];
  auto tile_offsets = [&](int li) {
    const bool live = li < xcd_count;
    const int tile = xcd_first + li, m0 = (tile / 6) * 128, n0 = (tile % 6) * 192;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (i * 4 + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int m = m0 + row;
      if (p.x_tiled) xo[i] = live ? (unsigned)(m >> 3) * 12u * 1024u + (unsigned)((m & 7) * 128 + g * 16) : OOB;   // [rows / 8][plane][K / 64][8][64]
      else xo[i] = live ? ((unsigned)m * (unsigned)(2 * K) + (unsigned)(g * 8)) * 2u : OOB;                          // rows [x0 | x1]
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int row = (j * 4 + wave) * 8 + (lane >> 3);
      const int g = (lane & 7) ^ ((row >> 1) & 7);
      const int q16 = row & 15;
      const int n = n0 + (row & ~31) + (q16 >> 2) * 8 + ((row >> 4) & 1) * 4 + (q16 & 3);   // channel held by that LDS row (gemm_sp.hip)
      if (wtiled) wo[j] = live ? (unsigned)((n0 + (row & ~7)) >> 3) * 18u * 1024u + (unsigned)((row & 7) * 128 + g * 16) : OOB;   // [Cout / 8][plane][K / 64][8][64]
      else wo[j] = live ? ((unsigned)n * (unsigned)(3 * K) + (unsigned)(g * 8)) * 2u : OOB;                                        // rows [w0 | w0b | w1]
    }
  };
  const unsigned x_pl1 = p.x_tiled ? 6u * 1024u : (unsigned)K * 2u, x_kstep = p.x_tiled ? 1024u : 128u;     // byte offsets of plane x1 and of one k0
  const unsigned w_pl1 = wtiled ? 12u * 1024u : (unsigned)K * 4u, w_kstep = wtiled ? 1024u : 128u;            // ... of plane w1 (the staged w0b between them is not read)
  auto issue_x = [&](int pl, int k0) {               // plane pl of k0 into X slot pl
    const unsigned so = (pl ? x_pl1 : 0u) + (unsigned)k0 * x_kstep;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const unsigned vo = xo[i]; __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(smem + pl * QXB + (i * 4 + wave) * 1024), 16, vo, so, 0, 0); }
  };
  auto issue_w = [&](int pl, int k0) {               // w0 (pl = 0) or w1 (1) of k0 into W slot pl
    const unsigned so = (pl ? w_pl1 : 0u) + (unsigned)k0 * w_kstep;
#pragma unroll
    for (int j = 0; j < 6; ++j) { const unsigned vo = wo[j]; __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(smem + 2 * QXB + pl * QWB + (j * 4 + wave) * 1024), 16, vo, so, 0, 0); }
  };
  auto tile_head = [&]() { issue_x(0, 0); issue_w(0, 0); issue_w(1, 0); };   // what the steady state has in flight when a k0 begins

  // fragment addressing (gemm_sp.hip): row = 16-aligned base + fr; the first 32 k at chunk fg, the second at chunk 4 + fg (byte ^ 64)
  const int frag_lane = fr * 128 + ((fg ^ ((lane >> 1) & 7)) << 4);
  const int xf = wm * 64 * 128 + frag_lane, wf = 2 * QXB + wn * 96 * 128 + frag_lane;
  const f16 dn = (f16)(1.f / 2048.f);
  const f16x8 dnv = {dn, dn, dn, dn, dn, dn, dn, dn};
  const float osc = p.out_scale;

  int stamp_tile = 0;
#define TTR_Q4_STAMP(k) do { if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp_tile < 24) p.dbg[stamp_tile * 16 + (k)] = __builtin_readcyclecounter(); } while (0)

  // (comparison, off: the matrix phase as a per-CU critical section - ConvParams::tile_ctr words 16 ..: one per CU, by XCC_ID and HW_ID's SE / SH / CU fields - so that a
  // workgroup's K loop starts when the other's ends and one's attention always runs beside the other's K loop.  Measured 3 - 5 % SLOWER: a K loop with the matrix pipe to
  // itself still takes 17 us of a 37 us period and the attention beside it 19 - the phases slow each other at the issue port, not by meeting.  The wait is bounded - a
  // workgroup alone on its CU finds the token free, a stuck one goes on without it after 20 us - and the token is given back on every path out.)
  unsigned* cu_token = nullptr;
  bool owns_token = false;
  if ((p.dbg_flags & 64) && p.tile_ctr && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    cu_token = p.tile_ctr + 16 + (((xcc & 7) << 8) | ((hw >> 8) & 0xFF));
  }
  auto take_token = [&]() {
    if (cu_token) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (!(owns_token = atomicCAS(cu_token, 0u, 1u) == 0u)) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000ull) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
  };
  auto give_token = [&]() {
    if (cu_token && owns_token) { __hip_atomic_store(cu_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); owns_token = false; }
  };
  const unsigned long long wg_t0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull, wg_c0 = p.dbg ? __builtin_readcyclecounter() : 0ull;
  tile_offsets(idx);
  tile_head();
  while (true) {
    TTR_Q4_STAMP(0);
    if (p.dbg && blockIdx.x == 0 && tid == 0 && stamp_tile < 24) p.dbg[stamp_tile * 16 + 11] = __builtin_amdgcn_s_memrealtime();   // (100 MHz: the shader clock the kernel ran at)
    const int ctile = xcd_first + idx;
    const int m0c = (ctile / 6) * 128, head = ctile % 6;
    f32x4 acc[6][4];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int next_idx = 0;                                   // the tile after this one: asked for here, back by the second step's wait
    if (p.tile_ctr && tid == 0) next_idx = J + (int)atomicAdd(p.tile_ctr + xcd, 1u);

    // ---- K loop: per 64-deep k0 the three products in gemm_sp.hip's order - (w0, x0) (w1, x0) (w0 / 2^11, x1), each over both 32-deep halves - so that every accumulator sees
    // the same MFMAs in the same order as there: the two kernels agree bit for bit.  The four plane tiles of a k0 are the whole 80 KB; a slot is asked for again as soon as
    // every wave has its fragments of it (the barrier of the next phase), two phases before it is read:
    //   phase A (reads X0, W0):  awaits X0(k) W0(k)   asks X1(k)            phase B (reads W1):  awaits W1(k)   asks X0(k+1) W0(k+1)         phase C (reads X1):  awaits X1(k)   asks W1(k+1)
    if (p.dbg_flags & 16) __builtin_amdgcn_s_setprio(2);   // the matrix phase in front of the other workgroup's vector phase at the issue arbiter (- 2 %)
    take_token();
    const int nk0 = (p.dbg_flags & 8) ? 1 : K / 64;      // (dbg_flags 8, timing experiment: one k0 only - the attention's cost alone)
    for (int k0 = 0; k0 < nk0; ++k0) {
      const bool last = k0 == nk0 - 1;
      if (last) give_token();
      f16x8 x[2][4], w0[2][6];
      // phase A
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __syncthreads();
      issue_x(1, k0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[kk][i] = *reinterpret_cast<const f16x8*>(smem + (xf ^ (kk * 64)) + i * 2048);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 6; ++j) w0[kk][j] = *reinterpret_cast<const f16x8*>(smem + (wf ^ (kk * 64)) + j * 2048);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[kk][j], x[kk][i], acc[j][i], 0, 0, 0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 6; ++j) w0[kk][j] = w0[kk][j] * dnv;
      // phase B (w1 in two halves of three channel blocks: 24 registers less at the peak)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __syncthreads();
      if (!last) { issue_x(0, k0 + 1); issue_w(0, k0 + 1); }
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        f16x8 w1[2][3];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 3; ++j) w1[kk][j] = *reinterpret_cast<const f16x8*>(smem + QWB + (wf ^ (kk * 64)) + (jh * 3 + j) * 2048);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[jh * 3 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[kk][j], x[kk][i], acc[jh * 3 + j][i], 0, 0, 0);
      }
      // phase C
      if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      __syncthreads();
      if (!last) issue_w(1, k0 + 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[kk][i] = *reinterpret_cast<const f16x8*>(smem + QXB + (xf ^ (kk * 64)) + i * 2048);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0[kk][j], x[kk][i], acc[j][i], 0, 0, 0);
    }
    give_token();
    if (p.dbg_flags & 16) __builtin_amdgcn_s_setprio(0);
    if (p.dbg_flags & 32) __builtin_amdgcn_s_setprio(2);   // (comparison: the other way round)
    TTR_Q4_STAMP(1);                                    // K loop done
    __syncthreads();                                    // every wave has read the last step's fragments: the stages become the attention's images
    TTR_Q4_STAMP(2);
    if (p.dbg_flags & 4) {                              // timing experiment (results are wrong): the K loop alone
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
      if (sum == 1.2345e30f) reinterpret_cast<float*>(p.out)[0] = sum;
      idx += J;
      if (idx >= xcd_count) break;
      tile_offsets(idx);
      tile_head();
      continue;
    }
    if (p.tile_ctr) { if (tid == 0) *reinterpret_cast<volatile int*>(smem + QLDS - 4) = next_idx; }   // (a word no attention image reaches; read behind the next barrier)
    else idx += J;

    // ---- attention (gemm_sp.hip's epilogue on wave tiles of 64 rows).  The tile is Q | K | V [128 rows][64] of one (crop, head): tile channel 96 wn + 32 t + dd is Q (t = 0),
    // K (t = 1) or V (t = 2), d = 32 wn + dd.  A lane holds, of row 64 wm + 16 i + fr, the channels 96 wn + 32 t + 8 fg + e (e < 4 in acc[2t][i], e >= 4 in acc[2t+1][i]),
    // i.e. d = 32 wn + 8 fg + e: an MFMA operand fragment of that row for the d half `wn`.  Wave (wm, wn) takes the 32 queries 64 wm + 32 wn + 16 qb + q (its row blocks
    // i = 2 wn + qb): its own d half of their Q (exact triple) stays in registers, the other half comes from the partner wave (wm, 1 - wn) through LDS, and the partner's
    // rows of this wave's half (blocks i = 2 (1 - wn) + qb) go the other way.
    RangeWatch rw;
    unsigned char* const sQ = smem;                     // [wm 2][reader wn 2][qb 2][3 planes][64 lanes][16 B] = 24 KB
    unsigned char* const sK = smem + 24576;             // [2 planes][128 rows][128 B]; V afterwards
    const float* const bp = p.bias + head * 192 + wn * 96 + fg * 8;
    auto tile_values = [&](int t, int i, float (&v)[8]) {       // bias + scale of the lane's 8 channels of block t, row block i
      const float4 b0 = *reinterpret_cast<const float4*>(bp + t * 32), b1 = *reinterpret_cast<const float4*>(bp + t * 32 + 4);
      const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = fmaf(acc[2 * t][i][e], osc, bv[e]); v[4 + e] = fmaf(acc[2 * t + 1][i][e], osc, bv[4 + e]); }
    };
    f16x8 fq[2][2][3];                                  // [query block][own d half, other d half][plane]
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // Q: blocks 2 wn, 2 wn + 1 stay; the other two go to the partner
      float v[8];
      tile_values(0, i, v);
      f16x8 a, b, c;
      split3_x8(v, a, b, c, rw);
      const bool mine = (i >> 1) == wn;
      const int qb = i & 1;
      if (mine) { fq[qb][0][0] = a; fq[qb][0][1] = b; fq[qb][0][2] = c; }
      else {
        unsigned char* d = sQ + ((wm * 2 + (1 - wn)) * 2 + qb) * 3072 + lane * 16;
        *reinterpret_cast<f16x8*>(d) = a; *reinterpret_cast<f16x8*>(d + 1024) = b; *reinterpret_cast<f16x8*>(d + 2048) = c;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                       // K (pair): key 64 wm + 16 i + fr -> LDS row 32 (2 wm + (i >> 1)) + perm(fr) + 8 (i & 1) (attn_split.hip's image)
      float v[8];
      tile_values(1, i, v);
      f16x8 a, b;
      split2_x8(v, a, b, rw);
      const int row = (2 * wm + (i >> 1)) * 32 + (((fr >> 2) & 1) << 4) + ((fr >> 3) << 2) + (fr & 3) + 8 * (i & 1);
      unsigned char* d = sK + row * 128 + (((wn * 4 + fg) ^ ((row >> 1) & 7)) << 4);
      *reinterpret_cast<f16x8*>(d) = a; *reinterpret_cast<f16x8*>(d + QKV) = b * dnv;
    }
    f16x8 vp[2][4];                                     // V pairs [plane][i]: kept until every wave is through with K
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[8];
      tile_values(2, i, v);
      split2_x8(v, vp[0][i], vp[1][i], rw);
      vp[1][i] = vp[1][i] * dnv;
    }
    rw.flush(p.range_flag, p.range_tag);
    TTR_Q4_STAMP(3);                                    // Q / K / V converted, Q and K written
    __syncthreads();
    TTR_Q4_STAMP(4);
    if (p.tile_ctr) idx = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(smem + QLDS - 4));
    const bool has_next = idx < xcd_count;
    if (has_next) tile_offsets(idx);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fq[qb][1][pl] = *reinterpret_cast<const f16x8*>(sQ + ((wm * 2 + wn) * 2 + qb) * 3072 + pl * 1024 + lane * 16);
    // S^T = K Q^T per query block: sacc[qb][kt], lane = query fr, LDS key rows 16 kt + 4 fg + r.  The two d halves of a score are summed in gemm_sp.hip's order - there the
    // wave that owns a query row has d half (row >> 4) & 1 in registers and takes it first - which here is the block's qb: half qb first, whichever wave it came from.
    f32x4 sacc[2][8];
    {
      const int krd = fr * 128 + (((wn * 4 + fg) ^ ((fr >> 1) & 7)) << 4);          // K fragment, kt = 0, this wave's d half (the other: ^ 64)
      const unsigned char* const kb[2] = {sK + krd, sK + (krd ^ 64)};
      const unsigned ka0 = (unsigned)(size_t)(lds_ptr)sK + (unsigned)krd, ka1 = (unsigned)(size_t)(lds_ptr)sK + (unsigned)(krd ^ 64);
      // a key tile's four fragments ([own, other d half] x [k0, k1]) are read while the tile before it runs its sixteen MFMAs (left alone the compiler reads a tile where it
      // is used: eight exposed LDS round trips).  READ names a register the MFMAs in front of which it belongs consume, WAIT the results of the MFMAs it belongs behind.
#define TTR_Q4_KREAD(kt, K, pin)                                                                                                                        \
      asm volatile("ds_read_b128 %0, %5 offset:%7\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%7\n\tds_read_b128 %3, %6 offset:%8"      \
                   : "=&v"(K[0][0]), "=&v"(K[0][1]), "=&v"(K[1][0]), "=&v"(K[1][1]), "+v"(pin) : "v"(ka0), "v"(ka1), "n"((kt) * 2048), "n"(QKV + (kt) * 2048));
#define TTR_Q4_KWAIT(K, r0, r1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(K[0][0]), "+v"(K[0][1]), "+v"(K[1][0]), "+v"(K[1][1]), "+v"(r0), "+v"(r1));
      auto s_tile = [&](auto first_c, const f16x8 (&K)[2][2], f32x4& a0, f32x4& a1) {   // K[hf][k0 / k1]; F: which of [own, other] is block 0's first half (block 1 takes the other first)
        constexpr int F = decltype(first_c)::value;
        f16x8 k0b[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) k0b[hf] = K[hf][0] * dnv;
        a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int h0 = o ? 1 - F : F, h1 = 1 - h0;
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(K[h0][0], fq[0][h0][0], a0, 0, 0, 0);   // (the two blocks' chains alternate: a dependent MFMA is never the next one)
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(K[h1][0], fq[1][h1][0], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b[h0], fq[0][h0][1], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b[h1], fq[1][h1][1], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b[h0], fq[0][h0][2], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0b[h1], fq[1][h1][2], a1, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(K[h0][1], fq[0][h0][0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(K[h1][1], fq[1][h1][0], a1, 0, 0, 0);
        }
      };
      auto s_loop = [&](auto first_c) {
        f16x8 KA[2][2], KB[2][2];
        TTR_Q4_KREAD(0, KA, fq[0][0][0])
        TTR_Q4_KWAIT(KA, fq[0][0][1], fq[0][0][2])
        TTR_Q4_KREAD(1, KB, KA[0][0])  s_tile(first_c, KA, sacc[0][0], sacc[1][0]);  TTR_Q4_KWAIT(KB, sacc[0][0], sacc[1][0])
        TTR_Q4_KREAD(2, KA, KB[0][0])  s_tile(first_c, KB, sacc[0][1], sacc[1][1]);  TTR_Q4_KWAIT(KA, sacc[0][1], sacc[1][1])
        TTR_Q4_KREAD(3, KB, KA[0][0])  s_tile(first_c, KA, sacc[0][2], sacc[1][2]);  TTR_Q4_KWAIT(KB, sacc[0][2], sacc[1][2])
        TTR_Q4_KREAD(4, KA, KB[0][0])  s_tile(first_c, KB, sacc[0][3], sacc[1][3]);  TTR_Q4_KWAIT(KA, sacc[0][3], sacc[1][3])
        TTR_Q4_KREAD(5, KB, KA[0][0])  s_tile(first_c, KA, sacc[0][4], sacc[1][4]);  TTR_Q4_KWAIT(KB, sacc[0][4], sacc[1][4])
        TTR_Q4_KREAD(6, KA, KB[0][0])  s_tile(first_c, KB, sacc[0][5], sacc[1][5]);  TTR_Q4_KWAIT(KA, sacc[0][5], sacc[1][5])
        TTR_Q4_KREAD(7, KB, KA[0][0])  s_tile(first_c, KA, sacc[0][6], sacc[1][6]);  TTR_Q4_KWAIT(KB, sacc[0][6], sacc[1][6])
        s_tile(first_c, KB, sacc[0][7], sacc[1][7]);
      };
      if (wn == 0) s_loop(std::integral_constant<int, 0>{});   // block 0's half 0 is this wave's own
      else s_loop(std::integral_constant<int, 1>{});
#undef TTR_Q4_KWAIT
#undef TTR_Q4_KREAD
    }
    TTR_Q4_STAMP(5);                                    // S done
    // softmax over the 128 keys of a query: 32 values here, the rest in lanes fr + 16 g'; probabilities as pairs (gemm_sp.hip)
    f16x8 fp[2][2][4];                                  // [query block][plane][32-key step]
    float rinv[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qb][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
      RangeWatch rp;                                    // (dead: the probabilities are <= 1)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        float ev[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ev[e] = __expf((sacc[qb][2 * s4 + (e >> 2)][e & 3] - mx) * 0.125f);
          sum += ev[e];
        }
        split2_x8(ev, fp[qb][0][s4], fp[qb][1][s4], rp);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      rinv[qb] = 1.0f / sum;
    }
    TTR_Q4_STAMP(6);                                    // softmax + probability planes done
    __syncthreads();                                    // every wave has read K
    // V: row = key 64 wm + 16 i + fr; the 8 values d = 32 wn + 8 fg + e go to column positions 32 wn + 4 fg + e (e < 4) and 32 wn + 16 + 4 fg + e - 4
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f16x8 v = vp[pl][i];
        unsigned char* d = sK + (wm * 64 + i * 16 + fr) * 128 + wn * 64 + fg * 8 + pl * QKV;
        *reinterpret_cast<f16x4*>(d) = f16x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f16x4*>(d + 32) = f16x4{v[4], v[5], v[6], v[7]};
      }
    __syncthreads();
    TTR_Q4_STAMP(7);                                    // V written
    // O^T = V^T P^T: A = V^T fragment (16 column positions x 32 keys) by transposed reads of the row-major planes (attn_split.hip)
    f32x4 oacc[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const unsigned vbase = (unsigned)(size_t)(lds_ptr)sK + (unsigned)((8 * fg + (fr >> 2)) * 128 + (fr & 3) * 8);
      // the reads of step s + 1 go out in front of step s's MFMAs (two register sets): READ's first instruction also names a register the MFMAs of the step before it
      // consume, WAIT names those MFMAs' results - the compiler keeps [READ(s + 1)] [MFMAs(s)] [WAIT(s + 1)] in that order
#define TTR_Q4_READ(s, lo, hi, pin)                                                                            \
      asm volatile("ds_read_b64_tr_b16 %0, %17 offset:%18\n\tds_read_b64_tr_b16 %1, %17 offset:%18+512\n\t"                                             \
                   "ds_read_b64_tr_b16 %2, %17 offset:%18+32\n\tds_read_b64_tr_b16 %3, %17 offset:%18+544\n\t"                                          \
                   "ds_read_b64_tr_b16 %4, %17 offset:%18+64\n\tds_read_b64_tr_b16 %5, %17 offset:%18+576\n\t"                                          \
                   "ds_read_b64_tr_b16 %6, %17 offset:%18+96\n\tds_read_b64_tr_b16 %7, %17 offset:%18+608\n\t"                                          \
                   "ds_read_b64_tr_b16 %8, %17 offset:%19\n\tds_read_b64_tr_b16 %9, %17 offset:%19+512\n\t"                                            \
                   "ds_read_b64_tr_b16 %10, %17 offset:%19+32\n\tds_read_b64_tr_b16 %11, %17 offset:%19+544\n\t"                                       \
                   "ds_read_b64_tr_b16 %12, %17 offset:%19+64\n\tds_read_b64_tr_b16 %13, %17 offset:%19+576\n\t"                                       \
                   "ds_read_b64_tr_b16 %14, %17 offset:%19+96\n\tds_read_b64_tr_b16 %15, %17 offset:%19+608"                                           \
                   : "=&v"(lo[0][0]), "=&v"(hi[0][0]), "=&v"(lo[0][1]), "=&v"(hi[0][1]), "=&v"(lo[0][2]), "=&v"(hi[0][2]), "=&v"(lo[0][3]), "=&v"(hi[0][3]),   \
                     "=&v"(lo[1][0]), "=&v"(hi[1][0]), "=&v"(lo[1][1]), "=&v"(hi[1][1]), "=&v"(lo[1][2]), "=&v"(hi[1][2]), "=&v"(lo[1][3]), "=&v"(hi[1][3]),   \
                     "+v"(pin)                                                                                                                        \
                   : "v"(vbase), "n"((s) * 4096), "n"(QKV + (s) * 4096));
#define TTR_Q4_WAIT(lo, hi)                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[0][2]), "+v"(lo[0][3]), "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[0][2]), "+v"(hi[0][3]), \
                   "+v"(lo[1][0]), "+v"(lo[1][1]), "+v"(lo[1][2]), "+v"(lo[1][3]), "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(hi[1][2]), "+v"(hi[1][3]), \
                   "+v"(oacc[0][0]), "+v"(oacc[0][1]), "+v"(oacc[0][2]), "+v"(oacc[0][3]), "+v"(oacc[1][0]), "+v"(oacc[1][1]), "+v"(oacc[1][2]), "+v"(oacc[1][3]));
#define TTR_Q4_MFMAS(s, lo, hi)                                                                                \
      _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                       \
        const f16x8 v0 = __builtin_shufflevector(lo[0][dt], hi[0][dt], 0, 1, 2, 3, 4, 5, 6, 7);                \
        const f16x8 v1 = __builtin_shufflevector(lo[1][dt], hi[1][dt], 0, 1, 2, 3, 4, 5, 6, 7);                \
        const f16x8 v0b = v0 * dnv;                                                                            \
        _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                                     \
          f32x4 a = oacc[qb][dt];                                                                              \
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, fp[qb][0][s], a, 0, 0, 0);                            \
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0b, fp[qb][1][s], a, 0, 0, 0);                           \
          oacc[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, fp[qb][0][s], a, 0, 0, 0);                 \
        }                                                                                                      \
      }
      f16x4 loA[2][4], hiA[2][4], loB[2][4], hiB[2][4];
      TTR_Q4_READ(0, loA, hiA, fp[0][0][0])
      TTR_Q4_WAIT(loA, hiA)
      TTR_Q4_READ(1, loB, hiB, loA[0][0])
      TTR_Q4_MFMAS(0, loA, hiA)
      TTR_Q4_WAIT(loB, hiB)
      TTR_Q4_READ(2, loA, hiA, loB[0][0])
      TTR_Q4_MFMAS(1, loB, hiB)
      TTR_Q4_WAIT(loA, hiA)
      TTR_Q4_READ(3, loB, hiB, loA[0][0])
      TTR_Q4_MFMAS(2, loA, hiA)
      TTR_Q4_WAIT(loB, hiB)
      TTR_Q4_MFMAS(3, loB, hiB)
#undef TTR_Q4_MFMAS
#undef TTR_Q4_WAIT
#undef TTR_Q4_READ
    }
    TTR_Q4_STAMP(8);                                    // P V done
    __syncthreads();                                    // every wave has read V: the stages are free again
    if (has_next) tile_head();                          // the next tile's first requests, in front of this tile's output stores
    asm volatile("" ::: "memory");
    TTR_Q4_STAMP(9);
    // out planes (exact triples): oacc[qb][2u], oacc[qb][2u+1] hold d = 32 u + 8 fg + 0..7 of query 64 wm + 32 wn + 16 qb + fr
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int orow = m0c + wm * 64 + wn * 32 + qb * 16 + fr;
      f16* const op = p.out_tiled ? reinterpret_cast<f16*>(p.out) + ((int64_t)(orow >> 3) * 18 + head) * 512 + (orow & 7) * 64 + fg * 8
                                  : reinterpret_cast<f16*>(p.out) + (int64_t)orow * (3 * 384) + head * 64 + fg * 8;
      const int opl = p.out_tiled ? 6 * 512 : 384;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = oacc[qb][2 * u][e] * rinv[qb]; v[4 + e] = oacc[qb][2 * u + 1][e] * rinv[qb]; }
        f16x8 a, b, c;
        RangeWatch ro;                                  // (dead: convex combinations of the V rows watched above)
        split3_x8(v, a, b, c, ro);
        *reinterpret_cast<f16x8*>(op + u * 32) = a; *reinterpret_cast<f16x8*>(op + opl + u * 32) = b; *reinterpret_cast<f16x8*>(op + 2 * opl + u * 32) = c;
      }
    }
    if (p.dbg_flags & 32) __builtin_amdgcn_s_setprio(0);
    TTR_Q4_STAMP(10);
    ++stamp_tile;
    if (!has_next) break;
  }
  leave();
  if (p.dbg && tid == 0 && blockIdx.x < 512) {   // (diagnostics: where and when each workgroup ran - 100 MHz start / end, HW_ID | XCC_ID << 32, shader clocks)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* d = p.dbg + 512 + blockIdx.x * 4;
    d[0] = wg_t0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = hw | ((unsigned long long)xcc << 32); d[3] = __builtin_readcyclecounter() - wg_c0;
  }
#undef TTR_Q4_STAMP
}

static int g_qkv_attn4 = 0;   // 1: launch_qkv_attn_split (gemm_sp.hip) hands its launches to this kernel
void set_qkv_attn4(int v) { g_qkv_attn4 = v; }
int qkv_attn4_enabled() { return g_qkv_attn4; }
static unsigned long long* g_q4_stamps = nullptr;
void set_qkv_attn4_stamps(unsigned long long* d) { g_q4_stamps = d; }

// same contract as launch_qkv_attn_split (gemm_sp.hip)
void launch_qkv_attn4(const void* x_pairs, const void* w_planes, const float* bias, float inv_scale, void* out_planes, int N, hipStream_t s, const void* w_tiled, int x_tiled,
                      int out_tiled) {
  if (N <= 0) return;
  if (((uintptr_t)x_pairs | (uintptr_t)w_planes | (uintptr_t)bias | (uintptr_t)out_planes) & 15) throw std::runtime_error("qkv_attn4: operands must be 16-byte aligned");
  if ((size_t)N * 128 * 384 * 6 >= ((size_t)1 << 31)) throw std::runtime_error("qkv_attn4: too many crops for 32-bit buffer offsets (the caller groups them)");
  ConvParams p{};
  p.in0 = x_pairs; p.C0 = 384; p.wgt = w_planes; p.wgt_tiled = w_tiled; p.bias = bias; p.out_scale = inv_scale;
  p.x_tiled = x_tiled; p.out_tiled = out_tiled; p.out = out_planes; p.M = N * 128; p.Cout = 1152;
  p = with_range_ctx(p);
  p.dbg = g_q4_stamps;
  // the key's bits beside 1 (comparisons kept for profiles/r06_qkv_attn4.txt): 2 = equal shares by stride instead of the counter; 4 = no issue priority for the K loop,
  // 8 = priority for the attention instead; 16 = the K loop alone, 32 = the attention alone (timing experiments, wrong results); 64 = the per-CU matrix-phase token
  const int v = g_qkv_attn4;
  p.tile_ctr = (v & 2) || (v & 16) ? nullptr : tile_counters(s);
  p.dbg_flags = ((v & 16) ? 4 : 0) | ((v & 32) ? 8 : 0) | ((v & 12) == 0 ? 16 : 0) | ((v & 8) ? 32 : 0) | ((v & 64) ? 64 : 0);
  static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)qkv_attn4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QLDS)); });
  const int T = N * 6;
  const int grid = std::min((T + 7) / 8 * 8, std::max(device_cu_count(256) * 2 / 8 * 8, 8));
  hipLaunchKernelGGL(qkv_attn4_kernel, dim3(grid), dim3(256), QLDS, s, p);
}

}  // namespace ttr
