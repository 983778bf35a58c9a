// Split-operand mode (split.h): elementwise helpers around the f16x4 GEMMs.
//   split_planes_kernel   fp32 [M][C] (row stride ld) -> f16 planes [M][3 C]  (x0 | x1 | x2), optional ReLU first
// They stand where a producer cannot write planes itself (LayerNorm / attention / pooling outputs in the first build of the mode);
// each is one streaming pass: 4 B read + 6 B written per element.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

template <int NPL>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, int ld, f16* __restrict__ out, int64_t M, int C, int relu, const int* skip, int skip_n,
                                                          unsigned* range_flag, unsigned range_tag) {
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  RangeWatch rw;
  const int cv = C >> 3;                                   // 8-channel groups per row
  const int64_t total = M * cv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cv;
    const int c = (int)(i - m * cv) << 3;
    const float* src = in + m * ld + c;
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st_split_n(out, m, C, c, v, NPL, rw);
  }
  rw.flush(range_flag, range_tag);
}

void launch_split_planes(const float* in, int ld, void* out, int64_t M, int C, int relu, hipStream_t s, int planes, const int* skip, int skip_n) {
  if (M <= 0) return;
  if (C % 8 || ld % 4 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) throw std::runtime_error("split_planes: C must be a multiple of 8 and the tensors 16-byte aligned");
  const int64_t total = M * (C >> 3);
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
  if (planes == 2) hipLaunchKernelGGL(split_planes_kernel<2>, dim3(grid), dim3(256), 0, s, in, ld, (f16*)out, M, C, relu, skip, skip_n, range_ctx().flag, range_ctx().tag);
  else hipLaunchKernelGGL(split_planes_kernel<3>, dim3(grid), dim3(256), 0, s, in, ld, (f16*)out, M, C, relu, skip, skip_n, range_ctx().flag, range_ctx().tag);
}

}  // namespace ttr

// ------------------------------------------------------------------------------------------------------------------
// Planes in, planes out: the elementwise layers between CRAFT's convolutions.  A value is joined from its three planes
// (exact), the fp32 kernels' arithmetic is applied unchanged (craft_ops.hip: max is order-free, the bilinear weights use
// the same spelled-out expression), and the result is split again (exact): bit-identical to the fp32 engine's tensors.
namespace ttr {
namespace {

struct V8 { float v[8]; };
template <int NPL> __device__ __forceinline__ V8 ld_planes(const f16* p, int C) {   // p -> plane 0 of 8 channels of a pixel
  const f16x8 a = *reinterpret_cast<const f16x8*>(p), b = *reinterpret_cast<const f16x8*>(p + C);
  V8 r;
  if constexpr (NPL == 3) {
    const f16x8 c = *reinterpret_cast<const f16x8*>(p + 2 * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = join3(a[e], b[e], c[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = join2(a[e], b[e]);
  }
  return r;
}
template <int NPL> __device__ __forceinline__ void st_planes(f16* p, int C, const V8& r, RangeWatch& rw) {
  if constexpr (NPL == 3) {
    f16x8 a, b, c;
    split3_x8(r.v, a, b, c, rw);
    *reinterpret_cast<f16x8*>(p) = a; *reinterpret_cast<f16x8*>(p + C) = b; *reinterpret_cast<f16x8*>(p + 2 * C) = c;
  } else {
    f16x8 a, b;
    split2_x8(r.v, a, b, rw);
    *reinterpret_cast<f16x8*>(p) = a; *reinterpret_cast<f16x8*>(p + C) = b;
  }
}

template <int NPL>
__global__ __launch_bounds__(256) void maxpool3x3s1_planes_kernel(const f16* __restrict__ in, f16* __restrict__ out, int B, int H, int W, int C) {
  const int Cc = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)B * H * W * Cc;
  if (idx >= total) return;
  const int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  const int x = (int)(t % W); t /= W;
  const int y = (int)(t % H); const int b = (int)(t / H);
  V8 m;
#pragma unroll
  for (int e = 0; e < 8; ++e) m.v[e] = -INFINITY;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const V8 v = ld_planes<NPL>(in + (((int64_t)b * H + yy) * W + xx) * (NPL * (int64_t)C) + cc * 8, C);
#pragma unroll
      for (int e = 0; e < 8; ++e) m.v[e] = fmaxf(m.v[e], v.v[e]);
    }
  RangeWatch rw;   // (not flushed: a maximum of values their producer has watched)
  st_planes<NPL>(out + (((int64_t)b * H + y) * W + x) * (NPL * (int64_t)C) + cc * 8, C, m, rw);
}

// F.interpolate(mode='bilinear', align_corners=False), exact x2: the expression of craft_ops.hip's bilerp()
__device__ __forceinline__ float bilerp_s(float v00, float v01, float v10, float v11, float lx0, float lx1, float ly0, float ly1) {
  const float top = fmaf(lx1, v01, lx0 * v00), bot = fmaf(lx1, v11, lx0 * v10);
  return fmaf(ly1, bot, ly0 * top);
}
template <int NPL>
__global__ __launch_bounds__(256) void upsample2x_planes_kernel(const f16* __restrict__ in, f16* __restrict__ out, int B, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, Cc = C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)B * Ho * Wo * Cc;
  if (idx >= total) return;
  const int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  const int xo = (int)(t % Wo); t /= Wo;
  const int yo = (int)(t % Ho); const int b = (int)(t / Ho);
  const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
  const f16* base = in + (int64_t)b * H * W * (NPL * (int64_t)C) + cc * 8;
  const int64_t ps = NPL * (int64_t)C;
  const V8 v00 = ld_planes<NPL>(base + ((int64_t)y0 * W + x0) * ps, C), v01 = ld_planes<NPL>(base + ((int64_t)y0 * W + x1) * ps, C);
  const V8 v10 = ld_planes<NPL>(base + ((int64_t)y1 * W + x0) * ps, C), v11 = ld_planes<NPL>(base + ((int64_t)y1 * W + x1) * ps, C);
  V8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o.v[e] = bilerp_s(v00.v[e], v01.v[e], v10.v[e], v11.v[e], lx0, lx1, ly0, ly1);
  RangeWatch rw;   // (not flushed: a convex combination of values their producer has watched)
  st_planes<NPL>(out + (((int64_t)b * Ho + yo) * Wo + xo) * ps + cc * 8, C, o, rw);
}

// CRAFT's conv1_1 (3 -> 64, 3x3, ReLU) from the u8 canvas straight into planes: conv1_direct_kernel's structure (craft_ops.hip)
// with split operands.  A pixel's 27 inputs are u8 / 255 in fp32 (the reference's division, tuatara.cpp:367-370): their three planes
// come from three 256-entry tables built once per workgroup; the weights are the layer's three planes [64][3][32].
// FULL: the canvas width is a multiple of 64 - every task stores all four 16-pixel blocks (the counted wait below relies on the number of store instructions)
template <int NPL, bool FULL>
__global__ __launch_bounds__(256, NPL == 2 ? 4 : 3) void conv1_split_kernel(const uint8_t* __restrict__ canvas, const f16* __restrict__ wgt /*[64][3][32]*/, const float* __restrict__ bias,
                                                         float out_scale, f16* __restrict__ out /*[M][3*64]*/, int B, int H, int W, unsigned* range_flag, unsigned range_tag) {
  // A wave takes 64 consecutive pixels of one row (W % 32 == 0: the last segment of a row may hold 32).  The three canvas rows around them come
  // in as 51 aligned dwords each (row bytes 3 x0 - 4 .. 3 x0 + 199: one pixel of halo either side; 3 W and 3 x0 - 4 are multiples of 4) into the
  // wave's own LDS strip, zero outside the image, and the 27 taps of a pixel are byte reads from there: byte 1 + 3 px + (k % 9) of row k / 9.
  // (The earlier form fetched every tap with its own global byte load: 32 per lane and 64 pixels instead of 3 dwords.)
  __shared__ f16 lut[3][256];
  __shared__ __attribute__((aligned(16))) uint8_t strip[4][3][208];
  RangeWatch rw;
  {
    f16x2 a, b, c = f16x2{(f16)0.f, (f16)0.f};
    if constexpr (NPL == 3) split3_pair((float)threadIdx.x / 255.0f, 0.f, a, b, c, rw);
    else split2_pair((float)threadIdx.x / 255.0f, 0.f, a, b, rw);
    lut[0][threadIdx.x] = a[0]; lut[1][threadIdx.x] = b[0]; lut[2][threadIdx.x] = c[0];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  f16x8 fw[3][4];   // [weight plane w0 | w0b | w1][jj]
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) fw[pl][jj] = *reinterpret_cast<const f16x8*>(wgt + n * 96 + pl * 32 + fg * 8);
  }
  float bv[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[t][e] = bias[32 * t + fg * 8 + e];
  int srow[8], soff[8];                                  // this lane's 8 inputs k = 8 fg + e: strip row k / 9, byte 1 + k % 9 (+ 3 px); k >= 27: none
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fg * 8 + e;
    srow[e] = k < 27 ? k / 9 : -1; soff[e] = 1 + (k % 9);
  }
  const int segs = (W + 63) >> 6;
  const int ntasks = B * H * segs;                       // (launcher: < 2^31)
  const int nwaves = (int)gridDim.x * 4, wave0 = (int)blockIdx.x * 4 + wave;
  uint8_t (*st)[208] = strip[wave];
  // The canvas rows of a task are REQUESTED a task ahead, in front of the previous task's stores, and awaited with a counted s_waitcnt behind them: vmcnt counts loads and
  // stores together in issue order, so a load behind 16 - 24 stores that is awaited with vmcnt(0) waits for every one of those stores to be acknowledged - once per
  // task, 100 of this kernel's 430 us per 8-page group (measured with the arithmetic switched off: the stores alone 283 us, the stores behind these loads 381;
  // profiles/r06_conv1_split.txt).  The loads are inline asm (buffer loads: lanes
  // outside the row or the image get an out-of-range offset and read zero) so that the compiler does not put its own vmcnt(0) in front of their use.
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const size_t canvas_bytes = (size_t)B * H * W * 3;
  const i32x4 rs = {(int)(unsigned)(uintptr_t)canvas, (int)(unsigned)((uintptr_t)canvas >> 32), (int)(unsigned)std::min<size_t>(canvas_bytes, 0xFFFFFFFFu), 0x00020000};
  constexpr unsigned OOB = 0x80000000u;
  struct Pos { int seg, by, y; };
  auto request = [&](const Pos& q, unsigned (&v)[3], bool live) {   // (not live - no next task -: three out-of-range loads, so that every task runs the same code path)
    const int x0 = q.seg << 6;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int yy = q.y - 1 + r;
      const bool ok = live && lane < 51 && yy >= 0 && yy < H && !(x0 == 0 && lane == 0) && 4 * lane < 3 * (W - x0) + 4;   // (inside the row: the right halo pixel too, where there is one)
      const unsigned off = ok ? (unsigned)(((q.by - q.y + yy) * W + x0) * 3 - 4 + 4 * lane) : OOB;                 // (launcher: the canvas is below 2 GB)
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v[r]) : "v"(off), "s"(rs) : "memory");
    }
  };
  Pos pos{wave0 % segs, wave0 / segs, 0};
  pos.y = pos.by % H;
  const int dseg = nwaves % segs, dby = nwaves / segs, dy = dby % H;
  unsigned cur[3] = {0u, 0u, 0u};
  if (wave0 < ntasks) {
    request(pos, cur, true);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]) : : "memory");
  }
  constexpr int kStores = 4 * 2 * NPL;                   // store instructions of a task with all four 16-pixel blocks inside the row
  for (int task = wave0; task < ntasks; task += nwaves) {
    const int seg = pos.seg, by = pos.by, y = pos.y;
    const int x0 = seg << 6, npx = min(64, W - x0);
    (void)y;
    if (lane < 51) {
#pragma unroll
      for (int r = 0; r < 3; ++r) *reinterpret_cast<unsigned*>(&st[r][4 * lane]) = cur[r];
    }
    // the next task's position and the request for its rows (uniform over the wave)
    Pos nx{seg + dseg, by + dby, y + dy};
    if (nx.seg >= segs) { nx.seg -= segs; ++nx.by; ++nx.y; }
    if (nx.y >= H) nx.y -= H;
    const bool has_next = task + nwaves < ntasks;
    unsigned nxt[3];
    request(nx, nxt, has_next);
    // (per block of 16 pixels: its 12 - 16 MFMAs, then its epilogue and stores - the accumulators of one block are live at a time, so that four workgroups share a
    // CU: with the four blocks' MFMAs first and their epilogues behind, 64 accumulator registers put the kernel at two waves per SIMD and 3.9 TB/s of its 6.3)
    const int64_t m0 = (int64_t)by * W + x0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int px = i * 16 + fr;
      f16x8 fx[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int byte = srow[e] >= 0 ? st[srow[e]][soff[e] + 3 * px] : 0;   // table entry 0 = (0, 0, 0)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fx[pl][e] = lut[pl][byte];
      }
      f32x4 acc[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[0][jj], fx[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[1][jj], fx[1], a, 0, 0, 0);
        if constexpr (NPL == 3) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[1][jj], fx[2], a, 0, 0, 0);
        acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[2][jj], fx[0], a, 0, 0, 0);
      }
      if (px < npx) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          V8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o.v[e] = fmaxf(fmaf(acc[2 * t][e], out_scale, bv[t][e]), 0.f);
            o.v[4 + e] = fmaxf(fmaf(acc[2 * t + 1][e], out_scale, bv[t][4 + e]), 0.f);
          }
          st_planes<NPL>(out + (m0 + px) * (NPL * 64) + 32 * t + fg * 8, 64, o, rw);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the requested rows have landed once all but this task's stores (issued behind them) are done.  ONE wait on ONE code path - with a branch around the wait the
    // compiler may copy the rows' registers on the way into the branch, before the data is there (conv1u.hip's last tile came out wrong that way) - so the count is the
    // smaller of the two a task can have where the width is not a multiple of 64 (a last segment of 32 pixels issues half the stores: W % 32 == 0; a count below the
    // real one only waits for more), and the full count in the kernel compiled for widths that are (FULL)
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]) : "n"(FULL ? kStores : kStores / 2) : "memory");
    cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
    pos = nx;
  }
  rw.flush(range_flag, range_tag);
}

}  // namespace

void launch_maxpool3x3s1_planes(const void* in, void* out, int B, int H, int W, int C, hipStream_t s, int planes) {
  if (C % 8) throw std::runtime_error("maxpool3x3 (planes): C % 8");
  const int64_t total = (int64_t)B * H * W * (C >> 3);
  const dim3 g((unsigned)((total + 255) / 256));
  if (planes == 2) hipLaunchKernelGGL(maxpool3x3s1_planes_kernel<2>, g, dim3(256), 0, s, (const f16*)in, (f16*)out, B, H, W, C);
  else hipLaunchKernelGGL(maxpool3x3s1_planes_kernel<3>, g, dim3(256), 0, s, (const f16*)in, (f16*)out, B, H, W, C);
}
void launch_upsample2x_planes(const void* in, void* out, int B, int H, int W, int C, hipStream_t s, int planes) {
  if (C % 8) throw std::runtime_error("upsample2x (planes): C % 8");
  const int64_t total = (int64_t)B * 4 * H * W * (C >> 3);
  const dim3 g((unsigned)((total + 255) / 256));
  if (planes == 2) hipLaunchKernelGGL(upsample2x_planes_kernel<2>, g, dim3(256), 0, s, (const f16*)in, (f16*)out, B, H, W, C);
  else hipLaunchKernelGGL(upsample2x_planes_kernel<3>, g, dim3(256), 0, s, (const f16*)in, (f16*)out, B, H, W, C);
}
void launch_conv1_split(const uint8_t* canvas, const void* wgt_planes, const float* bias, float out_scale, void* out, int B, int H, int W, hipStream_t s, int planes) {
  if (W % 32) throw std::runtime_error("conv1_split: the canvas width must be a multiple of 32");
  const int64_t tasks = (int64_t)B * H * ((W + 63) / 64);
  if (tasks >= ((int64_t)1 << 31) || (int64_t)B * H * W * 3 >= ((int64_t)1 << 31)) throw std::runtime_error("conv1_split: the canvas batch exceeds 2 GB (32-bit offsets); the caller groups the pages");
  const int grid = (int)std::min<int64_t>((tasks + 3) / 4, 256 * 16);
  const bool full = W % 64 == 0;
#define TTR_C1_LAUNCH(NPLV, FULLV) hipLaunchKernelGGL((conv1_split_kernel<NPLV, FULLV>), dim3(grid), dim3(256), 0, s, canvas, (const f16*)wgt_planes, bias, out_scale, (f16*)out, B, H, W, range_ctx().flag, range_ctx().tag)
  if (planes == 2) { if (full) TTR_C1_LAUNCH(2, true); else TTR_C1_LAUNCH(2, false); }
  else { if (full) TTR_C1_LAUNCH(3, true); else TTR_C1_LAUNCH(3, false); }
#undef TTR_C1_LAUNCH
}

}  // namespace ttr

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm (D = 384) of fp32 rows into planes: one wave per row, 48 lanes x 8 consecutive columns (two 16-byte loads, three
// 16-byte plane stores per lane); statistics as layernorm_kernel (parseq_ops.hip): mean, then the centred sum of squares.
namespace ttr {
namespace {
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_planes_kernel(const float* __restrict__ in, int in_ld, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, f16* __restrict__ out, int M, const int* skip, int skip_n, int tiled,
                                                              unsigned* range_flag, unsigned range_tag) {
  if (skip && __builtin_nontemporal_load(skip) >= skip_n) return;   // AR early exit (see ConvParams::skip)
  constexpr int D = 384;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const bool act = lane < 48;
  const int c = (act ? lane : 0) * 8;
  const float* x = in + (int64_t)row * in_ld + c;
  float v[8];
  {
    const float4 a = *reinterpret_cast<const float4*>(x), b = *reinterpret_cast<const float4*>(x + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  V8 o;
  ln384_row8(v, act, gamma + c, beta + c, eps, o.v);
  if (!act) return;
  // tiled: gemm_sp.hip's loader pieces [rows / 8][plane][6 blocks of 64 channels][8 rows][64] (ConvParams::x_tiled) - planes 6 x 512 halves apart
  RangeWatch rw;
  if (tiled) st_planes<NPL>(out + ((int64_t)(row >> 3) * (NPL * 6) + (c >> 6)) * 512 + (row & 7) * 64 + (c & 63), 6 * 512, o, rw);
  else st_planes<NPL>(out + (int64_t)row * (NPL * D) + c, D, o, rw);
  rw.flush(range_flag, range_tag);
}
}  // namespace

void launch_layernorm_planes(const float* in, int in_ld, const float* gamma, const float* beta, float eps, void* out, int M, hipStream_t s, int planes,
                             const int* skip, int skip_n, int tiled) {
  if (M <= 0) return;
  if (in_ld % 4 || (((uintptr_t)in | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15)) throw std::runtime_error("layernorm (planes): 16-byte alignment");
  if (tiled && M % 8) throw std::runtime_error("layernorm (planes): the tiled layout wants a multiple of 8 rows");
  if (planes == 2) hipLaunchKernelGGL(layernorm_planes_kernel<2>, dim3((M + 3) / 4), dim3(256), 0, s, in, in_ld, gamma, beta, eps, (f16*)out, M, skip, skip_n, tiled, range_ctx().flag, range_ctx().tag);
  else hipLaunchKernelGGL(layernorm_planes_kernel<3>, dim3((M + 3) / 4), dim3(256), 0, s, in, in_ld, gamma, beta, eps, (f16*)out, M, skip, skip_n, tiled, range_ctx().flag, range_ctx().tag);
}
}  // namespace ttr
