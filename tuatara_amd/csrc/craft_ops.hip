// CRAFT-side data-movement kernels (HBM-bound, 16-byte vectorised NHWC):
// page resize/pad/channel-swap (tuatara.cpp:349, :206-234), first-layer im2col with
// the /255 normalisation (:363-370), max-pools, bilinear x2 upsample and the heat-map
// channel extraction (:393-394).  The convolutions themselves are igemm.hip.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "resize_dev.h"

namespace ttr {

// ------------------------------------------------------------------ resize + pad + swap
// blockIdx.z = page of a batch of equally sized pages (source pages src_page bytes apart, canvases H*W*3 apart)
__global__ void resize_pad_u8_kernel(const uint8_t* src, size_t src_page, int sstride, ResizeGeom g, uint8_t* dst, int H, int W, int swap_rb) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  src += (size_t)blockIdx.z * src_page;
  dst += (size_t)blockIdx.z * H * W * 3;
  uint8_t px[3] = {0, 0, 0};
  if (y < g.dh && x < g.dw) resize_pixel_u8c3(src, sstride, g, y, x, px);
  uint8_t* d = dst + ((size_t)y * W + x) * 3;
  d[0] = swap_rb ? px[2] : px[0]; d[1] = px[1]; d[2] = swap_rb ? px[0] : px[2];
}

void launch_resize_pad_u8(const uint8_t* src, int sh, int sw, int sstride, uint8_t* dst, int th, int tw, int H, int W, int swap_rb, hipStream_t s,
                          int pages, size_t src_page) {
  ResizeGeom g = make_resize_geom(sh, sw, th, tw);
  hipLaunchKernelGGL(resize_pad_u8_kernel, dim3((W + 255) / 256, H, pages), dim3(256), 0, s, src, src_page, sstride, g, dst, H, W, swap_rb);
}

// ------------------------------------------------------------------ first layer im2col
// k = (ky*3+kx)*3 + c for k < 27, zero for k in [27,32); value = u8 / 255 (fp32 division as torch does)
template <typename T>
__global__ void im2col_l1_kernel(const uint8_t* __restrict__ canvas, T* __restrict__ out, int B, int H, int W) {
  int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t M = (int64_t)B * H * W;
  if (m >= M) return;
  int r = (int)(m % ((int64_t)H * W));
  int y = r / W, x = r % W;
  T vals[32];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int yy = y + ky - 1, xx = x + kx - 1;
      bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const uint8_t* p = canvas + (m + (int64_t)(ky - 1) * W + (kx - 1)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) vals[(ky * 3 + kx) * 3 + c] = ok ? (T)((float)p[c] / 255.0f) : (T)0.f;
    }
#pragma unroll
  for (int k = 27; k < 32; ++k) vals[k] = (T)0.f;
  uint4* o = reinterpret_cast<uint4*>(out + m * 32);
  const uint4* v = reinterpret_cast<const uint4*>(vals);
#pragma unroll
  for (int i = 0; i < (int)(32 * sizeof(T) / 16); ++i) o[i] = v[i];
}

void launch_im2col_l1(Precision prec, const uint8_t* canvas, void* out, int B, int H, int W, hipStream_t s) {
  int64_t M = (int64_t)B * H * W;
  dim3 grid((unsigned)((M + 255) / 256));
  if (prec == kBF16) hipLaunchKernelGGL(im2col_l1_kernel<bf16>, grid, dim3(256), 0, s, canvas, (bf16*)out, B, H, W);
  else hipLaunchKernelGGL(im2col_l1_kernel<float>, grid, dim3(256), 0, s, canvas, (float*)out, B, H, W);
}

// ------------------------------------------------------------------ first layer, direct (bf16)
// conv1_1 (3 -> 64, 3x3) without the im2col round trip: each wave builds the MFMA operand of 64
// pixels in registers (lane = pixel l&15, k = 8*(l>>4)..+7 with k = (ky*3+kx)*3+c, the same K
// order and the same bf16 values (u8/255) as im2col_l1_kernel), multiplies by the [64][32]
// weight held in 4 fragments, adds bias, applies ReLU and stores 16 bytes per lane.  Transposed
// MFMA + channel permutation as in gemm2.hip.  HBM-bound on the 128 B/pixel it writes.
__global__ __launch_bounds__(256) void conv1_direct_kernel(const uint8_t* __restrict__ canvas, const bf16* __restrict__ wgt /*[64][32]*/,
                                                          const float* __restrict__ bias, bf16* __restrict__ out, int B, int H, int W) {
  __shared__ bf16 lut[256];     // bf16(u8 / 255.0f): the division runs 256 times per workgroup instead of 27 times per pixel
  lut[threadIdx.x] = (bf16)((float)threadIdx.x / 255.0f);
  __syncthreads();
  const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
  const int64_t M = (int64_t)B * H * W;
  const int HW = H * W;
  bf16x8 fw[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int n = 32 * (jj >> 1) + (fr >> 2) * 8 + (jj & 1) * 4 + (fr & 3);
    fw[jj] = *reinterpret_cast<const bf16x8*>(wgt + n * 32 + fg * 8);
  }
  float bv[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[t][e] = bias[32 * t + fg * 8 + e];
  // the 8 (tap, channel) pairs this lane contributes: byte offset relative to the pixel's first byte, and the tap's (dy, dx)
  int off[8], dy[8], dx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fg * 8 + e, tap = k / 3;
    dy[e] = tap / 3 - 1; dx[e] = tap % 3 - 1;   // k >= 27: masked below
    off[e] = (dy[e] * W + dx[e]) * 3 + (k - tap * 3);
  }
  const int64_t nwaves = (int64_t)gridDim.x * 4, wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t g = wave0; g * 64 < M; g += nwaves) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = g * 64 + i * 16 + fr;
      bf16x8 fx;
      const bool mv = m < M;
      const int r = (int)((mv ? m : 0) % HW), y = r / W, x = r - y * W;
      const uint8_t* px = canvas + (mv ? m : 0) * 3;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int yy = y + dy[e], xx = x + dx[e];
        const bool ok = mv && (fg * 8 + e < 27) && yy >= 0 && yy < H && xx >= 0 && xx < W;
        fx[e] = ok ? lut[px[off[e]]] : (bf16)0.f;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[jj], fx, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = g * 64 + i * 16 + fr;
      if (m >= M) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)fmaxf(acc[2 * t][i][e] + bv[t][e], 0.f);
          o[4 + e] = (bf16)fmaxf(acc[2 * t + 1][i][e] + bv[t][4 + e], 0.f);
        }
        *reinterpret_cast<bf16x8*>(out + m * 64 + 32 * t + fg * 8) = o;
      }
    }
  }
}

void launch_conv1_direct(const uint8_t* canvas, const void* wgt, const float* bias, void* out, int B, int H, int W, hipStream_t s) {
  const int64_t M = (int64_t)B * H * W;
  const int grid = (int)std::min<int64_t>((M + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(conv1_direct_kernel, dim3(grid), dim3(256), 0, s, canvas, (const bf16*)wgt, bias, (bf16*)out, B, H, W);
}

// ------------------------------------------------------------------ pools / upsample (one 16-byte chunk per thread)
template <typename T> struct Chunk {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};
template <typename T> __device__ __forceinline__ Chunk<T> ld_chunk(const T* p) { Chunk<T> c; *reinterpret_cast<uint4*>(c.v) = *reinterpret_cast<const uint4*>(p); return c; }
template <typename T> __device__ __forceinline__ void st_chunk(T* p, const Chunk<T>& c) { *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(c.v); }

template <typename T>
__global__ void maxpool2x2_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int relu) {
  constexpr int N = Chunk<T>::N;
  const int Ho = H / 2, Wo = W / 2, Cc = C / N;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * Ho * Wo * Cc;
  if (idx >= total) return;
  int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  int xo = (int)(t % Wo); t /= Wo;
  int yo = (int)(t % Ho); int b = (int)(t / Ho);
  const T* p = in + (((int64_t)b * H + 2 * yo) * W + 2 * xo) * C + cc * N;
  Chunk<T> a = ld_chunk(p), b1 = ld_chunk(p + C), c = ld_chunk(p + (int64_t)W * C), d = ld_chunk(p + (int64_t)W * C + C), o;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float m = fmaxf(fmaxf((float)a.v[i], (float)b1.v[i]), fmaxf((float)c.v[i], (float)d.v[i]));
    if (relu) m = fmaxf(m, 0.f);
    o.v[i] = (T)m;
  }
  st_chunk(out + (((int64_t)b * Ho + yo) * Wo + xo) * C + cc * N, o);
}

// The same for W % 4 == 0, half the load instructions per output: a thread owns four consecutive outputs of a row and reads the
// 3 x 6 window above them once, coordinates clamped at the borders (a duplicated border pixel does not change a maximum): column maxima
// first, then three adjacent ones per output.  max is exact and order-free: bit-identical to the kernel below.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s1_strip_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int N = Chunk<T>::N;
  const int Cc = C / N, Wq = W / 4;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * H * Wq * Cc;
  if (idx >= total) return;
  const int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  const int xq = (int)(t % Wq); t /= Wq;
  const int y = (int)(t % H); const int b = (int)(t / H);
  const int x0 = 4 * xq;
  const T* base = in + (int64_t)b * H * W * C + cc * N;
  const int ry[3] = {max(y - 1, 0), y, min(y + 1, H - 1)};
  float col[6][N];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int x = min(max(x0 - 1 + c, 0), W - 1);
    const Chunk<T> v0 = ld_chunk(base + ((int64_t)ry[0] * W + x) * C), v1 = ld_chunk(base + ((int64_t)ry[1] * W + x) * C), v2 = ld_chunk(base + ((int64_t)ry[2] * W + x) * C);
#pragma unroll
    for (int i = 0; i < N; ++i) col[c][i] = fmaxf(fmaxf((float)v0.v[i], (float)v1.v[i]), (float)v2.v[i]);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    Chunk<T> o;
#pragma unroll
    for (int i = 0; i < N; ++i) o.v[i] = (T)fmaxf(fmaxf(col[d][i], col[d + 1][i]), col[d + 2][i]);
    st_chunk(out + (((int64_t)b * H + y) * W + x0 + d) * C + cc * N, o);
  }
}

template <typename T>
__global__ void maxpool3x3s1_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int N = Chunk<T>::N;
  const int Cc = C / N;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * H * W * Cc;
  if (idx >= total) return;
  int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  int x = (int)(t % W); t /= W;
  int y = (int)(t % H); int b = (int)(t / H);
  float m[N];
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = -INFINITY;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      Chunk<T> v = ld_chunk(in + (((int64_t)b * H + yy) * W + xx) * C + cc * N);
#pragma unroll
      for (int i = 0; i < N; ++i) m[i] = fmaxf(m[i], (float)v.v[i]);
    }
  Chunk<T> o;
#pragma unroll
  for (int i = 0; i < N; ++i) o.v[i] = (T)m[i];
  st_chunk(out + (((int64_t)b * H + y) * W + x) * C + cc * N, o);
}

// F.interpolate(mode='bilinear', align_corners=False) for an exact x2: src = 0.5*(dst+0.5)-0.5 clamped at 0
// (one spelled-out evaluation order for both kernels below - left to -ffp-contract the two contract differently in the last bit)
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, float lx0, float lx1, float ly0, float ly1) {
  const float top = fmaf(lx1, v01, lx0 * v00), bot = fmaf(lx1, v11, lx0 * v10);
  return fmaf(ly1, bot, ly0 * top);
}
template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int N = Chunk<T>::N;
  const int Ho = 2 * H, Wo = 2 * W, Cc = C / N;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * Ho * Wo * Cc;
  if (idx >= total) return;
  int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  int xo = (int)(t % Wo); t /= Wo;
  int yo = (int)(t % Ho); int b = (int)(t / Ho);
  float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
  int y0 = (int)sy, x0 = (int)sx;
  int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
  const T* base = in + (int64_t)b * H * W * C + cc * N;
  Chunk<T> v00 = ld_chunk(base + ((int64_t)y0 * W + x0) * C), v01 = ld_chunk(base + ((int64_t)y0 * W + x1) * C);
  Chunk<T> v10 = ld_chunk(base + ((int64_t)y1 * W + x0) * C), v11 = ld_chunk(base + ((int64_t)y1 * W + x1) * C), o;
#pragma unroll
  for (int i = 0; i < N; ++i)
    o.v[i] = (T)bilerp((float)v00.v[i], (float)v01.v[i], (float)v10.v[i], (float)v11.v[i], lx0, lx1, ly0, ly1);
  st_chunk(out + (((int64_t)b * Ho + yo) * Wo + xo) * C + cc * N, o);
}

// The same for even W, four times fewer load instructions per output: a thread owns the 2 x 4 output block under input pixels
// (i, j) and (i, j + 1) and fetches the 3 x 4 input window around them once (clamped at the borders) - 12 loads for 8 stores where the
// kernel above issues 32.  Every output is formed by the expression above from the same four source pixels: bit-identical results.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_block_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C) {
  constexpr int N = Chunk<T>::N;
  const int Wh = W / 2, Cc = C / N;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * H * Wh * Cc;
  if (idx >= total) return;
  const int cc = (int)(idx % Cc); int64_t t = idx / Cc;
  const int jp = (int)(t % Wh); t /= Wh;
  const int i = (int)(t % H); const int b = (int)(t / H);
  const int j = 2 * jp;
  const T* base = in + (int64_t)b * H * W * C + cc * N;
  const int ry[3] = {max(i - 1, 0), i, min(i + 1, H - 1)};
  const int cx[4] = {max(j - 1, 0), j, j + 1, min(j + 2, W - 1)};
  Chunk<T> w[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) w[a][c] = ld_chunk(base + ((int64_t)ry[a] * W + cx[c]) * C);
  const int Ho = 2 * H, Wo = 2 * W;
  // Window rows / columns of an output's source pixels are static: output row 2i reads rows (i - 1, i) = window (0, 1), row 2i + 1
  // rows (i, i + 1) = window (1, 2); columns 2j .. 2j + 3 read window columns (0, 1), (1, 2), (1, 2), (2, 3).  At a border the clamped
  // window holds the border pixel twice where the expression above names a neighbour with weight 0 (or the same pixel twice): same value.
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int yo = 2 * i + dy;
    const float sy = fmaxf(0.5f * ((float)yo + 0.5f) - 0.5f, 0.f);
    const float ly1 = sy - (float)(int)sy, ly0 = 1.f - ly1;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      const int xo = 2 * j + dx;
      const float sx = fmaxf(0.5f * ((float)xo + 0.5f) - 0.5f, 0.f);
      const float lx1 = sx - (float)(int)sx, lx0 = 1.f - lx1;
      constexpr int kC0[4] = {0, 1, 1, 2};
      const Chunk<T>&v00 = w[dy][kC0[dx]], &v01 = w[dy][kC0[dx] + 1], &v10 = w[dy + 1][kC0[dx]], &v11 = w[dy + 1][kC0[dx] + 1];
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e)
        o.v[e] = (T)bilerp((float)v00.v[e], (float)v01.v[e], (float)v10.v[e], (float)v11.v[e], lx0, lx1, ly0, ly1);
      st_chunk(out + (((int64_t)b * Ho + yo) * Wo + xo) * C + cc * N, o);
    }
  }
}

template <typename T>
__global__ void extract_heat_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int M) {
  int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  out[2 * m] = (float)in[(int64_t)m * ld];
  out[2 * m + 1] = (float)in[(int64_t)m * ld + 1];
}

#define TTR_DISPATCH(prec, kern, grid, block, s, ...)                                            \
  do {                                                                                           \
    if ((prec) == kBF16) hipLaunchKernelGGL(kern<bf16>, grid, block, 0, s, __VA_ARGS__);         \
    else hipLaunchKernelGGL(kern<float>, grid, block, 0, s, __VA_ARGS__);                        \
  } while (0)

static inline int chunk_elems(Precision p) { return p == kBF16 ? 8 : 4; }

static int g_upsample_block = 1;   // upsample2x: 2 x 4 output blocks per thread, maxpool3x3s1: 1 x 4 strips (0: one output chunk per thread)
void set_upsample_block(int v) { g_upsample_block = v; }

void launch_maxpool2x2(Precision prec, const void* in, void* out, int B, int H, int W, int C, int relu, hipStream_t s) {
  if (C % chunk_elems(prec) || (H | W) & 1) throw std::runtime_error("maxpool2x2: bad shape");
  int64_t total = (int64_t)B * (H / 2) * (W / 2) * (C / chunk_elems(prec));
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == kBF16) hipLaunchKernelGGL(maxpool2x2_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, B, H, W, C, relu);
  else hipLaunchKernelGGL(maxpool2x2_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C, relu);
}
void launch_maxpool3x3s1(Precision prec, const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
  if (C % chunk_elems(prec)) throw std::runtime_error("maxpool3x3: bad shape");
  if (W % 4 == 0 && g_upsample_block) {   // (the knob of the block-wise elementwise kernels)
    const int64_t ns = (int64_t)B * H * (W / 4) * (C / chunk_elems(prec));
    dim3 gs((unsigned)((ns + 255) / 256));
    if (prec == kBF16) hipLaunchKernelGGL(maxpool3x3s1_strip_kernel<bf16>, gs, dim3(256), 0, s, (const bf16*)in, (bf16*)out, B, H, W, C);
    else hipLaunchKernelGGL(maxpool3x3s1_strip_kernel<float>, gs, dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
    return;
  }
  int64_t total = (int64_t)B * H * W * (C / chunk_elems(prec));
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == kBF16) hipLaunchKernelGGL(maxpool3x3s1_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, B, H, W, C);
  else hipLaunchKernelGGL(maxpool3x3s1_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
}
void launch_upsample2x(Precision prec, const void* in, void* out, int B, int H, int W, int C, hipStream_t s) {
  if (C % chunk_elems(prec)) throw std::runtime_error("upsample2x: bad shape");
  if (W % 2 == 0 && W >= 4 && g_upsample_block) {
    const int64_t nb = (int64_t)B * H * (W / 2) * (C / chunk_elems(prec));
    dim3 gb((unsigned)((nb + 255) / 256));
    if (prec == kBF16) hipLaunchKernelGGL(upsample2x_block_kernel<bf16>, gb, dim3(256), 0, s, (const bf16*)in, (bf16*)out, B, H, W, C);
    else hipLaunchKernelGGL(upsample2x_block_kernel<float>, gb, dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
    return;
  }
  int64_t total = (int64_t)B * 4 * H * W * (C / chunk_elems(prec));
  dim3 grid((unsigned)((total + 255) / 256));
  if (prec == kBF16) hipLaunchKernelGGL(upsample2x_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, (bf16*)out, B, H, W, C);
  else hipLaunchKernelGGL(upsample2x_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, B, H, W, C);
}
void launch_extract_heat(Precision prec, const void* in, int ld, float* out, int M, hipStream_t s) {
  dim3 grid((M + 255) / 256);
  if (prec == kBF16) hipLaunchKernelGGL(extract_heat_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)in, ld, out, M);
  else hipLaunchKernelGGL(extract_heat_kernel<float>, grid, dim3(256), 0, s, (const float*)in, ld, out, M);
}

}  // namespace ttr
