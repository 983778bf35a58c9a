// GPU replacement for the OpenCV half of get_detected_boxes (tuatara.cpp:119-204) and
// for the crop + cv::resize step (tuatara.cpp:408-418, :436-441).
//
//   minmax      : min / max of both heat maps                                  (:120-121)
//   binarize    : (x-min)/(max-min), threshold (strict >), combine, init labels (:120-137)
//   ccl_merge   : 4-connected union-find, root = smallest pixel index           (:142)
//   ccl_flatten : path compression so every pixel points at its root
//   stats       : per-root area / bbox / max of normalised text map (atomics)   (:147-152, :162-165)
//   candidates  : roots with area >= 10 and max >= text_threshold               (:148, :154)
//   rowext      : per candidate, per row: min / max x of the link-masked pixels (:156-160)
//   pack_rows   : dense copy of those row extremes for the host
// The host then applies the rectangular dilation analytically on the row extremes
// (hull(S + K) only needs them), clips to the ROI (:166-174) and runs rotating
// calipers (:177-179) — geometry.cpp.  Label numbering = raster order of each
// component's first pixel = ascending root index (OpenCV's order).
#include "common.h"
#include "kernels.h"
#include "resize_dev.h"

namespace ttr {

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// All CCL kernels run over a batch of pages at once: blockIdx.y = page, every per-page array is
// strided by the page (npx pixels, max_cand candidates).
struct CclPage {
  const float* heat; float* tnorm; uint8_t* flags; int* parent; unsigned* mm; int* area; int* bbox; unsigned* maxt;
  int* cand_slot; int* cand; int* counters; int* rows_packed;
};
__device__ __forceinline__ CclPage ccl_page(const CclBuffers& b, const float* heat, int npx) {
  const size_t pg = blockIdx.y;
  CclPage c;
  c.heat = heat + pg * npx * 2; c.tnorm = b.tnorm + pg * npx; c.flags = b.flags + pg * npx; c.parent = b.parent + pg * npx;
  c.mm = b.mm + pg * 4; c.area = b.area + pg * npx; c.bbox = b.bbox + pg * npx * 4; c.maxt = b.maxt + pg * npx;
  c.cand_slot = b.cand_slot + pg * npx; c.cand = b.cand + pg * b.max_cand * 8; c.counters = b.counters + pg * 2;
  c.rows_packed = b.rows_packed + pg * npx * 2;
  return c;
}

__global__ void ccl_init_kernel(CclBuffers b, int pages) {
  int pg = blockIdx.x * blockDim.x + threadIdx.x;
  if (pg >= pages) return;
  unsigned* mm = b.mm + pg * 4; int* counters = b.counters + pg * 2;
  mm[0] = 0xFFFFFFFFu; mm[1] = 0u; mm[2] = 0xFFFFFFFFu; mm[3] = 0u; counters[0] = 0; counters[1] = 0;
}

// one workgroup reduction, then 4 atomics per workgroup (the per-wave form spent 48 us per page serialising on 4 words)
__global__ __launch_bounds__(256) void minmax_kernel(CclBuffers b, const float* __restrict__ heat_all, int npx) {
  const CclPage c = ccl_page(b, heat_all, npx);
  float tmin = INFINITY, tmax = -INFINITY, lmin = INFINITY, lmax = -INFINITY;
  const float4* h4 = reinterpret_cast<const float4*>(c.heat);   // two pixels per load (npx is even: H, W multiples of 16)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx / 2; i += gridDim.x * blockDim.x) {
    float4 v = h4[i];
    tmin = fminf(tmin, fminf(v.x, v.z)); tmax = fmaxf(tmax, fmaxf(v.x, v.z));
    lmin = fminf(lmin, fminf(v.y, v.w)); lmax = fmaxf(lmax, fmaxf(v.y, v.w));
  }
  if ((npx & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    float2 v = reinterpret_cast<const float2*>(c.heat)[npx - 1];
    tmin = fminf(tmin, v.x); tmax = fmaxf(tmax, v.x); lmin = fminf(lmin, v.y); lmax = fmaxf(lmax, v.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tmin = fminf(tmin, __shfl_xor(tmin, o)); tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    lmin = fminf(lmin, __shfl_xor(lmin, o)); lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  }
  __shared__ float red[4][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = tmin; red[w][1] = tmax; red[w][2] = lmin; red[w][3] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { tmin = fminf(tmin, red[k][0]); tmax = fmaxf(tmax, red[k][1]); lmin = fminf(lmin, red[k][2]); lmax = fmaxf(lmax, red[k][3]); }
    atomicMin(&c.mm[0], f2ord(tmin)); atomicMax(&c.mm[1], f2ord(tmax));
    atomicMin(&c.mm[2], f2ord(lmin)); atomicMax(&c.mm[3], f2ord(lmax));
  }
}

// flags: bit0 text_score, bit1 link_score, bit2 combined
__global__ void binarize_kernel(CclBuffers b, const float* __restrict__ heat_all, int npx, float low_text, float link_threshold) {
  const CclPage c = ccl_page(b, heat_all, npx);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const float tmin = ord2f(c.mm[0]), tmax = ord2f(c.mm[1]), lmin = ord2f(c.mm[2]), lmax = ord2f(c.mm[3]);
  float2 v = reinterpret_cast<const float2*>(c.heat)[i];
  float tn = __fdiv_rn(v.x - tmin, tmax - tmin), ln = __fdiv_rn(v.y - lmin, lmax - lmin);  // IEEE division like torch
  int ts = tn > low_text, ls = ln > link_threshold, comb = ts | ls;
  c.tnorm[i] = tn;
  c.flags[i] = (uint8_t)(ts | (ls << 1) | (comb << 2));
  c.parent[i] = comb ? i : -1;
  if (comb) {   // per-component statistics live at the component's root, and only a set pixel can become one: 28 B for ~5 % of the pixels
    c.area[i] = 0; c.maxt[i] = 0u; c.cand_slot[i] = -1;
    c.bbox[4 * i] = 0x7fffffff; c.bbox[4 * i + 1] = 0x7fffffff; c.bbox[4 * i + 2] = -1; c.bbox[4 * i + 3] = -1;
  }
}

// Lock-free union-find (ECL-CC style).  Hooking is a CAS on a *true* root (parent[a]==a),
// always larger root under smaller, so parents only ever decrease and the final root of a
// component is its smallest pixel index.  Path halving writes only to non-roots and only
// ancestor values, so it cannot undo a hook.  Loads are agent-scope (bypass the per-CU L1,
// which other CUs' stores never refresh).
__device__ __forceinline__ int uf_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int uf_find(int* parent, int i) {
  int p = uf_load(&parent[i]);
  while (p != i) {
    int gp = uf_load(&parent[p]);
    if (gp != p) __hip_atomic_store(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // halve the path
    i = p; p = gp;
  }
  return i;
}
// Read-only find for the passes that run after all unions are done.  (A halving find here could
// overwrite another thread's final `parent[i] = root` store with a mere ancestor, and the passes
// after this one treat parent[i] as the root: that lost pixels of a component once in a while.)
__device__ __forceinline__ int uf_find_ro(const int* parent, int i) {
  int p = uf_load(&parent[i]);
  while (p != i) { i = p; p = uf_load(&parent[i]); }
  return i;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }
    int old = atomicCAS(&parent[a], a, b);
    if (old == a) return;
    a = old;  // a was hooked by someone else meanwhile; retry from its new parent
  }
}

__global__ void ccl_merge_kernel(CclBuffers b, int H, int W) {
  int* parent = b.parent + (size_t)blockIdx.y * H * W;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W || parent[i] < 0) return;
  int x = i % W, y = i / W;
  if (x > 0 && parent[i - 1] >= 0) uf_union(parent, i, i - 1);
  if (y > 0 && parent[i - W] >= 0) uf_union(parent, i, i - W);
}

// A wave covers 64 consecutive pixels of one row (W is a multiple of 64 on every canvas the engine pads to; otherwise the
// per-pixel form below runs) and those mostly share a root: the lanes of one root are folded in the wave - count by popcount, x
// extremes from the first / last lane of the group (lanes are in x order), the text maximum by a masked wave maximum - and ONE lane
// issues the six atomics for the group.  Sums of ones, minima and maxima: the statistics are exactly those of the per-pixel form.
__global__ __launch_bounds__(256) void ccl_flatten_stats_kernel(CclBuffers b, int H, int W) {
  const CclPage c = ccl_page(b, nullptr, H * W);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool set = i < H * W && c.parent[i] >= 0;
  int r = -1;
  if (set) {
    r = uf_find_ro(c.parent, i);
    __hip_atomic_store(&c.parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // readers see the old ancestor or the root: both lead to r
  }
  const int x = i % W, y = i / W;
  const unsigned tbits = set ? __float_as_uint(fmaxf(c.tnorm[i], 0.f)) : 0u;
  if ((W & 63) != 0) {   // a wave may straddle two rows: per-pixel atomics
    if (set) {
      atomicAdd(&c.area[r], 1);
      atomicMin(&c.bbox[4 * r], x); atomicMin(&c.bbox[4 * r + 1], y);
      atomicMax(&c.bbox[4 * r + 2], x); atomicMax(&c.bbox[4 * r + 3], y);
      atomicMax(&c.maxt[r], tbits);
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(set);
  while (todo) {
    const int lead = __ffsll((long long)todo) - 1;
    const int rr = __shfl(r, lead);
    const unsigned long long grp = __ballot(set && r == rr);
    unsigned m = (set && r == rr) ? tbits : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == lead) {
      const int x1 = x - lane + (63 - __clzll((long long)grp));          // x of the group's last lane
      atomicAdd(&c.area[rr], __popcll(grp));
      atomicMin(&c.bbox[4 * rr], x); atomicMin(&c.bbox[4 * rr + 1], y);
      atomicMax(&c.bbox[4 * rr + 2], x1); atomicMax(&c.bbox[4 * rr + 3], y);
      atomicMax(&c.maxt[rr], m);
    }
    todo &= ~grp;
  }
}

__global__ void candidates_kernel(CclBuffers b, int npx, float text_threshold, int min_area) {
  const CclPage c = ccl_page(b, nullptr, npx);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx || c.parent[i] != i) return;
  if (c.area[i] < min_area) return;                                   // :148
  if (__uint_as_float(c.maxt[i]) < text_threshold) return;            // :154
  int slot = atomicAdd(&c.counters[0], 1);
  if (slot >= b.max_cand) return;
  int h = c.bbox[4 * i + 3] - c.bbox[4 * i + 1] + 1;
  int off = atomicAdd(&c.counters[1], h);
  c.cand_slot[i] = slot;
  int* cd = c.cand + 8 * slot;
  cd[0] = i; cd[1] = c.area[i]; cd[2] = c.bbox[4 * i]; cd[3] = c.bbox[4 * i + 1]; cd[4] = c.bbox[4 * i + 2]; cd[5] = c.bbox[4 * i + 3]; cd[6] = off; cd[7] = 0;
}

// rows_packed[r] = {INT_MAX, -1} for every row of every candidate (total rows <= npx)
__global__ void rowext_init_kernel(CclBuffers b, int npx) {
  const CclPage c = ccl_page(b, nullptr, npx);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c.counters[1] || i >= npx) return;
  c.rows_packed[2 * i] = 0x7fffffff; c.rows_packed[2 * i + 1] = -1;
}

// Per (candidate, row): the extent [min x, max x] of its pixels, for the host's rotating calipers.  A wave covers 64 consecutive pixels;
// when W % 64 == 0 they lie in one image row, x grows with the lane, and the pixels of one (candidate, row) entry are folded inside the
// wave - first lane = min, last lane = max, ONE lane issues the two atomics - instead of 64 lanes queueing on the same two addresses.
__global__ void rowext_kernel(CclBuffers b, int H, int W) {
  const CclPage c = ccl_page(b, nullptr, H * W);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int idx = -1, x = 0;
  if (i < H * W) {
    int r = c.parent[i];
    int slot = r >= 0 ? c.cand_slot[r] : -1;
    if (slot >= 0) {
      uint8_t f = c.flags[i];
      if (!((f & 2) && !(f & 1))) {   // segmap.setTo(0, link_score==1 & text_score==0)  (:160)
        const int* cd = c.cand + 8 * slot;
        x = i % W;
        idx = cd[6] + (i / W - cd[3]);
      }
    }
  }
  if (W % 64) {                       // a wave may straddle two rows: per-pixel atomics
    if (idx >= 0) { atomicMin(&c.rows_packed[2 * idx], x); atomicMax(&c.rows_packed[2 * idx + 1], x); }
    return;
  }
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  unsigned long long todo = __ballot(idx >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int lidx = __shfl(idx, leader);
    const unsigned long long same = __ballot(idx == lidx) & todo;
    const int first = __ffsll((long long)same) - 1, last = 63 - __clzll((long long)same);
    if (lane == first) { atomicMin(&c.rows_packed[2 * idx], x); atomicMax(&c.rows_packed[2 * idx + 1], x + (last - first)); }
    todo &= ~same;
  }
}

void launch_ccl(const float* heat, int pages, int H, int W, float text_threshold, float link_threshold, float low_text, int min_area, const CclBuffers& b, hipStream_t s) {
  const int npx = H * W;
  const dim3 blk(256), grid((npx + 255) / 256, pages);
  hipLaunchKernelGGL(ccl_init_kernel, dim3((pages + 63) / 64), dim3(64), 0, s, b, pages);
  hipLaunchKernelGGL(minmax_kernel, dim3(32, pages), blk, 0, s, b, heat, npx);
  hipLaunchKernelGGL(binarize_kernel, grid, blk, 0, s, b, heat, npx, low_text, link_threshold);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, blk, 0, s, b, H, W);
  hipLaunchKernelGGL(ccl_flatten_stats_kernel, grid, blk, 0, s, b, H, W);
  hipLaunchKernelGGL(candidates_kernel, grid, blk, 0, s, b, npx, text_threshold, min_area);
  hipLaunchKernelGGL(rowext_init_kernel, grid, blk, 0, s, b, npx);
  hipLaunchKernelGGL(rowext_kernel, grid, blk, 0, s, b, H, W);
}

// ------------------------------------------------------------------ crop-batch packer
// One workgroup per crop: OpenCV fixed-point bilinear resample of image[y0:y1, x0:x1] to 32x128.
// The reference swaps channels before cropping (:349) and again after the resize (:441); the
// resize is per channel, so the net effect is the caller's channel order — no swap here.
__global__ void pack_crops_kernel(const uint8_t* __restrict__ images, size_t page_bytes, int stride, const int* __restrict__ rects, uint8_t* __restrict__ out) {
  const int n = blockIdx.x;
  const int x0 = rects[5 * n], y0 = rects[5 * n + 1], x1 = rects[5 * n + 2], y1 = rects[5 * n + 3];
  const uint8_t* image = images + (size_t)rects[5 * n + 4] * page_bytes;
  uint8_t* o = out + (size_t)n * 32 * 128 * 3;
  if (x1 <= x0 || y1 <= y0) {
    for (int p = threadIdx.x; p < 32 * 128 * 3; p += blockDim.x) o[p] = 0;
    return;
  }
  ResizeGeom g = make_resize_geom(y1 - y0, x1 - x0, 32, 128);
  const uint8_t* src = image + (size_t)y0 * stride + (size_t)x0 * 3;
  for (int p = threadIdx.x; p < 32 * 128; p += blockDim.x) {
    uint8_t px[3];
    resize_pixel_u8c3(src, stride, g, p >> 7, p & 127, px);
    o[3 * p] = px[0]; o[3 * p + 1] = px[1]; o[3 * p + 2] = px[2];
  }
}

void launch_pack_crops(const uint8_t* images, size_t page_bytes, int stride, const int* rects5, uint8_t* out, int N, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(pack_crops_kernel, dim3(N), dim3(256), 0, s, images, page_bytes, stride, rects5, out);
}

}  // namespace ttr
