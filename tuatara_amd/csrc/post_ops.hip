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

__global__ void ccl_init_kernel(unsigned* mm, int* counters) {
  if (threadIdx.x == 0) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; mm[2] = 0xFFFFFFFFu; mm[3] = 0u; counters[0] = 0; counters[1] = 0; }
}

__global__ void minmax_kernel(const float* __restrict__ heat, int npx, unsigned* mm) {
  float tmin = INFINITY, tmax = -INFINITY, lmin = INFINITY, lmax = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    float2 v = reinterpret_cast<const float2*>(heat)[i];
    tmin = fminf(tmin, v.x); tmax = fmaxf(tmax, v.x); lmin = fminf(lmin, v.y); lmax = fmaxf(lmax, v.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    tmin = fminf(tmin, __shfl_xor(tmin, o)); tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    lmin = fminf(lmin, __shfl_xor(lmin, o)); lmax = fmaxf(lmax, __shfl_xor(lmax, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], f2ord(tmin)); atomicMax(&mm[1], f2ord(tmax));
    atomicMin(&mm[2], f2ord(lmin)); atomicMax(&mm[3], f2ord(lmax));
  }
}

// flags: bit0 text_score, bit1 link_score, bit2 combined
__global__ void binarize_kernel(const float* __restrict__ heat, int npx, const unsigned* __restrict__ mm, float low_text, float link_threshold,
                                float* __restrict__ tnorm, uint8_t* __restrict__ flags, int* __restrict__ parent,
                                int* __restrict__ area, int* __restrict__ bbox, unsigned* __restrict__ maxt, int* __restrict__ cand_slot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const float tmin = ord2f(mm[0]), tmax = ord2f(mm[1]), lmin = ord2f(mm[2]), lmax = ord2f(mm[3]);
  float2 v = reinterpret_cast<const float2*>(heat)[i];
  float tn = __fdiv_rn(v.x - tmin, tmax - tmin), ln = __fdiv_rn(v.y - lmin, lmax - lmin);  // IEEE division like torch
  int ts = tn > low_text, ls = ln > link_threshold, comb = ts | ls;
  tnorm[i] = tn;
  flags[i] = (uint8_t)(ts | (ls << 1) | (comb << 2));
  parent[i] = comb ? i : -1;
  area[i] = 0; maxt[i] = 0u; cand_slot[i] = -1;
  bbox[4 * i] = 0x7fffffff; bbox[4 * i + 1] = 0x7fffffff; bbox[4 * i + 2] = -1; bbox[4 * i + 3] = -1;
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

__global__ void ccl_merge_kernel(int* parent, int H, int W) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W || parent[i] < 0) return;
  int x = i % W, y = i / W;
  if (x > 0 && parent[i - 1] >= 0) uf_union(parent, i, i - 1);
  if (y > 0 && parent[i - W] >= 0) uf_union(parent, i, i - W);
}

__global__ void ccl_flatten_stats_kernel(int* parent, int H, int W, const float* __restrict__ tnorm, int* area, int* bbox, unsigned* maxt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W || parent[i] < 0) return;
  int r = uf_find_ro(parent, i);
  __hip_atomic_store(&parent[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // readers see the old ancestor or the root: both lead to r
  int x = i % W, y = i / W;
  atomicAdd(&area[r], 1);
  atomicMin(&bbox[4 * r], x); atomicMin(&bbox[4 * r + 1], y);
  atomicMax(&bbox[4 * r + 2], x); atomicMax(&bbox[4 * r + 3], y);
  atomicMax(&maxt[r], __float_as_uint(fmaxf(tnorm[i], 0.f)));
}

__global__ void candidates_kernel(const int* __restrict__ parent, int npx, const int* __restrict__ area, const int* __restrict__ bbox,
                                  const unsigned* __restrict__ maxt, float text_threshold, int min_area, int* cand_slot, int* cand, int* counters, int max_cand) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx || parent[i] != i) return;
  if (area[i] < min_area) return;                                   // :148
  if (__uint_as_float(maxt[i]) < text_threshold) return;            // :154
  int slot = atomicAdd(&counters[0], 1);
  if (slot >= max_cand) return;
  int h = bbox[4 * i + 3] - bbox[4 * i + 1] + 1;
  int off = atomicAdd(&counters[1], h);
  cand_slot[i] = slot;
  int* c = cand + 8 * slot;
  c[0] = i; c[1] = area[i]; c[2] = bbox[4 * i]; c[3] = bbox[4 * i + 1]; c[4] = bbox[4 * i + 2]; c[5] = bbox[4 * i + 3]; c[6] = off; c[7] = 0;
}

__global__ void rowext_init_kernel(const int* __restrict__ cand, const int* __restrict__ counters, int max_cand, int* rows_packed) {
  int slot = blockIdx.x;
  int n = min(counters[0], max_cand);
  if (slot >= n) return;
  const int* c = cand + 8 * slot;
  int h = c[5] - c[3] + 1, off = c[6];
  for (int r = threadIdx.x; r < h; r += blockDim.x) { rows_packed[2 * (off + r)] = 0x7fffffff; rows_packed[2 * (off + r) + 1] = -1; }
}

__global__ void rowext_kernel(const int* __restrict__ parent, const uint8_t* __restrict__ flags, int H, int W, const int* __restrict__ cand_slot,
                              const int* __restrict__ cand, int* rows_packed) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  int r = parent[i];
  if (r < 0) return;
  int slot = cand_slot[r];
  if (slot < 0) return;
  uint8_t f = flags[i];
  if ((f & 2) && !(f & 1)) return;  // segmap.setTo(0, link_score==1 & text_score==0)  (:160)
  const int* c = cand + 8 * slot;
  int x = i % W, y = i / W;
  int idx = c[6] + (y - c[3]);
  atomicMin(&rows_packed[2 * idx], x);
  atomicMax(&rows_packed[2 * idx + 1], x);
}

void launch_ccl(const float* heat, int H, int W, float text_threshold, float link_threshold, float low_text, int min_area, const CclBuffers& b, hipStream_t s) {
  const int npx = H * W;
  const dim3 blk(256), grid((npx + 255) / 256);
  hipLaunchKernelGGL(ccl_init_kernel, dim3(1), dim3(64), 0, s, b.mm, b.counters);
  hipLaunchKernelGGL(minmax_kernel, dim3(256), blk, 0, s, heat, npx, b.mm);
  hipLaunchKernelGGL(binarize_kernel, grid, blk, 0, s, heat, npx, b.mm, low_text, link_threshold, b.tnorm, b.flags, b.parent, b.area, b.bbox, b.maxt, b.cand_slot);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, blk, 0, s, b.parent, H, W);
  hipLaunchKernelGGL(ccl_flatten_stats_kernel, grid, blk, 0, s, b.parent, H, W, b.tnorm, b.area, b.bbox, b.maxt);
  hipLaunchKernelGGL(candidates_kernel, grid, blk, 0, s, b.parent, npx, b.area, b.bbox, b.maxt, text_threshold, min_area, b.cand_slot, b.cand, b.counters, b.max_cand);
  hipLaunchKernelGGL(rowext_init_kernel, dim3(b.max_cand), dim3(64), 0, s, b.cand, b.counters, b.max_cand, b.rows_packed);
  hipLaunchKernelGGL(rowext_kernel, grid, blk, 0, s, b.parent, b.flags, H, W, b.cand_slot, b.cand, b.rows_packed);
}

// ------------------------------------------------------------------ crop-batch packer
// One workgroup per crop: OpenCV fixed-point bilinear resample of image[y0:y1, x0:x1] to 32x128.
// The reference swaps channels before cropping (:349) and again after the resize (:441); the
// resize is per channel, so the net effect is the caller's channel order — no swap here.
__global__ void pack_crops_kernel(const uint8_t* __restrict__ image, int stride, const int* __restrict__ rects, uint8_t* __restrict__ out) {
  const int n = blockIdx.x;
  const int x0 = rects[4 * n], y0 = rects[4 * n + 1], x1 = rects[4 * n + 2], y1 = rects[4 * n + 3];
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

void launch_pack_crops(const uint8_t* image, int h, int w, int stride, const int* rects, uint8_t* out, int N, hipStream_t s) {
  (void)h; (void)w;
  if (N <= 0) return;
  hipLaunchKernelGGL(pack_crops_kernel, dim3(N), dim3(256), 0, s, image, stride, rects, out);
}

}  // namespace ttr
