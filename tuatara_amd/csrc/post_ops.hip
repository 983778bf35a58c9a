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
  if (pg == 0 && b.cal_ctr) *b.cal_ctr = 0;   // the hulls' scratch pool starts empty for this group (ccl_rects_kernel)
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

// ------------------------------------------------------------------ minAreaRect on the GPU
// One lane per candidate runs geometry.cpp's component_to_rect / min_area_rect as they stand (same operations in the same order and precision, no
// contraction): dilated row extremes -> points -> sort (x, y) + unique -> monotone-chain hull (double cross products) -> rotating calipers (float32) ->
// centre, sides (double sqrt), angle (double atan2).  Scratch per candidate (points, hull, edge vectors, inverse lengths: 9 floats per point) comes from
// one pool by atomic bump; a candidate that does not get its share reports status 2 and the host computes the group as before.
namespace {
struct DPt { float x, y; };
#pragma clang fp contract(off)
__device__ inline double d_cross(const DPt& o, const DPt& a, const DPt& b) {
  return ((double)a.x - o.x) * ((double)b.y - o.y) - ((double)a.y - o.y) * ((double)b.x - o.x);
}
__device__ inline bool d_less(const DPt& a, const DPt& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }
// sort (unless the caller has: `sorted`) + unique in place, hull into h (capacity 2 n); returns the hull's size (n < 3 after unique: the points themselves)
__device__ int d_convex_hull(DPt* p, int n, DPt* h, bool sorted) {
  if (!sorted)
  for (int gap = n >> 1; gap > 0; gap = gap == 2 ? 1 : (int)(gap * 5 / 11)) {   // shell sort (any correct sort gives std::sort's sequence: duplicates are removed below)
    for (int i = gap; i < n; ++i) {
      const DPt v = p[i];
      int j = i;
      for (; j >= gap && d_less(v, p[j - gap]); j -= gap) p[j] = p[j - gap];
      p[j] = v;
    }
  }
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (m == 0 || p[i].x != p[m - 1].x || p[i].y != p[m - 1].y) p[m++] = p[i];
  n = m;
  if (n < 3) { for (int i = 0; i < n; ++i) h[i] = p[i]; return n; }
  int k = 0;
  for (int i = 0; i < n; ++i) { while (k >= 2 && d_cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
  for (int i = n - 2, t = k + 1; i >= 0; --i) { while (k >= t && d_cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
  return k - 1;
}
__device__ void d_calipers(const DPt* points, int n, DPt* vect, float* inv_len, float out[6]) {
  float minarea = 3.402823466e+38f;
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4];
  float orientation = 0.f, base_a, base_b = 0.f;
  DPt pt0 = points[0];
  float left_x = pt0.x, right_x = pt0.x, top_y = pt0.y, bottom_y = pt0.y;
  for (int i = 0; i < n; ++i) {
    if (pt0.x < left_x) left_x = pt0.x, left = i;
    if (pt0.x > right_x) right_x = pt0.x, right = i;
    if (pt0.y > top_y) top_y = pt0.y, top = i;
    if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
    const DPt pt = points[i + 1 < n ? i + 1 : 0];
    const double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
    vect[i].x = (float)dx; vect[i].y = (float)dy;
    inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    pt0 = pt;
  }
  {
    double ax = vect[n - 1].x, ay = vect[n - 1].y;
    for (int i = 0; i < n; ++i) {
      const double bx = vect[i].x, by = vect[i].y;
      const double convexity = ax * by - ay * bx;
      if (convexity != 0) { orientation = convexity > 0 ? 1.f : -1.f; break; }
      ax = bx; ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int best_left = 0, best_bottom = 0;
  float best_a = 1.f, best_b = 0.f, best_w = 0.f, best_h = 0.f;
  for (int k = 0; k < n; ++k) {
    const float dp0 = +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y;
    const float dp1 = -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y;
    const float dp2 = -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y;
    const float dp3 = +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y;
    float maxcos = dp0 * inv_len[seq[0]];
    int main_element = 0;
    { const float c = dp1 * inv_len[seq[1]]; if (c > maxcos) { main_element = 1; maxcos = c; } }
    { const float c = dp2 * inv_len[seq[2]]; if (c > maxcos) { main_element = 2; maxcos = c; } }
    { const float c = dp3 * inv_len[seq[3]]; if (c > maxcos) { main_element = 3; maxcos = c; } }
    {
      const int pindex = main_element == 0 ? seq[0] : main_element == 1 ? seq[1] : main_element == 2 ? seq[2] : seq[3];
      const float lead_x = vect[pindex].x * inv_len[pindex], lead_y = vect[pindex].y * inv_len[pindex];
      if (main_element == 0) { base_a = lead_x; base_b = lead_y; }
      else if (main_element == 1) { base_a = lead_y; base_b = -lead_x; }
      else if (main_element == 2) { base_a = -lead_x; base_b = -lead_y; }
      else { base_a = -lead_y; base_b = lead_x; }
    }
    if (main_element == 0) { if (++seq[0] == n) seq[0] = 0; }
    else if (main_element == 1) { if (++seq[1] == n) seq[1] = 0; }
    else if (main_element == 2) { if (++seq[2] == n) seq[2] = 0; }
    else { if (++seq[3] == n) seq[3] = 0; }
    float dx = points[seq[1]].x - points[seq[3]].x, dy = points[seq[1]].y - points[seq[3]].y;
    const float width = dx * base_a + dy * base_b;
    dx = points[seq[2]].x - points[seq[0]].x; dy = points[seq[2]].y - points[seq[0]].y;
    const float height = -dx * base_b + dy * base_a;
    const float area = width * height;
    if (area <= minarea) {
      minarea = area;
      best_left = seq[3]; best_bottom = seq[0];
      best_a = base_a; best_b = base_b; best_w = width; best_h = height;
    }
  }
  const float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
  const float C1 = A1 * points[best_left].x + points[best_left].y * B1;
  const float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
  const float idet = 1.f / (A1 * B2 - A2 * B1);
  out[0] = (C1 * B2 - C2 * B1) * idet;
  out[1] = (A1 * C2 - A2 * C1) * idet;
  out[2] = A1 * best_w; out[3] = B1 * best_w;
  out[4] = A2 * best_h; out[5] = B2 * best_h;
}
// geometry.cpp: min_area_rect on n points (pts is sorted in place) up to the calipers; the sides and the angle (double sqrt / atan2: the device's libm is
// not the host's, and one ulp there can move boundingRect's integer rounding) are left to the host (geometry.cpp: finish_min_area_rect).
// Returns the record's kind: 1 = calipers' out[6], 3 = a segment (v = x0, y0, x1, y1), 4 = a point (v = x, y)
__device__ int d_min_area_rect_raw(DPt* pts, int n, DPt* hull, DPt* vect, float* inv_len, float v[6], bool sorted) {
  const int hn = d_convex_hull(pts, n, hull, sorted);
  for (int i = 0; i < 6; ++i) v[i] = 0.f;
  if (hn > 2) { d_calipers(hull, hn, vect, inv_len, v); return 1; }
  if (hn == 2) { v[0] = hull[0].x; v[1] = hull[0].y; v[2] = hull[1].x; v[3] = hull[1].y; return 3; }
  v[0] = hull[0].x; v[1] = hull[0].y;
  return 4;
}
}  // namespace

// One wave per candidate, lane 0 at work: the hull's scratch lives in the workgroup's LDS (a dependent access costs an LDS round trip instead of an L2 one:
// the sort, the hull and the calipers are chains of them) when the candidate has at most kRectLds points, else in the global pool.
// Record per candidate: 8 floats {kind, v[0..5], -}: kind (int bits) 0 nothing left, 1 calipers' raw result, 2 the scratch pool was exhausted, 3 segment, 4 point.
constexpr int kRectLds = 448;   // points: 9 x 448 floats = 16 KB
constexpr int kRectGridX = 256; // workgroups per page: a workgroup walks the candidates slot, slot + 256, ... (a page has tens to hundreds; max_cand is 4096)
namespace {
__device__ void rect_of_candidate(const CclBuffers& b, int H, int W, size_t pg, int slot, float* lds_scratch, int* lds_rows, int* lds_n) {
  const int* c = b.cand + (pg * b.max_cand + slot) * 8;
  float* rr = b.rects + (pg * b.max_cand + slot) * 8;
  const int* rows = b.rows_packed + pg * (size_t)H * W * 2 + 2 * (size_t)c[6];
  {   // the candidate's row extremes into LDS, all lanes (lane 0 alone would wait out an L2 round trip per row)
    const int hrows = c[5] - c[3] + 1;
    if (hrows <= 1024) {
      for (int i = threadIdx.x; i < 2 * hrows; i += 64) lds_rows[i] = rows[i];
      __syncthreads();
      rows = lds_rows;
    }
  }
  // geometry.cpp: component_to_rect
  const int x = c[2], y = c[3], w = c[4] - c[2] + 1, h = c[5] - c[3] + 1, size = c[1];
  const int niter = (int)sqrt((double)(size * min(w, h) / (w * h) * 2));   // tuatara.cpp:166, integer inside the sqrt
  const int sx = max(0, x - niter), sy = max(0, y - niter);
  const int ex = min(W, x + w + niter + 1), ey = min(H, y + h + niter + 1);
  const int k = 1 + niter, a = k / 2, back = k - 1 - a;
  const int oy0 = max(sy, y - back), oy1 = min(ey - 1, y + h - 1 + a);
  const int nmax = 2 * max(oy1 - oy0 + 1, 0);
  if (nmax == 0) { if (threadIdx.x == 0) reinterpret_cast<int*>(rr)[0] = 0; return; }
  const bool in_lds = !(nmax > kRectLds || b.cal_cap < 9 * kRectLds);   // (a pool smaller than the LDS share: the fallback path under test)
  if (!in_lds && threadIdx.x != 0) return;                               // the pool path is lane 0's alone
  float* scratch = lds_scratch;
  if (!in_lds) {
    const int need = 9 * nmax;
    const int off = atomicAdd(b.cal_ctr, need);
    if (off < 0 || off + need > b.cal_cap) { reinterpret_cast<int*>(rr)[0] = 2; return; }
    scratch = b.cal_pool + off;
  }
  DPt* pts = reinterpret_cast<DPt*>(scratch);                 // nmax points
  DPt* hull = pts + nmax;                                      // 2 nmax
  DPt* vect = hull + 2 * nmax;                                 // nmax
  float* inv_len = reinterpret_cast<float*>(vect + nmax);      // nmax
  int n = 0;
  if (in_lds) {   // a row per lane; the points land in any order (they are sorted below, and equal points are equal)
    if (threadIdx.x == 0) *lds_n = 0;
    __syncthreads();
    for (int oy = oy0 + (int)threadIdx.x; oy <= oy1; oy += 64) {
      int mn = 0x7fffffff, mx = -1;
      for (int s = max(y, oy - a); s <= min(y + h - 1, oy + back); ++s) {
        const int* r = rows + 2 * (s - y);
        if (r[1] < 0) continue;
        mn = min(mn, r[0]); mx = max(mx, r[1]);
      }
      if (mx < 0) continue;
      mn = max(mn - back, sx); mx = min(mx + a, ex - 1);
      const int cnt = mx != mn ? 2 : 1;
      const int at = atomicAdd(lds_n, cnt);
      pts[at] = DPt{(float)mn, (float)oy};
      if (cnt == 2) pts[at + 1] = DPt{(float)mx, (float)oy};
    }
  } else
  for (int oy = oy0; oy <= oy1; ++oy) {
    int mn = 0x7fffffff, mx = -1;
    for (int s = max(y, oy - a); s <= min(y + h - 1, oy + back); ++s) {
      const int* r = rows + 2 * (s - y);
      if (r[1] < 0) continue;
      mn = min(mn, r[0]); mx = max(mx, r[1]);
    }
    if (mx < 0) continue;
    mn = max(mn - back, sx); mx = min(mx + a, ex - 1);
    pts[n++] = DPt{(float)mn, (float)oy};
    if (mx != mn) pts[n++] = DPt{(float)mx, (float)oy};
  }
  bool sorted = false;
  if (in_lds) {   // the sort on all lanes: a point's place = the number of points before it in (x, y, index) order; through the hull's space and back
    __syncthreads();
    n = *lds_n;
    for (int i = threadIdx.x; i < n; i += 64) {
      const DPt v = pts[i];
      int rank = 0;
      for (int j = 0; j < n; ++j) { const DPt u = pts[j]; rank += (d_less(u, v) || (u.x == v.x && u.y == v.y && j < i)) ? 1 : 0; }
      hull[rank] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) pts[i] = hull[i];
    __syncthreads();
    sorted = true;
    if (threadIdx.x != 0) return;
  }
  if (n == 0) { reinterpret_cast<int*>(rr)[0] = 0; return; }
  float v6[6];
  const int kind = d_min_area_rect_raw(pts, n, hull, vect, inv_len, v6, sorted);
  reinterpret_cast<int*>(rr)[0] = kind;
  for (int i = 0; i < 6; ++i) rr[1 + i] = v6[i];
}
}  // namespace

__global__ __launch_bounds__(64) void ccl_rects_kernel(CclBuffers b, int H, int W) {
  __shared__ float lds_scratch[9 * kRectLds];
  __shared__ int lds_rows[2 * 1024];
  __shared__ int lds_n;
  const size_t pg = blockIdx.y;
  const int count = min(b.counters[pg * 2], b.max_cand);        // (uniform over the workgroup - one wave, so the barriers below order its lanes' LDS accesses only)
  for (int slot = blockIdx.x; slot < count; slot += gridDim.x) {
    rect_of_candidate(b, H, W, pg, slot, lds_scratch, lds_rows, &lds_n);
    __syncthreads();                                             // the next candidate reuses the LDS
  }
}

void launch_ccl_rects(const CclBuffers& b, int pages, int H, int W, hipStream_t s) {
  if (!b.rects || !b.cal_pool || !b.cal_ctr) throw std::runtime_error("ccl_rects: no buffers");
  hipLaunchKernelGGL(ccl_rects_kernel, dim3(std::min(b.max_cand, kRectGridX), pages), dim3(64), 0, s, b, H, W);
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
