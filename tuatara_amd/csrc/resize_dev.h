// Device restatement of OpenCV's 8-bit INTER_LINEAR resize (cv::resize as called at
// tuatara.cpp:223 and tuatara.cpp:440): 11-bit fixed-point weights, integer
// horizontal pass, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2 vertical pass,
// and the silent INTER_AREA substitution for an exact 2x2 decimation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttr {

struct ResizeAxis { int s0, s1; int w0, w1; };  // two source indices and their 11-bit weights

__device__ __forceinline__ int cv_floor_f(float v) { int i = (int)v; return i - (v < (float)i); }

// horizontal axis: weights are zeroed at the clamped borders
__device__ __forceinline__ ResizeAxis resize_axis_x(int d, int ssize, double scale) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = cv_floor_f(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  ResizeAxis a;
  a.s0 = s; a.s1 = s + 1 < ssize ? s + 1 : s;
  a.w0 = __float2int_rn((1.f - f) * 2048.f); a.w1 = __float2int_rn(f * 2048.f);
  return a;
}
// vertical axis: rows are clipped, weights are not
__device__ __forceinline__ ResizeAxis resize_axis_y(int d, int ssize, double scale) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = cv_floor_f(f);
  f -= (float)s;
  ResizeAxis a;
  a.s0 = s < 0 ? 0 : (s > ssize - 1 ? ssize - 1 : s);
  a.s1 = s + 1 < 0 ? 0 : (s + 1 > ssize - 1 ? ssize - 1 : s + 1);
  a.w0 = __float2int_rn((1.f - f) * 2048.f); a.w1 = __float2int_rn(f * 2048.f);
  return a;
}

struct ResizeGeom { int sh, sw, dh, dw; double scale_x, scale_y; int area2x2; int identity; };

__host__ __device__ inline ResizeGeom make_resize_geom(int sh, int sw, int dh, int dw) {
  ResizeGeom g;
  g.sh = sh; g.sw = sw; g.dh = dh; g.dw = dw;
  double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  g.scale_x = 1. / inv_x; g.scale_y = 1. / inv_y;
  g.identity = (sh == dh && sw == dw);
  g.area2x2 = (!g.identity && g.scale_x == 2.0 && g.scale_y == 2.0);
  return g;
}

// one output pixel (3 channels) of the resize of src (row stride `stride` bytes)
__device__ __forceinline__ void resize_pixel_u8c3(const uint8_t* src, int stride, const ResizeGeom& g, int dy, int dx, uint8_t* out3) {
  if (g.identity) {
    const uint8_t* s = src + (size_t)dy * stride + dx * 3;
    out3[0] = s[0]; out3[1] = s[1]; out3[2] = s[2];
    return;
  }
  if (g.area2x2) {
    const uint8_t* s0 = src + (size_t)(2 * dy) * stride + (2 * dx) * 3;
    const uint8_t* s1 = s0 + stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) out3[c] = (uint8_t)((s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2);
    return;
  }
  ResizeAxis ax = resize_axis_x(dx, g.sw, g.scale_x), ay = resize_axis_y(dy, g.sh, g.scale_y);
  const uint8_t* r0 = src + (size_t)ay.s0 * stride;
  const uint8_t* r1 = src + (size_t)ay.s1 * stride;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int h0 = r0[ax.s0 * 3 + c] * ax.w0 + r0[ax.s1 * 3 + c] * ax.w1;
    int h1 = r1[ax.s0 * 3 + c] * ax.w0 + r1[ax.s1 * 3 + c] * ax.w1;
    int v = (((ay.w0 * (h0 >> 4)) >> 16) + ((ay.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
    out3[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace ttr
