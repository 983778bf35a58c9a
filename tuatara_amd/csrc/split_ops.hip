// Split-operand mode (split.h): elementwise helpers around the f16x4 GEMMs.
//   split_planes_kernel   fp32 [M][C] (row stride ld) -> f16 planes [M][3 C]  (x0 | x1 | x2), optional ReLU first
// They stand where a producer cannot write planes itself (LayerNorm / attention / pooling outputs in the first build of the mode);
// each is one streaming pass: 4 B read + 6 B written per element.
#include <algorithm>

#include "common.h"
#include "kernels.h"
#include "split.h"

namespace ttr {

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, int ld, f16* __restrict__ out, int64_t M, int C, int relu) {
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
    f16x8 p0, p1, p2;
    split3_x8(v, p0, p1, p2);
    f16* dst = out + m * (3 * (int64_t)C) + c;
    *reinterpret_cast<f16x8*>(dst) = p0; *reinterpret_cast<f16x8*>(dst + C) = p1; *reinterpret_cast<f16x8*>(dst + 2 * C) = p2;
  }
}

void launch_split_planes(const float* in, int ld, void* out, int64_t M, int C, int relu, hipStream_t s) {
  if (M <= 0) return;
  if (C % 8 || ld % 4 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) throw std::runtime_error("split_planes: C must be a multiple of 8 and the tensors 16-byte aligned");
  const int64_t total = M * (C >> 3);
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, s, in, ld, (f16*)out, M, C, relu);
}

}  // namespace ttr
