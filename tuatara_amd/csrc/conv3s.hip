// CRAFT's classification head (conv_cls.*: 3x3 32->32, 3x3 32->32, 3x3 32->16, 1x1 16->16, 1x1 16->2 at half
// resolution; upstream craft.py, run inside the TorchScript module called at tuatara.cpp:376) in bf16.
//
// These layers have 32 (16) channels: 25 MB per layer per page through HBM for 3.6 GFLOP — bandwidth, not MFMA,
// bound.  conv3s_kernel is the small-channel sibling of conv3p.hip: a workgroup (4 waves) owns an 8 x 32 pixel patch,
// pulls its 10 x 34 x 32-channel halo patch into LDS with one burst of LDS-DMA loads (64-byte pixel rows, chunk XOR
// (slot>>1)&3 => conflict-free ds_read_b128 at any shift), keeps ALL nine taps' weights in registers (18 MFMA A
// fragments per lane) and issues 72 MFMAs per wave with no K loop and a single barrier.  With TAIL the last three layers
// run as one kernel: the 3x3 (32->16) result stays in registers, the two 1x1 layers are per-pixel register math
// (weights broadcast from LDS, 16 channels exchanged between two lanes), and only the f32 heat map [M][2] is written.
// Rounding points are those of the layer-per-kernel path (bf16 activations between layers, f32 heat map).
#include "common.h"
#include "kernels.h"

namespace ttr {

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int PH = 8, PW = 32, HW2 = PW + 2, NSLOT = (PH + 2) * HW2;   // 340 halo pixels
constexpr int NPIECE = (NSLOT + 15) / 16;                              // 1-KiB pieces of 16 pixels x 64 B
constexpr int XBYTES = NPIECE * 1024;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t s_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(bf16)x; }
}  // namespace

template <bool TAIL>
__global__ __launch_bounds__(256, 3) void conv3s_kernel(Conv3sParams p) {   // 4 waves per SIMD = 128 registers: without the cap hipcc hoists every (tap, tile) LDS address out of the patch loop (224 registers, two workgroups per CU)
  __shared__ __attribute__((aligned(1024))) unsigned char xs[XBYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int ptx = p.W / PW, pty = p.H / PH;
  const int M = p.B * p.H * p.W;
  // Persistent: the 18 weight fragments (and the tail's 1x1 tables) are fetched once per workgroup, not once per patch - as one
  // workgroup per patch the weights' 72 KB of per-lane 16-byte loads out-weighed the patch's 22 KB.  XCD x (blockIdx % 8) walks
  // a contiguous eighth of the patches, its workgroups side by side: neighbouring patches share their halo rows in that XCD's L2.
  const int ntiles = p.B * pty * ptx, per_xcd = (ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, t_first = xcd * per_xcd + (int)(blockIdx.x >> 3), t_end = min(ntiles, (xcd + 1) * per_xcd), t_step = (int)(gridDim.x >> 3);
  const __amdgpu_buffer_rsrc_t rsx = s_rsrc(p.in, (unsigned)((size_t)M * 64));
  // ---- all nine taps' weights as MFMA A fragments; output row q of tile jj is channel (q>>2)*8 + jj*4 + (q&3),
  // so a lane ends up with the 8 consecutive channels 8*fg .. 8*fg+7 of its pixel
  bf16x8 fw[2][9];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int n = (fr >> 2) * 8 + jj * 4 + (fr & 3);
    const bf16* wp = p.wgt + n * 288 + fg * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t) fw[jj][t] = *reinterpret_cast<const bf16x8*>(wp + t * 32);
  }
  // TAIL: conv_cls.6 (1x1 16->16) is ONE more MFMA per pixel tile - the rounded conv_cls.4 output of a lane (channels 8 fg .. + 7 of
  // its pixel; the padded channels 16-31 are zeros) IS the B fragment, W6 [out channel fr][k = 8 fg + e] the A fragment, b6 the C
  // operand - and leaves lane (pixel fr, fg) with out channels 4 fg .. + 3; conv_cls.8 (16->2) is 8 FMAs on those and a sum over fg.
  bf16x8 a6;
  f32x4 cb6;
  float w8a[4], w8b[4], b8v[2];
  if (TAIL) {
    a6 = *reinterpret_cast<const bf16x8*>(p.w6 + fr * 32 + fg * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) { cb6[r] = p.b6[4 * fg + r]; w8a[r] = (float)p.w8[4 * fg + r]; w8b[r] = (float)p.w8[32 + 4 * fg + r]; }
    b8v[0] = p.b8[0]; b8v[1] = p.b8[1];
  }
  float bv[8];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + fg * 8), b1 = *reinterpret_cast<const float4*>(p.bias + fg * 8 + 4);
    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
  }
  for (int tile = t_first; tile < t_end; tile += t_step) {
  const int b = tile / (pty * ptx), trem = tile - b * pty * ptx, ty = trem / ptx, tx = trem - ty * ptx;
  const int y0 = ty * PH, x0 = tx * PW;
  // ---- halo patch -> LDS (one burst)
  for (int piece = wave; piece < NPIECE; piece += 4) {
    const int pi = piece * 16 + (lane >> 2);
    const int pr = pi / HW2, pc = pi - pr * HW2;
    const int y = y0 - 1 + pr, x = x0 - 1 + pc;
    const int g = (lane & 3) ^ ((pi >> 1) & 3);
    const bool ok = pi < NSLOT && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const unsigned vo = ok ? (unsigned)((((b * p.H + y) * p.W + x) * 32 + g * 8) * 2) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(xs + piece * 1024), 16, vo, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 9 taps x 4 pixel tiles x 2 channel tiles.  Tile row r = 64*wave + 16*i + fr is patch pixel (r>>5, r&31).
  f32x4 acc[2][4];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[jj][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int pi0[4];
  int opq = 0;
  asm volatile("" : "+v"(opq));   // per patch: the 36 (tap, tile) LDS addresses below are not loop invariants hipcc may hoist out of the patch loop (and spill)
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int r = wave * 64 + i * 16 + fr; pi0[i] = (r >> 5) * HW2 + (r & 31) + opq; }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tapoff = (t / 3) * HW2 + (t % 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pi = pi0[i] + tapoff;
      const bf16x8 fx = *reinterpret_cast<const bf16x8*>(xs + pi * 64 + ((fg ^ ((pi >> 1) & 3)) << 4));
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[jj][t], fx, acc[jj][i], 0, 0, 0);
    }
  }

  // ---- epilogue
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 64 + i * 16 + fr;
    const int64_t m = ((int64_t)b * p.H + y0 + (r >> 5)) * p.W + x0 + (r & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = fmaxf(acc[0][i][e] + bv[e], 0.f); v[4 + e] = fmaxf(acc[1][i][e] + bv[4 + e], 0.f); }
    if (!TAIL) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
      *reinterpret_cast<bf16x8*>(p.out + m * 32 + fg * 8) = o;
    } else {
      bf16x8 xo;
#pragma unroll
      for (int e = 0; e < 8; ++e) xo[e] = (bf16)v[e];
      const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6, xo, cb6, 0, 0, 0);
      float o0 = 0.f, o1 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hr = bf16_round(fmaxf(d[r], 0.f));
        o0 = fmaf(w8a[r], hr, o0); o1 = fmaf(w8b[r], hr, o1);
      }
      o0 += __shfl_xor(o0, 16); o1 += __shfl_xor(o1, 16);
      o0 += __shfl_xor(o0, 32); o1 += __shfl_xor(o1, 32);
      o0 += b8v[0]; o1 += b8v[1];
      if (fg == 0) *reinterpret_cast<float2*>(p.heat + m * 2) = make_float2(o0, o1);
    }
  }
  __syncthreads();   // every wave is done with the patch before the next one's DMA overwrites it
  }
}

const char* conv3s_check(const Conv3sParams& p) {
  if (p.H % PH || p.W % PW || p.B <= 0) return "conv3s: H % 8, W % 32";
  if ((size_t)p.B * p.H * p.W * 64 >= ((size_t)1 << 31)) return "conv3s: tensor too large for 32-bit buffer offsets";
  if (!p.in || !p.wgt || !p.bias) return "conv3s: null operand";
  if (((uintptr_t)p.in & 15) || ((uintptr_t)p.wgt & 15) || ((uintptr_t)p.bias & 15)) return "conv3s: operand alignment";
  if (p.heat) { if (!p.w6 || !p.b6 || !p.w8 || !p.b8 || ((uintptr_t)p.heat & 7)) return "conv3s: tail operands"; }
  else if (!p.out || ((uintptr_t)p.out & 15)) return "conv3s: output";
  return nullptr;
}

static int g_c3s_wgs_per_cu = 3;   // persistent workgroups per CU (22 KB of LDS each; __launch_bounds__(256, 3): 3 waves per SIMD)
void set_conv3s_wgs_per_cu(int v) { g_c3s_wgs_per_cu = v < 1 ? 1 : (v > 8 ? 8 : v); }

void launch_conv3s(const Conv3sParams& p, hipStream_t s) {
  if (const char* e = conv3s_check(p)) throw std::runtime_error(e);
  const int tiles = p.B * (p.H / PH) * (p.W / PW);
  const int cus = device_cu_count(256);
  const int grid = std::max(8, std::min((tiles + 7) & ~7, cus * g_c3s_wgs_per_cu));   // a multiple of 8: an equal number of workgroups per XCD
  if (p.heat) hipLaunchKernelGGL(conv3s_kernel<true>, dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(conv3s_kernel<false>, dim3(grid), dim3(256), 0, s, p);
}

}  // namespace ttr
