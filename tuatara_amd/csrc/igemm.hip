// Implicit-GEMM convolution / linear kernel for gfx950 (MI355X).
//
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] (+ resid) )
//
// A is never materialised: row m is output pixel (b,y,x) of an NHWC tensor and
// k = (tap, channel) walks the 3x3 (optionally dilated) or 1x1 window; a second
// source tensor supplies a *virtual* channel concat (U-Net up-blocks), ReLU can be
// applied on load (CRAFT's skips are pre-ReLU BatchNorm outputs).  Linear layers
// are the 1x1 case with H=1.  Replaces the LibTorch conv/linear calls inside the
// TorchScript modules run at tuatara.cpp:376 and tuatara.cpp:307.
//
// Tiling: 256 threads = 4 wave64 in a 2x2 grid, each wave owns WMT x WNT MFMA tiles
// of 16x16 (block tile 32*WMT x 32*WNT), K step 32.  Operand tiles are staged
// global -> VGPR -> LDS (zero fill for the conv halo and ragged edges happens in
// registers), double-buffered with one barrier per K step; 16-byte chunks are XOR
// swizzled so ds_read_b128 fragment reads are bank-conflict free for bf16.
// T = bf16 uses v_mfma_f32_16x16x32_bf16; T = float uses 8 x v_mfma_f32_16x16x4_f32
// on the same fragment layout (exact fp32 products: the parity mode).
#include "common.h"
#include "kernels.h"

namespace ttr {

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

template <typename T> __device__ __forceinline__ f32x4 mma16(const Frag<T>& a, const Frag<T>& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<bf16>(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<float>(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
  // lane l holds k = 8*(l>>4)+j for j=0..7; MFMA j contracts k in {j, 8+j, 16+j, 24+j}
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
  return c;
}

// ReLU on a packed 16-byte chunk without unpacking
template <typename T> __device__ __forceinline__ uint4 relu_chunk(uint4 v);
template <> __device__ __forceinline__ uint4 relu_chunk<bf16>(uint4 v) {
  auto f = [](uint32_t x) { uint32_t neg = (x >> 15) & 0x00010001u; return x & ~(neg * 0xFFFFu); };
  return make_uint4(f(v.x), f(v.y), f(v.z), f(v.w));
}
template <> __device__ __forceinline__ uint4 relu_chunk<float>(uint4 v) {
  auto f = [](uint32_t x) { return (x >> 31) ? 0u : x; };
  return make_uint4(f(v.x), f(v.y), f(v.z), f(v.w));
}

template <typename T> struct Swz;
// 64-byte rows: 4 rows per 256-B bank row; h = {0,2,3,1}[(row>>2)&3] makes every
// ds_read_b128 lane group {rows q, chunk c} hit 16 distinct 16-B slots.
template <> struct Swz<bf16> { static __device__ __forceinline__ int f(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; } };
template <> struct Swz<float> { static __device__ __forceinline__ int f(int row) { return (row >> 1) & 7; } };

template <typename T, int WMT, int WNT>
__global__ __launch_bounds__(256) void igemm_kernel(ConvParams p) {
  if (p.skip && __builtin_nontemporal_load(p.skip) >= p.skip_n) return;   // AR early exit (ConvParams::skip)
  constexpr int BM = 32 * WMT, BN = 32 * WNT, BK = 32;
  constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk
  constexpr int CPR = BK / EPC;                // chunks per tile row
  constexpr int NA = BM * CPR / 256, NB = (BN * CPR + 255) / 256;
  constexpr int ROWS_PER_PASS = 256 / CPR;
  static_assert(BM * CPR % 256 == 0, "A tile must divide over 256 threads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);  // [2][(BM+BN)*CPR] chunks
  constexpr int TILE_CHUNKS = (BM + BN) * CPR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks i and i+8 share an XCD (and its L2); give each
  // group of 8 M-tiles one XCD apiece and let it sweep the N-tiles so the A rows
  // are re-read from that L2.
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  int bid = blockIdx.x;
  int grp = bid / (8 * tilesN), rem = bid - grp * 8 * tilesN;
  int gm = min(8, tilesM - grp * 8);
  const int tm = grp * 8 + rem % gm, tn = rem / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int Ctot = p.C0 + p.C1;
  const int K = p.ks * p.ks * Ctot;
  const int nk = K / BK;
  const int HW = p.H * p.W;

  // per-thread A rows (fixed for the whole K loop)
  const int a_c = tid % CPR;
  int a_pix[NA], a_y[NA], a_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int row = tid / CPR + i * ROWS_PER_PASS;
    int m = m0 + row;
    if (m < p.M) {
      int r = m % HW;
      a_pix[i] = m; a_y[i] = r / p.W; a_x[i] = r % p.W;
    } else { a_pix[i] = -1; a_y[i] = -100000; a_x[i] = -100000; }
  }

  uint4 ra[NA], rb[NB];
  int tap = 0, cc = 0;  // K cursor: tap index and channel offset inside the virtual concat

  auto load_tile = [&]() {
    int dy = 0, dx = 0;
    if (p.ks == 3) { int ky = tap / 3, kx = tap - ky * 3; dy = (ky - 1) * p.dil; dx = (kx - 1) * p.dil; }
    const bool s1 = cc >= p.C0;
    const T* src = reinterpret_cast<const T*>(s1 ? p.in1 : p.in0);
    const int C = s1 ? p.C1 : p.C0, ch = (s1 ? cc - p.C0 : cc) + a_c * EPC;
    const int relu = s1 ? p.relu1 : p.relu0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int yy = a_y[i] + dy, xx = a_x[i] + dx;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
        v = *reinterpret_cast<const uint4*>(src + (int64_t)(a_pix[i] + dy * p.W + dx) * C + ch);
        if (relu) v = relu_chunk<T>(v);
      }
      ra[i] = v;
    }
    const int k0 = tap * Ctot + cc;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int q = tid + i * 256, row = q / CPR, c = q % CPR;
      int n = n0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row < BN && n < p.Cout) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.wgt) + (int64_t)n * K + k0 + c * EPC);
      rb[i] = v;
    }
    cc += BK;
    if (cc == Ctot) { cc = 0; ++tap; }
  };
  auto store_tile = [&](int buf) {
    uint4* A = lds + buf * TILE_CHUNKS;
    uint4* Bt = A + BM * CPR;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row = tid / CPR + i * ROWS_PER_PASS;
      A[row * CPR + (a_c ^ Swz<T>::f(row))] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int q = tid + i * 256, row = q / CPR, c = q % CPR;
      if (row < BN) Bt[row * CPR + (c ^ Swz<T>::f(row))] = rb[i];
    }
  };

  f32x4 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile();
  store_tile(0);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile();  // global loads in flight under the MFMAs below
    const uint4* A = lds + cur * TILE_CHUNKS;
    const uint4* Bt = A + BM * CPR;
    Frag<T> fa[WMT], fb[WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      int row = wm * (WMT * 16) + i * 16 + fr;
      if constexpr (sizeof(T) == 2) {
        uint4 v = A[row * CPR + (fg ^ Swz<T>::f(row))];
        fa[i].v = *reinterpret_cast<bf16x8*>(&v);
      } else {
        uint4 v0 = A[row * CPR + ((2 * fg) ^ Swz<T>::f(row))], v1 = A[row * CPR + ((2 * fg + 1) ^ Swz<T>::f(row))];
        *reinterpret_cast<uint4*>(&fa[i].v[0]) = v0; *reinterpret_cast<uint4*>(&fa[i].v[4]) = v1;
      }
    }
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      int row = wn * (WNT * 16) + j * 16 + fr;
      if constexpr (sizeof(T) == 2) {
        uint4 v = Bt[row * CPR + (fg ^ Swz<T>::f(row))];
        fb[j].v = *reinterpret_cast<bf16x8*>(&v);
      } else {
        uint4 v0 = Bt[row * CPR + ((2 * fg) ^ Swz<T>::f(row))], v1 = Bt[row * CPR + ((2 * fg + 1) ^ Swz<T>::f(row))];
        *reinterpret_cast<uint4*>(&fb[j].v[0]) = v0; *reinterpret_cast<uint4*>(&fb[j].v[4]) = v1;
      }
    }
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int j = 0; j < WNT; ++j) acc[i][j] = mma16<T>(fa[i], fb[j], acc[i][j]);
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + r
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * (WMT * 16) + i * 16 + fg * 4 + r;
      if (m >= p.M) continue;
      const int64_t rrow = p.resid ? (int64_t)(p.resid_mod ? m % p.resid_mod : m) * p.resid_ld : 0;
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int n = n0 + wn * (WNT * 16) + j * 16 + fr;
        if (n >= p.Cout) continue;
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[n];
        if (p.resid) v += p.resid[rrow + n];
        if (p.act == kActRelu) v = fmaxf(v, 0.f);
        else if (p.act == kActGelu) v = gelu_exact(v);
        if (p.out) reinterpret_cast<T*>(p.out)[(int64_t)m * p.out_ld + n] = (T)v;
        if (p.out_relu) reinterpret_cast<T*>(p.out_relu)[(int64_t)m * p.out_ld + n] = (T)fmaxf(v, 0.f);
        if (p.out_f32) p.out_f32[(int64_t)m * p.out_f32_ld + n] = v;
      }
    }
  }
}

template <typename T, int WMT, int WNT>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
  constexpr int BM = 32 * WMT, BN = 32 * WNT;
  constexpr int CPR = 32 / (16 / sizeof(T));
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const size_t lds = 2 * (size_t)(BM + BN) * CPR * 16;
  if (lds > 48 * 1024) {
    static PerDeviceOnce once;
  once.run([&] { TTR_HIP_CHECK(hipFuncSetAttribute((const void*)igemm_kernel<T, WMT, WNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); });
  }
  hipLaunchKernelGGL((igemm_kernel<T, WMT, WNT>), dim3(tilesM * tilesN), dim3(256), lds, s, p);
}

template <typename T>
static void launch_t(const ConvParams& p, hipStream_t s) {
  if (p.M <= 1024 && p.Cout > 32) return launch_cfg<T, 2, 2>(p, s);  // skinny (decoder) GEMMs: more, smaller tiles
  if (p.Cout <= 32) return launch_cfg<T, 4, 1>(p, s);
  if (p.Cout <= 64) return launch_cfg<T, 4, 2>(p, s);
  return launch_cfg<T, 4, 4>(p, s);
}

const char* igemm_check(const ConvParams& p) {
  const int Ctot = p.C0 + p.C1;
  if (p.ks != 1 && p.ks != 3) return "igemm: ks must be 1 or 3";
  if (Ctot % 32 || p.C0 % 32) return "igemm: channel counts must be multiples of 32";
  if (p.C1 && !p.in1) return "igemm: in1 missing";
  if ((!p.in0 && !p.ln_in) || !p.wgt) return "igemm: null operand";
  if (p.M != p.B * p.H * p.W) return "igemm: M != B*H*W";
  if (p.M <= 0 || p.Cout <= 0) return "igemm: empty problem";
  if (p.out_pool && (p.relu0 || p.relu1 || (p.C0 + p.C1) % 64 || p.C0 % 64 || p.Cout % 8)) return "igemm: the fused max-pool output needs a gemm2-eligible layer";
  if ((int64_t)p.M * (int64_t)(Ctot > p.Cout ? Ctot : p.Cout) >= (1ll << 40)) return "igemm: problem too large";
  return nullptr;
}

static int g_gemm_cfg = 0;
static int g_sk_max_rows = 2048;
static int g_ws_min_rows = 8192;
void set_gemm_ws_min_rows(int m) { g_ws_min_rows = m; }
void set_skinny_max_rows(int m) { g_sk_max_rows = m; }
int g_store_policy = 1;
void set_store_policy(int v) { g_store_policy = v; }
int skinny_max_rows() { return g_gemm_cfg == 0 || g_gemm_cfg >= 7 ? g_sk_max_rows : 0; }
void set_gemm_config(int cfg) { g_gemm_cfg = cfg; }
int gemm_config() { return g_gemm_cfg; }

void launch_igemm(Precision prec, const ConvParams& p_in, hipStream_t s) {
  if (const char* e = igemm_check(p_in)) throw std::runtime_error(e);
  ConvParams p = p_in;
  p.store_policy = g_store_policy;
  if (p.ln_in) {   // LayerNorm fused into the GEMM prologue: only the skinny kernel implements it
    if (prec != kBF16) throw std::runtime_error("igemm: fused LayerNorm input is bf16-only");
    return launch_gemm_sk(p, s);
  }
  if (prec == kBF16 && g_gemm_cfg >= 0) {
    if ((g_gemm_cfg == 0 || g_gemm_cfg >= 7) && p.M <= g_sk_max_rows && gemm_sk_check(p) == nullptr) return launch_gemm_sk(p, s);
    if ((g_gemm_cfg == 0 || g_gemm_cfg >= 7) && g_ws_min_rows > 0 && p.M >= g_ws_min_rows && p.Cout >= 512 && gemm_ws_check(p) == nullptr)
      return launch_gemm_ws(p, s);   // K <= 384, wide N: weights in registers (qkv +18 %, fc1 +13 % over gemm2; N = 384 is better off in gemm2)
    // 3x3 layers whose image tiles into 8x32 patches: the patch-stationary kernel moves 1.7-4x fewer bytes L2 -> LDS
    // (measured +5..47 % over gemm2 on every such CRAFT layer; profiles/r01_gemm_sweep_v2.txt)
    if ((g_gemm_cfg == 7 || (g_gemm_cfg == 0 && p.Cout >= 32)) && conv3p_check(p) == nullptr) return launch_conv3p(p, s);
    if (gemm2_check(p) == nullptr) return launch_gemm2(p, g_gemm_cfg >= 7 ? 0 : g_gemm_cfg, s);
  }
  if (p.out_pool) throw std::runtime_error("igemm: fused max-pool output is only available in the bf16 gemm2 kernel");
  if (prec == kBF16) launch_t<bf16>(p, s); else launch_t<float>(p, s);
}

}  // namespace ttr
