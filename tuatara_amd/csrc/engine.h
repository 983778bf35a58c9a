// The engine of the MI355X-native tuatara hot path: declarations shared by its translation units.
//   engine.cpp          construction, weights -> device, workspaces, per-launch profile
//   engine_craft.cpp    the detector's forward pass (tuatara.cpp:363-394)
//   engine_parseq.cpp   the recogniser's forward pass (tuatara.cpp:307, :450-485)
//   engine_pages.cpp    image_to_data over batches of pages: detect / collect / recognise / finish, streamed batches, the sharded latency mode
//   comm.cpp            transports (RCCL, framed TCP), rendezvous, the ttr_comm_* entry points
//   capi.cpp            the C ABI of include/tuatara_hip.h;  capi_debug.cpp: the developer hooks of include/tuatara_hip_debug.h
//
// Re-implements the reference's pipeline function image_to_data (tuatara.cpp:314-512):
//   resize/pad/swap (:349-358) -> CRAFT (:363-394) -> get_detected_boxes (:400) ->
//   adjust_result_coordinates (:406) -> crop (:408-418) -> resize 128x32 (:436-448) ->
//   PARSeq (:450-485) -> argmax + Tokenizer (:486-505) -> format_output (:511)
// with every tensor op on the GPU (igemm.hip, craft_ops.hip, parseq_ops.hip, post_ops.hip)
// and only the per-component calipers + string decoding on the host (geometry.cpp).
// Differences by design: models are loaded once per engine (the reference reloads both
// per call, :336, :428), crops of all pages of a batch run as one PARSeq batch (the
// reference chunks by 4 over 6 threads, :452-475; logits are batch-invariant), the AR
// decoder keeps a K/V cache and leaves its loop once every crop has emitted EOS, like upstream PARSeq.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <chrono>
#include <dlfcn.h>
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <atomic>
#include <functional>
#include <condition_variable>
#include <thread>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tuatara_hip_debug.h"
#include "common.h"
#include "geometry.h"
#include "host_util.h"
#include "kernels.h"

namespace ttr {


extern thread_local std::string g_last_error;   // capi.cpp
extern unsigned long long* g_dec_dbg;   // device buffer for dec_ar phase stamps (diagnostics; engine_parseq.cpp)
// Kernel-selection knobs, per engine (ttr_engine_set_tuning); g_tuning_default seeds engines created afterwards (ttr_set_tuning)
struct Tuning {
  int dbg_bf16_out = 0;       // ttr_dbg_conv on a bf16 engine: take the kernel's bf16 output (the path the engine uses) instead of the f32 one
  int qkv_attn = 1;           // bf16 encoder: qkv projection + self-attention as one kernel (qkv_attn.hip) from qkv_attn_min crops on
  int qkv_attn_min = 160;
  int mlp_proj = 1;           // ... with the attention output projection in front of it in the same launch
  int mlp_min_rows = 49152;   // = 384 crops
  int dec_mlp_fused = 1;      // bf16 refinement pass: cross_out + norm2 + linear1 + GELU + linear2 + final norm through mlp_fused.hip
  int dec_mlp_min_rows = 16384;
  int mlp_fused = 1;          // bf16 encoder: norm2 + fc1 + GELU + fc2 + residual (+ the next LayerNorm) as one kernel (mlp_fused.hip)
  int tok_fuse = 1;           // bf16 AR steps: argmax of the previous step + token embedding + norm_c inside the self_kv skinny GEMM
  int ln_fuse = 1;            // bf16 decoder steps: LayerNorm computed inside the skinny GEMM's loader (gemm_sk ln_in)
  int fuse_first = 1;         // bf16: CRAFT conv1_1 fused into conv1_2's loader (conv3p FIRST)
  int qkv_attn_split = 1;     // split-operand engines, PARSeq encoder: qkv projection + self-attention as ONE launch (gemm_sp.hip, attention epilogue); needs enc_ln_pairs
  int skinny_split = 1;       // split-operand engines: linears of <= skinny_max_rows rows (the AR steps: one row per crop) on gemm_skx.hip
  int skinny_max_rows = 2048;
  int skx_ln_fuse = 1;        // split engines, <= 256 rows: the decoder's LayerNorm + linear pairs as one skinny launch (gemm_skx.hip, LayerNorm prologue)
  int embed_fold = 1;         // split engines, <= 256 crops: an AR step's embedding + norm_c inside its self_kv linear (gemm_skx.hip, token prologue)
  int argmax_fold = 1;        // split engines: an AR step's argmax inside the next step's embedding kernel (one dependent launch less per step)
  int sp_tiled_x = 1;         // ... and the encoder's activation planes (LayerNorm outputs, attention output, MLP hidden) are written and read as such pieces
  int sp_tiled_w = 1;         // split engines: gemm_sp.hip reads the recogniser's weight planes as contiguous 1-KiB pieces (Linear::wst)
  int ar_host_check = 10;     // split / fp32 engines, batches of <= 256 crops (the latency regime): from this AR step on the host looks at the done counter every fourth
                              // step and stops enqueuing steps once every crop has emitted EOS (upstream's loop does that check every step); 0 = never
  int enc_chunk = 0;          // crops per encoder group (0 = all crops at once)
  int enc_ln_pairs = 1;       // split-operand engines, PARSeq encoder: LayerNorm outputs as pairs (qkv and fc1 on three MFMAs per product: their inputs tolerate ~23.5 bits - 3 x 1280 crops: max |dlogit| 7.6e-4 vs 6.9e-4 with triples; proj and fc2 keep exact triples); 0 = triples
  int dec_planes = 1;         // split-operand engines: the decoder's layers hand each other planes (13 launches per AR step instead of 20); 0 = fp32 tensors + split passes
  int qkv_kv_pairs = 1;       // the qkv GEMM leaves the third plane of its K and V columns unwritten (the attention kernel reads them as pairs)
  int enc_fc2_pairs = 1;      // ... and the MLP hidden activation as pairs (fc2 on three MFMAs per product; 3 x 640 crops: max |dlogit| 5.1 - 6.7e-4 vs 5.6 - 7.6e-4 with triples); the attention output - the projection input - stays an exact triple: the one encoder linear whose result moves with the 24th bit (oracle/splitsim.py)
  int craft_products = 3;     // split-operand engines, CRAFT: 3 = activation pairs (~23.5 bits; the heat map stays at fp32 noise level), 4 = exact triples
  int gpu_calipers = 1;       // get_detected_boxes' per-component tail (dilated extremes -> hull -> minAreaRect, tuatara.cpp:162-179) on the GPU (post_ops.hip: ccl_rects_kernel);
                              // the host then reads 24 bytes per candidate instead of its row extremes and runs no calipers; 0 = geometry.cpp on the host
  int up_commute = 1;         // split engines, CRAFT's upconv2.0 / 3.0 / 4.0 (1x1 over cat(upsample(y), skip)): W_up . y at the low resolution, its bilinear upsample added in the
                              // epilogue of the skip half's 1x1 (gemm2.hip, ConvParams::up_z) - the upsampled tensors are never written; 0 = upsample kernel + two-source 1x1
  int head_tail = 1;          // ... and the two 1x1 layers behind conv_cls.4 inside its epilogue (no 16-channel tensors, two launches less); 0 = fp32 MFMA launches
  int first_fused = 1;        // pairs, pages that tile into 8 x 32 patches: conv1_1 evaluated inside conv1_2's kernel on the halo patch (conv3p.hip, FIRST on pairs) - the
                              // 64-channel full-resolution tensor is neither written nor read; bit-identical to the two launches (0: conv1_split_kernel + the plain tile)
  int head_packed = 1;        // ... with pairs: the 32-channel head tensors as 128-byte pixel rows [x0 | x1] and conv_cls.0 / .2 / .4 on packed pairs (two virtual
                              // chunks instead of three over zero-padded 64-channel rows: two thirds of the MFMAs, half the bytes); 0 = zero-padded rows
  int detector_only = 0;      // profiling: drop every detected box, so that a batch runs the detector + CCL only
  int bench_grid_boxes = 0;   // benchmark workload control (bench.py --boxes=grid40): the detector runs in full, then every page's boxes are replaced by a fixed 5 x 8 grid
  int split_planes = 1;       // split-operand engines: activations stay in planes between the layers (0: fp32 tensors + a split pass in front of every GEMM)
  int split_conv3p = 1;       // split-operand engines: 3x3 layers on the patch-stationary kernel (0: gemm2)
  int split_gemm = 1;         // split-operand engines: 0 = every layer on the fp32 MFMA kernel (A/B and tests)
  int craft_group = 16;       // pages per CRAFT launch group (activation workspace ~0.5 GB/page; every tensor must stay inside the 2 GiB window of 32-bit buffer offsets)
  int ar_tail_step = 12;      // with ar_early_exit: AR steps from this one on run as ONE launch of the fused kernel (which returns at once when the batch is done)
  int ar_crop_exit = 1;       // ... and, per crop, the two attention kernels of a step return for crops that have emitted EOS
  int ar_early_exit = 1;      // bf16 kernel-per-op AR loop: the steps' kernels return at once when every crop of the batch has emitted EOS (upstream's break)
  int decoder_mode = 1;       // 0 = kernel-per-op AR loop, 4/8/16 = fused kernel with that many crops per workgroup, else automatic
  int sp_hidden16 = 0;        // split engines, encoder MLP: the hidden activation (fc1 -> fc2, 1 GB per layer at 1280 crops) as 16-row pieces in the producing epilogue's lane
                              // order - a store instruction writes one contiguous KiB (gemm_sp.hip, x_tiled / out_tiled = 2); 2 = and those stores stream (nt) past the
                              // weights and activation rows the tiles re-read from L2; 0 = the loader's 8-row pieces
  int craft_lanes = 1;        // split engines, batches of >= 2 CRAFT launch groups: 2 = the groups on two staggered streams (Engine::lane_stream) - measured: no gain
                              // (403.0 / 403.3 against 403.7 / 404.3 pages/s: both lanes want the matrix pipe and the power budget); 1 = one after the other
  int recog_overlap = 1;      // streamed batches: the recogniser of batch j - 1 on a stream of its own, beside the detector of batch j (they share no buffer): the
                              // HBM-bound kernels and tile tails of one run under the other's matrix work.  Per-kernel times then include the neighbour's share of the chip
  int images_batch = 32;      // ttr_images_to_data: pages per streamed batch (same-sized images travel together)
  int range_guard = 1;        // split engines: every kernel that writes planes watches |x| < 65504 (split.h: RangeWatch); a tripped batch 1 = fails the call naming the
                              // layer, 2 = warns on stderr and returns the (saturated) result, 0 = not watched
  bool set(const std::string& k, int value) {
    if (k == "decoder_mode") decoder_mode = value;
    else if (k == "enc_chunk") enc_chunk = value;
    else if (k == "range_guard") range_guard = value;
    else if (k == "recog_overlap") recog_overlap = value;
    else if (k == "craft_lanes") craft_lanes = value == 2 ? 2 : 1;
    else if (k == "sp_hidden16") sp_hidden16 = value;
    else if (k == "images_batch") images_batch = value < 1 ? 1 : (value > 256 ? 256 : value);
    else if (k == "up_commute") up_commute = value;
    else if (k == "gpu_calipers") gpu_calipers = value;
    else if (k == "skinny_split") skinny_split = value;
    else if (k == "skinny_max_rows") skinny_max_rows = value;
    else if (k == "ar_host_check") ar_host_check = value;
    else if (k == "sp_tiled_w") sp_tiled_w = value;
    else if (k == "sp_tiled_x") sp_tiled_x = value;
    else if (k == "argmax_fold") argmax_fold = value;
    else if (k == "embed_fold") embed_fold = value;
    else if (k == "skx_ln_fuse") skx_ln_fuse = value;
    else if (k == "qkv_attn_split") qkv_attn_split = value;
    else if (k == "fuse_first") fuse_first = value;
    else if (k == "ln_fuse") ln_fuse = value;
    else if (k == "tok_fuse") tok_fuse = value;
    else if (k == "ar_early_exit") ar_early_exit = value;
    else if (k == "ar_crop_exit") ar_crop_exit = value;
    else if (k == "ar_tail_step") ar_tail_step = value;
    else if (k == "split_gemm") split_gemm = value;
    else if (k == "bench_grid_boxes") bench_grid_boxes = value;
    else if (k == "detector_only") detector_only = value;
    else if (k == "craft_products") craft_products = value == 4 ? 4 : 3;
    else if (k == "head_packed") head_packed = value;
    else if (k == "first_fused") first_fused = value;
    else if (k == "head_tail") head_tail = value;
    else if (k == "enc_ln_pairs") enc_ln_pairs = value;
    else if (k == "enc_fc2_pairs") enc_fc2_pairs = value;
    else if (k == "qkv_kv_pairs") qkv_kv_pairs = value;
    else if (k == "dec_planes") dec_planes = value;
    else if (k == "split_conv3p") split_conv3p = value;
    else if (k == "split_planes") split_planes = value;
    else if (k == "craft_group") craft_group = value < 1 ? 1 : (value > 32 ? 32 : value);
    else if (k == "mlp_fused") mlp_fused = value;     // 0 off, 1 from mlp_min_rows rows on, 2 always
    else if (k == "mlp_min_rows") mlp_min_rows = value;
    else if (k == "dec_mlp_fused") dec_mlp_fused = value;
    else if (k == "dec_mlp_min_rows") dec_mlp_min_rows = value;
    else if (k == "mlp_proj") mlp_proj = value;
    else if (k == "qkv_attn") qkv_attn = value;        // 0 off, 1 from qkv_attn_min crops on, 2 always
    else if (k == "qkv_attn_min") qkv_attn_min = value;
    else if (k == "dbg_bf16_out") dbg_bf16_out = value;
    else return false;
    return true;
  }
};
extern Tuning g_tuning_default;   // engine.cpp

// ------------------------------------------------------------------ small utilities
// roctx ranges around the host phases of a batch (rocprofv3 --marker-trace shows them next to the kernels).
// The marker library is looked up at run time: without it the ranges are no-ops.
struct Roctx {
  int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
  Roctx() {
    for (const char* n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
  }
};
inline Roctx& roctx() { static Roctx r; return r; }
struct RangeScope {
  bool on;
  explicit RangeScope(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~RangeScope() { if (on) roctx().pop(); }
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  size_t cap_seen = 0;   // the last capacity this buffer had (survives the window in which cap is 0 during a re-allocation)
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    void* old = p;
    p = nullptr; cap = 0;                       // a throwing hipFree must not leave a dangling pointer for the destructor
    if (old) TTR_HIP_CHECK(hipFree(old));
    // a buffer that grows again grows by a quarter at least: batches of slightly different crop counts must not re-allocate (hipFree waits for the whole
    // device) pass after pass - 288 GB of HBM make the slack free
    const size_t old_cap = cap_seen;
    size_t want = std::max(bytes, old_cap ? old_cap + old_cap / 4 : (size_t)0);
    want = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
    TTR_HIP_CHECK(hipMalloc(&p, want));
    cap = want; cap_seen = want;
  }
  template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// grow-only pinned host buffer: async copies to / from it need no staging and do not serialise the stream
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    void* old = p;
    p = nullptr; cap = 0;
    if (old) TTR_HIP_CHECK(hipHostFree(old));
    size_t want = (bytes + 65535) & ~(size_t)65535;
    TTR_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
  }
  template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
};

// A GEMM-shaped weight on the device: T [Cout_pad][K_pad] + f32 bias
struct Linear {
  DevBuf w, b;
  int cout = 0, k = 0;  // padded sizes as the kernel sees them
  int cout_valid = 0;   // != 0: the layer's own output count, below `cout` (PARSeq's head: 95 classes in 96 weight rows, so that it has f16 planes); the kernels get this as Cout
  DevBuf ws;            // split-operand engines (split.h): f16 [cout][3][k] = w0 | w0/2^11 | w1 of w S
  DevBuf wst;           // ... the same planes as the loader pieces of gemm_sp.hip (Engine::tile_planes): f16 [ceil(cout / 32) * 4][3][k / 64][8 rows][64]
  DevBuf wsp;           // ... CRAFT's 3x3 layers of 32 input channels, PACKED pairs form (conv3p.hip, NP = 2): f16 [cout][3][taps * 64], per tap
                        // plane 0 = [w0 (32) | w0 / 2^11 (32)], plane 1 unused, plane 2 = [w1 (32) | 0]; multiplies pixel rows [x0 (32) | x1 (32)]
  float inv_scale = 0;  // 1 / S
};

// ------------------------------------------------------------------ the engine
struct CraftConv { const char* name; int cin, cout, ks, dil; };

struct Result {
  std::vector<std::string> text;
  std::vector<float> bbox;   // 4 per item
  std::vector<int32_t> ids;  // 26 per item
};

struct CclBatch {   // device workspaces of the CCL stage for a batch of equally sized pages
  DevBuf tnorm, flags, parent, mm, area, bbox, maxt, cand_slot, cand, counters, rows;
  DevBuf rects, cal_pool, cal_ctr;   // GPU-side minAreaRect: per-candidate results, the hulls' scratch pool and its bump counter
  static constexpr int kCalCap = 4 << 20;   // floats (16 MB: ~230 k hull points per CRAFT group - half of that per lane when two detector lanes are active; a group that needs more falls back to the host's calipers)
  int pages = 0, npx = 0, max_cand = 0;
  int cal_cap_now = kCalCap;          // (tuning key gpu_calipers = 2 shrinks it to 512 floats: the host-fallback path under test)
  bool split_pool = false;            // two detector lanes are active for this batch: each takes half of the hulls' pool (set by detect_enqueue)
  CclBuffers view(int p0 = 0, int lane = 0) {   // the slices of pages p0.. (every array is strided by the page); lane: which half of the hulls' pool (two detector lanes)
    CclBuffers b;
    const size_t o = (size_t)p0 * npx;
    b.tnorm = tnorm.as<float>() + o; b.flags = flags.as<uint8_t>() + o; b.parent = parent.as<int>() + o; b.mm = mm.as<unsigned>() + (size_t)p0 * 4;
    b.area = area.as<int>() + o; b.bbox = bbox.as<int>() + o * 4; b.maxt = maxt.as<unsigned>() + o; b.cand_slot = cand_slot.as<int>() + o;
    b.cand = cand.as<int>() + (size_t)p0 * max_cand * 8; b.counters = counters.as<int>() + (size_t)p0 * 2; b.rows_packed = rows.as<int>() + o * 2;
    b.max_cand = max_cand;
    b.rects = rects.as<float>() + (size_t)p0 * max_cand * 8;
    b.cal_pool = cal_pool.as<float>() + (split_pool ? (size_t)lane * (kCalCap / 2) : 0); b.cal_ctr = cal_ctr.as<int>() + lane; b.cal_cap = std::min(cal_cap_now, split_pool ? kCalCap / 2 : kCalCap);
    return b;
  }
  void ensure(int pages_, int npx_, int max_cand_) {
    pages = pages_; npx = npx_; max_cand = max_cand_;
    const size_t n = (size_t)pages * npx;
    tnorm.ensure(n * 4); flags.ensure(n); parent.ensure(n * 4); mm.ensure((size_t)pages * 16);
    area.ensure(n * 4); bbox.ensure(n * 16); maxt.ensure(n * 4); cand_slot.ensure(n * 4);
    cand.ensure((size_t)pages * max_cand * 32); counters.ensure((size_t)pages * 8); rows.ensure(n * 8);
    rects.ensure((size_t)pages * max_cand * 32); cal_pool.ensure((size_t)kCalCap * 4); cal_ctr.ensure(64);
  }
};




#define TTR_NCCL_CHECK(expr)                                                                                          \
  do {                                                                                                                \
    ncclResult_t _r = (expr);                                                                                         \
    if (_r != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + ncclGetErrorString(_r) + " at " #expr);   \
  } while (0)

// Multi-GPU exchange in the C++ host (RCCL = the NCCL API of /opt/rocm/include/rccl/rccl.h): one process per GPU.
// The engine speaks to a Transport: two kinds of collective on device buffers, enqueued on a stream -
//   all_gather (data: the per-batch token ids on the engine's main stream; control: the small host-side exchanges - crop counts, status
//   headers, barriers - on the copy stream) and broadcast (latency mode's crop batch).
// RcclTransport is the product: two communicators per process (collectives of one communicator must be issued in one order on every rank,
// and the two kinds interleave differently from batch to batch).  SocketTransport carries the SAME calls over TCP through rank 0, staged
// through host memory, every call framed with a sequence number and its size so that a mismatched call sequence is an error, not a hang: it
// is what lets two ranks share ONE GPU (RCCL refuses two ranks on a device), i.e. what runs the engine's multi-rank code paths at world
// size 2 on a single-GPU box (tests/test_gpu_dist.py), and a fallback where RCCL cannot initialise.
struct Transport {
  virtual ~Transport() {}
  virtual const char* name() const = 0;
  virtual void all_gather(const void* d_send, void* d_recv, size_t bytes, bool control, hipStream_t stream) = 0;   // d_recv: world * bytes, by rank
  virtual void broadcast(void* d_buf, size_t bytes, int root, hipStream_t stream) = 0;
};

struct Comm {
  std::unique_ptr<Transport> tr;
  int rank = 0, world = 1;
  struct Engine* E = nullptr;
  DevBuf d_in, d_out;
  PinnedBuf h_in, h_out;
};

// The layout of a gathered batch (pure host logic, tests/test_comm_cpu.py drives it through ttr_gather_layout): every rank
// contributes its crops-per-page counts first; the payload then travels as `cap` = the largest rank total rows of 26 ids per rank.
// Nothing is truncated: a page may hold any number of crops.
struct GatherLayout {
  int world = 0, pages = 0, cap = 0;
  std::vector<int> total;      // crops of rank r
  std::vector<int64_t> first;  // row of (rank r, page p)'s first crop in the compacted [sum(total)][26] array
  static GatherLayout from_counts(const int32_t* counts, int world, int pages) {
    GatherLayout L;
    L.world = world; L.pages = pages; L.total.assign(world, 0); L.first.assign((size_t)world * pages + 1, 0);
    int64_t run = 0;
    for (int r = 0; r < world; ++r)
      for (int p = 0; p < pages; ++p) {
        const int c = counts[(size_t)r * pages + p];
        if (c < 0) throw std::runtime_error("gather: negative crop count");
        L.first[(size_t)r * pages + p] = run;
        run += c; L.total[r] += c;
      }
    L.first[(size_t)world * pages] = run;
    for (int r = 0; r < world; ++r) L.cap = std::max(L.cap, L.total[r]);
    return L;
  }
};

struct Engine {
  ttr_config cfg;
  Tuning tn = g_tuning_default;
  Precision prec;
  size_t es;  // element size of T
  hipStream_t stream = nullptr;
  std::unique_ptr<HostPool> host_pool;
  hipStream_t recog_stream = nullptr;             // tn.recog_overlap: the recogniser of a streamed batch runs here
  hipStream_t copy_stream = nullptr;              // device -> host copies of one page group's components while the next group's CRAFT runs
  hipEvent_t copy_ev = nullptr, done_ev[2] = {nullptr, nullptr};   // done_ev[slot]: a batch's token ids have landed
  hipEvent_t evr[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};   // per slot: before the packer, after it, after PARSeq
  // hand-over points of a batch are polled, not slept on: a blocking wait costs tens of microseconds of wake-up per sync
  static void spin_event(hipEvent_t e) {
    for (;;) {
      const hipError_t r = hipEventQuery(e);
      if (r == hipSuccess) return;
      if (r != hipErrorNotReady) hip_fail("hipEventQuery", r, __FILE__, __LINE__);
    }
  }
  std::vector<hipEvent_t> group_ev;               // per page group: component counters are on the host
  std::mutex mu;
  Tokenizer tok;

  // CRAFT
  std::map<std::string, Linear> craft;
  // split-operand engines: conv_cls.6 / conv_cls.8 as the epilogue of conv_cls.4's packed-pairs tile (ConvParams::tail_*, conv3p.hip)
  struct HeadTail { DevBuf w6, w8, b6, b8; float s6 = 0, s8 = 0; } head_tail;
  // PARSeq
  std::map<std::string, Linear> pq;               // linears by upstream name
  std::map<std::string, DevBuf> pqf;              // f32 vectors (LayerNorm params, pos embed, ...)
  DevBuf fc1_packed[12];                          // bf16 engines: encoder fc1 / fc2 / attn.proj weights as mlp_fused.hip's LDS images
  DevBuf proj_packed[12];                         // bf16 engines: encoder attn.proj weights k-step-major [12][384][32] (mlp_fused.hip, PROJ)
  DevBuf dec_ffn1_packed, dec_ffn2_packed, dec_co_packed;   // bf16 engines: decoder linear1 / linear2 / cross_attn.out_proj as mlp_fused images (refinement pass)
  DevBuf fc2_packed[12];                          // bf16 engines: encoder fc2 weights chunk-major [48][384][32] for mlp_fused.hip
  DevBuf qself;                                   // f32 [26][384]

  // workspaces
  std::vector<std::unique_ptr<DevBuf>> craft_ws_set[2];   // per-layer activations; [1]: the second detector lane's (tn.craft_lanes)
  int ws_sel = 0;                                 // which set ws() hands out
  std::vector<std::unique_ptr<DevBuf>>& craft_ws_cur() { return craft_ws_set[ws_sel]; }
  // tn.craft_lanes = 2: a batch's CRAFT launch groups alternate between the main stream and lane_stream, the second lane half a group behind the first
  // (lane_go: recorded by the first group once its full-resolution layers are through), so that one lane's matrix-bound first half runs beside the other's
  // HBM-bound U-Net tail; lane_done: the second lane's last CCL is enqueued (the main stream waits for it before the batch's closing events)
  hipStream_t lane_stream = nullptr;
  hipEvent_t lane_go = nullptr, lane_done = nullptr, resize_done = nullptr;
  bool lane_go_pending = false;                   // craft_forward_split records lane_go behind slice3.20 when set
  int craft_ws_npl = 0;                           // planes per value the split CRAFT workspaces were laid out for
  DevBuf pq_ws[24];
  DevBuf canvas, heat, staging_img, crops, rects_dev, logits, ar_logits, ids_dev, tokens;
  CclBatch ccl;
  PinnedBuf h_counters, h_cand, h_rows, h_rects_f, h_rects[2], h_ids[2];   // pinned staging of the small host <-> device transfers
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float stage_ms[4] = {0, 0, 0, 0};
  float host_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // host wall-clock splits of the last run_pages (ttr_last_host_us)
  static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  // optional per-launch timing of the igemm kernel (bench.py's roofline): events bracket every launch
  int profiling = 0;                                   // 0 off, 1 = CRAFT conv launches only, 2 = every conv / GEMM launch
  int prof_stage = 0;                                  // 0 = CRAFT convs, 1 = PARSeq encoder (ViT) + batched decoder GEMMs, 2 = per-step AR decoder GEMMs
  std::vector<hipEvent_t> prof_pool;
  // Every timed launch carries its KIND (which kernel family / which layer role), its ALGORITHMIC flops (2 x MACs of the layer: the figure
  // SURVEY.md section 8(d) prices the roofline with) and the flops the matrix cores EXECUTE for it (x 3 or x 4 in the split-operand mode).
  struct ProfRec { int stage; int kind; double alg, exec; int launches; double bytes = 0; };
  struct ProfKind { std::string name; int stage = 0; double ms = 0, alg = 0, exec = 0; long launches = 0; double bytes = 0; };   // bytes: ALGORITHMIC HBM bytes (every operand read once, every result written once), 0 = not tallied
  std::vector<ProfKind> prof_kinds;
  std::map<std::string, int> prof_kind_ids;
  int kind_id(const char* name) {   // (a kind is a name in a stage: the decoder's linears run in the batched stage and in the AR steps)
    const std::string key = std::string(name) + "#" + std::to_string(prof_stage);
    auto it = prof_kind_ids.find(key);
    if (it != prof_kind_ids.end()) return it->second;
    const int id = (int)prof_kinds.size();
    prof_kinds.push_back(ProfKind{name, prof_stage});
    prof_kind_ids[key] = id;
    return id;
  }
  bool seg_open = false;                               // profiling == 1: an event pair brackets a RUN of consecutive CRAFT conv launches of one kind
  int seg_kind = -1;                                   // (an event record between two kernels costs ~8 us of idle GPU)
  double seg_alg = 0, seg_exec = 0, seg_bytes = 0; int seg_launches = 0;
  std::vector<ProfRec> prof_recs;
  double prof_ms[3] = {0, 0, 0}, prof_flops[3] = {0, 0, 0};
  long prof_launches[3] = {0, 0, 0};

  template <class F> void timed(const char* kind, double alg_flops, double exec_flops, F&& launch, double alg_bytes = 0) {
    if (!profiling || (profiling == 1 && prof_stage != 0)) { launch(); return; }
    const int k = kind_id(kind);
    if (profiling == 1) {   // the timed region of bench.py: one event pair per run of same-kind convolutions, closed by the next kind or by prof_break()
      if (seg_open && seg_kind != k) prof_break();
      const size_t i = prof_recs.size();
      while (prof_pool.size() < 2 * (i + 1)) { hipEvent_t e; TTR_HIP_CHECK(hipEventCreate(&e)); prof_pool.push_back(e); }
      if (!seg_open) { TTR_HIP_CHECK(hipEventRecord(prof_pool[2 * i], stream)); seg_open = true; seg_kind = k; seg_alg = seg_exec = seg_bytes = 0; seg_launches = 0; }
      launch();
      seg_alg += alg_flops; seg_exec += exec_flops; seg_bytes += alg_bytes; ++seg_launches;
      return;
    }
    const size_t i = prof_recs.size();
    while (prof_pool.size() < 2 * (i + 1)) { hipEvent_t e; TTR_HIP_CHECK(hipEventCreate(&e)); prof_pool.push_back(e); }
    TTR_HIP_CHECK(hipEventRecord(prof_pool[2 * i], stream));
    launch();
    TTR_HIP_CHECK(hipEventRecord(prof_pool[2 * i + 1], stream));
    prof_recs.push_back(ProfRec{prof_stage, k, alg_flops, exec_flops, 1, alg_bytes});
  }
  void prof_break() {       // call before any kernel that is not a CRAFT convolution, and at the end of CRAFT
    if (!seg_open) return;
    TTR_HIP_CHECK(hipEventRecord(prof_pool[2 * prof_recs.size() + 1], stream));
    prof_recs.push_back(ProfRec{0, seg_kind, seg_alg, seg_exec, seg_launches, seg_bytes});
    seg_open = false;
  }
  void igemm(const ConvParams& p, double true_flops, const char* kind = "igemm");
  // split-operand engines: the layer as four f16 MFMAs per product (gemm2.hip, SP) when its shape allows; the fp32 inputs
  // are written as planes first (split_ops.hip)
  std::map<const void*, const Linear*> split_by_w;   // fp32 weight pointer -> its Linear (the one with the planes)
  DevBuf split_in[2];
  bool split_gemm(const ConvParams& p, double true_flops, const char* kind);
  void prof_break_if_craft() { if (prof_stage == 0) prof_break(); }
  // Folds the records whose events have completed (one stream: they complete in order).  With streamed batches the newest records
  // belong to a pass that is still running: they stay, with their events, for the next call.
  void prof_collect();

  // ---- construction
  void upload_linear(Linear& L, const float* w, int cout, int k, const float* bias, int cout_pad, int k_pad,
                     const std::vector<int>* kmap = nullptr, bool own = true);
  void tile_planes(Linear& L);   // L.ws -> L.wst (gemm_sp.hip's contiguous loader pieces)
  bool drop_w1 = false;          // (experiment, load_craft: the weight planes' low parts as zeros)
  void upload_f32(DevBuf& d, const float* p, size_t n);

  static const std::vector<CraftConv>& craft_convs() {
    static const std::vector<CraftConv> v = {
        {"slice1.0", 3, 64, 3, 1},     {"slice1.3", 64, 64, 3, 1},    {"slice1.7", 64, 128, 3, 1},   {"slice1.10", 128, 128, 3, 1},
        {"slice2.14", 128, 256, 3, 1}, {"slice2.17", 256, 256, 3, 1}, {"slice3.20", 256, 256, 3, 1}, {"slice3.24", 256, 512, 3, 1},
        {"slice3.27", 512, 512, 3, 1}, {"slice4.30", 512, 512, 3, 1}, {"slice4.34", 512, 512, 3, 1}, {"slice4.37", 512, 512, 3, 1},
        {"slice5.1", 512, 1024, 3, 6}, {"slice5.2", 1024, 1024, 1, 1},
        {"upconv1.0", 1536, 512, 1, 1}, {"upconv1.3", 512, 256, 3, 1}, {"upconv2.0", 768, 256, 1, 1}, {"upconv2.3", 256, 128, 3, 1},
        {"upconv3.0", 384, 128, 1, 1},  {"upconv3.3", 128, 64, 3, 1},  {"upconv4.0", 192, 64, 1, 1},  {"upconv4.3", 64, 32, 3, 1},
        {"conv_cls.0", 32, 32, 3, 1},   {"conv_cls.2", 32, 32, 3, 1},  {"conv_cls.4", 32, 16, 3, 1},  {"conv_cls.6", 16, 16, 1, 1},
        {"conv_cls.8", 16, 2, 1, 1}};
    return v;
  }

  void load_craft(const std::string& dir);

  // Row order of the qkv weight for the fused qkv + attention launch: row n = 192 h + c of the head-major matrix is tile channel c of head h,
  // c = 96 wn + 32 t + dd -> Q (t = 0), K (t = 1), V (t = 2), d = 32 wn + dd; upstream (timm) row = 384 t + 64 h + d
  static int qkv_tile_row(int n) {
    const int h = n / 192, c = n % 192, wn = c / 96, t = (c % 96) / 32, dd = c % 32;
    return 384 * t + 64 * h + 32 * wn + dd;
  }

  void load_head_tail(WeightFile& wf);
  void load_parseq(const std::string& dir);

  bool verbose = false;
  // ---- range guard of the split-operand mode (split.h: RangeWatch; kernels.h: range_ctx)
  // One sticky word PER STAGE AND SLOT (a word = 0, or the tag of the first layer whose planes left the f16 range): with streamed batches the detector of batch j
  // (main stream) and the recogniser of batch j - 1 (recog_stream) are on the GPU together, and a word shared by both blamed the wrong batch.  A word is copied to
  // its host slot and cleared by the stream that wrote it, right behind the kernels that could set it; the detector's is verified in detect_collect, before the
  // batch's boxes are used, the recogniser's in finish.
  enum { kRangeDet0 = 0, kRangeDet1 = 1, kRangeRec0 = 2, kRangeRec1 = 3, kRangeStage = 4, kRangeWords = 8 };
  DevBuf range_word;                               // [kRangeWords] device words
  PinnedBuf h_range;                               // [kRangeWords] host copies
  std::vector<std::string> range_names;            // tag - 1 -> layer name
  std::map<std::string, unsigned> range_ids;
  std::string range_scope;                         // prefix of the tags sgemm() forms from its profile kinds ("encoder.blocks.3.", "decoder.")
  unsigned* range_flag_ptr(int w = kRangeStage) { return prec == kSplit && tn.range_guard ? range_word.as<unsigned>() + w : nullptr; }
  void range_use(int w) { range_ctx().flag = range_flag_ptr(w); }   // this thread's launches from here on watch word w
  void range_tag(const std::string& layer);        // names the layer whose launches follow (this thread's range_ctx().tag)
  void range_fetch(int w);                         // on `stream`, behind the kernels that watch word w: its copy to the host, then its clear
  void range_verify(int w, const char* where);     // after the sync that covers the copy: a tripped word fails the call (or warns)
  // ---- multi-GPU (ttr_engine_attach_comm): every batch's token ids are all-gathered on the stream, device buffer to device buffer
  Comm* comm = nullptr;
  DevBuf gath_dev[2];
  PinnedBuf h_gath[2];
  struct Gathered { int world = 0, pages = 0; std::vector<int32_t> counts, ids; } last_gathered;
  // small host buffers of every rank, concatenated by rank (also the barrier): staged through device memory on the copy stream
  void allgather_host(const void* mine, size_t bytes, void* all);
  Engine(const std::string& dir, const ttr_config& c);
  ~Engine();

  // ---- CRAFT
  DevBuf& ws(size_t idx, size_t bytes, bool zero_new = false);

  void conv(const char* name, const void* in0, int C0, const void* in1, int C1, int relu0, int B, int H, int W, void* out, int act,
            float* out_f32 = nullptr, void* out_relu = nullptr, void* out_pool = nullptr, int pool_relu = 0);

  // canvas u8 [B][H][W][3] (device) -> heat f32 [B][H/2][W/2][2] (device)
  void craft_forward(const uint8_t* d_canvas, int B, int H, int W, float* d_heat);


  // ---- CRAFT, split-operand engines: every tensor between the convolutions lives as f16 planes ([pixel][x0 | x1 | x2], 6 bytes per
  // value); the convolutions' epilogues write them (bias, ReLU, ReLU copy, 2x2 max-pool fused), so no fp32 tensor and no separate
  // split pass exists up to the 32-channel head, which stays on the fp32 MFMA kernel (thin layers: 3 % of the FLOPs).
  void upconv_commuted(const char* name, const void* y_lo, int C0, const void* skip, int C1, int B, int H, int W, float* z, void* out);
  const uint8_t* sconv_canvas = nullptr;   // set around the sconv of slice1.3: its input tensor does not exist, conv1_1 is evaluated from this canvas inside the kernel (tn.first_fused)
  void sconv(const char* name, const void* in0, int C0, const void* in1, int C1, int B, int H, int W, void* out, int act,
             void* out_relu = nullptr, void* out_pool = nullptr, int pool_relu = 0, int out_planes = -1, int out_ld = 0, bool packed = false, float* tail_heat = nullptr);
  void craft_forward_split(const uint8_t* d_canvas, int B, int H, int W, float* d_heat);

  // ---- PARSeq
  // split-operand linear on planes: in [M][3 K] -> out (planes [M][3 out_ld] or fp32 [M][out_ld]) and / or out_f32 (+ fp32 residual)
  void sgemm(const Linear& L, const void* in_planes, int M, void* out, int out_ld, int act, int out_planes,
             float* out_f32 = nullptr, int out_f32_ld = 0, const float* resid = nullptr, int resid_ld = 0, int np = 4, int resid_mod = 0, int out_full_cols = 0,
             const char* kind = nullptr, int x_tiled = 0, int out_tiled = 0);
  // out = L(LayerNorm(x)) for the decoder's per-step rows: the skinny GEMM normalises its own activation rows (bf16, few rows);
  // otherwise the LayerNorm kernel writes `scratch` and the plain GEMM follows
  void ln_gemm(const float* x, const std::string& ln_name, float eps, void* scratch, const Linear& L, int M, void* out, int out_ld, int act,
               float* out_f32 = nullptr, int out_f32_ld = 0);
  void gemm(const Linear& L, const void* in, int M, void* out, int out_ld, int act, float* out_f32 = nullptr, int out_f32_ld = 0,
            const float* resid = nullptr, int resid_ld = 0, int resid_mod = 0);
  // AR early exit: while set, the decoder's per-step launches carry the batch's done counter (ConvParams::skip); only the skinny
  // GEMM and the per-row attention kernels honour it, which are the ones the bf16 AR steps use
  const int* cur_skip = nullptr; int cur_skip_n = 0;
  DevBuf ar_done;
  PinnedBuf h_ar_done;
  bool streaming_recog = false;   // set around the recogniser of a STREAMED batch: no host-side wait there (the stream holds the next batch's detector)
  size_t kvcache_zeroed = 0;
  void ln(const float* x, const std::string& name, float eps, void* out, int M);

  // The decoder tail of the split-operand engines: the same layers as decoder_tail() below, handing each other f16 planes (split.h) instead of
  // fp32 tensors + split passes - 13 launches per AR step instead of 20.  sa: planes [rows][3 * 384] (self-attention output).
  void decoder_tail_split(const void* sa, int N, int R, const float* resid_pos, int resid_mod, float* tgt, void* pa, void* pb, void* p1536, float* q384,
                          void* t384, const void* kvmem, float* logits_out, int logits_ld, const int* done_tok = nullptr, int done_col = 0);

  // decoder tail shared by the AR steps (R = 1) and the refinement pass (R = 26):
  // sa T [rows][384] -> logits f32 (row stride logits_ld)
  void decoder_tail(const void* sa, int N, int R, const float* resid_pos, int resid_mod, float* tgt, void* t384, void* t384b, void* t1536,
                    const void* kvmem, float* logits_out, int logits_ld, const int* done_tok = nullptr, int done_col = 0);

  // crops u8 [N][32][128][3] (device) -> logits f32 [N][26][95], ids i32 [N][26] (device); d_ar optional
  void parseq_forward(const uint8_t* d_crops, int N, float* d_logits, float* d_ar, int* d_ids);

  // ---- post-processing of one page's heat map: GPU CCL + host calipers
  struct PageBoxes { std::vector<RRect> det; };

  // CCL kernels of pages [p0, p0 + pages) of a batch of `total` pages, then their component counters -> host; group `g`'s event
  // fires when the counters have landed
  void ccl_launch(const float* d_heat, int p0, int pages, int total, int g, int H2, int W2, int lane = 0);
  // boxes of pages [p0, p0 + pages): waits for the group's counters, pulls candidates + row extremes over on the copy stream
  // (the main stream may already be running the next group's CRAFT), then the host calipers
  void ccl_collect(int p0, int pages, int g, int H2, int W2, std::vector<std::vector<RRect>>& det);
  // run f(page) for every page on the engine's host threads
  void parallel_pages(int pages, const std::function<void(int)>& f) { host_pool->run(pages, f); }

  // ---- the hot path over a batch of same-sized device pages, in four phases so that several batches can be in flight:
  //   detect_enqueue   resize + CRAFT + CCL kernels of a batch on the stream
  //   detect_collect   host: wait for each CRAFT group's components, calipers -> boxes -> crop rectangles
  //   recog_enqueue    crop rectangles -> packer -> PARSeq -> token ids back (stream)
  //   finish           wait, decode the ids per page
  // run_pages runs them in that order for one batch.  stream_push(j) runs detect_enqueue(j), recog_enqueue(j-1), detect_collect(j),
  // finish(j-2): the stream holds C(j) P(j-1) behind whatever is running, so the GPU works through the previous batch's recogniser
  // while the host turns batch j's components into boxes, and still has a whole recogniser queued while the host decodes batch j-2,
  // returns to the caller and comes back with batch j+1 — no GPU idle at any hand-over; one stream, every kernel still runs alone.
  // Host staging (crop rectangles, token ids) and the completion events exist twice (slot = batch parity).
  struct PageBatch {
    const uint8_t* d_pages = nullptr; int n = 0, h = 0, w = 0;
    CanvasGeom g{}; int H = 0, W = 0, H2 = 0, W2 = 0; size_t page_bytes = 0;
    std::vector<std::vector<RRect>> boxes;
    std::vector<int> rects, page_of;
    int N = 0, slot = 0, group = 16;
    int det_groups = -1;               // CRAFT groups enqueued for it (group_ev[det_groups]: behind the copy of its detector range word)
    bool live = false, enqueued = false;
    std::vector<int32_t> all_counts;   // with a communicator: crops per page of every rank [world][n]
    int cap = 0;                       // ... and the largest rank total (rows of the gathered payload per rank)
  };
  PageBatch q1, q2;        // streamed batches: q1 = boxes known (recogniser enqueued or not), q2 = older, recogniser enqueued, results not yet returned

  void detect_enqueue(PageBatch& B);

  // With a communicator attached a batch is a collective: a {status, pages} header travels before anything whose size depends on the
  // ranks' inputs, so that a rank that failed in its detector (`pre`: what detect_enqueue threw; or the box extraction below) or passed
  // another page count makes the call fail on EVERY rank - instead of leaving the others inside a gather that never completes.
  void detect_collect(PageBatch& B, std::exception_ptr pre = nullptr);
  void detect_collect_local(PageBatch& B);

  void recog_enqueue(PageBatch& B);

  void finish(PageBatch& B, std::vector<Result>& results);

  // the stage entry points (ttr_craft_heatmap, ttr_parseq_logits, ...) share workspaces with the batches: with the recogniser of a streamed batch on a stream of
  // its own they would race with it
  void refuse_while_streaming(const char* what) const { if (q1.live || q2.live) throw std::runtime_error(std::string(what) + ": streamed batches are in flight: call ttr_stream_flush until it returns none"); }
  void run_pages(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& results);

  // Latency mode (the reference's 6-thread fan-out over chunks of the crop batch, tuatara.cpp:450-485, across
  // GPUs): rank 0 detects and packs the crop batch, the batch is broadcast, rank r recognises the contiguous shard r of
  // ceil(N / world) crops, the ids are all-gathered, rank 0 decodes and returns the pages' results (the other ranks pass no pages and
  // return n empty results).  Collective over the engine's communicator.
  void run_pages_sharded(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& results);

  // Streamed form: returns the results of the batch pushed TWO calls earlier (prev_n = its page count, 0 for the first two pushes).
  // The pages of a batch must stay valid until its results have been returned (the crop packer reads them one push later).
  void stream_push(const uint8_t* d_pages, int n, int h, int w, std::vector<Result>& prev_results, int& prev_n);
  // results of the oldest batch in flight (prev_n = 0: none left)
  void stream_flush(std::vector<Result>& prev_results, int& prev_n);

  // ---- image_to_data over a LIST of host images of any sizes (the reference loads both models per call, tuatara.cpp:336, :428, and
  // takes one image per call): images of equal size travel together as batches of <= images_batch pages through the streamed path above.  Staging: four
  // pinned host buffers + four device buffers; a helper thread gathers batch j + 1's rows into its pinned buffer and starts its host-to-device copy on the
  // upload stream while the caller's thread is inside stream_push(j) - the detector of a batch waits for its pages' copy event, nothing else does.
  struct HostImage { const uint8_t* data; int h, w; std::ptrdiff_t row_stride; };
  static constexpr int kStageSlots = 4;   // a batch's pages must outlive its recogniser (two pushes later) while the helper fills the next slot
  PinnedBuf stage_host[kStageSlots];
  DevBuf stage_dev[kStageSlots];
  hipStream_t up_stream = nullptr;
  hipEvent_t up_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  // results[i] = what run_pages returns for image i alone.  An unreadable entry (null / empty / short stride: the reference's "Error reading image from file",
  // tuatara.cpp:344-347) and every image of a batch that failed on the GPU (e.g. the range guard) keep an empty result and are listed in `failed` (input
  // indices; first_error: the first failure's message); everything else is delivered - what a loop over image_to_data does with one bad image.
  void run_images(const std::vector<HostImage>& imgs, std::vector<Result>& results, std::vector<int>& failed, std::string& first_error);
  int stream_fail_age = 0;   // set by stream_push / stream_flush before they throw: 0 = the batch being pushed never entered the pipeline, 2 = the batch whose results were due failed and has left it
};

}  // namespace ttr

// ---- shared by the C ABI translation units
struct ttr_engine { std::unique_ptr<ttr::Engine> e; };
struct ttr_result { ttr::Result r; };

// Every entry point: the engine's lock, and the engine's device made current for the calling thread (HIP's current device is per
// thread: allocations, hipFuncSetAttribute and device queries inside the call must hit the device the stream belongs to).
struct EngineScope {
  std::lock_guard<std::mutex> lk;
  explicit EngineScope(ttr::Engine& E) : lk(E.mu) {
    TTR_HIP_CHECK(hipSetDevice(E.cfg.device));
    ttr::range_ctx().flag = E.range_flag_ptr(); ttr::range_ctx().tag = 0;   // this thread's launches belong to this engine until the scope ends
  }
  ~EngineScope() { ttr::range_ctx().flag = nullptr; ttr::range_ctx().tag = 0; }
};

#define TTR_GUARD_BEGIN try {
#define TTR_GUARD_END(rc)                                   \
  }                                                         \
  catch (const std::exception& ex) { ttr::g_last_error = ex.what(); return rc; } \
  catch (...) { ttr::g_last_error = "unknown error"; return rc; }

struct ttr_comm { std::unique_ptr<ttr::Comm> c; };
