// Host-callable launchers for every HIP kernel of the engine (definitions in *.hip).
#pragma once
#include <stdexcept>
#include "common.h"

namespace ttr {

// ---- split-operand range guard (split.h: RangeWatch).  The engine points the calling thread's context at its sticky flag word and names the layer it is about
// to launch (a tag = index into its table of names); every launcher of a kernel that writes planes hands both to the kernel.  Thread-local: an engine runs
// under its own lock on one thread at a time, two engines on two threads do not see each other's context.  flag == nullptr: not watched.
struct RangeCtx { unsigned* flag = nullptr; unsigned tag = 0; };
RangeCtx& range_ctx();                                   // engine.cpp
inline ConvParams with_range_ctx(const ConvParams& p) {  // a launcher's copy of its parameters, the guard filled in unless the caller set it
  ConvParams q = p;
  if (!q.range_flag) { q.range_flag = range_ctx().flag; q.range_tag = range_ctx().tag; }
  return q;
}

unsigned* tile_counters(hipStream_t s);                  // engine.cpp: ConvParams::tile_ctr for launches on that stream

// ---- igemm.hip
const char* igemm_check(const ConvParams& p);
void launch_igemm(Precision prec, const ConvParams& p, hipStream_t s);
// which kernel serves bf16 problems: -1 = igemm.hip only, 0 = automatic, 1..6 = force that gemm2 tile configuration,
// 7 = conv3p wherever it applies (gemm2 automatic elsewhere), 8 = never conv3p (gemm2 automatic)
void set_gemm_config(int cfg);
int gemm_config();

// ---- gemm2.hip (bf16, LDS-DMA staged)
const char* gemm2_check(const ConvParams& p);   // nullptr when gemm2 can run the problem
void launch_gemm2(const ConvParams& p, int cfg, hipStream_t s);
void set_gemm2_split_dbg(int v);     // split mode timing experiments (wrong results): 1 = no output stores
void set_gemm2_split_wreg(int v);    // split mode: 1 (default) w0b from the W0 tile in registers on the big tiles, 0 = staged copy (A/B)
void set_gemm2_split_stream(int v);   // split mode, pairs, plain GEMM shapes: gemm_sp.hip's kernel (1, default) or gemm2's (0)
void set_gemm2_split_stream4(int v);  // the same for activation triples (1, default)
bool gemm_sp_eligible(const ConvParams& p);
void set_gemm_sp_sched(int v);
void set_gemm_sp_few(int v);
void set_gemm_sp_epi(int v);
void set_gemm_sp_ks3(int v);
void set_gemm_sp_dbg(int v);
void set_gemm_sp_stag(int v);
void set_gemm_sp_stagger(int v);          // start delay by workgroup (ConvParams::cu_stagger), ticks of 10 ns; 0 = off
void set_gemm_sp_stagger_groups(int v);
void set_qkv_attn_dbg(int v);
void set_qkv_attn_stamps(unsigned long long* dev_buf);   // >= 24 * 16 u64, or null: phase stamps of the fused qkv + attention launch's workgroup 0
bool gemm_sp_ks3_eligible(const ConvParams& p);
void launch_gemm_sp_ks3(const ConvParams& p, int cfg, hipStream_t s);
void set_gemm_skx_ln_max_rows(int v);
void launch_gemm_sp(const ConvParams& p, int cfg, hipStream_t s);   // gemm_sp.hip: streamlined split-pairs GEMM (cfg 2 = 256 x 128 tiles, 6 = 128 x 256)
void set_gemm2_split_cfg(int v);
void set_gemm2_split_few(int v);     // split mode: force gemm2 tile configuration 1..6 (0 = automatic)
void set_gemm2_split_reuse(int v);   // split mode: 1 (default) reuse-order K loop, 0 plane-major order (A/B)
void set_gemm2_x_ring3(int v);   // 1 (default): activation tiles two K steps ahead where LDS allows; 0: one step (measurement)
// device table float2[1024] {Phi(x_i), Phi(x_i + 1/64) - Phi(x_i)}, x_i = -8 + i/64, of the GELU epilogues (built on first use)
const void* gelu_lut_for_current_device();
const void* gelu_hermite_lut_for_current_device();   // float4[512]: one cubic of Phi per interval of 1/32, the split mode's GELU (common.h: gelu_hermite)
// GELU by that table (g = the table, in LDS)
// ---- gemm_skx.hip (split-operand skinny linear, M <= 64 rows, whole K per workgroup: the AR steps of a single page)
bool gemm_skx_eligible(const ConvParams& p);
bool gemm_skx_ln_eligible(const ConvParams& p);   // ... with the LayerNorm prologue (ConvParams::ln_in: fp32 rows of 384, <= 256 of them)
void launch_gemm_skx(const ConvParams& p, hipStream_t s);
// ---- gemm_sk.hip (bf16 skinny GEMM, whole K resident: the per-step decoder linears)
const char* gemm_sk_check(const ConvParams& p);
void launch_gemm_sk(const ConvParams& p, hipStream_t s);
void set_skinny_max_rows(int m);   // problems with M <= m rows go to gemm_sk (0 = never)
void set_mlp_store_nt(int v);
void set_mlp_stamps(unsigned long long* dev_buf);       // >= 48*8 u64 or null
void launch_dec_cross_attn_mfma(const bf16* q, const bf16* kvmem, bf16* out, int N, int R, hipStream_t s);   // attn_dec2.hip: R <= 32 query rows per crop
void launch_dec_cross_attn_split(const float* q, const float* kvmem, void* out_planes, int N, int R, hipStream_t s);   // attn_cross_split.hip: f16x4 refinement pass, triples out
void set_dec_cross_rows_hsplit(int v);                   // head groups (workgroups) per row of the per-row cross-attention at <= 128 rows: 4 (default); 1: all 12 heads in one
void set_dec_cross_split(int v);                         // 1 (default): that kernel; 0: dec_cross_attn_crop_kernel
void set_dec_cross_crop(int v);                          // 1 (default): fp32 / f16x4 refinement-pass cross-attention with one workgroup per crop
void set_dec_cross_mfma(int v);                          // 1 (default): refinement-pass cross-attention on the matrix cores
void set_dec_self_refine(int v);                         // 1 (default): refinement-pass self-attention as one workgroup per crop
void set_conv3p_stamps(unsigned long long* dev_buf);    // >= 2*24*8 u64, or null: phase stamps of conv3p_first2 workgroup 0
void set_gemm_ws_stamps(unsigned long long* dev_buf);   // >= 2*24*8 u64, or null: phase stamps of gemm_ws workgroup 0
extern int g_store_policy;          // cache policy of the big streaming output stores: 0 default, 1 nt, 2 sc0 sc1 nt
void set_store_policy(int v);
// ---- conv1u.hip (CRAFT's upconv4.0 skip half: 1x1 over 128 channels + upsampled addend, persistent, weights resident)
bool conv1u_eligible(const ConvParams& p);
void launch_conv1u(const ConvParams& p, hipStream_t s);
// ---- qkv_attn4.hip (the fused qkv + attention tile as four-wave workgroups, two per CU; tuning key "qkv_attn4", off by default)
void set_qkv_attn4(int v);
int qkv_attn4_enabled();
void set_qkv_attn4_stamps(unsigned long long* dev_buf);
void launch_qkv_attn4(const void* x_pairs, const void* w_planes, const float* bias, float inv_scale, void* out_planes, int N, hipStream_t s, const void* w_tiled, int x_tiled,
                      int out_tiled);
void set_gemm2_up_resident(int v);
void set_gemm2_up_2d(int v);
// ---- conv3h.hip (CRAFT's packed-pairs head layers, persistent)
bool conv3h_eligible(const ConvParams& p);
void launch_conv3h(const ConvParams& p, hipStream_t s);
void set_conv3h_wgs_per_cu(int v);
void set_conv3h_stamps(unsigned long long* dev_buf);   // >= 24 * 8 u64 or null: phase stamps of workgroup 0, wave 0
void set_conv3p_head_persistent(int v);
void set_gemm_ws_dbg_flags(int f);
void set_gemm_ws_lean(int v);      // 0: always the run-time-activation epilogue (A/B and tests)
int skinny_max_rows();             // the current threshold (0 when gemm_sk is not in use)
// ---- gemm_ws.hip (bf16 linear, K <= 384, weights resident in registers: the ViT encoder's qkv / proj / fc1)
const char* gemm_ws_check(const ConvParams& p);
void launch_gemm_ws(const ConvParams& p, hipStream_t s);
void set_gemm_ws_min_rows(int m);   // linears with K <= 384 and at least this many rows use gemm_ws (0 = never)
// ---- conv3s.hip (bf16, CRAFT's 32-channel head: 3x3 conv, optionally with the two 1x1 layers behind it fused)
struct Conv3sParams {
  const bf16* in;        // [B][H][W][32]
  const bf16* wgt;       // [32][9][32]  (output channels beyond the layer's own are zero rows)
  const float* bias;     // [32]
  bf16* out;             // [B][H][W][32] = relu(conv3x3)                      (when heat == nullptr)
  const bf16* w6; const float* b6;   // tail: 1x1 16->16 (+ReLU), weights [32][32] zero padded
  const bf16* w8; const float* b8;   // tail: 1x1 16->2, weights [2][32] zero padded
  float* heat;           // tail output f32 [B][H][W][2]; non-null selects the fused conv_cls.4 + .6 + .8 kernel
  int B, H, W;
};
const char* conv3s_check(const Conv3sParams& p);
void launch_conv3s(const Conv3sParams& p, hipStream_t s);
void set_conv3s_wgs_per_cu(int v);   // persistent grid: workgroups per CU (default 4)
// ---- conv3p.hip (bf16 3x3 conv with a patch-stationary input tile)
const char* conv3p_check(const ConvParams& p);   // nullptr when conv3p can run the layer
void launch_conv3p(const ConvParams& p, hipStream_t s);
int conv3p_split_bn(const ConvParams& p);   // split-operand layers: the tile width launch_conv3p picks (128 / 64 / 32)
void set_conv3p_narrow_wide(int v);   // split-operand layers on 8 x 32 patches: 64-wide tiles when 128-wide ones would not fill the workgroup slots (1, default)
void set_conv3p_narrow_frac(int v);   // ... fewer than v / 4 tiles per CU (default 8)
void set_conv3p_deep_w(int v);           // 1 (default): those 32-wide tiles request a tap's weights three taps ahead (four weight stages)
void set_conv3p_deep_w64(int v);         // ... and the 64-wide tiles while fewer than v per CU (default: always; 0 = never)
void set_conv3p_narrowest_frac(int v);   // 32-wide tiles when the 64-wide ones number fewer than v / 4 per CU (default: always; 0 = never)
void set_conv3p_single_stage_max_cin(int c);
void set_conv3p_force_bn128(int v);
void set_conv3p_c64_waves(int w);
void set_conv3p_c128_waves(int w);
void set_conv3p_first_persistent(int v);
void set_conv3p_narrow_bn64(int v);
void set_conv3p_c32_tile(int v);   // Cout <= 32 on 32-wide tiles (default 1)
// n pseudo-random values, uniform in [-scale, scale) (benchmark inputs)
void launch_fill_random(Precision prec, void* p, size_t n, unsigned seed, float scale, hipStream_t s);

// ---- split_ops.hip (split-operand mode, split.h)
// fp32 [M][C] (row stride ld) -> f16 planes [M][3 C], optional ReLU first
void launch_split_planes(const float* in, int ld, void* out, int64_t M, int C, int relu, hipStream_t s, int planes = 3, const int* skip = nullptr, int skip_n = 0);
// planes [B][H][W][3 C] in and out: CRAFT's 3x3 / stride-1 max-pool and its bilinear x2 upsampling (bit-identical to the fp32 kernels)
void launch_maxpool3x3s1_planes(const void* in, void* out, int B, int H, int W, int C, hipStream_t s, int planes = 3);
void launch_upsample2x_planes(const void* in, void* out, int B, int H, int W, int C, hipStream_t s, int planes = 3);
// LayerNorm over 384 columns of fp32 rows (stride in_ld) -> planes [M][3 * 384]
void launch_layernorm_planes(const float* in, int in_ld, const float* gamma, const float* beta, float eps, void* out, int M, hipStream_t s, int planes = 3,
                             const int* skip = nullptr, int skip_n = 0, int tiled = 0);   // tiled: the planes as gemm_sp.hip's loader pieces (ConvParams::x_tiled)
// attn_split.hip: ViT encoder self-attention, qkv planes [N*128][3][1152] -> planes [N*128][3][384]
void launch_attn_enc_split(const void* qkv_planes, void* out_planes, int N, hipStream_t s);
// gemm_sp.hip: qkv projection + self-attention of the ViT encoder as one launch.  x_pairs [N*128][2][384] (LayerNorm output as pairs), weight planes
// [1152][3][384] and bias [1152] with the rows in head-major order (192 h + 64 c + d <- upstream 384 c + 64 h + d) -> out triples [N*128][3][384]
void launch_qkv_attn_split(const void* x_pairs, const void* w_planes, const float* bias, float inv_scale, void* out_planes, int N, hipStream_t s,
                           const void* w_tiled = nullptr, int x_tiled = 0, int out_tiled = 0);   // w_tiled: the same planes as loader pieces (Engine::tile_planes), optional
// CRAFT's conv1_1 + bias + ReLU from the u8 canvas into planes [M][3 * 64]; wgt_planes f16 [64][3][32] (k = (ky*3+kx)*3+c, 27 used)
void launch_conv1_split(const uint8_t* canvas, const void* wgt_planes, const float* bias, float out_scale, void* out, int B, int H, int W, hipStream_t s, int planes = 3);

// ---- craft_ops.hip
// OpenCV-style 8-bit INTER_LINEAR resize of src[sh,sw,3] to [th,tw], zero pad to [H,W], optional channel swap.
// `pages` equally sized pages in one launch: sources src_page bytes apart, canvases H*W*3 bytes apart
void launch_resize_pad_u8(const uint8_t* src, int sh, int sw, int sstride, uint8_t* dst, int th, int tw, int H, int W, int swap_rb, hipStream_t s,
                          int pages = 1, size_t src_page = 0);
// canvas u8 [B,H,W,3] -> first-layer im2col matrix T [B*H*W][32] (27 taps*channels, /255, zero padded)
void launch_im2col_l1(Precision prec, const uint8_t* canvas, void* out, int B, int H, int W, hipStream_t s);
// bf16 only: conv1_1 + bias + ReLU straight from the u8 canvas (same arithmetic as im2col_l1 + igemm, no [M][32] round trip)
void launch_conv1_direct(const uint8_t* canvas, const void* wgt /*bf16 [64][32]*/, const float* bias, void* out /*bf16 [M][64]*/, int B, int H, int W, hipStream_t s);
void launch_maxpool2x2(Precision prec, const void* in, void* out, int B, int H, int W, int C, int relu, hipStream_t s);
void launch_maxpool3x3s1(Precision prec, const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
void launch_upsample2x(Precision prec, const void* in, void* out, int B, int H, int W, int C, hipStream_t s);
void set_upsample_block(int v);   // 1 (default): a thread forms a 2 x 4 output block from one 3 x 4 input window  // in [B,H,W,C] -> out [B,2H,2W,C]
// T [M][ld] -> f32 [M][2] (the two heat-map channels)
void launch_extract_heat(Precision prec, const void* in, int ld, float* out, int M, hipStream_t s);

// ---- parseq_ops.hip
void launch_patchify(Precision prec, const uint8_t* crops, void* out, int N, int ld, hipStream_t s);   // out [N*128][ld], ld >= 96: columns 96 .. ld-1 are zeroed
void launch_layernorm(Precision prec, const float* in, int in_ld, const float* gamma, const float* beta, float eps, void* out, int out_ld, int M, int D, hipStream_t s,
                      const int* skip = nullptr, int skip_n = 0);
void launch_attn_enc(Precision prec, const void* qkv, void* out, int N, hipStream_t s);  // qkv T [N*128][1152] -> out T [N*128][384]
void launch_attn_enc2(const bf16* qkv, bf16* out, int N, hipStream_t s);                  // bf16, second generation (attn_enc2.hip)
// qkv projection + self-attention fused (qkv_attn.hip): x bf16 [N*128][384] (LayerNorm output), w [1152][384], bias [1152] -> out [N*128][384]
void launch_qkv_attn(const bf16* x, const bf16* w, const float* bias, bf16* out, int N, hipStream_t s);
void set_attn_impl(int v);                                                               // 0: bf16 also uses the first generation
// content token embedding + norm_c.  rows (n, i) for i in [i0,i1): out row n*(i1-i0)+(i-i0)
void launch_dec_embed_ln(Precision prec, int* tokens, const float* emb, const float* pos_q, const float* gamma, const float* beta, float eps,
                         void* out, int N, int i0, int i1, hipStream_t s, const int* skip = nullptr, int skip_n = 0, int planes = 0,
                         const float* prev_logits = nullptr, int prev_ld = 0, int C = 0, int* done_count = nullptr, int eos = 0);   // prev_logits: column i0's token = argmax of these rows first (an AR step)   // planes = 3 (fp32 engines): f16 triple planes out
// self attention of R query rows per crop against the K/V cache [N][26][768].
// mode 0 (AR): R == 1, query index qi0, keys 0..qi0.  mode 1 (refine): R == 26, cloze mask + EOS key padding.
// skip / skip_n: the kernel returns at once when *skip >= skip_n (AR early exit, ConvParams::skip); bf16 per-row kernels only
void launch_dec_self_attn(Precision prec, const float* q /*[26][384] f32*/, const void* kvcache, const int* tokens, void* out, int N, int R, int qi0, int mode, hipStream_t s,
                          const int* skip = nullptr, int skip_n = 0, int planes = 0);
// cross attention of rows [N*R] (Q: T [N*R][384]) against kvmem T [N*128][768]
void launch_dec_cross_attn(Precision prec, const void* q, const void* kvmem, void* out, int N, int R, hipStream_t s, const int* skip = nullptr, int skip_n = 0,
                           const int* done_tok = nullptr, int done_col = 0, int planes = 0);   // done_tok [N][26]: AR steps skip crops with EOS (0) in columns 1 .. done_col
// tokens[n*tok_ld + col] = argmax over C of logits[n*ld ..]
// skip / skip_n: AR early exit; done_count: incremented once per crop whose FIRST EOS (id eos, columns 1 .. col) is the token formed here
void launch_argmax(const float* logits, int ld, int C, int* tokens, int tok_ld, int col, int N, hipStream_t s, const int* skip = nullptr, int skip_n = 0,
                   int* done_count = nullptr, int eos = 0);
void launch_fill_i32(int* p, int value, int n, int stride, hipStream_t s);

// ---- mlp_fused.hip: x_out = x + fc2(GELU(fc1(LayerNorm(x)))) [+ y = LayerNorm_next(x_out)] for the ViT encoder blocks (bf16, E = 384)
struct MlpParams {
  const float* x;          // [M][384] f32 residual stream
  float* x_out;            // [M][384] f32; may be x (every row is read and written by one wave)
  const float *ln_g, *ln_b; float ln_eps;            // the block's norm2
  const bf16* w1p; const float* b1;                  // fc1 as 48 chunk images (pack_mlp_w1), bias [1536]
  const bf16* w2p; const float* b2;                  // fc2 as 48 chunk images (pack_mlp_w2), bias [384]
  const float *nln_g, *nln_b; float nln_eps; bf16* nln_out;   // optional: LayerNorm of x_out -> bf16 [M][384]
  const bf16* att; const bf16* wpp; const float* bp;  // optional: x' = x + att . Wp^T + bp first (att bf16 [M][384], Wp as 12 k-step images, pack_mlp_w2)
  const void* gelu_lut;    // set by the launcher
  int no_x_store;          // 1: x_out is not written (the caller only wants nln_out: the last encoder block - nothing reads its residual stream)
  int store_nt;            // set by the launcher: streaming policy on the epilogue stores
  unsigned long long* dbg; // optional [48][8] shader-clock stamps of workgroup 0 / wave 0 over its first panel (diagnostics)
  int M;
  int stagger;             // set by the launcher: (groups << 16) | microseconds - workgroup (blockIdx / 8) % groups starts that many steps late, to take the
                           // panels' HBM phases (front loads, epilogue stores) of the groups out of lock-step; 0 = all start together
  int ablate;              // timing experiments of the stamps build (results are wrong): 1 no weight DMA after the first items, 2 no GELU, 4 no GEMM2, 8 no GEMM1, 16 no epilogue stores
};
void launch_mlp_fused(const MlpParams& p, hipStream_t s);
void set_mlp_ablate(int v);    // MlpParams::ablate of the stamps build (tools/mlp_stamps.py)
void set_mlp_stagger(int v);   // MlpParams::stagger for the following launches
// host-side packing of the weight operands into the kernel's LDS images (bf16 bits): w1 f32 [1536][384] -> 48 x 24 KiB;
// w f32 [384][K] (fc2: K = 1536, attention projection: K = 384) -> K/32 x 24 KiB
void pack_mlp_w1(const float* w1, uint16_t* out);
void pack_mlp_w2(const float* w, int K, uint16_t* out);

// ---- dec_fused.hip: the whole autoregressive decode (<= 26 steps) of PARSeq as one persistent kernel (bf16)
struct DecArParams {
  const bf16 *w_selfkv, *w_selfout, *w_crossq, *w_crossout, *w_ffn1, *w_ffn2, *w_head;   // [N][K] row-major
  const float *b_selfkv, *b_selfout, *b_crossq, *b_crossout, *b_ffn1, *b_ffn2, *b_head;
  const float *emb, *posq, *qself;                 // [97][384] (pre-scaled), [26][384], [26][384] = Wq.norm_q(pos_q)+bq
  const float *g_c, *b_c, *g_1, *b_1, *g_2, *b_2, *g_f, *b_f;   // norm_c, norm1, norm2, decoder.norm
  const bf16* kvmem;     // [N][128][768] cross-attention K|V of the encoder memory
  bf16* kvcache;         // [N][26][768]  self-attention K|V of the content stream (out: all 26 rows)
  int* tokens;           // [N][26] out: BOS, then the greedy tokens
  float* ar_logits;      // optional [N][26][95]
  const void* gelu_lut;  // gelu_lut_for_current_device()
  unsigned long long* dbg;   // optional [26][16] phase stamps of workgroup 0 (diagnostics), else null
  int N, nsteps;         // nsteps: 25 (logits of the 26th step are never used) or 26
  // Tail form (the kernel-per-op loop ran steps 0 .. first_step-1): the tokens of positions <= first_step-1 are read from `tokens`,
  // token first_step is the argmax of prev_logits (step first_step-1's logits, row stride prev_ld), and the launch returns at once
  // when *skip >= skip_n (every crop of the batch has emitted EOS); a workgroup whose own crops all have is done as well.
  int first_step; const float* prev_logits; int prev_ld; const int* skip; int skip_n;
};
void launch_dec_ar(const DecArParams& p, int crops_per_workgroup, hipStream_t s);

// ---- post_ops.hip
struct CclBuffers {
  // device arrays for a batch of pages, each strided by the page: [pages][npx] unless noted
  float* tnorm; uint8_t* flags; int* parent;
  unsigned* mm;           // [pages][4] ordered-uint min/max of the two maps
  int* area; int* bbox;   // bbox: 4 ints per root (minx,miny,maxx,maxy)
  unsigned* maxt;         // per root max of tnorm (as uint bits; tnorm >= 0)
  int* cand_slot;         // per pixel: slot of the candidate rooted here, or -1
  int* cand;              // [pages][max_cand][8]: root, area, minx, miny, maxx, maxy, row_offset, pad
  int* counters;          // [pages][2]: n_cand, total_rows
  int* rows_packed;       // [pages][npx][2]: per candidate row {min x, max x} of the link-masked pixels
  int max_cand;
  // GPU-side minAreaRect (rects_kernel): per candidate 8 floats {kind, v[0..5], -} (kind as int bits: 0 nothing left, 1 the calipers' raw result, 2 the scratch
  // pool was full - the host's calipers take that group -, 3 a segment, 4 a point; geometry.cpp: finish_min_area_rect turns 1 / 3 / 4 into the RotatedRect with
  // the host's sqrt / atan2); scratch for the hulls: one pool per batch, bump-allocated
  float* rects;           // [pages][max_cand][8]
  float* cal_pool; int* cal_ctr; int cal_cap;   // pool of cal_cap floats, *cal_ctr = floats handed out (zeroed by ccl_init_kernel of page 0)
};
void launch_ccl(const float* heat /*[pages][H][W][2]*/, int pages, int H, int W, float text_threshold, float link_threshold, float low_text, int min_area,
                const CclBuffers& b, hipStream_t s);
// crops: rects5[n] = {x0,y0,x1,y1,page} (clamped, x1/y1 exclusive) of images u8 [pages][h,w,3] (page stride page_bytes)
// -> out u8 [N][32][128][3]; one launch for the crops of every page of a batch
void launch_pack_crops(const uint8_t* images, size_t page_bytes, int stride, const int* rects5, uint8_t* out, int N, hipStream_t s);
// get_detected_boxes' per-component tail on the GPU (tuatara.cpp:162-179: niter, ROI, dilation, findNonZero + minAreaRect): one lane per candidate, geometry.cpp's
// arithmetic step for step (float32 calipers, double where OpenCV is double) -> CclBuffers::rects.  After launch_ccl on the same stream.
void launch_ccl_rects(const CclBuffers& b, int pages, int H, int W, hipStream_t s);

}  // namespace ttr
