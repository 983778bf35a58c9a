/* Developer diagnostics of the MI355X tuatara engine: unit-test hooks for single kernels, micro-benchmarks, phase stamps and the
 * PROCESS-WIDE kernel-variant switches of the .hip files.  Not part of the drop-in surface (include/tuatara_hip.h); exported by
 * the same library so that tests/ and tools/ can reach the kernels without the whole pipeline. */
#ifndef TUATARA_HIP_DEBUG_H
#define TUATARA_HIP_DEBUG_H

#include "tuatara_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- debug / unit-test hook: one implicit-GEMM conv layer on host tensors ------------------- */
/* in f32 NHWC [B][H][W][C0] (+ optional in1 [..][C1] virtual concat), wgt f32 [Cout][ks][ks][C0+C1],
 * out f32 NHWC [B][H][W][Cout].  act: 0 none, 1 relu, 2 gelu. */
int ttr_dbg_conv(ttr_engine* e, const float* in0, int C0, const float* in1, int C1, int relu0, int relu1, int B, int H, int W,
                 int ks, int dil, const float* wgt, const float* bias, int Cout, int act, float* out);

/* bf16 engines: one conv layer (single source) with the 2x2 max-pool fused into its epilogue, as CRAFT's trunk uses it.
 * out_full (optional) f32 [B][H][W][Cout] and out_pool f32 [B][H/2][W/2][Cout] receive the bf16 results widened to f32. */
/* One split-operand linear layer of the f16x4 engine on its own: out[M][N] = act(x w^T + bias (+ resid)), np = 3 (activation pairs) / 4 (triples),
 * out_planes = 0 (fp32 from the kernel) / 2 / 3 (f16 planes from the kernel, joined on the host), cfg = gemm2's tile configuration (0 = automatic). */
int ttr_dbg_split_gemm(ttr_engine* e, const float* x, int M, int K, const float* w, const float* bias, int N, int np, int act, int out_planes,
                       const float* resid, int cfg, float* out);
int ttr_dbg_conv_pool(ttr_engine* e, const float* in0, int C0, int B, int H, int W, int ks, const float* wgt, const float* bias,
                      int Cout, int act, int pool_relu, float* out_full, float* out_pool);
/* Which kernel serves bf16 layers: -1 = first-generation igemm only, 0 = automatic (default),
 * 1..6 = force that gemm2 tile configuration where it applies.  Process-wide; for tuning and tests. */
void ttr_set_gemm_config(int cfg);
/* PARSeq autoregressive loop in bf16 mode: 0 = one kernel per op (the f32 mode's schedule), 4 / 8 / 16 = the
 * fused persistent kernel with that many crops per workgroup, anything else = automatic (default). */
void ttr_set_decoder_mode(int mode);
/* Process-wide knobs by name: the defaults of engines created afterwards for the per-engine keys of ttr_engine_set_tuning, and the
 * kernel-variant switches that live in the kernel files.  Kernel selection: "gemm_config", "decoder_mode" (as above), "enc_chunk" (crops per PARSeq
 * encoder group, 0 = all at once), "sk_max_rows" / "ws_min_rows" (row counts up to / from which linears use the skinny / the
 * weight-stationary GEMM), "mlp_fused" (0 off, 1 = from "mlp_min_rows" rows on (default), 2 = always), "ln_fuse" (decoder
 * LayerNorms inside the skinny GEMM), "tok_fuse" (AR steps: argmax + token embedding + norm_c inside the self_kv skinny GEMM), "self_refine" (refinement-pass
 * self-attention as one workgroup per crop), "cross_mfma" (refinement-pass cross-attention on the matrix cores), "dec_mlp_fused" / "dec_mlp_min_rows" (refinement pass: cross_out + norm2 + FFN + final norm through the
 * fused block kernel from that many rows on; 1 = when its 128-row panels fill the CUs' last round to 65 %, 2 = always), "fuse_first", "ws_lean", "store_policy" (0 default, 1 streaming, 2 system-scope streaming
 * output stores), "g2_x_ring3", "c3_*" (conv3p variants: "c3_c32" 32-wide tiles for Cout <= 32, "c3_narrow64" 64-wide tiles on the
 * 16x16-patch maps: 0 never / 1 always / 2 when the chip would be under-filled), "c3s_wgs" (persistent conv3s workgroups per CU),
 * "upsample_block" (2x4-block bilinear kernel), "craft_group" (pages per CRAFT launch group), "ar_early_exit" / "ar_crop_exit" /
 * "ar_tail_step" (AR loop: batch-level exit, per-crop exit in the step's attention kernels, step from which the fused tail kernel
 * takes over), "mlp_store_nt".  Timing experiments (results unspecified): "mlp_stagger",
 * "mlp_ablate" / "pair_ablate" (libraries built with -DMLP_ABLATE_BUILDS).  Diagnostics: "dec_stamps" (1 fused decoder, 2 gemm_ws, 3 mlp_fused
 * phase stamps, read back with ttr_dbg_dec_stamps), "dbg_bf16_out", "ws_dbg_flags".
 * Returns 0, or -1 for an unknown key.  Selection knobs change fp32 summation order at most (never a rounding point). */
int ttr_set_tuning(const char* key, int value);
/* host wall-clock splits (microseconds) of the engine's last batch: [0] enqueue resize+CRAFT+CCL, [1] wait for the component
 * counters, [2] wait for candidates / row extremes, [3] calipers, [4] crop rectangles + PARSeq enqueue, [5] wait for the GPU,
 * [6] event read-back, [7] token decode */
void ttr_last_host_us(ttr_engine* e, float out[8]);
/* test hook for the ViT encoder self-attention kernels: qkv f32 [N][128][1152] (rounded to the engine's type) -> out [N][128][384] */
int ttr_dbg_attn_enc(ttr_engine* e, const float* qkv, int N, float* out);
/* test hook for the decoder's cross-attention kernels of the split-operand / fp32 engines (parseq_ops.hip, attn_cross_split.hip; which one runs follows the
 * tuning keys "cross_split", "cross_crop", "cross_rows_hsplit"): q f32 [N * R][384] (R query rows per crop), kvmem f32 [N * 128][768] (K | V of a crop's 128
 * memory tokens) -> out f32 [N * R][384], joined from the exact triples the kernels write */
int ttr_dbg_cross_attn(ttr_engine* e, const float* q, const float* kvmem, int N, int R, float* out);
/* test hook for qkv_attn.hip (bf16 engines): x f32 [N][128][384], w [1152][384], b [1152] -> self-attention output [N][128][384] */
int ttr_dbg_qkv_attn(ttr_engine* e, const float* x, int N, const float* w, const float* b, float* out);
/* test hook for mlp_fused.hip (bf16 engines): x_out = x + fc2(GELU(fc1(LayerNorm(x)))) over f32 rows [M][384] with weights
 * w1 [1536][384], w2 [384][1536] (rounded to bf16 inside); nln_out (may be NULL) = LayerNorm(x_out; nln_g, nln_b), bf16 values as f32.
 * With att != NULL the attention output projection runs first in the same launch: x is replaced by x + att . wp^T + bp
 * (att f32 [M][384] and wp [384][384] rounded to bf16). */
int ttr_dbg_mlp(ttr_engine* e, const float* x, int M, const float* ln_g, const float* ln_b, float eps, const float* w1, const float* b1, const float* w2,
                const float* b2, const float* nln_g, const float* nln_b, float* x_out, float* nln_out, const float* att, const float* wp, const float* bp);
/* diagnostics: after ttr_set_tuning("dec_stamps", 1 / 2 / 3) workgroup 0 of the fused AR kernel / gemm_ws / mlp_fused records
 * shader-clock stamps into a 416-entry buffer ([26 steps][16 phases], [2 waves][24 panels][8], [48 chunks][8]); this copies them out.
 * Returns -1 when stamps are off. */
int ttr_dbg_dec_stamps(unsigned long long* out);
/* the first n (<= 4096) words of the same buffer: qkv_attn4.hip also records where and when each workgroup ran (words 512 + 4 b ..: start, end in 100 MHz ticks, HW_ID | XCC_ID << 32, shader clocks) */
int ttr_dbg_dec_stamps_ext(unsigned long long* out, int n);
/* Times one conv / linear layer on device-generated random data (no host traffic): average
 * microseconds per launch over `iters` back-to-back launches.  f32_resid != 0 selects the PARSeq
 * residual form (f32 residual in, f32 out) instead of a bf16/T output. */
int ttr_bench_conv(ttr_engine* e, int B, int H, int W, int C0, int C1, int ks, int dil, int Cout, int act, int f32_resid,
                   int iters, float* avg_us);

/* ---- host-side geometry hooks (no GPU touched; used by the CPU test-suite) -------------------- */
/* cv::minAreaRect stand-in used at tuatara.cpp:179,:248: n points (x,y) float32 -> {cx,cy,w,h,angle}. */
int ttr_dbg_min_area_rect(const float* xy, int n, float* rect5);
/* the TCP rendezvous of ttr_comm_create_tcp on its own (no GPU, no RCCL): rank 0 listens on addr:port and hands its `bytes` bytes to the
 * world - 1 ranks that connect.  Returns 0, or -1 (ttr_last_error).  tests/test_comm_cpu.py runs it with two processes. */
int ttr_dbg_tcp_share(int rank, int world, const char* addr, int port, void* buf, size_t bytes);
/* tuatara.cpp:162-179 for one component given its stats and per-row x extremes
 * rows[(y1-y0+1)][2] = {min x, max x} ({INT_MAX,-1} = empty row).  Returns 1 if a rect was produced. */
int ttr_dbg_component_rect(int area, int x0, int y0, int x1, int y1, const int32_t* rows, int H, int W, float* rect5);
/* adjust_result_coordinates + boundingRect + format (tuatara.cpp:236-274, :416): rect5 in heat-map
 * pixels -> adjusted rect5, crop xywh (unclamped) and the tesseract bbox. */
int ttr_dbg_box_geometry(const float* rect5, float ratio, float* adjusted5, int32_t* xywh, float* bbox4);


#ifdef __cplusplus
}
#endif
#endif
