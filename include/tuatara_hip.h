/* C ABI of the MI355X-native tuatara engine (libtuatara_hip.so).
 *
 * This is the drop-in boundary for the reference's one hot path,
 *   std::vector<OutputItem> image_to_data(cv::Mat, std::string, std::string)
 *   (/root/reference/tuatara.h:8-13, implemented at tuatara.cpp:314-512),
 * exported as plain C so any host language can bind it (INTEGRATION.md shows the
 * C++ shim that keeps tuatara.h's signature and the pybind11 module `pytuatara`
 * of bindings/python.cpp:43-58).  No exceptions cross this boundary: every call
 * returns 0 on success or a negative code and ttr_last_error() holds the message.
 * Inputs are borrowed for the duration of the call; results are engine-allocated
 * and released with ttr_result_free.
 */
#ifndef TUATARA_HIP_H
#define TUATARA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ttr_engine ttr_engine;
typedef struct ttr_result ttr_result;

enum { TTR_PREC_BF16 = 0, TTR_PREC_F32 = 1 };
enum { TTR_ORDER_AS_IS = 0 };  /* channel order: the engine reproduces "swap, detect; swap back, recognise"
                                  (tuatara.cpp:349, :441) relative to whatever the caller passes */

/* The constants the reference hard-codes (tuatara.cpp:352-353, :397-399, :148). */
typedef struct ttr_config {
  int precision;         /* TTR_PREC_BF16 (throughput) or TTR_PREC_F32 (parity mode: fp32 MFMA) */
  int device;            /* HIP device ordinal */
  int canvas_size;       /* 1024   tuatara.cpp:352 */
  float mag_ratio;       /* 1.0    tuatara.cpp:353 */
  float text_threshold;  /* 0.7    tuatara.cpp:397 */
  float link_threshold;  /* 0.4    tuatara.cpp:398 */
  float low_text;        /* 0.4    tuatara.cpp:399 */
  int min_area;          /* 10     tuatara.cpp:148 */
  int strict_crops;      /* 0: clamp crops to the image; 1: fail like the reference's cv::Exception at :416 */
  int max_components;    /* capacity for CCL candidates per page (default 4096) */
  int verbose;
  int bench_grid_boxes;  /* 0.  Benchmark only (--boxes=grid40): the detector runs in full, then every page's boxes are
                            replaced by a fixed 5 x 8 grid of 150 x 40 px boxes so that the recogniser sees exactly 40 crops per page */
} ttr_config;

void ttr_config_default(ttr_config* cfg);

/* weights_dir holds craft.ttrw + parseq.ttrw (tools/convert_weights.py makes them from the
 * reference's craft_traced_torchscript_model.pt / parseq_torchscript.bin, tuatara.cpp:333,:423). */
ttr_engine* ttr_create(const char* weights_dir, const ttr_config* cfg);
void ttr_destroy(ttr_engine* e);
const char* ttr_last_error(void);
const char* ttr_version(void);

/* ---- the hot path: replaces image_to_data (tuatara.cpp:314-512) ---------------------------- */
/* Host image, u8 HWC 3 channels, row_stride in bytes.  Not modified (the reference swaps
 * the caller's channels in place at :349; callers never rely on that). */
int ttr_image_to_data(ttr_engine* e, const uint8_t* hwc_u8, int h, int w, int row_stride, ttr_result** out);
/* Batch of n same-sized pages already resident in device memory (contiguous [n][h][w][3] u8).
 * out[i] receives page i's result. */
int ttr_pages_to_data_dev(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out);
/* Streamed form of ttr_pages_to_data_dev for a sequence of batches (same contract per batch, results two calls later):
 * ttr_stream_push(j) enqueues the detector of batch j, then the recogniser of batch j-1, turns batch j's components into boxes on the
 * host while the GPU works, and returns batch j-2's results in out_prev[0 .. *n_prev) (*n_prev = 0 on the first two pushes).  The
 * GPU always has a whole detector or recogniser pass queued while the host decodes, returns and comes back with the next batch; one
 * stream, kernels still run one at a time.  The pages of a batch must stay valid until its results have been returned.
 * ttr_stream_flush returns the oldest batch still in flight (*n_prev = 0: none left; call it until then).  out_prev must hold as many
 * entries as the largest batch.  The synchronous calls refuse to run while streamed batches are in flight. */
int ttr_stream_push(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out_prev, int* n_prev);
int ttr_stream_flush(ttr_engine* e, ttr_result** out_prev, int* n_prev);

int ttr_result_count(const ttr_result* r);
const char* ttr_result_text(const ttr_result* r, int i);
const float* ttr_result_bbox(const ttr_result* r, int i);   /* {x1,y1,x2,y2}, tuatara.cpp:272 */
const int32_t* ttr_result_ids(const ttr_result* r, int i);  /* 26 argmax token ids of crop i */
void ttr_result_free(ttr_result* r);
/* bulk views for bindings (valid until ttr_result_free): all boxes [count][4], all ids [count][26]; the texts of all
 * items, each followed by '\n' (no token maps to '\n'), copied into buf when cap suffices; returns the bytes needed. */
const float* ttr_result_bboxes(const ttr_result* r);
const int32_t* ttr_result_ids_all(const ttr_result* r);
int ttr_result_texts(const ttr_result* r, char* buf, size_t cap);
/* the same for a batch of results in one call (any output may be NULL): counts[n], bboxes[total][4], ids[total][26], texts as
 * above ('\n' after each item, copied when texts_cap >= *texts_need).  Returns the total item count. */
int ttr_results_gather(ttr_result* const* rs, int n, int32_t* counts, float* bboxes, int32_t* ids, char* texts, size_t texts_cap,
                       size_t* texts_need);

/* ---- stage-level entry points (BASELINE.json configs 2-3; used by the parity tests) -------- */
/* CRAFT forward (tuatara.cpp:363-394): canvas u8 [H][W][3] (H,W multiples of 32, already resized,
 * padded and in the channel order CRAFT must see) -> heat f32 [H/2][W/2][2]. */
int ttr_craft_heatmap(ttr_engine* e, const uint8_t* canvas, int H, int W, float* heat);
/* get_detected_boxes (tuatara.cpp:119-204): heat f32 [H2][W2][2] -> rects {cx,cy,w,h,angle} in heat-map
 * pixels, ordered by component label (raster order of first pixel).  Returns count in *n. */
int ttr_ccl_boxes(ttr_engine* e, const float* heat, int H2, int W2, float* rects5, int max_rects, int* n);
/* resize_aspect_ratio + channel swap (tuatara.cpp:349, :206-234): image -> canvas u8 [*H][*W][3] */
int ttr_resize_canvas(ttr_engine* e, const uint8_t* hwc_u8, int h, int w, int row_stride, uint8_t* canvas, size_t canvas_cap,
                      int* H, int* W, float* ratio);
/* adjust_result_coordinates + boundingRect + crop + cv::resize (tuatara.cpp:406-418, :436-441):
 * rects in heat-map pixels -> crops u8 [n][32][128][3] in the order PARSeq sees; boxes_out (optional)
 * receives the adjusted rects {cx,cy,w,h,angle} in image pixels. */
int ttr_pack_crops(ttr_engine* e, const uint8_t* hwc_u8, int h, int w, int row_stride, const float* rects5, int n,
                   float ratio, uint8_t* crops, float* boxes_out);
/* PARSeq forward (tuatara.cpp:443-446 + :307): crops u8 [n][32][128][3] -> logits f32 [n][26][95];
 * ar_logits (optional) receives the autoregressive pass's logits, ids (optional) the argmax ids [n][26]. */
int ttr_parseq_logits(ttr_engine* e, const uint8_t* crops, int n, float* logits, float* ar_logits, int32_t* ids);
/* Tokenizer::decode + EOS cut (tuatara.cpp:61-78, :497-502) on 26 ids; buf needs >= 27 bytes. */
int ttr_decode_ids(const int32_t* ids, int n, char* buf);

/* ---- debug / unit-test hook: one implicit-GEMM conv layer on host tensors ------------------- */
/* in f32 NHWC [B][H][W][C0] (+ optional in1 [..][C1] virtual concat), wgt f32 [Cout][ks][ks][C0+C1],
 * out f32 NHWC [B][H][W][Cout].  act: 0 none, 1 relu, 2 gelu. */
int ttr_dbg_conv(ttr_engine* e, const float* in0, int C0, const float* in1, int C1, int relu0, int relu1, int B, int H, int W,
                 int ks, int dil, const float* wgt, const float* bias, int Cout, int act, float* out);

/* bf16 engines: one conv layer (single source) with the 2x2 max-pool fused into its epilogue, as CRAFT's trunk uses it.
 * out_full (optional) f32 [B][H][W][Cout] and out_pool f32 [B][H/2][W/2][Cout] receive the bf16 results widened to f32. */
int ttr_dbg_conv_pool(ttr_engine* e, const float* in0, int C0, int B, int H, int W, int ks, const float* wgt, const float* bias,
                      int Cout, int act, int pool_relu, float* out_full, float* out_pool);
/* Which kernel serves bf16 layers: -1 = first-generation igemm only, 0 = automatic (default),
 * 1..6 = force that gemm2 tile configuration where it applies.  Process-wide; for tuning and tests. */
void ttr_set_gemm_config(int cfg);
/* PARSeq autoregressive loop in bf16 mode: 0 = one kernel per op (the f32 mode's schedule), 4 / 8 / 16 = the
 * fused persistent kernel with that many crops per workgroup, anything else = automatic (default). */
void ttr_set_decoder_mode(int mode);
/* Process-wide tuning knobs by name.  Kernel selection: "gemm_config", "decoder_mode" (as above), "enc_chunk" (crops per PARSeq
 * encoder group, 0 = all at once), "sk_max_rows" / "ws_min_rows" (row counts up to / from which linears use the skinny / the
 * weight-stationary GEMM), "mlp_fused" (0 off, 1 = from "mlp_min_rows" rows on (default), 2 = always), "ln_fuse" (decoder
 * LayerNorms inside the skinny GEMM), "tok_fuse" (AR steps: argmax + token embedding + norm_c inside the self_kv skinny GEMM), "self_refine" (refinement-pass
 * self-attention as one workgroup per crop), "cross_mfma" (refinement-pass cross-attention on the matrix cores), "dec_mlp_fused" / "dec_mlp_min_rows" (refinement pass: cross_out + norm2 + FFN + final norm through the
 * fused block kernel from that many rows on), "fuse_first", "ws_lean", "store_policy" (0 default, 1 streaming, 2 system-scope streaming
 * output stores), "g2_x_ring3", "c3_*" (conv3p variants).  Diagnostics: "dec_stamps" (1 fused decoder, 2 gemm_ws, 3 mlp_fused
 * phase stamps, read back with ttr_dbg_dec_stamps), "dbg_bf16_out", "ws_dbg_flags".
 * Returns 0, or -1 for an unknown key.  Selection knobs change fp32 summation order at most (never a rounding point). */
int ttr_set_tuning(const char* key, int value);
/* host wall-clock splits (microseconds) of the engine's last batch: [0] enqueue resize+CRAFT+CCL, [1] wait for the component
 * counters, [2] wait for candidates / row extremes, [3] calipers, [4] crop rectangles + PARSeq enqueue, [5] wait for the GPU,
 * [6] event read-back, [7] token decode */
void ttr_last_host_us(ttr_engine* e, float out[8]);
/* test hook for the ViT encoder self-attention kernels: qkv f32 [N][128][1152] (rounded to the engine's type) -> out [N][128][384] */
int ttr_dbg_attn_enc(ttr_engine* e, const float* qkv, int N, float* out);
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
/* Times one conv / linear layer on device-generated random data (no host traffic): average
 * microseconds per launch over `iters` back-to-back launches.  f32_resid != 0 selects the PARSeq
 * residual form (f32 residual in, f32 out) instead of a bf16/T output. */
int ttr_bench_conv(ttr_engine* e, int B, int H, int W, int C0, int C1, int ks, int dil, int Cout, int act, int f32_resid,
                   int iters, float* avg_us);

/* ---- host-side geometry hooks (no GPU touched; used by the CPU test-suite) -------------------- */
/* cv::minAreaRect stand-in used at tuatara.cpp:179,:248: n points (x,y) float32 -> {cx,cy,w,h,angle}. */
int ttr_dbg_min_area_rect(const float* xy, int n, float* rect5);
/* tuatara.cpp:162-179 for one component given its stats and per-row x extremes
 * rows[(y1-y0+1)][2] = {min x, max x} ({INT_MAX,-1} = empty row).  Returns 1 if a rect was produced. */
int ttr_dbg_component_rect(int area, int x0, int y0, int x1, int y1, const int32_t* rows, int H, int W, float* rect5);
/* adjust_result_coordinates + boundingRect + format (tuatara.cpp:236-274, :416): rect5 in heat-map
 * pixels -> adjusted rect5, crop xywh (unclamped) and the tesseract bbox. */
int ttr_dbg_box_geometry(const float* rect5, float ratio, float* adjusted5, int32_t* xywh, float* bbox4);

/* ---- device memory helpers (so callers need no HIP bindings) -------------------------------- */
void* ttr_dev_alloc(size_t bytes);
void ttr_dev_free(void* p);
int ttr_dev_upload(void* dst, const void* src, size_t bytes);
int ttr_dev_download(void* dst, const void* src, size_t bytes);
int ttr_dev_sync(ttr_engine* e);
/* last batch: milliseconds spent in each stage on the GPU stream (hipEvents): {craft, post, pack, parseq} */
int ttr_last_stage_ms(ttr_engine* e, float ms[4]);
/* per-launch HIP-event timing of the conv / GEMM kernels, accumulated over calls while on (on = 1: the CRAFT convolution launches
 * only, ~100 events per 32-page step; on = 2: every launch, ~1400 events, which costs ~8 % of throughput):
 * index 0 = CRAFT convolutions, 1 = PARSeq encoder (ViT) and batched decoder GEMMs, 2 = the per-step AR decoder GEMMs.  flops = algorithmic 2*M*N*K of the unpadded layers. */
int ttr_set_profiling(ttr_engine* e, int on);
int ttr_get_profile(ttr_engine* e, double ms[3], double flops[3], long long launches[3]);

#ifdef __cplusplus
}
#endif
#endif /* TUATARA_HIP_H */
