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

/* TTR_PREC_F16X4: fp32-equivalent results on the f16 matrix cores - every fp32 activation as three f16 planes, every weight as an
   f16 pair, four MFMAs per product into one fp32 accumulator (tuatara_amd/csrc/split.h).  The reference computes in fp32
   (tuatara.cpp:363-376, :443-446, :307); this mode meets its outputs at the level of fp32 rounding noise and is the default.
   TTR_PREC_BF16: operands rounded to bf16 (fastest; logits differ by up to ~1e-1).  TTR_PREC_F32: fp32 MFMA throughout. */
enum { TTR_PREC_BF16 = 0, TTR_PREC_F32 = 1, TTR_PREC_F16X4 = 2 };
enum { TTR_ORDER_AS_IS = 0 };  /* channel order: the engine reproduces "swap, detect; swap back, recognise"
                                  (tuatara.cpp:349, :441) relative to whatever the caller passes */

/* The constants the reference hard-codes (tuatara.cpp:352-353, :397-399, :148). */
typedef struct ttr_config {
  int precision;         /* TTR_PREC_F16X4 (default), TTR_PREC_BF16 or TTR_PREC_F32 */
  int device;            /* HIP device ordinal */
  int canvas_size;       /* 1024   tuatara.cpp:352 */
  float mag_ratio;       /* 1.0    tuatara.cpp:353 */
  float text_threshold;  /* 0.7    tuatara.cpp:397 */
  float link_threshold;  /* 0.4    tuatara.cpp:398 */
  float low_text;        /* 0.4    tuatara.cpp:399 */
  int min_area;          /* 10     tuatara.cpp:148 */
  int strict_crops;      /* 0: clamp crops to the image; 1: fail like the reference's cv::Exception at :416 */
  int max_components;    /* capacity for CCL candidates per page (default 4096) */
  int verbose;           /* 1: the reference's progress lines on stdout (tuatara.cpp:328-329, :342, :386, :421, :434, :488, :509); TUATARA_VERBOSE=1 does the same */
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
 * GPU always has a whole detector or recogniser pass queued while the host decodes, returns and comes back with the next batch.  Two
 * streams by default (tuning key "recog_overlap"): batch j-1's recogniser runs on a stream of its own beside batch j's detector (they
 * share no buffer); results are identical to the synchronous call's.  The pages of a batch must stay valid until its results have been returned.
 * ttr_stream_flush returns the oldest batch still in flight (*n_prev = 0: none left; call it until then).  out_prev must hold as many
 * entries as the largest batch.  The synchronous calls refuse to run while streamed batches are in flight. */
int ttr_stream_push(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out_prev, int* n_prev);
int ttr_stream_flush(ttr_engine* e, ttr_result** out_prev, int* n_prev);
/* image_to_data over a LIST of host images of any sizes (u8 HWC, 3 channels each; hs[i] x ws[i]; row_strides in bytes, NULL = tightly packed):
 * what a caller of the reference writes as a loop over image_to_data (/root/reference/bindings/run_ocr.py:92, examples/resume.cpp:11), with the models
 * loaded once (the reference reloads both per call, tuatara.cpp:336, :428).  Images of equal size travel together as batches through the streamed path
 * above; their rows are gathered into pinned staging buffers and copied to the device on an upload stream of their own while the previous batch is on the
 * GPU.  out[i] receives image i's result - input order, whatever the batching.  Every result equals what ttr_image_to_data returns for that image.
 * Returns 0; -1 when the call itself could not run (out[] untouched); k > 0 when k images failed - an unreadable entry (NULL, empty, a stride shorter than
 * a row: the reference's "Error reading image from file", tuatara.cpp:344-347) or the images of a batch that failed on the GPU: those keep EMPTY results, every
 * other out[i] is delivered, and ttr_last_error() lists the failed indices with the first failure's message - what a loop over image_to_data gives. */
int ttr_images_to_data(ttr_engine* e, const uint8_t* const* images, const int* hs, const int* ws, const int* row_strides, int n, ttr_result** out);

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

/* ---- multi-GPU: RCCL in the C++ host -----------------------------------------------------------------
 * One process per GPU, one engine per process.  The OCR path has no data-path collective: pages are independent.  The one exchange is
 * the gather of the decoded token ids, and it runs device buffer to device buffer with ncclAllGather (/opt/rocm/include/rccl/rccl.h:678)
 * on the engine's stream: attach a communicator and every batch's ids are gathered behind its recogniser pass.  Records are variable
 * length - every rank's crops-per-page counts travel first, then the payload (the largest rank total rows of 26 ids per rank); nothing
 * is truncated.  There is no torch in this path. */
typedef struct ttr_comm ttr_comm;
#define TTR_COMM_ID_BYTES 256
/* ncclGetUniqueId (x2: a data and a control communicator): call on ONE rank and hand the bytes to the others by any means ... */
int ttr_comm_unique_id(void* id256);
ttr_comm* ttr_comm_create(ttr_engine* e, int rank, int world, const void* id256);
/* ... or let rank 0 listen on addr:port (TCP) and hand them over itself (single node: addr = 127.0.0.1) */
ttr_comm* ttr_comm_create_tcp(ttr_engine* e, int rank, int world, const char* addr, int port);
/* The same communicator over a TCP transport through rank 0 instead of RCCL (device buffers staged through host memory, every collective
 * framed with a sequence number and its size: a mismatched call sequence raises instead of hanging).  For ranks that share ONE GPU - RCCL
 * refuses two ranks on a device -, which is how the multi-rank paths run at world size 2 on a single-GPU box, and as a fallback.
 * Rendezvous (both forms): rank 0 listens on addr:port for TUATARA_COMM_TIMEOUT seconds (default 120). */
ttr_comm* ttr_comm_create_socket(ttr_engine* e, int rank, int world, const char* addr, int port);
const char* ttr_comm_transport(const ttr_comm* c);   /* "rccl" or "socket" */
/* one JSON object about this rank's end of the communicator: {"rank", "world", "transport", "rccl_version" (ncclGetVersion), "device" (HIP ordinal),
 * "pci_bus_id", "gpu" (gcnArchName), "pid"} - what a scaling run prints so that its rank -> GPU map can be read afterwards.  Returns the length. */
int ttr_comm_describe(const ttr_comm* c, char* buf, size_t cap);
void ttr_comm_destroy(ttr_comm* c);
int ttr_comm_rank(const ttr_comm* c);
int ttr_comm_world(const ttr_comm* c);
/* c != NULL: from now on ttr_pages_to_data_dev / ttr_stream_push / ttr_stream_flush all-gather the token ids of every batch (every rank
 * must then make the same sequence of calls with the same page counts: a {status, pages} header travels first, and a rank that failed in
 * its detector or passed another page count makes the call fail on EVERY rank instead of leaving the others in the collective);
 * c == NULL: detach. */
int ttr_engine_attach_comm(ttr_engine* e, ttr_comm* c);
/* The gathered ids of the batch whose results the last such call returned: counts[world][pages] crops per page, ids[sum][26] in
 * (rank, page, crop) order.  Returns the number of id rows (and the sizes through world / pages / ids_need); buffers that are too
 * small or NULL are not written. */
int ttr_last_gathered(ttr_engine* e, int* world, int* pages, int32_t* counts, size_t counts_cap, int32_t* ids, size_t ids_cap, size_t* ids_need);
/* bytes of every rank concatenated by rank into all[world * bytes] (small host buffers; also the barrier of the benchmark) */
int ttr_comm_allgather_host(ttr_comm* c, const void* mine, size_t bytes, void* all);
/* the framing of a gathered batch, host logic only (no GPU): counts[world][pages] -> cap (payload rows per rank), total[world],
 * first[world * pages + 1] (row of each page's first crop in the compacted array) */
int ttr_gather_layout(const int32_t* counts, int world, int pages, int* cap, int32_t* total, int64_t* first);
/* Latency mode (the reference's 6-thread fan-out over the crop batch, tuatara.cpp:450-485, across GPUs): rank 0 passes the pages, detects
 * and packs the crop batch; it is broadcast, rank r recognises shard r of ceil(N / world) crops, the ids are all-gathered and rank 0
 * receives the pages' results (the other ranks pass d_pages = NULL and get n empty results).  Collective.  Returns the result count. */
int ttr_pages_to_data_dev_sharded(ttr_comm* c, const uint8_t* d_pages, int n, int h, int w, ttr_result** out);

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
 * ar_logits (optional) receives the autoregressive pass's logits - per crop defined up to and including its EOS step (upstream leaves
 * its loop when every crop has emitted EOS; behind a crop's own EOS the bf16 engine skips it, and zero-fills the steps behind the batch's
 * exit) -, ids (optional) the argmax ids [n][26]. */
int ttr_parseq_logits(ttr_engine* e, const uint8_t* crops, int n, float* logits, float* ar_logits, int32_t* ids);
/* Tokenizer::decode + EOS cut (tuatara.cpp:61-78, :497-502) on 26 ids; buf needs >= 27 bytes. */
int ttr_decode_ids(const int32_t* ids, int n, char* buf);

/* Per-engine kernel-selection knobs by name (they change fp32 summation order at most, never a rounding point; tests and
 * tools/ use them to compare kernel generations on the same engine): "mlp_fused" (0 off, 1 = from "mlp_min_rows" rows on (default),
 * 2 = always), "mlp_proj", "qkv_attn" (0 / 1 = from "qkv_attn_min" crops on / 2), "dec_mlp_fused" / "dec_mlp_min_rows" (refinement
 * pass through the fused block kernel), "ln_fuse", "tok_fuse" (decoder LayerNorms / token embedding inside the skinny GEMM),
 * "ar_early_exit" (AR steps return at once when every crop has emitted EOS), "decoder_mode" (0 = one kernel per op, 4 / 8 / 16 = the
 * fused persistent AR kernel, else automatic), "fuse_first" (CRAFT conv1_1 inside conv1_2's loader), "enc_chunk", "dbg_bf16_out".
 * Any other key is handed to the process-wide diagnostics setter of tuatara_hip_debug.h.  Returns 0, or -1 for an unknown key. */
int ttr_engine_set_tuning(ttr_engine* e, const char* key, int value);

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
/* the same records by kernel kind, as JSON text written to buf (at most cap - 1 characters + terminator; returns the full length or -1):
 *   [{"kind": "conv3p_kernel<128,NP=3>", "stage": 0, "launches": n, "ms": t, "alg_flops": a, "exec_flops": x}, ...]
 * alg_flops = 2 x MACs of the layers (the figure a roofline is priced with), exec_flops = what the matrix cores execute for them (x 3 / x 4 in
 * the split-operand precision).  With on = 1 a record spans a run of consecutive launches of one kind (inter-kernel gaps included). */
int ttr_get_profile_kinds(ttr_engine* e, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TUATARA_HIP_H */
