// The C ABI of include/tuatara_hip.h: extern "C", plain pointers and sizes, no exceptions across it.
#include "engine.h"

namespace ttr {

thread_local std::string g_last_error;

}  // namespace ttr

using namespace ttr;

static void run_locked(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out) {
  std::vector<Result> res;
  e->e->run_pages(d_pages, n, h, w, res);
  for (int i = 0; i < n; ++i) { out[i] = new ttr_result(); out[i]->r = std::move(res[i]); }
}

extern "C" {

void ttr_config_default(ttr_config* c) {
  c->precision = TTR_PREC_F16X4; c->device = 0; c->canvas_size = 1024; c->mag_ratio = 1.0f;
  c->text_threshold = 0.7f; c->link_threshold = 0.4f; c->low_text = 0.4f; c->min_area = 10;
  c->strict_crops = 0; c->max_components = 4096; c->verbose = 0;
}

const char* ttr_last_error(void) { return g_last_error.c_str(); }

const char* ttr_version(void) { return "tuatara-mi355x 0.1 (gfx950)"; }

ttr_engine* ttr_create(const char* weights_dir, const ttr_config* cfg) {
  TTR_GUARD_BEGIN
  if (!weights_dir || !*weights_dir) throw std::runtime_error("Please provide a value for weights_dir");  // tuatara.cpp:315-318
  ttr_config c;
  if (cfg) c = *cfg; else ttr_config_default(&c);
  if (c.max_components <= 0) c.max_components = 4096;
  std::unique_ptr<ttr_engine> h(new ttr_engine());
  h->e.reset(new Engine(weights_dir, c));
  return h.release();
  TTR_GUARD_END(nullptr)
}

void ttr_destroy(ttr_engine* e) { delete e; }

int ttr_pages_to_data_dev(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out) {
  TTR_GUARD_BEGIN
  if (!e || !out) throw std::runtime_error("null argument");
  EngineScope lk(*e->e);
  run_locked(e, d_pages, n, h, w, out);
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_stream_push(ttr_engine* e, const uint8_t* d_pages, int n, int h, int w, ttr_result** out_prev, int* n_prev) {
  TTR_GUARD_BEGIN
  if (!e || !out_prev || !n_prev) throw std::runtime_error("null argument");
  EngineScope lk(*e->e);
  std::vector<Result> res;
  int np = 0;
  e->e->stream_push(d_pages, n, h, w, res, np);
  for (int i = 0; i < np; ++i) { out_prev[i] = new ttr_result(); out_prev[i]->r = std::move(res[i]); }
  *n_prev = np;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_stream_flush(ttr_engine* e, ttr_result** out_prev, int* n_prev) {
  TTR_GUARD_BEGIN
  if (!e || !out_prev || !n_prev) throw std::runtime_error("null argument");
  EngineScope lk(*e->e);
  std::vector<Result> res;
  int np = 0;
  e->e->stream_flush(res, np);
  for (int i = 0; i < np; ++i) { out_prev[i] = new ttr_result(); out_prev[i]->r = std::move(res[i]); }
  *n_prev = np;
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_image_to_data(ttr_engine* e, const uint8_t* img, int h, int w, int row_stride, ttr_result** out) {
  TTR_GUARD_BEGIN
  if (!e || !out) throw std::runtime_error("null argument");
  if (!img || h <= 0 || w <= 0) throw std::runtime_error("Error reading image from file");  // tuatara.cpp:344-347
  Engine& E = *e->e;
  EngineScope lk(E);
  E.staging_img.ensure((size_t)h * w * 3);
  TTR_HIP_CHECK(hipMemcpy2DAsync(E.staging_img.p, (size_t)w * 3, img, row_stride, (size_t)w * 3, h, hipMemcpyHostToDevice, E.stream));
  run_locked(e, E.staging_img.as<uint8_t>(), 1, h, w, out);
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_images_to_data(ttr_engine* e, const uint8_t* const* images, const int* hs, const int* ws, const int* row_strides, int n, ttr_result** out) {
  TTR_GUARD_BEGIN
  if (!e || !out || n < 0 || (n > 0 && (!images || !hs || !ws))) throw std::runtime_error("null argument");
  Engine& E = *e->e;
  EngineScope lk(E);
  std::vector<Engine::HostImage> imgs((size_t)n);
  for (int i = 0; i < n; ++i) imgs[i] = Engine::HostImage{images[i], hs[i], ws[i], row_strides ? (std::ptrdiff_t)row_strides[i] : (std::ptrdiff_t)ws[i] * 3};
  std::vector<Result> res;
  std::vector<int> failed;
  std::string first;
  E.run_images(imgs, res, failed, first);
  for (int i = 0; i < n; ++i) { out[i] = new ttr_result(); out[i]->r = std::move(res[i]); }
  if (!failed.empty()) {   // partial failure: every other image's result stands; the failed ones are empty (header)
    std::string msg = std::to_string(failed.size()) + " of " + std::to_string(n) + " images failed (indices";
    for (size_t k = 0; k < failed.size() && k < 16; ++k) msg += " " + std::to_string(failed[k]);
    if (failed.size() > 16) msg += " ...";
    g_last_error = msg + "): " + first;
    return (int)failed.size();
  }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_result_count(const ttr_result* r) { return r ? (int)r->r.text.size() : 0; }

const char* ttr_result_text(const ttr_result* r, int i) { return r->r.text[i].c_str(); }

const float* ttr_result_bbox(const ttr_result* r, int i) { return &r->r.bbox[4 * (size_t)i]; }

const int32_t* ttr_result_ids(const ttr_result* r, int i) { return &r->r.ids[26 * (size_t)i]; }

void ttr_result_free(ttr_result* r) { delete r; }

const float* ttr_result_bboxes(const ttr_result* r) { return r && !r->r.bbox.empty() ? r->r.bbox.data() : nullptr; }

const int32_t* ttr_result_ids_all(const ttr_result* r) { return r && !r->r.ids.empty() ? r->r.ids.data() : nullptr; }

int ttr_results_gather(ttr_result* const* rs, int n, int32_t* counts, float* bboxes, int32_t* ids, char* texts, size_t texts_cap, size_t* texts_need) {
  if (!rs || n < 0) return -1;
  size_t total = 0, need = 0;
  for (int i = 0; i < n; ++i) {
    const size_t c = rs[i] ? rs[i]->r.text.size() : 0;
    if (counts) counts[i] = (int32_t)c;
    total += c;
    if (rs[i]) for (const auto& t : rs[i]->r.text) need += t.size() + 1;
  }
  if (texts_need) *texts_need = need;
  size_t ob = 0, oi = 0, ot = 0;
  for (int i = 0; i < n; ++i) {
    if (!rs[i]) continue;
    const Result& r = rs[i]->r;
    if (bboxes && !r.bbox.empty()) { memcpy(bboxes + ob, r.bbox.data(), r.bbox.size() * 4); ob += r.bbox.size(); }
    if (ids && !r.ids.empty()) { memcpy(ids + oi, r.ids.data(), r.ids.size() * 4); oi += r.ids.size(); }
    if (texts && texts_cap >= need) for (const auto& t : r.text) { memcpy(texts + ot, t.data(), t.size()); ot += t.size(); texts[ot++] = '\n'; }
  }
  return (int)total;
}

int ttr_result_texts(const ttr_result* r, char* buf, size_t cap) {
  if (!r) return 0;
  size_t need = 0;
  for (const auto& t : r->r.text) need += t.size() + 1;
  if (!buf || cap < need) return (int)need;
  size_t o = 0;
  for (const auto& t : r->r.text) { memcpy(buf + o, t.data(), t.size()); o += t.size(); buf[o++] = '\n'; }
  return (int)need;
}

int ttr_craft_heatmap(ttr_engine* e, const uint8_t* canvas, int H, int W, float* heat_out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  E.refuse_while_streaming("ttr_craft_heatmap");
  E.canvas.ensure((size_t)H * W * 3);
  E.heat.ensure((size_t)H * W / 4 * 2 * 4);
  TTR_HIP_CHECK(hipMemcpyAsync(E.canvas.p, canvas, (size_t)H * W * 3, hipMemcpyHostToDevice, E.stream));
  E.craft_forward(E.canvas.as<uint8_t>(), 1, H, W, E.heat.as<float>());
  TTR_HIP_CHECK(hipMemcpyAsync(heat_out, E.heat.p, (size_t)H * W / 4 * 2 * 4, hipMemcpyDeviceToHost, E.stream));
  E.range_fetch(Engine::kRangeStage);
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  E.range_verify(Engine::kRangeStage, "ttr_craft_heatmap");
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_ccl_boxes(ttr_engine* e, const float* heat, int H2, int W2, float* rects5, int max_rects, int* n) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  E.refuse_while_streaming("ttr_ccl_boxes");
  E.heat.ensure((size_t)H2 * W2 * 2 * 4);
  TTR_HIP_CHECK(hipMemcpyAsync(E.heat.p, heat, (size_t)H2 * W2 * 2 * 4, hipMemcpyHostToDevice, E.stream));
  E.ccl_launch(E.heat.as<float>(), 0, 1, 1, 0, H2, W2);
  std::vector<std::vector<RRect>> dets;
  dets.assign(1, std::vector<RRect>());
  E.ccl_collect(0, 1, 0, H2, W2, dets);
  const std::vector<RRect>& det = dets[0];
  *n = (int)det.size();
  for (int i = 0; i < (int)det.size() && i < max_rects; ++i) {
    rects5[5 * i] = det[i].cx; rects5[5 * i + 1] = det[i].cy; rects5[5 * i + 2] = det[i].w; rects5[5 * i + 3] = det[i].h; rects5[5 * i + 4] = det[i].angle;
  }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_resize_canvas(ttr_engine* e, const uint8_t* img, int h, int w, int row_stride, uint8_t* canvas, size_t cap, int* H, int* W, float* ratio) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  E.refuse_while_streaming("ttr_resize_canvas");
  const CanvasGeom g = canvas_geometry(h, w, E.cfg.canvas_size, E.cfg.mag_ratio);
  *H = g.h32; *W = g.w32; *ratio = g.ratio;
  const size_t need = (size_t)g.h32 * g.w32 * 3;
  if (cap < need) throw std::runtime_error("canvas buffer too small");
  E.staging_img.ensure((size_t)h * w * 3);
  E.canvas.ensure(need);
  TTR_HIP_CHECK(hipMemcpy2DAsync(E.staging_img.p, (size_t)w * 3, img, row_stride, (size_t)w * 3, h, hipMemcpyHostToDevice, E.stream));
  launch_resize_pad_u8(E.staging_img.as<uint8_t>(), h, w, w * 3, E.canvas.as<uint8_t>(), g.target_h, g.target_w, g.h32, g.w32, 1, E.stream);
  TTR_HIP_CHECK(hipMemcpyAsync(canvas, E.canvas.p, need, hipMemcpyDeviceToHost, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_pack_crops(ttr_engine* e, const uint8_t* img, int h, int w, int row_stride, const float* rects5, int n, float ratio, uint8_t* crops_out,
                   float* boxes_out) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  E.refuse_while_streaming("ttr_pack_crops");
  if (n <= 0) return 0;
  std::vector<int> rects((size_t)n * 5, 0);
  for (int i = 0; i < n; ++i) {
    RRect r{rects5[5 * i], rects5[5 * i + 1], rects5[5 * i + 2], rects5[5 * i + 3], rects5[5 * i + 4]};
    RRect b = adjust_coordinates(r, 1.f / ratio, 1.f / ratio);
    if (boxes_out) { boxes_out[5 * i] = b.cx; boxes_out[5 * i + 1] = b.cy; boxes_out[5 * i + 2] = b.w; boxes_out[5 * i + 3] = b.h; boxes_out[5 * i + 4] = b.angle; }
    int xywh[4];
    bounding_rect(b, xywh);
    rects[5 * i] = std::max(xywh[0], 0); rects[5 * i + 1] = std::max(xywh[1], 0);
    rects[5 * i + 2] = std::min(xywh[0] + xywh[2], w); rects[5 * i + 3] = std::min(xywh[1] + xywh[3], h);
  }
  E.staging_img.ensure((size_t)h * w * 3);
  E.rects_dev.ensure(rects.size() * 4);
  E.crops.ensure((size_t)n * 32 * 128 * 3);
  TTR_HIP_CHECK(hipMemcpy2DAsync(E.staging_img.p, (size_t)w * 3, img, row_stride, (size_t)w * 3, h, hipMemcpyHostToDevice, E.stream));
  TTR_HIP_CHECK(hipMemcpyAsync(E.rects_dev.p, rects.data(), rects.size() * 4, hipMemcpyHostToDevice, E.stream));
  launch_pack_crops(E.staging_img.as<uint8_t>(), 0, w * 3, E.rects_dev.as<int>(), E.crops.as<uint8_t>(), n, E.stream);
  TTR_HIP_CHECK(hipMemcpyAsync(crops_out, E.crops.p, (size_t)n * 32 * 128 * 3, hipMemcpyDeviceToHost, E.stream));
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_parseq_logits(ttr_engine* e, const uint8_t* crops, int n, float* logits, float* ar_logits, int32_t* ids) {
  TTR_GUARD_BEGIN
  Engine& E = *e->e;
  EngineScope lk(E);
  E.refuse_while_streaming("ttr_parseq_logits");
  if (n <= 0) return 0;
  E.crops.ensure((size_t)n * 32 * 128 * 3);
  E.logits.ensure((size_t)n * 26 * 95 * 4);
  E.ids_dev.ensure((size_t)n * 26 * 4);
  if (ar_logits) E.ar_logits.ensure((size_t)n * 26 * 95 * 4);
  TTR_HIP_CHECK(hipMemcpyAsync(E.crops.p, crops, (size_t)n * 32 * 128 * 3, hipMemcpyHostToDevice, E.stream));
  E.parseq_forward(E.crops.as<uint8_t>(), n, E.logits.as<float>(), ar_logits ? E.ar_logits.as<float>() : nullptr, E.ids_dev.as<int>());
  TTR_HIP_CHECK(hipMemcpyAsync(logits, E.logits.p, (size_t)n * 26 * 95 * 4, hipMemcpyDeviceToHost, E.stream));
  if (ar_logits) TTR_HIP_CHECK(hipMemcpyAsync(ar_logits, E.ar_logits.p, (size_t)n * 26 * 95 * 4, hipMemcpyDeviceToHost, E.stream));
  if (ids) TTR_HIP_CHECK(hipMemcpyAsync(ids, E.ids_dev.p, (size_t)n * 26 * 4, hipMemcpyDeviceToHost, E.stream));
  E.range_fetch(Engine::kRangeStage);
  TTR_HIP_CHECK(hipStreamSynchronize(E.stream));
  E.range_verify(Engine::kRangeStage, "ttr_parseq_logits");
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_decode_ids(const int32_t* ids, int n, char* buf) {
  TTR_GUARD_BEGIN
  static const Tokenizer tok;
  std::string s = tok.decode(ids, n);
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
  TTR_GUARD_END(-1)
}

void* ttr_dev_alloc(size_t bytes) { void* p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }

void ttr_dev_free(void* p) { if (p) (void)hipFree(p); }

int ttr_dev_upload(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }

int ttr_dev_download(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }

int ttr_dev_sync(ttr_engine* e) {   // every stream the engine enqueues work on (the recogniser of a streamed batch and the second detector lane have their own)
  if (!e) return -1;
  bool ok = hipStreamSynchronize(e->e->stream) == hipSuccess;
  ok = (hipStreamSynchronize(e->e->recog_stream) == hipSuccess) && ok;
  ok = (hipStreamSynchronize(e->e->lane_stream) == hipSuccess) && ok;
  return ok ? 0 : -1;
}

int ttr_set_profiling(ttr_engine* e, int on) {
  TTR_GUARD_BEGIN
  if (!e) throw std::runtime_error("null argument");
  Engine& E = *e->e;
  EngineScope lk(E);
  E.profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
  E.prof_recs.clear();
  for (int i = 0; i < 3; ++i) { E.prof_ms[i] = 0; E.prof_flops[i] = 0; E.prof_launches[i] = 0; }
  for (auto& k : E.prof_kinds) { k.ms = 0; k.alg = 0; k.exec = 0; k.launches = 0; k.bytes = 0; }
  return 0;
  TTR_GUARD_END(-1)
}

// The same records by kernel kind, as JSON text: [{"kind": name, "stage": 0|1|2, "launches": n, "ms": t, "alg_flops": a, "exec_flops": x, "alg_bytes": b}, ...]
// (alg_bytes: the detector layers' algorithmic HBM bytes - every operand read once, every result written once, at the engine's plane sizes; 0 where not tallied)
// (alg_flops: 2 x MACs of the layers, SURVEY.md section 8(d)'s figure; exec_flops: what the matrix cores execute for them).  Returns the
// text's length (without the terminator); the text is truncated to cap - 1 characters.
int ttr_get_profile_kinds(ttr_engine* e, char* buf, size_t cap) {
  TTR_GUARD_BEGIN
  if (!e) throw std::runtime_error("null argument");
  Engine& E = *e->e;
  EngineScope lk(E);
  E.prof_collect();
  std::string s = "[";
  bool first = true;
  for (const auto& k : E.prof_kinds) {
    if (!k.launches) continue;
    std::string name;            // JSON string: quotes, backslashes and control characters escaped
    for (const char ch : k.name) {
      if (ch == '"' || ch == '\\') { name += '\\'; name += ch; }
      else if ((unsigned char)ch < 0x20) { char u[8]; snprintf(u, sizeof u, "\\u%04x", (unsigned)(unsigned char)ch); name += u; }
      else name += ch;
    }
    char line[320];
    snprintf(line, sizeof line, "\", \"stage\": %d, \"launches\": %lld, \"ms\": %.6f, \"alg_flops\": %.6e, \"exec_flops\": %.6e, \"alg_bytes\": %.6e}", k.stage, (long long)k.launches, k.ms, k.alg, k.exec, k.bytes);
    s += first ? "{\"kind\": \"" : ", {\"kind\": \"";
    s += name; s += line; first = false;
  }
  s += "]";
  if (buf && cap) { const size_t n = std::min(s.size(), cap - 1); memcpy(buf, s.data(), n); buf[n] = 0; }
  return (int)s.size();
  TTR_GUARD_END(-1)
}

int ttr_get_profile(ttr_engine* e, double ms[3], double flops[3], long long launches[3]) {
  TTR_GUARD_BEGIN
  if (!e) throw std::runtime_error("null argument");
  Engine& E = *e->e;
  EngineScope lk(E);
  E.prof_collect();          // records whose events completed since the last batch was finished
  for (int i = 0; i < 3; ++i) { ms[i] = E.prof_ms[i]; flops[i] = E.prof_flops[i]; launches[i] = E.prof_launches[i]; }
  return 0;
  TTR_GUARD_END(-1)
}

int ttr_last_stage_ms(ttr_engine* e, float ms[4]) { memcpy(ms, e->e->stage_ms, sizeof(float) * 4); return 0; }

}  // extern "C"
