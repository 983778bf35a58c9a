// image_to_data (tuatara.h:13 / tuatara.cpp:314-512) as a thin C++ shim over the C ABI.
#include "../../include/tuatara.h"

#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>

#include "../../include/tuatara_hip.h"

namespace {
std::mutex g_mu;
std::map<std::string, ttr_engine*> g_engines;  // one engine per (weights_dir, precision); lives for the process

ttr_engine* engine_for(const std::string& weights_dir) {
  std::lock_guard<std::mutex> lk(g_mu);
  ttr_config cfg;
  ttr_config_default(&cfg);
  if (const char* p = std::getenv("TUATARA_PRECISION")) {   // default: TTR_PREC_F16X4 (fp32-equivalent, the reference computes in fp32)
    const std::string v(p);
    cfg.precision = v == "f32" ? TTR_PREC_F32 : v == "bf16" ? TTR_PREC_BF16 : TTR_PREC_F16X4;
  }
  if (const char* p = std::getenv("TUATARA_STRICT_CROPS")) cfg.strict_crops = std::atoi(p);
  if (const char* p = std::getenv("TUATARA_DEVICE")) cfg.device = std::atoi(p);
  std::string key = weights_dir + "#" + std::to_string(cfg.precision) + "#" + std::to_string(cfg.device);
  auto it = g_engines.find(key);
  if (it != g_engines.end()) return it->second;
  ttr_engine* e = ttr_create(weights_dir.c_str(), &cfg);
  if (e) g_engines[key] = e;
  return e;
}
}  // namespace

std::vector<OutputItem> image_to_data(const uint8_t* image, int rows, int cols, std::ptrdiff_t row_stride, std::string weights_dir,
                                      std::string outputs_dir) {
  if (weights_dir.empty()) {  // tuatara.cpp:315-318
    std::cerr << "Please provide a value for weights_dir" << std::endl;
    return {};
  }
  if (outputs_dir.empty()) {  // tuatara.cpp:320-323 (never used afterwards, there or here)
    std::cerr << "Please provide a value for outputs_dir" << std::endl;
    return {};
  }
  ttr_engine* e = engine_for(weights_dir);
  if (!e) {  // tuatara.cpp:337-340, :429-432
    std::cerr << "error loading craft/parseq model: " << ttr_last_error() << std::endl;
    return {};
  }
  if (!image || rows <= 0 || cols <= 0) {  // tuatara.cpp:344-347
    std::cerr << "Error reading image from file";
    return {};
  }
  ttr_result* r = nullptr;
  if (ttr_image_to_data(e, image, rows, cols, row_stride ? (int)row_stride : cols * 3, &r) != 0) {
    std::cerr << "tuatara: " << ttr_last_error() << std::endl;
    return {};
  }
  std::vector<OutputItem> out(ttr_result_count(r));
  for (size_t i = 0; i < out.size(); ++i) {
    out[i].text = ttr_result_text(r, (int)i);
    const float* b = ttr_result_bbox(r, (int)i);
    out[i].bbox.assign(b, b + 4);
  }
  ttr_result_free(r);
  return out;
}

std::vector<std::vector<OutputItem>> images_to_data(const std::vector<ImageView>& images, std::string weights_dir, std::string outputs_dir) {
  if (weights_dir.empty()) { std::cerr << "Please provide a value for weights_dir" << std::endl; return {}; }   // tuatara.cpp:315-318
  if (outputs_dir.empty()) { std::cerr << "Please provide a value for outputs_dir" << std::endl; return {}; }   // tuatara.cpp:320-323
  ttr_engine* e = engine_for(weights_dir);
  if (!e) { std::cerr << "error loading craft/parseq model: " << ttr_last_error() << std::endl; return {}; }     // tuatara.cpp:337-340, :429-432
  const int n = (int)images.size();
  std::vector<const uint8_t*> ptr(n);
  std::vector<int> hs(n), ws(n), st(n);
  for (int i = 0; i < n; ++i) {
    // (an unreadable entry - tuatara.cpp:344-347 - yields an empty list for that image alone: the engine prints the reference's message)
    ptr[i] = images[i].data; hs[i] = images[i].rows; ws[i] = images[i].cols;
    st[i] = images[i].row_stride ? (int)images[i].row_stride : images[i].cols * 3;
  }
  std::vector<ttr_result*> rs(n, nullptr);
  const int rc = n ? ttr_images_to_data(e, ptr.data(), hs.data(), ws.data(), st.data(), n, rs.data()) : 0;
  if (rc < 0) {   // the call could not run at all
    std::cerr << "tuatara: " << ttr_last_error() << std::endl;
    return {};
  }
  if (rc > 0) std::cerr << "tuatara: " << ttr_last_error() << std::endl;   // some images failed: theirs stay empty, the rest are returned (a loop over image_to_data)
  std::vector<std::vector<OutputItem>> out(n);
  for (int i = 0; i < n; ++i) {
    out[i].resize(ttr_result_count(rs[i]));
    for (size_t k = 0; k < out[i].size(); ++k) {
      out[i][k].text = ttr_result_text(rs[i], (int)k);
      const float* b = ttr_result_bbox(rs[i], (int)k);
      out[i][k].bbox.assign(b, b + 4);
    }
    ttr_result_free(rs[i]);
  }
  return out;
}
