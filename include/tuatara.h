// C++ drop-in for /root/reference/tuatara.h:8-13 on top of the C ABI (tuatara_hip.h).
//
//   struct OutputItem { std::string text; std::vector<float> bbox; };
//   std::vector<OutputItem> image_to_data(cv::Mat image, std::string weights_dir, std::string outputs_dir);
//
// The cv::Mat overload exists only when OpenCV headers are present (the reference's header
// includes them unconditionally, tuatara.h:3-4); the raw-pointer overload is always there and
// is what bindings/python.cpp uses.  Error convention as the reference: message on std::cerr
// and an empty vector (tuatara.cpp:315-323, :337-340, :344-347).  The engine is created on the
// first call for a weights_dir and cached (the reference reloads both models per call).
#ifndef TUATARA_H
#define TUATARA_H
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct OutputItem {
  std::string text;
  std::vector<float> bbox;  // x1, y1, x2, y2
};

// image: u8 HWC, 3 channels, `row_stride` bytes per row (0 = tightly packed).  Not modified.
std::vector<OutputItem> image_to_data(const uint8_t* image, int rows, int cols, std::ptrdiff_t row_stride, std::string weights_dir,
                                      std::string outputs_dir);

// The same over a list of images of any sizes: one entry of the result per image, in input order (what a caller of the reference writes as a loop over
// image_to_data; here the list shares one engine, same-sized images travel as batches and the host-to-device copies run beside the GPU's work).
// An error (see above) prints its message and returns an empty list.
struct ImageView { const uint8_t* data; int rows, cols; std::ptrdiff_t row_stride; };   // row_stride 0 = tightly packed
std::vector<std::vector<OutputItem>> images_to_data(const std::vector<ImageView>& images, std::string weights_dir, std::string outputs_dir);

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
inline std::vector<OutputItem> image_to_data(cv::Mat image, std::string weights_dir, std::string outputs_dir) {
  if (image.empty() || image.type() != CV_8UC3) return image_to_data(nullptr, 0, 0, 0, weights_dir, outputs_dir);
  return image_to_data(image.data, image.rows, image.cols, (std::ptrdiff_t)image.step, weights_dir, outputs_dir);
}
inline std::vector<std::vector<OutputItem>> images_to_data(const std::vector<cv::Mat>& images, std::string weights_dir, std::string outputs_dir) {
  std::vector<ImageView> v;
  for (const cv::Mat& m : images) v.push_back(m.empty() || m.type() != CV_8UC3 ? ImageView{nullptr, 0, 0, 0} : ImageView{m.data, m.rows, m.cols, (std::ptrdiff_t)m.step});
  return images_to_data(v, weights_dir, outputs_dir);
}
#endif
#endif

#endif  // TUATARA_H
