// Host-side geometry and text decoding of the engine (no GPU, no OpenCV).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ttr {

// cv::RotatedRect stand-in: centre, size, angle in degrees, all float32.
struct RRect { float cx = 0, cy = 0, w = 0, h = 0, angle = 0; };
struct Pt2f { float x, y; };

void rect_points(const RRect& r, Pt2f pt[4]);                 // cv::RotatedRect::points      (tuatara.cpp:181,:241,:258)
void bounding_rect(const RRect& r, int xywh[4]);              // cv::RotatedRect::boundingRect (tuatara.cpp:416)
void tesseract_bbox(const RRect& r, float bbox[4]);           // rotated_rect_to_tesseract_format (tuatara.cpp:256-274)
RRect min_area_rect(const Pt2f* pts, int n);                  // cv::minAreaRect              (tuatara.cpp:179,:248)
// the tail of cv::minAreaRect behind the hull: kind 1 = the rotating calipers' raw result out[6] (corner + two side vectors), 3 = a two-point hull
// (x0, y0, x1, y1), 4 = one point -> centre, sides (double sqrt), angle (double atan2, degrees).  min_area_rect ends in it; the GPU calipers
// (post_ops.hip: ccl_rects_kernel) hand their raw result to it, so that sqrt / atan2 are the host's libm on both paths
RRect finish_min_area_rect(int kind, const float v[6]);
RRect adjust_coordinates(const RRect& r, float ratio_w, float ratio_h, float ratio_net = 2.f);  // tuatara.cpp:236-253

// One CCL candidate as the GPU reports it (post_ops.hip): stats of the combined-map
// component and the per-row x extremes of its link-masked pixels.
struct Component {
  int root, area, x0, y0, x1, y1;       // bbox inclusive
  const int* rows;                      // [(y1-y0+1)][2] = {min x, max x}; {INT_MAX,-1} for an empty row
};
// tuatara.cpp:162-179 on the row extremes: niter (integer arithmetic), ROI, rectangular
// dilation with OpenCV's anchor, findNonZero + minAreaRect.  Returns false if nothing is left.
bool component_to_rect(const Component& c, int H, int W, RRect* out);

// resize_aspect_ratio's integer/float bookkeeping (tuatara.cpp:206-234)
struct CanvasGeom { int target_h, target_w, h32, w32; float ratio; };
CanvasGeom canvas_geometry(int height, int width, int square_size, float mag_ratio);

// Tokenizer (tuatara.cpp:25-117) with the reference's id table quirks.
struct Tokenizer {
  std::string itos;          // 98 entries
  int eos_id, bos_id, pad_id;  // 88, 96, 97
  Tokenizer();
  // argmax ids of one row -> filter(eos_id) -> ids2tok -> cut at first EOS char (tuatara.cpp:108-116, :93-99, :497-502)
  std::string decode(const int* ids, int n) const;
};

}  // namespace ttr
