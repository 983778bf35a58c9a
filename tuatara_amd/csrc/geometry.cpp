// Host-side geometry and text decoding.  Replaces the OpenCV calls the reference makes
// after connected-component labelling (tuatara.cpp:162-179, :236-274, :416) and its
// Tokenizer (tuatara.cpp:25-117).  float32 throughout where OpenCV is float32.
#include "geometry.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cfloat>
#include <map>

namespace ttr {

static const double kPi = 3.1415926535897932384626433832795;  // CV_PI

static inline int cv_floor(double v) { int i = (int)v; return i - (v < i); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (v > i); }

void rect_points(const RRect& r, Pt2f pt[4]) {
  double ang = r.angle * kPi / 180.;
  float b = (float)std::cos(ang) * 0.5f;
  float a = (float)std::sin(ang) * 0.5f;
  pt[0].x = r.cx - a * r.h - b * r.w;
  pt[0].y = r.cy + b * r.h - a * r.w;
  pt[1].x = r.cx + a * r.h - b * r.w;
  pt[1].y = r.cy - b * r.h - a * r.w;
  pt[2].x = 2 * r.cx - pt[0].x;
  pt[2].y = 2 * r.cy - pt[0].y;
  pt[3].x = 2 * r.cx - pt[1].x;
  pt[3].y = 2 * r.cy - pt[1].y;
}

void bounding_rect(const RRect& r, int xywh[4]) {
  Pt2f p[4];
  rect_points(r, p);
  float mnx = std::min(std::min(p[0].x, p[1].x), std::min(p[2].x, p[3].x));
  float mny = std::min(std::min(p[0].y, p[1].y), std::min(p[2].y, p[3].y));
  float mxx = std::max(std::max(p[0].x, p[1].x), std::max(p[2].x, p[3].x));
  float mxy = std::max(std::max(p[0].y, p[1].y), std::max(p[2].y, p[3].y));
  xywh[0] = cv_floor(mnx);
  xywh[1] = cv_floor(mny);
  xywh[2] = cv_ceil(mxx) - xywh[0] + 1;
  xywh[3] = cv_ceil(mxy) - xywh[1] + 1;
}

void tesseract_bbox(const RRect& r, float bbox[4]) {
  Pt2f v[4];
  rect_points(r, v);
  float min_x = std::min(std::min(v[0].x, v[1].x), std::min(v[2].x, v[3].x));
  float min_y = std::min(std::min(v[0].y, v[1].y), std::min(v[2].y, v[3].y));
  float max_x = std::max(std::max(v[0].x, v[1].x), std::max(v[2].x, v[3].x));
  float max_y = std::max(std::max(v[0].y, v[1].y), std::max(v[2].y, v[3].y));
  bbox[0] = std::round(min_x); bbox[1] = std::round(min_y); bbox[2] = std::round(max_x); bbox[3] = std::round(max_y);
}

// ---------------------------------------------------------------- convex hull (monotone chain, exact on integer-valued input)
static double cross(const Pt2f& o, const Pt2f& a, const Pt2f& b) {
  return ((double)a.x - o.x) * ((double)b.y - o.y) - ((double)a.y - o.y) * ((double)b.x - o.x);
}

static std::vector<Pt2f> convex_hull(std::vector<Pt2f> p) {
  std::sort(p.begin(), p.end(), [](const Pt2f& a, const Pt2f& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
  p.erase(std::unique(p.begin(), p.end(), [](const Pt2f& a, const Pt2f& b) { return a.x == b.x && a.y == b.y; }), p.end());
  const int n = (int)p.size();
  if (n < 3) return p;
  std::vector<Pt2f> h(2 * n);
  int k = 0;
  for (int i = 0; i < n; ++i) { while (k >= 2 && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
  for (int i = n - 2, t = k + 1; i >= 0; --i) { while (k >= t && cross(h[k - 2], h[k - 1], p[i]) <= 0) --k; h[k++] = p[i]; }
  h.resize(k - 1);
  return h;
}

// ---------------------------------------------------------------- rotating calipers (float32, minimum-area mode)
// The four calipers sides are (a,b), (-b,a), (-a,-b), (b,-a); at every step the side
// making the smallest angle with its polygon edge becomes flush with it.  out = corner,
// edge vector 1, edge vector 2.
static void rotating_calipers_min_area(const Pt2f* points, int n, float out[6]) {
  float minarea = FLT_MAX;
  std::vector<float> inv_len(n);
  std::vector<Pt2f> vect(n);
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4];
  float orientation = 0.f, base_a, base_b = 0.f;
  Pt2f pt0 = points[0];
  float left_x = pt0.x, right_x = pt0.x, top_y = pt0.y, bottom_y = pt0.y;
  for (int i = 0; i < n; ++i) {
    if (pt0.x < left_x) left_x = pt0.x, left = i;
    if (pt0.x > right_x) right_x = pt0.x, right = i;
    if (pt0.y > top_y) top_y = pt0.y, top = i;
    if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
    Pt2f pt = points[i + 1 < n ? i + 1 : 0];
    double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
    vect[i].x = (float)dx; vect[i].y = (float)dy;
    inv_len[i] = (float)(1. / std::sqrt(dx * dx + dy * dy));
    pt0 = pt;
  }
  {
    double ax = vect[n - 1].x, ay = vect[n - 1].y;
    for (int i = 0; i < n; ++i) {
      double bx = vect[i].x, by = vect[i].y;
      double convexity = ax * by - ay * bx;
      if (convexity != 0) { orientation = convexity > 0 ? 1.f : -1.f; break; }
      ax = bx; ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int best_left = 0, best_bottom = 0;
  float best_a = 1.f, best_b = 0.f, best_w = 0.f, best_h = 0.f;
  for (int k = 0; k < n; ++k) {
    float dp[4] = {
        +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y,
        -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y,
        -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y,
        +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y,
    };
    float maxcos = dp[0] * inv_len[seq[0]];
    int main_element = 0;
    for (int i = 1; i < 4; ++i) {
      float cosalpha = dp[i] * inv_len[seq[i]];
      if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
    }
    {
      int pindex = seq[main_element];
      float lead_x = vect[pindex].x * inv_len[pindex], lead_y = vect[pindex].y * inv_len[pindex];
      switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
      }
    }
    seq[main_element] += 1;
    if (seq[main_element] == n) seq[main_element] = 0;
    float dx = points[seq[1]].x - points[seq[3]].x, dy = points[seq[1]].y - points[seq[3]].y;
    float width = dx * base_a + dy * base_b;
    dx = points[seq[2]].x - points[seq[0]].x; dy = points[seq[2]].y - points[seq[0]].y;
    float height = -dx * base_b + dy * base_a;
    float area = width * height;
    if (area <= minarea) {
      minarea = area;
      best_left = seq[3]; best_bottom = seq[0];
      best_a = base_a; best_b = base_b; best_w = width; best_h = height;
    }
  }
  float A1 = best_a, B1 = best_b, A2 = -best_b, B2 = best_a;
  float C1 = A1 * points[best_left].x + points[best_left].y * B1;
  float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
  float idet = 1.f / (A1 * B2 - A2 * B1);
  out[0] = (C1 * B2 - C2 * B1) * idet;
  out[1] = (A1 * C2 - A2 * C1) * idet;
  out[2] = A1 * best_w; out[3] = B1 * best_w;
  out[4] = A2 * best_h; out[5] = B2 * best_h;
}

RRect finish_min_area_rect(int kind, const float v[6]) {
  RRect box;
  if (kind == 1) {
    box.cx = v[0] + (v[2] + v[4]) * 0.5f;
    box.cy = v[1] + (v[3] + v[5]) * 0.5f;
    box.w = (float)std::sqrt((double)v[2] * v[2] + (double)v[3] * v[3]);
    box.h = (float)std::sqrt((double)v[4] * v[4] + (double)v[5] * v[5]);
    box.angle = (float)std::atan2((double)v[3], (double)v[2]);
  } else if (kind == 3) {
    box.cx = (v[0] + v[2]) * 0.5f;
    box.cy = (v[1] + v[3]) * 0.5f;
    double dx = v[2] - v[0], dy = v[3] - v[1];
    box.w = (float)std::sqrt(dx * dx + dy * dy);
    box.h = 0;
    box.angle = (float)std::atan2(dy, dx);
  } else if (kind == 4) {
    box.cx = v[0]; box.cy = v[1];
  }
  box.angle = (float)(box.angle * 180 / kPi);
  return box;
}

RRect min_area_rect(const Pt2f* pts, int n) {
  std::vector<Pt2f> hull = convex_hull(std::vector<Pt2f>(pts, pts + n));
  const int hn = (int)hull.size();
  float v[6] = {0, 0, 0, 0, 0, 0};
  if (hn > 2) { rotating_calipers_min_area(hull.data(), hn, v); return finish_min_area_rect(1, v); }
  if (hn == 2) { v[0] = hull[0].x; v[1] = hull[0].y; v[2] = hull[1].x; v[3] = hull[1].y; return finish_min_area_rect(3, v); }
  if (hn == 1) { v[0] = hull[0].x; v[1] = hull[0].y; return finish_min_area_rect(4, v); }
  return finish_min_area_rect(0, v);
}

RRect adjust_coordinates(const RRect& r, float ratio_w, float ratio_h, float ratio_net) {
  Pt2f c[4];
  rect_points(r, c);
  for (int i = 0; i < 4; ++i) { c[i].x *= (ratio_w * ratio_net); c[i].y *= (ratio_h * ratio_net); }
  return min_area_rect(c, 4);
}

bool component_to_rect(const Component& c, int H, int W, RRect* out) {
  const int x = c.x0, y = c.y0, w = c.x1 - c.x0 + 1, h = c.y1 - c.y0 + 1, size = c.area;
  const int niter = (int)std::sqrt((double)(size * std::min(w, h) / (w * h) * 2));  // tuatara.cpp:166, integer inside the sqrt
  const int sx = std::max(0, x - niter), sy = std::max(0, y - niter);               // :168-169
  const int ex = std::min(W, x + w + niter + 1), ey = std::min(H, y + h + niter + 1);  // :170-171
  const int k = 1 + niter, a = k / 2, back = k - 1 - a;  // MORPH_RECT k x k, anchor (k/2,k/2): source s lights [s-back, s+a]
  std::vector<Pt2f> pts;
  pts.reserve(2 * (h + k));
  for (int oy = std::max(sy, y - back); oy <= std::min(ey - 1, y + h - 1 + a); ++oy) {
    int mn = INT_MAX, mx = -1;
    for (int s = std::max(y, oy - a); s <= std::min(y + h - 1, oy + back); ++s) {
      const int* r = c.rows + 2 * (s - y);
      if (r[1] < 0) continue;
      mn = std::min(mn, r[0]); mx = std::max(mx, r[1]);
    }
    if (mx < 0) continue;
    mn = std::max(mn - back, sx); mx = std::min(mx + a, ex - 1);
    pts.push_back(Pt2f{(float)mn, (float)oy});
    if (mx != mn) pts.push_back(Pt2f{(float)mx, (float)oy});
  }
  if (pts.empty()) return false;
  *out = min_area_rect(pts.data(), (int)pts.size());
  return true;
}

CanvasGeom canvas_geometry(int height, int width, int square_size, float mag_ratio) {
  CanvasGeom g;
  float target_size = mag_ratio * std::max(height, width);
  if (target_size > square_size) target_size = (float)square_size;
  g.ratio = target_size / std::max(height, width);
  g.target_h = (int)(height * g.ratio);
  g.target_w = (int)(width * g.ratio);
  g.h32 = g.target_h % 32 != 0 ? g.target_h + (32 - g.target_h % 32) : g.target_h;
  g.w32 = g.target_w % 32 != 0 ? g.target_w + (32 - g.target_w % 32) : g.target_w;
  return g;
}

Tokenizer::Tokenizer() {
  const std::string charset =
      "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&"
      "\\'()*+,-./:;<=>?@[\\]^_`{|}~";
  itos = charset;
  itos.insert(itos.begin(), ']');
  itos.push_back('[');
  itos.push_back('P');
  std::map<char, size_t> stoi;
  for (size_t i = 0; i < itos.size(); ++i) stoi[itos[i]] = i;  // duplicates: last index wins
  eos_id = (int)stoi[']'];
  bos_id = (int)stoi['['];
  pad_id = (int)stoi['P'];
}

std::string Tokenizer::decode(const int* ids, int n) const {
  std::string s;
  for (int i = 0; i < n; ++i) {
    if (ids[i] == eos_id) continue;
    if (ids[i] < 0 || ids[i] >= (int)itos.size()) continue;
    char ch = itos[ids[i]];
    if (ch == ']') break;
    s.push_back(ch);
  }
  return s;
}

}  // namespace ttr
