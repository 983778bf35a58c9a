/* CPU oracle — post-processing half.  TEST INFRASTRUCTURE ONLY: linked/loaded by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
 * product path.
 *
 * PARITY UNPINNED.  Plain-C restatement of the non-model steps of
 * /root/reference/tuatara.cpp, following it line by line.  The reference delegates
 * the arithmetic to OpenCV 4 (unpinned: CMakeLists.txt:9 `find_package(OpenCV 4)`,
 * setup.sh:24,30) which is absent from /root/reference and from this image, and the
 * reference has no tests or golden vectors.  OpenCV calls are
 * therefore restated from their documented semantics and marked [OpenCV]:
 *   cv::resize INTER_LINEAR 8U   (tuatara.cpp:223, :440)  11-bit fixed point
 *   cv::threshold THRESH_BINARY  (:131-132)
 *   cv::connectedComponentsWithStats 4-connectivity (:142)  raster-first-pixel label order
 *   cv::dilate MORPH_RECT        (:173-174)  anchor k/2, border ignored
 *   cv::findNonZero / cv::minAreaRect (:178-179, :248)
 *   cv::RotatedRect::points / boundingRect (:181, :241, :258, :416)
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/liboracle_post.so oracle/post.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.1415926535897932384626433832795 /* CV_PI */

/* ------------------------------------------------------------------ Tokenizer
 * tuatara.cpp:25-117.  charset literal (:32-34) contains "\\'" = backslash AND
 * apostrophe => 95 chars; itos = EOS + charset + BOS + PAD (:36-39) = 98 entries;
 * stoi is a std::map filled in index order (:41-43) so duplicate chars resolve to
 * the LAST index: eos_id = stoi[']'] = 88, bos_id = stoi['['] = 96, pad_id = 97. */
static const char ORC_CHARSET[] =
    "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&"
    "\\'()*+,-./:;<=>?@[\\]^_`{|}~";

int orc_tokenizer_table(char *itos /* >=99 bytes */, int *eos_id, int *bos_id, int *pad_id) {
  int n = 0;
  itos[n++] = ']'; /* EOS, :37 */
  for (const char *p = ORC_CHARSET; *p; ++p) itos[n++] = *p; /* :36 */
  itos[n++] = '['; /* BOS, :38 */
  itos[n++] = 'P'; /* PAD, :39 */
  itos[n] = 0;
  int stoi[256];
  for (int i = 0; i < 256; ++i) stoi[i] = -1;
  for (int i = 0; i < n; ++i) stoi[(unsigned char)itos[i]] = i; /* :41-43, last wins */
  *eos_id = stoi[(unsigned char)']']; /* :45 */
  *bos_id = stoi[(unsigned char)'[']; /* :46 */
  *pad_id = stoi[(unsigned char)'P']; /* :47 */
  return n;
}

/* decode() for one row: ids -> filter(ids != eos_id) (:108-116) -> ids2tok (:93-99)
 * -> caller's cut at the first EOS char (:497-502).  out needs n+1 bytes. */
int orc_decode_ids(const int64_t *ids, int n, char *out) {
  char itos[100];
  int eos, bos, pad;
  orc_tokenizer_table(itos, &eos, &bos, &pad);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (ids[i] == eos) continue;
    char c = itos[ids[i]];
    if (c == ']') break; /* tokenizer.EOS, :498 */
    out[m++] = c;
  }
  out[m] = 0;
  return m;
}

/* torch::softmax(-1) (:486) then per-row max(-1) (:101-106); first maximal index wins. */
void orc_softmax_argmax(const float *logits, int rows, int C, int64_t *ids, float *probs) {
  for (int r = 0; r < rows; ++r) {
    const float *x = logits + (size_t)r * C;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = x[c] > mx ? x[c] : mx;
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(x[c] - mx);
    int best = 0;
    float bp = expf(x[0] - mx) / sum;
    for (int c = 1; c < C; ++c) {
      float p = expf(x[c] - mx) / sum;
      if (p > bp) { bp = p; best = c; }
    }
    ids[r] = best;
    if (probs) probs[r] = bp;
  }
}

/* ------------------------------------------------------------------ cv::resize, 8UC3
 * [OpenCV] resize.cpp generic path for CV_8U + INTER_LINEAR: coefficients are
 * cvRound(w * 2048) shorts, horizontal pass keeps ints, vertical pass is
 *   ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
 * Special case: an exact 2x2 decimation silently becomes INTER_AREA (box mean). */
static inline int orc_cvround(double v) { return (int)lrint(v); }
static inline int orc_cvfloor(double v) { int i = (int)v; return i - (v < i); }
static inline int orc_cvceil(double v) { int i = (int)v; return i + (v > i); }
static inline short orc_sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_resize_linear_u8c3(const uint8_t *src, int sh, int sw, int sstride, uint8_t *dst, int dh, int dw, int dstride) {
  const int cn = 3;
  if (sh == dh && sw == dw) {
    for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dstride, src + (size_t)y * sstride, (size_t)sw * cn);
    return;
  }
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int iscale_x = orc_cvround(scale_x), iscale_y = orc_cvround(scale_y);
  int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 && fabs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const uint8_t *s0 = src + (size_t)(2 * y) * sstride + (2 * x) * cn + c;
          const uint8_t *s1 = s0 + sstride;
          dst[(size_t)y * dstride + x * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
        }
    return;
  }
  int *xofs = (int *)malloc(sizeof(int) * dw);
  short *ialpha = (short *)malloc(sizeof(short) * dw * 2);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = orc_cvfloor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    ialpha[dx * 2] = orc_sat_short(orc_cvround((1.f - fx) * 2048));
    ialpha[dx * 2 + 1] = orc_sat_short(orc_cvround(fx * 2048));
  }
  int *row0 = (int *)malloc(sizeof(int) * dw * cn), *row1 = (int *)malloc(sizeof(int) * dw * cn);
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = orc_cvfloor(fy);
    fy -= sy;
    /* vertical: rows are clipped, weights are NOT zeroed (resizeGeneric_ clips sy+k) */
    short b0 = orc_sat_short(orc_cvround((1.f - fy) * 2048)), b1 = orc_sat_short(orc_cvround(fy * 2048));
    int sy0 = sy < 0 ? 0 : sy > sh - 1 ? sh - 1 : sy;
    int sy1 = sy + 1 < 0 ? 0 : sy + 1 > sh - 1 ? sh - 1 : sy + 1;
    const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
      int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
      for (int c = 0; c < cn; ++c) {
        row0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
        row1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
      }
    }
    uint8_t *D = dst + (size_t)dy * dstride;
    for (int i = 0; i < dw * cn; ++i) {
      int v = (((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2;
      D[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  }
  free(xofs); free(ialpha); free(row0); free(row1);
}

/* resize_aspect_ratio, tuatara.cpp:206-234.  Returns sizes; `out` (if non-NULL) must
 * hold th32*tw32*3 bytes and receives the zero-padded canvas (:228-229). */
void orc_resize_aspect_ratio_dims(int height, int width, int square_size, float mag_ratio, int *target_h, int *target_w,
                                  int *th32, int *tw32, float *ratio) {
  int mx = height > width ? height : width;
  float target_size = mag_ratio * mx;              /* :211 */
  if (target_size > square_size) target_size = (float)square_size; /* :213-215 */
  *ratio = target_size / mx;                        /* :217 */
  *target_h = (int)(height * *ratio);               /* :219 */
  *target_w = (int)(width * *ratio);                /* :220 */
  *th32 = *target_h % 32 != 0 ? *target_h + (32 - *target_h % 32) : *target_h; /* :225 */
  *tw32 = *target_w % 32 != 0 ? *target_w + (32 - *target_w % 32) : *target_w; /* :226 */
}

void orc_resize_aspect_ratio(const uint8_t *img, int height, int width, int stride, int square_size, float mag_ratio,
                             uint8_t *out) {
  int th, tw, th32, tw32; float ratio;
  orc_resize_aspect_ratio_dims(height, width, square_size, mag_ratio, &th, &tw, &th32, &tw32, &ratio);
  memset(out, 0, (size_t)th32 * tw32 * 3);
  orc_resize_linear_u8c3(img, height, width, stride, out, th, tw, tw32 * 3);
}

/* ------------------------------------------------------------------ RotatedRect helpers
 * rect = {cx, cy, w, h, angle_deg} as float32, like cv::RotatedRect. */
void orc_rect_points(const float *r, float *pt /*8*/) { /* [OpenCV] RotatedRect::points */
  double _angle = r[4] * ORC_PI / 180.;
  float b = (float)cos(_angle) * 0.5f;
  float a = (float)sin(_angle) * 0.5f;
  pt[0] = r[0] - a * r[3] - b * r[2];
  pt[1] = r[1] + b * r[3] - a * r[2];
  pt[2] = r[0] + a * r[3] - b * r[2];
  pt[3] = r[1] - b * r[3] - a * r[2];
  pt[4] = 2 * r[0] - pt[0];
  pt[5] = 2 * r[1] - pt[1];
  pt[6] = 2 * r[0] - pt[2];
  pt[7] = 2 * r[1] - pt[3];
}

void orc_bounding_rect(const float *r, int *xywh) { /* [OpenCV] RotatedRect::boundingRect */
  float pt[8];
  orc_rect_points(r, pt);
  float mnx = fminf(fminf(pt[0], pt[2]), fminf(pt[4], pt[6])), mxx = fmaxf(fmaxf(pt[0], pt[2]), fmaxf(pt[4], pt[6]));
  float mny = fminf(fminf(pt[1], pt[3]), fminf(pt[5], pt[7])), mxy = fmaxf(fmaxf(pt[1], pt[3]), fmaxf(pt[5], pt[7]));
  xywh[0] = orc_cvfloor(mnx);
  xywh[1] = orc_cvfloor(mny);
  xywh[2] = orc_cvceil(mxx) - xywh[0] + 1;
  xywh[3] = orc_cvceil(mxy) - xywh[1] + 1;
}

/* rotated_rect_to_tesseract_format, tuatara.cpp:256-274 */
void orc_tesseract_bbox(const float *r, float *bbox) {
  float v[8];
  orc_rect_points(r, v);
  float min_x = fminf(fminf(v[0], v[2]), fminf(v[4], v[6]));
  float min_y = fminf(fminf(v[1], v[3]), fminf(v[5], v[7]));
  float max_x = fmaxf(fmaxf(v[0], v[2]), fmaxf(v[4], v[6]));
  float max_y = fmaxf(fmaxf(v[1], v[3]), fmaxf(v[5], v[7]));
  bbox[0] = roundf(min_x); bbox[1] = roundf(min_y); bbox[2] = roundf(max_x); bbox[3] = roundf(max_y);
}

/* [OpenCV] cv::minAreaRect on n points: convex hull, then the smallest enclosing rectangle
 * with a side collinear to a hull edge, packed into float32 {center,size,angle} the way
 * minAreaRect packs the calipers' (corner, edge1, edge2) output. */
typedef struct { double x, y; } orc_pt;
static int orc_pt_cmp(const void *a, const void *b) {
  const orc_pt *p = (const orc_pt *)a, *q = (const orc_pt *)b;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  if (p->y != q->y) return p->y < q->y ? -1 : 1;
  return 0;
}
static double orc_cross(orc_pt o, orc_pt a, orc_pt b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

static int orc_convex_hull(orc_pt *p, int n, orc_pt *h) {
  qsort(p, n, sizeof(orc_pt), orc_pt_cmp);
  int m = 0;
  for (int i = 0; i < n; ++i) { /* dedupe */
    if (m && p[m - 1].x == p[i].x && p[m - 1].y == p[i].y) continue;
    p[m++] = p[i];
  }
  n = m;
  if (n < 3) { for (int i = 0; i < n; ++i) h[i] = p[i]; return n; }
  int k = 0;
  for (int i = 0; i < n; ++i) { while (k >= 2 && orc_cross(h[k - 2], h[k - 1], p[i]) <= 0) k--; h[k++] = p[i]; }
  for (int i = n - 2, t = k + 1; i >= 0; --i) { while (k >= t && orc_cross(h[k - 2], h[k - 1], p[i]) <= 0) k--; h[k++] = p[i]; }
  return k - 1;
}

/* [OpenCV] rotatingCalipers(), CALIPERS_MINAREARECT, float32 as in OpenCV: the four
 * calipers sides are (a,b), (-b,a), (-a,-b), (b,-a); each step makes the side with the
 * smallest angle to its polygon edge flush with it and evaluates width*height.
 * out = corner, edge vector 1, edge vector 2. */
static void orc_rotating_calipers(const float *px, const float *py, int n, float *out) {
  float minarea = 3.402823466e+38F;
  float *inv = (float *)malloc(sizeof(float) * n), *vx = (float *)malloc(sizeof(float) * n), *vy = (float *)malloc(sizeof(float) * n);
  int left = 0, bottom = 0, right = 0, top = 0, seq[4];
  float orientation = 0.f, base_a, base_b = 0.f;
  float p0x = px[0], p0y = py[0];
  float left_x = p0x, right_x = p0x, top_y = p0y, bottom_y = p0y;
  for (int i = 0; i < n; ++i) {
    if (p0x < left_x) { left_x = p0x; left = i; }
    if (p0x > right_x) { right_x = p0x; right = i; }
    if (p0y > top_y) { top_y = p0y; top = i; }
    if (p0y < bottom_y) { bottom_y = p0y; bottom = i; }
    int j = i + 1 < n ? i + 1 : 0;
    double dx = px[j] - p0x, dy = py[j] - p0y;
    vx[i] = (float)dx; vy[i] = (float)dy;
    inv[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    p0x = px[j]; p0y = py[j];
  }
  {
    double ax = vx[n - 1], ay = vy[n - 1];
    for (int i = 0; i < n; ++i) {
      double bx = vx[i], by = vy[i];
      double convexity = ax * by - ay * bx;
      if (convexity != 0) { orientation = convexity > 0 ? 1.f : -1.f; break; }
      ax = bx; ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int bl = 0, bb = 0;
  float ba = 1.f, bbv = 0.f, bw = 0.f, bh = 0.f;
  for (int k = 0; k < n; ++k) {
    float dp[4];
    dp[0] = +base_a * vx[seq[0]] + base_b * vy[seq[0]];
    dp[1] = -base_b * vx[seq[1]] + base_a * vy[seq[1]];
    dp[2] = -base_a * vx[seq[2]] - base_b * vy[seq[2]];
    dp[3] = +base_b * vx[seq[3]] - base_a * vy[seq[3]];
    float maxcos = dp[0] * inv[seq[0]];
    int me = 0;
    for (int i = 1; i < 4; ++i) {
      float c = dp[i] * inv[seq[i]];
      if (c > maxcos) { me = i; maxcos = c; }
    }
    {
      int pi = seq[me];
      float lx = vx[pi] * inv[pi], ly = vy[pi] * inv[pi];
      switch (me) {
        case 0: base_a = lx; base_b = ly; break;
        case 1: base_a = ly; base_b = -lx; break;
        case 2: base_a = -lx; base_b = -ly; break;
        default: base_a = -ly; base_b = lx; break;
      }
    }
    seq[me] += 1;
    if (seq[me] == n) seq[me] = 0;
    float dx = px[seq[1]] - px[seq[3]], dy = py[seq[1]] - py[seq[3]];
    float width = dx * base_a + dy * base_b;
    dx = px[seq[2]] - px[seq[0]]; dy = py[seq[2]] - py[seq[0]];
    float height = -dx * base_b + dy * base_a;
    float area = width * height;
    if (area <= minarea) { minarea = area; bl = seq[3]; bb = seq[0]; ba = base_a; bbv = base_b; bw = width; bh = height; }
  }
  float A1 = ba, B1 = bbv, A2 = -bbv, B2 = ba;
  float C1 = A1 * px[bl] + py[bl] * B1;
  float C2 = A2 * px[bb] + py[bb] * B2;
  float idet = 1.f / (A1 * B2 - A2 * B1);
  out[0] = (C1 * B2 - C2 * B1) * idet;
  out[1] = (A1 * C2 - A2 * C1) * idet;
  out[2] = A1 * bw; out[3] = B1 * bw;
  out[4] = A2 * bh; out[5] = B2 * bh;
  free(inv); free(vx); free(vy);
}

/* mode 0: OpenCV-style float32 rotating calipers (what the reference runs).
 * mode 1: exhaustive search over hull edges in double (independent cross-check for tests). */
static void orc_min_area_rect_mode(const double *xy, int n, float *rect, int mode) {
  orc_pt *p = (orc_pt *)malloc(sizeof(orc_pt) * (n + 1)), *h = (orc_pt *)malloc(sizeof(orc_pt) * (2 * n + 2));
  for (int i = 0; i < n; ++i) { p[i].x = (float)xy[2 * i]; p[i].y = (float)xy[2 * i + 1]; }  /* hull.convertTo(CV_32F) */
  int hn = orc_convex_hull(p, n, h);
  rect[0] = rect[1] = rect[2] = rect[3] = rect[4] = 0.f;
  if (hn > 2) {
    float f0x, f0y, f1x, f1y, f2x, f2y;
    if (mode == 0) {
      float *hx = (float *)malloc(sizeof(float) * hn), *hy = (float *)malloc(sizeof(float) * hn), out[6];
      for (int i = 0; i < hn; ++i) { hx[i] = (float)h[i].x; hy[i] = (float)h[i].y; }
      orc_rotating_calipers(hx, hy, hn, out);
      f0x = out[0]; f0y = out[1]; f1x = out[2]; f1y = out[3]; f2x = out[4]; f2y = out[5];
      free(hx); free(hy);
    } else {
      double best = -1, o0x = 0, o0y = 0, o1x = 0, o1y = 0, o2x = 0, o2y = 0;
      for (int i = 0; i < hn; ++i) {
        orc_pt a = h[i], b = h[(i + 1) % hn];
        double dx = b.x - a.x, dy = b.y - a.y, len = sqrt(dx * dx + dy * dy);
        double ux = dx / len, uy = dy / len, nx = -uy, ny = ux;
        double mnu = 1e300, mxu = -1e300, mnn = 1e300, mxn = -1e300;
        for (int j = 0; j < hn; ++j) {
          double pu = h[j].x * ux + h[j].y * uy, pn = h[j].x * nx + h[j].y * ny;
          if (pu < mnu) mnu = pu; if (pu > mxu) mxu = pu;
          if (pn < mnn) mnn = pn; if (pn > mxn) mxn = pn;
        }
        double area = (mxu - mnu) * (mxn - mnn);
        if (best < 0 || area < best) {
          best = area;
          o0x = ux * mnu + nx * mnn; o0y = uy * mnu + ny * mnn;
          o1x = ux * (mxu - mnu); o1y = uy * (mxu - mnu);
          o2x = nx * (mxn - mnn); o2y = ny * (mxn - mnn);
        }
      }
      f0x = (float)o0x; f0y = (float)o0y; f1x = (float)o1x; f1y = (float)o1y; f2x = (float)o2x; f2y = (float)o2y;
    }
    rect[0] = f0x + (f1x + f2x) * 0.5f;
    rect[1] = f0y + (f1y + f2y) * 0.5f;
    rect[2] = (float)sqrt((double)f1x * f1x + (double)f1y * f1y);
    rect[3] = (float)sqrt((double)f2x * f2x + (double)f2y * f2y);
    rect[4] = (float)atan2((double)f1y, (double)f1x);
  } else if (hn == 2) {
    rect[0] = ((float)h[0].x + (float)h[1].x) * 0.5f;
    rect[1] = ((float)h[0].y + (float)h[1].y) * 0.5f;
    double dx = (float)h[1].x - (float)h[0].x, dy = (float)h[1].y - (float)h[0].y;
    rect[2] = (float)sqrt(dx * dx + dy * dy);
    rect[3] = 0;
    rect[4] = (float)atan2(dy, dx);
  } else if (hn == 1) {
    rect[0] = (float)h[0].x; rect[1] = (float)h[0].y;
  }
  rect[4] = (float)(rect[4] * 180 / ORC_PI);
  free(p); free(h);
}

void orc_min_area_rect(const double *xy, int n, float *rect) { orc_min_area_rect_mode(xy, n, rect, 0); }
void orc_min_area_rect_exhaustive(const double *xy, int n, float *rect) { orc_min_area_rect_mode(xy, n, rect, 1); }

/* ------------------------------------------------------------------ connected components
 * [OpenCV] connectedComponentsWithStats(img8u, labels32S, stats, centroids, 4): two-pass
 * union-find; final label numbers follow the raster order of each component's first pixel.
 * stats row = {LEFT, TOP, WIDTH, HEIGHT, AREA}.  Returns nLabels (background = 0). */
static int orc_find(int *par, int i) { while (par[i] != i) { par[i] = par[par[i]]; i = par[i]; } return i; }

int orc_connected_components4(const uint8_t *img, int H, int W, int32_t *labels, int32_t *stats /*5*nLabels, cap H*W/2+2*/) {
  int *par = (int *)malloc(sizeof(int) * ((size_t)H * W / 2 + 2));
  int np = 1;
  par[0] = 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      size_t i = (size_t)y * W + x;
      if (!img[i]) { labels[i] = 0; continue; }
      int up = y > 0 && img[i - W] ? labels[i - W] : 0;
      int lf = x > 0 && img[i - 1] ? labels[i - 1] : 0;
      if (!up && !lf) { par[np] = np; labels[i] = np++; }
      else if (up && lf) {
        int a = orc_find(par, up), b = orc_find(par, lf);
        if (a < b) par[b] = a; else par[a] = b;
        labels[i] = a < b ? a : b;
      } else labels[i] = up ? up : lf;
    }
  int *remap = (int *)calloc(np, sizeof(int));
  int n = 1;
  for (int i = 1; i < np; ++i) { int r = orc_find(par, i); if (r == i) remap[i] = n++; }
  for (int i = 1; i < np; ++i) remap[i] = remap[orc_find(par, i)];
  for (int k = 0; k < n; ++k) { stats[5 * k] = W; stats[5 * k + 1] = H; stats[5 * k + 2] = -1; stats[5 * k + 3] = -1; stats[5 * k + 4] = 0; }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      size_t i = (size_t)y * W + x;
      int k = labels[i] ? remap[labels[i]] : 0;
      labels[i] = k;
      int32_t *s = stats + 5 * k;
      if (x < s[0]) s[0] = x; if (y < s[1]) s[1] = y;
      if (x > s[2]) s[2] = x; if (y > s[3]) s[3] = y;
      s[4]++;
    }
  for (int k = 0; k < n; ++k) { stats[5 * k + 2] = stats[5 * k + 2] - stats[5 * k] + 1; stats[5 * k + 3] = stats[5 * k + 3] - stats[5 * k + 1] + 1; }
  free(par); free(remap);
  return n;
}

/* ------------------------------------------------------------------ get_detected_boxes
 * tuatara.cpp:119-204.  textmap/linkmap: H*W float32 with element stride `es`
 * (the reference passes channel slices of the [H,W,2] output, :393-394).
 * rects: up to max_rects x 5 floats (heatmap pixel units).  labels_out (optional): H*W.
 * Returns the number of boxes (det.size()). */
int orc_get_detected_boxes(const float *textmap, const float *linkmap, int es, int H, int W, float text_threshold,
                           float link_threshold, float low_text, float *rects, int max_rects, int32_t *labels_out,
                           float *textnorm_out) {
  size_t npx = (size_t)H * W;
  float *tn = (float *)malloc(sizeof(float) * npx), *ln = (float *)malloc(sizeof(float) * npx);
  float tmin = textmap[0], tmax = textmap[0], lmin = linkmap[0], lmax = linkmap[0];
  for (size_t i = 0; i < npx; ++i) {
    float t = textmap[i * es], l = linkmap[i * es];
    tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax;
    lmin = l < lmin ? l : lmin; lmax = l > lmax ? l : lmax;
  }
  float td = tmax - tmin, ld = lmax - lmin;
  for (size_t i = 0; i < npx; ++i) { /* :120-121, fp32 elementwise (x - min) / (max - min) */
    tn[i] = (textmap[i * es] - tmin) / td;
    ln[i] = (linkmap[i * es] - lmin) / ld;
  }
  if (textnorm_out) memcpy(textnorm_out, tn, sizeof(float) * npx);
  uint8_t *ts = (uint8_t *)malloc(npx), *ls = (uint8_t *)malloc(npx), *comb = (uint8_t *)malloc(npx);
  for (size_t i = 0; i < npx; ++i) {
    ts[i] = tn[i] > low_text;        /* :131 THRESH_BINARY, strict > */
    ls[i] = ln[i] > link_threshold;  /* :132 */
    int s = ts[i] + ls[i];           /* :136 min(max(a+b,0),1) -> :137 8U */
    comb[i] = (uint8_t)(s > 1 ? 1 : s);
  }
  int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * npx);
  int32_t *stats = (int32_t *)malloc(sizeof(int32_t) * 5 * (npx / 2 + 2));
  int nLabels = orc_connected_components4(comb, H, W, labels, stats); /* :142 */
  if (labels_out) memcpy(labels_out, labels, sizeof(int32_t) * npx);

  uint8_t *segmap = (uint8_t *)malloc(npx), *tmp = (uint8_t *)malloc(npx);
  double *pts = (double *)malloc(sizeof(double) * 2 * npx);
  int ndet = 0;
  for (int k = 1; k < nLabels; ++k) {           /* :146 */
    int size = stats[5 * k + 4];
    if (size < 10) continue;                    /* :147-148 */
    float maxv = -INFINITY;                     /* :150-152 minMaxLoc under mask */
    for (size_t i = 0; i < npx; ++i) if (labels[i] == k && tn[i] > maxv) maxv = tn[i];
    if ((double)maxv < (double)text_threshold) continue; /* :154 */
    for (size_t i = 0; i < npx; ++i) {
      segmap[i] = labels[i] == k ? 255 : 0;     /* :156-157 */
      if (ls[i] == 1 && ts[i] == 0) segmap[i] = 0; /* :160 */
    }
    int x = stats[5 * k], y = stats[5 * k + 1], w = stats[5 * k + 2], h = stats[5 * k + 3];
    int niter = (int)sqrt((double)(size * (w < h ? w : h) / (w * h) * 2)); /* :166, all-int inside */
    int sx = x - niter > 0 ? x - niter : 0;             /* :168 */
    int sy = y - niter > 0 ? y - niter : 0;             /* :169 */
    int ex = x + w + niter + 1 < W ? x + w + niter + 1 : W; /* :170 */
    int ey = y + h + niter + 1 < H ? y + h + niter + 1 : H; /* :171 */
    int ks = 1 + niter, ax = ks / 2, ay = ks / 2;       /* :173 MORPH_RECT, default anchor = center */
    /* :174 dilate the ROI in place; reads reach into the parent image, image border ignored */
    memcpy(tmp, segmap, npx);
    for (int yy = sy; yy < ey; ++yy)
      for (int xx = sx; xx < ex; ++xx) {
        uint8_t m = 0;
        for (int j = 0; j < ks && !m; ++j) {
          int y2 = yy + j - ay;
          if (y2 < 0 || y2 >= H) continue;
          for (int i = 0; i < ks; ++i) {
            int x2 = xx + i - ax;
            if (x2 < 0 || x2 >= W) continue;
            if (tmp[(size_t)y2 * W + x2]) { m = 255; break; }
          }
        }
        segmap[(size_t)yy * W + xx] = m;
      }
    int np = 0;                                  /* :177-178 findNonZero */
    for (int yy = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx)
        if (segmap[(size_t)yy * W + xx]) { pts[2 * np] = xx; pts[2 * np + 1] = yy; np++; }
    /* :180-198 compute box[], ratio fix and rotation but never use them (:200 pushes `rectangle`) */
    if (ndet < max_rects) orc_min_area_rect(pts, np, rects + 5 * ndet); /* :179, :200 */
    ndet++;
  }
  free(tn); free(ln); free(ts); free(ls); free(comb); free(labels); free(stats); free(segmap); free(tmp); free(pts);
  return ndet;
}

/* adjust_result_coordinates, tuatara.cpp:236-253 (ratio_net = 2) */
void orc_adjust_result_coordinates(const float *rects, int n, float ratio_w, float ratio_h, float *out) {
  for (int i = 0; i < n; ++i) {
    float c[8];
    orc_rect_points(rects + 5 * i, c);           /* :241 */
    double d[8];
    for (int j = 0; j < 4; ++j) {                /* :243-246 */
      c[2 * j] *= (ratio_w * 2.f);
      c[2 * j + 1] *= (ratio_h * 2.f);
      d[2 * j] = c[2 * j]; d[2 * j + 1] = c[2 * j + 1];
    }
    orc_min_area_rect(d, 4, out + 5 * i);        /* :248 */
  }
}

/* crop (:416) + cv::resize to 128x32 (:440) + BGR2RGB (:441).  `image` is the
 * caller's image AFTER the in-place swap at :349.  clamp != 0 clips the rect to the
 * image (the reference would throw); returns 0 on success, -1 if the
 * (unclamped) rect leaves the image or is empty. */
int orc_crop_resize(const uint8_t *image, int H, int W, int stride, const float *rect, int clamp, uint8_t *out /*32*128*3*/) {
  int r[4];
  orc_bounding_rect(rect, r);
  int x0 = r[0], y0 = r[1], x1 = r[0] + r[2], y1 = r[1] + r[3];
  if (clamp) { if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W) x1 = W; if (y1 > H) y1 = H; }
  if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x1 <= x0 || y1 <= y0) return -1;
  uint8_t tmp[32 * 128 * 3];
  orc_resize_linear_u8c3(image + (size_t)y0 * stride + x0 * 3, y1 - y0, x1 - x0, stride, tmp, 32, 128, 128 * 3);
  for (int i = 0; i < 32 * 128; ++i) { out[3 * i] = tmp[3 * i + 2]; out[3 * i + 1] = tmp[3 * i + 1]; out[3 * i + 2] = tmp[3 * i]; }
  return 0;
}
