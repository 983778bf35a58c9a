// Host-only driver for the sanitizer builds (tests/test_sanitizers_cpu.py): the engine's untrusted-input parsers and its host
// threading, compiled WITHOUT HIP under -fsanitize=address,undefined (and -fsanitize=thread for the pool).
//   host_san ttrw <file>      load a .ttrw weight file (tuatara_amd/csrc/host_util.h: WeightFile); "ok <tensors>" or "rejected: <why>"
//   host_san png <file>       decode a PNG (examples/png_decode.h); "ok <w>x<h>" or "rejected: <why>"
//   host_san pool <threads> <rounds>   HostPool stress: uneven work, exceptions thrown inside tasks, pool reuse and teardown
//   host_san geom <seed> <n>  calipers / component_to_rect / tokenizer on random inputs (geometry.cpp)
// Every outcome of a malformed input must be a clean C++ exception: the sanitizers turn anything else into a non-zero exit.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../examples/png_decode.h"
#include "../../tuatara_amd/csrc/geometry.h"
#include "../../tuatara_amd/csrc/host_util.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  try {
    if (mode == "ttrw") {
      ttr::WeightFile wf(argv[2]);
      size_t total = 0;
      for (const auto& kv : wf.t) total += kv.second.data.size();
      printf("ok %zu tensors %zu values\n", wf.t.size(), total);
    } else if (mode == "png") {
      const pngdec::Image im = pngdec::read(argv[2]);
      printf("ok %dx%d\n", im.cols, im.rows);
    } else if (mode == "pool") {
      const int threads = atoi(argv[2]), rounds = argc > 3 ? atoi(argv[3]) : 200;
      long long checksum = 0;
      for (int rep = 0; rep < 3; ++rep) {
        ttr::HostPool pool(threads);
        std::vector<long long> out(257);
        for (int r = 0; r < rounds; ++r) {
          const int n = 1 + (r * 37) % 257;
          std::function<void(int)> f = [&](int i) {
            long long s = 0;
            for (int k = 0; k < (i % 7) * 1000; ++k) s += k ^ i;
            out[i] = s + i;
            if (r % 11 == 3 && i == n / 2) throw std::runtime_error("task failure");
          };
          try { pool.run(n, f); } catch (const std::runtime_error&) { checksum += 1; }
          for (int i = 0; i < n; ++i) checksum += out[i] & 1;
        }
      }
      printf("ok pool %lld\n", checksum);
    } else if (mode == "geom") {
      std::mt19937 rng((unsigned)atoi(argv[2]));
      const int n = argc > 3 ? atoi(argv[3]) : 200;
      std::uniform_real_distribution<float> u(-50.f, 500.f);
      double acc = 0;
      for (int it = 0; it < n; ++it) {
        const int m = 1 + (int)(rng() % 40);
        std::vector<ttr::Pt2f> pts(m);
        for (auto& p : pts) { p.x = u(rng); p.y = u(rng); }
        if (it % 5 == 0) for (auto& p : pts) p.y = pts[0].y;                 // collinear
        if (it % 7 == 0) for (auto& p : pts) p = pts[0];                     // all the same point
        const ttr::RRect r = ttr::min_area_rect(pts.data(), m);
        float bb[4]; int xywh[4];
        ttr::tesseract_bbox(r, bb); ttr::bounding_rect(r, xywh);
        acc += r.w + r.h + bb[0] + xywh[2];
        // a component given by row extremes (some rows empty)
        const int H = 64, W = 96, y0 = (int)(rng() % 40), y1 = y0 + (int)(rng() % 20), x0 = (int)(rng() % 60), x1 = x0 + (int)(rng() % 30);
        std::vector<int> rows(2 * (y1 - y0 + 1));
        int area = 0;
        for (int y = y0; y <= y1; ++y) {
          if (rng() % 6 == 0) { rows[2 * (y - y0)] = 2147483647; rows[2 * (y - y0) + 1] = -1; continue; }
          const int a = x0 + (int)(rng() % (x1 - x0 + 1)), b = a + (int)(rng() % (x1 - a + 1));
          rows[2 * (y - y0)] = a; rows[2 * (y - y0) + 1] = b; area += b - a + 1;
        }
        ttr::Component c{0, area, x0, y0, x1, y1, rows.data()};
        ttr::RRect out;
        if (ttr::component_to_rect(c, H, W, &out)) acc += out.w;
      }
      ttr::Tokenizer tok;
      for (int it = 0; it < n; ++it) {
        int ids[26];
        for (int& v : ids) v = (int)(rng() % 120) - 10;                       // ids outside the table too
        acc += (double)tok.decode(ids, 26).size();
      }
      printf("ok geom %.3f\n", acc);
    } else {
      return 2;
    }
  } catch (const std::exception& ex) {
    printf("rejected: %s\n", ex.what());
  }
  return 0;
}
