// Counterpart of the reference's example CLIs (examples/resume.cpp:7-12, examples/table.cpp:9-10):
//   ocr_cli <image.png> <weights_dir> <outputs_dir>
// reads the image as BGR (what cv::imread(path, cv::IMREAD_COLOR) hands the reference), calls image_to_data and,
// unlike the reference (which discards the result), prints one "x1 y1 x2 y2<TAB>text" line per item.
//   ocr_cli --decode-only <image.png> <out.raw>   writes the decoded BGR bytes (tests of the PNG reader; no GPU).
#include <cstdio>
#include <iostream>

#include "../include/tuatara.h"
#include "png_decode.h"

int main(int argc, const char** argv) {
  try {
    if (argc == 4 && std::string(argv[1]) == "--decode-only") {
      pngdec::Image img = pngdec::read(argv[2]);
      FILE* f = fopen(argv[3], "wb");
      if (!f) throw std::runtime_error("cannot write output");
      fwrite(img.bgr.data(), 1, img.bgr.size(), f);
      fclose(f);
      printf("%d %d\n", img.rows, img.cols);
      return 0;
    }
    if (argc != 4) {
      std::cerr << "usage: ocr_cli <image.png> <weights_dir> <outputs_dir>" << std::endl;
      return 2;
    }
    pngdec::Image img = pngdec::read(argv[1]);
    std::vector<OutputItem> items = image_to_data(img.bgr.data(), img.rows, img.cols, (std::ptrdiff_t)img.cols * 3, argv[2], argv[3]);
    for (const OutputItem& it : items) printf("%g %g %g %g\t%s\n", it.bbox[0], it.bbox[1], it.bbox[2], it.bbox[3], it.text.c_str());
    return 0;
  } catch (const std::exception& ex) {
    std::cerr << "ocr_cli: " << ex.what() << std::endl;
    return 1;
  }
}
