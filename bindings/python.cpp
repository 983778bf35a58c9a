// pybind11 module `pytuatara` — same surface as /root/reference/bindings/python.cpp:43-58:
//   pytuatara.image_to_data(image, weights_dir, outputs_dir) -> list[{"text": str, "bbox": [x1,y1,x2,y2]}]
// image: uint8 array with 3 dimensions (else RuntimeError("Input array should have 3 dimensions"),
// python.cpp:15-17).  Unlike the reference this copy honours strides and rejects != 3 channels
// instead of silently mis-copying, and the GIL is released while the GPU works.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>

#include "../include/tuatara.h"

namespace py = pybind11;

static py::list image_to_data_wrapper(py::array_t<unsigned char, py::array::c_style | py::array::forcecast> image, std::string weights_dir,
                                      std::string output_dir) {
  py::buffer_info buf = image.request();
  if (buf.ndim != 3) throw std::runtime_error("Input array should have 3 dimensions");
  if (buf.shape[2] != 3) throw std::runtime_error("Input array should have 3 channels");
  const int rows = (int)buf.shape[0], cols = (int)buf.shape[1];
  std::vector<OutputItem> items;
  {
    py::gil_scoped_release nogil;
    items = image_to_data(static_cast<const uint8_t*>(buf.ptr), rows, cols, (std::ptrdiff_t)cols * 3, weights_dir, output_dir);
  }
  py::list result;
  for (const auto& item : items) {
    py::dict d;
    d["text"] = item.text;
    d["bbox"] = item.bbox;
    result.append(d);
  }
  return result;
}

// pytuatara.images_to_data(images, weights_dir, outputs_dir) -> list (one entry per image, input order) of the lists image_to_data returns.
// images: a sequence of uint8 arrays [H, W, 3] of any sizes.  What a caller of the reference writes as a loop over image_to_data (bindings/run_ocr.py:92),
// on one cached engine: same-sized images travel as batches, the host-to-device copies run beside the GPU's work, the GIL is released meanwhile.
static py::list images_to_data_wrapper(py::sequence images, std::string weights_dir, std::string output_dir) {
  std::vector<py::array_t<unsigned char, py::array::c_style | py::array::forcecast>> keep;   // contiguous uint8 views / copies, alive for the call
  std::vector<ImageView> views;
  for (py::handle h : images) {
    auto a = py::array_t<unsigned char, py::array::c_style | py::array::forcecast>::ensure(h);
    if (!a) throw std::runtime_error("images_to_data: every image must convert to a uint8 array");
    py::buffer_info buf = a.request();
    if (buf.ndim != 3) throw std::runtime_error("Input array should have 3 dimensions");
    if (buf.shape[2] != 3) throw std::runtime_error("Input array should have 3 channels");
    views.push_back(ImageView{static_cast<const uint8_t*>(buf.ptr), (int)buf.shape[0], (int)buf.shape[1], (std::ptrdiff_t)buf.shape[1] * 3});
    keep.push_back(std::move(a));
  }
  std::vector<std::vector<OutputItem>> pages;
  {
    py::gil_scoped_release nogil;
    pages = images_to_data(views, weights_dir, output_dir);
  }
  py::list result;
  for (const auto& items : pages) {
    py::list page;
    for (const auto& item : items) {
      py::dict d;
      d["text"] = item.text;
      d["bbox"] = item.bbox;
      page.append(d);
    }
    result.append(page);
  }
  return result;
}

PYBIND11_MODULE(pytuatara, m) {
  m.doc() = "Tuatara ocr (MI355X-native engine)";
  m.def("image_to_data", &image_to_data_wrapper, py::arg("image"), py::arg("weights_dir"), py::arg("outputs_dir"),
        "Extract text and bounding boxes from an image");
  m.def("images_to_data", &images_to_data_wrapper, py::arg("images"), py::arg("weights_dir"), py::arg("outputs_dir"),
        "image_to_data over a sequence of images of any sizes: one list of {text, bbox} per image, in input order");
}
