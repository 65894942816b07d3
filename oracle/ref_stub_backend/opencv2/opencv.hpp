// TEST INFRASTRUCTURE (oracle/Makefile.ref: oracle/_ref/tandem_backend_run).  Stand-in for <opencv2/opencv.hpp> so that the
// reference's tandem/src/tandem/tandem_backend.cpp compiles UNCHANGED on a box without OpenCV.  That file uses cv::Mat as a
// reference-counted byte container only (tandem_backend.h:51-55, tandem_backend.cpp:141-145,166,234-240: copy, vector of,
// `.data`), so this is all a Mat is here: shallow copies that share one buffer, or a view of memory the caller owns.
#pragma once
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {
class Mat {
public:
  Mat() : data(nullptr), rows(0), cols(0), type_(CV_8U) {}
  Mat(int rows_, int cols_, int type) : rows(rows_), cols(cols_), type_(type) {  // owning, zero-filled
    own_ = std::make_shared<std::vector<unsigned char>>((size_t) rows_ * cols_ * elem(type), (unsigned char) 0);
    data = own_->data();
  }
  Mat(int rows_, int cols_, int type, void *external) : data((unsigned char *) external), rows(rows_), cols(cols_), type_(type) {}  // view, as cv::Mat(h, w, type, ptr)
  template <class T> T &at(int r, int c) { return reinterpret_cast<T *>(data)[(size_t) r * cols + c]; }
  template <class T> const T &at(int r, int c) const { return reinterpret_cast<const T *>(data)[(size_t) r * cols + c]; }
  unsigned char *data;
  int rows, cols;
private:
  static size_t elem(int type) { return type == CV_32F ? 4 : 1; }
  int type_;
  std::shared_ptr<std::vector<unsigned char>> own_;
};
}  // namespace cv
