// TEST INFRASTRUCTURE (oracle/ref_stub): the reference's numeric_cuda.h uses boost::format in a debug printer only.
#pragma once
#include <sstream>
#include <string>
namespace boost {
class format {
  std::ostringstream s_;
 public:
  explicit format(const char* f) { s_ << f; }
  template <class T> format& operator%(const T& v) { s_ << ' ' << v; return *this; }
  std::string str() const { return s_.str(); }
};
}  // namespace boost
