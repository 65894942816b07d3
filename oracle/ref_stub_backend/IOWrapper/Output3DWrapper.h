// TEST INFRASTRUCTURE: stand-in for tandem/src/IOWrapper/Output3DWrapper.h, whose real form pulls in Eigen, Sophus and the DSO
// frame types.  tandem_backend.cpp calls three of its virtuals (tandem_backend.cpp:199, 273-274); they are declared here with
// the reference's signatures (Output3DWrapper.h:200-209) and the same empty default bodies.
#pragma once
#include <cstddef>

namespace dso {
namespace IOWrap {
class Output3DWrapper {
public:
  Output3DWrapper() {}
  virtual ~Output3DWrapper() {}
  virtual void pushDrKfImage(unsigned char *bgr) {}
  virtual void pushDrKfDepth(float const *image, float depth_min, float depth_max) {}
  virtual void pushDrKfConfidence(float const *image) {}
  virtual void pushDrMesh(size_t num, float const *vert, float const *cols) {}
};
}  // namespace IOWrap
}  // namespace dso
