// dr_fusion.h -- header-compatible replacement for TANDEM's
//   tandem/libdr/dr_fusion/src/dr_fusion/dr_fusion.h
// Same structs, class, members and signatures; every call forwards to the C ABI of libdr_mi355x.so.
// Protocol violations print and exit(EXIT_FAILURE) like tsdf_volume.cu:520-524,635-653,703-713.
#ifndef DR_FUSION_DR_FUSION_H
#define DR_FUSION_DR_FUSION_H

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "dr_mi355x.h"

struct DrFusionOptions {  // dr_fusion.h:18-36 -- layout-identical to drf_options_t
  float voxel_size;
  int num_buckets;
  int bucket_size;
  int num_blocks;
  int block_size;
  int max_sdf_weight;
  float truncation_distance;
  float max_sensor_depth;
  float min_sensor_depth;
  int num_render_streams;

  float fx;
  float fy;
  float cx;
  float cy;
  int height;
  int width;
};
static_assert(sizeof(DrFusionOptions) == sizeof(drf_options_t), "DrFusionOptions must match drf_options_t");

struct DrMesh {  // dr_fusion.h:38-42
  size_t num = 0;
  float *vert = nullptr;
  float *cols = nullptr;
};

class DrFusion {  // dr_fusion.h:44-73
public:
  DrFusion(struct DrFusionOptions const &options) : n_render_(options.num_render_streams), impl(nullptr) {
    check(drf_create(reinterpret_cast<const drf_options_t *>(&options), 0, &impl));
    // the reference mallocs 2 x 720 MB here (dr_fusion.cpp:36-37); allocated lazily with the first mesh instead
    dr_mesh_vert = nullptr;
    dr_mesh_cols = nullptr;
  }
  ~DrFusion() { drf_destroy(impl); free(dr_mesh_vert); free(dr_mesh_cols); }
  DrFusion(const DrFusion &) = delete;
  DrFusion &operator=(const DrFusion &) = delete;

  void IntegrateScanAsync(unsigned char *bgr, float *depth, float const *pose) { check(drf_integrate_scan_async(impl, bgr, depth, pose)); }
  void RenderAsync(std::vector<float const *> camera_poses) { check(drf_render_async(impl, camera_poses.data(), (int) camera_poses.size())); }
  void GetRenderResult(std::vector<unsigned char *> &bgr, std::vector<float *> &depth) {
    if ((!bgr.empty()) || (!depth.empty())) { fprintf(stderr, "Input vectors must be empty.\n"); exit(EXIT_FAILURE); }  // tsdf_volume.cu:715-718
    std::vector<uint8_t *> b(n_render_ > 0 ? n_render_ : 1);
    std::vector<float *> d(n_render_ > 0 ? n_render_ : 1);
    check(drf_get_render_result(impl, b.data(), d.data(), n_render_));
    for (int i = 0; i < n_render_; i++) { bgr.push_back(b[i]); depth.push_back(d[i]); }
  }
  void SaveMeshToFile(std::string const &filename, float lower_corner[3], float upper_corner[3]) { check(drf_save_mesh(impl, filename.c_str(), lower_corner, upper_corner)); }
  struct DrMesh GetMesh(float lower_corner[3], float upper_corner[3]) {  // caller owns vert / cols (dr_fusion.cpp:95-148)
    check(drf_extract_mesh_async(impl, lower_corner, upper_corner));
    size_t ntri = 0;
    check(drf_mesh_num_triangles(impl, &ntri));
    DrMesh m;
    m.vert = (float *) malloc(sizeof(float) * (ntri ? ntri : 1) * 9);
    m.cols = (float *) malloc(sizeof(float) * (ntri ? ntri : 1) * 9);
    check(drf_get_mesh_sync(impl, 3 * ntri, &m.num, m.vert, m.cols));
    return m;
  }
  void ExtractMeshAsync(float lower_corner[3], float upper_corner[3]) { check(drf_extract_mesh_async(impl, lower_corner, upper_corner)); }
  void GetMeshSync() {
    if (!dr_mesh_vert) {
      dr_mesh_vert = (float *) malloc(sizeof(float) * dr_mesh_num_max * 3);
      dr_mesh_cols = (float *) malloc(sizeof(float) * dr_mesh_num_max * 3);
    }
    check(drf_get_mesh_sync(impl, dr_mesh_num_max, &dr_mesh_num, dr_mesh_vert, dr_mesh_cols));
  }
  void Synchronize() { check(drf_synchronize(impl)); }

  size_t dr_mesh_num = 0;
  const size_t dr_mesh_num_max = 60000000;
  float *dr_mesh_vert;
  float *dr_mesh_cols;

private:
  static void check(int status) {
    if (status != DR_OK) { fprintf(stderr, "%s\n", dr_last_error()); exit(EXIT_FAILURE); }
  }
  int n_render_;
  drf_t *impl;
};

#endif  // DR_FUSION_DR_FUSION_H
