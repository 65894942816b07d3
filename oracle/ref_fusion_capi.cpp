// TEST INFRASTRUCTURE (oracle/): C entry points over the REFERENCE's own DrFusion, compiled for the host.
// The reference sources are compiled where they lie under /root/reference (oracle/Makefile.ref, CPU stand-in of the CUDA
// runtime in oracle/ref_stub/); this file only drives them and dumps their state so that oracle/tsdf_oracle.c (the
// restatement that travels to the GPU box) can be compared with the reference itself:
//   DrFusion                            ref:tandem/libdr/dr_fusion/src/dr_fusion/dr_fusion.h:18-73, dr_fusion.cpp
//   TsdfVolume / HashTable / Voxel      ref:.../tsdfvh/tsdf_volume.{h,cu}, hash_table.{h,cu}, voxel.h
//   GetPoint3d / Project / float4x4     ref:.../utils/utils.h:93-108, utils/matrix_utils.h:821-826,914-922,958-1083
// Only tests/ and oracle/gen_golden_*.py may load the library built from this file (oracle/_ref/libdr_fusion_ref.so).
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include <map>
#include <tuple>
#include <utility>
#include <iostream>
#include <fstream>
#include <cmath>
#include <cfloat>
// the dump needs the volume behind DrFusion's private unique_ptr and TsdfVolume's protected coordinate maps
#define private public
#define protected public
#include "dr_fusion.h"
#include "utils/rgbd_sensor.h"
#include "tsdfvh/tsdf_volume.h"
#undef private
#undef protected

using refusion::tsdfvh::TsdfVolume;
using refusion::tsdfvh::Voxel;
using refusion::tsdfvh::HashEntry;

extern "C" {

void* refdrf_create(const DrFusionOptions* o) { return new DrFusion(*o); }
void refdrf_destroy(void* h) { delete (DrFusion*)h; }

void refdrf_integrate(void* h, unsigned char* bgr, float* depth, const float* pose16) {
  ((DrFusion*)h)->IntegrateScanAsync(bgr, depth, pose16);
  ((DrFusion*)h)->Synchronize();
}

// the reference's call order is Integrate -> Render -> GetRenderResult; n must equal num_render_streams
void refdrf_render(void* h, const float* const* poses, int n, unsigned char** bgr_out, float** depth_out) {
  DrFusion* f = (DrFusion*)h;
  std::vector<float const*> p(poses, poses + n);
  f->RenderAsync(p);
  std::vector<unsigned char*> b;
  std::vector<float*> d;
  f->GetRenderResult(b, d);
  size_t px = (size_t)f->volume_->options_.height * f->volume_->options_.width;
  for (int i = 0; i < n; ++i) {
    memcpy(bgr_out[i], b[i], px * 3);
    memcpy(depth_out[i], d[i], px * sizeof(float));
  }
}

int refdrf_num_blocks(void* h) {
  TsdfVolume* v = ((DrFusion*)h)->volume_;
  int n = 0;
  for (int i = 0; i < v->num_entries_; ++i) n += v->entries_[i].pointer != kFreeEntry;
  return n;
}
int refdrf_num_allocated_counter(void* h) { return ((DrFusion*)h)->volume_->num_allocated_blocks_; }

// canonical dump: every occupied hash entry -> (block coordinate, bs^3 voxels of 8 bytes {f32 sdf, u8 x3 colour, u8 weight})
int refdrf_export_blocks(void* h, int max_blocks, int* coords, unsigned char* voxels) {
  TsdfVolume* v = ((DrFusion*)h)->volume_;
  int bs = v->block_size_, nv = bs * bs * bs, n = 0;
  for (int i = 0; i < v->num_entries_ && n < max_blocks; ++i) {
    const HashEntry& e = v->entries_[i];
    if (e.pointer == kFreeEntry) continue;
    coords[3 * n + 0] = e.position.x; coords[3 * n + 1] = e.position.y; coords[3 * n + 2] = e.position.z;
    for (int j = 0; j < nv; ++j) {
      const Voxel& x = v->voxel_blocks_[e.pointer].at(j);
      unsigned char* o = voxels + ((size_t)n * nv + j) * 8;
      memcpy(o, &x.sdf, 4); o[4] = x.color.x; o[5] = x.color.y; o[6] = x.color.z; o[7] = x.weight;
    }
    ++n;
  }
  return n;
}

long refdrf_get_mesh(void* h, const float* lo, const float* hi, long max_tri, float* vert, float* cols) {
  DrFusion* f = (DrFusion*)h;
  float l[3] = {lo[0], lo[1], lo[2]}, u[3] = {hi[0], hi[1], hi[2]};
  DrMesh m = f->GetMesh(l, u);   // num = vertex count (3 per triangle), vert/cols = library-owned arrays
  long ntri = (long)(m.num / 3);
  long n = ntri < max_tri ? ntri : max_tri;
  memcpy(vert, m.vert, (size_t)n * 9 * sizeof(float));
  memcpy(cols, m.cols, (size_t)n * 9 * sizeof(float));
  return ntri;
}

// ---- the header-only arithmetic, function by function (pins for the restatement's helpers) ----
void ref_combine(float sdf, const unsigned char* c, unsigned char w, float vsdf, const unsigned char* vc, unsigned char vw,
                 unsigned char max_weight, float* sdf_out, unsigned char* c_out, unsigned char* w_out) {
  Voxel a; a.sdf = sdf; a.color = make_uchar3(c[0], c[1], c[2]); a.weight = w;
  Voxel b; b.sdf = vsdf; b.color = make_uchar3(vc[0], vc[1], vc[2]); b.weight = vw;
  a.Combine(b, max_weight);
  *sdf_out = a.sdf; c_out[0] = a.color.x; c_out[1] = a.color.y; c_out[2] = a.color.z; *w_out = a.weight;
}
// all (c, vc) in 0..255 x 0..255 for one weight: out[c*256+vc] = combined colour channel (voxel.h:30-35, vw = 1)
void ref_combine_colour_table(unsigned char w, unsigned char* out) {
  for (int c = 0; c < 256; ++c)
    for (int vc = 0; vc < 256; ++vc) {
      Voxel a; a.sdf = 0; a.color = make_uchar3(c, c, c); a.weight = w;
      Voxel b; b.sdf = 0; b.color = make_uchar3(vc, vc, vc); b.weight = 1;
      a.Combine(b, 255);
      out[c * 256 + vc] = a.color.x;
    }
}
static refusion::RgbdSensor sensor_of(const float* k4, int rows, int cols) {
  refusion::RgbdSensor s; s.fx = k4[0]; s.fy = k4[1]; s.cx = k4[2]; s.cy = k4[3]; s.depth_factor = 5000; s.rows = rows; s.cols = cols;
  return s;
}
void ref_point3d(const float* k4, int rows, int cols, int i, float depth, float* out3) {
  float3 p = refusion::GetPoint3d(i, depth, sensor_of(k4, rows, cols));
  out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}
void ref_project(const float* k4, int rows, int cols, const float* p3, int* out2) {
  int2 q = refusion::Project(make_float3(p3[0], p3[1], p3[2]), sensor_of(k4, rows, cols));
  out2[0] = q.x; out2[1] = q.y;
}
float ref_norm(const float* p3) { return refusion::norm(make_float3(p3[0], p3[1], p3[2])); }
void ref_inverse4(const float* m16, float* out16) {
  refusion::float4x4 m(m16);
  refusion::float4x4 inv = m.getInverse();
  memcpy(out16, inv.ptr(), 16 * sizeof(float));
}
void ref_xform(const float* m16, const float* p3, float* out3) {
  refusion::float4x4 m(m16);
  float3 r = m * make_float3(p3[0], p3[1], p3[2]);
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
// coordinate maps of a volume with the given voxel/block size (tsdf_volume.cu:103-145)
void ref_world_maps(void* h, const float* p3, int* global_voxel3, int* block3, int* local3, float* back3) {
  TsdfVolume* v = ((DrFusion*)h)->volume_;
  float3 p = make_float3(p3[0], p3[1], p3[2]);
  int3 g = v->WorldToGlobalVoxel(p), b = v->WorldToBlock(p), l = v->WorldToLocalVoxel(p);
  float3 w = v->GlobalVoxelToWorld(g);
  global_voxel3[0] = g.x; global_voxel3[1] = g.y; global_voxel3[2] = g.z;
  block3[0] = b.x; block3[1] = b.y; block3[2] = b.z;
  local3[0] = l.x; local3[1] = l.y; local3[2] = l.z;
  back3[0] = w.x; back3[1] = w.y; back3[2] = w.z;
}
int ref_hash(void* h, const int* p3) { return ((DrFusion*)h)->volume_->Hash(make_int3(p3[0], p3[1], p3[2])); }

}  // extern "C"
