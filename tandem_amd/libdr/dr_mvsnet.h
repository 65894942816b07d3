// dr_mvsnet.h -- header-compatible replacement for TANDEM's
//   tandem/libdr/dr_mvsnet/src/dr_mvsnet/dr_mvsnet.h
// Same classes, members and signatures (DrMvsnetOutput, DrMvsnet, test_dr_mvsnet), so
// tandem_backend.cpp / FullSystem.cpp compile unchanged; every call forwards to the C ABI of
// libdr_mi355x.so (include/dr_mi355x.h).  Error behaviour follows the reference: protocol
// violations print to stderr and exit(EXIT_FAILURE) (dr_mvsnet.cpp:100-102,156-157).
#ifndef DR_MVSNET_H
#define DR_MVSNET_H

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "dr_mi355x.h"

class DrMvsnetOutput {  // dr_mvsnet.h:12-34
public:
  DrMvsnetOutput(int height, int width) : height(height), width(width) {
    depth = (float *) malloc(sizeof(float) * width * height);
    confidence = (float *) malloc(sizeof(float) * width * height);
    depth_dense = (float *) malloc(sizeof(float) * width * height);
    confidence_dense = (float *) malloc(sizeof(float) * width * height);
  }
  // A VIEW of the engine's page-locked result block (DrMvsnet::GetResultView; no reference counterpart): the arrays are not owned.
  DrMvsnetOutput(int height, int width, float *d, float *c, float *dd, float *cd)
      : depth(d), confidence(c), depth_dense(dd), confidence_dense(cd), height(height), width(width), view_(true) {}
  ~DrMvsnetOutput() { if (!view_) { free(depth); free(confidence); free(depth_dense); free(confidence_dense); } }
  DrMvsnetOutput(const DrMvsnetOutput &) = delete;
  DrMvsnetOutput &operator=(const DrMvsnetOutput &) = delete;

  float *depth;
  float *confidence;
  float *depth_dense;
  float *confidence_dense;
  const int height;
  const int width;

private:
  const bool view_ = false;
};

class DrMvsnet {  // dr_mvsnet.h:36-66
public:
  // `filename` names a TDMW weight blob (tandem_amd/weights.py) where the reference took a TorchScript archive.
  explicit DrMvsnet(char const *filename) : impl(nullptr), height_(0), width_(0) {
    if (drm_create(filename, 0, &impl) != DR_OK) {
      // the reference only prints when CUDA is unavailable and crashes later (dr_mvsnet.cpp:22-23); we stop here.
      fprintf(stderr, "DrMvsnet: %s\n", dr_last_error());
      exit(EXIT_FAILURE);
    }
  }
  ~DrMvsnet() { delete spare_; drm_destroy(impl); }
  DrMvsnet(const DrMvsnet &) = delete;
  DrMvsnet &operator=(const DrMvsnet &) = delete;

  // Blocking for last input. Non-blocking for this input.
  void CallAsync(int height, int width, int view_num, int ref_index, unsigned char **bgrs, float const *intrinsic_matrix,
                 float **cam_to_worlds, float depth_min, float depth_max, float discard_percentage, bool debug_print = false) {
    if (debug_print) {
      printf("--- DrMvsnet::CallAsync ---\nW=%d, H=%d, view_num=%d, ref_index=%d, depth_min=%f, depth_max=%f, discard_percentage=%f\n",
             width, height, view_num, ref_index, depth_min, depth_max, discard_percentage);
    }
    height_ = height; width_ = width;
    check(drm_call_async(impl, height, width, view_num, ref_index, (const uint8_t *const *) bgrs, intrinsic_matrix,
                         (const float *const *) cam_to_worlds, depth_min, depth_max, discard_percentage));
    // The result object GetResult() will hand out is allocated and its pages TOUCHED here, while the device works on the window: TandemBackend
    // never deletes its DrMvsnetOutputs, so every result lands in 4.9 MB of fresh memory, and the first-touch page faults of that memory
    // (1200 of them at 640 x 480) were paid inside GetResult, on the caller's critical path -- more time than the copy itself.
    if (spare_ && (spare_->height != height || spare_->width != width)) { delete spare_; spare_ = nullptr; }
    if (!spare_ && !result_views_) {
      spare_ = new DrMvsnetOutput(height, width);
      const size_t bytes = sizeof(float) * (size_t) width * height;
      memset(spare_->depth, 0, bytes); memset(spare_->confidence, 0, bytes); memset(spare_->depth_dense, 0, bytes); memset(spare_->confidence_dense, 0, bytes);
    }
  }
  // Blocking.  Ownership of the result passes to the caller (delete it), as in the reference.
  // With SetResultViews(true) (extension, below) the object's four maps are VIEWS of the engine's page-locked result block instead of copies.
  DrMvsnetOutput *GetResult() {
    if (result_views_) return GetResultView();
    DrMvsnetOutput *out = spare_ ? spare_ : new DrMvsnetOutput(height_, width_);
    spare_ = nullptr;
    if (drm_get_result(impl, out->depth, out->confidence, out->depth_dense, out->confidence_dense) != DR_OK) {
      delete out;
      check(DR_ERR_PROTOCOL);
    }
    return out;
  }
  // ---- extensions (no reference counterpart): the operator boundary without its two host copies ----
  // GetResult() whose maps are VIEWS of the page-locked block the device wrote them to (no 4.9 MB copy at 640 x 480).  Valid while the
  // next CallAsync is processed; overwritten by the one after it.  Delete the object as usual (the arrays are not freed).
  DrMvsnetOutput *GetResultView() {
    const float *d, *c, *dd, *cd;
    check(drm_get_result_view(impl, &d, &c, &dd, &cd));
    return new DrMvsnetOutput(height_, width_, const_cast<float *>(d), const_cast<float *>(c), const_cast<float *>(dd), const_cast<float *>(cd));
  }
  // Page-locked memory for key-frame images: CallAsync uploads windows whose images ALL live in such memory in place, skipping the
  // gather into the engine's staging block (6.45 MB at 640 x 480 x 7).
  // GetResult() without its 4.9 MB host copy (640 x 480): the maps handed out are views (GetResultView) -- valid while the NEXT CallAsync is processed, overwritten by
  // the one after it.  That is exactly how TandemBackend uses a result (tandem_backend.cpp:147-177: read during the call that follows, never touched again), so its
  // unchanged code may run on views; a caller that keeps results longer must not switch this on.
  void SetResultViews(bool on) { result_views_ = on; }
  // The key-frame feature cache (drm_set_feature_cache): FeatureNet runs on the window's NEW image only; call once after construction.  0 = off.
  void SetFeatureCache(int key_frames) { check(drm_set_feature_cache(impl, key_frames)); }
  static unsigned char *AllocImage(size_t bytes) { return static_cast<unsigned char *>(drm_host_alloc(bytes)); }
  static void FreeImage(unsigned char *p) { drm_host_free(p); }

  // Blocking
  void Wait() { check(drm_wait(impl)); }
  // Non-blocking
  bool Ready() { return drm_ready(impl) != 0; }

private:
  static void check(int status) {
    if (status != DR_OK) { fprintf(stderr, "%s\n", dr_last_error()); exit(EXIT_FAILURE); }
  }
  drm_t *impl;
  int height_, width_;
  DrMvsnetOutput *spare_ = nullptr;  // the next GetResult()'s object, pages already touched (CallAsync)
  bool result_views_ = false;
};

// test_dr_mvsnet (dr_mvsnet.cpp:376-556): feeds a stored window through CallAsync/Ready/GetResult
// `repetitions` (+5 warm-up) times and passes iff mean-abs error < 1e-2 for stage-3 depth and confidence.
// `filename_inputs` is a TDMS sample file (tools/export_fixture.py) where the reference read sample_inputs.pt:
//   "TDMS0001" | int32 V,H,W,ref_index | float depth_min,depth_max,discard | float K[9] | float c2w[V*16]
//   | u8 bgr[V*H*W*3] | float depth_ref[H*W] | float confidence_ref[H*W]
// `out_folder` (dr_mvsnet.cpp:515-538): the first repetition's filtered depth map is written to
// <out_folder>pred_outputs.npy (H x W float32, NumPy format) where the reference pickles the same tensor to pred_outputs.pt.
inline bool test_dr_mvsnet(DrMvsnet &model, char const *filename_inputs, bool print = false, int repetitions = 1,
                           char const *out_folder = NULL) {
  FILE *f = fopen(filename_inputs, "rb");
  if (!f) { fprintf(stderr, "test_dr_mvsnet: cannot open %s\n", filename_inputs); return false; }
  char magic[8]; int hdr[4]; float sc[3], K[9];
  bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, "TDMS0001", 8) && fread(hdr, 4, 4, f) == 4 && fread(sc, 4, 3, f) == 3 &&
            fread(K, 4, 9, f) == 9;
  if (!ok) { fclose(f); fprintf(stderr, "test_dr_mvsnet: bad sample file\n"); return false; }
  const int V = hdr[0], H = hdr[1], W = hdr[2], ref = hdr[3];
  const size_t npx = (size_t) H * W;
  std::vector<float> c2w((size_t) V * 16), dref(npx), cref(npx);
  std::vector<unsigned char> img((size_t) V * npx * 3);
  ok = fread(c2w.data(), 4, c2w.size(), f) == c2w.size() && fread(img.data(), 1, img.size(), f) == img.size() &&
       fread(dref.data(), 4, npx, f) == npx && fread(cref.data(), 4, npx, f) == npx;
  fclose(f);
  if (!ok) { fprintf(stderr, "test_dr_mvsnet: truncated sample file\n"); return false; }
  std::vector<unsigned char *> bgrs(V);
  std::vector<float *> c2ws(V);
  for (int v = 0; v < V; v++) { bgrs[v] = img.data() + (size_t) v * npx * 3; c2ws[v] = c2w.data() + 16 * v; }
  if (print) printf("View Num: %d, ref index: %d\n", V, ref);
  double e1 = 0, e2 = 0, e3 = 0;
  bool correct = true;
  const int warmup = (repetitions == 1) ? 0 : 5;
  for (int rep = 0; rep < repetitions + warmup; rep++) {
    if (rep == warmup) e1 = e2 = e3 = 0;
    auto t = std::chrono::high_resolution_clock::now();
    model.CallAsync(H, W, V, ref, bgrs.data(), K, c2ws.data(), sc[0], sc[1], sc[2]);
    e1 += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t).count();
    t = std::chrono::high_resolution_clock::now();
    (void) model.Ready();
    e2 += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t).count();
    t = std::chrono::high_resolution_clock::now();
    DrMvsnetOutput *out = model.GetResult();
    e3 += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t).count();
    double ed = 0, ec = 0;
    for (size_t i = 0; i < npx; i++) { ed += std::fabs(out->depth[i] - dref[i]); ec += std::fabs(out->confidence[i] - cref[i]); }
    ed /= npx; ec /= npx;
    const double atol = 1e-2;  // dr_mvsnet.cpp:509
    if (print) printf("Correctness:\n\tDepth correct     : %d, error: %g\n\tConfidence correct: %d, error: %g\n", ed < atol, ed, ec < atol, ec);
    correct &= ed < atol;
    correct &= ec < atol;
    if (out_folder && rep == 0) {
      const std::string out_name = std::string(out_folder) + "pred_outputs.npy";
      printf("Writing Result to: %s\n", out_name.c_str());
      FILE *fo = fopen(out_name.c_str(), "wb");
      if (fo) {
        char hdr[128];
        int n = snprintf(hdr, sizeof hdr, "{'descr': '<f4', 'fortran_order': False, 'shape': (%d, %d), }", H, W);
        while ((10 + n + 1) % 64) hdr[n++] = ' ';
        hdr[n++] = '\n';
        const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char) (n & 255), (unsigned char) (n >> 8)};
        fwrite(pre, 1, 10, fo); fwrite(hdr, 1, (size_t) n, fo); fwrite(out->depth, 4, npx, fo);
        fclose(fo);
      } else {
        fprintf(stderr, "test_dr_mvsnet: cannot write %s\n", out_name.c_str());
      }
    }
    delete out;
  }
  if (print) {
    printf("Performance:\n\tCallAsync     : %f ms\n\tReady         : %f ms\n\tGetResult     : %f ms\n", e1 / (1000.0 * repetitions),
           e2 / (1000.0 * repetitions), e3 / (1000.0 * repetitions));
    printf(correct ? "All looks good!\n" : "There has been an error. Do not use the model.\n");
  }
  return correct;
}

#endif  // DR_MVSNET_H
