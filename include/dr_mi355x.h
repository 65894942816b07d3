/* dr_mi355x.h -- C ABI of libdr_mi355x.so: the MI355X-native drop-in for TANDEM's libdr
 * operator API (DrMvsnet + DrFusion).  Plain pointers and sizes only; no torch / HIP types.
 *
 * Every entry point below replaces one member of the reference's C++ interface
 *   tandem/libdr/dr_mvsnet/src/dr_mvsnet/dr_mvsnet.h   (class DrMvsnet, DrMvsnetOutput)
 *   tandem/libdr/dr_fusion/src/dr_fusion/dr_fusion.h   (class DrFusion, DrFusionOptions, DrMesh)
 * The header-compatible C++ shim classes that forward to this ABI live in
 * tandem_amd/libdr/{dr_mvsnet.h,dr_fusion.h}; INTEGRATION.md shows how TANDEM links them.
 *
 * Error convention: the reference prints and exit()s on protocol violations and has no return
 * codes (dr_mvsnet.cpp:100-102,156-157; tsdf_volume.cu:520-524).  The C ABI returns an int
 * status instead (0 = ok) and keeps the message in dr_last_error(); the C++ shim reproduces the
 * reference's exit(EXIT_FAILURE) behaviour on non-zero status.
 */
#ifndef DR_MI355X_H
#define DR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DR_OK = 0,
  DR_ERR_ARG = 1,       /* bad argument (null pointer, aliasing views, unsupported size) */
  DR_ERR_PROTOCOL = 2,  /* call-order violation (reference: exit(EXIT_FAILURE)) */
  DR_ERR_DEVICE = 3,    /* HIP error / no GPU */
  DR_ERR_IO = 4,        /* weight blob missing or malformed */
  DR_ERR_CAPACITY = 5,  /* hash table / block pool exhausted (reference: KERNEL_ABORT trap, heap.cu:16) */
  DR_ERR_UNSUPPORTED = 6
};

/* Thread-local message of the last failing call in this thread ("" if none). */
const char *dr_last_error(void);
/* Library version string, e.g. "dr_mi355x 0.1 gfx950". */
const char *dr_version(void);

/* ------------------------------------------------------------------ DrMvsnet */
typedef struct drm_s drm_t;

/* DrMvsnet::DrMvsnet(char const* filename)                      dr_mvsnet.h:38, dr_mvsnet.cpp:20-26
 * `weights_path` names a TDMW blob (tandem_amd/weights.py) instead of a TorchScript archive. */
int drm_create(const char *weights_path, int device, drm_t **out);
/* DrMvsnet::~DrMvsnet(): waits for pending work, joins the worker  dr_mvsnet.cpp:28-38 */
void drm_destroy(drm_t *h);
/* DrMvsnet::CallAsync(...)                                      dr_mvsnet.h:43-53, dr_mvsnet.cpp:125-283
 * bgrs[v] -> H*W*3 u8 interleaved BGR; K9 row-major full-res intrinsics; c2ws[v] -> 16 floats
 * row-major camera-to-world.  Inputs are copied before return.  Blocks while the previous call
 * is still being processed ("Blocking for last input. Non-blocking for this input").
 * Returns DR_ERR_ARG if two bgr / c2w pointers alias (reference: exit, :153-160). */
int drm_call_async(drm_t *h, int height, int width, int view_num, int ref_index,
                   const uint8_t *const *bgrs, const float *K9, const float *const *c2ws,
                   float depth_min, float depth_max, float discard_percentage);
/* DrMvsnet::Ready()  non-blocking, 1 = no unprocessed input      dr_mvsnet.h:62, dr_mvsnet.cpp:54 */
int drm_ready(drm_t *h);
/* DrMvsnet::Wait()   blocking                                    dr_mvsnet.h:59, dr_mvsnet.cpp:109-119 */
int drm_wait(drm_t *h);
/* DrMvsnet::GetResult()  blocking; fills four caller-owned H*W float arrays (the members of
 * DrMvsnetOutput, dr_mvsnet.h:12-34).  A second call without a new drm_call_async returns
 * DR_ERR_PROTOCOL (reference: exit, dr_mvsnet.cpp:100-103). */
int drm_get_result(drm_t *h, float *depth, float *confidence, float *depth_dense, float *confidence_dense);
/* The operator boundary without its two host copies (no reference counterpart; DrMvsnet::GetResultView / AllocImage in the shim):
 * drm_get_result_view = drm_get_result returning POINTERS into the page-locked block the device wrote the four maps to (no 4.9 MB copy).
 * Two blocks alternate: the maps stay valid while the next drm_call_async is processed and are overwritten by the one after it; they
 * die with the engine and with a change of resolution.  Same protocol errors as drm_get_result.
 * drm_host_alloc / drm_host_free = page-locked host memory.  When EVERY image of a drm_call_async lives in page-locked memory (from
 * here, hipHostMalloc or hipHostRegister) the upload reads it in place instead of gathering the window into the engine's own staging
 * block first; the call still returns only after the copies have completed, so the caller may reuse the images at once. */
int drm_get_result_view(drm_t *h, const float **depth, const float **confidence, const float **depth_dense, const float **confidence_dense);
void *drm_host_alloc(size_t bytes);
void drm_host_free(void *p);

/* Extension (no reference counterpart): the KEY-FRAME FEATURE CACHE.  TANDEM's sliding window re-sends six of its seven images with every key frame
 * (ref: tandem/src/FullSystem/FullSystem.cpp:1162-1171 pushes frameHessians[i]->image_bgr for every active key frame), and FeatureNet
 * (ref: cva_mvsnet/models/module.py:496-531) is per image.  drm_set_feature_cache(h, capacity) makes the engine keep the three feature maps (and the
 * u8 image) of the last `capacity` images (0 = off, the default; capacity must exceed the window's view count): a window of which at most one image
 * is new runs FeatureNet on that one view.  A hit is found by a 128-bit key over a sample of the image and made exact on the device: every uploaded
 * image is compared byte for byte with the cached copy, and a difference drops the cache and repeats the window without it before a result leaves the
 * engine.  Results are bit-identical with the cache on and off.  Call it before the first window; changing it on a configured engine re-plans the
 * engine (upload the window again).  drm_feature_cache_stats: out[0] views answered by the cache, [1] views computed, [2] windows that took the batch
 * path, [3] key collisions caught by the device compare, [4] 1 if the single-view plan exists for the current window shape, [5] entries. */
int drm_set_feature_cache(drm_t *h, int capacity);
int drm_feature_cache_stats(drm_t *h, uint64_t out[6]);

/* --- device-resident / measurement / introspection hooks (no reference counterpart) --- */
/* Upload a window (same arguments as drm_call_async) and keep it resident in HBM. Synchronous. */
int drm_upload(drm_t *h, int height, int width, int view_num, int ref_index, const uint8_t *const *bgrs,
               const float *K9, const float *const *c2ws, float depth_min, float depth_max,
               float discard_percentage);
/* Enqueue `iters` complete forwards (pre-process .. edge filter) of the resident window on the
 * engine stream; returns after a stream synchronise.  ms_total (may be NULL) = hipEvent time. */
int drm_forward(drm_t *h, int iters, float *ms_total);
/* Autotune the convolution plan of the resident window: time the first max_candidates tile/pass configurations of each
 * layer's cost-model ranking on the device and keep the fastest.  before_ms / after_ms (may be NULL) = summed layer time.
 * Opt-in: a tuned engine's results differ from an untuned one's at the 1e-7 level (fp32 accumulation order). */
int drm_autotune(drm_t *h, int max_candidates, float *before_ms, float *after_ms);
/* --- view sharding (BASELINE configs[2]): one source view (or a few) per GPU, the host sum-reduces the
 * partial cost volumes over the ranks (RCCL).  No reference counterpart: the reference has no inference-time
 * collective.  Protocol per rank: drm_set_view_shard(total source views of the window) -> drm_upload(sub-window =
 * [reference view, this rank's source views], view_num may be 1) -> for p in 0..2 { drm_forward_phase(p);
 * all-reduce(sum) the tensor "volume<p+1>" in place } -> drm_forward_phase(3) -> drm_download.  With the shard set,
 * each rank's cost volume is sum_over_local_views((gate + 1) * (warp - ref)^2) / total, so the reduced volume is the
 * unsharded one up to fp32 summation order.  View-aggregation models only (module.py:1097-1108). */
int drm_set_view_shard(drm_t *h, int nsrc_total); /* 0 = off (default) */
/* Enqueue phase 0..3 of the resident window and wait for it: 0 = pre-process, FeatureNet, cost volume 1;
 * p = 1,2: regularise + regress stage p, cost volume p+1; 3 = regularise + regress stage 3, edge filter. */
int drm_forward_phase(drm_t *h, int phase);
/* The same collective INSIDE the engine, for hosts without a collective library of their own (TANDEM's C++ back-end):
 * rank 0 draws an id (drm_comm_unique_id) and hands the 128 bytes to every rank by any means; each rank calls
 * drm_comm_init on its engine (RCCL communicator on the engine's device, librccl bound with dlopen).  From then on a
 * sharded window (drm_set_view_shard > 0) needs no host step between phases: drm_call_async / drm_forward enqueue, on
 * the engine's stream, each cost-volume kernel followed by ncclReduce(sum) of that volume to rank 0; rank 0 alone
 * regularises and regresses the stage and ncclBroadcast()s its depth map (stage 3: depth and confidence) -- what the next
 * stage's hypotheses hang on -- back to every rank, so all ranks end with the same four output maps.  ((n-1)/n x 354 MB per
 * depth map over xGMI instead of an all-reduce's 2(n-1)/n; DR_SHARD_ALLREDUCE=1 selects the all-reduce
 * form, in which every rank regularises redundantly.)  FeatureNet's stage-2/3 heads overlap the first reduce on the side
 * stream.  Use at most as many ranks as the window has source views.  drm_forward_phase keeps the host-reduced protocol. */
int drm_comm_available(void);                /* DR_OK when RCCL (librccl.so.1, or $DR_RCCL_LIB) can be bound in this process */
int drm_comm_unique_id(uint8_t id[128]);
int drm_comm_init(drm_t *h, int rank, int world, const uint8_t id[128]);
int drm_comm_destroy(drm_t *h);
/* Ranks of the engine's communicator as RCCL reports them (ncclCommCount): 0 = no communicator, -1 = the bound library lacks the call. */
int drm_comm_count(drm_t *h, int *nranks);
/* Device pointer and element count of a named internal tensor ("volume1".."volume3", "feat1", "depth2", ...) AS IT LIES IN
 * MEMORY: "feat1".."feat3" carry a one-pixel zero border in H and W (the cost-volume kernels read them that way), so for
 * those the pointer is the padded base and *nfloats = V * (H + 2) * (W + 2) * C; "volume1" (32 channels) is stored as two consecutive
 * (D, H, W, 16) halves (channels 0-15 | 16-31).  drm_get_tensor returns the logical (D, H, W, C) block in every case. */
int drm_device_tensor(drm_t *h, const char *name, void **dptr, size_t *nfloats);
/* Copy the last forward's stage-3 outputs to host (same four arrays as drm_get_result). */
int drm_download(drm_t *h, float *depth, float *confidence, float *depth_dense, float *confidence_dense);
/* Unfiltered depth / confidence of stage 1..3 (h_s*w_s floats each). */
int drm_get_stage_output(drm_t *h, int stage, float *depth, float *confidence);
/* Named internal tensor of the last forward, copied to host in its device layout (channels-last).
 * n_max = capacity of out in floats; *n = element count; dims[4] = {D|V, H, W, C}. */
int drm_get_tensor(drm_t *h, const char *name, float *out, size_t n_max, size_t *n, int dims[4]);
/* Per-kernel timing of one forward (hipEvents around every launch). names: '\n'-separated. */
int drm_profile(drm_t *h, char *names, size_t names_cap, float *ms, int cap, int *count);
/* Algorithmic work of one forward of the resident window (SURVEY.md 8d "layer-boundary" model). */
int drm_work(drm_t *h, double *flops, double *bytes);

/* Kernel unit-test hook: run one convolution layer through the engine's packer + MFMA kernel.
 * in  : (D,H,W,Cin) channels-last, host.  weight: torch layout (Cout,Cin,kd,kh,kw) for conv,
 * (Cin,Cout,kd,kh,kw) for transposed.  scale/bias: per-Cout affine applied before ReLU (may be NULL).
 * add : optional residual, output-shaped (or (D,H/2,W/2,Cout) when add_up2).  out: (Do,Ho,Wo,Cout). */
int drm_debug_conv(int device, const float *in, int D, int H, int W, int Cin, const float *weight, int Cout,
                   int kd, int kh, int kw, int sd, int sh, int sw, int transposed, const float *scale,
                   const float *bias, int relu, const float *add, int add_up2, float *out, int out_dims[3]);

/* Kernel unit-test hook for the fused tail of CostRegNet (conv11 + prob in one launch, csrc/tail_kernels.h; cva_mvsnet/models/module.py:571-575,598-599):
 * x (D/2, h/2, w/2, 16) and skip (D, h, w, 8) channels-last, host; w_deconv (16, 8, 3, 3, 3) and w_prob (1, 8, 3, 3, 3) in torch layout; scale8 / bias8 =
 * the folded BatchNorm of conv11; qy / zchunk: tile and depth-chunk overrides (0: chosen by size); form: 1 = the transposed convolution on the matrix pipe
 * (k_tail_m), 0 = on the vector pipe (k_tail).  out: (D, h, w) logits. */
int drm_debug_tail(int device, const float *x, const float *skip, const float *w_deconv, const float *scale8, const float *bias8, const float *w_prob, int D,
                   int h, int w, int qy, int zchunk, int form, float *out);

/* ------------------------------------------------------------------ DrFusion */
/* struct DrFusionOptions                                         dr_fusion.h:18-36 (same field order) */
typedef struct {
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
  float fx, fy, cx, cy;
  int height, width;
} drf_options_t;

typedef struct drf_s drf_t;

/* DrFusion::DrFusion(DrFusionOptions const&)                     dr_fusion.h:46, dr_fusion.cpp:8-38 */
int drf_create(const drf_options_t *opt, int device, drf_t **out);
/* DrFusion::~DrFusion()                                          dr_fusion.cpp:41-46 */
void drf_destroy(drf_t *h);
/* DrFusion::IntegrateScanAsync(bgr, depth, pose)                 dr_fusion.h:50, tsdf_volume.cu:515-598
 * H*W*3 u8 BGR, H*W f32 metres (0 = invalid), 16-float row-major cam-to-world; inputs are copied
 * to pinned memory before return.  Wrong call order -> DR_ERR_PROTOCOL (reference: exit). */
int drf_integrate_scan_async(drf_t *h, const uint8_t *bgr, const float *depth, const float *pose16);
/* DrFusion::RenderAsync(std::vector<float const*>)               dr_fusion.h:52, tsdf_volume.cu:634-700
 * n must equal num_render_streams (may be 0). */
int drf_render_async(drf_t *h, const float *const *poses16, int n);
/* DrFusion::GetRenderResult(bgr, depth)                          dr_fusion.h:54, tsdf_volume.cu:702-737
 * Fills n library-owned pinned pointers, valid until the next drf_get_render_result. */
int drf_get_render_result(drf_t *h, uint8_t **bgr, float **depth, int n);
/* DrFusion::ExtractMeshAsync(lower, upper)                        dr_fusion.h:60, tsdf_volume.cu:759-779,
 * marching_cubes/mesh_extractor.cu:136-281.  Marching cubes over the lattice lower + g * voxel_size; legal where
 * IntegrateScanAsync is (after GetRenderResult); at most one extraction may be pending. */
int drf_extract_mesh_async(drf_t *h, const float lower[3], const float upper[3]);
/* DrFusion::GetMeshSync()                                         dr_fusion.h:61, tsdf_volume.cu:781-838
 * Waits for the pending extraction and copies it out: vert / cols hold num_max vertices (3 floats each);
 * *num = 3 * triangles; vert[9t + 3k + 0..2] = position of vertex k of triangle t, cols[...] = its colour as RGB in
 * [0, 1].  DR_ERR_CAPACITY if 3 * triangles > num_max (the mesh stays pending) or above 20 M triangles. */
int drf_get_mesh_sync(drf_t *h, size_t num_max, size_t *num, float *vert, float *cols);
/* Size of the pending mesh (waits for it, does not consume it) -- lets a binding allocate exactly; no reference
 * counterpart (the reference preallocates 2 x 720 MB, dr_fusion.cpp:36-37). */
int drf_mesh_num_triangles(drf_t *h, size_t *ntri);
/* DrFusion::SaveMeshToFile(filename, lower, upper)                dr_fusion.h:56, dr_fusion.cpp:74-93, mesh.cu:24-66
 * Synchronous extraction written as Wavefront OBJ: "v x y z r g b" per vertex, "f i i+1 i+2" per triangle. */
int drf_save_mesh(drf_t *h, const char *filename, const float lower[3], const float upper[3]);
/* Device pointers of render stream `stream`'s result (H*W*3 u8 BGR, H*W f32 depth), valid from drf_get_render_result
 * until the next drf_render_async; no reference counterpart.  Feeds drt_append_dense_reference(on_device = 1). */
int drf_get_render_device(drf_t *h, int stream, const uint8_t **d_bgr, const float **d_depth);
/* DrFusion::Synchronize()                                        dr_fusion.h:64 */
int drf_synchronize(drf_t *h);

/* --- introspection / measurement hooks (no reference counterpart) --- */
/* Counters: [0] allocated blocks, [1] voxels updated by the last scan (band + carve),
 * [2] voxels updated in total, [3] round-trip voxel mismatches (must stay 0). */
int drf_stats(drf_t *h, uint64_t out[4]);
/* Blocks the integration kernel has visited since creation (visible blocks of every scan; each is one 4 KB read whether or not a voxel of it was
 * updated): with out[2] of drf_stats the kernel's exact HBM bytes are 4096 * visited + 8 * updated. */
int drf_visited_blocks(drf_t *h, uint64_t *total);
/* Canonical dump for bit-exact comparison: coords[3*i..] block coordinates, voxels[4096*i..] the
 * 512 8-byte voxels {f32 sdf, u8 b,g,r, u8 weight} of block i in index order x*64+y*8+z. */
int drf_export_blocks(drf_t *h, int max_blocks, int32_t *coords, uint8_t *voxels, int *n);
/* The engine divides by voxel_size, fx and fy with a 3-instruction exact sequence (reciprocal + FMA correction) after
 * checking it against IEEE division for all 2^32 dividends at construction: *enabled = 1 if every check passed (else the
 * kernels use IEEE division), *mismatches = number of disagreeing dividends found. */
int drf_fast_div_status(drf_t *h, int *enabled, uint64_t *mismatches);
/* Test hook: out[i] = Combine(a[i], b[i], max_weight) evaluated by the integration kernel's own device function
 * (voxel.h:21-50), n 8-byte voxels {f32 sdf, u8 b,g,r, u8 weight} each -- lets a test sweep every colour/weight case. */
int drf_test_combine(drf_t *h, size_t n, const uint8_t *a, const uint8_t *b, int max_weight, uint8_t *out);
/* Integrate scans already resident in HBM (bench path): d_* are device pointers. */
int drf_integrate_device(drf_t *h, const void *d_bgr, const void *d_depth, const float *pose16);
/* Device-side scratch allocation helpers so a host without a HIP runtime binding can stage inputs. */
int dr_device_alloc(int device, size_t bytes, void **dptr);
int dr_device_free(void *dptr);
int dr_memcpy_h2d(void *dptr, const void *src, size_t bytes);
int dr_memcpy_d2h(void *dst, const void *dptr, size_t bytes);
int dr_memcpy_d2d(void *dst, const void *src, size_t bytes); /* returns after the copy has completed */
/* Time `iters` back-to-back integrations of `nscans` resident scans with hipEvents on the
 * integration stream.  ms / kernel_ms (integrate kernel only) may be NULL. */
int drf_bench_integrate(drf_t *h, const void *d_bgr, const void *d_depth, const float *poses16, int nscans,
                        float *ms, float *kernel_ms);

/* BASELINE configs[3] loop (dr_debug_example.cpp:78-162) over nframes frames resident in HBM (d_bgr: nframes*H*W*3 u8,
 * d_depth: nframes*H*W f32, poses16: nframes*16): allocate + integrate per frame and, if render != 0, one ray-cast per
 * render stream from the frame's pose with the D2H of its result.  ms[0] whole run (hipEvents), ms[1..4] sums of the
 * allocate / integrate / ray-cast / D2H intervals, ms[5] host wall clock. */
int drf_bench_sequence(drf_t *h, const void *d_bgr, const void *d_depth, const float *poses16, int nframes, int render, float ms[6]);
/* Test hook: host pointers (page-locked, library-owned) of the last (back = 0) / second-to-last (back = 1) render of `stream` written by
 * drf_bench_sequence.  In that loop the allocation of scan k + 1 runs beside the ray-cast of scan k (a device-resident sequence is the only
 * caller that reaches this overlap: through the operator API GetRenderResult(k) returns before IntegrateScanAsync(k + 1) is called). */
int drf_bench_render_host(drf_t *h, int stream, int back, const uint8_t **bgr, const float **depth);

/* ======================================================================================================
 * DrCoarseTracker -- the dense coarse tracker operator.  Replaces
 *   tandem/libdr/cuda_coarse_tracker/include/public/cuda_coarse_tracker.h   (class CudaCoarseTracker)
 * member for member (Eigen arguments become plain arrays: matrices row-major, double where the reference is), plus
 * the dense-depth hand-off of CoarseTracker::setCoarseTrackingRef (src/FullSystem/CoarseTracker.cpp:655-725).
 * ====================================================================================================== */
typedef struct drt_s drt_t;
/* CudaCoarseTracker(w, h, setting_huberTH, setting_coarseCutoffTH)            cuda_coarse_tracker.h:11, .cpp:63-69 */
int drt_create(int w, int h, float setting_huberTH, float setting_coarseCutoffTH, int device, drt_t **out);
void drt_destroy(drt_t *t);                                                   /* ~CudaCoarseTracker / free(), .cpp:142-199 */
/* setK(w, h, fx, fy, cx, cy): w, h must equal the constructor's               cuda_coarse_tracker.h:13, .cpp:358-372 */
int drt_set_k(drt_t *t, int w, int h, float fx, float fy, float cx, float cy);
/* init(n_max = 0 -> w*h); a second call is DR_ERR_PROTOCOL                      cuda_coarse_tracker.h:17, .cpp:101-140 */
int drt_init(drt_t *t, int n_max);
/* setReference(n, pc_u, pc_v, pc_idepth, pc_color, ref_exposure, ref_aff_g2l)  cuda_coarse_tracker.h:21, .cpp:71-89 */
int drt_set_reference(drt_t *t, int n, const float *pc_u, const float *pc_v, const float *pc_idepth, const float *pc_color,
                      float ref_exposure, const double ref_aff_g2l[2]);
/* setNew(dInew): 3*w*h floats, (I, dx, dy) interleaved per pixel                cuda_coarse_tracker.h:23, .cpp:91-94 */
int drt_set_new(drt_t *t, const float *dInew);
/* Vec6 calcRes(refToNew 4x4, new_exposure, aff_g2l, cutoffTH)                   cuda_coarse_tracker.h:25, .cpp:217-288
 * refToNew row-major.  out6 = the reference's Vec6 (E, numTermsInE, shiftT/num, 0, shiftRT/num, saturated/numTermsInE);
 * sums7 (may be NULL) = the 7 raw sums in the order of cuda_coarse_tracker_private.h:8-15. */
int drt_calc_res(drt_t *t, const double refToNew[16], float new_exposure, const double aff_g2l[2], float cutoffTH, double out6[6],
                 double sums7[7]);
/* calcG(H_out 8x8, b_out 8, new_exposure, aff_g2l)                              cuda_coarse_tracker.h:27, .cpp:290-356
 * Uses the warped buffers of the last calcRes.  H row-major, scaled as the reference; raw45 (may be NULL) = the 45
 * unscaled upper-triangular sums. */
int drt_calc_g(drt_t *t, double H_out[64], double b_out[8], float new_exposure, const double aff_g2l[2], double raw45[45]);
/* The dense-depth branch of CoarseTracker::setCoarseTrackingRef (CoarseTracker.cpp:655-725), on the device: forward-warp
 * `depth` (w*h, metres, <= 0 invalid; sampled every `step` pixels) with p' = KRKi * (x*d, y*d, d) + Kt into the tracker's
 * reference frame (z-buffer minimum), then append every pixel of rows/cols [2, size-2) that received a depth -- and has
 * idepth0 <= 0 unless dense_only -- as (x, y, 1/depth, dIp0[3*i]) after the current points, row-major.  KRKi (row-major)
 * and Kt are the caller's float products K*R*Ki and K*t (:672-673).  on_device != 0: depth / idepth0 / dIp0 are device
 * pointers (e.g. a DrFusion render left in HBM).  *n_out = new point count.  DR_ERR_CAPACITY above n_max. */
int drt_append_dense_reference(drt_t *t, const float *depth, const float KRKi[9], const float Kt[3], int step, int dense_only,
                               const float *idepth0, const float *dIp0, int on_device, int *n_out);
/* synchronize / startTiming / endTimingMilliseconds                              cuda_coarse_tracker.h:30-34 */
int drt_synchronize(drt_t *t);
int drt_start_timing(drt_t *t);
int drt_end_timing_ms(drt_t *t, float *ms);
/* --- introspection hooks (no reference counterpart) --- */
int drt_get_points(drt_t *t, float *pc_u, float *pc_v, float *pc_idepth, float *pc_color, int cap, int *n); /* arrays may be NULL */
int drt_get_warped(drt_t *t, int which, float *out, int cap); /* which: 0 u, 1 v, 2 dx, 3 dy, 4 idepth, 5 residual, 6 weight */
int drt_get_zbuffer(drt_t *t, float *out);                    /* w*h projected depths of the last append, -1 = empty */

#ifdef __cplusplus
}
#endif
#endif /* DR_MI355X_H */
