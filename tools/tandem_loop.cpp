// tandem_loop -- BASELINE configs[4] stand-in: TANDEM's back-end loop on one GPU through the header-compatible libdr shim.
// Drives DrMvsnet and DrFusion exactly in the order TandemBackendImpl does (ref:tandem/src/tandem/tandem_backend.cpp):
//   caller thread, per keyframe k   (:220-283)  GetResult(k-1)                               [blocking]
//   worker,       per keyframe k   (:137-217)  CallAsync(k)  ->  IntegrateScanAsync(k-1, depth of k-1)
//                                               ->  RenderAsync({pose k})  ->  GetRenderResult  ->  memcpy of the rendered depth
//                                               [->  ExtractMeshAsync + GetMeshSync every mesh_freq-th call]
// so the depth network of keyframe k runs while keyframe k-1 is fused and ray-cast.  The DSO front-end is replaced by a
// stored keyframe window (a TDMS sample, tools/export_fixture.py) whose poses are moved rigidly from keyframe to keyframe
// (the network result is invariant, the map keeps growing).  Prints ONE JSON line: keyframes/s and the mean time of each call.
//   usage: tandem_loop <weights.tdmw> <window.tdms> <keyframes> [voxel_size=0.01] [mesh_freq=0] [dense_tracking=1]
// TANDEM_LOOP_SERIAL=1 (measurement aid, not TANDEM's order) waits for the depth network of keyframe k before fusing k-1:
// the un-overlapped sum on the same map, to separate GPU contention from host gaps.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dr_fusion.h"
#include "dr_mvsnet.h"

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <weights.tdmw> <window.tdms> <keyframes> [voxel_size] [mesh_freq] [dense_tracking]\n", argv[0]); return 2; }
  const int n_kf = atoi(argv[3]);
  const float voxel = argc > 4 ? (float) atof(argv[4]) : 0.01f;   // FullSystem.cpp:260
  const int mesh_freq = argc > 5 ? atoi(argv[5]) : 0;
  const bool dense_tracking = argc > 6 ? atoi(argv[6]) != 0 : true;
  FILE *f = fopen(argv[2], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
  char magic[8]; int hdr[4]; float sc[3], K[9];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "TDMS0001", 8) || fread(hdr, 4, 4, f) != 4 || fread(sc, 4, 3, f) != 3 || fread(K, 4, 9, f) != 9) return 2;
  const int V = hdr[0], H = hdr[1], W = hdr[2], ref = hdr[3];
  const size_t npx = (size_t) H * W;
  std::vector<float> c2w0((size_t) V * 16);
  std::vector<unsigned char> img((size_t) V * npx * 3);
  if (fread(c2w0.data(), 4, c2w0.size(), f) != c2w0.size() || fread(img.data(), 1, img.size(), f) != img.size()) return 2;
  fclose(f);

  DrMvsnet mvsnet(argv[1]);
  DrFusionOptions o;   // FullSystem::initDr, FullSystem.cpp:259-276 (TANDEM's values, voxel size on the command line)
  o.voxel_size = voxel; o.num_buckets = 1000000; o.bucket_size = 10; o.num_blocks = 1000000; o.block_size = 8; o.max_sdf_weight = 64;
  o.truncation_distance = 4 * voxel; o.max_sensor_depth = 10.f; o.min_sensor_depth = 0.1f; o.num_render_streams = dense_tracking ? 1 : 0;
  o.fx = K[0]; o.fy = K[4]; o.cx = K[2]; o.cy = K[5]; o.height = H; o.width = W;
  DrFusion fusion(o);

  std::vector<unsigned char *> bgrs(V);
  for (int v = 0; v < V; v++) bgrs[v] = img.data() + (size_t) v * npx * 3;
  std::vector<float> c2w_cur(c2w0), c2w_prev(c2w0), tracker_depth(npx);
  std::vector<float *> cur_ptr(V), prev_ptr(V);
  for (int v = 0; v < V; v++) { cur_ptr[v] = c2w_cur.data() + 16 * v; prev_ptr[v] = c2w_prev.data() + 16 * v; }
  float lower[3] = {-5, -5, -5}, upper[3] = {5, 5, 5};   // tandem_backend.cpp:80-81

  const bool serial = getenv("TANDEM_LOOP_SERIAL") != nullptr;
  double t_get = 0, t_call = 0, t_int = 0, t_render = 0, t_getrender = 0, t_mesh = 0;
  int meshes = 0;
  size_t rendered = 0;
  DrMvsnetOutput *prev = nullptr;
  const int warm = 3;
  Clock::time_point t_begin = Clock::now();
  for (int k = 0; k < n_kf + warm; k++) {
    if (k == warm) { t_begin = Clock::now(); t_get = t_call = t_int = t_render = t_getrender = t_mesh = 0; meshes = 0; }
    // keyframe k's window: the stored one, moved rigidly (x advances 4 cm per keyframe on a slow arc)
    const float ang = 0.01f * k, cs = std::cos(ang), sn = std::sin(ang), tx = 0.04f * k;
    for (int v = 0; v < V; v++) {
      const float *a = c2w0.data() + 16 * v; float *b = c2w_cur.data() + 16 * v;
      for (int c = 0; c < 4; c++) {       // b = S * a,  S = rot_y(ang) with translation (tx, 0, 0)
        b[c] = cs * a[c] + sn * a[8 + c] + (c == 3 ? tx : 0.f);
        b[4 + c] = a[4 + c];
        b[8 + c] = -sn * a[c] + cs * a[8 + c];
        b[12 + c] = a[12 + c];
      }
    }
    // --- caller: TandemBackendImpl::CallAsync step 1 (:264-268)
    Clock::time_point t = Clock::now();
    DrMvsnetOutput *out_prev = nullptr;
    if (k > 0) out_prev = mvsnet.GetResult();
    t_get += ms_since(t);
    // --- worker: CallSequential 3.5 (:147-160)
    t = Clock::now();
    mvsnet.CallAsync(H, W, V, ref, bgrs.data(), K, cur_ptr.data(), sc[0], sc[1], sc[2], false);
    if (serial) mvsnet.Wait();
    t_call += ms_since(t);
    if (out_prev) {
      t = Clock::now();
      fusion.IntegrateScanAsync(bgrs[ref], out_prev->depth, prev_ptr[ref]);                      // (:166)
      t_int += ms_since(t);
      t = Clock::now();
      std::vector<float const *> poses;
      if (dense_tracking) poses.push_back(cur_ptr[ref]);                                          // (:171-172)
      fusion.RenderAsync(poses);
      t_render += ms_since(t);
      t = Clock::now();
      std::vector<unsigned char *> rb;
      std::vector<float *> rd;
      fusion.GetRenderResult(rb, rd);                                                             // (:175-177)
      if (dense_tracking) {
        memcpy(tracker_depth.data(), rd[0], sizeof(float) * npx);                                 // (:179)
        rendered = 0;
        for (size_t i = 0; i < npx; i += 97) rendered += tracker_depth[i] > 0;
      }
      t_getrender += ms_since(t);
      if (mesh_freq > 0 && (k % mesh_freq) == 0) {                                                // (:194-200)
        t = Clock::now();
        fusion.ExtractMeshAsync(lower, upper);
        fusion.GetMeshSync();
        t_mesh += ms_since(t);
        meshes++;
      }
    }
    delete prev;
    prev = out_prev;
    c2w_prev = c2w_cur;
  }
  DrMvsnetOutput *last = mvsnet.GetResult();
  fusion.Synchronize();
  const double total = ms_since(t_begin);
  size_t valid = 0;
  for (size_t i = 0; i < npx; i++) valid += last->depth[i] > 0;
  delete last; delete prev;
  printf("{\"keyframes\": %d, \"keyframes_per_s\": %.3f, \"ms_per_keyframe\": %.4f, \"height\": %d, \"width\": %d, \"views\": %d, "
         "\"voxel_size\": %g, \"serial\": %d, \"dense_tracking\": %d, \"mesh_every\": %d, \"meshes\": %d, "
         "\"mean_ms\": {\"GetResult_wait\": %.4f, \"CallAsync\": %.4f, \"IntegrateScanAsync\": %.4f, \"RenderAsync\": %.4f, "
         "\"GetRenderResult\": %.4f, \"mesh\": %.4f}, \"valid_depth_fraction\": %.4f, \"rendered_sample\": %zu}\n",
         n_kf, 1e3 * n_kf / total, total / n_kf, H, W, V, voxel, (int) serial, (int) dense_tracking, mesh_freq, meshes, t_get / n_kf, t_call / n_kf,
         t_int / n_kf, t_render / n_kf, t_getrender / n_kf, meshes ? t_mesh / meshes : 0.0, (double) valid / npx, rendered);
  return 0;
}
