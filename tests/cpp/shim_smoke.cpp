// Compiles the header-compatible libdr shim exactly the way TANDEM's callers use it
// (tandem_backend.cpp:147-200, dr_debug_example.cpp:78-162) and runs it when a GPU is present.
//   usage: shim_smoke <weights.tdmw> <sample.tdms>
#include <cstdio>
#include <cstring>
#include <vector>

#include "dr_fusion.h"
#include "dr_mvsnet.h"

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <weights.tdmw> <sample.tdms>\n", argv[0]); return 2; }
  DrMvsnet mvsnet(argv[1]);
  if (!test_dr_mvsnet(mvsnet, argv[2], true, 3)) return 1;

  // one fuse + render round with the depth map TANDEM would fuse (tandem_backend.cpp:166-177)
  FILE *f = fopen(argv[2], "rb");
  char magic[8]; int hdr[4]; float sc[3], K[9];
  if (fread(magic, 1, 8, f) != 8 || fread(hdr, 4, 4, f) != 4 || fread(sc, 4, 3, f) != 3 || fread(K, 4, 9, f) != 9) return 2;
  const int V = hdr[0], H = hdr[1], W = hdr[2], ref = hdr[3];
  std::vector<float> c2w((size_t) V * 16);
  std::vector<unsigned char> img((size_t) V * H * W * 3);
  if (fread(c2w.data(), 4, c2w.size(), f) != c2w.size() || fread(img.data(), 1, img.size(), f) != img.size()) return 2;
  fclose(f);
  std::vector<unsigned char *> bgrs(V);
  std::vector<float *> c2ws(V);
  for (int v = 0; v < V; v++) { bgrs[v] = img.data() + (size_t) v * H * W * 3; c2ws[v] = c2w.data() + 16 * v; }
  mvsnet.CallAsync(H, W, V, ref, bgrs.data(), K, c2ws.data(), sc[0], sc[1], sc[2]);
  DrMvsnetOutput *out = mvsnet.GetResult();

  // the boundary extensions: the same window from page-locked images, the result as a view of the engine's pinned block -- same maps
  {
    std::vector<unsigned char *> pb(V);
    for (int v = 0; v < V; v++) { pb[v] = DrMvsnet::AllocImage((size_t) H * W * 3); if (!pb[v]) return 3; memcpy(pb[v], bgrs[v], (size_t) H * W * 3); }
    mvsnet.CallAsync(H, W, V, ref, pb.data(), K, c2ws.data(), sc[0], sc[1], sc[2]);
    for (int v = 0; v < V; v++) memset(pb[v], 0, (size_t) H * W * 3);  // inputs are copied before CallAsync returns
    DrMvsnetOutput *view = mvsnet.GetResultView();
    const bool same = !memcmp(view->depth, out->depth, sizeof(float) * H * W) && !memcmp(view->confidence_dense, out->confidence_dense, sizeof(float) * H * W);
    printf("mvsnet: pinned upload + result view %s the copying path\n", same ? "equal" : "DIFFER FROM");
    delete view;
    for (int v = 0; v < V; v++) DrMvsnet::FreeImage(pb[v]);
    if (!same) return 4;
  }
  // round 6's extensions on the same object: GetResult() handing out views (SetResultViews) and the key-frame feature cache (SetFeatureCache) -- same maps again,
  // and the previous result's views stay readable while the next window is processed (TandemBackend reads result k - 1 after CallAsync(k), tandem_backend.cpp:147-166)
  {
    mvsnet.SetResultViews(true);
    mvsnet.SetFeatureCache(V + 2);
    mvsnet.CallAsync(H, W, V, ref, bgrs.data(), K, c2ws.data(), sc[0], sc[1], sc[2]);
    DrMvsnetOutput *a = mvsnet.GetResult();
    mvsnet.CallAsync(H, W, V, ref, bgrs.data(), K, c2ws.data(), sc[0], sc[1], sc[2]);  // answered by the cache; its result goes to the OTHER pinned block
    const bool kept = !memcmp(a->depth, out->depth, sizeof(float) * H * W) && !memcmp(a->depth_dense, out->depth_dense, sizeof(float) * H * W);
    DrMvsnetOutput *b = mvsnet.GetResult();
    const bool same = kept && !memcmp(b->depth, out->depth, sizeof(float) * H * W) && !memcmp(b->confidence, out->confidence, sizeof(float) * H * W);
    printf("mvsnet: result views + feature cache %s the copying path\n", same ? "equal" : "DIFFER FROM");
    delete a; delete b;
    mvsnet.SetResultViews(false);
    mvsnet.SetFeatureCache(0);
    if (!same) return 5;
  }

  DrFusionOptions o;
  o.voxel_size = 0.01f; o.num_buckets = 50000; o.bucket_size = 10; o.num_blocks = 100000; o.block_size = 8; o.max_sdf_weight = 64;
  o.truncation_distance = 0.04f; o.max_sensor_depth = 10.f; o.min_sensor_depth = 0.1f; o.num_render_streams = 1;
  o.fx = K[0]; o.fy = K[4]; o.cx = K[2]; o.cy = K[5]; o.height = H; o.width = W;
  DrFusion fusion(o);
  fusion.IntegrateScanAsync(bgrs[ref], out->depth, c2ws[ref]);
  fusion.RenderAsync({c2ws[ref]});
  std::vector<unsigned char *> rb;
  std::vector<float *> rd;
  fusion.GetRenderResult(rb, rd);
  size_t hit = 0;
  for (int i = 0; i < H * W; i++) hit += rd[0][i] > 0;
  printf("fusion: rendered %zu / %d pixels\n", hit, H * W);
  // mesh, the way TandemBackend asks for it (tandem_backend.cpp:194-200)
  float lower[3] = {-5, -5, -5}, upper[3] = {5, 5, 5};
  fusion.ExtractMeshAsync(lower, upper);
  fusion.GetMeshSync();
  printf("fusion: mesh with %zu vertices, first (%g %g %g)\n", fusion.dr_mesh_num, fusion.dr_mesh_vert[0], fusion.dr_mesh_vert[1], fusion.dr_mesh_vert[2]);
  DrMesh m = fusion.GetMesh(lower, upper);
  const bool mesh_ok = fusion.dr_mesh_num > 300 && fusion.dr_mesh_num % 3 == 0 && m.num == fusion.dr_mesh_num;
  free(m.vert); free(m.cols);
  delete out;
  return hit > (size_t) (H * W) / 4 && mesh_ok ? 0 : 1;
}
