// march_emul.hip -- host emulation of k_conv_m (tandem_amd/csrc/conv_march.h), run by tests/test_march_plan.py on the CPU.
//
// There is no GPU where the CPU suite runs, so this program executes the kernel's DATA FLOW on the host: plan_conv
// builds the real launch (host-only arena), then for every workgroup the producer's DMA pieces fill a ring image with
// the kernel's own index helpers (march_piece_entry, march_plane_offset, conv_a_slot), the consumer lanes gather their
// MFMA operands with the kernel's own addressing (march_bpos, conv_a_unit, tap table, weight sections) and a scalar
// model of v_mfma_f32_16x16x4_f32 accumulates them; the epilogue uses march_out_index.  The result is compared with a
// direct convolution.  The ring protocol is checked on the way (a load may only overwrite a slot every reader has
// released; the consumer never waits for a load the producer could not have issued).  What this does NOT cover is the
// timing side of the protocol on real hardware -- that is the GPU tests' job.
#include <cmath>
#include <cstdio>
#include <random>

#include "../../tandem_amd/csrc/conv_mfma.h"

namespace dr {
std::string &last_error_slot() {
  static std::string s;
  return s;
}
}  // namespace dr
using namespace dr;

struct F4 { float v[4]; };

template <int CI>
static bool emulate(const ConvLaunch &c, const float *in, float *out, const float *add) {
  const ConvArgs &a = c.args;
  const MarchArgs &m = c.march;
  const int NUP = c.nup, CT = c.ct, PT = c.pt, NS = m.geo.KZ * m.geo.NPI, TPC = 16 / CI;
  for (unsigned bz = 0; bz < c.grid.z; ++bz) {
    const int ct0 = (int)bz * CT;
    for (unsigned b = 0; b < c.grid.x; ++b) {
      const int nwg = (int)c.grid.x, id = (int)(b & 7u) * (nwg >> 3) + (int)(b >> 3);
      int s0, s1;
      march_range(m.steps, id, nwg, s0, s1);
      if (s0 >= s1) continue;
      std::vector<F4> ring((size_t)m.R * m.PS), wl((size_t)NS * m.wsec);
      int loaded = 0, released = 0, L = 0;
      for (int po = 0; po < m.NPO; ++po) {
        if (released != L) { printf("emul: pass %d starts with %d of %d loads released\n", po, released, L); return false; }
        for (int e = 0; e < NS * NUP * CT; ++e)
          for (int lane = 0; lane < 64; ++lane) {
            const float4 w = a.wpk[march_weight_src(a, m, po, e, NUP, CT, ct0) + lane];
            wl[(size_t)e * 64 + lane] = F4{{w.x, w.y, w.z, w.w}};
          }
        // the producer's load list of this pass, in issue order
        struct Load { int zc, plane, pass, iy0, ix0; };
        std::vector<Load> loads;
        for (int s = s0; s < s1;) {
          const MarchSeg sg = march_segment(m.geo, s, s1);
          int zc, py0, px0;
          march_tile_origin(a, m, sg.col, zc, py0, px0);
          for (int l = 0; l < sg.nl; ++l) {
            int plane, pi;
            march_load_plane(m.geo, sg, l, plane, pi);
            loads.push_back({zc, plane, po * m.geo.NPI + pi, m.rm ? 0 : py0 * a.sy - a.py, px0 * a.sx - a.px});
          }
          s += sg.zb - sg.za;
        }
        const int Lpass = L;
        auto do_load = [&](int i) -> bool {
          if (i >= m.R && released < i - m.R + 1) { printf("emul: load %d would overwrite a slot still in use (released %d, R %d)\n", i, released, m.R); return false; }
          const Load &ld = loads[i - Lpass];
          const long long pofs = march_plane_offset(a, m, ld.zc, ld.plane, ld.iy0, ld.ix0, ld.pass, CI);
          for (int pw = 0; pw < kMarchProducers; ++pw)
            for (int it = 0; it < m.nit; ++it)
              for (int lane = 0; lane < 64; ++lane) {
                int rel; unsigned yx;
                march_piece_entry<CI>(a, m, pw, it, lane, rel, yx);
                F4 v{{0.f, 0.f, 0.f, 0.f}};
                if (march_piece_inside(a, m, yx, ld.iy0, ld.ix0)) for (int k = 0; k < 4; ++k) v.v[k] = in[pofs + rel + k];
                const size_t dst = (size_t)(i % m.R) * m.PS + (size_t)(it * kMarchProducers + pw) * 64 + lane;
                if (dst >= ring.size() || (it * kMarchProducers + pw) * 64 + lane >= m.PS) { printf("emul: DMA piece outside its ring slot\n"); return false; }
                ring[dst] = v;
              }
          return true;
        };
        const int raw = m.NPO == 1 ? 0 : (po == 0 ? 1 : 2);
        for (int s = s0; s < s1;) {
          const MarchSeg sg = march_segment(m.geo, s, s1);
          int zc, py0, px0;
          march_tile_origin(a, m, sg.col, zc, py0, px0);
          MarchCursor cur;  // the kernel's division-free bookkeeping, checked against march_section_load
          cur.begin(m.geo, sg, L, m.R);
          for (int z = sg.za; z < sg.zb; cur.next_step(m.geo, m.R), ++z) {
            const bool W = m.wino != 0;  // march_consumer_w: four accumulator planes (one per Winograd point), chunks of x taps with three raw kernel rows each
            const int NRP = NUP / 3;
            std::vector<float> acc((size_t)(W ? 4 : 1) * m.ncw * CT * PT * 64 * 4, 0.f);
            const size_t pstride = (size_t)m.ncw * CT * PT * 64 * 4;
            for (int sec = 0; sec < NS; ++sec) {
              const int rel = march_section_load(m.geo, sg, z, sec);
              if (rel < 0) continue;
              const int idx = L + rel;
              if (cur.rel0 + sec != rel || (cur.slot0 + sec) % m.R != idx % m.R) { printf("emul: cursor disagrees with march_section_load (step %d section %d)\n", z, sec); return false; }
              if (idx - Lpass >= (int)loads.size()) { printf("emul: section reads load %d beyond the producer's list\n", idx); return false; }
              while (loaded <= idx) { if (!do_load(loaded)) return false; ++loaded; }
              if (idx < released) { printf("emul: section reads load %d after releasing it\n", idx); return false; }
              if (W) {
                for (int wave = 0; wave < m.ncw; ++wave)
                  for (int r = 0; r < NRP; ++r)
                    for (int ct = 0; ct < CT; ++ct)
                      for (int pt = 0; pt < PT; ++pt) {
                        F4 u[4][64], v[4][64];
                        for (int lane = 0; lane < 64; ++lane) {
                          const int j = lane & 15, g = lane >> 4, sub = (4 * g) / CI, c4 = ((4 * g) % CI) / 4;
                          F4 gk[3], d[4];
                          for (int k = 0; k < 3; ++k) gk[k] = wl[(size_t)sec * m.wsec + (size_t)((r * 3 + k) * CT + ct) * 64 + lane];
                          for (int q = 0; q < 4; ++q) {
                            const int slot = conv_a_unit<CI>(march_bpos(a, wave, pt, PT, j) + m.tap2d[r * TPC + sub] + q * a.TXI, c4);
                            if (slot < 0 || slot >= m.PS) { printf("emul: operand slot %d outside the plane (%d slots)\n", slot, m.PS); return false; }
                            d[q] = ring[(size_t)(idx % m.R) * m.PS + slot];
                          }
                          for (int e = 0; e < 4; ++e) {
                            u[0][lane].v[e] = gk[0].v[e]; u[3][lane].v[e] = gk[2].v[e];
                            u[1][lane].v[e] = march_w_u1(gk[0].v[e], gk[1].v[e], gk[2].v[e]);
                            u[2][lane].v[e] = march_w_u2(gk[0].v[e], gk[1].v[e], gk[2].v[e]);
                            v[0][lane].v[e] = d[0].v[e] - d[2].v[e]; v[1][lane].v[e] = d[1].v[e] + d[2].v[e];
                            v[2][lane].v[e] = d[2].v[e] - d[1].v[e]; v[3][lane].v[e] = d[1].v[e] - d[3].v[e];
                          }
                        }
                        for (int pp = 0; pp < 4; ++pp)
                          for (int col = 0; col < 16; ++col)
                            for (int row = 0; row < 16; ++row) {
                              float &dd = acc[pp * pstride + ((((size_t)wave * CT + ct) * PT + pt) * 64 + ((row >> 2) * 16 + col)) * 4 + (row & 3)];
                              for (int sidx = 0; sidx < 4; ++sidx)
                                for (int g = 0; g < 4; ++g) dd = std::fmaf(u[pp][g * 16 + row].v[sidx], v[pp][g * 16 + col].v[sidx], dd);
                            }
                      }
              } else
              for (int wave = 0; wave < m.ncw; ++wave)
                for (int u = 0; u < NUP; ++u)
                  for (int ct = 0; ct < CT; ++ct)
                    for (int pt = 0; pt < PT; ++pt) {
                      F4 av[64], bv[64];
                      for (int lane = 0; lane < 64; ++lane) {
                        const int j = lane & 15, g = lane >> 4, sub = (4 * g) / CI, c4 = ((4 * g) % CI) / 4;
                        av[lane] = wl[(size_t)sec * m.wsec + (size_t)(u * CT + ct) * 64 + lane];
                        const int slot = conv_a_unit<CI>(march_bpos(a, wave, pt, PT, j) + m.tap2d[u * TPC + sub], c4);
                        if (slot < 0 || slot >= m.PS) { printf("emul: operand slot %d outside the plane (%d slots)\n", slot, m.PS); return false; }
                        bv[lane] = ring[(size_t)(idx % m.R) * m.PS + slot];
                      }
                      // four v_mfma_f32_16x16x4_f32: D[row][col] += sum_{g} A[row][g] * B[g][col], A/B from lane (row|col, g), component s
                      for (int col = 0; col < 16; ++col)
                        for (int row = 0; row < 16; ++row) {
                          float &d = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + ((row >> 2) * 16 + col)) * 4 + (row & 3)];
                          for (int sidx = 0; sidx < 4; ++sidx)
                            for (int g = 0; g < 4; ++g) d = std::fmaf(av[g * 16 + row].v[sidx], bv[g * 16 + col].v[sidx], d);
                        }
                    }
              if (march_section_releases(m.geo, sg, z, sec)) {
                if (idx + 1 < released) { printf("emul: release counter would go backwards\n"); return false; }
                released = idx + 1;
              }
            }
            for (int wave = 0; wave < m.ncw; ++wave)
              for (int pt = 0; pt < PT; ++pt)
                for (int lane = 0; lane < 64; ++lane) {
                  const int j = lane & 15, g = lane >> 4;
                  const int tau = wave * PT + pt, xt = tau % a.TXT, yt = tau / a.TXT;
                  const int qy = py0 + yt, qx = px0 + xt * 16 + j;
                  if (qy >= a.nPH || qx >= a.nPW) continue;
                  for (int ct = 0; ct < CT; ++ct) {
                    const int c0 = (ct0 + ct) * 16 + 4 * g;
                    if (c0 >= a.rows_valid) continue;
                    for (int ro = 0; ro < (W ? 2 : 1); ++ro) {
                    const size_t ob = march_out_index(a, m, zc, z, W ? 2 * qy + ro : qy, qx, c0);
                    for (int r = 0; r < 4; ++r) {
                      const size_t ai = ((((size_t)wave * CT + ct) * PT + pt) * 64 + lane) * 4 + r;
                      float v = acc[ai];
                      if (W) v = ro == 0 ? (acc[ai] + acc[pstride + ai]) + acc[2 * pstride + ai] : (acc[pstride + ai] - acc[2 * pstride + ai]) - acc[3 * pstride + ai];
                      if (raw == 2) v += out[ob + r];
                      if (raw != 1) {
                        v = v * a.scale[c0 + r] + a.bias[c0 + r];
                        if (a.relu) v = std::max(v, 0.f);
                        if (a.add_mode == 1) v += add[ob + r];
                      }
                      out[ob + r] = v;
                    }
                    }
                  }
                }
          }
          L += sg.nl;
          s += sg.zb - sg.za;
        }
        if (loaded != L || released != L) { printf("emul: pass %d ends with %d loaded / %d released of %d\n", po, loaded, released, L); return false; }
      }
    }
  }
  return true;
}

struct Case { const char *name; int D, H, W, Cin, Cout, kd; bool relu, add; };

static int run_case(const Case &cs, int max_plans) {
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  const int taps = cs.kd * 9;
  std::vector<float> in((size_t)cs.D * cs.H * cs.W * cs.Cin), w((size_t)cs.Cout * cs.Cin * taps), sc(cs.Cout), bi(cs.Cout);
  for (auto &v : in) v = U(rng);
  for (auto &v : w) v = U(rng) * 0.2f;
  for (auto &v : sc) v = 0.5f + 0.5f * std::fabs(U(rng));
  for (auto &v : bi) v = 0.3f * U(rng);
  const size_t on = (size_t)cs.D * cs.H * cs.W * cs.Cout;
  std::vector<float> add(on);
  for (auto &v : add) v = U(rng);
  // direct convolution (torch Conv2d/Conv3d semantics, zero padding 1), channels-last
  std::vector<float> ref(on);
  for (int z = 0; z < cs.D; ++z) for (int y = 0; y < cs.H; ++y) for (int x = 0; x < cs.W; ++x) for (int co = 0; co < cs.Cout; ++co) {
    double s = 0;
    for (int tz = 0; tz < cs.kd; ++tz) for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
      const int iz = z + tz - cs.kd / 2, iy = y + ty - 1, ix = x + tx - 1;
      if (iz < 0 || iz >= cs.D || iy < 0 || iy >= cs.H || ix < 0 || ix >= cs.W) continue;
      for (int ci = 0; ci < cs.Cin; ++ci)
        s += (double)w[((((size_t)co * cs.Cin + ci) * cs.kd + tz) * 3 + ty) * 3 + tx] * in[(((size_t)iz * cs.H + iy) * cs.W + ix) * cs.Cin + ci];
    }
    double v = s * sc[co] + bi[co];
    if (cs.relu) v = std::max(v, 0.0);
    if (cs.add) v += add[(((size_t)z * cs.H + y) * cs.W + x) * cs.Cout + co];
    ref[(((size_t)z * cs.H + y) * cs.W + x) * cs.Cout + co] = (float)v;
  }
  ConvLayer L;
  L.Cin = cs.Cin; L.Cout = cs.Cout; L.kd = cs.kd; L.kh = 3; L.kw = 3; L.weight = w.data(); L.scale = sc; L.bias = bi; L.relu = cs.relu;
  const ConvMode mode = cs.Cout == 8 ? CONV_XPAIR : CONV_NORMAL;
  int done = 0, fails = 0;
  for (int rank = 0; rank < 400 && done < max_plans; ++rank) {
    DeviceArena arena;
    arena.host_only = true;
    std::vector<float> out(on, -777.f);
    ConvPlanOut P = plan_conv(L, mode, in.data(), cs.D, cs.H, cs.W, cs.Cin, out.data(), cs.add ? add.data() : nullptr, 1, arena, rank);
    if (rank >= P.ncand) break;
    const ConvLaunch &c = P.launches.at(0);
    if (c.async != 2) continue;
    const bool ok = c.ci == 8 ? emulate<8>(c, in.data(), out.data(), add.data()) : emulate<16>(c, in.data(), out.data(), add.data());
    double worst = 0;
    for (size_t i = 0; i < on; ++i) worst = std::max(worst, (double)std::fabs(out[i] - ref[i]) / (1.0 + std::fabs(ref[i])));
    const bool pass = ok && worst < 2e-5;
    printf("%-22s plan %s w=%d ci=%d nup=%d ct=%d pt=%d tile %dx%d R=%d PS=%d NPI=%d NPO=%d grid %ux%u steps %d: %s (max rel err %.2e)\n", cs.name, c.march.rm ? "rows" : (c.march.wino ? "wino" : "tile"), c.ncw, c.ci, c.nup, c.ct, c.pt,
           c.args.TY, c.args.TXT * 16, c.march.R, c.march.PS, c.march.geo.NPI, c.march.NPO, c.grid.x, c.grid.z, c.march.steps, pass ? "ok" : "FAIL", worst);
    ++done;
    if (!pass) ++fails;
  }
  if (!done) { printf("%-22s no marching plan was produced\n", cs.name); return 1; }
  return fails;
}

int main(int argc, char **argv) {
  setenv("DR_CONV_MARCH", "2", 1);  // marching candidates first in the planner's ranking
  setenv("DR_CONV_ROWMARCH", "2", 1);
  if (argc > 2 && !strcmp(argv[2], "wino")) setenv("DR_CONV_WINO", "2", 1);  // the Winograd form of the 3-D march first (march_consumer_w)
  const int max_plans = argc > 1 ? atoi(argv[1]) : 3;
  const Case cases[] = {
      {"xpair3d_c16", 5, 20, 40, 16, 8, 3, true, false},   // s2.conv0's type
      {"xpair3d_c32", 4, 18, 36, 32, 8, 3, true, false},   // s1.conv0: two outer channel passes through raw partial sums
      {"xpair3d_c8", 3, 34, 66, 8, 8, 3, true, false},     // s3.conv0: 8-channel records
      {"conv3d_16_16", 4, 18, 24, 16, 16, 3, true, false}, // conv2
      {"xpair2d_c8", 2, 20, 70, 8, 8, 1, true, false},     // fn.conv0.1
      {"conv2d_16_16", 3, 19, 33, 16, 16, 1, true, true},  // fn.conv1.x (+ a residual add to cover add_mode 1)
      {"conv2d_32_32", 2, 17, 20, 32, 32, 1, true, false}, // fn.conv2.x: inner channel passes, two row tiles
      {"conv2d_32_16", 2, 16, 32, 32, 16, 1, false, false},// fn.out2: no ReLU
      // wide enough for the row march (a strip is at least 8 position tiles)
      {"rows_xpair_c8", 2, 9, 300, 8, 8, 1, true, false},  // fn.conv0.1
      {"rows_16_16", 2, 7, 170, 16, 16, 1, true, false},   // fn.conv1.x
      {"rows_32_32", 2, 6, 168, 32, 32, 1, true, false},   // fn.conv2.x: two channel slices per row, two row tiles
      {"rows_32_16", 3, 5, 330, 32, 16, 1, false, false},  // fn.out2
      {"rows_xpair_c32", 1, 11, 520, 32, 8, 1, true, false}, // fn.out3's shape without the fused skip
  };
  int fails = 0;
  for (const Case &cs : cases) fails += run_case(cs, max_plans);
  printf(fails ? "MARCH EMULATION: %d FAILED\n" : "MARCH EMULATION: all ok\n", fails);
  return fails ? 1 : 0;
}
