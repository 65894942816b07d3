// conv_emul.hip -- host emulation of k_conv and k_conv_a (tandem_amd/csrc/conv_mfma.h), run by tests/test_conv_plan.py on the CPU.
//
// There is no GPU where the CPU suite runs, so this program executes the generic kernel's DATA FLOW on the host for every layer
// type the planner knows: plan_conv builds the real launch (host-only arena: packed weights, tap tables, parity classes,
// epilogue geometry), then for every (class, tile, row group) the halo tile is staged with the kernel's index rules, the lanes
// gather their MFMA operands through the tap table and the packed weight array exactly as conv_kloop does, a scalar model of
// v_mfma_f32_16x16x4_f32 accumulates them, and conv_epilogue's index arithmetic (output multipliers, class offsets, parity
// rows, residual / upsample add) places the results.  The output is compared with a direct evaluation of the layer's definition
// (torch semantics: Conv3d, ConvTranspose3d(k=3, s=2, p=1, output_padding=1), Conv2d over a nearest x2 upsampling).
// What this covers is the planner -- tap offsets, weight packing (XPAIR / X8 shifts, the three parity forms, the summed kernel
// entries of ConvLayer::up2), class and row-group bookkeeping -- for candidates the GPU tests may never rank first.
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

static bool emulate(const ConvLaunch &c, const float *in, float *out, const float *add) {
  const ConvArgs &a = c.args;
  const int CI = c.ci, CT = c.ct, PT = c.pt, TPC = (c.bf3 ? 32 : 16) / CI;
  const int NP = a.TZI * a.TYI * a.TXI;
  std::vector<float> tile((size_t)NP * CI);
  // k_conv_b (conv_bf3.h): the staged tile as the kernel lays it out -- records of (CI + 4) * 4 bytes, [CI hi bf16 | CI lo bf16 | pad]
  const int RB = (CI + 4) * 4, LO = 2 * CI;
  std::vector<unsigned char> tileb(c.bf3 ? (size_t)NP * RB : 0, 0xff);
  const unsigned short *wpb = reinterpret_cast<const unsigned short *>(a.wpk);
  const unsigned ncls = a.class_loop > 0 ? (unsigned)a.class_loop : c.grid.y;  // (class loop: one workgroup walks the classes, grid.y = 1)
  for (unsigned ic = 0; ic < ncls; ++ic) {
    const ConvClass &cls = a.cls[ic];
    for (unsigned bz = 0; bz < c.grid.z; ++bz) {
      const int ct0 = (int)bz * CT;
      for (int td = 0; td < a.tilesD; ++td) for (int th = 0; th < a.tilesH; ++th) for (int tw = 0; tw < a.tilesW; ++tw) {
        const int pz0 = td * a.TZ, py0 = th * a.TY, px0 = tw * a.TXT * 16;
        const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;
        std::vector<float> acc((size_t)4 * CT * PT * 64 * 4, 0.f);
        for (int p = 0; p < a.npass; ++p) {
          for (int pos = 0; pos < NP; ++pos) {  // stage CI channels of the halo tile, zero outside the tensor
            const int x = pos % a.TXI, y = (pos / a.TXI) % a.TYI, z = pos / (a.TXI * a.TYI);
            const int gz = iz0 + z, gy = iy0 + y, gx = ix0 + x;
            const bool inside = gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
            for (int ch = 0; ch < CI; ++ch)
              tile[(size_t)pos * CI + ch] = inside ? in[(((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + p * CI + ch] : 0.f;
          }
          if (c.bf3) {
            for (int e = 0; e < NP * (CI / 4); ++e) {  // the staging step: element e = (position, group of four channels) -> 8 bytes of hi, 8 of lo
              const int pos = e / (CI / 4), c4 = e % (CI / 4);
              unsigned short *hi = reinterpret_cast<unsigned short *>(tileb.data() + (size_t)pos * RB + c4 * 8), *lo = reinterpret_cast<unsigned short *>(tileb.data() + (size_t)pos * RB + c4 * 8 + LO);
              for (int r = 0; r < 4; ++r) {
                const float v = tile[(size_t)pos * CI + c4 * 4 + r];
                hi[r] = bf16_rne(v);
                lo[r] = bf16_rne(v - bf16_value(hi[r]));
              }
            }
            for (int wave = 0; wave < 4; ++wave)
              for (int u = 0; u < cls.NU; ++u)
                for (int ct = 0; ct < CT; ++ct)
                  for (int pt = 0; pt < PT; ++pt) {
                    float ah[64][8], al[64][8], bh[64][8], bl[64][8];
                    for (int lane = 0; lane < 64; ++lane) {
                      const int j = lane & 15, g = lane >> 4;
                      const size_t frag = (size_t)cls.w_base * 8 + ((((size_t)p * cls.NU + u) * a.ctTot + ct0 + ct) * 2) * 64 * 8;  // (w_base counts 16-byte units)
                      for (int s = 0; s < 8; ++s) { ah[lane][s] = bf16_value(wpb[frag + (size_t)lane * 8 + s]); al[lane][s] = bf16_value(wpb[frag + 64 * 8 + (size_t)lane * 8 + s]); }
                      const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
                      const long long baseb = (long long)(((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx) * RB + ((8 * g) % CI) * 2;
                      const long long addr = baseb + (long long)a.tapoff[cls.tap_base + u * TPC + (8 * g) / CI] * RB;
                      const bool inside = addr >= 0 && addr + LO + 16 <= (long long)NP * RB;
                      for (int s = 0; s < 8; ++s) {
                        bh[lane][s] = inside ? bf16_value(*reinterpret_cast<const unsigned short *>(tileb.data() + addr + 2 * s)) : NAN;
                        bl[lane][s] = inside ? bf16_value(*reinterpret_cast<const unsigned short *>(tileb.data() + addr + LO + 2 * s)) : NAN;
                      }
                    }
                    for (int col = 0; col < 16; ++col)
                      for (int row = 0; row < 16; ++row) {  // v_mfma_f32_16x16x32_bf16 x 3: A row = lane & 15, B column = lane & 15, K slot (lane >> 4, s) pairs with itself
                        float &d = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + ((row >> 2) * 16 + col)) * 4 + (row & 3)];
                        for (int term = 0; term < 3; ++term)
                          for (int g = 0; g < 4; ++g)
                            for (int s = 0; s < 8; ++s) {
                              const float wv = term == 0 ? al[g * 16 + row][s] : ah[g * 16 + row][s];
                              const float xv = term == 1 ? bl[g * 16 + col][s] : bh[g * 16 + col][s];
                              if (ah[g * 16 + row][s] != 0.f) d += wv * xv;  // (a padded tap / row carries weight 0 and may point anywhere)
                            }
                      }
                  }
          } else
          for (int wave = 0; wave < 4; ++wave)
            for (int u = 0; u < cls.NU; ++u)
              for (int ct = 0; ct < CT; ++ct)
                for (int pt = 0; pt < PT; ++pt) {
                  float av[64][4], bv[64][4];
                  for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 15, g = lane >> 4;
                    const float4 w = a.wpk[cls.w_base + (((size_t)p * cls.NU + u) * a.ctTot + ct0 + ct) * 64 + lane];
                    av[lane][0] = w.x; av[lane][1] = w.y; av[lane][2] = w.z; av[lane][3] = w.w;
                    const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
                    const int base = ((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx;
                    const int tap = u * TPC + (4 * g) / CI, c0 = (4 * g) % CI;
                    const int tpos = base + a.tapoff[cls.tap_base + tap];
                    for (int s = 0; s < 4; ++s) bv[lane][s] = (tpos >= 0 && tpos < NP) ? tile[(size_t)tpos * CI + c0 + s] : NAN;  // outside the tile: a planner bug unless its weight is 0
                  }
                  for (int col = 0; col < 16; ++col)
                    for (int row = 0; row < 16; ++row) {
                      float &d = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + ((row >> 2) * 16 + col)) * 4 + (row & 3)];
                      for (int s = 0; s < 4; ++s)
                        for (int g = 0; g < 4; ++g) {
                          const float wv = av[g * 16 + row][s];
                          if (wv != 0.f) d = std::fmaf(wv, bv[g * 16 + col][s], d);  // (a padded tap / row carries weight 0 and may point anywhere)
                        }
                    }
                }
        }
        for (int wave = 0; wave < 4; ++wave)  // conv_epilogue
          for (int pt = 0; pt < PT; ++pt)
            for (int lane = 0; lane < 64; ++lane) {
              const int j = lane & 15, g = lane >> 4;
              const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
              const int qz = pz0 + zt, qy = py0 + yt, qx = px0 + xt * 16 + j;
              if (qz >= a.nPD || qy >= a.nPH || qx >= a.nPW) continue;
              for (int ct = 0; ct < CT; ++ct) {
                const int c0 = (ct0 + ct) * 16 + 4 * g;
                if (c0 >= a.rows_valid) continue;
                int oz = qz * a.omz + cls.ooz, oy = qy * a.omy + cls.ooy, ox = qx * a.omx + cls.oox, ch = c0;
                if (a.par_rows) {
                  const int q = c0 / a.par_rows, bits = (a.par_map >> (3 * q)) & 7;
                  ch = c0 - q * a.par_rows;
                  oz += (bits >> 2) & 1; oy += (bits >> 1) & 1; ox += bits & 1;
                }
                const size_t ob = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + ch;
                size_t ab = ob;
                if (a.add_mode == 2) ab = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + ch;
                for (int r = 0; r < 4; ++r) {
                  float v = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + lane) * 4 + r];
                  v = v * a.scale[c0 + r] + a.bias[c0 + r];
                  if (a.relu) v = std::max(v, 0.f);
                  if (a.add_mode) v += add[ab + r];
                  if (std::isnan(v)) { printf("emul: an operand outside the staged tile reached a non-zero weight\n"); return false; }
                  out[ob + r] = v;
                }
              }
            }
      }
    }
  }
  return true;
}

// k_conv_w (conv_wino.h): k_conv's staging, then per chunk of taps the four rows d0..d3 of every lane's column, the input transform in
// fp32 as the kernel does it, four implicit GEMMs (one per Winograd point, weights [chunk][point][row tile][lane]), the output transform and
// two output rows per position.
static bool emulate_wino(const ConvLaunch &c, const float *in, float *out, const float *add) {
  const ConvArgs &a = c.args;
  const int CI = c.ci, CT = c.ct, PT = c.pt, TPC = 16 / CI;
  const int NP = a.TZI * a.TYI * a.TXI;
  const ConvClass &cls = a.cls[0];
  const int NU = cls.NU, NR = NU / 4;
  if (c.grid.y != 1 || a.sy != 2 || a.omy != 2 || (NU & 3)) { printf("emul: not a k_conv_w launch\n"); return false; }
  std::vector<float> tile((size_t)NP * CI);
  for (unsigned bz = 0; bz < c.grid.z; ++bz) {
    const int ct0 = (int)bz * CT;
    for (int td = 0; td < a.tilesD; ++td) for (int th = 0; th < a.tilesH; ++th) for (int tw = 0; tw < a.tilesW; ++tw) {
      const int pz0 = td * a.TZ, py0 = th * a.TY, px0 = tw * a.TXT * 16;
      const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;
      std::vector<float> acc((size_t)4 * 4 * CT * PT * 64 * 4, 0.f);  // [point][wave][ct][pt][lane][4]
      auto A = [&](int pp, int wave, int ct, int pt, int lane, int r) -> float & { return acc[(((((size_t)pp * 4 + wave) * CT + ct) * PT + pt) * 64 + lane) * 4 + r]; };
      for (int p = 0; p < a.npass; ++p) {
        for (int pos = 0; pos < NP; ++pos) {
          const int x = pos % a.TXI, y = (pos / a.TXI) % a.TYI, z = pos / (a.TXI * a.TYI);
          const int gz = iz0 + z, gy = iy0 + y, gx = ix0 + x;
          const bool inside = gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
          for (int ch = 0; ch < CI; ++ch) tile[(size_t)pos * CI + ch] = inside ? in[(((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + p * CI + ch] : 0.f;
        }
        for (int wave = 0; wave < 4; ++wave)
          for (int r = 0; r < NR; ++r)
            for (int ct = 0; ct < CT; ++ct)
              for (int pt = 0; pt < PT; ++pt) {
                float av[4][64][4], v[4][64][4];
                for (int lane = 0; lane < 64; ++lane) {
                  const int j = lane & 15, g = lane >> 4;
                  for (int pp = 0; pp < 4; ++pp) {
                    const float4 w = a.wpk[cls.w_base + (((size_t)p * NU + r * 4 + pp) * a.ctTot + ct0 + ct) * 64 + lane];
                    av[pp][lane][0] = w.x; av[pp][lane][1] = w.y; av[pp][lane][2] = w.z; av[pp][lane][3] = w.w;
                  }
                  const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
                  const int base = ((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx;
                  const int tap = r * TPC + (4 * g) / CI, c0 = (4 * g) % CI;
                  float d[4][4];
                  for (int q = 0; q < 4; ++q) {
                    const int tpos = base + a.tapoff[cls.tap_base + tap] + q * a.TXI;
                    for (int s = 0; s < 4; ++s) d[q][s] = (tpos >= 0 && tpos < NP) ? tile[(size_t)tpos * CI + c0 + s] : NAN;
                  }
                  for (int s = 0; s < 4; ++s) { v[0][lane][s] = d[0][s] - d[2][s]; v[1][lane][s] = d[1][s] + d[2][s]; v[2][lane][s] = d[2][s] - d[1][s]; v[3][lane][s] = d[1][s] - d[3][s]; }
                }
                for (int pp = 0; pp < 4; ++pp)
                  for (int col = 0; col < 16; ++col)
                    for (int row = 0; row < 16; ++row) {
                      float &dd = A(pp, wave, ct, pt, (row >> 2) * 16 + col, row & 3);
                      for (int s = 0; s < 4; ++s)
                        for (int g = 0; g < 4; ++g) {
                          const float wv = av[pp][g * 16 + row][s];
                          if (wv != 0.f) dd = std::fmaf(wv, v[pp][g * 16 + col][s], dd);
                        }
                    }
              }
      }
      for (int wave = 0; wave < 4; ++wave)
        for (int pt = 0; pt < PT; ++pt)
          for (int lane = 0; lane < 64; ++lane) {
            const int j = lane & 15, g = lane >> 4;
            const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
            const int qz = pz0 + zt, qy = py0 + yt, qx = px0 + xt * 16 + j;
            if (qz >= a.nPD || qy >= a.nPH || qx >= a.nPW) continue;
            for (int ct = 0; ct < CT; ++ct) {
              const int c0 = (ct0 + ct) * 16 + 4 * g;
              if (c0 >= a.rows_valid) continue;
              const int oz = qz * a.omz + cls.ooz, ox = qx * a.omx + cls.oox;
              for (int ro = 0; ro < 2; ++ro) {
                const int oy = qy * 2 + ro;
                const size_t ob = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + c0;
                size_t ab = ob;
                if (a.add_mode == 2) ab = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + c0;
                for (int r = 0; r < 4; ++r) {
                  const float m0 = A(0, wave, ct, pt, lane, r), m1 = A(1, wave, ct, pt, lane, r), m2 = A(2, wave, ct, pt, lane, r), m3 = A(3, wave, ct, pt, lane, r);
                  float vv = ro == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
                  vv = vv * a.scale[c0 + r] + a.bias[c0 + r];
                  if (a.relu) vv = std::max(vv, 0.f);
                  if (a.add_mode) vv += add[ab + r];
                  if (std::isnan(vv)) { printf("emul: an operand outside the staged tile reached a non-zero weight\n"); return false; }
                  out[ob + r] = vv;
                }
              }
            }
          }
    }
  }
  return true;
}

// The persistent LDS-DMA kernel k_conv_a: the tile image every DMA piece produces (conv_a_slot: slot -> staged element, zeros from the
// zero buffer), the workgroups' tile lists (XCD ranges, round-robin inside), operands through conv_a_unit, 8 waves of PT position tiles.
template <int CI>
static bool emulate_async(const ConvLaunch &c, const float *in, float *out, const float *add, long long *tiles_seen) {
  const ConvArgs &a = c.args;
  const ConvClass &cls = a.cls[0];
  const int CT = c.ct, PT = c.pt, TPC = 16 / CI, NP = a.TZI * a.TYI * a.TXI, NU = cls.NU;
  const int ntiles = a.tilesD * a.tilesH * a.tilesW, per_xcd = (ntiles + 7) >> 3;
  std::vector<float> image((size_t)a.a_slots * 4);
  for (unsigned bz = 0; bz < c.grid.z; ++bz) {
    const int ct0 = (int)bz * CT;
    for (unsigned blk = 0; blk < c.grid.x; ++blk) {
      const int xcd = blk & 7, wi = blk >> 3, nw = c.grid.x >> 3;
      const int t_lo = xcd * per_xcd, t_hi = std::min(ntiles, t_lo + per_xcd);
      const int my_tiles = t_lo + wi < t_hi ? (t_hi - t_lo - wi + nw - 1) / nw : 0;
      for (int k = 0; k < my_tiles; ++k) {
        int b = t_lo + wi + k * nw;
        if (bz == 0) ++*tiles_seen;
        const int tw = b % a.tilesW;
        b /= a.tilesW;
        const int pz0 = (b / a.tilesH) * a.TZ, py0 = (b % a.tilesH) * a.TY, px0 = tw * a.TXT * 16;
        const int iz0 = pz0 * a.sz - a.pz, iy0 = py0 * a.sy - a.py, ix0 = px0 * a.sx - a.px;
        std::vector<float> acc((size_t)8 * CT * PT * 64 * 4, 0.f);
        for (int p = 0; p < a.npass; ++p) {
          for (int s = 0; s < a.a_slots; ++s) {
            int pos, c4;
            conv_a_slot<CI>(s, pos, c4);
            const int x = pos % a.TXI, y = (pos / a.TXI) % a.TYI, z = pos / (a.TXI * a.TYI);
            const int gz = iz0 + z, gy = iy0 + y, gx = ix0 + x;
            const bool inside = pos < NP && gz >= 0 && gz < a.inD && gy >= 0 && gy < a.inH && gx >= 0 && gx < a.inW;
            for (int r = 0; r < 4; ++r)
              image[(size_t)s * 4 + r] = inside ? in[(((size_t)gz * a.inH + gy) * a.inW + gx) * a.inC + p * CI + c4 * 4 + r] : (pos < NP ? 0.f : NAN);
          }
          for (int wave = 0; wave < 8; ++wave)
            for (int u = 0; u < NU; ++u)
              for (int ct = 0; ct < CT; ++ct)
                for (int pt = 0; pt < PT; ++pt) {
                  float av[64][4], bv[64][4];
                  for (int lane = 0; lane < 64; ++lane) {
                    const int j = lane & 15, g = lane >> 4, c4 = ((4 * g) % CI) / 4;
                    const float4 w = a.wpk[cls.w_base + (((size_t)p * NU + u) * a.ctTot + ct0 + ct) * 64 + lane];
                    av[lane][0] = w.x; av[lane][1] = w.y; av[lane][2] = w.z; av[lane][3] = w.w;
                    const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
                    const int bpos = ((zt * a.sz) * a.TYI + yt * a.sy) * a.TXI + (xt * 16 + j) * a.sx;
                    const int tpos = bpos + a.tapoff[cls.tap_base + u * TPC + (4 * g) / CI];
                    const int slot = tpos >= 0 ? conv_a_unit<CI>(tpos, c4) : -1;
                    for (int r = 0; r < 4; ++r) bv[lane][r] = (slot >= 0 && slot < a.a_slots) ? image[(size_t)slot * 4 + r] : NAN;
                  }
                  for (int col = 0; col < 16; ++col)
                    for (int row = 0; row < 16; ++row) {
                      float &d = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + ((row >> 2) * 16 + col)) * 4 + (row & 3)];
                      for (int r = 0; r < 4; ++r)
                        for (int g = 0; g < 4; ++g) {
                          const float wv = av[g * 16 + row][r];
                          if (wv != 0.f) d = std::fmaf(wv, bv[g * 16 + col][r], d);
                        }
                    }
                }
        }
        for (int wave = 0; wave < 8; ++wave)
          for (int pt = 0; pt < PT; ++pt)
            for (int lane = 0; lane < 64; ++lane) {
              const int j = lane & 15, g = lane >> 4;
              const int tau = wave * PT + pt, xt = tau % a.TXT, yt = (tau / a.TXT) % a.TY, zt = tau / (a.TXT * a.TY);
              const int qz = pz0 + zt, qy = py0 + yt, qx = px0 + xt * 16 + j;
              if (qz >= a.nPD || qy >= a.nPH || qx >= a.nPW) continue;
              for (int ct = 0; ct < CT; ++ct) {
                const int c0 = (ct0 + ct) * 16 + 4 * g;
                if (c0 >= a.rows_valid) continue;
                int oz = qz * a.omz + cls.ooz, oy = qy * a.omy + cls.ooy, ox = qx * a.omx + cls.oox, ch = c0;
                if (a.par_rows) {
                  const int q = c0 / a.par_rows, bits = (a.par_map >> (3 * q)) & 7;
                  ch = c0 - q * a.par_rows;
                  oz += (bits >> 2) & 1; oy += (bits >> 1) & 1; ox += bits & 1;
                }
                const size_t ob = (((size_t)oz * a.outH + oy) * a.outW + ox) * a.outC + ch;
                size_t ab = ob;
                if (a.add_mode == 2) ab = (((size_t)oz * a.addH + (oy >> 1)) * a.addW + (ox >> 1)) * a.outC + ch;
                for (int r = 0; r < 4; ++r) {
                  float v = acc[((((size_t)wave * CT + ct) * PT + pt) * 64 + lane) * 4 + r];
                  v = v * a.scale[c0 + r] + a.bias[c0 + r];
                  if (a.relu) v = std::max(v, 0.f);
                  if (a.add_mode) v += add[ab + r];
                  if (std::isnan(v)) { printf("emul: an operand outside the staged image reached a non-zero weight\n"); return false; }
                  out[ob + r] = v;
                }
              }
            }
      }
    }
  }
  return true;
}

// kind: 0 conv (any stride), 1 ConvTranspose3d(k=3, pad=1, output_padding = stride - 1), 2 Conv2d 3x3 over the nearest x2 upsampling (one launch
// per row parity), 3 the same with both row parities as the two classes of one launch
struct Case { const char *name; int kind, D, H, W, Cin, Cout, kd, kh, kw, sd, sh, sw; bool relu; int add; /* 0 none, 1 same, 2 up2 */ };

static int run_case(const Case &cs, int max_plans) {
  std::mt19937 rng(4321);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  const int taps = cs.kd * cs.kh * cs.kw;
  std::vector<float> in((size_t)cs.D * cs.H * cs.W * cs.Cin), w((size_t)cs.Cout * cs.Cin * taps), sc(cs.Cout), bi(cs.Cout);
  for (auto &v : in) v = U(rng);
  for (auto &v : w) v = U(rng) * 0.2f;
  for (auto &v : sc) v = 0.5f + 0.5f * std::fabs(U(rng));
  for (auto &v : bi) v = 0.3f * U(rng);
  int oD, oH, oW;
  if (cs.kind == 1) { oD = cs.D * cs.sd; oH = cs.H * cs.sh; oW = cs.W * cs.sw; }
  else if (cs.kind >= 2) { oD = cs.D; oH = 2 * cs.H; oW = 2 * cs.W; }
  else { oD = (cs.D + 2 * (cs.kd / 2) - cs.kd) / cs.sd + 1; oH = (cs.H + 2 * (cs.kh / 2) - cs.kh) / cs.sh + 1; oW = (cs.W + 2 * (cs.kw / 2) - cs.kw) / cs.sw + 1; }
  const size_t on = (size_t)oD * oH * oW * cs.Cout;
  std::vector<float> add(cs.add == 2 ? (size_t)oD * (oH / 2) * (oW / 2) * cs.Cout : on);
  for (auto &v : add) v = U(rng);
  auto at = [&](int z, int y, int x, int c) -> double {
    if (z < 0 || z >= cs.D || y < 0 || y >= cs.H || x < 0 || x >= cs.W) return 0.0;
    return in[(((size_t)z * cs.H + y) * cs.W + x) * cs.Cin + c];
  };
  std::vector<float> ref(on);
  for (int z = 0; z < oD; ++z) for (int y = 0; y < oH; ++y) for (int x = 0; x < oW; ++x) for (int co = 0; co < cs.Cout; ++co) {
    double s = 0;
    for (int tz = 0; tz < cs.kd; ++tz) for (int ty = 0; ty < cs.kh; ++ty) for (int tx = 0; tx < cs.kw; ++tx)
      for (int ci = 0; ci < cs.Cin; ++ci) {
        double v = 0, wt;
        if (cs.kind == 1) {  // out[o] += in[i] * w[ci][co][k] with o = i * s - pad + k, pad = k / 2 (weight layout (Cin, Cout, kd, kh, kw))
          const int nz = z + cs.kd / 2 - tz, ny = y + cs.kh / 2 - ty, nx = x + cs.kw / 2 - tx;
          if (nz % cs.sd || ny % cs.sh || nx % cs.sw) continue;
          v = at(nz / cs.sd, ny / cs.sh, nx / cs.sw, ci);
          if (nz < 0 || ny < 0 || nx < 0) v = 0;
          wt = w[((((size_t)ci * cs.Cout + co) * cs.kd + tz) * cs.kh + ty) * cs.kw + tx];
        } else {
          wt = w[((((size_t)co * cs.Cin + ci) * cs.kd + tz) * cs.kh + ty) * cs.kw + tx];
          if (cs.kind >= 2) {  // the upsampled image: up[Y][X] = in[Y / 2][X / 2], zero outside [0, 2H) x [0, 2W)
            const int Y = y + ty - 1, X = x + tx - 1;
            v = (Y < 0 || Y >= oH || X < 0 || X >= oW) ? 0.0 : at(z, Y / 2, X / 2, ci);
          } else v = at(z * cs.sd + tz - cs.kd / 2, y * cs.sh + ty - cs.kh / 2, x * cs.sw + tx - cs.kw / 2, ci);
        }
        s += wt * v;
      }
    double v = s * sc[co] + bi[co];
    if (cs.relu) v = std::max(v, 0.0);
    if (cs.add == 1) v += add[(((size_t)z * oH + y) * oW + x) * cs.Cout + co];
    if (cs.add == 2) v += add[(((size_t)z * (oH / 2) + y / 2) * (oW / 2) + x / 2) * cs.Cout + co];
    ref[(((size_t)z * oH + y) * oW + x) * cs.Cout + co] = (float)v;
  }
  ConvLayer L;
  L.Cin = cs.Cin; L.Cout = cs.Cout; L.kd = cs.kd; L.kh = cs.kh; L.kw = cs.kw; L.sd = cs.sd; L.sh = cs.sh; L.sw = cs.sw;
  L.transposed = cs.kind == 1; L.weight = w.data(); L.scale = sc; L.bias = bi; L.relu = cs.relu;
  const ConvMode mode = (cs.kind == 0 && cs.sw == 1 && cs.Cout == 8) ? CONV_XPAIR : ((cs.kind == 0 && cs.sw == 1 && cs.Cout == 1) ? CONV_X8 : CONV_NORMAL);
  int done = 0, fails = 0, n_async = 0, n_wino = 0;
  for (int rank = 0; rank < 400 && done < max_plans; rank += 3) {
    std::vector<float> out(on, -777.f);
    DeviceArena arena;
    arena.host_only = true;
    ConvLaunch c{};
    bool ok = true;
    int ncand = 0;
    for (int py = 0; py < (cs.kind == 2 ? 2 : 1) && ok; ++py) {  // up2: one launch per row parity
      L.up2 = cs.kind == 2 ? 1 + py : 0;
      ConvPlanOut P = plan_conv(L, mode, in.data(), cs.D, cs.H, cs.W, cs.Cin, out.data(), cs.add ? add.data() : nullptr, cs.add == 2 ? 2 : 1, arena, rank);
      ncand = P.ncand;
      c = P.launches.at(0);
      if (c.async == 2) { ok = false; printf("%-26s rank %d: unexpected k_conv_m plan\n", cs.name, rank); break; }
      if (c.async == 1) {
        long long seen = 0;
        ok = c.ci == 4 ? emulate_async<4>(c, in.data(), out.data(), add.data(), &seen)
                       : (c.ci == 8 ? emulate_async<8>(c, in.data(), out.data(), add.data(), &seen) : emulate_async<16>(c, in.data(), out.data(), add.data(), &seen));
        if (ok && seen != (long long)c.args.tilesD * c.args.tilesH * c.args.tilesW) { ok = false; printf("emul: the workgroups' tile lists cover %lld of %d tiles\n", seen, c.args.tilesD * c.args.tilesH * c.args.tilesW); }
        ++n_async;
      } else if (c.async == 4) { ok = emulate_wino(c, in.data(), out.data(), add.data()); ++n_wino; }
      else ok = emulate(c, in.data(), out.data(), add.data());
    }
    if (rank >= ncand) break;
    double worst = 0;
    for (size_t i = 0; i < on; ++i) worst = std::max(worst, (double)std::fabs(out[i] - ref[i]) / (1.0 + std::fabs(ref[i])));
    const bool pass = ok && worst < (c.bf3 ? 1e-4 : 2e-5) && (c.bf3 != 0) == (conv_bf3_policy() && cs.Cin % 8 == 0);  // (bf16 x 3: the dropped w_l x_l term and the lo terms' rounding, ~2^-16 per product)
    printf("%-26s plan rank %3d %s ci=%d ct=%d pt=%d tile %dx%dx%d classes %u rows %d: %s (max rel err %.2e)\n", cs.name, rank, c.async == 4 ? "k_conv_w" : (c.async ? "k_conv_a" : (c.bf3 ? "k_conv_b" : "k_conv")), c.ci, c.ct, c.pt, c.args.TZ, c.args.TY,
           c.args.TXT * 16, c.args.class_loop > 0 ? (unsigned)c.args.class_loop : c.grid.y, c.args.rows_valid, pass ? "ok" : "FAIL", worst);
    ++done;
    if (!pass) ++fails;
  }
  if (!done) { printf("%-26s no plan was produced\n", cs.name); return 1; }
  return fails;
}

int main(int argc, char **argv) {
  // argv[2] = "async": rank the persistent LDS-DMA kernel's plans first (layers no k_conv_a plan fits fall back to k_conv); default: k_conv only.
  // DR_CONV_BF16X3=1 in the environment: every layer with Cin % 8 == 0 is planned for and emulated as k_conv_b (three bf16 terms).
  // (k_conv_m has its own emulation: march_emul.hip)
  setenv("DR_CONV_ASYNC", argc > 2 && !strcmp(argv[2], "async") ? "1" : "0", 1);
  setenv("DR_CONV_WINO", argc > 2 && !strcmp(argv[2], "wino") ? "2" : "0", 1);  // "wino": rank k_conv_w's plans first wherever the Winograd form applies
  setenv("DR_CONV_MARCH", "0", 1);
  setenv("DR_CONV_ROWMARCH", "0", 1);
  setenv("DR_CONV_NO_TUNED", "1", 1);
  const int max_plans = argc > 1 ? atoi(argv[1]) : 4;
  const char *form = getenv("DR_DECONV_FORM");
  printf("parity form of the transposed layers: %s\n", form ? form : "default");
  const Case cases[] = {
      {"conv2d_5x5_s2_8_16", 0, 2, 13, 22, 8, 16, 1, 5, 5, 1, 2, 2, true, 0},        // fn.conv1.0
      {"conv2d_1x1_16_32_up2add", 0, 2, 8, 20, 16, 32, 1, 1, 1, 1, 1, 1, false, 2},   // fn.skip2
      {"xpair2d_4_8", 0, 2, 9, 36, 4, 8, 1, 3, 3, 1, 1, 1, true, 0},                  // fn.conv0.0
      {"conv3d_s2_8_16", 0, 6, 10, 20, 8, 16, 3, 3, 3, 2, 2, 2, true, 0},             // conv1
      {"conv3d_s122_32_64", 0, 1, 6, 12, 32, 64, 3, 3, 3, 1, 2, 2, true, 0},          // conv5 at D = 4 stages
      {"conv3d_64_64", 0, 3, 4, 6, 64, 64, 3, 3, 3, 1, 1, 1, true, 0},                // conv6
      {"x8_prob_8_1", 0, 5, 6, 24, 8, 1, 3, 3, 3, 1, 1, 1, false, 0},                 // prob on the MFMA form
      {"deconv_16_8_skip", 1, 3, 5, 9, 16, 8, 3, 3, 3, 2, 2, 2, true, 1},             // conv11
      {"deconv_32_16_skip", 1, 3, 4, 10, 32, 16, 3, 3, 3, 2, 2, 2, true, 1},          // conv9
      {"deconv_64_32_s122", 1, 1, 4, 6, 64, 32, 3, 3, 3, 1, 2, 2, true, 1},           // conv7 at D = 4 stages
      {"up2_32_8_inplace_add", 2, 2, 7, 19, 32, 8, 1, 3, 3, 1, 1, 1, false, 1},       // the folded out.stage3's phase layers
      {"up2_16_16", 2, 1, 5, 33, 16, 16, 1, 3, 3, 1, 1, 1, true, 0},
      // large enough for the persistent kernel's tiles (8 waves x 2-4 position tiles)
      {"conv2d_3x3_16_16", 0, 2, 24, 48, 16, 16, 1, 3, 3, 1, 1, 1, true, 0},           // fn.conv1.x
      {"conv2d_3x3_32_16", 0, 2, 16, 40, 32, 16, 1, 3, 3, 1, 1, 1, false, 0},          // fn.out2: two channel passes
      {"conv3d_16_16_skip", 0, 6, 12, 32, 16, 16, 3, 3, 3, 1, 1, 1, true, 1},          // conv2 (+ a residual add)
      {"xpair3d_16_8", 0, 5, 10, 70, 16, 8, 3, 3, 3, 1, 1, 1, true, 0},                // conv0
      // k_conv_w shapes (even H): Cin = 8 XPAIR, two channel passes, two row tiles, an upsample add
      {"xpair2d_8_8", 0, 2, 12, 36, 8, 8, 1, 3, 3, 1, 1, 1, true, 0},                  // fn.conv0.1
      {"conv2d_3x3_32_32_up2add", 0, 2, 8, 24, 32, 32, 1, 3, 3, 1, 1, 1, false, 2},    // fn.conv2.x shape with an upsample add
      {"conv3d_32_32", 0, 4, 6, 20, 32, 32, 3, 3, 3, 1, 1, 1, true, 0},                // conv4
      {"xpair3d_32_8", 0, 4, 8, 40, 32, 8, 3, 3, 3, 1, 1, 1, true, 0},                 // s1.conv0
      // depth axes of extent 1 and 2 (stage 3's coarse levels): taps that only ever read padding are pruned (prune_taps)
      {"conv3d_64_64_D1", 0, 1, 6, 10, 64, 64, 3, 3, 3, 1, 1, 1, true, 0},             // s3.conv6
      {"conv3d_s2_32_64_D2", 0, 2, 8, 12, 32, 64, 3, 3, 3, 2, 2, 2, true, 0},          // s3.conv5
      {"conv3d_32_32_D2", 0, 2, 6, 20, 32, 32, 3, 3, 3, 1, 1, 1, true, 0},             // s3.conv4 (nothing to prune, two-plane halo)
      {"deconv_64_32_D1", 1, 1, 4, 6, 64, 32, 3, 3, 3, 2, 2, 2, true, 1},              // s3.conv7
  };
  int fails = 0;
  for (const Case &cs : cases) fails += run_case(cs, max_plans);
  printf(fails ? "CONV EMULATION: %d FAILED\n" : "CONV EMULATION: all ok\n", fails);
  return fails ? 1 : 0;
}
