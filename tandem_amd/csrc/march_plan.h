// march_plan.h -- the iteration space of the marching convolution kernel (conv_march.h), shared by the producer waves,
// the consumer waves and the host (planner + the CPU simulation of the ring protocol in tests/cpp/march_sim.cpp).
//
// A launch walks STEPS.  A step is one output plane of one (y, x) tile: (column, z), linear index column * Dc + z.
// A workgroup owns a contiguous range of steps; inside it, a SEGMENT is a maximal run of steps of one column.
// Inputs arrive as LOADS, numbered from 0 per workgroup in the order the producer issues them, load i into ring slot
// i % R:
//   KZ = 3 (3-D layers): the loads of a segment [za, zb) are the input planes max(za-1, 0) .. min(zb, Dc-1) of the
//          column -- every plane of the z-halo is fetched once per segment, not once per output plane;
//   KZ = 1 (2-D layers): every (view, tile) is its own column of one step;
//   NPI channel passes inside a step multiply both (load = one channel slice of one plane).
// A step's K loop runs SECTIONS sec = dz * NPI + pi (dz < KZ, pi < NPI); a section reads exactly one load.
#pragma once
#if defined(__HIPCC__)
#define DR_HD __host__ __device__
#else
#define DR_HD
#endif

namespace dr {

struct MarchGeom {
  int Dc;   // steps (output planes) per column
  int KZ;   // input planes per step: 1 or 3
  int NPI;  // channel passes inside a step
};
struct MarchSeg {
  int col, za, zb;  // steps za .. zb-1 of column col
  int p0;           // first input plane loaded for the segment
  int nl;           // loads of the segment
};

DR_HD inline MarchSeg march_segment(const MarchGeom &g, int s, int s1) {
  MarchSeg r;
  r.col = s / g.Dc;
  r.za = s - r.col * g.Dc;
  r.zb = r.za + (s1 - s) < g.Dc ? r.za + (s1 - s) : g.Dc;
  if (g.KZ == 3) {
    r.p0 = r.za > 0 ? r.za - 1 : 0;
    const int p1 = r.zb < g.Dc ? r.zb : g.Dc - 1;
    r.nl = (p1 - r.p0 + 1) * g.NPI;
  } else {
    r.p0 = r.za;
    r.nl = (r.zb - r.za) * g.NPI;
  }
  return r;
}
// Load read by section `sec` of step z, relative to the first load of the segment; -1: the plane lies outside the volume
// (zero padding along z: the section is skipped, its MFMAs are not issued at all).
DR_HD inline int march_section_load(const MarchGeom &g, const MarchSeg &sg, int z, int sec) {
  const int dz = sec / g.NPI, pi = sec - dz * g.NPI;
  const int plane = g.KZ == 3 ? z - 1 + dz : z;
  if (plane < 0 || plane >= g.Dc) return -1;
  return (plane - sg.p0) * g.NPI + pi;
}
// True when no later section of this wave reads that load again (it is released right after the section).
DR_HD inline bool march_section_releases(const MarchGeom &g, const MarchSeg &sg, int z, int sec) {
  return g.KZ == 1 || sec / g.NPI == 0 || z == sg.zb - 1;
}
// Input plane and channel pass (inside the step) of the l-th load of a segment.
DR_HD inline void march_load_plane(const MarchGeom &g, const MarchSeg &sg, int l, int &plane, int &pi) {
  plane = sg.p0 + l / g.NPI;
  pi = l % g.NPI;
}
// The consumer's view of a segment without divisions in its loops: section (dz, pi) of the current step reads load
// L + rel0 + dz * NPI + pi (== march_section_load) out of ring slot (that index) % R; both advance by NPI per step.
struct MarchCursor {
  int rel0, slot0;
  DR_HD void begin(const MarchGeom &g, const MarchSeg &sg, int L, int R) {
    rel0 = (sg.za - (g.KZ == 3 ? 1 : 0) - sg.p0) * g.NPI;  // -NPI when the segment starts at z = 0
    slot0 = (L + rel0 + 2 * R) % R;
  }
  DR_HD void next_step(const MarchGeom &g, int R) {
    rel0 += g.NPI;
    slot0 = slot0 + g.NPI >= R ? slot0 + g.NPI - R : slot0 + g.NPI;
  }
};

// Balanced step range of workgroup `id` of `n`.
DR_HD inline void march_range(long long steps, int id, int n, int &s0, int &s1) {
  s0 = (int)(steps * id / n);
  s1 = (int)(steps * (id + 1) / n);
}

}  // namespace dr
