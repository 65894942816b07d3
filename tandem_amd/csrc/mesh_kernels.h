// Marching-cubes mesh extraction over the hashed TSDF volume (included by dr_fusion.hip after the voxel/hash helpers).
//
// Reference behaviour being reproduced: ExtractMeshKernel / ExtractMeshAtPosition / TrilinearInterpolation /
// VertexInterpolation, marching_cubes/mesh_extractor.cu:24-265, and the GetMeshSync output layout,
// tsdfvh/tsdf_volume.cu:781-838.  The reference walks the DENSE lattice lower..upper in steps of one voxel (10^9 cells
// for TANDEM's (-5..5 m)^3 box at 1 cm), does 64 hash lookups per cell and appends triangles with atomicAdd
// (arbitrary order).  Here:
//   * every expression of the reference is separable per axis -- a lattice coordinate g fixes, per axis, the cell
//     position, the two corner positions, their trilinear weights and the voxel indices of the 2x2 (corner, offset)
//     samples -- so k_mc_axes evaluates those float expressions LITERALLY once per axis coordinate (3*n entries);
//   * cells are visited per ALLOCATED block (one 256-lane workgroup each, blocks in sorted key order): the cells a
//     block owns are those whose centre voxel (the GetVoxel(position) that supplies the colour) lies in it, a
//     contiguous g-range per axis found by binary search in the monotone axis table; the block's 10^3 voxel
//     neighbourhood is staged in LDS once and serves all 64 samples of each of its <= 9^3 cells;
//   * triangle order is deterministic: count pass -> exclusive scan over blocks -> emit pass, with a workgroup scan
//     over cells inside a block.
// Same fp32 expressions, no contraction => vertices equal the oracle's bit for bit (as a set of triangles).
#pragma once

namespace dr {

struct McAxis {  // everything the reference derives from one lattice coordinate along one axis
  int m[4];      // global voxel index of sample (corner c, offset k) at m[2*c + k]
  int mc;        // global voxel index of the cell position itself (colour voxel)
  float w[2];    // trilinear weight of corner c
  float q[2];    // position of corner c (c = 0: -half voxel, 1: +half voxel)
};

struct McArgs {
  const McAxis *ax[3];
  int n[3];
  const unsigned long long *sorted_keys;
  int nblk;
  unsigned *counts;          // [nblk] triangles per block (count pass)
  const unsigned *offsets;   // [nblk] exclusive scan of counts (emit pass)
  float *vert, *cols;        // [cap_tri * 9]
  unsigned cap_tri;
};

__global__ void k_mc_axes(McAxis *out, int n, float lo, float vs) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const float pa = (float)g * vs + lo;  // mesh_extractor.cu:253-256
  const float hv = vs / 2.0f;           // :141 (P), M = -P
  McAxis a;
  a.q[0] = pa + (-hv);
  a.q[1] = pa + hv;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float vp = a.q[c] / vs;         // :31
    a.w[c] = vp - floorf(vp);             // :32-34
    const float da = a.q[c] - vs / 2.0f;  // :28-30 pos_dual
    const float s0 = da + 0.0f, s1 = da + vs;
    a.m[2 * c] = f2i(s0 / vs + signf_(s0) * 0.5f);  // GetVoxel -> WorldToGlobalVoxel, tsdf_volume.cu:109-115
    a.m[2 * c + 1] = f2i(s1 / vs + signf_(s1) * 0.5f);
  }
  a.mc = f2i(pa / vs + signf_(pa) * 0.5f);
  out[g] = a;
}

__device__ inline Voxel voxel_at(const FusionDev &d, int vx, int vy, int vz) {  // GetVoxel by global voxel index
  Voxel z; z.sdf = 0.f; z.c[0] = z.c[1] = z.c[2] = 0; z.weight = 0;
  I3 b; b.x = floor_div(vx, kBS); b.y = floor_div(vy, kBS); b.z = floor_div(vz, kBS);
  const int p = find_block(d, b);
  if (p < 0) return z;
  return d.vox[(size_t)p * 512 + pos_mod(vx, kBS) * 64 + pos_mod(vy, kBS) * 8 + pos_mod(vz, kBS)];
}

// VertexInterpolation, mesh_extractor.cu:105-134 (isolevel 0, both colours = the cell's centre voxel)
__device__ inline F3 mc_vertex_pos(F3 p1, F3 p2, float d1, float d2) {
  if (fabsf(0.0f - d1) < 0.00001f) return p1;
  if (fabsf(0.0f - d2) < 0.00001f) return p2;
  if (fabsf(d1 - d2) < 0.00001f) return p1;
  const float mu = (0.0f - d1) / (d2 - d1);
  F3 r;
  r.x = p1.x + mu * (p2.x - p1.x);
  r.y = p1.y + mu * (p2.y - p1.y);
  r.z = p1.z + mu * (p2.z - p1.z);
  return r;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_mc_cells(const FusionDev d, const McArgs a) {
  __shared__ Voxel nb[1000];   // voxels [8B-1, 8B+8]^3 of this block's neighbourhood, index (lx*10 + ly)*10 + lz
  __shared__ int nbptr[27];
  __shared__ int rng[3][2];
  __shared__ unsigned wsum[4];
  __shared__ unsigned run_base;
  __shared__ float sdist[EMIT ? 8 : 1][EMIT ? 256 : 1];  // emit pass: corner values, indexed by a RUNTIME corner id (keeps them out of scratch)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.x;
  const I3 B = unpack_key(a.sorted_keys[bi]);
  if (tid < 3) {  // cells owned by this block along axis `tid`: g with floor(mc / 8) == B
    const McAxis *ax = a.ax[tid];
    const int n = a.n[tid], b = tid == 0 ? B.x : (tid == 1 ? B.y : B.z);
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (floor_div(ax[mid].mc, kBS) >= b) hi = mid; else lo = mid + 1; }
    const int g0 = lo;
    hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (floor_div(ax[mid].mc, kBS) >= b + 1) hi = mid; else lo = mid + 1; }
    rng[tid][0] = g0; rng[tid][1] = lo;
  }
  if (tid >= 64 && tid < 64 + 27) {
    const int k = tid - 64;
    I3 q; q.x = B.x + k / 9 - 1; q.y = B.y + (k / 3) % 3 - 1; q.z = B.z + k % 3 - 1;
    nbptr[k] = find_block(d, q);
  }
  if (tid == 0) run_base = 0;
  __syncthreads();
  const int gx0 = rng[0][0], gy0 = rng[1][0], gz0 = rng[2][0];
  const int rx = rng[0][1] - gx0, ry = rng[1][1] - gy0, rz = rng[2][1] - gz0;
  if (rx <= 0 || ry <= 0 || rz <= 0 || nbptr[13] < 0) {
    if (!EMIT && tid == 0) a.counts[bi] = 0;
    return;
  }
  for (int i = tid; i < 1000; i += 256) {
    const int lx = i / 100, ly = (i / 10) % 10, lz = i % 10;
    const int p = nbptr[((lx + 7) >> 3) * 9 + ((ly + 7) >> 3) * 3 + ((lz + 7) >> 3)];
    Voxel v; v.sdf = 0.f; v.c[0] = v.c[1] = v.c[2] = 0; v.weight = 0;
    if (p >= 0) v = d.vox[(size_t)p * 512 + ((lx + 7) & 7) * 64 + ((ly + 7) & 7) * 8 + ((lz + 7) & 7)];
    nb[i] = v;
  }
  __syncthreads();
  const int ox = B.x * kBS - 1, oy = B.y * kBS - 1, oz = B.z * kBS - 1;
  auto fetch = [&](int vx, int vy, int vz) -> Voxel {
    const unsigned lx = (unsigned)(vx - ox), ly = (unsigned)(vy - oy), lz = (unsigned)(vz - oz);
    if (lx < 10u && ly < 10u && lz < 10u) return nb[(lx * 10 + ly) * 10 + lz];
    return voxel_at(d, vx, vy, vz);  // only if float rounding pushes a sample outside the +-1 neighbourhood
  };
  const int ncell = rx * ry * rz;
  unsigned my_total = 0;
  for (int base = 0; base < ncell; base += 256) {
    const int cell = base + tid;
    unsigned ntri = 0;
    unsigned long long row = ~0ull;
    float qx[2] = {0.f, 0.f}, qy[2] = {0.f, 0.f}, qz[2] = {0.f, 0.f};  // corner positions per axis (c = 0: -half voxel, 1: +half)
    float dist[8];
    Voxel cv;
    if (cell < ncell) {
      const int cx = cell % rx, cy = (cell / rx) % ry, cz = cell / (rx * ry);  // x fastest, as the reference's lattice index
      const McAxis X = a.ax[0][gx0 + cx], Y = a.ax[1][gy0 + cy], Z = a.ax[2][gz0 + cz];
      // cube corners in Bourke order v0..v7 = p010 p110 p100 p000 p011 p111 p101 p001 (mesh_extractor.cu:192-199)
      bool ok = true;
      qx[0] = X.q[0]; qx[1] = X.q[1]; qy[0] = Y.q[0]; qy[1] = Y.q[1]; qz[0] = Z.q[0]; qz[1] = Z.q[1];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sx = (k == 1 || k == 2 || k == 5 || k == 6), sy = (k == 0 || k == 1 || k == 4 || k == 5), sz = k >> 2;
        float acc = 0.0f;
        if (ok) {
          const float wx = X.w[sx], wy = Y.w[sy], wz = Z.w[sz];
          // sample order of TrilinearInterpolation: 000 100 010 001 110 011 101 111 (bit0 = x, bit1 = y, bit2 = z)
          const int order[8] = {0, 1, 2, 4, 3, 6, 5, 7};
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const int o = order[s], kx = o & 1, ky = (o >> 1) & 1, kz = o >> 2;
            const Voxel v = fetch(X.m[2 * sx + kx], Y.m[2 * sy + ky], Z.m[2 * sz + kz]);
            if (v.weight == 0) ok = false;
            const float fa = kx ? wx : (1.0f - wx), fb = ky ? wy : (1.0f - wy), fc = kz ? wz : (1.0f - wz);
            acc += fa * fb * fc * v.sdf;
          }
        }
        dist[k] = acc;
      }
      if (ok) {
        unsigned cube = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) cube |= (dist[k] < 0.0f ? 1u : 0u) << k;
        row = kMcTri[cube];  // all-F for cube 0 / 255 (edgeTable == 0)
#pragma unroll
        for (int i = 0; i < 15; i += 3) ntri += ((row >> (4 * i)) & 15) != 15;
        cv = fetch(X.mc, Y.mc, Z.mc);
      }
    }
    if (!EMIT) { my_total += ntri; continue; }
#pragma unroll
    for (int k = 0; k < 8; ++k) sdist[k][tid] = dist[k];  // own column only: no barrier needed
    // deterministic order inside the block: exclusive scan of ntri over the 256 cells of this round
    unsigned incl = ntri;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned before = run_base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    const unsigned round_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    unsigned t_out = a.offsets[bi] + before + incl - ntri;
    for (int i = 0; i < 15 && ((row >> (4 * i)) & 15) != 15; i += 3, ++t_out) {
      if (t_out >= a.cap_tri) break;
      float *vv = a.vert + (size_t)t_out * 9, *cc = a.cols + (size_t)t_out * 9;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int e = (int)((row >> (4 * (i + k))) & 15);
        const int c1 = kMcEdgeCorner[e][0], c2 = kMcEdgeCorner[e][1];
        // corner k sits at (+x for k in {1,2,5,6}, +y for k in {0,1,4,5}, +z for k >= 4): bit tables 0x66 / 0x33
        F3 p1, p2;
        p1.x = ((0x66 >> c1) & 1) ? qx[1] : qx[0]; p1.y = ((0x33 >> c1) & 1) ? qy[1] : qy[0]; p1.z = (c1 >> 2) ? qz[1] : qz[0];
        p2.x = ((0x66 >> c2) & 1) ? qx[1] : qx[0]; p2.y = ((0x33 >> c2) & 1) ? qy[1] : qy[0]; p2.z = (c2 >> 2) ? qz[1] : qz[0];
        const F3 r = mc_vertex_pos(p1, p2, sdist[c1][tid], sdist[c2][tid]);
        vv[3 * k] = r.x; vv[3 * k + 1] = r.y; vv[3 * k + 2] = r.z;
        cc[3 * k] = (float)cv.c[2] / 255.f;  // GetMeshSync swaps BGR -> RGB (tsdf_volume.cu:810-812)
        cc[3 * k + 1] = (float)cv.c[1] / 255.f;
        cc[3 * k + 2] = (float)cv.c[0] / 255.f;
      }
    }
    __syncthreads();
    if (tid == 0) run_base += round_total;
    __syncthreads();
  }
  if (!EMIT) {
    for (int off = 32; off > 0; off >>= 1) my_total += __shfl_down(my_total, off);
    if (lane == 0) wsum[wave] = my_total;
    __syncthreads();
    if (tid == 0) a.counts[bi] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

__global__ void k_mc_total(const unsigned *counts, const unsigned *offsets, int nblk, unsigned long long *total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *total = nblk > 0 ? (unsigned long long)offsets[nblk - 1] + counts[nblk - 1] : 0ull;
}

}  // namespace dr
