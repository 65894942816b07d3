// dr_fusion.hip -- MI355X engine behind the DrFusion operator API (C ABI: include/dr_mi355x.h).
//
// Replaces tandem/libdr/dr_fusion/src (CUDA, managed memory, try-lock hash inserts):
//   HashTable / Heap        tsdfvh/hash_table.cu, heap.cu  -> dense direct-mapped block grid (512^3 cells, one load per
//                                                           look-up) + presence bitmap + request list for block
//                                                           coordinates in [-256, 256)^3; a lock-free open-addressing
//                                                           table (64-bit CAS) outside it; bump pool
//   AllocateFromDepthKernel tsdfvh/tsdf_volume.cu:317-434 -> k_allocate   (one lane per pixel, DDA; new blocks are requested
//                                                           once per wave) + k_alloc_commit (pool slots, bitmap, superblock flags)
//   IntegrateScanKernel     tsdfvh/tsdf_volume.cu:436-513 -> k_cull (allocated blocks whose centre projects into the image)
//                                                           + k_integrate (one 64-lane wave per VISIBLE block, 8 voxels
//                                                           per lane, coalesced 4 KB block read-modify-write, one 12-byte
//                                                           pixel record per voxel) + k_fold_counter
//   GenerateRgbDepthKernel  tsdfvh/tsdf_volume.cu:600-632 -> k_raycast2 (one lane per pixel, sphere tracing; exact 3-instruction
//                                                           division verified against IEEE for all dividends, shared
//                                                           corner coordinates, empty-superblock skip) + k_raycast_fix;
//                                                           k_raycast = the literal form (DR_RAYCAST_V1, parity hook)
//   TsdfVolume::{IntegrateScanAsync,RenderAsync,GetRenderResult}  tsdf_volume.cu:515-737 -> FusionEngine
//   MeshExtractor / ExtractMeshAsync / GetMeshSync  marching_cubes/mesh_extractor.cu, tsdf_volume.cu:739-838
//                                                         -> mesh_kernels.h (per allocated block, LDS neighbourhood)
//
// Semantics follow the canonical form fixed by the CPU oracle (oracle/tsdf_oracle.c header): the voxel
// state keyed by block coordinate is bit-identical; hash slots and pool indices are implementation detail.
// fp32 arithmetic is written in the reference's expression order and compiled with -ffp-contract=off;
// divisions and square roots are IEEE-correct (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <memory>

#include <rocprim/rocprim.hpp>  // device radix sort (block keys) and exclusive scan (triangle offsets) of the mesh path

#include "dr_common.h"
#define DR_MC_CONST __device__ static const
#include "mc_tables.h"

namespace dr {

constexpr int kBS = 8;        // voxel block edge (DrFusionOptions::block_size must be 8, as TANDEM sets it)
constexpr int kMaxDDA = 4096;  // cap on DDA steps per ray (the reference loops unboundedly)
constexpr unsigned long long kEmptyKey = ~0ull;

struct Voxel {  // tsdfvh/voxel.h:13-19 -- 8 bytes
  float sdf;
  unsigned char c[3];
  unsigned char weight;
};
static_assert(sizeof(Voxel) == 8, "voxel layout");

struct PixRec { float dep, sd; unsigned col; };  // written by k_allocate (one lane per pixel), read by k_integrate

struct FusionDev {  // everything the kernels need, passed by value
  drf_options_t o;
  unsigned long long *keys;  // [cap] packed block coordinate or kEmptyKey
  int *vals;                 // [cap] pool index
  unsigned cmask;            // cap - 1
  unsigned long long *blk_key;  // [num_blocks] key of pool block i
  Voxel *vox;                // [num_blocks * 512]
  int *n_alloc;              // allocated pool blocks
  int *err;                  // [0] pool exhausted, [1] coordinate out of packing range
  unsigned long long *cnt;   // [0] voxels updated by the current scan, [1] total, [2] round-trip mismatches
  float *sd;                 // [H*W] per-pixel surface distance |GetPoint3d(i, depth)| of the current scan
  unsigned char *super[2];   // per level: 1 = some block of this superblock of the dense grid is allocated (ray-cast empty-space skip)
  PixRec *pix;               // [H*W] {depth, surface distance, packed BGR} of the current scan: ONE gather per voxel in k_integrate
  unsigned *present;         // kPresentBits^3-bit map: bit set <=> that block is allocated (a cache of `grid`, no state)
  // Dense direct-mapped block index for the block coordinates [-256, 256)^3 (+-10 m at 5 mm voxels, +-20 m at 1 cm): one
  // int per cell, 512 MiB of the 288 GB -- 0 = absent, -1 = requested by the allocation pass of the current scan,
  // p + 1 = pool block p.  A block lookup inside the region is ONE load (ray-cast: 9 lookups per sphere-tracing step);
  // blocks outside it live in the open-addressing table above (keys / vals).
  int *grid;
  unsigned *req;             // [num_blocks] grid cells requested by the current scan (k_allocate -> k_alloc_commit)
  int *req_count;
  int *vis;                  // [num_blocks] pool blocks that pass IntegrateScanKernel's per-block test for the current scan
  int *vis_count;            // (k_cull -> k_integrate; == req_count + 1)
  unsigned *wg_upd;          // [integrate grid] voxels updated per workgroup of k_integrate (summed by k_fold_counter)
  float vs_rcp, fx_rcp, fy_rcp;  // correctly rounded reciprocals of voxel_size, fx, fy for the exact fast divisions below
  int fast_div;              // 1: the three reciprocals passed the exhaustive check against IEEE division (verify_fast_div)
};
constexpr int kPresentBits = 9;  // blocks within [-256, 256)^3: 16 MiB bitmap, L2/MALL resident
constexpr int kGridBits = kPresentBits;

// ---- CUDA float->int conversion semantics (cvt.rzi: saturate, NaN -> 0), see oracle header (4) ----
// v_cvt_i32_f32 IS that conversion (truncate, saturate, NaN -> 0: CDNA ISA "V_CVT_I32_F32"); written as inline asm
// because a C cast leaves out-of-range inputs undefined for the optimiser.  tests/test_fusion_gpu.py holds the kernels
// to the oracle's explicit branches bit for bit, out-of-range projections included.
__device__ inline int f2i(float f) {
  int r;
  asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
__device__ inline unsigned char f2u8(float f) {
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (unsigned char)f;
}

// Exact fp32 division by a per-engine constant b (voxel_size, fx, fy) in three instructions instead of the ~12 of the IEEE
// sequence (v_div_scale/v_rcp/4 v_fma/v_div_fmas/v_div_fixup + two denormal-mode switches): with y = RN(1/b),
//   q0 = RN(a*y);  r = a - b*q0 (exact, one FMA);  q = RN(q0 + r*y)  ==  RN(a/b)
// (Markstein's correction step: q0 is within one ulp of a/b, the FMA residual is exact, the final FMA rounds correctly).
// Not taken on trust: FusionEngine's constructor checks q against a/b for ALL 2^32 dividends for each of the three
// divisors (k_verify_fast_div, ~10 ms each) and the kernels use the IEEE division if any finite dividend with
// 2^-100 <= |a| <= 2^100 (or a = 0) disagrees; dividends outside that range always take the IEEE path (in_fast_range).
__device__ inline float div_exact(float a, float b, float y) {
  const float q0 = a * y;
  const float r = __builtin_fmaf(-q0, b, a);
  return __builtin_fmaf(r, y, q0);
}
__device__ inline bool in_fast_range(float a) {  // 0, or 2^-100 <= |a| <= 2^100 (also false for NaN / Inf)
  const float m = fabsf(a);
  return a == 0.0f || (m >= 7.8886090522101181e-31f && m <= 1.2676506002282294e30f);
}
__global__ void k_verify_fast_div(float b, float y, unsigned long long *mismatches) {
  unsigned long long bad = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float a = __uint_as_float((unsigned)i);
    if (!in_fast_range(a)) continue;
    const float q = div_exact(a, b, y), e = a / b;
    // +-0 compare equal on purpose: every use adds a signed 0.5 / subtracts floor() / feeds f2i (see get_voxel2)
    if (!(q == e)) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };
struct Mat { float m[16]; };

__device__ inline float norm3(F3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }  // utils.h:44-46
__device__ inline F3 xform(const Mat &T, F3 v) {                                          // matrix_utils.h:914-922
  F3 r;
  r.x = T.m[0] * v.x + T.m[1] * v.y + T.m[2] * v.z + T.m[3] * 1.0f;
  r.y = T.m[4] * v.x + T.m[5] * v.y + T.m[6] * v.z + T.m[7] * 1.0f;
  r.z = T.m[8] * v.x + T.m[9] * v.y + T.m[10] * v.z + T.m[11] * 1.0f;
  return r;
}
__device__ inline F3 point3d(const drf_options_t &o, int i, float depth) {  // utils.h:93-101
  const int v = i / o.width, u = i - o.width * v;
  F3 p;
  p.z = depth;
  p.x = ((float)u - o.cx) * p.z / o.fx;
  p.y = ((float)v - o.cy) * p.z / o.fy;
  return p;
}
__device__ inline void project(const drf_options_t &o, F3 p, int &px, int &py) {  // utils.h:103-108
  const float x = (o.fx * p.x) / p.z + o.cx;
  const float y = (o.fy * p.y) / p.z + o.cy;
  px = f2i(roundf(x));
  py = f2i(roundf(y));
}
__device__ inline float signf_(float n) { return (float)((n > 0) - (n < 0)); }
__device__ inline int signi(float n) { return (n > 0) - (n < 0); }

// ---- block-coordinate hash table ----
__device__ inline bool pack_key(I3 p, unsigned long long &k) {
  const int B = 1 << 20;
  if (p.x < -B || p.x >= B || p.y < -B || p.y >= B || p.z < -B || p.z >= B) return false;
  k = ((unsigned long long)(unsigned)(p.x + B) << 42) | ((unsigned long long)(unsigned)(p.y + B) << 21) | (unsigned long long)(unsigned)(p.z + B);
  return true;
}
__device__ inline I3 unpack_key(unsigned long long k) {
  const int B = 1 << 20;
  I3 p;
  p.x = (int)((k >> 42) & 0x1fffff) - B;
  p.y = (int)((k >> 21) & 0x1fffff) - B;
  p.z = (int)(k & 0x1fffff) - B;
  return p;
}
__device__ inline unsigned hash_key(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned)k;
}
__device__ inline bool grid_index(I3 p, unsigned &idx) {  // dense region [-256, 256)^3, z fastest
  constexpr int H = 1 << (kGridBits - 1);
  const unsigned x = (unsigned)(p.x + H), y = (unsigned)(p.y + H), z = (unsigned)(p.z + H);
  if ((x | y | z) >> kGridBits) return false;
  idx = (x << (2 * kGridBits)) | (y << kGridBits) | z;
  return true;
}
// Occupancy of the dense grid at two coarser levels, for the ray-caster's empty-space skip: a "superblock" of level L is
// a cube of (1 << kSuperShift[L])^3 blocks; super[L][i] = 1 as soon as any block inside it is allocated.
// Level 0: 32^3 blocks (1.28 m at 5 mm voxels, 4 KiB of flags), level 1: 8^3 blocks (0.32 m, 256 KiB).
constexpr int kSuperLevels = 2;
constexpr int kSuperShift[kSuperLevels] = {5, 3};
template <int SH>
__device__ inline unsigned super_index(unsigned idx) {
  constexpr unsigned M = (1u << kGridBits) - 1;
  const unsigned x = idx >> (2 * kGridBits), y = (idx >> kGridBits) & M, z = idx & M;
  return ((x >> SH) << (2 * (kGridBits - SH))) | ((y >> SH) << (kGridBits - SH)) | (z >> SH);
}
__device__ inline int find_block_table(const FusionDev &d, I3 p) {  // blocks outside the dense region
  unsigned long long key;
  if (!pack_key(p, key)) return -1;
  unsigned s = hash_key(key) & d.cmask;
  for (unsigned probe = 0; probe <= d.cmask; ++probe) {
    const unsigned long long cur = d.keys[s];
    if (cur == key) return __hip_atomic_load(&d.vals[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // -1 while the inserting lane has not stored the pool index yet (vals starts at -1)
    if (cur == kEmptyKey) return -1;
    s = (s + 1) & d.cmask;
  }
  return -1;
}
__device__ inline int find_block(const FusionDev &d, I3 p) {
  unsigned idx;
  if (grid_index(p, idx)) return d.grid[idx] - 1;  // 0 / -1 (absent / only requested) -> negative
  return find_block_table(d, p);
}
// Insert-if-absent (HashTable::AllocateBlock, hash_table.cu:80-115, without the try-lock drop).
// The allocation DDA asks for ~60 blocks per pixel and, once a map exists, almost all of them are there already: a dense
// presence bitmap answers that with one load from a 16 MiB array.  A block that is NOT there yet is wanted by every ray
// that crosses it -- a few hundred lanes at about the same moment -- so the insert itself is made once per block:
//   * inside the dense region the first lane to turn the block's grid cell from 0 (absent) to -1 (requested) appends the
//     cell to the request list (lanes of one wave that want the same cell elect one of them first, so the word is hit by
//     one CAS per wave instead of one per lane; everybody else sees -1 with a plain load); k_alloc_commit then hands out
//     the pool blocks, one lane per request, no contention.  (Measured on the BASELINE configs[3] loop: the old path --
//     atomic load + CAS + atomicOr per lane per new block -- took 0.42-1.4 ms per frame while the map grew, 0.04 built.)
//   * outside it (|block coordinate| >= 256) the open-addressing table takes the insert directly, as before.
__device__ inline void allocate_block_table(const FusionDev &d, I3 p) {
  unsigned long long key;
  if (!pack_key(p, key)) { d.err[1] = 1; return; }
  unsigned s = hash_key(key) & d.cmask;
  for (unsigned probe = 0; probe <= d.cmask; ++probe) {
    unsigned long long cur = __hip_atomic_load(&d.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return;
    if (cur == kEmptyKey) {
      cur = atomicCAS(&d.keys[s], kEmptyKey, key);
      if (cur == kEmptyKey) {  // we own the slot: take a pool block
        const int idx = atomicAdd(d.n_alloc, 1);
        if (idx >= d.o.num_blocks) { d.err[0] = 1; return; }  // (vals[s] stays -1: the key is there, the block is not)
        d.blk_key[idx] = key;
        __hip_atomic_store(&d.vals[s], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // readers in OTHER kernels see -1 or idx, never garbage
        atomicAdd(&d.n_alloc[3], 1);  // blocks living in the table (outside the dense grid)
        return;
      }
      if (cur == key) return;
    }
    s = (s + 1) & d.cmask;
  }
  d.err[0] = 1;
}
// Called by every lane of the wave in lockstep (`want` false for lanes that have nothing to insert at this DDA step).
__device__ inline void allocate_block(const FusionDev &d, I3 p, bool want) {
  unsigned idx = 0;
  const bool in_grid = want && grid_index(p, idx);
  bool need = in_grid && !((d.present[idx >> 5] >> (idx & 31)) & 1u);
  if (need) need = __hip_atomic_load(&d.grid[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
  unsigned long long todo = __ballot(need);
  if (todo) {
    const int lane = (int)(threadIdx.x & 63);
    bool won = false;
    while (todo) {  // one CAS per distinct cell per wave
      const int leader = __ffsll((long long)todo) - 1;
      const unsigned lidx = (unsigned)__builtin_amdgcn_readlane((int)idx, leader);
      const unsigned long long same = __ballot(need && idx == lidx);
      if (lane == leader) won = atomicCAS(&d.grid[idx], 0, -1) == 0;
      todo &= ~same;
    }
    const unsigned long long wm = __ballot(won);  // the wave's new requests go to the list with ONE atomicAdd
    if (wm) {
      int base = 0;
      if (lane == __ffsll((long long)wm) - 1) base = atomicAdd(d.req_count, __popcll(wm));
      base = __builtin_amdgcn_readlane(base, __ffsll((long long)wm) - 1);
      if (won) {
        const int r = base + __popcll(wm & ((1ull << lane) - 1));
        if (r < d.o.num_blocks) d.req[r] = idx; else d.err[0] = 1;
      }
    }
  }
  if (want && !in_grid) allocate_block_table(d, p);
}
// One lane per requested cell: take a pool block (one atomicAdd per wave), publish it in the grid and the bitmap.
__global__ __launch_bounds__(256) void k_alloc_commit(const FusionDev d) {
  const int n = min(*d.req_count, d.o.num_blocks);
  const int lane = threadIdx.x & 63;
  for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; i0 < n; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + lane;
    const bool act = i < n;
    const unsigned long long m = __ballot(act);
    int base = 0;
    if (lane == 0) base = atomicAdd(d.n_alloc, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (!act) continue;
    const int p = base + __popcll(m & ((1ull << lane) - 1));
    const unsigned idx = d.req[i];
    if (p >= d.o.num_blocks) { d.err[0] = 1; d.grid[idx] = 0; continue; }
    constexpr int H = 1 << (kGridBits - 1);
    I3 c; c.x = (int)(idx >> (2 * kGridBits)) - H; c.y = (int)((idx >> kGridBits) & ((1u << kGridBits) - 1)) - H; c.z = (int)(idx & ((1u << kGridBits) - 1)) - H;
    unsigned long long key = 0;
    pack_key(c, key);
    d.blk_key[p] = key;
    d.grid[idx] = p + 1;
    atomicOr(&d.present[idx >> 5], 1u << (idx & 31));
    d.super[0][super_index<kSuperShift[0]>(idx)] = 1;  // plain stores: every writer writes the same value
    d.super[1][super_index<kSuperShift[1]>(idx)] = 1;
  }
}

// ---- coordinate maps, tsdf_volume.cu:109-145 ----
__device__ inline I3 world_to_global_voxel(const drf_options_t &o, F3 p) {
  const float vs = o.voxel_size;
  I3 r;
  r.x = f2i(p.x / vs + signf_(p.x) * 0.5f);
  r.y = f2i(p.y / vs + signf_(p.y) * 0.5f);
  r.z = f2i(p.z / vs + signf_(p.z) * 0.5f);
  return r;
}
__device__ inline int floor_div(int v, int bs) { return v < 0 ? (v - bs + 1) / bs : v / bs; }
__device__ inline int pos_mod(int v, int bs) { const int r = v % bs; return r < 0 ? r + bs : r; }
__device__ inline void world_to_block_local(const drf_options_t &o, F3 p, I3 &blk, int &local) {
  const I3 v = world_to_global_voxel(o, p);
  constexpr int bs = kBS;
  blk.x = floor_div(v.x, bs); blk.y = floor_div(v.y, bs); blk.z = floor_div(v.z, bs);
  local = pos_mod(v.x, bs) * bs * bs + pos_mod(v.y, bs) * bs + pos_mod(v.z, bs);  // voxel_block.h:37-41
}

// ------------------------------------------------------------------ allocation
__global__ __launch_bounds__(256) void k_allocate(const FusionDev d, const unsigned char *__restrict__ bgr, const float *__restrict__ depth, const Mat T) {
  const drf_options_t &o = d.o;
  const int size = o.height * o.width;
  const float trunc = o.truncation_distance;
  const float bsz = o.block_size * o.voxel_size;
  F3 start; start.x = T.m[3]; start.y = T.m[7]; start.z = T.m[11];
  // every lane of a wave walks its own ray, but the insert helper is called in lockstep (it elects one lane per distinct
  // block): lanes without a (valid) pixel, and lanes whose ray has ended, go along with want = false
  const int i_end = ((size + 63) / 64) * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < i_end; i += gridDim.x * blockDim.x) {
    bool live = i < size;
    const float dep = live ? depth[i] : 0.0f;
    // IntegrateScanKernel recomputes distance(0, GetPoint3d(idx, depth[idx])) for every voxel that projects to
    // pixel idx (tsdf_volume.cu:485-486); it depends on the pixel only, so it is evaluated once here.
    if (live) {
      const float sdist = norm3(point3d(o, i, dep));
      d.sd[i] = sdist;
      PixRec r; r.dep = dep; r.sd = sdist;
      r.col = (unsigned)bgr[3 * i] | ((unsigned)bgr[3 * i + 1] << 8) | ((unsigned)bgr[3 * i + 2] << 16);
      d.pix[i] = r;
    }
    if (dep < o.min_sensor_depth || dep > o.max_sensor_depth) live = false;
    const F3 point = xform(T, point3d(o, live ? i : 0, dep));
    if (point.x == 0 && point.y == 0 && point.z == 0) live = false;
    F3 dv; dv.x = point.x - start.x; dv.y = point.y - start.y; dv.z = point.z - start.z;
    const float dn = norm3(dv);
    F3 dir; dir.x = dv.x / dn; dir.y = dv.y / dn; dir.z = dv.z / dn;
    const float surf = norm3(dv);
    const float reach = surf + trunc;
    F3 re; re.x = start.x + dir.x * reach; re.y = start.y + dir.y * reach; re.z = start.z + dir.z * reach;
    I3 bp, be, st;
    bp.x = f2i(floorf(start.x / bsz)); bp.y = f2i(floorf(start.y / bsz)); bp.z = f2i(floorf(start.z / bsz));
    be.x = f2i(floorf(re.x / bsz)); be.y = f2i(floorf(re.y / bsz)); be.z = f2i(floorf(re.z / bsz));
    st.x = signi(dir.x); st.y = signi(dir.y); st.z = signi(dir.z);
    F3 dt, mt;
    dt.x = (dir.x != 0) ? fabsf(bsz / dir.x) : FLT_MAX;
    dt.y = (dir.y != 0) ? fabsf(bsz / dir.y) : FLT_MAX;
    dt.z = (dir.z != 0) ? fabsf(bsz / dir.z) : FLT_MAX;
    const float bdx = (bp.x + (float)st.x) * bsz, bdy = (bp.y + (float)st.y) * bsz, bdz = (bp.z + (float)st.z) * bsz;
    mt.x = (dir.x != 0) ? (bdx - start.x) / dir.x : FLT_MAX;
    mt.y = (dir.y != 0) ? (bdy - start.y) / dir.y : FLT_MAX;
    mt.z = (dir.z != 0) ? (bdz - start.z) / dir.z : FLT_MAX;
    I3 diff; diff.x = diff.y = diff.z = 0;
    bool neg = false;
    if (bp.x != be.x && dir.x < 0) { diff.x--; neg = true; }
    if (bp.y != be.y && dir.y < 0) { diff.y--; neg = true; }
    if (bp.z != be.z && dir.z < 0) { diff.z--; neg = true; }
    allocate_block(d, bp, live);
    if (__any(live && neg)) {
      if (live && neg) { bp.x += diff.x; bp.y += diff.y; bp.z += diff.z; }
      allocate_block(d, bp, live && neg);
    }
    int steps = 0;
    for (;;) {
      const bool go = live && (bp.x != be.x || bp.y != be.y || bp.z != be.z) && steps++ < kMaxDDA;
      if (!__any(go)) break;
      if (go) {
        if (mt.x < mt.y) {
          if (mt.x < mt.z) { bp.x += st.x; mt.x += dt.x; } else { bp.z += st.z; mt.z += dt.z; }
        } else {
          if (mt.y < mt.z) { bp.y += st.y; mt.y += dt.y; } else { bp.z += st.z; mt.z += dt.z; }
        }
      }
      allocate_block(d, bp, go);
    }
  }
}

// ------------------------------------------------------------------ integration
// Voxel::Combine (voxel.h:21-50).  The colour channels are  uchar((c*w + vc*vw) / (w + vw))  with integer-valued
// operands (c, vc <= 255, w <= 255, vw = 1): the quotient is either an exact integer or at least 1/(w+vw) away from
// one, so the truncated IEEE quotient equals floor(num/den) and can be taken from a 1-ulp reciprocal with a bias of
// half that gap -- bit-identical to the reference's division, a quarter of the instructions.
// (IntegrateScanKernel always passes vw = 1; the general case keeps the division.)
__device__ inline void combine(Voxel &a, const Voxel &b, unsigned char max_weight) {
  const float w = (float)a.weight, vw = (float)b.weight;
  const float den = w + vw;
  if (b.weight == 1) {
    const float r = __builtin_amdgcn_rcpf(den), bias = 0.5f * r;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.c[k] = (unsigned char)(int)(((float)a.c[k] * w + (float)b.c[k]) * r + bias);
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) a.c[k] = f2u8(((float)a.c[k] * w + (float)b.c[k] * vw) / den);
  }
  a.sdf = (a.sdf * w + b.sdf * vw) / den;
  unsigned char nw = (unsigned char)(a.weight + b.weight);
  if (nw > max_weight) nw = max_weight;
  a.weight = nw;
}

// Test hook (drf_test_combine): the SAME device function on arbitrary voxel pairs, so that the reciprocal shortcut above can be
// checked exhaustively against the reference's Voxel::Combine.
__global__ void k_test_combine(const Voxel *__restrict__ a, const Voxel *__restrict__ b, Voxel *__restrict__ out, size_t n, int max_weight) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    Voxel v = a[i];
    combine(v, b[i], (unsigned char)max_weight);
    out[i] = v;
  }
}

// One WAVE per allocated pool block (4 waves per workgroup, grid-strided): per-block work (pose transform of the
// block origin, frustum test) is done once per wave, then 8 iterations of 64 voxels (lane = y*8+z of slab x), each
// a coalesced 512-byte read-modify-write of the block.
// IntegrateScanKernel's per-block test (tsdf_volume.cu:451-470: block origin in front of the camera plane, block centre
// projects into the image) for EVERY allocated block, one LANE per block; survivors go to the visible list (one
// atomicAdd per wave).  The map keeps growing while the camera sees a room-sized part of it: with one wave per allocated
// block this test was half of k_integrate's instructions at 110 k blocks and would dominate at a million.
__global__ __launch_bounds__(256) void k_cull(const FusionDev d, const Mat Ti) {
  const drf_options_t &o = d.o;
  constexpr int bs = kBS;
  const float vs = o.voxel_size;
  const int lane = threadIdx.x & 63;
  const int n_blocks = min(*d.n_alloc, o.num_blocks);  // written by k_alloc_commit earlier on this stream
  __shared__ int wcount[4], wbase;
  for (int e0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; (e0 & ~255) < n_blocks; e0 += gridDim.x * blockDim.x) {  // uniform per workgroup
    const int e = e0 + lane;
    bool vis = false;
    if (e < n_blocks) {
      const I3 P = unpack_key(d.blk_key[e]);
      F3 position; position.x = P.x * vs * bs; position.y = P.y * vs * bs; position.z = P.z * vs * bs;
      const F3 pc = xform(Ti, position);
      if (!(pc.z < 0)) {
        F3 center;  // tsdf_volume.cu:461-465 -- the half-block offset is added in double
        center.x = (float)((double)pc.x + 0.5 * (double)vs * (double)bs);
        center.y = (float)((double)pc.y + 0.5 * (double)vs * (double)bs);
        center.z = (float)((double)pc.z + 0.5 * (double)vs * (double)bs);
        int cx, cy;
        project(o, center, cx, cy);
        vis = cx >= 0 && cy >= 0 && cx < o.width && cy < o.height;
      }
    }
    // one atomicAdd per WORKGROUP and iteration (a single address takes ~10^8 atomics/s: one per wave was measurable)
    const unsigned long long m = __ballot(vis);
    const int wave = threadIdx.x >> 6;
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
      wbase = tot ? atomicAdd(d.vis_count, tot) : 0;
    }
    __syncthreads();
    int base = wbase;
    for (int w = 0; w < wave; ++w) base += wcount[w];
    if (vis) d.vis[base + __popcll(m & ((1ull << lane) - 1))] = e;
    __syncthreads();
  }
}

// A voxel is 8 bytes at an 8-byte-aligned address, but `Voxel` itself only promises the alignment of its float: read as a
// struct it becomes a dword load plus a byte load per field (18 gathers per ray-cast sample).  One 64-bit load instead,
// and one 128-bit load for two voxels that are neighbours in z.
struct alignas(8) Voxel8 { unsigned lo, hi; };
struct __attribute__((packed, aligned(8))) Voxel16 { unsigned a, b, c, d; };
__device__ inline Voxel unpack_voxel(unsigned lo, unsigned hi) {
  Voxel v;
  v.sdf = __uint_as_float(lo);
  v.c[0] = (unsigned char)(hi & 255u); v.c[1] = (unsigned char)((hi >> 8) & 255u); v.c[2] = (unsigned char)((hi >> 16) & 255u);
  v.weight = (unsigned char)(hi >> 24);
  return v;
}
__device__ inline Voxel load_voxel(const Voxel *p) {
  const Voxel8 t = *reinterpret_cast<const Voxel8 *>(p);
  return unpack_voxel(t.lo, t.hi);
}

// One WAVE per VISIBLE block (k_cull's list).
#ifndef DR_INTEGRATE_WAVES
#define DR_INTEGRATE_WAVES 1
#endif
__global__ __launch_bounds__(256, DR_INTEGRATE_WAVES) void k_integrate(const FusionDev d, const unsigned char *__restrict__ bgr,
                                                   const float *__restrict__ depth, const Mat T, const Mat Ti) {
  const drf_options_t &o = d.o;
  constexpr int bs = kBS;
  const float vs = o.voxel_size, trunc = o.truncation_distance, inv_vs = 1.0f / o.voxel_size;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int by = lane >> 3, bz = lane & 7;
  unsigned upd = 0;
  const int n_vis = *d.vis_count;
  // The block of iteration i is named by two dependent look-ups (vis[] -> blk_key[]).  Both are wave-uniform (scalar
  // loads) and fetched ahead -- the list entry two blocks ahead, the key one block ahead -- so that no iteration starts
  // with a round trip to L2 before it can issue its voxel loads.
  const int stride = gridDim.x * 4;
  int iv = blockIdx.x * 4 + wave;
  int e_next = iv < n_vis ? d.vis[iv] : 0;
  int e_next2 = iv + stride < n_vis ? d.vis[iv + stride] : 0;
  unsigned long long key_next = d.blk_key[e_next];
  for (; iv < n_vis; iv += stride) {
    const int e = e_next;
    const I3 P = unpack_key(key_next);
    e_next = e_next2;
    key_next = d.blk_key[e_next];
    e_next2 = iv + 2 * stride < n_vis ? d.vis[iv + 2 * stride] : 0;
    F3 position; position.x = P.x * vs * bs; position.y = P.y * vs * bs; position.z = P.z * vs * bs;
    const F3 pc = xform(Ti, position);
    Voxel *blk_vox = d.vox + (size_t)e * (bs * bs * bs);
    // the block's 4 KB first: these are the loads that go to HBM, and nothing below depends on them until the combine --
    // the corner test, 8 projections and 8 pixel gathers run under their latency (sched_barrier: hipcc otherwise sinks
    // them behind the gathers).  Requesting the NEXT block's 4 KB here as well (two blocks per wave in flight, 121 VGPRs)
    // measured 8 % slower: after the changes above the kernel is bound by vector-ALU issue, not by latency.
    Voxel cur8[bs];
#pragma unroll
    for (int bx = 0; bx < bs; ++bx) cur8[bx] = load_voxel(blk_vox + bx * (bs * bs) + lane);
    __builtin_amdgcn_sched_barrier(0);
    // UpdateVoxel's world -> cam -> world round trip (below) is, per voxel, the map  vp -> T*(Ti*vp)  = an affine map
    // plus fp32 rounding noise (< 1e-4 m for |coordinates| < 64 m).  An affine deviation is extremal at the corners of
    // the block's voxel lattice, so if its 8 corner voxels come back within 0.1 voxel of themselves, every voxel of the
    // block comes back within 0.1 + noise/vs < 0.25 voxel, i.e. onto itself (see the per-voxel test's comment): the
    // round trip is then skipped for the whole block.  Lanes 0..7 test one corner each; any failure, a far-away block or
    // a sub-2 mm grid leaves the per-voxel test in charge.
    bool block_same = false;
    {
      const int kx = (lane & 1) ? bs - 1 : 0, ky = (lane & 2) ? bs - 1 : 0, kz = (lane & 4) ? bs - 1 : 0;
      F3 cv; cv.x = position.x + kx * vs; cv.y = position.y + ky * vs; cv.z = position.z + kz * vs;
      const F3 cw = xform(T, xform(Ti, cv));
      const bool near = fabsf(cw.x * inv_vs - (float)(P.x * bs + kx)) < 0.1f && fabsf(cw.y * inv_vs - (float)(P.y * bs + ky)) < 0.1f &&
                        fabsf(cw.z * inv_vs - (float)(P.z * bs + kz)) < 0.1f;
      const bool small = fabsf(cv.x) < 64.f && fabsf(cv.y) < 64.f && fabsf(cv.z) < 64.f && fabsf(pc.x) < 64.f && fabsf(pc.y) < 64.f && fabsf(pc.z) < 64.f;
      block_same = (__ballot(near && small) & 0xffull) == 0xffull && vs >= 0.002f;
    }
    if (block_same) {
      // Fast path (every voxel is known to map onto itself): the 8 slabs' loads are independent, so issue all of them
      // before any is consumed -- the kernel is bound by the depth -> surface/colour -> voxel load chain, not by ALU.
      F3 vpc[bs];
      int pix[bs];
      float dep8[bs], sd8[bs];
      unsigned col8[bs];
#pragma unroll
      for (int bx = 0; bx < bs; ++bx) {
        F3 vp; vp.x = position.x + bx * vs; vp.y = position.y + by * vs; vp.z = position.z + bz * vs;
        vpc[bx] = xform(Ti, vp);
        int ix, iy;
        project(o, vpc[bx], ix, iy);
        const bool inb = ix >= 0 && iy >= 0 && ix < o.width && iy < o.height;
        pix[bx] = inb ? iy * o.width + ix : -1;
        const int idx = inb ? pix[bx] : 0;
        const PixRec r = d.pix[idx];  // depth[idx], d.sd[idx], bgr[3 idx ..] in one 12-byte gather
        dep8[bx] = r.dep; sd8[bx] = r.sd; col8[bx] = r.col;
      }
#pragma unroll
      for (int bx = 0; bx < bs; ++bx) {
        const float dep = dep8[bx], sd = sd8[bx];
        if (pix[bx] < 0 || dep <= 0 || dep < o.min_sensor_depth || dep > o.max_sensor_depth) continue;
        const float vd = norm3(vpc[bx]);
        Voxel v;
        bool hit = false;
        if (vd > sd - trunc && vd < sd + trunc && dep < o.max_sensor_depth) { v.sdf = sd - vd; hit = true; }
        else if (vd < sd - trunc) { v.sdf = trunc; hit = true; }
        if (!hit) continue;
        v.c[0] = (unsigned char)(col8[bx] & 255u); v.c[1] = (unsigned char)((col8[bx] >> 8) & 255u); v.c[2] = (unsigned char)((col8[bx] >> 16) & 255u);
        v.weight = 1;
        Voxel cur = cur8[bx];
        combine(cur, v, (unsigned char)o.max_sdf_weight);
        blk_vox[bx * (bs * bs) + lane] = cur;
        ++upd;
      }
      continue;
    }
#pragma unroll 2
    for (int bx = 0; bx < bs; ++bx) {
      const int li = bx * (bs * bs) + lane;
      F3 vp; vp.x = position.x + bx * vs; vp.y = position.y + by * vs; vp.z = position.z + bz * vs;
      vp = xform(Ti, vp);
      int ix, iy;
      project(o, vp, ix, iy);
      if (!(ix >= 0 && iy >= 0 && ix < o.width && iy < o.height)) continue;
      const int idx = iy * o.width + ix;
      const float dep = depth[idx];
      if (dep <= 0) continue;
      if (dep < o.min_sensor_depth) continue;
      if (dep > o.max_sensor_depth) continue;
      const float sd = d.sd[idx];
      const float vd = norm3(vp);
      Voxel v;
      bool hit = false;
      if (vd > sd - trunc && vd < sd + trunc && dep < o.max_sensor_depth) { v.sdf = sd - vd; hit = true; }
      else if (vd < sd - trunc) { v.sdf = trunc; hit = true; }
      if (!hit) continue;
      v.c[0] = bgr[3 * idx]; v.c[1] = bgr[3 * idx + 1]; v.c[2] = bgr[3 * idx + 2];
      v.weight = 1;
      // UpdateVoxel re-derives block and voxel from the world position (tsdf_volume.cu:303-315):
      //   g = trunc(wp/vs + sign(wp)*0.5) per axis, block = floor(g/8), local = g mod 8.
      // Sufficient test without the three divisions: if |wp * (1/vs) - n| < 0.25 for this lane's own global voxel
      // index n on every axis, then |wp/vs - n| < 0.3 and the truncation above yields exactly n (for n = 0 as well),
      // i.e. the round trip lands on this lane's voxel.  Otherwise the literal path decides.
      Voxel *dst = blk_vox + li;
      if (!block_same) {
      const F3 wp = xform(T, vp);
      const bool same = fabsf(wp.x * inv_vs - (float)(P.x * bs + bx)) < 0.25f && fabsf(wp.y * inv_vs - (float)(P.y * bs + by)) < 0.25f &&
                        fabsf(wp.z * inv_vs - (float)(P.z * bs + bz)) < 0.25f;
      if (!same) {
        I3 blk; int local;
        world_to_block_local(o, wp, blk, local);
        if (blk.x != P.x || blk.y != P.y || blk.z != P.z || local != li) {
          atomicAdd(&d.cnt[2], 1ull);
          const int target = find_block(d, blk);
          if (target < 0) continue;
          dst = d.vox + (size_t)target * (bs * bs * bs) + local;
        }
      }
      }
      Voxel cur = *dst;
      combine(cur, v, (unsigned char)o.max_sdf_weight);
      *dst = cur;
      ++upd;
    }
  }
  // voxels updated: one plain store per workgroup (k_fold_counter adds them up) instead of an atomicAdd per wave on one
  // address -- with one block per wave that atomic was the longest thing the kernel did
  for (int off = 32; off > 0; off >>= 1) upd += __shfl_down(upd, off);
  __shared__ unsigned wupd[4];
  if (lane == 0) wupd[wave] = upd;
  __syncthreads();
  if (threadIdx.x == 0) d.wg_upd[blockIdx.x] = wupd[0] + wupd[1] + wupd[2] + wupd[3];
}

// ------------------------------------------------------------------ raycast
__device__ inline Voxel get_voxel(const FusionDev &d, F3 p) {  // tsdf_volume.cu:147-160
  Voxel z; z.sdf = 0.f; z.c[0] = z.c[1] = z.c[2] = 0; z.weight = 0;
  I3 blk; int local;
  world_to_block_local(d.o, p, blk, local);
  const int b = find_block(d, blk);
  if (b < 0) return z;
  return d.vox[(size_t)b * (kBS * kBS * kBS) + local];
}

__device__ inline Voxel get_interpolated_voxel(const FusionDev &d, F3 pos) {  // tsdf_volume.cu:161-289
  const Voxel v0 = get_voxel(d, pos);
  if (v0.weight == 0) return v0;
  const float vs = d.o.voxel_size, hv = vs / 2.0f;
  F3 pd; pd.x = pos.x - hv; pd.y = pos.y - hv; pd.z = pos.z - hv;
  F3 vp; vp.x = pos.x / vs; vp.y = pos.y / vs; vp.z = pos.z / vs;
  F3 w; w.x = vp.x - floorf(vp.x); w.y = vp.y - floorf(vp.y); w.z = vp.z - floorf(vp.z);
  float dist = 0.0f, cx = 0.0f, cy = 0.0f, cz = 0.0f;
  Voxel v = v0;
  // corner order of the reference: 000 100 010 001 110 011 101 111
  const int order[8] = {0, 1, 2, 4, 3, 6, 5, 7};  // bit0 = x, bit1 = y, bit2 = z
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = order[k];
    F3 q; q.x = pd.x + ((c & 1) ? vs : 0.0f); q.y = pd.y + ((c & 2) ? vs : 0.0f); q.z = pd.z + ((c & 4) ? vs : 0.0f);
    v = get_voxel(d, q);
    const float a = (c & 1) ? w.x : (1.0f - w.x), b = (c & 2) ? w.y : (1.0f - w.y), cc = (c & 4) ? w.z : (1.0f - w.z);
    const float wt = a * b * cc;
    const Voxel &src = v.weight == 0 ? v0 : v;
    dist += wt * src.sdf;
    cx = cx + (float)src.c[0] * wt;
    cy = cy + (float)src.c[1] * wt;
    cz = cz + (float)src.c[2] * wt;
  }
  v.c[0] = f2u8(cx); v.c[1] = f2u8(cy); v.c[2] = f2u8(cz);
  v.weight = v0.weight;
  v.sdf = dist;
  return v;
}

#ifdef DR_PARITY_HOOKS  // the literal first generation as a whole-image kernel (DR_RAYCAST_V1); k_raycast_fix below is its per-pixel form
__global__ __launch_bounds__(64) void k_raycast(const FusionDev d, const Mat pose, unsigned char *__restrict__ bgr,
                                                float *__restrict__ depth_out) {
  const drf_options_t &o = d.o;
  const int size = o.height * o.width;
  // One wave = one 8x8 pixel tile (a row of 64 pixels fans out over ~6 voxel blocks at 2 m, a tile over 1-2, and
  // PMC showed 4.5 GB of L2 misses per 640x480 render with row-wise waves); tiles are dealt to the 8 XCDs in bands of
  // rows so that neighbouring tiles share an L2.  Sizes that are not multiples of 8 keep the row-wise order.
  const bool tiled = (o.width % 8 == 0) && (o.height % 8 == 0) && blockDim.x == 64;
  const int ntile = tiled ? size / 64 : 0, per_xcd = (ntile + 7) >> 3;
  for (int w0 = blockIdx.x; w0 < (tiled ? 8 * per_xcd : (size + 63) / 64); w0 += gridDim.x) {
    int i;
    if (tiled) {
      const int t = (w0 & 7) * per_xcd + (w0 >> 3);
      if (t >= ntile) continue;
      const int tw = o.width / 8, tx = t % tw, ty = t / tw;
      i = (ty * 8 + (threadIdx.x >> 3)) * o.width + tx * 8 + (threadIdx.x & 7);
    } else {
      i = w0 * 64 + threadIdx.x;
      if (i >= size) continue;
    }
    float cur = 0.f;
    while (cur < o.max_sensor_depth) {
      const Voxel v = get_interpolated_voxel(d, xform(pose, point3d(o, i, cur)));
      if (v.weight == 0) cur += o.truncation_distance; else cur += v.sdf;
      if (v.weight != 0 && v.sdf < o.voxel_size) break;
    }
    if (cur < o.max_sensor_depth) {
      const Voxel v = get_interpolated_voxel(d, xform(pose, point3d(o, i, cur)));
      bgr[3 * i] = v.c[0]; bgr[3 * i + 1] = v.c[1]; bgr[3 * i + 2] = v.c[2];
      depth_out[i] = cur;
    } else {
      bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = 0;
      depth_out[i] = 0.0f;
    }
  }
}

#endif  // DR_PARITY_HOOKS

// ---- ray-cast, second generation: same arithmetic, a fraction of the instructions and of the dependent loads ----
// What GetInterpolatedVoxel costs when it is written out literally (above): 9 GetVoxel calls = 27 IEEE divisions by
// voxel_size + 3 for the weights, and 9 block look-ups, each a probe chain into the hash table -- per sphere-tracing step,
// ~100 steps per pixel.  Here:
//   * every division by voxel_size / fx / fy is div_exact (3 instructions, verified equal to the IEEE quotient);
//   * the 8 dual-grid corners differ per axis in ONE of two coordinates, so 6 voxel coordinates are computed, not 24
//     (each coordinate goes through exactly the expression the reference evaluates for it);
//   * blocks are looked up in the dense grid (one load), once per distinct block of the 2x2x2 corner set (almost
//     always one) and shared with the centre voxel's look-up; the 8 corner loads are then independent of each other;
//   * colour is only interpolated for the final sample of a ray.
template <bool FAST>
__device__ inline float div_by(float a, float b, float y) { return FAST ? div_exact(a, b, y) : a / b; }

// Block look-up of the fast ray-caster: dense grid only.  A coordinate outside the grid is absent if the table holds no
// block at all (d.n_alloc[3] counts table inserts; the usual case), otherwise the pixel bails out to the literal pass.
__device__ inline int find_block_xyz(const FusionDev &d, int x, int y, int z, bool far_blocks, bool &bail) {
  I3 p; p.x = x; p.y = y; p.z = z;
  unsigned idx;
  if (grid_index(p, idx)) return d.grid[idx] - 1;
  if (far_blocks) bail = true;
  return -1;
}

template <bool FAST, bool COLOUR>
__device__ inline Voxel interp_voxel(const FusionDev &d, F3 pos, bool far_blocks, bool &bail, int *empty_cell = nullptr) {  // == get_interpolated_voxel(d, pos), tsdf_volume.cu:161-289
  const float vs = d.o.voxel_size, hv = vs / 2.0f, y = d.vs_rcp;
  Voxel zero; zero.sdf = 0.f; zero.c[0] = zero.c[1] = zero.c[2] = 0; zero.weight = 0;
  // GetVoxel(position): WorldToGlobalVoxel (tsdf_volume.cu:109-113), then block = floor(g / 8), local = g mod 8
  const float qx = div_by<FAST>(pos.x, vs, y), qy = div_by<FAST>(pos.y, vs, y), qz = div_by<FAST>(pos.z, vs, y);
  const int g0x = f2i(qx + signf_(pos.x) * 0.5f), g0y = f2i(qy + signf_(pos.y) * 0.5f), g0z = f2i(qz + signf_(pos.z) * 0.5f);
  const int c0x = g0x >> 3, c0y = g0y >> 3, c0z = g0z >> 3;
  const int b0 = find_block_xyz(d, c0x, c0y, c0z, far_blocks, bail);
  if (empty_cell) {  // dense-grid cell of the centre voxel's block when that block does not exist (else -1)
    I3 c; c.x = c0x; c.y = c0y; c.z = c0z;
    unsigned ci;
    *empty_cell = (b0 < 0 && grid_index(c, ci)) ? (int)ci : -1;
  }
  Voxel v0 = zero;
  if (b0 >= 0) v0 = load_voxel(d.vox + (size_t)b0 * 512 + (((g0x & 7) << 6) | ((g0y & 7) << 3) | (g0z & 7)));
  if (v0.weight == 0) return v0;
  const float pdx = pos.x - hv, pdy = pos.y - hv, pdz = pos.z - hv;
  const float wx = qx - floorf(qx), wy = qy - floorf(qy), wz = qz - floorf(qz);  // voxel_position = position / voxel_size is q
  // per-axis corner coordinates: pos_dual + 0.0f and pos_dual + voxel_size
  int gx[2], gy[2], gz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float ax = pdx + (j ? vs : 0.0f), ay = pdy + (j ? vs : 0.0f), az = pdz + (j ? vs : 0.0f);
    gx[j] = f2i(div_by<FAST>(ax, vs, y) + signf_(ax) * 0.5f);
    gy[j] = f2i(div_by<FAST>(ay, vs, y) + signf_(ay) * 0.5f);
    gz[j] = f2i(div_by<FAST>(az, vs, y) + signf_(az) * 0.5f);
  }
  const int bx0 = gx[0] >> 3, bx1 = gx[1] >> 3, by0 = gy[0] >> 3, by1 = gy[1] >> 3, bz0 = gz[0] >> 3, bz1 = gz[1] >> 3;
  auto look = [&](int x, int yy, int z) { return (x == c0x && yy == c0y && z == c0z) ? b0 : find_block_xyz(d, x, yy, z, far_blocks, bail); };
  int P[8];  // pool block of corner c (bit0 = x, bit1 = y, bit2 = z)
  P[0] = look(bx0, by0, bz0);
  P[1] = bx1 == bx0 ? P[0] : look(bx1, by0, bz0);
  P[2] = by1 == by0 ? P[0] : look(bx0, by1, bz0);
  P[3] = bx1 == bx0 ? P[2] : (by1 == by0 ? P[1] : look(bx1, by1, bz0));
  P[4] = bz1 == bz0 ? P[0] : look(bx0, by0, bz1);
  P[5] = bz1 == bz0 ? P[1] : (bx1 == bx0 ? P[4] : look(bx1, by0, bz1));
  P[6] = bz1 == bz0 ? P[2] : (by1 == by0 ? P[4] : look(bx0, by1, bz1));
  P[7] = bz1 == bz0 ? P[3] : (bx1 == bx0 ? P[6] : (by1 == by0 ? P[5] : look(bx1, by1, bz1)));
  Voxel cv[8];
  // The two z-corners of an (x, y) pair are neighbours in memory (z is the fastest voxel index) whenever they lie in the
  // same block: one 16-byte load then brings both -- 4 gathers per sample instead of 8 for 7 lanes in 8.
  if (bz1 == bz0 && gz[1] == gz[0] + 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int local = ((gx[c & 1] & 7) << 6) | ((gy[(c >> 1) & 1] & 7) << 3) | (gz[0] & 7);
      cv[c] = zero; cv[c | 4] = zero;
      if (P[c] >= 0) {
        const Voxel16 t = *reinterpret_cast<const Voxel16 *>(d.vox + (size_t)P[c] * 512 + local);
        cv[c] = unpack_voxel(t.a, t.b); cv[c | 4] = unpack_voxel(t.c, t.d);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int local = ((gx[c & 1] & 7) << 6) | ((gy[(c >> 1) & 1] & 7) << 3) | (gz[(c >> 2) & 1] & 7);
      cv[c] = zero;
      if (P[c] >= 0) cv[c] = load_voxel(d.vox + (size_t)P[c] * 512 + local);
    }
  }
  float dist = 0.0f, cx = 0.0f, cy = 0.0f, cz = 0.0f;
  const int order[8] = {0, 1, 2, 4, 3, 6, 5, 7};  // the reference's corner order: 000 100 010 001 110 011 101 111
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = order[k];
    const float a = (c & 1) ? wx : (1.0f - wx), b = (c & 2) ? wy : (1.0f - wy), cc = (c & 4) ? wz : (1.0f - wz);
    const float wt = a * b * cc;
    const Voxel &src = cv[c].weight == 0 ? v0 : cv[c];
    dist += wt * src.sdf;
    if (COLOUR) {
      cx = cx + (float)src.c[0] * wt;
      cy = cy + (float)src.c[1] * wt;
      cz = cz + (float)src.c[2] * wt;
    }
  }
  Voxel v;
  v.c[0] = f2u8(cx); v.c[1] = f2u8(cy); v.c[2] = f2u8(cz);
  v.weight = v0.weight;
  v.sdf = dist;
  return v;
}
// interp_voxel in TWO memory round trips.  The statistics of the bench loop (DR_RAYCAST_STATS, r3): a ray takes ~62
// samples, 52 of them inside allocated, carved space (the reference allocates every block between the camera and the
// surface), and all lanes of a wave need about the same number -- the kernel is a chain of dependent gathers, each as slow
// as the slowest of a wave's 64 lanes (some lane always misses L2).  interp_voxel has four dependent stages per sample
// (centre block -> centre voxel -> neighbour blocks -> corner voxels); here every block look-up (centre + the 2x2x2 corner
// blocks, computed from the position alone) is issued at once, then every voxel load (centre + 8 corners, unconditional
// 8-byte loads from a clamped address, masked afterwards) at once.  Same values, same arithmetic, same result.
template <bool FAST, bool COLOUR>
__device__ inline Voxel interp_voxel2(const FusionDev &d, F3 pos, bool far_blocks, bool &bail, int *empty_cell = nullptr) {
  const float vs = d.o.voxel_size, hv = vs / 2.0f, y = d.vs_rcp;
  Voxel zero; zero.sdf = 0.f; zero.c[0] = zero.c[1] = zero.c[2] = 0; zero.weight = 0;
  const float qx = div_by<FAST>(pos.x, vs, y), qy = div_by<FAST>(pos.y, vs, y), qz = div_by<FAST>(pos.z, vs, y);
  const int g0x = f2i(qx + signf_(pos.x) * 0.5f), g0y = f2i(qy + signf_(pos.y) * 0.5f), g0z = f2i(qz + signf_(pos.z) * 0.5f);
  const float pdx = pos.x - hv, pdy = pos.y - hv, pdz = pos.z - hv;
  int gx[2], gy[2], gz[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float ax = pdx + (j ? vs : 0.0f), ay = pdy + (j ? vs : 0.0f), az = pdz + (j ? vs : 0.0f);
    gx[j] = f2i(div_by<FAST>(ax, vs, y) + signf_(ax) * 0.5f);
    gy[j] = f2i(div_by<FAST>(ay, vs, y) + signf_(ay) * 0.5f);
    gz[j] = f2i(div_by<FAST>(az, vs, y) + signf_(az) * 0.5f);
  }
  // ---- round trip 1: nine block look-ups (identical addresses coalesce in the load unit) ----
  auto cell_of = [&](int x, int yy, int z, bool &ok) { I3 p; p.x = x; p.y = yy; p.z = z; unsigned idx = 0; ok = grid_index(p, idx); return ok ? idx : 0u; };
  bool ok0, okc[8];
  const unsigned i0 = cell_of(g0x >> 3, g0y >> 3, g0z >> 3, ok0);
  unsigned ic[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) ic[c] = cell_of(gx[c & 1] >> 3, gy[(c >> 1) & 1] >> 3, gz[(c >> 2) & 1] >> 3, okc[c]);
  int b0 = d.grid[i0];
  int P[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) P[c] = d.grid[ic[c]];
  b0 = ok0 ? b0 - 1 : -1;
  if (!ok0 && far_blocks) bail = true;
  if (empty_cell) *empty_cell = (b0 < 0 && ok0) ? (int)i0 : -1;
  if (b0 < 0) return zero;  // (weight 0: the corner look-ups above were speculative)
#pragma unroll
  for (int c = 0; c < 8; ++c) P[c] = okc[c] ? P[c] - 1 : -1;
  // ---- round trip 2: the centre voxel and the eight corners ----
  const Voxel8 t0 = *reinterpret_cast<const Voxel8 *>(d.vox + (size_t)b0 * 512 + (((g0x & 7) << 6) | ((g0y & 7) << 3) | (g0z & 7)));
  Voxel8 tc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int local = ((gx[c & 1] & 7) << 6) | ((gy[(c >> 1) & 1] & 7) << 3) | (gz[(c >> 2) & 1] & 7);
    tc[c] = *reinterpret_cast<const Voxel8 *>(d.vox + (size_t)(P[c] >= 0 ? P[c] : b0) * 512 + local);
  }
  const Voxel v0 = unpack_voxel(t0.lo, t0.hi);
  if (v0.weight == 0) return v0;
  // the far-block bail of the literal order: a corner outside the dense grid only matters once the centre voxel has weight
#pragma unroll
  for (int c = 0; c < 8; ++c) if (!okc[c] && far_blocks) bail = true;
  const float wx = qx - floorf(qx), wy = qy - floorf(qy), wz = qz - floorf(qz);
  float dist = 0.0f, cx = 0.0f, cy = 0.0f, cz = 0.0f;
  const int order[8] = {0, 1, 2, 4, 3, 6, 5, 7};  // the reference's corner order: 000 100 010 001 110 011 101 111
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = order[k];
    const float a = (c & 1) ? wx : (1.0f - wx), b = (c & 2) ? wy : (1.0f - wy), cc = (c & 4) ? wz : (1.0f - wz);
    const float wt = a * b * cc;
    Voxel cvx = unpack_voxel(tc[c].lo, tc[c].hi);
    if (P[c] < 0) cvx = zero;
    const Voxel &src = cvx.weight == 0 ? v0 : cvx;
    dist += wt * src.sdf;
    if (COLOUR) {
      cx = cx + (float)src.c[0] * wt;
      cy = cy + (float)src.c[1] * wt;
      cz = cz + (float)src.c[2] * wt;
    }
  }
  Voxel v;
  v.c[0] = f2u8(cx); v.c[1] = f2u8(cy); v.c[2] = f2u8(cz);
  v.weight = v0.weight;
  v.sdf = dist;
  return v;
}
// Round 4 measured two more samplers against this one on the bench map (profiles/r04_experiments.txt, 2) and removed both: PAIRED
// gathers (one 8-byte load for the two grid cells and one 16-byte load for the two voxels of a z-corner pair: 4 + 4 gathers, half of
// them not naturally aligned) and the centre voxel SELECTED from the eight corners instead of fetched (8 + 8 gathers, 14 more selects):
// 0.43 ms per render against 0.34 for this sampler.  The kernel is bound by the L1's tag path (PMC, r3: 131.6 M line accesses per
// render, 20 per gather instruction -- the 64 rays of a tile sit in 64 different z-columns of a block), and neither variant lowers the
// number of distinct lines a sample touches; both add instructions.
// Pixels are flagged for the literal pass (k_raycast_fix) with depth -1 when a sample leaves the range div_exact was
// verified on, or needs a block outside the dense grid while the table is not empty.  Neither happens in a room-sized map.
// How many further samples q + j * trunc * dir (j = 1..k) stay inside the superblock of `cell`, shrunk by one voxel on
// every side.  Approximate float arithmetic on purpose: it only has to be conservative (the margin is 5 mm against
// errors of ~1e-5 m), the samples' own positions are never used.
template <int SH>
__device__ inline int skip_steps(unsigned cell, F3 q, F3 dirw, F3 inv_dir, float vs, float inv_trunc) {
  constexpr int H = 1 << (kGridBits - 1);
  constexpr unsigned M = (1u << kGridBits) - 1, SM = ~((1u << SH) - 1u);
  constexpr float hi_off = (float)(kBS << SH) - 1.5f;  // the superblock's n = 8 << SH voxels cover [(g - 0.5) vs, (g + n - 0.5) vs)
  const float gx = (float)((int)(((cell >> (2 * kGridBits)) & SM) - H) * kBS);
  const float gy = (float)((int)((((cell >> kGridBits) & M) & SM) - H) * kBS);
  const float gz = (float)((int)(((cell & M) & SM) - H) * kBS);
  float t = 1e30f;
  if (dirw.x != 0.f) t = fminf(t, ((gx + (dirw.x > 0.f ? hi_off : 0.5f)) * vs - q.x) * inv_dir.x);
  if (dirw.y != 0.f) t = fminf(t, ((gy + (dirw.y > 0.f ? hi_off : 0.5f)) * vs - q.y) * inv_dir.y);
  if (dirw.z != 0.f) t = fminf(t, ((gz + (dirw.z > 0.f ? hi_off : 0.5f)) * vs - q.z) * inv_dir.z);
  return (int)fminf(t * inv_trunc - 0.5f, 256.f);
}
// STATS (DR_RAYCAST_STATS=1, a measuring build of the same loop): per-launch totals of the ray loop in st[] --
// [0] lane iterations, [1] longest ray, [2] sum over waves of their longest ray (what the wave pays), [3] samples whose
// centre block does not exist, [4] skip events, [5] skipped steps, [6] samples with weight != 0, [7] waves, [8..] histogram
// of the waves' longest rays in buckets of 16 iterations.
// SAMPLER: 1 = interp_voxel2 (9 + 9 gathers in two round trips; the product's), 0 = interp_voxel (round 2's four stages; parity build)
template <bool FAST, bool STATS = false, int SAMPLER = 1>
__global__ __launch_bounds__(64) void k_raycast2(const FusionDev d, const Mat pose, unsigned char *__restrict__ bgr,
                                                 float *__restrict__ depth_out, int *__restrict__ n_flagged, unsigned long long *st = nullptr) {
  const drf_options_t &o = d.o;
  const int size = o.height * o.width;
  const bool far_blocks = d.n_alloc[3] != 0;
  // one wave = one 8x8 pixel tile, tiles dealt to the 8 XCDs in bands of rows (see k_raycast)
  const bool tiled = (o.width % 8 == 0) && (o.height % 8 == 0) && blockDim.x == 64;
  const int ntile = tiled ? size / 64 : 0, per_xcd = (ntile + 7) >> 3;
  for (int w0 = blockIdx.x; w0 < (tiled ? 8 * per_xcd : (size + 63) / 64); w0 += gridDim.x) {
    int i;
    if (tiled) {
      const int t = (w0 & 7) * per_xcd + (w0 >> 3);
      if (t >= ntile) continue;
      const int tw = o.width / 8, tx = t % tw, ty = t / tw;
      i = (ty * 8 + (threadIdx.x >> 3)) * o.width + tx * 8 + (threadIdx.x & 7);
    } else {
      i = w0 * 64 + threadIdx.x;
      if (i >= size) continue;
    }
    // GetPoint3d(i, cur, sensor) (utils.h:93-101): x = (u - cx) * z / fx, the pixel part is constant along the ray
    const int pv = i / o.width, pu = i - o.width * pv;
    const float ucx = (float)pu - o.cx, vcy = (float)pv - o.cy;
    bool bail = false;
    auto sample_pos = [&](float cur) {
      F3 p;
      p.z = cur;
      const float tx = ucx * cur, ty = vcy * cur;
      p.x = div_by<FAST>(tx, o.fx, d.fx_rcp);
      p.y = div_by<FAST>(ty, o.fy, d.fy_rcp);
      const F3 q = xform(pose, p);
      if (FAST && !(in_fast_range(q.x) && in_fast_range(q.y) && in_fast_range(q.z) && in_fast_range(tx) && in_fast_range(ty))) bail = true;
      return q;
    };
    // Empty-space skip.  A sample whose centre voxel lies in a block that does not exist returns weight 0 and the ray
    // advances by the truncation distance (2 cm at TANDEM's settings: ~150 look-ups across a room).  d.super[] marks the
    // superblocks (32^3 and 8^3 blocks) of the dense grid that hold any block at all; while the ray stays inside an empty one
    // (shrunk by a voxel on every side: three orders of magnitude above the float error of the approximate ray used
    // here) every sample is known to return weight 0, so `cur` takes the same sequence of float additions -- the
    // result is bit-identical -- without transforming, dividing or loading anything.  DR_RAYCAST_NO_SKIP=1 turns it off.
    F3 dirw, inv_dir;
    {
      const float lx = ucx / o.fx, ly = vcy / o.fy;
      dirw.x = pose.m[0] * lx + pose.m[1] * ly + pose.m[2];
      dirw.y = pose.m[4] * lx + pose.m[5] * ly + pose.m[6];
      dirw.z = pose.m[8] * lx + pose.m[9] * ly + pose.m[10];
      inv_dir.x = 1.0f / dirw.x; inv_dir.y = 1.0f / dirw.y; inv_dir.z = 1.0f / dirw.z;
    }
    const float inv_trunc = 1.0f / o.truncation_distance, vs = o.voxel_size;
    float cur = 0.f;
    unsigned n_it = 0, n_miss = 0, n_skip = 0, n_skipped = 0, n_full = 0;
    while (cur < o.max_sensor_depth) {
      const F3 q = sample_pos(cur);
      int cell = -1;
      const Voxel v = SAMPLER == 1 ? interp_voxel2<FAST, false>(d, q, far_blocks, bail, d.super[0] ? &cell : nullptr)
                                   : interp_voxel<FAST, false>(d, q, far_blocks, bail, d.super[0] ? &cell : nullptr);
      if (bail) break;
      if (STATS) { ++n_it; n_miss += cell >= 0; n_full += v.weight != 0; }
      if (v.weight == 0) {
        cur += o.truncation_distance;
        if (cell >= 0) {
          int k = 0;
          if (d.super[0][super_index<kSuperShift[0]>((unsigned)cell)] == 0) k = skip_steps<kSuperShift[0]>((unsigned)cell, q, dirw, inv_dir, vs, inv_trunc);
          else if (d.super[1][super_index<kSuperShift[1]>((unsigned)cell)] == 0) k = skip_steps<kSuperShift[1]>((unsigned)cell, q, dirw, inv_dir, vs, inv_trunc);
          if (STATS && k > 0) { ++n_skip; n_skipped += k; }
          for (; k > 0 && cur < o.max_sensor_depth; --k) cur += o.truncation_distance;
        }
      } else cur += v.sdf;
      if (v.weight != 0 && v.sdf < o.voxel_size) break;
    }
    if (STATS) {
      unsigned mx = n_it, sum = n_it, sm = n_miss, ss = n_skip, sk = n_skipped, sf = n_full;
      for (int off = 32; off; off >>= 1) {
        mx = max(mx, (unsigned)__shfl_xor((int)mx, off)); sum += __shfl_xor((int)sum, off); sm += __shfl_xor((int)sm, off);
        ss += __shfl_xor((int)ss, off); sk += __shfl_xor((int)sk, off); sf += __shfl_xor((int)sf, off);
      }
      if (threadIdx.x == 0) {
        atomicAdd(&st[0], (unsigned long long)sum); atomicMax(&st[1], (unsigned long long)mx); atomicAdd(&st[2], (unsigned long long)mx);
        atomicAdd(&st[3], (unsigned long long)sm); atomicAdd(&st[4], (unsigned long long)ss); atomicAdd(&st[5], (unsigned long long)sk);
        atomicAdd(&st[6], (unsigned long long)sf); atomicAdd(&st[7], 1ull); atomicAdd(&st[8 + min(mx / 16u, 23u)], 1ull);
      }
    }
    if (!bail && cur < o.max_sensor_depth) {
      const F3 qf = sample_pos(cur);
      const Voxel v = SAMPLER == 1 ? interp_voxel2<FAST, true>(d, qf, far_blocks, bail) : interp_voxel<FAST, true>(d, qf, far_blocks, bail);
      bgr[3 * i] = v.c[0]; bgr[3 * i + 1] = v.c[1]; bgr[3 * i + 2] = v.c[2];
      depth_out[i] = cur;
    } else {
      bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = 0;
      depth_out[i] = 0.0f;
    }
    if (bail) { depth_out[i] = -1.0f; atomicAdd(n_flagged, 1); }
  }
}
// The literal ray-caster for the pixels k_raycast2 flagged; exits at once when there are none.
__global__ __launch_bounds__(64) void k_raycast_fix(const FusionDev d, const Mat pose, unsigned char *__restrict__ bgr,
                                                    float *__restrict__ depth_out, int *__restrict__ n_flagged) {
  if (*n_flagged == 0) return;
  const drf_options_t &o = d.o;
  const int size = o.height * o.width;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < size; i += gridDim.x * blockDim.x) {
    if (!(depth_out[i] == -1.0f)) continue;
    float cur = 0.f;
    while (cur < o.max_sensor_depth) {
      const Voxel v = get_interpolated_voxel(d, xform(pose, point3d(o, i, cur)));
      if (v.weight == 0) cur += o.truncation_distance; else cur += v.sdf;
      if (v.weight != 0 && v.sdf < o.voxel_size) break;
    }
    if (cur < o.max_sensor_depth) {
      const Voxel v = get_interpolated_voxel(d, xform(pose, point3d(o, i, cur)));
      bgr[3 * i] = v.c[0]; bgr[3 * i + 1] = v.c[1]; bgr[3 * i + 2] = v.c[2];
      depth_out[i] = cur;
    } else {
      bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = 0;
      depth_out[i] = 0.0f;
    }
  }
}
__global__ void k_zero_int(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 0; }

__global__ __launch_bounds__(256) void k_fold_counter(unsigned long long *cnt, int *req_count, const unsigned *wg_upd, int n_wg) {
  // end of scan: this scan's update count -> last / total, request and visible lists emptied
  __shared__ unsigned long long part[256];
  unsigned long long a = 0;
  for (int i = threadIdx.x; i < n_wg; i += 256) a += wg_upd[i];
  part[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) { const unsigned long long u = part[0] + cnt[0]; cnt[1] += u; cnt[3] = u; cnt[0] = 0; cnt[4] += (unsigned long long)req_count[1]; req_count[0] = 0; req_count[1] = 0; }  // cnt[4]: blocks k_integrate visited (each one a 4 KB read), all scans
}
__global__ void k_fill_keys(unsigned long long *keys, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) keys[i] = kEmptyKey;
}

}  // namespace dr
#include "mesh_kernels.h"
namespace dr {

// cofactor inverse on the host in the reference's term order (matrix_utils.h:958-1083), fp32, no contraction
static void inverse4_host(const float *e, float *out) {
  float inv[16];
  auto t3 = [&](int a, int b, int c) { return e[a] * e[b] * e[c]; };
  inv[0] = t3(5, 10, 15) - t3(5, 11, 14) - t3(9, 6, 15) + t3(9, 7, 14) + t3(13, 6, 11) - t3(13, 7, 10);
  inv[4] = -t3(4, 10, 15) + t3(4, 11, 14) + t3(8, 6, 15) - t3(8, 7, 14) - t3(12, 6, 11) + t3(12, 7, 10);
  inv[8] = t3(4, 9, 15) - t3(4, 11, 13) - t3(8, 5, 15) + t3(8, 7, 13) + t3(12, 5, 11) - t3(12, 7, 9);
  inv[12] = -t3(4, 9, 14) + t3(4, 10, 13) + t3(8, 5, 14) - t3(8, 6, 13) - t3(12, 5, 10) + t3(12, 6, 9);
  inv[1] = -t3(1, 10, 15) + t3(1, 11, 14) + t3(9, 2, 15) - t3(9, 3, 14) - t3(13, 2, 11) + t3(13, 3, 10);
  inv[5] = t3(0, 10, 15) - t3(0, 11, 14) - t3(8, 2, 15) + t3(8, 3, 14) + t3(12, 2, 11) - t3(12, 3, 10);
  inv[9] = -t3(0, 9, 15) + t3(0, 11, 13) + t3(8, 1, 15) - t3(8, 3, 13) - t3(12, 1, 11) + t3(12, 3, 9);
  inv[13] = t3(0, 9, 14) - t3(0, 10, 13) - t3(8, 1, 14) + t3(8, 2, 13) + t3(12, 1, 10) - t3(12, 2, 9);
  inv[2] = t3(1, 6, 15) - t3(1, 7, 14) - t3(5, 2, 15) + t3(5, 3, 14) + t3(13, 2, 7) - t3(13, 3, 6);
  inv[6] = -t3(0, 6, 15) + t3(0, 7, 14) + t3(4, 2, 15) - t3(4, 3, 14) - t3(12, 2, 7) + t3(12, 3, 6);
  inv[10] = t3(0, 5, 15) - t3(0, 7, 13) - t3(4, 1, 15) + t3(4, 3, 13) + t3(12, 1, 7) - t3(12, 3, 5);
  inv[14] = -t3(0, 5, 14) + t3(0, 6, 13) + t3(4, 1, 14) - t3(4, 2, 13) - t3(12, 1, 6) + t3(12, 2, 5);
  inv[3] = -t3(1, 6, 11) + t3(1, 7, 10) + t3(5, 2, 11) - t3(5, 3, 10) - t3(9, 2, 7) + t3(9, 3, 6);
  inv[7] = t3(0, 6, 11) - t3(0, 7, 10) - t3(4, 2, 11) + t3(4, 3, 10) + t3(8, 2, 7) - t3(8, 3, 6);
  inv[11] = -t3(0, 5, 11) + t3(0, 7, 9) + t3(4, 1, 11) - t3(4, 3, 9) - t3(8, 1, 7) + t3(8, 3, 5);
  inv[15] = t3(0, 5, 10) - t3(0, 6, 9) - t3(4, 1, 10) + t3(4, 2, 9) + t3(8, 1, 6) - t3(8, 2, 5);
  const float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
  const float detr = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}

// Hands a finished render to the host: both images are written straight into the pinned result buffers by ONE kernel (16-byte
// stores over PCIe) instead of two copy-engine transfers behind the ray-cast -- each of those costs its own launch and completion
// latency, which for 2 MB is most of the time (0.14 ms for the pair against 0.05 ms here, DESIGN.md "render hand-off").
__global__ __launch_bounds__(256) void k_publish(const unsigned char *__restrict__ a, unsigned char *__restrict__ ha, size_t na,
                                                 const unsigned char *__restrict__ b, unsigned char *__restrict__ hb, size_t nb) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  const size_t qa = na >> 4, qb = nb >> 4;
  for (size_t i = t; i < qa + qb; i += nt) {
    if (i < qa) reinterpret_cast<uint4 *>(ha)[i] = reinterpret_cast<const uint4 *>(a)[i];
    else reinterpret_cast<uint4 *>(hb)[i - qa] = reinterpret_cast<const uint4 *>(b)[i - qa];
  }
  for (size_t i = (qa << 4) + t; i < na; i += nt) ha[i] = a[i];
  for (size_t i = (qb << 4) + t; i < nb; i += nt) hb[i] = b[i];
}

// ------------------------------------------------------------------ engine
constexpr int kDefaultFusionPriority = 1;  // 0 least (the reference's), 1 normal (measured best in the TandemBackend loop), 2 greatest
class FusionEngine {
 public:
  FusionEngine(const drf_options_t &o, int device) : device_(device), o_(o) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) fail(DR_ERR_DEVICE, "DrFusion: no HIP device %d (found %d) -- the MI355X path has no CPU fallback", device, n);
    if (o.block_size != 8) fail(DR_ERR_UNSUPPORTED, "DrFusion: block_size must be 8 (got %d)", o.block_size);
    if (o.num_blocks >= (1 << 30) - 1) fail(DR_ERR_UNSUPPORTED, "DrFusion: num_blocks must be below 2^30 - 1 (got %d)", o.num_blocks);
    if (o.height <= 0 || o.width <= 0 || o.num_blocks <= 0 || o.num_buckets <= 0 || o.bucket_size <= 0 || o.num_render_streams < 0)
      fail(DR_ERR_ARG, "DrFusion: invalid options");
    DR_HIP(hipSetDevice(device_));
    // Stream priority of the integrate and render streams.  The reference creates them at the LEAST priority with a note that
    // higher may be better (tsdf_volume.cu:64-70).  DR_FUSION_PRIORITY=low|normal|high selects it here; see DESIGN.md
    // "TandemBackend loop" for the measurement behind the default.
    int least, greatest;
    DR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int lo = kDefaultFusionPriority == 2 ? greatest : (kDefaultFusionPriority == 1 ? 0 : least);
    if (const char *e = getenv("DR_FUSION_PRIORITY")) lo = !strcmp(e, "high") ? greatest : (!strcmp(e, "normal") ? 0 : (!strcmp(e, "low") ? least : lo));
    DR_HIP(hipStreamCreateWithPriority(&int_stream_, hipStreamNonBlocking, lo));
    npix_ = (size_t)o.height * o.width;
    size_t cap = 1024;
    const size_t want = std::max((size_t)o.num_buckets * (size_t)o.bucket_size, (size_t)2 * o.num_blocks);
    while (cap < want) cap <<= 1;
    d_.o = o;
    d_.keys = dalloc<unsigned long long>(cap);
    d_.vals = dalloc<int>(cap);
    // -1 everywhere: an insert publishes the key (CAS) BEFORE it stores the pool index, so a concurrent reader -- the ray-cast of scan k
    // beside the allocation of scan k + 1 (enqueue_scan) -- may match a key whose value is not there yet; it then reads -1 = absent, which
    // is the state the block was in a moment ago (its voxels are still all unobserved: the same ray-cast result either way)
    DR_HIP(hipMemsetAsync(d_.vals, 0xFF, cap * sizeof(int), int_stream_));
    d_.cmask = (unsigned)(cap - 1);
    d_.blk_key = dalloc<unsigned long long>(o.num_blocks);
    d_.vox = dalloc<Voxel>((size_t)o.num_blocks * 512);
    d_.n_alloc = dalloc<int>(4);
    d_.err = d_.n_alloc + 1;
    d_.cnt = dalloc<unsigned long long>(8);
    d_.sd = dalloc<float>(npix_);
    d_.pix = dalloc<PixRec>(npix_);
    for (int l = 0; l < kSuperLevels; ++l) d_.super[l] = dalloc<unsigned char>((size_t)1 << (3 * (kGridBits - kSuperShift[l])));
    d_.present = dalloc<unsigned>((size_t)1 << (3 * kPresentBits - 5));
    DR_HIP(hipMemsetAsync(d_.present, 0, (size_t)1 << (3 * kPresentBits - 3), int_stream_));
    d_.grid = dalloc<int>((size_t)1 << (3 * kGridBits));
    DR_HIP(hipMemsetAsync(d_.grid, 0, sizeof(int) << (3 * kGridBits), int_stream_));
    for (int l = 0; l < kSuperLevels; ++l) DR_HIP(hipMemsetAsync(d_.super[l], 0, (size_t)1 << (3 * (kGridBits - kSuperShift[l])), int_stream_));
    d_.req = dalloc<unsigned>(o.num_blocks);
    d_.req_count = dalloc<int>(4);
    d_.vis_count = d_.req_count + 1;
    d_.vis = dalloc<int>(o.num_blocks);
    d_.wg_upd = dalloc<unsigned>(65536);
    DR_HIP(hipMemsetAsync(d_.req_count, 0, 16, int_stream_));
    setup_fast_div();
    hipLaunchKernelGGL(k_fill_keys, dim3(1024), dim3(256), 0, int_stream_, d_.keys, cap);
    DR_HIP(hipMemsetAsync(d_.vox, 0, (size_t)o.num_blocks * 512 * sizeof(Voxel), int_stream_));  // hash_table.cu:28-32
    DR_HIP(hipMemsetAsync(d_.n_alloc, 0, 16, int_stream_));
    DR_HIP(hipMemsetAsync(d_.cnt, 0, 64, int_stream_));
    d_bgr_in_ = dalloc<unsigned char>(npix_ * 3);
    d_depth_in_ = dalloc<float>(npix_);
    DR_HIP(hipHostMalloc((void **)&h_bgr_in_, npix_ * 3, hipHostMallocDefault));
    DR_HIP(hipHostMalloc((void **)&h_depth_in_, npix_ * 4, hipHostMallocDefault));
    // 12 workgroups per CU: enough to keep every SIMD's 4 resident waves busy, few enough that the blocks being worked on
    // at any moment are neighbours in the pool (sweep on the bench map: 1024 / 3072 / 4096 / 6144 / 8192 / 16384
    // workgroups -> 0.341 / 0.327 / 0.329 / 0.363 / 0.443 / 0.697 ms per scan)
    integrate_grid_ = std::min(cdiv(o.num_blocks, 4), 3072);
    if (const char *e = hook_env("DR_INT_GRID")) integrate_grid_ = std::min(65536, std::max(1, atoi(e)));  // tuning hook
    DR_HIP(hipEventCreateWithFlags(&int_done_, hipEventDisableTiming));
    for (int i = 0; i < o.num_render_streams; ++i) {
      Render r;
      DR_HIP(hipStreamCreateWithPriority(&r.stream, hipStreamNonBlocking, lo));
      r.d_bgr = dalloc<unsigned char>(npix_ * 3);
      r.d_depth = dalloc<float>(npix_);
      r.d_flag = dalloc<int>(4);
      for (int k = 0; k < 2; ++k) {  // double-buffered host results ("blocked"/"free", tsdf_volume.cu:846-872)
        DR_HIP(hipHostMalloc((void **)&r.h_bgr[k], npix_ * 3, hipHostMallocDefault));
        DR_HIP(hipHostMalloc((void **)&r.h_depth[k], npix_ * 4, hipHostMallocDefault));
        DR_HIP(hipHostGetDevicePointer((void **)&r.hd_bgr[k], r.h_bgr[k], 0));
        DR_HIP(hipHostGetDevicePointer((void **)&r.hd_depth[k], r.h_depth[k], 0));
      }
      DR_HIP(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
      DR_HIP(hipEventCreateWithFlags(&r.cast, hipEventDisableTiming));
      renders_.push_back(r);
    }
    DR_HIP(hipStreamSynchronize(int_stream_));
  }
  ~FusionEngine() {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    (void)hipFree(d_.keys); (void)hipFree(d_.vals); (void)hipFree(d_.blk_key); (void)hipFree(d_.vox);
    (void)hipFree(d_.n_alloc); (void)hipFree(d_.cnt); (void)hipFree(d_.sd); (void)hipFree(d_.pix); (void)hipFree(d_.super[0]); (void)hipFree(d_.super[1]); (void)hipFree(d_.present); (void)hipFree(d_.grid); (void)hipFree(d_.req); (void)hipFree(d_.req_count); (void)hipFree(d_.vis); (void)hipFree(d_.wg_upd); (void)hipFree(d_bgr_in_); (void)hipFree(d_depth_in_);
    (void)hipHostFree(h_bgr_in_); (void)hipHostFree(h_depth_in_);
    for (auto &r : renders_) {
      (void)hipFree(r.d_bgr); (void)hipFree(r.d_depth); (void)hipFree(r.d_flag);
      for (int k = 0; k < 2; ++k) { (void)hipHostFree(r.h_bgr[k]); (void)hipHostFree(r.h_depth[k]); }
      (void)hipEventDestroy(r.done); (void)hipEventDestroy(r.cast); (void)hipStreamDestroy(r.stream);
    }
    (void)hipEventDestroy(int_done_);
    (void)hipStreamDestroy(int_stream_);
    (void)hipFree(mesh_axis_); (void)hipFree(mesh_keys_); (void)hipFree(mesh_total_); (void)hipFree(mesh_counts_);
    (void)hipFree(mesh_offsets_); (void)hipFree(mesh_tmp_); (void)hipFree(mesh_vert_); (void)hipFree(mesh_cols_);
    if (mesh_done_) (void)hipEventDestroy(mesh_done_);
  }

  // tsdf_volume.cu:515-598
  void integrate_scan_async(const uint8_t *bgr, const float *depth, const float *pose16) {
    if (!bgr || !depth || !pose16) fail(DR_ERR_ARG, "IntegrateScanAsync: null argument");
    expect(kIntegrate, "Please call the functions like Integration -> RenderAsync -> GetRenderResults.");
    next_ = kRender;
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipEventSynchronize(int_done_));  // previous scan's use of the pinned staging buffers
    memcpy(h_bgr_in_, bgr, npix_ * 3);
    memcpy(h_depth_in_, depth, npix_ * 4);
    DR_HIP(hipMemcpyAsync(d_bgr_in_, h_bgr_in_, npix_ * 3, hipMemcpyHostToDevice, int_stream_));
    DR_HIP(hipMemcpyAsync(d_depth_in_, h_depth_in_, npix_ * 4, hipMemcpyHostToDevice, int_stream_));
    // the ray-casts read the volume; their result copies do not (the reference waits for the copies, tsdf_volume.cu:553-556)
    enqueue_scan(d_bgr_in_, d_depth_in_, pose16, true);
    DR_HIP(hipEventRecord(int_done_, int_stream_));
  }
  void launch_raycast(hipStream_t st, unsigned char *d_bgr, float *d_depth, int *d_flag, const Mat &P) {
    const dim3 grid(8 * cdiv(cdiv((int)npix_, 64), 8)), block(64);
#ifdef DR_PARITY_HOOKS
    if (raycast_v1_) { hipLaunchKernelGGL(k_raycast, grid, block, 0, st, d_, P, d_bgr, d_depth); return; }
#endif
    hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, st, d_flag);
    FusionDev dv = d_;
    if (raycast_no_skip_) dv.super[0] = nullptr;  // DR_RAYCAST_NO_SKIP=1: every sample is looked up (A/B and parity hook)
#ifdef DR_PARITY_HOOKS
    if (raycast_stats_ && d_.fast_div) {  // DR_RAYCAST_STATS=1: a synchronous, counting launch of the same loop (prints to stderr)
      if (!d_rstats_) d_rstats_ = dalloc<unsigned long long>(32);
      DR_HIP(hipMemsetAsync(d_rstats_, 0, 32 * 8, st));
      hipLaunchKernelGGL((k_raycast2<true, true>), grid, block, 0, st, dv, P, d_bgr, d_depth, d_flag, d_rstats_);
      unsigned long long h[32];
      DR_HIP(hipMemcpyAsync(h, d_rstats_, sizeof h, hipMemcpyDeviceToHost, st));
      DR_HIP(hipStreamSynchronize(st));
      fprintf(stderr, "raycast stats: waves %llu  iterations/lane %.1f  longest ray %llu  mean wave-longest %.1f | per lane: missing-block samples %.1f, full samples %.1f, skip events %.2f (%.1f steps)\n  wave-longest histogram (x16):",
              h[7], (double)h[0] / (64.0 * h[7]), h[1], (double)h[2] / h[7], (double)h[3] / (64.0 * h[7]), (double)h[6] / (64.0 * h[7]), (double)h[4] / (64.0 * h[7]), (double)h[5] / (64.0 * h[7]));
      for (int i = 0; i < 24; ++i) fprintf(stderr, " %llu", h[8 + i]);
      fprintf(stderr, "\n");
    } else
    if (raycast_sampler_ == 0) {  // DR_RAYCAST_SAMPLER=0: the four-stage sampler of round 2 (parity hook)
      if (d_.fast_div) hipLaunchKernelGGL((k_raycast2<true, false, 0>), grid, block, 0, st, dv, P, d_bgr, d_depth, d_flag, (unsigned long long *)nullptr);
      else hipLaunchKernelGGL((k_raycast2<false, false, 0>), grid, block, 0, st, dv, P, d_bgr, d_depth, d_flag, (unsigned long long *)nullptr);
    } else
#endif
    if (d_.fast_div) hipLaunchKernelGGL((k_raycast2<true, false, 1>), grid, block, 0, st, dv, P, d_bgr, d_depth, d_flag, (unsigned long long *)nullptr);
    else hipLaunchKernelGGL((k_raycast2<false, false, 1>), grid, block, 0, st, dv, P, d_bgr, d_depth, d_flag, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(k_raycast_fix, dim3(512), dim3(64), 0, st, d_, P, d_bgr, d_depth, d_flag);
  }
  // render -> host (k_publish); DR_RENDER_D2H=copy: the two hipMemcpyAsync of round 2 (A/B hook)
  void publish_render(int i) {
    auto &r = renders_[i];
    if (render_copy_) {  // (parity build only)
      DR_HIP(hipMemcpyAsync(r.h_bgr[free_slot_], r.d_bgr, npix_ * 3, hipMemcpyDeviceToHost, r.stream));
      DR_HIP(hipMemcpyAsync(r.h_depth[free_slot_], r.d_depth, npix_ * 4, hipMemcpyDeviceToHost, r.stream));
      return;
    }
    hipLaunchKernelGGL(k_publish, dim3(128), dim3(256), 0, r.stream, (const unsigned char *)r.d_depth, (unsigned char *)r.hd_depth[free_slot_], npix_ * 4,
                       (const unsigned char *)r.d_bgr, r.hd_bgr[free_slot_], npix_ * 3);
  }
  // tsdf_volume.cu:634-700
  void render_async(const float *const *poses, int n) {
    expect(kRender, "Please call the functions like IntegrateScanAsync -> RenderAsync -> GetRenderResult.");
    if (n != (int)renders_.size()) fail(DR_ERR_PROTOCOL, "Can only render exactly as many poses as streams. Streams: %zu, Poses: %d.", renders_.size(), n);
    next_ = kGetRender;
    DR_HIP(hipSetDevice(device_));
    free_slot_ ^= 1;  // write into the buffers NOT handed out by the last GetRenderResult
    for (int i = 0; i < n; ++i) {
      Render &r = renders_[i];
      Mat P; memcpy(P.m, poses[i], 64);
      DR_HIP(hipStreamWaitEvent(r.stream, int_done_, 0));
      launch_raycast(r.stream, r.d_bgr, r.d_depth, r.d_flag, P);
      DR_HIP(hipEventRecord(r.cast, r.stream));
      publish_render(i);
      DR_HIP(hipEventRecord(r.done, r.stream));
    }
  }
  // tsdf_volume.cu:702-737
  void get_render_result(uint8_t **bgr, float **depth, int n) {
    expect(kGetRender, "Please call the functions in a loop: IntegrateScanAsync -> RenderAsync -> GetRenderResult.");
    if (n != (int)renders_.size()) fail(DR_ERR_ARG, "GetRenderResult: expected %zu outputs", renders_.size());
    next_ = kIntegrate;
    DR_HIP(hipSetDevice(device_));
    for (int i = 0; i < n; ++i) {
      DR_HIP(hipEventSynchronize(renders_[i].done));
      bgr[i] = renders_[i].h_bgr[free_slot_];
      depth[i] = renders_[i].h_depth[free_slot_];
    }
    check_device_flags();
  }
  // Device-resident result of render stream i (the buffers GetRenderResult copies from): valid from GetRenderResult
  // until the next RenderAsync.  Lets a consumer on the same GPU (the coarse tracker's dense-depth hand-off) skip the
  // D2H + H2D round trip.
  void get_render_device(int i, const uint8_t **d_bgr, const float **d_depth) {
    if (i < 0 || i >= (int)renders_.size()) fail(DR_ERR_ARG, "get_render_device: stream %d of %zu", i, renders_.size());
    if (next_ != kIntegrate) fail(DR_ERR_PROTOCOL, "get_render_device: call after GetRenderResult");
    if (d_bgr) *d_bgr = renders_[i].d_bgr;
    if (d_depth) *d_depth = renders_[i].d_depth;
  }
  // Test hook: the page-locked host copies of the last (back = 0) and the second-to-last (back = 1) ray-cast of render stream i that
  // bench_sequence wrote -- the second-to-last one ran BESIDE the allocation of the last scan (enqueue_scan), which is what a test of
  // that overlap has to look at.
  void bench_render_host(int i, int back, const uint8_t **bgr, const float **depth) {
    if (i < 0 || i >= (int)renders_.size() || back < 0 || back > 1) fail(DR_ERR_ARG, "bench_render_host: stream %d of %zu, back %d", i, renders_.size(), back);
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipDeviceSynchronize());
    const int slot = free_slot_ ^ back;
    if (bgr) *bgr = renders_[i].h_bgr[slot];
    if (depth) *depth = renders_[i].h_depth[slot];
  }
  void synchronize() {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipDeviceSynchronize());
    check_device_flags();
  }
  void stats(uint64_t out[4]) {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipDeviceSynchronize());
    unsigned long long c[4]; int na[2];
    DR_HIP(hipMemcpy(c, d_.cnt, 32, hipMemcpyDeviceToHost));
    DR_HIP(hipMemcpy(na, d_.n_alloc, 8, hipMemcpyDeviceToHost));
    out[0] = (uint64_t)std::min(na[0], o_.num_blocks); out[1] = c[3]; out[2] = c[1]; out[3] = c[2];
  }
  // blocks k_integrate has read since the engine was created (one 4 KB read each, whether or not any voxel of the block was updated): with
  // `updated_total` this gives the kernel's HBM bytes exactly -- 4096 x visited + 8 x updated -- for the counter calibration in DESIGN.md
  uint64_t visited_blocks() {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    DR_HIP(hipMemcpy(&v, d_.cnt + 4, 8, hipMemcpyDeviceToHost));
    return v;
  }
  void export_blocks(int max_blocks, int32_t *coords, uint8_t *voxels, int *n) {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipDeviceSynchronize());
    int na = 0;
    DR_HIP(hipMemcpy(&na, d_.n_alloc, 4, hipMemcpyDeviceToHost));
    na = std::min(std::min(na, o_.num_blocks), max_blocks);
    std::vector<unsigned long long> keys(na);
    DR_HIP(hipMemcpy(keys.data(), d_.blk_key, (size_t)na * 8, hipMemcpyDeviceToHost));
    const int B = 1 << 20;
    for (int i = 0; i < na; ++i) {
      coords[3 * i] = (int)((keys[i] >> 42) & 0x1fffff) - B;
      coords[3 * i + 1] = (int)((keys[i] >> 21) & 0x1fffff) - B;
      coords[3 * i + 2] = (int)(keys[i] & 0x1fffff) - B;
    }
    DR_HIP(hipMemcpy(voxels, d_.vox, (size_t)na * 4096, hipMemcpyDeviceToHost));
    if (n) *n = na;
  }
  void fast_div_status(int *enabled, unsigned long long *mismatches) const { if (enabled) *enabled = d_.fast_div; if (mismatches) *mismatches = fast_div_mismatches_; }
  void test_combine(size_t n, const uint8_t *a, const uint8_t *b, int max_weight, uint8_t *out) {
    DR_HIP(hipSetDevice(device_));
    Voxel *da = nullptr, *db = nullptr, *dout = nullptr;
    DR_HIP(hipMalloc(&da, n * 8)); DR_HIP(hipMalloc(&db, n * 8)); DR_HIP(hipMalloc(&dout, n * 8));
    DR_HIP(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
    DR_HIP(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_combine, dim3(2048), dim3(256), 0, int_stream_, da, db, dout, n, max_weight);
    DR_HIP(hipStreamSynchronize(int_stream_));
    DR_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
    DR_HIP(hipFree(da)); DR_HIP(hipFree(db)); DR_HIP(hipFree(dout));
  }
  // ---- marching cubes: TsdfVolume::ExtractMeshAsync / GetMeshSync (tsdf_volume.cu:759-838) ----
  void extract_mesh_async(const float *lower, const float *upper) {
    if (!lower || !upper) fail(DR_ERR_ARG, "ExtractMeshAsync: null argument");
    expect(kIntegrate, "Please call this functions after GetRenderResult.");
    if (mesh_pending_) fail(DR_ERR_PROTOCOL, "mesh_extractor should be NULL (fetch the previous mesh with GetMeshSync first)");
    launch_mesh(lower, upper);
    mesh_pending_ = true;
  }
  size_t mesh_num_triangles() {  // blocks until the pending extraction is done; does not consume it
    if (!mesh_pending_) fail(DR_ERR_PROTOCOL, "mesh_extractor should not be NULL (did you call ExtractMeshAsync before)?");
    return finish_mesh();
  }
  // vert / cols hold num_max vertices (3 floats each).  The reference compares num_max with the TRIANGLE count
  // (tsdf_volume.cu:796) and would overrun for meshes above num_max / 3 triangles; here the vertex count is checked.
  void get_mesh_sync(size_t num_max, size_t *num, float *vert, float *cols) {
    expect(kIntegrate, "Please call this functions after GetRenderResult.");
    if (!num || !vert || !cols) fail(DR_ERR_ARG, "GetMeshSync: null argument");
    const size_t ntri = mesh_num_triangles();
    if (num_max < 3 * ntri) fail(DR_ERR_CAPACITY, "Did not provide enough storage for mesh (%zu vertices > %zu).", 3 * ntri, num_max);
    DR_HIP(hipMemcpy(vert, mesh_vert_, ntri * 36, hipMemcpyDeviceToHost));
    DR_HIP(hipMemcpy(cols, mesh_cols_, ntri * 36, hipMemcpyDeviceToHost));
    *num = 3 * ntri;  // 1 triangle = 3 vert (tsdf_volume.cu:800)
    mesh_pending_ = false;
  }
  // DrFusion::SaveMeshToFile (dr_fusion.cpp:74-93): synchronous extraction, then Mesh::SaveToFile(filename, bgr=true)
  // (mesh.cu:24-66): one "v x y z r g b" line per vertex, one "f i i+1 i+2" line per triangle.
  void save_mesh(const char *filename, const float *lower, const float *upper) {
    if (!filename || !lower || !upper) fail(DR_ERR_ARG, "SaveMeshToFile: null argument");
    if (mesh_pending_) fail(DR_ERR_PROTOCOL, "SaveMeshToFile: an ExtractMeshAsync is pending, call GetMeshSync first");
    launch_mesh(lower, upper);
    const size_t ntri = finish_mesh();
    std::vector<float> v(ntri * 9), c(ntri * 9);
    DR_HIP(hipMemcpy(v.data(), mesh_vert_, ntri * 36, hipMemcpyDeviceToHost));
    DR_HIP(hipMemcpy(c.data(), mesh_cols_, ntri * 36, hipMemcpyDeviceToHost));
    FILE *f = fopen(filename, "w");
    if (!f) fail(DR_ERR_IO, "SaveMeshToFile: cannot open %s", filename);
    for (size_t i = 0; i < ntri * 3; ++i)
      fprintf(f, "v %g %g %g %g %g %g\n", v[3 * i], v[3 * i + 1], v[3 * i + 2], c[3 * i], c[3 * i + 1], c[3 * i + 2]);
    for (size_t i = 1; i <= ntri * 3; i += 3) fprintf(f, "f %zu %zu %zu\n", i, i + 1, i + 2);
    if (fclose(f) != 0) fail(DR_ERR_IO, "SaveMeshToFile: write to %s failed", filename);
  }

  // bench path: inputs already resident in HBM
  void integrate_device(const void *d_bgr, const void *d_depth, const float *pose16) {
    DR_HIP(hipSetDevice(device_));
    enqueue_scan((const unsigned char *)d_bgr, (const float *)d_depth, pose16);
  }
  void bench_integrate(const void *d_bgr, const void *d_depth, const float *poses, int nscans, float *ms, float *kernel_ms) {
    DR_HIP(hipSetDevice(device_));
    std::vector<hipEvent_t> ev(2 * (size_t)nscans + 2);
    for (auto &e : ev) DR_HIP(hipEventCreate(&e));
    DR_HIP(hipEventRecord(ev[0], int_stream_));
    for (int s = 0; s < nscans; ++s) {
      kernel_events_[0] = ev[2 + 2 * s]; kernel_events_[1] = ev[3 + 2 * s];
      enqueue_scan((const unsigned char *)d_bgr + (size_t)s * npix_ * 3, (const float *)d_depth + (size_t)s * npix_, poses + 16 * s);
    }
    kernel_events_[0] = kernel_events_[1] = nullptr;
    DR_HIP(hipEventRecord(ev[1], int_stream_));
    DR_HIP(hipStreamSynchronize(int_stream_));
    float t = 0, k = 0;
    DR_HIP(hipEventElapsedTime(&t, ev[0], ev[1]));
    for (int s = 0; s < nscans; ++s) { float q = 0; DR_HIP(hipEventElapsedTime(&q, ev[2 + 2 * s], ev[3 + 2 * s])); k += q; }
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (ms) *ms = t;
    if (kernel_ms) *kernel_ms = k;
    check_device_flags();
  }

  // BASELINE configs[3] loop (dr_debug_example.cpp:78-162: GetRenderResult(k-1) / IntegrateScanAsync(k) / RenderAsync(k) per
  // frame, map growing) over `n` frames whose inputs are resident in HBM: allocate + integrate on the integration stream,
  // then -- render != 0 -- one ray-cast per render stream from the frame's own pose with the D2H of its result into the
  // pinned double buffers, ordered by the same events as the operator path.  ms[0] = first allocate .. last copy
  // (hipEvents), ms[1..4] = sums of the allocate / integrate / ray-cast / D2H intervals, ms[5] = host wall clock.
  void bench_sequence(const void *d_bgr, const void *d_depth, const float *poses, int n, int render, float ms[6]) {
    if (!d_bgr || !d_depth || !poses || !ms || n <= 0) fail(DR_ERR_ARG, "bench_sequence: bad argument");
    expect(kIntegrate, "bench_sequence starts where IntegrateScanAsync may be called.");
    DR_HIP(hipSetDevice(device_));
    const int nr = render ? (int)renders_.size() : 0;
    const size_t per = 4 + (size_t)3 * nr;
    std::vector<hipEvent_t> ev(per * n);
    for (auto &e : ev) DR_HIP(hipEventCreate(&e));
    DR_HIP(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < n; ++s) {
      hipEvent_t *e = &ev[per * s];
      DR_HIP(hipEventRecord(e[0], int_stream_));
      kernel_events_[0] = e[1]; kernel_events_[1] = e[2]; kernel_events_[2] = e[3];
      // (the voxel update waits for the previous frame's ray-casts inside enqueue_scan; allocation, commit and cull run beside them)
      enqueue_scan((const unsigned char *)d_bgr + (size_t)s * npix_ * 3, (const float *)d_depth + (size_t)s * npix_, poses + 16 * s, true);
      kernel_events_[0] = kernel_events_[1] = kernel_events_[2] = nullptr;
      DR_HIP(hipEventRecord(int_done_, int_stream_));
      free_slot_ ^= 1;
      for (int i = 0; i < nr; ++i) {
        Render &r = renders_[i];
        Mat P; memcpy(P.m, poses + 16 * s, 64);
        DR_HIP(hipStreamWaitEvent(r.stream, int_done_, 0));
        DR_HIP(hipEventRecord(e[4 + 3 * i], r.stream));
        launch_raycast(r.stream, r.d_bgr, r.d_depth, r.d_flag, P);
        DR_HIP(hipEventRecord(e[5 + 3 * i], r.stream));
        DR_HIP(hipEventRecord(r.cast, r.stream));
        publish_render(i);
        DR_HIP(hipEventRecord(e[6 + 3 * i], r.stream));
        DR_HIP(hipEventRecord(r.done, r.stream));
      }
    }
    DR_HIP(hipDeviceSynchronize());
    ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int k = 0; k < 5; ++k) ms[k] = 0.f;
    float q = 0.f;
    hipEvent_t last = nr ? ev[per * (n - 1) + 6 + 3 * (nr - 1)] : ev[per * (n - 1) + 2];
    DR_HIP(hipEventElapsedTime(&ms[0], ev[0], last));
    for (int s = 0; s < n; ++s) {
      hipEvent_t *e = &ev[per * s];
      DR_HIP(hipEventElapsedTime(&q, e[0], e[3])); ms[1] += q;  // allocate + commit + cull (may run beside the previous frame's ray-cast)
      DR_HIP(hipEventElapsedTime(&q, e[1], e[2])); ms[2] += q;  // k_integrate alone
      for (int i = 0; i < nr; ++i) {
        DR_HIP(hipEventElapsedTime(&q, e[4 + 3 * i], e[5 + 3 * i])); ms[3] += q;
        DR_HIP(hipEventElapsedTime(&q, e[5 + 3 * i], e[6 + 3 * i])); ms[4] += q;
      }
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    check_device_flags();
  }

 private:
  enum Next { kIntegrate, kRender, kGetRender };
  void expect(Next want, const char *msg) {
    static const char *names[] = {"IntegrateScanAsync", "RenderAsync", "GetRenderResult"};
    if (next_ != want) fail(DR_ERR_PROTOCOL, "%s You should have called %s", msg, names[next_]);
  }
  // allocate -> integrate on int_stream_, no host round trip: the number of allocated blocks lives on the
  // device, so the integrate grid is fixed (a few workgroups per CU) and strides over [0, *n_alloc).
  // wait_renders: order the voxel UPDATE behind the ray-casts of the previous scan (they read the voxels).  The allocation pass, the
  // commit and the cull of this scan do not wait: they only turn absent blocks into present, EMPTY ones (zero voxels, weight 0), write
  // the per-pixel records and the visible list -- a ray that meets such a block mid-flight takes the same step as through an absent one
  // (weight 0 either way; the empty-space skip is conservative in both states), so a ray-cast of scan k that runs beside the allocation
  // of scan k + 1 returns the same image bit for bit, and a quarter of the scan (0.08 of 0.46 ms at 5 mm) leaves the critical path.
  void enqueue_scan(const unsigned char *d_bgr, const float *d_depth, const float *pose16, bool wait_renders = false) {
    Mat T, Ti;
    memcpy(T.m, pose16, 64);
    inverse4_host(T.m, Ti.m);
    hipLaunchKernelGGL(k_allocate, dim3(cdiv((int)npix_, 256)), dim3(256), 0, int_stream_, d_, d_bgr, d_depth, T);
    hipLaunchKernelGGL(k_alloc_commit, dim3(64), dim3(256), 0, int_stream_, d_);
    hipLaunchKernelGGL(k_cull, dim3(512), dim3(256), 0, int_stream_, d_, Ti);
    if (kernel_events_[2]) DR_HIP(hipEventRecord(kernel_events_[2], int_stream_));  // timing hook: end of allocate + commit + cull
    if (wait_renders) for (auto &r : renders_) DR_HIP(hipStreamWaitEvent(int_stream_, r.cast, 0));
    if (kernel_events_[0]) DR_HIP(hipEventRecord(kernel_events_[0], int_stream_));  // timing hooks: [0]..[1] brackets k_integrate alone
    hipLaunchKernelGGL(k_integrate, dim3(integrate_grid_), dim3(256), 0, int_stream_, d_, d_bgr, d_depth, T, Ti);
    if (kernel_events_[1]) DR_HIP(hipEventRecord(kernel_events_[1], int_stream_));
    hipLaunchKernelGGL(k_fold_counter, dim3(1), dim3(256), 0, int_stream_, d_.cnt, d_.req_count, d_.wg_upd, integrate_grid_);
    DR_HIP(hipGetLastError());
  }
  static int f2i_host(float f) {  // make_int3(float...) on CUDA: cvt.rzi (saturating, NaN -> 0)
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
  }
  // Enqueue the whole extraction on int_stream_ (after the last integration, before the next one).
  void launch_mesh(const float *lower, const float *upper) {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(int_stream_));  // the block count lives on the device
    int na = 0;
    DR_HIP(hipMemcpy(&na, d_.n_alloc, 4, hipMemcpyDeviceToHost));
    const int nblk = std::min(na, o_.num_blocks);
    const float vs = o_.voxel_size;
    int n[3];
    size_t ntab = 0;
    for (int a = 0; a < 3; ++a) {  // ExtractMeshKernel, mesh_extractor.cu:241-245
      n[a] = f2i_host(fabsf(lower[a] - upper[a]) / vs);
      if (n[a] > (1 << 22)) fail(DR_ERR_ARG, "ExtractMesh: %d lattice cells along axis %d (box too large for voxel_size %g)", n[a], a, vs);
      ntab += (size_t)std::max(n[a], 0);
    }
    if (!mesh_total_) mesh_total_ = dalloc<unsigned long long>(1);
    if (nblk <= 0 || n[0] <= 0 || n[1] <= 0 || n[2] <= 0) {
      DR_HIP(hipMemsetAsync(mesh_total_, 0, 8, int_stream_));
      DR_HIP(hipEventRecord(mesh_done_ev(), int_stream_));
      return;
    }
    if (ntab > mesh_axis_cap_) {
      if (mesh_axis_) DR_HIP(hipFree(mesh_axis_));
      mesh_axis_ = dalloc<McAxis>(ntab);
      mesh_axis_cap_ = ntab;
    }
    if (!mesh_keys_) {
      mesh_keys_ = dalloc<unsigned long long>(o_.num_blocks);
      mesh_counts_ = dalloc<unsigned>(o_.num_blocks);
      mesh_offsets_ = dalloc<unsigned>(o_.num_blocks);
      size_t t1 = 0, t2 = 0;
      DR_HIP(rocprim::radix_sort_keys(nullptr, t1, d_.blk_key, mesh_keys_, (size_t)o_.num_blocks, 0, 63, int_stream_));
      DR_HIP(rocprim::exclusive_scan(nullptr, t2, mesh_counts_, mesh_offsets_, 0u, (size_t)o_.num_blocks, rocprim::plus<unsigned>(), int_stream_));
      mesh_tmp_bytes_ = std::max(t1, t2);
      mesh_tmp_ = dalloc<unsigned char>(mesh_tmp_bytes_);
      // the reference's MeshExtractor::Init(20000000, ...) (tsdf_volume.cu:776): 72 B per triangle, 1.44 GB of HBM
      mesh_vert_ = dalloc<float>((size_t)kMeshMaxTriangles * 9);
      mesh_cols_ = dalloc<float>((size_t)kMeshMaxTriangles * 9);
    }
    McArgs a{};
    McAxis *p = mesh_axis_;
    for (int k = 0; k < 3; ++k) {
      hipLaunchKernelGGL(k_mc_axes, dim3(cdiv(n[k], 256)), dim3(256), 0, int_stream_, p, n[k], lower[k], vs);
      a.ax[k] = p; a.n[k] = n[k];
      p += n[k];
    }
    size_t tb = mesh_tmp_bytes_;
    DR_HIP(rocprim::radix_sort_keys(mesh_tmp_, tb, d_.blk_key, mesh_keys_, (size_t)nblk, 0, 63, int_stream_));
    a.sorted_keys = mesh_keys_; a.nblk = nblk; a.counts = mesh_counts_; a.offsets = mesh_offsets_;
    a.vert = mesh_vert_; a.cols = mesh_cols_; a.cap_tri = kMeshMaxTriangles;
    hipLaunchKernelGGL((k_mc_cells<false>), dim3(nblk), dim3(256), 0, int_stream_, d_, a);
    tb = mesh_tmp_bytes_;
    DR_HIP(rocprim::exclusive_scan(mesh_tmp_, tb, mesh_counts_, mesh_offsets_, 0u, (size_t)nblk, rocprim::plus<unsigned>(), int_stream_));
    hipLaunchKernelGGL((k_mc_cells<true>), dim3(nblk), dim3(256), 0, int_stream_, d_, a);
    hipLaunchKernelGGL(k_mc_total, dim3(1), dim3(1), 0, int_stream_, mesh_counts_, mesh_offsets_, nblk, mesh_total_);
    DR_HIP(hipGetLastError());
    DR_HIP(hipEventRecord(mesh_done_ev(), int_stream_));
  }
  hipEvent_t mesh_done_ev() {
    if (!mesh_done_) DR_HIP(hipEventCreateWithFlags(&mesh_done_, hipEventDisableTiming));
    return mesh_done_;
  }
  size_t finish_mesh() {
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipEventSynchronize(mesh_done_ev()));
    unsigned long long t = 0;
    DR_HIP(hipMemcpy(&t, mesh_total_, 8, hipMemcpyDeviceToHost));
    if (t > kMeshMaxTriangles) fail(DR_ERR_CAPACITY, "Triangles limit reached! (%llu > %u)", t, kMeshMaxTriangles);
    return (size_t)t;
  }
  // Correctly rounded reciprocals of the three per-engine divisors, each checked against IEEE division over all 2^32
  // dividends (see div_exact); DR_FUSION_IEEE_DIV=1 forces the IEEE sequences (A/B and fallback testing).
  static float rcp_rn(float b) { return (float)(1.0 / (double)b); }  // double rounding cannot bite: 1/b is never within 2^-49 of a float midpoint
  void setup_fast_div() {
    d_.vs_rcp = rcp_rn(o_.voxel_size); d_.fx_rcp = rcp_rn(o_.fx); d_.fy_rcp = rcp_rn(o_.fy);
    d_.fast_div = 0;
    if (hook_env("DR_FUSION_IEEE_DIV")) return;
    unsigned long long *bad = dalloc<unsigned long long>(1), h = 0;
    DR_HIP(hipMemsetAsync(bad, 0, 8, int_stream_));
    const float b[3] = {o_.voxel_size, o_.fx, o_.fy}, y[3] = {d_.vs_rcp, d_.fx_rcp, d_.fy_rcp};
    for (int k = 0; k < 3; ++k)
      if (b[k] > 0.0f && b[k] < 1e30f) hipLaunchKernelGGL(k_verify_fast_div, dim3(4096), dim3(256), 0, int_stream_, b[k], y[k], bad);
      else h = 1;
    DR_HIP(hipMemcpyAsync(&fast_div_mismatches_, bad, 8, hipMemcpyDeviceToHost, int_stream_));
    DR_HIP(hipStreamSynchronize(int_stream_));
    DR_HIP(hipFree(bad));
    fast_div_mismatches_ += h;
    d_.fast_div = fast_div_mismatches_ == 0 ? 1 : 0;
  }
  void check_device_flags() {
    int f[4];
    DR_HIP(hipMemcpy(f, d_.n_alloc, 16, hipMemcpyDeviceToHost));
    if (f[1] || f[2]) fail(DR_ERR_CAPACITY, "DrFusion: block pool exhausted (num_blocks=%d) or block coordinate out of range", o_.num_blocks);
  }

  struct Render {
    hipStream_t stream;
    unsigned char *d_bgr, *h_bgr[2], *hd_bgr[2];   // hd_*: the pinned result buffers as the device addresses them
    float *d_depth, *h_depth[2], *hd_depth[2];
    int *d_flag;  // pixels the fast ray-caster handed to the literal pass
    hipEvent_t done, cast;  // result on the host / ray-cast kernels finished (the volume may be written again)
  };
  int device_;
  drf_options_t o_;
  FusionDev d_{};
  size_t npix_ = 0;
  hipStream_t int_stream_ = nullptr;
  hipEvent_t int_done_ = nullptr;
  hipEvent_t kernel_events_[3] = {nullptr, nullptr, nullptr};
  unsigned char *d_bgr_in_ = nullptr, *h_bgr_in_ = nullptr;
  float *d_depth_in_ = nullptr, *h_depth_in_ = nullptr;
  int integrate_grid_ = 4096;
  unsigned long long fast_div_mismatches_ = 0;
  // switches, read once here.  Product: the empty-space skip's A/B switch (a run-time flag of the same kernel) and DR_FUSION_IEEE_DIV (the
  // IEEE-division instances are the product's fallback when the exact fast division fails its check).  Parity build (-DDR_PARITY_HOOKS):
  // the superseded generations -- the literal ray-caster, round 2's four-stage sampler, copy-engine result transfers, statistics.
  bool raycast_no_skip_ = hook_env("DR_RAYCAST_NO_SKIP") != nullptr;
#ifdef DR_PARITY_HOOKS
  bool raycast_v1_ = hook_env("DR_RAYCAST_V1") != nullptr;
  int raycast_sampler_ = hook_env("DR_RAYCAST_SAMPLER") ? std::max(0, std::min(1, atoi(hook_env("DR_RAYCAST_SAMPLER")))) : (getenv("DR_RAYCAST_UNSTAGED") ? 0 : 1);
  bool render_copy_ = hook_env("DR_RENDER_D2H") && !strcmp(hook_env("DR_RENDER_D2H"), "copy");
  bool raycast_stats_ = hook_env("DR_RAYCAST_STATS") != nullptr;     // measuring hook: iteration statistics of k_raycast2 on stderr
  unsigned long long *d_rstats_ = nullptr;
#else
  static constexpr bool render_copy_ = false;
#endif
  std::vector<Render> renders_;
  int free_slot_ = 0;
  Next next_ = kIntegrate;
  // mesh extraction state (allocated with the first ExtractMeshAsync)
  static constexpr unsigned kMeshMaxTriangles = 20000000;
  bool mesh_pending_ = false;
  hipEvent_t mesh_done_ = nullptr;
  McAxis *mesh_axis_ = nullptr;
  size_t mesh_axis_cap_ = 0, mesh_tmp_bytes_ = 0;
  unsigned long long *mesh_keys_ = nullptr, *mesh_total_ = nullptr;
  unsigned *mesh_counts_ = nullptr, *mesh_offsets_ = nullptr;
  unsigned char *mesh_tmp_ = nullptr;
  float *mesh_vert_ = nullptr, *mesh_cols_ = nullptr;
};

}  // namespace dr

// ==================================================================== C ABI
using dr::guarded;
struct drf_s {
  std::unique_ptr<dr::FusionEngine> e;
};
// every C-ABI entry point goes through this: a NULL handle is an argument error, not a crash
static inline dr::FusionEngine *eng(drf_s *h) {
  if (!h || !h->e) dr::fail(DR_ERR_ARG, "NULL handle");
  return h->e.get();
}

extern "C" {

int drf_create(const drf_options_t *opt, int device, drf_t **out) {
  return guarded([&] {
    if (!opt || !out) dr::fail(DR_ERR_ARG, "drf_create: null argument");
    auto *h = new drf_s();
    try { h->e.reset(new dr::FusionEngine(*opt, device)); } catch (...) { delete h; throw; }
    *out = h;
  });
}
void drf_destroy(drf_t *h) { delete h; }
int drf_integrate_scan_async(drf_t *h, const uint8_t *bgr, const float *depth, const float *pose16) {
  return guarded([&] { eng(h)->integrate_scan_async(bgr, depth, pose16); });
}
int drf_render_async(drf_t *h, const float *const *poses16, int n) { return guarded([&] { eng(h)->render_async(poses16, n); }); }
int drf_get_render_result(drf_t *h, uint8_t **bgr, float **depth, int n) { return guarded([&] { eng(h)->get_render_result(bgr, depth, n); }); }
int drf_extract_mesh_async(drf_t *h, const float *lower, const float *upper) {
  return guarded([&] { eng(h)->extract_mesh_async(lower, upper); });
}
int drf_get_mesh_sync(drf_t *h, size_t num_max, size_t *num, float *vert, float *cols) {
  return guarded([&] { eng(h)->get_mesh_sync(num_max, num, vert, cols); });
}
int drf_mesh_num_triangles(drf_t *h, size_t *ntri) {
  return guarded([&] { if (!ntri) dr::fail(DR_ERR_ARG, "drf_mesh_num_triangles: null argument"); *ntri = eng(h)->mesh_num_triangles(); });
}
int drf_save_mesh(drf_t *h, const char *filename, const float *lower, const float *upper) {
  return guarded([&] { eng(h)->save_mesh(filename, lower, upper); });
}
int drf_get_render_device(drf_t *h, int stream, const uint8_t **d_bgr, const float **d_depth) {
  return guarded([&] { eng(h)->get_render_device(stream, d_bgr, d_depth); });
}
int drf_synchronize(drf_t *h) { return guarded([&] { eng(h)->synchronize(); }); }
int drf_stats(drf_t *h, uint64_t out[4]) { return guarded([&] { eng(h)->stats(out); }); }
int drf_export_blocks(drf_t *h, int max_blocks, int32_t *coords, uint8_t *voxels, int *n) {
  return guarded([&] { eng(h)->export_blocks(max_blocks, coords, voxels, n); });
}
int drf_fast_div_status(drf_t *h, int *enabled, uint64_t *mismatches) {
  return guarded([&] { unsigned long long m = 0; eng(h)->fast_div_status(enabled, &m); if (mismatches) *mismatches = m; });
}
int drf_test_combine(drf_t *h, size_t n, const uint8_t *a, const uint8_t *b, int max_weight, uint8_t *out) {
  return guarded([&] { if (!a || !b || !out) dr::fail(DR_ERR_ARG, "drf_test_combine: null argument"); eng(h)->test_combine(n, a, b, max_weight, out); });
}
int drf_integrate_device(drf_t *h, const void *d_bgr, const void *d_depth, const float *pose16) {
  return guarded([&] { eng(h)->integrate_device(d_bgr, d_depth, pose16); });
}
int dr_device_alloc(int device, size_t bytes, void **dptr) {
  return guarded([&] { DR_HIP(hipSetDevice(device)); DR_HIP(hipMalloc(dptr, bytes)); });
}
int dr_device_free(void *dptr) { return guarded([&] { DR_HIP(hipFree(dptr)); }); }
int dr_memcpy_h2d(void *dptr, const void *src, size_t bytes) { return guarded([&] { DR_HIP(hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice)); }); }
int dr_memcpy_d2d(void *dst, const void *src, size_t bytes) {
  // a device-to-device hipMemcpy may return before the copy has run, and the engines' streams are non-blocking:
  // synchronise so that whatever the caller enqueues next (on any stream) sees the data
  return guarded([&] {
    hipPointerAttribute_t at;  // synchronise the device that owns the destination, not whichever is current on this thread --
    int prev = -1;             // and leave the caller's current device as it was (a helper must not move a host thread between GPUs)
    (void)hipGetDevice(&prev);
    const bool moved = hipPointerGetAttributes(&at, dst) == hipSuccess && at.device != prev;
    if (moved) DR_HIP(hipSetDevice(at.device));
    const hipError_t e1 = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice);
    const hipError_t e2 = e1 == hipSuccess ? hipDeviceSynchronize() : e1;
    if (moved && prev >= 0) (void)hipSetDevice(prev);
    DR_HIP(e2);
  });
}
int dr_memcpy_d2h(void *dst, const void *dptr, size_t bytes) { return guarded([&] { DR_HIP(hipMemcpy(dst, dptr, bytes, hipMemcpyDeviceToHost)); }); }
int drf_bench_sequence(drf_t *h, const void *d_bgr, const void *d_depth, const float *poses16, int nframes, int render, float ms[6]) {
  return guarded([&] { eng(h)->bench_sequence(d_bgr, d_depth, poses16, nframes, render, ms); });
}
int drf_visited_blocks(drf_t *h, uint64_t *total) {
  return guarded([&] { if (!total) dr::fail(DR_ERR_ARG, "drf_visited_blocks: null pointer"); *total = eng(h)->visited_blocks(); });
}
int drf_bench_render_host(drf_t *h, int stream, int back, const uint8_t **bgr, const float **depth) {
  return guarded([&] { eng(h)->bench_render_host(stream, back, bgr, depth); });
}
int drf_bench_integrate(drf_t *h, const void *d_bgr, const void *d_depth, const float *poses16, int nscans, float *ms, float *kernel_ms) {
  return guarded([&] { eng(h)->bench_integrate(d_bgr, d_depth, poses16, nscans, ms, kernel_ms); });
}

}  // extern "C"
