/* tsdf_oracle.c -- CPU ORACLE for the DrFusion hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * A plain-C, single-threaded restatement of the reference's hashed-voxel TSDF
 * (tandem/libdr/dr_fusion/src, a ReFusion derivative), function by function:
 *   Voxel / Combine                 tsdfvh/voxel.h:13-53
 *   block-local voxel index         tsdfvh/voxel_block.h:37-41        (x*bs*bs + y*bs + z)
 *   World<->voxel/block maps        tsdfvh/tsdf_volume.cu:103-145
 *   GetVoxel / GetInterpolatedVoxel tsdfvh/tsdf_volume.cu:147-289
 *   UpdateVoxel                     tsdfvh/tsdf_volume.cu:303-315
 *   AllocateFromDepthKernel         tsdfvh/tsdf_volume.cu:317-434
 *   IntegrateScanKernel             tsdfvh/tsdf_volume.cu:436-513
 *   GenerateRgbDepthKernel          tsdfvh/tsdf_volume.cu:600-632
 *   ExtractMeshKernel & helpers     marching_cubes/mesh_extractor.cu:24-265, GetMeshSync tsdf_volume.cu:781-838
 *   GetPoint3d / Project / norm ... utils/utils.h:44-108
 *   float4x4 * float3, getInverse   utils/matrix_utils.h:914-922, :958-1083
 *
 * PARITY UNPINNED BY THE REFERENCE: dr_fusion has no tests, no fixtures and no golden vectors
 * (SURVEY.md section 4 / 8c), and its CUDA sources cannot be built here (cudaMallocManaged,
 * nvcc-only).  This restatement therefore DEFINES the canonical result the HIP path is held to:
 *   (1) allocated set = every block the DDA of every valid pixel visits (no bucket overflow: the
 *       reference's try-lock insert that silently drops contended inserts, hash_table.cu:103-114,
 *       is not reproduced);
 *   (2) state = map block coordinate -> 512 voxels; hash-entry slot / heap pointer are not state;
 *   (3) integration visits allocated blocks only (the reference walks all 10 M hash entries and
 *       aliases every free one onto block (0,0,0), tsdf_volume.cu:451-455 -- a defect we do not inherit);
 *   (4) float->int conversions follow CUDA's cvt.rzi (saturating, NaN -> 0), since the reference
 *       runs on CUDA and uses out-of-range conversions in Project();
 *   (5) DDA walks are capped at ORACLE_MAX_DDA steps (the reference loops forever on overshoot);
 *   (6) a mesh is a SET of triangles: the reference appends with atomicAdd (mesh.cu:19-22), so its order is
 *       arbitrary; this restatement emits in lattice order and tests compare sorted triangle lists.
 * All arithmetic is fp32 in the reference's expression order; build with -ffp-contract=off.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../tandem_amd/csrc/mc_tables.h" /* Paul Bourke's public-domain triangle table, shared constants */

#define ORACLE_MAX_DDA 4096

typedef struct {
  float voxel_size;
  int num_buckets, bucket_size, num_blocks, block_size, max_sdf_weight;
  float truncation_distance, max_sensor_depth, min_sensor_depth;
  int num_render_streams;
  float fx, fy, cx, cy;
  int height, width;
} tsdf_options; /* == DrFusionOptions, dr_fusion.h:18-36 */

typedef struct { float sdf; unsigned char c[3]; unsigned char weight; } voxel_t; /* voxel.h:13-19: 8 bytes */
typedef struct { float x, y, z; } f3;
typedef struct { int x, y, z; } i3;

typedef struct {
  tsdf_options o;
  int nblk, cap_blk;      /* allocated blocks */
  i3 *coord;              /* [cap_blk] */
  voxel_t *vox;           /* [cap_blk * bs^3] */
  int *table;             /* open-addressing map: slot -> block index or -1 */
  unsigned tmask;
  unsigned long long updated_last, updated_total, mismatches;
  unsigned long long dda_capped; /* rays whose DDA ran into ORACLE_MAX_DDA (the reference would not terminate) */
  int overflow;
} tsdf_t;

/* ---- CUDA conversion semantics ---- */
static int f2i(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return INT_MAX;
  if (f <= -2147483648.0f) return INT_MIN;
  return (int)f;
}
static unsigned char f2u8(float f) {
  if (!(f > 0.0f)) return 0;
  if (f >= 255.0f) return 255;
  return (unsigned char)f;
}

/* ---- utils.h:44-108 ---- */
static float norm3(f3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static f3 sub3(f3 a, f3 b) { f3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static f3 add3(f3 a, f3 b) { f3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static f3 mul3(f3 a, float b) { f3 r = {a.x * b, a.y * b, a.z * b}; return r; }
static int signi(float n) { return (n > 0) - (n < 0); }
static float signf_(float n) { return (float)((n > 0) - (n < 0)); }
static f3 xform(const float *m, f3 v) { /* matrix_utils.h:914-922, row-major m[16] */
  f3 r;
  r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 1.0f;
  r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 1.0f;
  r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 1.0f;
  return r;
}
static f3 point3d(const tsdf_options *o, int i, float depth) { /* utils.h:93-101 */
  int v = i / o->width, u = i - o->width * v;
  f3 p;
  p.z = depth;
  p.x = ((float)u - o->cx) * p.z / o->fx;
  p.y = ((float)v - o->cy) * p.z / o->fy;
  return p;
}
static void project(const tsdf_options *o, f3 p, int *px, int *py) { /* utils.h:103-108 */
  float x = (o->fx * p.x) / p.z + o->cx;
  float y = (o->fy * p.y) / p.z + o->cy;
  *px = f2i(roundf(x));
  *py = f2i(roundf(y));
}
/* cofactor inverse in the reference's term order (matrix_utils.h:958-1083): inv = adj * (1/det) */
static void inverse4(const float *e, float *out) {
  float inv[16];
#define T3(a, b, c) (e[a] * e[b] * e[c])
  inv[0] = T3(5, 10, 15) - T3(5, 11, 14) - T3(9, 6, 15) + T3(9, 7, 14) + T3(13, 6, 11) - T3(13, 7, 10);
  inv[4] = -T3(4, 10, 15) + T3(4, 11, 14) + T3(8, 6, 15) - T3(8, 7, 14) - T3(12, 6, 11) + T3(12, 7, 10);
  inv[8] = T3(4, 9, 15) - T3(4, 11, 13) - T3(8, 5, 15) + T3(8, 7, 13) + T3(12, 5, 11) - T3(12, 7, 9);
  inv[12] = -T3(4, 9, 14) + T3(4, 10, 13) + T3(8, 5, 14) - T3(8, 6, 13) - T3(12, 5, 10) + T3(12, 6, 9);
  inv[1] = -T3(1, 10, 15) + T3(1, 11, 14) + T3(9, 2, 15) - T3(9, 3, 14) - T3(13, 2, 11) + T3(13, 3, 10);
  inv[5] = T3(0, 10, 15) - T3(0, 11, 14) - T3(8, 2, 15) + T3(8, 3, 14) + T3(12, 2, 11) - T3(12, 3, 10);
  inv[9] = -T3(0, 9, 15) + T3(0, 11, 13) + T3(8, 1, 15) - T3(8, 3, 13) - T3(12, 1, 11) + T3(12, 3, 9);
  inv[13] = T3(0, 9, 14) - T3(0, 10, 13) - T3(8, 1, 14) + T3(8, 2, 13) + T3(12, 1, 10) - T3(12, 2, 9);
  inv[2] = T3(1, 6, 15) - T3(1, 7, 14) - T3(5, 2, 15) + T3(5, 3, 14) + T3(13, 2, 7) - T3(13, 3, 6);
  inv[6] = -T3(0, 6, 15) + T3(0, 7, 14) + T3(4, 2, 15) - T3(4, 3, 14) - T3(12, 2, 7) + T3(12, 3, 6);
  inv[10] = T3(0, 5, 15) - T3(0, 7, 13) - T3(4, 1, 15) + T3(4, 3, 13) + T3(12, 1, 7) - T3(12, 3, 5);
  inv[14] = -T3(0, 5, 14) + T3(0, 6, 13) + T3(4, 1, 14) - T3(4, 2, 13) - T3(12, 1, 6) + T3(12, 2, 5);
  inv[3] = -T3(1, 6, 11) + T3(1, 7, 10) + T3(5, 2, 11) - T3(5, 3, 10) - T3(9, 2, 7) + T3(9, 3, 6);
  inv[7] = T3(0, 6, 11) - T3(0, 7, 10) - T3(4, 2, 11) + T3(4, 3, 10) + T3(8, 2, 7) - T3(8, 3, 6);
  inv[11] = -T3(0, 5, 11) + T3(0, 7, 9) + T3(4, 1, 11) - T3(4, 3, 9) - T3(8, 1, 7) + T3(8, 3, 5);
  inv[15] = T3(0, 5, 10) - T3(0, 6, 9) - T3(4, 1, 10) + T3(4, 2, 9) + T3(8, 1, 6) - T3(8, 2, 5);
#undef T3
  float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
  float detr = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}
void tsdf_inverse4(const float *e, float *out) { inverse4(e, out); }

/* ---- block map (implementation detail of the oracle, not reference state) ---- */
static unsigned hash_i3(i3 p) {
  unsigned h = (unsigned)p.x * 73856093u ^ (unsigned)p.y * 19349669u ^ (unsigned)p.z * 83492791u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return h;
}
static int find_block(const tsdf_t *t, i3 p) {
  unsigned s = hash_i3(p) & t->tmask;
  for (;;) {
    int b = t->table[s];
    if (b < 0) return -1;
    if (t->coord[b].x == p.x && t->coord[b].y == p.y && t->coord[b].z == p.z) return b;
    s = (s + 1) & t->tmask;
  }
}
static void allocate_block(tsdf_t *t, i3 p) { /* hash_table.cu:80-115 without the contention drop */
  unsigned s = hash_i3(p) & t->tmask;
  for (;;) {
    int b = t->table[s];
    if (b < 0) break;
    if (t->coord[b].x == p.x && t->coord[b].y == p.y && t->coord[b].z == p.z) return;
    s = (s + 1) & t->tmask;
  }
  if (t->nblk >= t->cap_blk) { t->overflow = 1; return; }
  t->table[s] = t->nblk;
  t->coord[t->nblk] = p;
  t->nblk++; /* voxels are zero-initialised at creation (hash_table.cu:28-32) */
}

/* ---- coordinate maps, tsdf_volume.cu:109-145 ---- */
static i3 world_to_global_voxel(const tsdf_t *t, f3 p) {
  float vs = t->o.voxel_size;
  i3 r = {f2i(p.x / vs + signf_(p.x) * 0.5f), f2i(p.y / vs + signf_(p.y) * 0.5f), f2i(p.z / vs + signf_(p.z) * 0.5f)};
  return r;
}
static int fdiv_block(int v, int bs) { return v < 0 ? (v - bs + 1) / bs : v / bs; }
static i3 world_to_block(const tsdf_t *t, f3 p) {
  i3 v = world_to_global_voxel(t, p);
  int bs = t->o.block_size;
  i3 r = {fdiv_block(v.x, bs), fdiv_block(v.y, bs), fdiv_block(v.z, bs)};
  return r;
}
static int pmod(int v, int bs) { int r = v % bs; return r < 0 ? r + bs : r; }
static i3 world_to_local_voxel(const tsdf_t *t, f3 p) {
  i3 v = world_to_global_voxel(t, p);
  int bs = t->o.block_size;
  i3 r = {pmod(v.x, bs), pmod(v.y, bs), pmod(v.z, bs)};
  return r;
}
static int local_index(const tsdf_t *t, i3 l) { int bs = t->o.block_size; return l.x * bs * bs + l.y * bs + l.z; }
static int nvox(const tsdf_t *t) { int bs = t->o.block_size; return bs * bs * bs; }

static voxel_t get_voxel(const tsdf_t *t, f3 p) { /* tsdf_volume.cu:147-160 */
  voxel_t z = {0.0f, {0, 0, 0}, 0};
  int b = find_block(t, world_to_block(t, p));
  if (b < 0) return z;
  return t->vox[(size_t)b * nvox(t) + local_index(t, world_to_local_voxel(t, p))];
}

static voxel_t get_interpolated_voxel(const tsdf_t *t, f3 pos) { /* tsdf_volume.cu:161-289 */
  voxel_t v0 = get_voxel(t, pos);
  if (v0.weight == 0) return v0;
  float vs = t->o.voxel_size;
  f3 half = {vs / 2.0f, vs / 2.0f, vs / 2.0f};
  f3 pd = sub3(pos, half);
  f3 vp = {pos.x / vs, pos.y / vs, pos.z / vs};
  f3 w = {vp.x - floorf(vp.x), vp.y - floorf(vp.y), vp.z - floorf(vp.z)};
  float dist = 0.0f;
  f3 col = {0.0f, 0.0f, 0.0f};
  /* corner order and weight association exactly as the reference writes them */
  static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 1}};
  voxel_t v = v0;
  for (int k = 0; k < 8; ++k) {
    f3 off = {corner[k][0] ? vs : 0.0f, corner[k][1] ? vs : 0.0f, corner[k][2] ? vs : 0.0f};
    v = get_voxel(t, add3(pd, off));
    float a = corner[k][0] ? w.x : (1.0f - w.x);
    float b = corner[k][1] ? w.y : (1.0f - w.y);
    float c = corner[k][2] ? w.z : (1.0f - w.z);
    float wt = a * b * c;
    const voxel_t *src = v.weight == 0 ? &v0 : &v;
    f3 vc = {(float)src->c[0], (float)src->c[1], (float)src->c[2]};
    dist += wt * src->sdf;
    col.x = col.x + vc.x * wt;
    col.y = col.y + vc.y * wt;
    col.z = col.z + vc.z * wt;
  }
  v.c[0] = f2u8(col.x); v.c[1] = f2u8(col.y); v.c[2] = f2u8(col.z);
  v.weight = v0.weight;
  v.sdf = dist;
  return v;
}

static void combine(voxel_t *a, const voxel_t *b, unsigned char max_weight) { /* voxel.h:21-50 */
  float w = (float)a->weight, vw = (float)b->weight;
  for (int k = 0; k < 3; ++k) a->c[k] = f2u8(((float)a->c[k] * w + (float)b->c[k] * vw) / (w + vw));
  a->sdf = (a->sdf * w + b->sdf * vw) / (w + vw);
  unsigned char nw = (unsigned char)(a->weight + b->weight);
  if (nw > max_weight) nw = max_weight;
  a->weight = nw;
}

/* ---- public API ---- */
tsdf_t *tsdf_create(const tsdf_options *o) {
  tsdf_t *t = (tsdf_t *)calloc(1, sizeof(tsdf_t));
  t->o = *o;
  t->cap_blk = o->num_blocks;
  t->coord = (i3 *)malloc(sizeof(i3) * (size_t)t->cap_blk);
  t->vox = (voxel_t *)calloc((size_t)t->cap_blk * nvox(t), sizeof(voxel_t));
  unsigned cap = 1024;
  while (cap < 2u * (unsigned)t->cap_blk) cap <<= 1;
  t->tmask = cap - 1;
  t->table = (int *)malloc(sizeof(int) * cap);
  for (unsigned i = 0; i < cap; ++i) t->table[i] = -1;
  return t;
}
void tsdf_destroy(tsdf_t *t) {
  if (!t) return;
  free(t->coord); free(t->vox); free(t->table); free(t);
}

/* AllocateFromDepthKernel, tsdf_volume.cu:317-434 */
static void allocate_from_depth(tsdf_t *t, const float *depth, const float *T) {
  const tsdf_options *o = &t->o;
  float trunc = o->truncation_distance;
  float bsz = o->block_size * o->voxel_size;
  f3 start = {T[3], T[7], T[11]};
  int size = o->height * o->width;
  for (int i = 0; i < size; ++i) {
    if (depth[i] < o->min_sensor_depth || depth[i] > o->max_sensor_depth) continue;
    f3 pu = point3d(o, i, depth[i]);
    f3 point = xform(T, pu);
    if (point.x == 0 && point.y == 0 && point.z == 0) continue;
    f3 d = sub3(point, start);
    float dn = norm3(d);
    f3 dir = {d.x / dn, d.y / dn, d.z / dn};
    float surf = norm3(sub3(point, start));
    f3 rs = start;
    f3 re = add3(start, mul3(dir, surf + trunc));
    i3 bp = {f2i(floorf(rs.x / bsz)), f2i(floorf(rs.y / bsz)), f2i(floorf(rs.z / bsz))};
    i3 be = {f2i(floorf(re.x / bsz)), f2i(floorf(re.y / bsz)), f2i(floorf(re.z / bsz))};
    i3 st = {signi(dir.x), signi(dir.y), signi(dir.z)};
    f3 dt;
    dt.x = (dir.x != 0) ? fabsf(bsz / dir.x) : FLT_MAX;
    dt.y = (dir.y != 0) ? fabsf(bsz / dir.y) : FLT_MAX;
    dt.z = (dir.z != 0) ? fabsf(bsz / dir.z) : FLT_MAX;
    f3 bd = {(bp.x + (float)st.x) * bsz, (bp.y + (float)st.y) * bsz, (bp.z + (float)st.z) * bsz};
    f3 mt;
    mt.x = (dir.x != 0) ? (bd.x - rs.x) / dir.x : FLT_MAX;
    mt.y = (dir.y != 0) ? (bd.y - rs.y) / dir.y : FLT_MAX;
    mt.z = (dir.z != 0) ? (bd.z - rs.z) / dir.z : FLT_MAX;
    i3 diff = {0, 0, 0};
    int neg = 0;
    if (bp.x != be.x && dir.x < 0) { diff.x--; neg = 1; }
    if (bp.y != be.y && dir.y < 0) { diff.y--; neg = 1; }
    if (bp.z != be.z && dir.z < 0) { diff.z--; neg = 1; }
    allocate_block(t, bp);
    if (neg) { bp.x += diff.x; bp.y += diff.y; bp.z += diff.z; allocate_block(t, bp); }
    int steps = 0;
    while ((bp.x != be.x || bp.y != be.y || bp.z != be.z) && steps++ < ORACLE_MAX_DDA) {
      if (mt.x < mt.y) {
        if (mt.x < mt.z) { bp.x += st.x; mt.x += dt.x; } else { bp.z += st.z; mt.z += dt.z; }
      } else {
        if (mt.y < mt.z) { bp.y += st.y; mt.y += dt.y; } else { bp.z += st.z; mt.z += dt.z; }
      }
      allocate_block(t, bp);
    }
    if (steps > ORACLE_MAX_DDA) t->dda_capped++;
  }
}

/* UpdateVoxel, tsdf_volume.cu:303-315 (position-keyed, exactly as the reference re-hashes it) */
static int update_voxel(tsdf_t *t, f3 wp, const voxel_t *v, int from_block, int from_index) {
  int b = find_block(t, world_to_block(t, wp));
  if (b < 0) return 0;
  int li = local_index(t, world_to_local_voxel(t, wp));
  if (b != from_block || li != from_index) {
#ifdef _OPENMP
#pragma omp atomic
#endif
    t->mismatches++;
  }
  combine(&t->vox[(size_t)b * nvox(t) + li], v, (unsigned char)t->o.max_sdf_weight);
  return 1;
}

/* IntegrateScanKernel, tsdf_volume.cu:436-513, over allocated blocks only */
static void integrate_scan(tsdf_t *t, const unsigned char *bgr, const float *depth, const float *T, const float *Ti) {
  const tsdf_options *o = &t->o;
  int bs = o->block_size;
  float vs = o->voxel_size, trunc = o->truncation_distance;
  unsigned long long upd = 0;
  int nb = t->nblk;
  /* Built with -fopenmp (libtsdf_oracle_omp.so, bench.py's multi-core CPU baseline only): blocks are independent as long as
   * every voxel's world -> camera -> world round trip lands on itself (mismatch counter 0, which the baseline asserts). */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : upd)
#endif
  for (int e = 0; e < nb; ++e) {
    i3 P = t->coord[e];
    int ix, iy;
    f3 position = {P.x * vs * bs, P.y * vs * bs, P.z * vs * bs};
    f3 pc = xform(Ti, position);
    if (pc.z < 0) continue;
    f3 center = {(float)(pc.x + 0.5 * vs * bs), (float)(pc.y + 0.5 * vs * bs), (float)(pc.z + 0.5 * vs * bs)};
    project(o, center, &ix, &iy);
    if (!(ix >= 0 && iy >= 0 && ix < o->width && iy < o->height)) continue;
    for (int bx = 0; bx < bs; bx++) for (int by = 0; by < bs; by++) for (int bz = 0; bz < bs; bz++) {
      f3 vp = {position.x + bx * vs, position.y + by * vs, position.z + bz * vs};
      vp = xform(Ti, vp);
      project(o, vp, &ix, &iy);
      if (!(ix >= 0 && iy >= 0 && ix < o->width && iy < o->height)) continue;
      int idx = iy * o->width + ix;
      float d = depth[idx];
      if (d <= 0) continue;
      if (d < o->min_sensor_depth) continue;
      if (d > o->max_sensor_depth) continue;
      f3 p3 = point3d(o, idx, d);
      float sd = norm3(p3);
      float vd = norm3(vp);
      voxel_t v;
      v.c[0] = bgr[3 * idx]; v.c[1] = bgr[3 * idx + 1]; v.c[2] = bgr[3 * idx + 2];
      v.weight = 1;
      int li = bx * bs * bs + by * bs + bz;
      if (vd > sd - trunc && vd < sd + trunc && d < o->max_sensor_depth) {
        v.sdf = sd - vd;
        upd += update_voxel(t, xform(T, vp), &v, e, li);
      } else if (vd < sd - trunc) {
        v.sdf = trunc;
        upd += update_voxel(t, xform(T, vp), &v, e, li);
      }
    }
  }
  t->updated_last = upd;
  t->updated_total += upd;
}

/* DrFusion::IntegrateScanAsync -> TsdfVolume::IntegrateScanAsync, tsdf_volume.cu:515-598 */
int tsdf_integrate(tsdf_t *t, const unsigned char *bgr, const float *depth, const float *pose16) {
  float inv[16];
  inverse4(pose16, inv);
  allocate_from_depth(t, depth, pose16);
  integrate_scan(t, bgr, depth, pose16, inv);
  return t->overflow;
}

/* GenerateRgbDepthKernel, tsdf_volume.cu:600-632 */
void tsdf_render(const tsdf_t *t, const float *pose16, unsigned char *bgr_out, float *depth_out) {
  const tsdf_options *o = &t->o;
  int size = o->height * o->width;
  for (int i = 0; i < size; ++i) {
    float cur = 0;
    while (cur < o->max_sensor_depth) {
      f3 p = xform(pose16, point3d(o, i, cur));
      voxel_t v = get_interpolated_voxel(t, p);
      if (v.weight == 0) cur += o->truncation_distance; else cur += v.sdf;
      if (v.weight != 0 && v.sdf < o->voxel_size) break;
    }
    if (cur < o->max_sensor_depth) {
      f3 p = xform(pose16, point3d(o, i, cur));
      voxel_t v = get_interpolated_voxel(t, p);
      bgr_out[3 * i] = v.c[0]; bgr_out[3 * i + 1] = v.c[1]; bgr_out[3 * i + 2] = v.c[2];
      depth_out[i] = cur;
    } else {
      bgr_out[3 * i] = bgr_out[3 * i + 1] = bgr_out[3 * i + 2] = 0;
      depth_out[i] = 0.0f;
    }
  }
}

int tsdf_num_blocks(const tsdf_t *t) { return t->nblk; }
unsigned long long tsdf_dda_capped(const tsdf_t *t) { return t->dda_capped; }
void tsdf_stats(const tsdf_t *t, unsigned long long out[4]) {
  out[0] = (unsigned long long)t->nblk; out[1] = t->updated_last; out[2] = t->updated_total; out[3] = t->mismatches;
}
/* canonical dump in allocation order: coords[3*i..], voxels[8*nvox*i..] */
int tsdf_export_blocks(const tsdf_t *t, int max_blocks, int *coords, unsigned char *voxels) {
  int n = t->nblk < max_blocks ? t->nblk : max_blocks;
  for (int i = 0; i < n; ++i) { coords[3 * i] = t->coord[i].x; coords[3 * i + 1] = t->coord[i].y; coords[3 * i + 2] = t->coord[i].z; }
  memcpy(voxels, t->vox, (size_t)n * nvox(t) * sizeof(voxel_t));
  return n;
}

/* ---------------------------------------------------------------- marching cubes
 * TrilinearInterpolation, mesh_extractor.cu:24-103: distance only (the interpolated colour is computed by the
 * reference but never used by ExtractMeshAtPosition); false as soon as one of the 8 voxels is unobserved. */
static int mc_trilinear(const tsdf_t *t, f3 position, float *distance) {
  float vs = t->o.voxel_size;
  f3 half = {vs / 2.0f, vs / 2.0f, vs / 2.0f};
  f3 pd = sub3(position, half);
  f3 vp = {position.x / vs, position.y / vs, position.z / vs};
  f3 w = {vp.x - floorf(vp.x), vp.y - floorf(vp.y), vp.z - floorf(vp.z)};
  static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}, {1, 1, 1}};
  float d = 0.0f;
  for (int k = 0; k < 8; ++k) {
    f3 off = {corner[k][0] ? vs : 0.0f, corner[k][1] ? vs : 0.0f, corner[k][2] ? vs : 0.0f};
    voxel_t v = get_voxel(t, add3(pd, off));
    if (v.weight == 0) return 0;
    float a = corner[k][0] ? w.x : (1.0f - w.x);
    float b = corner[k][1] ? w.y : (1.0f - w.y);
    float c = corner[k][2] ? w.z : (1.0f - w.z);
    d += a * b * c * v.sdf;
  }
  *distance = d;
  return 1;
}

/* VertexInterpolation, mesh_extractor.cu:105-134, with c1 == c2 == colour of the cell's centre voxel (as the caller
 * passes it) and isolevel 0: out[0..2] = position, out[3..5] = colour / 255 in the voxel's (BGR) channel order. */
static void mc_vertex(f3 p1, f3 p2, float d1, float d2, const unsigned char *c, float *out) {
  const float iso = 0.0f;
  f3 p;
  if (fabsf(iso - d1) < 0.00001f) p = p1;
  else if (fabsf(iso - d2) < 0.00001f) p = p2;
  else if (fabsf(d1 - d2) < 0.00001f) p = p1;
  else {
    float mu = (iso - d1) / (d2 - d1);
    p.x = p1.x + mu * (p2.x - p1.x);
    p.y = p1.y + mu * (p2.y - p1.y);
    p.z = p1.z + mu * (p2.z - p1.z);
  }
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
  out[3] = (float)c[0] / 255.f; out[4] = (float)c[1] / 255.f; out[5] = (float)c[2] / 255.f;
}

/* ExtractMeshKernel + ExtractMeshAtPosition (mesh_extractor.cu:136-265) over the dense lattice, then the
 * GetMeshSync layout (tsdf_volume.cu:800-832): vert[9*t + 3*k + 0..2] = position of vertex k of triangle t,
 * cols[9*t + 3*k + 0..2] = (colour.z, colour.y, colour.x) -- i.e. RGB from the BGR voxel.  Returns the number of
 * triangles found; at most max_tri are written. */
long tsdf_extract_mesh(const tsdf_t *t, const float *lower, const float *upper, long max_tri, float *vert, float *cols) {
  float vs = t->o.voxel_size;
  f3 size = {fabsf(lower[0] - upper[0]), fabsf(lower[1] - upper[1]), fabsf(lower[2] - upper[2])};
  int nx = f2i(size.x / vs), ny = f2i(size.y / vs), nz = f2i(size.z / vs);
  const float P = vs / 2.0f, M = -P;
  /* cube corners in Bourke order v0..v7 = p010 p110 p100 p000 p011 p111 p101 p001 (mesh_extractor.cu:192-199) */
  static const int cs[8][3] = {{0, 1, 0}, {1, 1, 0}, {1, 0, 0}, {0, 0, 0}, {0, 1, 1}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}};
  long ntri = 0;
  for (int gz = 0; gz < nz; ++gz) for (int gy = 0; gy < ny; ++gy) for (int gx = 0; gx < nx; ++gx) {
    f3 pos = {(float)gx * vs + lower[0], (float)gy * vs + lower[1], (float)gz * vs + lower[2]};
    f3 p[8];
    float d[8];
    int ok = 1;
    for (int k = 0; k < 8 && ok; ++k) {
      f3 off = {cs[k][0] ? P : M, cs[k][1] ? P : M, cs[k][2] ? P : M};
      p[k] = add3(pos, off);
      ok = mc_trilinear(t, p[k], &d[k]);
    }
    if (!ok) continue;
    unsigned cube = 0;
    for (int k = 0; k < 8; ++k) if (d[k] < 0.0f) cube |= 1u << k;
    if (cube == 0 || cube == 255) continue; /* edgeTable[cube] == 0 */
    voxel_t v = get_voxel(t, pos);
    unsigned long long row = kMcTri[cube];
    for (int i = 0; i < 15 && ((row >> (4 * i)) & 15) != 15; i += 3) {
      if (ntri < max_tri) {
        for (int k = 0; k < 3; ++k) {
          int e = (int)((row >> (4 * (i + k))) & 15);
          int a = kMcEdgeCorner[e][0], b = kMcEdgeCorner[e][1];
          float o6[6];
          mc_vertex(p[a], p[b], d[a], d[b], v.c, o6);
          float *vv = vert + 9 * ntri + 3 * k, *cc = cols + 9 * ntri + 3 * k;
          vv[0] = o6[0]; vv[1] = o6[1]; vv[2] = o6[2];
          cc[0] = o6[5]; cc[1] = o6[4]; cc[2] = o6[3];
        }
      }
      ++ntri;
    }
  }
  return ntri;
}

/* ---- pins: the helpers above, one by one, for tests/test_ref_fusion.py, which compares each with the REFERENCE's own
 * header code compiled for the host (oracle/_ref/libdr_fusion_ref.so, oracle/ref_fusion_capi.cpp) ---- */
void tsdf_pin_combine(float sdf, const unsigned char *c, unsigned char w, float vsdf, const unsigned char *vc, unsigned char vw,
                      unsigned char max_weight, float *sdf_out, unsigned char *c_out, unsigned char *w_out) {
  voxel_t a = {sdf, {c[0], c[1], c[2]}, w}, b = {vsdf, {vc[0], vc[1], vc[2]}, vw};
  combine(&a, &b, max_weight);
  *sdf_out = a.sdf; c_out[0] = a.c[0]; c_out[1] = a.c[1]; c_out[2] = a.c[2]; *w_out = a.weight;
}
void tsdf_pin_combine_colour_table(unsigned char w, unsigned char *out) {
  for (int c = 0; c < 256; ++c)
    for (int vc = 0; vc < 256; ++vc) {
      voxel_t a = {0.0f, {(unsigned char)c, (unsigned char)c, (unsigned char)c}, w};
      voxel_t b = {0.0f, {(unsigned char)vc, (unsigned char)vc, (unsigned char)vc}, 1};
      combine(&a, &b, 255);
      out[c * 256 + vc] = a.c[0];
    }
}
static tsdf_options pin_opts(const float *k4, int rows, int cols) {
  tsdf_options o; memset(&o, 0, sizeof o);
  o.fx = k4[0]; o.fy = k4[1]; o.cx = k4[2]; o.cy = k4[3]; o.height = rows; o.width = cols;
  return o;
}
void tsdf_pin_point3d(const float *k4, int rows, int cols, int i, float depth, float *out3) {
  tsdf_options o = pin_opts(k4, rows, cols);
  f3 p = point3d(&o, i, depth);
  out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}
void tsdf_pin_project(const float *k4, int rows, int cols, const float *p3, int *out2) {
  tsdf_options o = pin_opts(k4, rows, cols);
  f3 p = {p3[0], p3[1], p3[2]};
  project(&o, p, &out2[0], &out2[1]);
}
float tsdf_pin_norm(const float *p3) { f3 p = {p3[0], p3[1], p3[2]}; return norm3(p); }
void tsdf_pin_xform(const float *m16, const float *p3, float *out3) {
  f3 p = {p3[0], p3[1], p3[2]};
  f3 r = xform(m16, p);
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void tsdf_pin_world_maps(const tsdf_t *t, const float *p3, int *global_voxel3, int *block3, int *local3, float *back3) {
  f3 p = {p3[0], p3[1], p3[2]};
  i3 g = world_to_global_voxel(t, p), b = world_to_block(t, p), l = world_to_local_voxel(t, p);
  global_voxel3[0] = g.x; global_voxel3[1] = g.y; global_voxel3[2] = g.z;
  block3[0] = b.x; block3[1] = b.y; block3[2] = b.z;
  local3[0] = l.x; local3[1] = l.y; local3[2] = l.z;
  /* GlobalVoxelToWorld, tsdf_volume.cu:103-107 */
  back3[0] = g.x * t->o.voxel_size; back3[1] = g.y * t->o.voxel_size; back3[2] = g.z * t->o.voxel_size;
}
