/* tracker_oracle.c -- CPU ORACLE for the dense coarse tracker.  TEST INFRASTRUCTURE, NOT
 * PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Plain-C restatement of
 *   calcResKernelNew / getInterpolatedElement33   libdr/cuda_coarse_tracker/src/cuda_coarse_tracker_private.cu:21-214
 *   calcGKernel                                    .../cuda_coarse_tracker_private.cu:260-372
 *   CudaCoarseTracker::{setK,calcRes,calcG} host   .../cuda_coarse_tracker.cpp:217-356,358-372
 *   AffLight::fromToVecExposure                    .../cuda_coarse_tracker.cpp:40-49
 *   dense-depth reprojection into the tracker's reference frame   src/FullSystem/CoarseTracker.cpp:655-725
 *
 * PARITY UNPINNED BY THE REFERENCE: the library's own driver (src/main.cu) reads fixtures (cct_data/*.npy) that are
 * not shipped, and there are no tests.  This restatement defines the canonical result:
 *   (1) per-point quantities (warped_u/v/dx/dy/idepth/residual/weight, reprojected depth, appended points) are fp32 in
 *       the reference's expression order, no FMA contraction -> the HIP path must match them BIT FOR BIT;
 *   (2) the reductions (7 calcRes sums, 45 calcG sums) are accumulated here in double over points in index order;
 *       the reference accumulates in float through cub::BlockReduce + atomicAdd (order arbitrary, ~1e-6 relative
 *       noise), so sums are compared with a relative tolerance stated in the tests;
 *   (3) Ki = K^-1 by cofactors * (1/det) in double (Eigen's 3x3 inverse), then cast to float as the host code does;
 *   (4) the 3-vector products of the reprojection follow Eigen's unrolled reduction a0 + (a1 + a2) (redux_novec_unroller splits [0,3) into
 *       [0,1) + [1,3)); -DTRK_SUM_LEFT builds (a0 + a1) + a2 instead (libtracker_oracle_left.so).  PINNED since round 5: the reference's own
 *       lines (CoarseTracker.cpp:654-723) compiled for the host (oracle/_ref/libdense_handoff_ref.so, both orders) give the same appended
 *       points bit for bit, and tests/test_ref_handoff.py caps what the order can move (a pixel index at a rounding tie, 1 ulp of idepth);
 *   (5) reprojected candidates with depth <= 0 in the target frame are dropped (in the reference they enter the
 *       z-buffer test `proj < 0 ? set : min`, making the result depend on visiting order; such pixels are discarded
 *       by the `mvs_depth <= 0` test afterwards unless overwritten);
 *   (6) appended dense points are written contiguously after the n0 sparse points (the reference pre-increments the
 *       count, CoarseTracker.cpp:715-721, leaving slot n0 stale and dropping the last point -- a defect not inherited).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int w, h;
  float fx, fy, cx, cy;
  double Ki[9];
  float huber, coarse_cutoff;
  int n, n_max;
  float *pc_u, *pc_v, *pc_idepth, *pc_color;
  float *dInew; /* 3*w*h: (I, dx, dy) interleaved */
  float ref_exposure;
  double ref_aff[2];
  float *warped[7]; /* u v dx dy idepth residual weight */
  int num_terms_in_warped;
} trk_t;

trk_t *trk_create(int w, int h, float huber, float coarse_cutoff, int n_max) {
  trk_t *t = (trk_t *)calloc(1, sizeof(trk_t));
  t->w = w; t->h = h; t->huber = huber; t->coarse_cutoff = coarse_cutoff;
  t->n_max = n_max > 0 ? n_max : w * h;
  t->pc_u = (float *)calloc(t->n_max, 4); t->pc_v = (float *)calloc(t->n_max, 4);
  t->pc_idepth = (float *)calloc(t->n_max, 4); t->pc_color = (float *)calloc(t->n_max, 4);
  t->dInew = (float *)calloc((size_t)3 * w * h, 4);
  for (int k = 0; k < 7; ++k) t->warped[k] = (float *)calloc(t->n_max, 4);
  return t;
}
void trk_destroy(trk_t *t) {
  if (!t) return;
  free(t->pc_u); free(t->pc_v); free(t->pc_idepth); free(t->pc_color); free(t->dInew);
  for (int k = 0; k < 7; ++k) free(t->warped[k]);
  free(t);
}

/* cuda_coarse_tracker.cpp:358-372 */
void trk_set_k(trk_t *t, float fx, float fy, float cx, float cy) {
  t->fx = fx; t->fy = fy; t->cx = cx; t->cy = cy;
  double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
  double c[9];
  c[0] = K[4] * K[8] - K[5] * K[7]; c[1] = K[2] * K[7] - K[1] * K[8]; c[2] = K[1] * K[5] - K[2] * K[4];
  c[3] = K[5] * K[6] - K[3] * K[8]; c[4] = K[0] * K[8] - K[2] * K[6]; c[5] = K[2] * K[3] - K[0] * K[5];
  c[6] = K[3] * K[7] - K[4] * K[6]; c[7] = K[1] * K[6] - K[0] * K[7]; c[8] = K[0] * K[4] - K[1] * K[3];
  double det = K[0] * c[0] + K[1] * c[3] + K[2] * c[6];
  double inv = 1.0 / det;
  for (int i = 0; i < 9; ++i) t->Ki[i] = c[i] * inv;
}

int trk_set_reference(trk_t *t, int n, const float *u, const float *v, const float *id, const float *col, float ref_exposure,
                      const double *ref_aff) {
  if (n > t->n_max) return 1;
  t->n = n;
  memcpy(t->pc_u, u, 4 * (size_t)n); memcpy(t->pc_v, v, 4 * (size_t)n);
  memcpy(t->pc_idepth, id, 4 * (size_t)n); memcpy(t->pc_color, col, 4 * (size_t)n);
  t->ref_exposure = ref_exposure; t->ref_aff[0] = ref_aff[0]; t->ref_aff[1] = ref_aff[1];
  return 0;
}
void trk_set_new(trk_t *t, const float *dInew) { memcpy(t->dInew, dInew, (size_t)12 * t->w * t->h); }

/* AffLight::fromToVecExposure, cuda_coarse_tracker.cpp:40-49 */
static void aff_ll(float exposureF, float exposureT, const double *g2F, const double *g2T, float *ax, float *ay) {
  if (exposureF == 0 || exposureT == 0) exposureT = exposureF = 1;
  double a = exp(g2T[0] - g2F[0]) * exposureT / exposureF;
  double b = g2T[1] - a * g2F[1];
  *ax = (float)a; *ay = (float)b;
}

/* getInterpolatedElement33, cuda_coarse_tracker_private.cu:21-37 */
static void interp33(const float *mat, float x, float y, int width, float out[3]) {
  int ix = (int)x, iy = (int)y;
  float dx = x - ix, dy = y - iy;
  float dxdy = dx * dy;
  const float *bp = mat + 3 * (ix + iy * width);
  for (int c = 0; c < 3; ++c)
    out[c] = dxdy * bp[3 * (1 + width) + c] + (dy - dxdy) * bp[3 * width + c] + (dx - dxdy) * bp[3 + c] + (1.0f - dx - dy + dxdy) * bp[c];
}

static void matvec3(const float *A, const float *x, float *o) { /* numeric_cuda Matmul: out += A(r,k)*x(k), k ascending, out = 0 */
  for (int r = 0; r < 3; ++r) {
    float s = 0.0f;
    for (int k = 0; k < 3; ++k) s += A[3 * r + k] * x[k];
    o[r] = s;
  }
}

/* CudaCoarseTracker::calcRes, cuda_coarse_tracker.cpp:217-288 + calcResKernelNew.  refToNew: 4x4 row-major double.
 * out6 as the reference's Vec6; sums7 (may be NULL): the 7 raw sums in double. */
void trk_calc_res(trk_t *t, const double *refToNew, float new_exposure, const double *aff_g2l, float cutoffTH, double *out6, double *sums7) {
  float r2n[16], Kif[9];
  for (int i = 0; i < 16; ++i) r2n[i] = (float)refToNew[i];
  for (int i = 0; i < 9; ++i) Kif[i] = (float)t->Ki[i];
  float ax, ay;
  aff_ll(t->ref_exposure, new_exposure, t->ref_aff, aff_g2l, &ax, &ay);
  const float huber = t->huber;
  const float maxEnergy = 2 * huber * cutoffTH - huber * huber;
  const int w = t->w, h = t->h;
  const float fx = t->fx, fy = t->fy, cx = t->cx, cy = t->cy;
  double S[7] = {0, 0, 0, 0, 0, 0, 0}; /* E, numTermsInE, numTermsInWarped, numSaturated, shiftT, shiftRT, shiftNum */
  float R[9], tt[3], RKi[9];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = r2n[4 * r + c]; tt[r] = r2n[4 * r + 3]; }
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    float s = 0.0f;
    for (int k = 0; k < 3; ++k) s += R[3 * r + k] * Kif[3 * k + c];
    RKi[3 * r + c] = s;
  }
  for (int i = 0; i < t->n; ++i) {
    for (int k = 0; k < 7; ++k) t->warped[k][i] = 0;
    float id = t->pc_idepth[i], x = t->pc_u[i], y = t->pc_v[i];
    float xy1[3] = {x, y, 1.0f}, pt[3];
    matvec3(RKi, xy1, pt);
    for (int r = 0; r < 3; ++r) pt[r] += tt[r] * id;
    float u = pt[0] / pt[2], v = pt[1] / pt[2];
    float Ku = fx * u + cx, Kv = fy * v + cy;
    float new_idepth = id / pt[2];
    if (i % 32 == 0) {
      float a[3], b[3], c3[3];
      matvec3(Kif, xy1, a);
      for (int r = 0; r < 3; ++r) a[r] += tt[r] * id;
      float KuT = fx * (a[0] / a[2]) + cx, KvT = fy * (a[1] / a[2]) + cy;
      matvec3(Kif, xy1, b);
      for (int r = 0; r < 3; ++r) b[r] -= tt[r] * id;
      float KuT2 = fx * (b[0] / b[2]) + cx, KvT2 = fy * (b[1] / b[2]) + cy;
      matvec3(RKi, xy1, c3);
      for (int r = 0; r < 3; ++r) c3[r] -= tt[r] * id;
      float Ku3 = fx * (c3[0] / c3[2]) + cx, Kv3 = fy * (c3[1] / c3[2]) + cy;
      float sT = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      float sRT = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      S[4] += sT; S[5] += sRT; S[6] += 2.0;
    }
    if (Ku > 2 && Kv > 2 && Ku < w - 3 && Kv < h - 3 && new_idepth > 0) {
      float refColor = t->pc_color[i], hit[3];
      interp33(t->dInew, Ku, Kv, w, hit);
      if (isfinite(hit[0])) {
        float residual = hit[0] - (ax * refColor + ay);
        float hw = fabsf(residual) < huber ? 1 : huber / fabsf(residual);
        if (fabsf(residual) > cutoffTH) {
          S[0] += maxEnergy; S[1] += 1; S[3] += 1;
        } else {
          S[0] += hw * residual * residual * (2 - hw); S[1] += 1; S[2] += 1;
          t->warped[4][i] = new_idepth; t->warped[0][i] = u; t->warped[1][i] = v;
          t->warped[2][i] = hit[1]; t->warped[3][i] = hit[2]; t->warped[5][i] = residual; t->warped[6][i] = hw;
        }
      }
    }
  }
  if (sums7) memcpy(sums7, S, sizeof S);
  out6[0] = S[0]; out6[1] = S[1]; out6[2] = S[4] / S[6]; out6[3] = 0; out6[4] = S[5] / S[6]; out6[5] = S[3] / S[1];
  t->num_terms_in_warped = (int)S[2];
}

/* CudaCoarseTracker::calcG, cuda_coarse_tracker.cpp:290-356 + calcGKernel.  H: 8x8 row-major, b: 8; raw45 (may be
 * NULL): the 45 unscaled upper-triangular sums of J w J^T (J in R^9) in double. */
void trk_calc_g(trk_t *t, double *H, double *b, float new_exposure, const double *aff_g2l, double *raw45) {
  float ax, ay;
  aff_ll(t->ref_exposure, new_exposure, t->ref_aff, aff_g2l, &ax, &ay);
  const float a = ax, b0 = (float)t->ref_aff[1];
  double acc[45];
  for (int k = 0; k < 45; ++k) acc[k] = 0;
  for (int i = 0; i < t->n; ++i) {
    float J[9];
    const float dx = t->warped[2][i] * t->fx, dy = t->warped[3][i] * t->fy;
    const float u = t->warped[0][i], v = t->warped[1][i], id = t->warped[4][i];
    J[0] = id * dx; J[1] = id * dy; J[2] = -id * (u * dx + v * dy);
    J[3] = -(u * v * dx + dy + dy * v * v); J[4] = u * v * dy + dx + dx * u * u; J[5] = u * dy - v * dx;
    J[6] = a * (b0 - t->pc_color[i]); J[7] = -1; J[8] = t->warped[5][i];
    const float w = t->warped[6][i];
    int k = 0;
    for (int j1 = 0; j1 < 9; ++j1) {
      const float Jw = J[j1] * w;
      for (int j2 = j1; j2 < 9; ++j2) acc[k++] += (double)(Jw * J[j2]);
    }
  }
  if (raw45) memcpy(raw45, acc, sizeof acc);
  const double factor = 1.0 / t->num_terms_in_warped;
  for (int r = 0; r < 8; ++r) {
    for (int c = 0; c < 8; ++c) {
      int lo = r < c ? r : c, hi = r < c ? c : r;
      H[8 * r + c] = acc[lo * 9 + hi - lo * (lo + 1) / 2] * factor;
    }
    b[r] = acc[r * 9 + 8 - r * (r + 1) / 2] * factor;
  }
  /* SCALE_XI_ROT 1, SCALE_XI_TRANS 0.5, SCALE_A 10, SCALE_B 1000 (cuda_coarse_tracker.cpp:11-18, :343-354) */
  const double s[8] = {1, 1, 1, 0.5, 0.5, 0.5, 10, 1000};
  for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) H[8 * r + c] *= s[r] * s[c]; b[r] *= s[r]; }
}

int trk_n(const trk_t *t) { return t->n; }
void trk_get_points(const trk_t *t, float *u, float *v, float *id, float *col) {
  memcpy(u, t->pc_u, 4 * (size_t)t->n); memcpy(v, t->pc_v, 4 * (size_t)t->n);
  memcpy(id, t->pc_idepth, 4 * (size_t)t->n); memcpy(col, t->pc_color, 4 * (size_t)t->n);
}
void trk_get_warped(const trk_t *t, int k, float *out) { memcpy(out, t->warped[k], 4 * (size_t)t->n); }

/* CoarseTracker::setCoarseTrackingRef dense-depth branch, CoarseTracker.cpp:655-725: forward-warp the depth map into
 * the tracker's reference frame (z-buffer min), then append every pixel with a projected depth (and no sparse idepth,
 * unless dense_only) to the point list.  KRKi (3x3 row-major) and Kt are the caller's float products (:672-673).
 * proj_out (may be NULL): the w*h projected depth map (-1 = empty).  Returns the new point count or -1 on overflow. */
int trk_append_dense(trk_t *t, const float *depth, const float *KRKi, const float *Kt, int step, int dense_only, const float *idepth0,
                     const float *dIp0, float *proj_out) {
  const int w = t->w, h = t->h;
  float *proj = (float *)malloc(sizeof(float) * (size_t)w * h);
  for (int i = 0; i < w * h; ++i) proj[i] = -1.f;
  for (int y = 0; y < h; y += step)
    for (int x = 0; x < w; x += step) {
      const size_t i = (size_t)x + (size_t)y * w;
      const float d = depth[i];
      if (d <= 0.f) continue;
      const float o[3] = {x * d, y * d, d};
      float p[3];
#ifdef TRK_SUM_LEFT
      for (int r = 0; r < 3; ++r) p[r] = ((KRKi[3 * r] * o[0] + KRKi[3 * r + 1] * o[1]) + KRKi[3 * r + 2] * o[2]) + Kt[r];
#else
      for (int r = 0; r < 3; ++r) p[r] = (KRKi[3 * r] * o[0] + (KRKi[3 * r + 1] * o[1] + KRKi[3 * r + 2] * o[2])) + Kt[r];
#endif
      const float pd = p[2];
      if (!(pd > 0.f)) continue;
      const int pu = (int)(p[0] / p[2] + 0.5f), pv = (int)(p[1] / p[2] + 0.5f);
      if (pu > w - 4 || pv > h - 4 || pu < 3 || pv < 3) continue;
      float *q = proj + pu + (size_t)pv * w;
      if (*q < 0) *q = pd; else *q = pd < *q ? pd : *q;
    }
  int n = t->n;
  for (int y = 2; y < h - 2; ++y)
    for (int x = 2; x < w - 2; ++x) {
      const int i = x + y * w;
      const float m = proj[i];
      if (m <= 0) continue;
      if (dense_only || idepth0[i] <= 0) {
        if (n >= t->n_max) { free(proj); return -1; }
        t->pc_u[n] = (float)x; t->pc_v[n] = (float)y; t->pc_idepth[n] = 1.f / m; t->pc_color[n] = dIp0[3 * i];
        ++n;
      }
    }
  if (proj_out) memcpy(proj_out, proj, sizeof(float) * (size_t)w * h);
  free(proj);
  t->n = n;
  return n;
}

/* ---- pin: the exact float inputs this restatement's calcRes / calcG "kernels" work from, so that tests/test_ref_tracker.py
 * can hand the SAME inputs to the reference's own kernels compiled for the host (oracle/_ref/libcoarse_tracker_ref.so) ---- */
void trk_kernel_inputs(const trk_t *t, const double *refToNew, float new_exposure, const double *aff_g2l, float cutoffTH,
                       float *r2n16, float *Ki9, float *aff2, float *maxEnergy, float *ref_aff_b) {
  for (int i = 0; i < 16; ++i) r2n16[i] = (float)refToNew[i];
  for (int i = 0; i < 9; ++i) Ki9[i] = (float)t->Ki[i];
  aff_ll(t->ref_exposure, new_exposure, t->ref_aff, aff_g2l, &aff2[0], &aff2[1]);
  *maxEnergy = 2 * t->huber * cutoffTH - t->huber * t->huber;
  *ref_aff_b = (float)t->ref_aff[1];
}
