// dr_tracker.hip -- MI355X engine behind the dense coarse tracker operator (C ABI: drt_* in include/dr_mi355x.h).
//
// Replaces tandem/libdr/cuda_coarse_tracker (CUDA + cub + Eigen) and the dense-depth hand-off loop of
// CoarseTracker::setCoarseTrackingRef (src/FullSystem/CoarseTracker.cpp:655-725) -- SURVEY 8(f) rows 3 and 4:
//   calcResKernelNew  cuda_coarse_tracker_private.cu:39-214  -> k_trk_res   (one lane per reference point)
//   calcGKernel       cuda_coarse_tracker_private.cu:260-372 -> k_trk_g     (grid-stride, 45 sums per lane)
//   cub::BlockReduce + atomicAdd(float)                       -> wave shuffles + per-workgroup partials in DOUBLE,
//                                                               folded in a fixed order by k_trk_fold: the sums are
//                                                               deterministic and closer to exact than the reference's
//   host loop over 307 k pixels + D2H of the rendered depth   -> k_trk_project (atomicMin z-buffer on the float bit
//                                                               pattern) + row count / scan / write (row-major,
//                                                               deterministic append to the device point list)
// Per-point quantities follow the oracle (oracle/tracker_oracle.c) bit for bit: fp32, the reference's expression
// order, -ffp-contract=off.  These kernels move a few MB per call: they are launch-latency bound, not roofline bound.
#include <cmath>
#include <memory>

#include "dr_common.h"

namespace dr {

constexpr int kTrkThreads = 256;
constexpr int kTrkMaxBlocks = 512;

struct TrkDev {
  int w, h, n;
  float fx, fy, cx, cy;
  const float *pc_u, *pc_v, *pc_idepth, *pc_color;
  const float *dInew;
  float *warped[7];  // u v dx dy idepth residual weight
};

struct TrkResArgs {
  float R[9], t[3], Ki[9], RKi[9];
  float ax, ay, huber, maxEnergy, cutoffTH;
};

__device__ inline void matvec3(const float *A, float x, float y, float z, float *o) {  // numeric_cuda Matmul order
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float s = 0.0f;
    s += A[3 * r] * x; s += A[3 * r + 1] * y; s += A[3 * r + 2] * z;
    o[r] = s;
  }
}

// Block sum of NS doubles per lane -> partial[blockIdx.x * NS + k]; deterministic (fixed shuffle tree, fixed wave order).
template <int NS>
__device__ inline void block_partials(const double *priv, double *partial) {
  __shared__ double ws[kTrkThreads / 64][NS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double v = priv[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) ws[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double s = 0.0;
    for (int w = 0; w < kTrkThreads / 64; ++w) s += ws[w][threadIdx.x];
    partial[(size_t)blockIdx.x * NS + threadIdx.x] = s;
  }
}

// calcResKernelNew, cuda_coarse_tracker_private.cu:39-214
__global__ __launch_bounds__(kTrkThreads) void k_trk_res(const TrkDev d, const TrkResArgs a, double *partial) {
  double priv[7] = {0, 0, 0, 0, 0, 0, 0};  // E, numTermsInE, numTermsInWarped, numSaturated, shiftT, shiftRT, shiftNum
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.n) {
    float wu = 0, wv = 0, wdx = 0, wdy = 0, wid = 0, wres = 0, ww = 0;
    const float id = d.pc_idepth[i], x = d.pc_u[i], y = d.pc_v[i];
    float pt[3];
    matvec3(a.RKi, x, y, 1.0f, pt);
#pragma unroll
    for (int r = 0; r < 3; ++r) pt[r] += a.t[r] * id;
    const float u = pt[0] / pt[2], v = pt[1] / pt[2];
    const float Ku = d.fx * u + d.cx, Kv = d.fy * v + d.cy;
    const float new_idepth = id / pt[2];
    if (i % 32 == 0) {
      float p1[3], p2[3], p3[3];
      matvec3(a.Ki, x, y, 1.0f, p1);
#pragma unroll
      for (int r = 0; r < 3; ++r) p1[r] += a.t[r] * id;
      const float KuT = d.fx * (p1[0] / p1[2]) + d.cx, KvT = d.fy * (p1[1] / p1[2]) + d.cy;
      matvec3(a.Ki, x, y, 1.0f, p2);
#pragma unroll
      for (int r = 0; r < 3; ++r) p2[r] -= a.t[r] * id;
      const float KuT2 = d.fx * (p2[0] / p2[2]) + d.cx, KvT2 = d.fy * (p2[1] / p2[2]) + d.cy;
      matvec3(a.RKi, x, y, 1.0f, p3);
#pragma unroll
      for (int r = 0; r < 3; ++r) p3[r] -= a.t[r] * id;
      const float Ku3 = d.fx * (p3[0] / p3[2]) + d.cx, Kv3 = d.fy * (p3[1] / p3[2]) + d.cy;
      float sT = (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
      sT += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
      float sRT = (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
      sRT += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
      priv[4] = sT; priv[5] = sRT; priv[6] = 2.0;
    }
    if (Ku > 2 && Kv > 2 && Ku < d.w - 3 && Kv < d.h - 3 && new_idepth > 0) {
      const float refColor = d.pc_color[i];
      // getInterpolatedElement33, :21-37
      const int ix = (int)Ku, iy = (int)Kv;
      const float dx = Ku - ix, dy = Kv - iy, dxdy = dx * dy;
      const float *bp = d.dInew + 3 * (ix + iy * d.w);
      float hit[3];
#pragma unroll
      for (int c = 0; c < 3; ++c)
        hit[c] = dxdy * bp[3 * (1 + d.w) + c] + (dy - dxdy) * bp[3 * d.w + c] + (dx - dxdy) * bp[3 + c] + (1.0f - dx - dy + dxdy) * bp[c];
      if (isfinite(hit[0])) {
        const float residual = hit[0] - (a.ax * refColor + a.ay);
        const float hw = fabsf(residual) < a.huber ? 1 : a.huber / fabsf(residual);
        if (fabsf(residual) > a.cutoffTH) {
          priv[0] = a.maxEnergy; priv[1] = 1; priv[3] = 1;
        } else {
          priv[0] = hw * residual * residual * (2 - hw); priv[1] = 1; priv[2] = 1;
          wid = new_idepth; wu = u; wv = v; wdx = hit[1]; wdy = hit[2]; wres = residual; ww = hw;
        }
      }
    }
    d.warped[0][i] = wu; d.warped[1][i] = wv; d.warped[2][i] = wdx; d.warped[3][i] = wdy;
    d.warped[4][i] = wid; d.warped[5][i] = wres; d.warped[6][i] = ww;
  }
  block_partials<7>(priv, partial);
}

// calcGKernel, cuda_coarse_tracker_private.cu:260-372: 45 upper-triangular sums of (J w) J^T, J in R^9
__global__ __launch_bounds__(kTrkThreads) void k_trk_g(const TrkDev d, float a, float b0, double *partial) {
  double priv[45];
#pragma unroll
  for (int k = 0; k < 45; ++k) priv[k] = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += gridDim.x * blockDim.x) {
    float J[9];
    const float dx = d.warped[2][i] * d.fx, dy = d.warped[3][i] * d.fy;
    const float u = d.warped[0][i], v = d.warped[1][i], id = d.warped[4][i];
    J[0] = id * dx; J[1] = id * dy; J[2] = -id * (u * dx + v * dy);
    J[3] = -(u * v * dx + dy + dy * v * v); J[4] = u * v * dy + dx + dx * u * u; J[5] = u * dy - v * dx;
    J[6] = a * (b0 - d.pc_color[i]); J[7] = -1; J[8] = d.warped[5][i];
    const float w = d.warped[6][i];
    int k = 0;
#pragma unroll
    for (int j1 = 0; j1 < 9; ++j1) {
      const float Jw = J[j1] * w;
#pragma unroll
      for (int j2 = j1; j2 < 9; ++j2) priv[k++] += (double)(Jw * J[j2]);
    }
  }
  block_partials<45>(priv, partial);
}

// out[k] = sum over workgroups of partial[b * NS + k], in a fixed order (one wave per column)
__global__ void k_trk_fold(const double *partial, int nblocks, int ns, double *out) {
  const int k = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * ns + k];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) out[k] = s;
}

// ---- dense-depth reprojection, CoarseTracker.cpp:655-725 ----
__global__ void k_trk_fill(unsigned *p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
struct TrkProjArgs { float KRKi[9], Kt[3]; int step; };
__global__ void k_trk_project(const float *__restrict__ depth, int w, int h, TrkProjArgs a, unsigned *zbuf) {
  const int nx = (w + a.step - 1) / a.step, ny = (h + a.step - 1) / a.step;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nx * ny) return;
  const int x = (j % nx) * a.step, y = (j / nx) * a.step;
  const float dz = depth[(size_t)x + (size_t)y * w];
  if (dz <= 0.f) return;
  const float o0 = x * dz, o1 = y * dz, o2 = dz;
  float p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) p[r] = (a.KRKi[3 * r] * o0 + (a.KRKi[3 * r + 1] * o1 + a.KRKi[3 * r + 2] * o2)) + a.Kt[r];  // Eigen: a0 + (a1 + a2)
  const float pd = p[2];
  if (!(pd > 0.f)) return;
  const int pu = (int)(p[0] / p[2] + 0.5f), pv = (int)(p[1] / p[2] + 0.5f);
  if (pu > w - 4 || pv > h - 4 || pu < 3 || pv < 3) return;
  atomicMin(&zbuf[pu + (size_t)pv * w], __float_as_uint(pd));  // positive floats order like their bit patterns
}
__device__ inline bool trk_take(const unsigned *zbuf, const float *idepth0, int dense_only, int i) {
  const unsigned z = zbuf[i];
  return z != 0x7f800000u && (dense_only || idepth0[i] <= 0);
}
__global__ __launch_bounds__(kTrkThreads) void k_trk_row_count(const unsigned *zbuf, const float *idepth0, int dense_only, int w, int h, int *rowcnt) {
  const int y = blockIdx.x + 2;
  int c = 0;
  for (int x = 2 + threadIdx.x; x < w - 2; x += kTrkThreads) c += trk_take(zbuf, idepth0, dense_only, x + y * w);
  __shared__ int ws[kTrkThreads / 64];
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { int s = 0; for (int k = 0; k < kTrkThreads / 64; ++k) s += ws[k]; rowcnt[blockIdx.x] = s; }
}
__global__ void k_trk_row_scan(int *rowcnt, int nrows, int n0, int n_max, int *result) {  // exclusive scan in place, single lane: <= a few thousand rows
  if (threadIdx.x || blockIdx.x) return;
  int s = n0;
  for (int r = 0; r < nrows; ++r) { const int c = rowcnt[r]; rowcnt[r] = s; s += c; }
  result[0] = s;
  result[1] = s > n_max;
}
__global__ __launch_bounds__(kTrkThreads) void k_trk_row_write(const unsigned *zbuf, const float *idepth0, const float *dIp0, int dense_only, int w, int h,
                                                              const int *rowoff, const int *result, float *pc_u, float *pc_v, float *pc_idepth, float *pc_color) {
  if (result[1]) return;  // overflow: nothing is written, the host reports it
  const int y = blockIdx.x + 2;
  __shared__ int ws[kTrkThreads / 64];
  __shared__ int base;
  if (threadIdx.x == 0) base = rowoff[blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int x0 = 2; x0 < w - 2; x0 += kTrkThreads) {
    const int x = x0 + threadIdx.x;
    const bool take = x < w - 2 && trk_take(zbuf, idepth0, dense_only, x + y * w);
    const unsigned long long m = __ballot(take);
    if (lane == 0) ws[wave] = __popcll(m);
    __syncthreads();
    int before = base;
    for (int k = 0; k < wave; ++k) before += ws[k];
    if (take) {
      const int dst = before + __popcll(m & ((1ull << lane) - 1ull));
      const int i = x + y * w;
      pc_u[dst] = (float)x; pc_v[dst] = (float)y; pc_idepth[dst] = 1.f / __uint_as_float(zbuf[i]); pc_color[dst] = dIp0[3 * (size_t)i];
    }
    __syncthreads();
    if (threadIdx.x == 0) base += ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ engine
class TrackerEngine {
 public:
  TrackerEngine(int w, int h, float huber, float coarse_cutoff, int device) : device_(device), w_(w), h_(h), huber_(huber), cutoff_(coarse_cutoff) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) fail(DR_ERR_DEVICE, "DrCoarseTracker: no HIP device %d (found %d) -- the MI355X path has no CPU fallback", device, n);
    if (w <= 0 || h <= 0) fail(DR_ERR_ARG, "DrCoarseTracker: invalid image size %dx%d", w, h);
    DR_HIP(hipSetDevice(device_));
    int lo, hi;
    DR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    DR_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));  // cuda_coarse_tracker.cpp:63-68: greatest priority
  }
  ~TrackerEngine() {
    (void)hipSetDevice(device_);
    (void)hipStreamSynchronize(stream_);
    release();
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    (void)hipStreamDestroy(stream_);
  }
  void set_k(int w, int h, float fx, float fy, float cx, float cy) {  // cuda_coarse_tracker.cpp:358-372
    if (w != w_ || h != h_) fail(DR_ERR_ARG, "CudaCoarseTracker::setK wrong h,w.");
    fx_ = fx; fy_ = fy; cx_ = cx; cy_ = cy;
    const double K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    double c[9];
    c[0] = K[4] * K[8] - K[5] * K[7]; c[1] = K[2] * K[7] - K[1] * K[8]; c[2] = K[1] * K[5] - K[2] * K[4];
    c[3] = K[5] * K[6] - K[3] * K[8]; c[4] = K[0] * K[8] - K[2] * K[6]; c[5] = K[2] * K[3] - K[0] * K[5];
    c[6] = K[3] * K[7] - K[4] * K[6]; c[7] = K[1] * K[6] - K[0] * K[7]; c[8] = K[0] * K[4] - K[1] * K[3];
    const double det = K[0] * c[0] + K[1] * c[3] + K[2] * c[6], inv = 1.0 / det;
    for (int i = 0; i < 9; ++i) Ki_[i] = c[i] * inv;
    have_k_ = true;
  }
  void init(int n_max) {  // cuda_coarse_tracker.cpp:101-140
    if (n_max_ != 0) fail(DR_ERR_PROTOCOL, "Cannot call CudaCoarseTracker::init more than once.");
    DR_HIP(hipSetDevice(device_));
    n_max_ = n_max > 0 ? n_max : w_ * h_;
    for (int k = 0; k < 4; ++k) { pc_[k] = dalloc<float>(n_max_); DR_HIP(hipHostMalloc((void **)&h_pc_[k], (size_t)n_max_ * 4, hipHostMallocDefault)); }
    for (int k = 0; k < 7; ++k) warped_[k] = dalloc<float>(n_max_);
    dInew_ = dalloc<float>((size_t)3 * w_ * h_);
    DR_HIP(hipHostMalloc((void **)&h_dInew_, (size_t)12 * w_ * h_, hipHostMallocDefault));
    partial_ = dalloc<double>((size_t)std::max(cdiv(n_max_, kTrkThreads), kTrkMaxBlocks) * 45);
    sums_ = dalloc<double>(64);
    DR_HIP(hipHostMalloc((void **)&h_sums_, 64 * 8, hipHostMallocDefault));
  }
  void set_reference(int n, const float *u, const float *v, const float *id, const float *col, float ref_exposure, const double *ref_aff) {
    need_init();
    if (n < 0 || n > n_max_) fail(DR_ERR_ARG, "Called CudaCoarseTracker::setReference with n > n_max points.");
    if (n && (!u || !v || !id || !col)) fail(DR_ERR_ARG, "setReference: null argument");
    if (!ref_aff) fail(DR_ERR_ARG, "setReference: null argument");
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(stream_));  // the pinned staging buffers may still be in flight
    const float *src[4] = {u, v, id, col};
    for (int k = 0; k < 4; ++k) {
      memcpy(h_pc_[k], src[k], (size_t)n * 4);
      DR_HIP(hipMemcpyAsync(pc_[k], h_pc_[k], (size_t)n * 4, hipMemcpyHostToDevice, stream_));
    }
    n_ = n; ref_exposure_ = ref_exposure; ref_aff_[0] = ref_aff[0]; ref_aff_[1] = ref_aff[1];
    num_terms_in_warped_ = 0;
  }
  void set_new(const float *dInew) {
    need_init();
    if (!dInew) fail(DR_ERR_ARG, "setNew: null argument");
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(stream_));
    memcpy(h_dInew_, dInew, (size_t)12 * w_ * h_);
    DR_HIP(hipMemcpyAsync(dInew_, h_dInew_, (size_t)12 * w_ * h_, hipMemcpyHostToDevice, stream_));
  }
  // AffLight::fromToVecExposure, cuda_coarse_tracker.cpp:40-49
  void aff_ll(float new_exposure, const double *g2T, float &ax, float &ay) const {
    float eF = ref_exposure_, eT = new_exposure;
    if (eF == 0 || eT == 0) eT = eF = 1;
    const double a = std::exp(g2T[0] - ref_aff_[0]) * eT / eF;
    const double b = g2T[1] - a * ref_aff_[1];
    ax = (float)a; ay = (float)b;
  }
  void calc_res(const double *refToNew, float new_exposure, const double *aff_g2l, float cutoffTH, double *out6, double *sums7) {
    need_init();
    if (!have_k_) fail(DR_ERR_PROTOCOL, "calcRes: setK has not been called");
    if (!refToNew || !aff_g2l || !out6) fail(DR_ERR_ARG, "calcRes: null argument");
    DR_HIP(hipSetDevice(device_));
    TrkResArgs a;
    float Kif[9];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) a.R[3 * r + c] = (float)refToNew[4 * r + c]; a.t[r] = (float)refToNew[4 * r + 3]; }
    for (int i = 0; i < 9; ++i) Kif[i] = a.Ki[i] = (float)Ki_[i];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {  // RKi = R * Ki per thread in the reference (:96-97): same for all
      float s = 0.0f;
      for (int k = 0; k < 3; ++k) s += a.R[3 * r + k] * Kif[3 * k + c];
      a.RKi[3 * r + c] = s;
    }
    aff_ll(new_exposure, aff_g2l, a.ax, a.ay);
    a.huber = huber_; a.cutoffTH = cutoffTH;
    a.maxEnergy = 2 * huber_ * cutoffTH - huber_ * huber_;  // energy for r = cutoff (:230)
    const int nb = std::max(1, cdiv(n_, kTrkThreads));
    hipLaunchKernelGGL(k_trk_res, dim3(nb), dim3(kTrkThreads), 0, stream_, dev(), a, partial_);
    hipLaunchKernelGGL(k_trk_fold, dim3(7), dim3(64), 0, stream_, partial_, nb, 7, sums_);
    DR_HIP(hipMemcpyAsync(h_sums_, sums_, 7 * 8, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipStreamSynchronize(stream_));
    const double *S = h_sums_;
    out6[0] = S[0]; out6[1] = S[1]; out6[2] = S[4] / S[6]; out6[3] = 0; out6[4] = S[5] / S[6]; out6[5] = S[3] / S[1];
    if (sums7) memcpy(sums7, S, 56);
    num_terms_in_warped_ = (int)S[2];
  }
  void calc_g(double *H, double *b, float new_exposure, const double *aff_g2l, double *raw45) {
    need_init();
    if (!H || !b || !aff_g2l) fail(DR_ERR_ARG, "calcG: null argument");
    DR_HIP(hipSetDevice(device_));
    float ax, ay;
    aff_ll(new_exposure, aff_g2l, ax, ay);
    const int nb = std::max(1, std::min(cdiv(n_, kTrkThreads), kTrkMaxBlocks));
    hipLaunchKernelGGL(k_trk_g, dim3(nb), dim3(kTrkThreads), 0, stream_, dev(), ax, (float)ref_aff_[1], partial_);
    hipLaunchKernelGGL(k_trk_fold, dim3(45), dim3(64), 0, stream_, partial_, nb, 45, sums_);
    DR_HIP(hipMemcpyAsync(h_sums_, sums_, 45 * 8, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipStreamSynchronize(stream_));
    const double *acc = h_sums_;
    if (raw45) memcpy(raw45, acc, 45 * 8);
    const double factor = 1.0 / num_terms_in_warped_;
    for (int r = 0; r < 8; ++r) {
      for (int c = 0; c < 8; ++c) {
        const int lo = std::min(r, c), hi = std::max(r, c);
        H[8 * r + c] = acc[lo * 9 + hi - lo * (lo + 1) / 2] * factor;
      }
      b[r] = acc[r * 9 + 8 - r * (r + 1) / 2] * factor;
    }
    const double s[8] = {1, 1, 1, 0.5, 0.5, 0.5, 10, 1000};  // SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_A, SCALE_B (:11-18, :343-354)
    for (int r = 0; r < 8; ++r) { for (int c = 0; c < 8; ++c) H[8 * r + c] *= s[r] * s[c]; b[r] *= s[r]; }
  }
  // CoarseTracker.cpp:655-725 on the device.  depth / idepth0 / dIp0: host pointers, or device pointers if on_device.
  int append_dense(const float *depth, const float *KRKi, const float *Kt, int step, int dense_only, const float *idepth0, const float *dIp0, int on_device) {
    need_init();
    if (!depth || !KRKi || !Kt || !dIp0 || (!dense_only && !idepth0) || step <= 0) fail(DR_ERR_ARG, "appendDenseReference: bad argument");
    if (w_ < 8 || h_ < 8) fail(DR_ERR_ARG, "appendDenseReference: image too small");
    DR_HIP(hipSetDevice(device_));
    const size_t npix = (size_t)w_ * h_;
    if (!zbuf_) {
      zbuf_ = dalloc<unsigned>(npix); rowcnt_ = dalloc<int>(h_ + 2);
      up_depth_ = dalloc<float>(npix); up_idepth_ = dalloc<float>(npix); up_dip_ = dalloc<float>(npix * 3);
    }
    const float *d_depth = depth, *d_id = idepth0, *d_dip = dIp0;
    if (!on_device) {
      DR_HIP(hipMemcpyAsync(up_depth_, depth, npix * 4, hipMemcpyHostToDevice, stream_));
      DR_HIP(hipMemcpyAsync(up_dip_, dIp0, npix * 12, hipMemcpyHostToDevice, stream_));
      if (idepth0) DR_HIP(hipMemcpyAsync(up_idepth_, idepth0, npix * 4, hipMemcpyHostToDevice, stream_));
      d_depth = up_depth_; d_dip = up_dip_; d_id = idepth0 ? up_idepth_ : nullptr;
    }
    TrkProjArgs a;
    memcpy(a.KRKi, KRKi, 36); memcpy(a.Kt, Kt, 12); a.step = step;
    hipLaunchKernelGGL(k_trk_fill, dim3(256), dim3(256), 0, stream_, zbuf_, npix, 0x7f800000u);
    const int nl = cdiv(w_, step) * cdiv(h_, step);
    hipLaunchKernelGGL(k_trk_project, dim3(cdiv(nl, 256)), dim3(256), 0, stream_, d_depth, w_, h_, a, zbuf_);
    const int nrows = h_ - 4;
    hipLaunchKernelGGL(k_trk_row_count, dim3(nrows), dim3(kTrkThreads), 0, stream_, zbuf_, d_id, dense_only, w_, h_, rowcnt_ + 2);
    hipLaunchKernelGGL(k_trk_row_scan, dim3(1), dim3(1), 0, stream_, rowcnt_ + 2, nrows, n_, n_max_, rowcnt_);
    hipLaunchKernelGGL(k_trk_row_write, dim3(nrows), dim3(kTrkThreads), 0, stream_, zbuf_, d_id, d_dip, dense_only, w_, h_, rowcnt_ + 2, rowcnt_,
                       pc_[0], pc_[1], pc_[2], pc_[3]);
    int res[2];
    DR_HIP(hipMemcpyAsync(res, rowcnt_, 8, hipMemcpyDeviceToHost, stream_));
    DR_HIP(hipStreamSynchronize(stream_));
    if (res[1]) fail(DR_ERR_CAPACITY, "appendDenseReference: %d points > n_max %d", res[0], n_max_);
    n_ = res[0];
    return n_;
  }
  void get_points(float *u, float *v, float *id, float *col, int cap, int *n) {
    need_init();
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(stream_));
    if (n) *n = n_;
    if (cap < n_) fail(DR_ERR_ARG, "get_points: capacity %d < %d points", cap, n_);
    float *dst[4] = {u, v, id, col};
    for (int k = 0; k < 4; ++k) if (dst[k]) DR_HIP(hipMemcpy(dst[k], pc_[k], (size_t)n_ * 4, hipMemcpyDeviceToHost));
  }
  void get_warped(int k, float *out, int cap) {
    need_init();
    if (k < 0 || k > 6 || !out || cap < n_) fail(DR_ERR_ARG, "get_warped: bad argument");
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(stream_));
    DR_HIP(hipMemcpy(out, warped_[k], (size_t)n_ * 4, hipMemcpyDeviceToHost));
  }
  void get_zbuffer(float *out) {  // projected depth map of the last appendDenseReference, -1 = empty
    if (!zbuf_ || !out) fail(DR_ERR_PROTOCOL, "get_zbuffer: no dense reference has been appended");
    DR_HIP(hipSetDevice(device_));
    DR_HIP(hipStreamSynchronize(stream_));
    DR_HIP(hipMemcpy(out, zbuf_, (size_t)w_ * h_ * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)w_ * h_; ++i) if (std::isinf(out[i])) out[i] = -1.f;
  }
  void synchronize() { DR_HIP(hipSetDevice(device_)); DR_HIP(hipStreamSynchronize(stream_)); }
  void start_timing() {  // cuda_coarse_tracker.cpp:374-381
    if (timing_) fail(DR_ERR_PROTOCOL, "CudaCoarseTracker::startTiming. Did not destroy events before correctly.");
    DR_HIP(hipSetDevice(device_));
    if (!ev0_) { DR_HIP(hipEventCreate(&ev0_)); DR_HIP(hipEventCreate(&ev1_)); }
    DR_HIP(hipEventRecord(ev0_, stream_));
    timing_ = true;
  }
  float end_timing_ms() {
    if (!timing_) fail(DR_ERR_PROTOCOL, "CudaCoarseTracker::endTimingMilliseconds. Did not start before.");
    DR_HIP(hipEventRecord(ev1_, stream_));
    DR_HIP(hipEventSynchronize(ev1_));
    float ms = -1.f;
    DR_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
    timing_ = false;
    return ms;
  }

 private:
  void need_init() { if (!n_max_) fail(DR_ERR_PROTOCOL, "CudaCoarseTracker: init has not been called"); }
  TrkDev dev() const {
    TrkDev d;
    d.w = w_; d.h = h_; d.n = n_; d.fx = fx_; d.fy = fy_; d.cx = cx_; d.cy = cy_;
    d.pc_u = pc_[0]; d.pc_v = pc_[1]; d.pc_idepth = pc_[2]; d.pc_color = pc_[3];
    d.dInew = dInew_;
    for (int k = 0; k < 7; ++k) d.warped[k] = warped_[k];
    return d;
  }
  void release() {
    for (int k = 0; k < 4; ++k) { (void)hipFree(pc_[k]); if (h_pc_[k]) (void)hipHostFree(h_pc_[k]); }
    for (int k = 0; k < 7; ++k) (void)hipFree(warped_[k]);
    (void)hipFree(dInew_); if (h_dInew_) (void)hipHostFree(h_dInew_);
    (void)hipFree(partial_); (void)hipFree(sums_); if (h_sums_) (void)hipHostFree(h_sums_);
    (void)hipFree(zbuf_); (void)hipFree(rowcnt_); (void)hipFree(up_depth_); (void)hipFree(up_idepth_); (void)hipFree(up_dip_);
  }
  int device_, w_, h_;
  float huber_;
  [[maybe_unused]] float cutoff_;  // setting_coarseCutoffTH: stored, never read -- as in the reference (calcRes gets cutoffTH per call)
  float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
  double Ki_[9] = {};
  bool have_k_ = false, timing_ = false;
  int n_ = 0, n_max_ = 0, num_terms_in_warped_ = 0;
  float ref_exposure_ = 0;
  double ref_aff_[2] = {0, 0};
  hipStream_t stream_ = nullptr;
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
  float *pc_[4] = {}, *h_pc_[4] = {}, *warped_[7] = {};
  float *dInew_ = nullptr, *h_dInew_ = nullptr;
  double *partial_ = nullptr, *sums_ = nullptr, *h_sums_ = nullptr;
  unsigned *zbuf_ = nullptr;
  int *rowcnt_ = nullptr;
  float *up_depth_ = nullptr, *up_idepth_ = nullptr, *up_dip_ = nullptr;
};

}  // namespace dr

using dr::guarded;
struct drt_s {
  std::unique_ptr<dr::TrackerEngine> e;
};
// every C-ABI entry point goes through this: a NULL handle is an argument error, not a crash
static inline dr::TrackerEngine *eng(drt_s *h) {
  if (!h || !h->e) dr::fail(DR_ERR_ARG, "NULL handle");
  return h->e.get();
}

extern "C" {

int drt_create(int w, int h, float setting_huberTH, float setting_coarseCutoffTH, int device, drt_t **out) {
  return guarded([&] {
    if (!out) dr::fail(DR_ERR_ARG, "drt_create: null argument");
    auto *t = new drt_s();
    try { t->e.reset(new dr::TrackerEngine(w, h, setting_huberTH, setting_coarseCutoffTH, device)); } catch (...) { delete t; throw; }
    *out = t;
  });
}
void drt_destroy(drt_t *t) { delete t; }
int drt_set_k(drt_t *t, int w, int h, float fx, float fy, float cx, float cy) { return guarded([&] { eng(t)->set_k(w, h, fx, fy, cx, cy); }); }
int drt_init(drt_t *t, int n_max) { return guarded([&] { eng(t)->init(n_max); }); }
int drt_set_reference(drt_t *t, int n, const float *pc_u, const float *pc_v, const float *pc_idepth, const float *pc_color, float ref_exposure,
                      const double ref_aff_g2l[2]) {
  return guarded([&] { eng(t)->set_reference(n, pc_u, pc_v, pc_idepth, pc_color, ref_exposure, ref_aff_g2l); });
}
int drt_set_new(drt_t *t, const float *dInew) { return guarded([&] { eng(t)->set_new(dInew); }); }
int drt_calc_res(drt_t *t, const double refToNew[16], float new_exposure, const double aff_g2l[2], float cutoffTH, double out6[6], double sums7[7]) {
  return guarded([&] { eng(t)->calc_res(refToNew, new_exposure, aff_g2l, cutoffTH, out6, sums7); });
}
int drt_calc_g(drt_t *t, double H_out[64], double b_out[8], float new_exposure, const double aff_g2l[2], double raw45[45]) {
  return guarded([&] { eng(t)->calc_g(H_out, b_out, new_exposure, aff_g2l, raw45); });
}
int drt_append_dense_reference(drt_t *t, const float *depth, const float KRKi[9], const float Kt[3], int step, int dense_only, const float *idepth0,
                               const float *dIp0, int on_device, int *n_out) {
  return guarded([&] { const int n = eng(t)->append_dense(depth, KRKi, Kt, step, dense_only, idepth0, dIp0, on_device); if (n_out) *n_out = n; });
}
int drt_get_points(drt_t *t, float *pc_u, float *pc_v, float *pc_idepth, float *pc_color, int cap, int *n) {
  return guarded([&] { eng(t)->get_points(pc_u, pc_v, pc_idepth, pc_color, cap, n); });
}
int drt_get_warped(drt_t *t, int which, float *out, int cap) { return guarded([&] { eng(t)->get_warped(which, out, cap); }); }
int drt_get_zbuffer(drt_t *t, float *out) { return guarded([&] { eng(t)->get_zbuffer(out); }); }
int drt_synchronize(drt_t *t) { return guarded([&] { eng(t)->synchronize(); }); }
int drt_start_timing(drt_t *t) { return guarded([&] { eng(t)->start_timing(); }); }
int drt_end_timing_ms(drt_t *t, float *ms) { return guarded([&] { const float v = eng(t)->end_timing_ms(); if (ms) *ms = v; }); }

}  // extern "C"
