// TEST INFRASTRUCTURE (oracle/_ref/libdense_handoff_ref.so, oracle/Makefile.ref).  The few Eigen / Sophus types the dense-depth branch of
// CoarseTracker::setCoarseTrackingRef (tandem/src/FullSystem/CoarseTracker.cpp:654-723) is written in, so that the reference's OWN lines
// compile in this image (Eigen and Sophus are absent).  Only what that block uses; names as in tandem/src/util/NumType.h:48,97,100,120.
//
// The ONE numerical choice made here: the order of a 3-term inner product.  Eigen evaluates a fixed-size 3-vector reduction without
// vectorisation through redux_novec_unroller<Func, Evaluator, 0, 3>, which splits [0,3) into [0,1) + [1,3): a0 + (a1 + a2).  That order is the
// default below; -DHANDOFF_SUM_LEFT builds (a0 + a1) + a2 instead.  tests/test_ref_handoff.py runs both and caps what the choice can move.
// Everything else -- the z-buffer rule, the rounding of the projected pixel, the bounds test, the pre-incremented append -- is the reference's text.
#pragma once
#include <cmath>

#ifdef HANDOFF_SUM_LEFT
#define HANDOFF_SUM3(a0, a1, a2) (((a0) + (a1)) + (a2))
#else
#define HANDOFF_SUM3(a0, a1, a2) ((a0) + ((a1) + (a2)))
#endif

template <class T> struct Vec3T {
  T v[3];
  Vec3T() : v{0, 0, 0} {}
  Vec3T(T a, T b, T c) : v{a, b, c} {}
  T &operator[](int i) { return v[i]; }
  T operator[](int i) const { return v[i]; }
  T &operator()(int i) { return v[i]; }
  T operator()(int i) const { return v[i]; }
  const T *data() const { return v; }
  template <class U> Vec3T<U> cast() const { return Vec3T<U>((U)v[0], (U)v[1], (U)v[2]); }
  Vec3T operator+(const Vec3T &o) const { return Vec3T(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
};
template <class T> struct Mat33T {
  T m[3][3];
  Mat33T() : m{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}} {}
  T &operator()(int r, int c) { return m[r][c]; }
  T operator()(int r, int c) const { return m[r][c]; }
  template <class U> Mat33T<U> cast() const {
    Mat33T<U> o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[r][c] = (U)m[r][c];
    return o;
  }
  Mat33T operator*(const Mat33T &b) const {
    Mat33T o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[r][c] = HANDOFF_SUM3(m[r][0] * b.m[0][c], m[r][1] * b.m[1][c], m[r][2] * b.m[2][c]);
    return o;
  }
  Vec3T<T> operator*(const Vec3T<T> &x) const {
    Vec3T<T> o;
    for (int r = 0; r < 3; r++) o.v[r] = HANDOFF_SUM3(m[r][0] * x.v[0], m[r][1] * x.v[1], m[r][2] * x.v[2]);
    return o;
  }
  Mat33T transpose() const {
    Mat33T o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[r][c] = m[c][r];
    return o;
  }
};
typedef Mat33T<float> Mat33f;
typedef Vec3T<float> Vec3f;
struct Mat44 {
  double m[4][4];
  Mat44() { for (auto &r : m) for (double &x : r) x = 0; }
  Mat44 &matrix() { return *this; }
  const Mat44 &matrix() const { return *this; }
  double &operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
};
// Rigid transform in double (Sophus::SE3d keeps a quaternion; its rotation matrix differs from this one at the 1e-16 level, which only moves
// the float casts KRKi / Kt -- and those are INPUTS of the restatement under test, taken from this library).
struct SE3 {
  Mat33T<double> R;
  Vec3T<double> t;
  SE3() { R(0, 0) = R(1, 1) = R(2, 2) = 1; }
  explicit SE3(const Mat44 &T) {
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R(r, c) = T(r, c); t[r] = T(r, 3); }
  }
  SE3 inverse() const {
    SE3 o;
    o.R = R.transpose();
    const Vec3T<double> x = o.R * t;
    o.t = Vec3T<double>(-x[0], -x[1], -x[2]);
    return o;
  }
  SE3 operator*(const SE3 &b) const {
    SE3 o;
    o.R = R * b.R;
    o.t = (R * b.t) + t;
    return o;
  }
  Mat33T<double> rotationMatrix() const { return R; }
  Vec3T<double> translation() const { return t; }
};
