// mini_eigen.h — TEST INFRASTRUCTURE. A small, eager (no expression templates) implementation of the subset of the
// Eigen 3 public API that the Cerberus factor sources use, written from the API documentation so that the
// reference's own .cpp files can be compiled UNMODIFIED from /root/reference (Eigen itself is not installed in this
// image and there is no network). Only oracle/ref_build uses it; nothing in the product does.
//
// Semantics kept: column-major default storage, RowMajor option, Map over raw buffers, fixed/dynamic sizes,
// coefficient order of Quaternion (x, y, z, w), Quaternion <-> rotation matrix formulas, LLT lower factor,
// ascending eigenvalues of SelfAdjointEigenSolver. Every arithmetic expression is evaluated immediately into a
// plain Matrix, so there are no aliasing surprises; views (block/segment/transpose/col/...) are pointer + strides.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <numeric>
#include <type_traits>
#include <vector>

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2, Unaligned = 0, Aligned = 16 };

template <class T, int R, int C, int Opt = 0> class Matrix;
template <class T, int R, int C> class View;
template <class T, int R, int C> class Array;
template <class T, int N> class DiagonalMatrix;
template <class T, int Opt = 0> class Quaternion;
template <class Derived> class MatrixBase;
template <class Derived> class QuaternionBase;
template <class PlainType, int MapOptions = 0, class StrideType = void> class Map;

namespace internal {
template <class D> struct traits;
template <class T, int R, int C, int O> struct traits<Matrix<T, R, C, O>> { typedef T Scalar; enum { Rows = R, Cols = C, IsView = 0 }; };
template <class T, int R, int C> struct traits<View<T, R, C>> { typedef T Scalar; enum { Rows = R, Cols = C, IsView = 1 }; };
template <class D> struct traits<const D> : traits<D> {};
constexpr int pick(int a, int b) { return a != Dynamic ? a : b; }
constexpr int mul(int a, int b) { return (a == Dynamic || b == Dynamic) ? Dynamic : a * b; }
constexpr int minsz(int a, int b) { return (a == Dynamic || b == Dynamic) ? Dynamic : (a < b ? a : b); }
struct SizeTag {};

template <class T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)> struct Storage;
template <class T, int R, int C> struct Storage<T, R, C, false> {
  T d[R * C > 0 ? R * C : 1] = {};
  static constexpr Index rows() { return R; }
  static constexpr Index cols() { return C; }
  void resize(Index r, Index c) { assert(r == R && c == C); (void)r; (void)c; }
  T *ptr() const { return const_cast<T *>(d); }
};
template <class T, int R, int C> struct Storage<T, R, C, true> {
  std::vector<T> d;
  Index r_ = (R == Dynamic ? 0 : R), c_ = (C == Dynamic ? 0 : C);
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  void resize(Index r, Index c) {
    assert((R == Dynamic || r == R) && (C == Dynamic || c == C));
    if (r != r_ || c != c_ || (Index)d.size() != r * c) { r_ = r; c_ = c; d.assign((size_t)(r * c), T(0)); }
  }
  T *ptr() const { return const_cast<T *>(d.data()); }
};
}  // namespace internal

template <class D> struct CommaInitializer;

// ---------------------------------------------------------------------------------------------------- MatrixBase
template <class Derived> class MatrixBase {
 public:
  typedef typename internal::traits<Derived>::Scalar Scalar;
  enum {
    RowsAtCompileTime = internal::traits<Derived>::Rows,
    ColsAtCompileTime = internal::traits<Derived>::Cols,
    SizeAtCompileTime = internal::mul(internal::traits<Derived>::Rows, internal::traits<Derived>::Cols),
    IsVectorAtCompileTime = (internal::traits<Derived>::Rows == 1 || internal::traits<Derived>::Cols == 1)
  };
  typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;
  typedef Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> TransposedPlain;

  Derived &derived() { return *static_cast<Derived *>(this); }
  const Derived &derived() const { return *static_cast<const Derived *>(this); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }
  Index rs() const { return derived().rstride_(); }
  Index cs() const { return derived().cstride_(); }
  Scalar *base() const { return derived().ptr_(); }
  Scalar &ref(Index i, Index j) const {
    assert(i >= 0 && i < rows() && j >= 0 && j < cols());
    return base()[i * rs() + j * cs()];
  }
  Scalar &operator()(Index i, Index j) { return ref(i, j); }
  const Scalar &operator()(Index i, Index j) const { return ref(i, j); }
  Scalar &vref(Index i) const { return (ColsAtCompileTime == 1 || (RowsAtCompileTime != 1 && cols() == 1)) ? ref(i, 0) : ref(0, i); }
  Scalar &operator()(Index i) { return vref(i); }
  const Scalar &operator()(Index i) const { return vref(i); }
  Scalar &operator[](Index i) { return vref(i); }
  const Scalar &operator[](Index i) const { return vref(i); }
  Scalar &x() { return vref(0); }
  Scalar &y() { return vref(1); }
  Scalar &z() { return vref(2); }
  Scalar &w() { return vref(3); }
  const Scalar &x() const { return vref(0); }
  const Scalar &y() const { return vref(1); }
  const Scalar &z() const { return vref(2); }
  const Scalar &w() const { return vref(3); }
  const Scalar &coeff(Index i, Index j) const { return ref(i, j); }
  Scalar &coeffRef(Index i, Index j) { return ref(i, j); }

  // ---- views
  template <int BR, int BC> View<Scalar, BR, BC> block(Index i, Index j) const {
    assert(i >= 0 && j >= 0 && i + BR <= rows() && j + BC <= cols());
    return View<Scalar, BR, BC>(base() + i * rs() + j * cs(), BR, BC, rs(), cs());
  }
  template <int BR, int BC> View<Scalar, BR, BC> block(Index i, Index j, Index, Index) const { return block<BR, BC>(i, j); }
  View<Scalar, Dynamic, Dynamic> block(Index i, Index j, Index r, Index c) const {
    assert(i >= 0 && j >= 0 && r >= 0 && c >= 0 && i + r <= rows() && j + c <= cols());
    return View<Scalar, Dynamic, Dynamic>(base() + i * rs() + j * cs(), r, c, rs(), cs());
  }
  enum { IsRowVec = (RowsAtCompileTime == 1 && ColsAtCompileTime != 1) };
  template <int N> struct Seg { typedef View<Scalar, IsRowVec ? 1 : N, IsRowVec ? N : 1> type; };
  template <int N> typename Seg<N>::type segment(Index i) const {
    assert(i >= 0 && i + N <= size());
    return IsRowVec ? typename Seg<N>::type(base() + i * cs(), 1, N, rs(), cs()) : typename Seg<N>::type(base() + i * rs(), N, 1, rs(), cs());
  }
  template <int N> typename Seg<N>::type segment(Index i, Index) const { return segment<N>(i); }
  typename Seg<Dynamic>::type segment(Index i, Index n) const {
    assert(i >= 0 && n >= 0 && i + n <= size());
    return IsRowVec ? typename Seg<Dynamic>::type(base() + i * cs(), 1, n, rs(), cs()) : typename Seg<Dynamic>::type(base() + i * rs(), n, 1, rs(), cs());
  }
  template <int N> typename Seg<N>::type head() const { return segment<N>(0); }
  template <int N> typename Seg<N>::type tail() const { return segment<N>(size() - N); }
  typename Seg<Dynamic>::type head(Index n) const { return segment(0, n); }
  typename Seg<Dynamic>::type tail(Index n) const { return segment(size() - n, n); }
  View<Scalar, RowsAtCompileTime, 1> col(Index j) const { return View<Scalar, RowsAtCompileTime, 1>(base() + j * cs(), rows(), 1, rs(), cs()); }
  View<Scalar, 1, ColsAtCompileTime> row(Index i) const { return View<Scalar, 1, ColsAtCompileTime>(base() + i * rs(), 1, cols(), rs(), cs()); }
  template <int N> View<Scalar, RowsAtCompileTime, N> leftCols() const { return View<Scalar, RowsAtCompileTime, N>(base(), rows(), N, rs(), cs()); }
  template <int N> View<Scalar, RowsAtCompileTime, N> rightCols() const { return View<Scalar, RowsAtCompileTime, N>(base() + (cols() - N) * cs(), rows(), N, rs(), cs()); }
  template <int N> View<Scalar, RowsAtCompileTime, N> middleCols(Index j) const { return View<Scalar, RowsAtCompileTime, N>(base() + j * cs(), rows(), N, rs(), cs()); }
  View<Scalar, RowsAtCompileTime, Dynamic> leftCols(Index n) const { return View<Scalar, RowsAtCompileTime, Dynamic>(base(), rows(), n, rs(), cs()); }
  View<Scalar, RowsAtCompileTime, Dynamic> rightCols(Index n) const { return View<Scalar, RowsAtCompileTime, Dynamic>(base() + (cols() - n) * cs(), rows(), n, rs(), cs()); }
  View<Scalar, RowsAtCompileTime, Dynamic> middleCols(Index j, Index n) const { return View<Scalar, RowsAtCompileTime, Dynamic>(base() + j * cs(), rows(), n, rs(), cs()); }
  template <int N> View<Scalar, N, ColsAtCompileTime> topRows() const { return View<Scalar, N, ColsAtCompileTime>(base(), N, cols(), rs(), cs()); }
  template <int N> View<Scalar, N, ColsAtCompileTime> bottomRows() const { return View<Scalar, N, ColsAtCompileTime>(base() + (rows() - N) * rs(), N, cols(), rs(), cs()); }
  View<Scalar, Dynamic, ColsAtCompileTime> topRows(Index n) const { return View<Scalar, Dynamic, ColsAtCompileTime>(base(), n, cols(), rs(), cs()); }
  View<Scalar, Dynamic, ColsAtCompileTime> bottomRows(Index n) const { return View<Scalar, Dynamic, ColsAtCompileTime>(base() + (rows() - n) * rs(), n, cols(), rs(), cs()); }
  View<Scalar, Dynamic, ColsAtCompileTime> middleRows(Index i, Index n) const { return View<Scalar, Dynamic, ColsAtCompileTime>(base() + i * rs(), n, cols(), rs(), cs()); }
  template <int BR, int BC> View<Scalar, BR, BC> topLeftCorner() const { return block<BR, BC>(0, 0); }
  template <int BR, int BC> View<Scalar, BR, BC> topRightCorner() const { return block<BR, BC>(0, cols() - BC); }
  template <int BR, int BC> View<Scalar, BR, BC> bottomLeftCorner() const { return block<BR, BC>(rows() - BR, 0); }
  template <int BR, int BC> View<Scalar, BR, BC> bottomRightCorner() const { return block<BR, BC>(rows() - BR, cols() - BC); }
  View<Scalar, Dynamic, Dynamic> topLeftCorner(Index r, Index c) const { return block(0, 0, r, c); }
  View<Scalar, Dynamic, Dynamic> bottomRightCorner(Index r, Index c) const { return block(rows() - r, cols() - c, r, c); }
  View<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const { return View<Scalar, ColsAtCompileTime, RowsAtCompileTime>(base(), cols(), rows(), cs(), rs()); }
  View<Scalar, ColsAtCompileTime, RowsAtCompileTime> adjoint() const { return transpose(); }
  View<Scalar, internal::minsz(RowsAtCompileTime, ColsAtCompileTime), 1> diagonal() const {
    return View<Scalar, internal::minsz(RowsAtCompileTime, ColsAtCompileTime), 1>(base(), std::min(rows(), cols()), 1, rs() + cs(), 0);
  }
  PlainObject eval() const { return PlainObject(*this); }
  const Derived &noalias() const { return derived(); }
  Derived &noalias() { return derived(); }

  // ---- in-place
  Derived &setZero() { return setConstant(Scalar(0)); }
  Derived &setOnes() { return setConstant(Scalar(1)); }
  Derived &setConstant(const Scalar &v) {
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) = v;
    return derived();
  }
  Derived &fill(const Scalar &v) { return setConstant(v); }
  Derived &setIdentity() {
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) = (i == j) ? Scalar(1) : Scalar(0);
    return derived();
  }
  Derived &setRandom() {
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) = Scalar(2.0 * std::rand() / RAND_MAX - 1.0);
    return derived();
  }
  template <class O> Derived &operator+=(const MatrixBase<O> &o) {
    typename MatrixBase<O>::PlainObject t(o);
    assert(t.rows() == rows() && t.cols() == cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) += t.ref(i, j);
    return derived();
  }
  template <class O> Derived &operator-=(const MatrixBase<O> &o) {
    typename MatrixBase<O>::PlainObject t(o);
    assert(t.rows() == rows() && t.cols() == cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) -= t.ref(i, j);
    return derived();
  }
  template <class O> Derived &operator*=(const MatrixBase<O> &o) { derived() = PlainObject((*this) * o); return derived(); }
  Derived &operator*=(const Scalar &s) {
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) *= s;
    return derived();
  }
  Derived &operator/=(const Scalar &s) {
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) ref(i, j) /= s;
    return derived();
  }
  CommaInitializer<Derived> operator<<(const Scalar &s) const;
  template <class O> CommaInitializer<Derived> operator<<(const MatrixBase<O> &o) const;

  // ---- reductions
  Scalar sum() const { Scalar s(0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s += ref(i, j); return s; }
  Scalar mean() const { return sum() / Scalar(size()); }
  Scalar trace() const { Scalar s(0); for (Index i = 0; i < std::min(rows(), cols()); ++i) s += ref(i, i); return s; }
  Scalar squaredNorm() const { Scalar s(0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) s += ref(i, j) * ref(i, j); return s; }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  Scalar maxCoeff() const { Scalar m = ref(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) m = std::max(m, ref(i, j)); return m; }
  Scalar minCoeff() const { Scalar m = ref(0, 0); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) m = std::min(m, ref(i, j)); return m; }
  PlainObject normalized() const { PlainObject t(*this); Scalar n = t.norm(); if (n > Scalar(0)) t /= n; return t; }
  void normalize() { Scalar n = norm(); if (n > Scalar(0)) (*this) /= n; }
  template <class O> Scalar dot(const MatrixBase<O> &o) const {
    assert(size() == o.size());
    Scalar s(0); for (Index i = 0; i < size(); ++i) s += vref(i) * o.vref(i); return s;
  }
  template <class O> Matrix<Scalar, 3, 1> cross(const MatrixBase<O> &o) const {
    Matrix<Scalar, 3, 1> r;
    r(0) = vref(1) * o.vref(2) - vref(2) * o.vref(1);
    r(1) = vref(2) * o.vref(0) - vref(0) * o.vref(2);
    r(2) = vref(0) * o.vref(1) - vref(1) * o.vref(0);
    return r;
  }
  bool allFinite() const { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (!std::isfinite(ref(i, j))) return false; return true; }
  bool hasNaN() const { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (std::isnan(ref(i, j))) return true; return false; }

  // ---- coefficient-wise
  template <class F> PlainObject unary(F f) const {
    PlainObject t(internal::SizeTag(), rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) t.ref(i, j) = f(ref(i, j));
    return t;
  }
  PlainObject cwiseSqrt() const { return unary([](Scalar v) { return std::sqrt(v); }); }
  PlainObject cwiseAbs() const { return unary([](Scalar v) { return std::abs(v); }); }
  PlainObject cwiseInverse() const { return unary([](Scalar v) { return Scalar(1) / v; }); }
  template <class O> PlainObject cwiseProduct(const MatrixBase<O> &o) const {
    PlainObject t(internal::SizeTag(), rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) t.ref(i, j) = ref(i, j) * o.ref(i, j);
    return t;
  }
  template <class O> PlainObject cwiseQuotient(const MatrixBase<O> &o) const {
    PlainObject t(internal::SizeTag(), rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) t.ref(i, j) = ref(i, j) / o.ref(i, j);
    return t;
  }
  Array<Scalar, RowsAtCompileTime, ColsAtCompileTime> array() const;
  DiagonalMatrix<Scalar, SizeAtCompileTime> asDiagonal() const;
  template <int RF, int CF> Matrix<Scalar, internal::mul(RowsAtCompileTime, RF), internal::mul(ColsAtCompileTime, CF)> replicate() const {
    Matrix<Scalar, internal::mul(RowsAtCompileTime, RF), internal::mul(ColsAtCompileTime, CF)> t(internal::SizeTag(), rows() * RF, cols() * CF);
    for (Index j = 0; j < cols() * CF; ++j) for (Index i = 0; i < rows() * RF; ++i) t.ref(i, j) = ref(i % rows(), j % cols());
    return t;
  }

  // ---- dense solvers
  PlainObject inverse() const;
  Scalar determinant() const;
  template <class Dummy = void> auto jacobiSvd(unsigned flags = 0) const;
};

// ---------------------------------------------------------------------------------------------------- Matrix
template <class T, int R, int C, int Opt> class Matrix : public MatrixBase<Matrix<T, R, C, Opt>> {
  typedef MatrixBase<Matrix<T, R, C, Opt>> Base;
  internal::Storage<T, R, C> s_;
  enum { IsRowMajor = ((Opt & RowMajor) && R != 1 && C != 1) || (R == 1 && C != 1) };

 public:
  typedef T Scalar;
  Index rows_() const { return s_.rows(); }
  Index cols_() const { return s_.cols(); }
  Index rstride_() const { return IsRowMajor ? s_.cols() : 1; }
  Index cstride_() const { return IsRowMajor ? 1 : s_.rows(); }
  T *ptr_() const { return s_.ptr(); }
  T *data() { return s_.ptr(); }
  const T *data() const { return s_.ptr(); }

  Matrix() {}
  Matrix(internal::SizeTag, Index r, Index c) { s_.resize(r, c); }
  Matrix(const Matrix &o) = default;
  Matrix &operator=(const Matrix &o) = default;
  template <class O> Matrix(const MatrixBase<O> &o) { assign(o); }
  template <class O> Matrix &operator=(const MatrixBase<O> &o) { assign(o); return *this; }
  template <int AR, int AC> Matrix(const Array<T, AR, AC> &a);
  template <int AR, int AC> Matrix &operator=(const Array<T, AR, AC> &a);
  // one argument: size (vectors) — or the single coefficient of a 1x1
  template <class I, typename std::enable_if<std::is_integral<I>::value, int>::type = 0> explicit Matrix(I n) {
    if (R == 1 && C == 1) s_.ptr()[0] = T(n);
    else if (R == Dynamic || C == Dynamic) s_.resize(R == Dynamic ? (Index)n : R, C == Dynamic ? (R == Dynamic ? 1 : (Index)n) : C);
  }
  explicit Matrix(const T *p) { for (Index i = 0; i < this->size(); ++i) this->vref(i) = p[i]; }
  // two arguments: (rows, cols) for dynamic sizes, coefficients for fixed 2-vectors
  template <class A, class B> Matrix(const A &a, const B &b) {
    if (R != Dynamic && C != Dynamic) {
      if (R * C == 2 && (R == 1 || C == 1)) { s_.ptr()[0] = T(a); s_.ptr()[1] = T(b); }
      else assert((Index)a == R && (Index)b == C);   // fixed-size (rows, cols): sizes only
    } else s_.resize((Index)a, (Index)b);
  }
  Matrix(const T &a, const T &b, const T &c) { s_.resize(R == Dynamic ? 3 : R, C == Dynamic ? 1 : C); assert(this->size() == 3); T *p = s_.ptr(); p[0] = a; p[1] = b; p[2] = c; }
  Matrix(const T &a, const T &b, const T &c, const T &d) { s_.resize(R == Dynamic ? 4 : R, C == Dynamic ? 1 : C); assert(this->size() == 4); T *p = s_.ptr(); p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
  Matrix(std::initializer_list<T> l) {
    if (R == Dynamic || C == Dynamic) s_.resize(R == Dynamic ? (Index)l.size() : R, C == Dynamic ? (R == Dynamic ? 1 : (Index)l.size()) : C);
    assert((Index)l.size() == this->size());
    Index i = 0; for (const T &v : l) this->vref(i++) = v;
  }

  void resize(Index r, Index c) { s_.resize(r, c); }
  void resize(Index n) { s_.resize(R == Dynamic ? n : R, C == Dynamic ? (R == Dynamic ? 1 : n) : C); }
  void conservativeResize(Index r, Index c) {
    Matrix old(*this); s_.resize(r, c);
    for (Index j = 0; j < std::min(c, old.cols()); ++j) for (Index i = 0; i < std::min(r, old.rows()); ++i) this->ref(i, j) = old.ref(i, j);
  }
  template <class O> void assign(const MatrixBase<O> &o) {
    if ((const void *)&o == (const void *)this) return;
    if (internal::traits<O>::IsView) {  // a view may alias this object's storage: go through a temporary
      typename MatrixBase<O>::PlainObject t(internal::SizeTag(), o.rows(), o.cols());
      for (Index j = 0; j < o.cols(); ++j) for (Index i = 0; i < o.rows(); ++i) t.ref(i, j) = o.ref(i, j);
      copy_from(t);
    } else copy_from(o);
  }
  template <class O> void copy_from(const MatrixBase<O> &o) {
    Index r = o.rows(), c = o.cols();
    if ((R == 1 || C == 1) && ((R != Dynamic && R != r) || (C != Dynamic && C != c)) ) {  // vector <- transposed vector
      s_.resize(c, r);
      for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) this->ref(j, i) = T(o.ref(i, j));
      return;
    }
    s_.resize(r, c);
    for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) this->ref(i, j) = T(o.ref(i, j));
  }
  template <class NewT> Matrix<NewT, R, C, Opt> cast() const {
    Matrix<NewT, R, C, Opt> t(internal::SizeTag(), this->rows(), this->cols());
    for (Index j = 0; j < this->cols(); ++j) for (Index i = 0; i < this->rows(); ++i) t.ref(i, j) = NewT(this->ref(i, j));
    return t;
  }

  static Matrix Zero() { Matrix m; m.setZero(); return m; }
  static Matrix Zero(Index n) { Matrix m(n); m.setZero(); return m; }
  static Matrix Zero(Index r, Index c) { Matrix m(internal::SizeTag(), r, c); m.setZero(); return m; }
  static Matrix Ones() { Matrix m; m.setOnes(); return m; }
  static Matrix Ones(Index n) { Matrix m(n); m.setOnes(); return m; }
  static Matrix Ones(Index r, Index c) { Matrix m(internal::SizeTag(), r, c); m.setOnes(); return m; }
  static Matrix Constant(const T &v) { Matrix m; m.setConstant(v); return m; }
  static Matrix Constant(Index n, const T &v) { Matrix m(n); m.setConstant(v); return m; }
  static Matrix Constant(Index r, Index c, const T &v) { Matrix m(internal::SizeTag(), r, c); m.setConstant(v); return m; }
  static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
  static Matrix Identity(Index r, Index c) { Matrix m(internal::SizeTag(), r, c); m.setIdentity(); return m; }
  static Matrix Random() { Matrix m; m.setRandom(); return m; }
  static Matrix Random(Index n) { Matrix m(n); m.setRandom(); return m; }
  static Matrix Random(Index r, Index c) { Matrix m(internal::SizeTag(), r, c); m.setRandom(); return m; }
  static Matrix UnitX() { Matrix m; m.setZero(); m(0) = T(1); return m; }
  static Matrix UnitY() { Matrix m; m.setZero(); m(1) = T(1); return m; }
  static Matrix UnitZ() { Matrix m; m.setZero(); m(2) = T(1); return m; }
};

// ---------------------------------------------------------------------------------------------------- View
template <class T, int R, int C> class View : public MatrixBase<View<T, R, C>> {
 protected:
  T *p_;
  Index r_, c_, rs_, cs_;

 public:
  typedef T Scalar;
  View(T *p, Index r, Index c, Index rs, Index cs) : p_(p), r_(r), c_(c), rs_(rs), cs_(cs) {}
  View(const View &) = default;
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  Index rstride_() const { return rs_; }
  Index cstride_() const { return cs_; }
  T *ptr_() const { return p_; }
  T *data() const { return p_; }
  template <class O> void store(const MatrixBase<O> &o) const {
    typename MatrixBase<O>::PlainObject t(o);  // evaluate first: source may alias
    if (t.rows() == r_ && t.cols() == c_) {
      for (Index j = 0; j < c_; ++j) for (Index i = 0; i < r_; ++i) p_[i * rs_ + j * cs_] = t.ref(i, j);
    } else {
      assert((r_ == 1 || c_ == 1) && t.rows() == c_ && t.cols() == r_);
      for (Index j = 0; j < c_; ++j) for (Index i = 0; i < r_; ++i) p_[i * rs_ + j * cs_] = t.ref(j, i);
    }
  }
  View &operator=(const View &o) { store(o); return *this; }
  template <class O> View &operator=(const MatrixBase<O> &o) { store(o); return *this; }
  template <int AR, int AC> View &operator=(const Array<T, AR, AC> &a);
};

// ---------------------------------------------------------------------------------------------------- Map / Ref
template <class T, int R, int C, int O, int MO, class S> class Map<Matrix<T, R, C, O>, MO, S> : public View<T, R, C> {
  enum { IsRowMajor = ((O & RowMajor) && R != 1 && C != 1) || (R == 1 && C != 1) };

 public:
  Map(const T *p) : View<T, R, C>(const_cast<T *>(p), R, C, IsRowMajor ? C : 1, IsRowMajor ? 1 : R) { static_assert(R != Dynamic && C != Dynamic, "sizes required"); }
  Map(const T *p, Index n) : View<T, R, C>(const_cast<T *>(p), R == Dynamic ? n : R, C == Dynamic ? (R == Dynamic ? 1 : n) : C, 1, 1) {
    this->rs_ = IsRowMajor ? this->c_ : 1; this->cs_ = IsRowMajor ? 1 : this->r_;
  }
  Map(const T *p, Index r, Index c) : View<T, R, C>(const_cast<T *>(p), r, c, IsRowMajor ? c : 1, IsRowMajor ? 1 : r) {}
  Map(const Map &) = default;
  Map &operator=(const Map &o) { this->store(o); return *this; }
  template <class OO> Map &operator=(const MatrixBase<OO> &o) { this->store(o); return *this; }
};
template <class T, int R, int C, int O, int MO, class S> class Map<const Matrix<T, R, C, O>, MO, S> : public Map<Matrix<T, R, C, O>, MO, S> {
 public:
  using Map<Matrix<T, R, C, O>, MO, S>::Map;
};

// Ref<const M>: by-value copy of the argument (the reference only uses it for read-only parameters).
template <class M> class Ref : public std::remove_const<M>::type {
  typedef typename std::remove_const<M>::type Plain;

 public:
  template <class O> Ref(const MatrixBase<O> &o) : Plain(o) {}
  Ref(const Ref &) = default;
};

// ---------------------------------------------------------------------------------------------------- operators
#define ME_RES_SAME(A, B) Matrix<typename A::Scalar, internal::pick(A::RowsAtCompileTime, B::RowsAtCompileTime), internal::pick(A::ColsAtCompileTime, B::ColsAtCompileTime)>
template <class A, class B> ME_RES_SAME(MatrixBase<A>, MatrixBase<B>) operator+(const MatrixBase<A> &a, const MatrixBase<B> &b) {
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  ME_RES_SAME(MatrixBase<A>, MatrixBase<B>) t(internal::SizeTag(), a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) t.ref(i, j) = a.ref(i, j) + b.ref(i, j);
  return t;
}
template <class A, class B> ME_RES_SAME(MatrixBase<A>, MatrixBase<B>) operator-(const MatrixBase<A> &a, const MatrixBase<B> &b) {
  assert(a.rows() == b.rows() && a.cols() == b.cols());
  ME_RES_SAME(MatrixBase<A>, MatrixBase<B>) t(internal::SizeTag(), a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) t.ref(i, j) = a.ref(i, j) - b.ref(i, j);
  return t;
}
template <class A> typename MatrixBase<A>::PlainObject operator-(const MatrixBase<A> &a) { return a.unary([](typename MatrixBase<A>::Scalar v) { return -v; }); }
template <class A, class B>
Matrix<typename MatrixBase<A>::Scalar, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime> operator*(const MatrixBase<A> &a, const MatrixBase<B> &b) {
  typedef typename MatrixBase<A>::Scalar S;
  assert(a.cols() == b.rows());
  Matrix<S, MatrixBase<A>::RowsAtCompileTime, MatrixBase<B>::ColsAtCompileTime> t(internal::SizeTag(), a.rows(), b.cols());
  const Index n = a.rows(), m = b.cols(), k = a.cols();
  for (Index j = 0; j < m; ++j)
    for (Index i = 0; i < n; ++i) {
      S s(0);
      for (Index l = 0; l < k; ++l) s += a.ref(i, l) * b.ref(l, j);
      t.ref(i, j) = s;
    }
  return t;
}
template <class A> typename MatrixBase<A>::PlainObject operator*(const MatrixBase<A> &a, const typename MatrixBase<A>::Scalar &s) { return a.unary([s](typename MatrixBase<A>::Scalar v) { return v * s; }); }
template <class A> typename MatrixBase<A>::PlainObject operator*(const typename MatrixBase<A>::Scalar &s, const MatrixBase<A> &a) { return a.unary([s](typename MatrixBase<A>::Scalar v) { return s * v; }); }
template <class A> typename MatrixBase<A>::PlainObject operator/(const MatrixBase<A> &a, const typename MatrixBase<A>::Scalar &s) { return a.unary([s](typename MatrixBase<A>::Scalar v) { return v / s; }); }
template <class A, class B> bool operator==(const MatrixBase<A> &a, const MatrixBase<B> &b) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) if (a.ref(i, j) != b.ref(i, j)) return false;
  return true;
}
template <class A> std::ostream &operator<<(std::ostream &os, const MatrixBase<A> &a) {
  for (Index i = 0; i < a.rows(); ++i) {
    for (Index j = 0; j < a.cols(); ++j) os << (j ? " " : "") << a.ref(i, j);
    if (i + 1 < a.rows()) os << "\n";
  }
  return os;
}

// ---------------------------------------------------------------------------------------------------- comma init
template <class D> struct CommaInitializer {
  typedef typename MatrixBase<D>::Scalar Scalar;
  D &m; Index row, col, brows;
  CommaInitializer(D &mm, const Scalar &s) : m(mm), row(0), col(1), brows(1) { m.ref(0, 0) = s; }
  template <class O> CommaInitializer(D &mm, const MatrixBase<O> &o) : m(mm), row(0), col(o.cols()), brows(o.rows()) { m.block(0, 0, o.rows(), o.cols()) = o; }
  CommaInitializer &operator,(const Scalar &s) {
    if (col == m.cols()) { row += brows; col = 0; brows = 1; }
    m.ref(row, col++) = s;
    return *this;
  }
  template <class O> CommaInitializer &operator,(const MatrixBase<O> &o) {
    if (col == m.cols()) { row += brows; col = 0; brows = o.rows(); }
    m.block(row, col, o.rows(), o.cols()) = o;
    col += o.cols();
    return *this;
  }
  D &finished() { return m; }
};
template <class D> CommaInitializer<D> MatrixBase<D>::operator<<(const Scalar &s) const { return CommaInitializer<D>(const_cast<D &>(derived()), s); }
template <class D> template <class O> CommaInitializer<D> MatrixBase<D>::operator<<(const MatrixBase<O> &o) const { return CommaInitializer<D>(const_cast<D &>(derived()), o); }

// ---------------------------------------------------------------------------------------------------- Array
template <class T, int R, int C> class Array {
 public:
  typedef T Scalar;
  Matrix<T, R, C> m;
  Array() {}
  explicit Array(const Matrix<T, R, C> &mm) : m(mm) {}
  Index rows() const { return m.rows(); }
  Index cols() const { return m.cols(); }
  Index size() const { return m.size(); }
  T &operator()(Index i) { return m(i); }
  const T &operator()(Index i) const { return m(i); }
  T &operator()(Index i, Index j) { return m(i, j); }
  const T &operator()(Index i, Index j) const { return m(i, j); }
  template <class F> Array map(F f) const { return Array(m.unary(f)); }
  Array square() const { return map([](T v) { return v * v; }); }
  Array sqrt() const { return map([](T v) { return std::sqrt(v); }); }
  Array abs() const { return map([](T v) { return std::abs(v); }); }
  Array inverse() const { return map([](T v) { return T(1) / v; }); }
  Array exp() const { return map([](T v) { return std::exp(v); }); }
  T sum() const { return m.sum(); }
  T mean() const { return m.mean(); }
  T maxCoeff() const { return m.maxCoeff(); }
  T minCoeff() const { return m.minCoeff(); }
  Matrix<T, R, C> matrix() const { return m; }
  Array operator+(const T &s) const { return map([s](T v) { return v + s; }); }
  Array operator-(const T &s) const { return map([s](T v) { return v - s; }); }
  Array operator*(const T &s) const { return map([s](T v) { return v * s; }); }
  Array operator/(const T &s) const { return map([s](T v) { return v / s; }); }
  Array operator+(const Array &o) const { return Array(Matrix<T, R, C>(m + o.m)); }
  Array operator-(const Array &o) const { return Array(Matrix<T, R, C>(m - o.m)); }
  Array operator*(const Array &o) const { return Array(m.cwiseProduct(o.m)); }
  Array operator/(const Array &o) const { return Array(m.cwiseQuotient(o.m)); }
  Array operator-() const { return map([](T v) { return -v; }); }
  Array<int, R, C> operator>(const T &s) const { return cmp([s](T v) { return v > s; }); }
  Array<int, R, C> operator<(const T &s) const { return cmp([s](T v) { return v < s; }); }
  Array<int, R, C> operator>=(const T &s) const { return cmp([s](T v) { return v >= s; }); }
  Array<int, R, C> operator<=(const T &s) const { return cmp([s](T v) { return v <= s; }); }
  template <class F> Array<int, R, C> cmp(F f) const {
    Array<int, R, C> b; b.m.resize(rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) b.m.ref(i, j) = f(m.ref(i, j));
    return b;
  }
  // (cond).select(then, else)
  template <class S> Array<S, R, C> select(const Array<S, R, C> &a, const S &e) const {
    Array<S, R, C> o; o.m.resize(rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) o.m.ref(i, j) = m.ref(i, j) ? a.m.ref(i, j) : e;
    return o;
  }
  template <class S> Array<S, R, C> select(const Array<S, R, C> &a, int e) const { return select(a, S(e)); }
  template <class S> Array<S, R, C> select(const Array<S, R, C> &a, const Array<S, R, C> &e) const {
    Array<S, R, C> o; o.m.resize(rows(), cols());
    for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) o.m.ref(i, j) = m.ref(i, j) ? a.m.ref(i, j) : e.m.ref(i, j);
    return o;
  }
};
template <class T, int R, int C> Array<T, R, C> operator+(const T &s, const Array<T, R, C> &a) { return a + s; }
template <class T, int R, int C> Array<T, R, C> operator*(const T &s, const Array<T, R, C> &a) { return a * s; }
template <class T, int R, int C> Array<T, R, C> operator-(const T &s, const Array<T, R, C> &a) { return a.map([s](T v) { return s - v; }); }
template <class T, int R, int C> Array<T, R, C> operator/(const T &s, const Array<T, R, C> &a) { return a.map([s](T v) { return s / v; }); }
template <class D> Array<typename MatrixBase<D>::Scalar, MatrixBase<D>::RowsAtCompileTime, MatrixBase<D>::ColsAtCompileTime> MatrixBase<D>::array() const {
  return Array<Scalar, RowsAtCompileTime, ColsAtCompileTime>(PlainObject(*this));
}
template <class T, int R, int C, int O> template <int AR, int AC> Matrix<T, R, C, O>::Matrix(const Array<T, AR, AC> &a) { assign(a.m); }
template <class T, int R, int C, int O> template <int AR, int AC> Matrix<T, R, C, O> &Matrix<T, R, C, O>::operator=(const Array<T, AR, AC> &a) { assign(a.m); return *this; }
template <class T, int R, int C> template <int AR, int AC> View<T, R, C> &View<T, R, C>::operator=(const Array<T, AR, AC> &a) { store(a.m); return *this; }

// ---------------------------------------------------------------------------------------------------- diagonal
template <class T, int N> class DiagonalMatrix {
  Matrix<T, N, 1> d_;

 public:
  DiagonalMatrix() {}
  explicit DiagonalMatrix(Index n) : d_(n) {}
  template <class O> explicit DiagonalMatrix(const MatrixBase<O> &v) : d_(v) {}
  Matrix<T, N, 1> &diagonal() { return d_; }
  const Matrix<T, N, 1> &diagonal() const { return d_; }
  Index rows() const { return d_.size(); }
  Index cols() const { return d_.size(); }
  void setZero() { d_.setZero(); }
  void setIdentity() { d_.setOnes(); }
  Matrix<T, N, N> toDenseMatrix() const {
    Matrix<T, N, N> m(internal::SizeTag(), rows(), rows()); m.setZero();
    for (Index i = 0; i < rows(); ++i) m.ref(i, i) = d_(i);
    return m;
  }
};
template <class A, class T, int N> typename MatrixBase<A>::PlainObject operator*(const MatrixBase<A> &a, const DiagonalMatrix<T, N> &d) {
  assert(a.cols() == d.rows());
  typename MatrixBase<A>::PlainObject t(internal::SizeTag(), a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) t.ref(i, j) = a.ref(i, j) * d.diagonal()(j);
  return t;
}
template <class A, class T, int N> typename MatrixBase<A>::PlainObject operator*(const DiagonalMatrix<T, N> &d, const MatrixBase<A> &a) {
  assert(a.rows() == d.rows());
  typename MatrixBase<A>::PlainObject t(internal::SizeTag(), a.rows(), a.cols());
  for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) t.ref(i, j) = d.diagonal()(i) * a.ref(i, j);
  return t;
}
template <class D> DiagonalMatrix<typename MatrixBase<D>::Scalar, MatrixBase<D>::SizeAtCompileTime> MatrixBase<D>::asDiagonal() const {
  Matrix<Scalar, SizeAtCompileTime, 1> v(internal::SizeTag(), size(), 1);
  for (Index i = 0; i < size(); ++i) v(i) = vref(i);
  return DiagonalMatrix<Scalar, SizeAtCompileTime>(v);
}

// ---------------------------------------------------------------------------------------------------- LU / LLT / eig
template <class D> typename MatrixBase<D>::PlainObject MatrixBase<D>::inverse() const {
  // Gauss-Jordan elimination with partial pivoting
  const Index n = rows();
  assert(n == cols());
  Matrix<Scalar, Dynamic, Dynamic> a(*this);
  Matrix<Scalar, Dynamic, Dynamic> inv = Matrix<Scalar, Dynamic, Dynamic>::Identity(n, n);
  for (Index c = 0; c < n; ++c) {
    Index p = c; Scalar best = std::abs(a.ref(c, c));
    for (Index r = c + 1; r < n; ++r) if (std::abs(a.ref(r, c)) > best) { best = std::abs(a.ref(r, c)); p = r; }
    if (p != c) for (Index j = 0; j < n; ++j) { std::swap(a.ref(p, j), a.ref(c, j)); std::swap(inv.ref(p, j), inv.ref(c, j)); }
    const Scalar piv = a.ref(c, c);
    for (Index j = 0; j < n; ++j) { a.ref(c, j) /= piv; inv.ref(c, j) /= piv; }
    for (Index r = 0; r < n; ++r) {
      if (r == c) continue;
      const Scalar f = a.ref(r, c);
      if (f == Scalar(0)) continue;
      for (Index j = 0; j < n; ++j) { a.ref(r, j) -= f * a.ref(c, j); inv.ref(r, j) -= f * inv.ref(c, j); }
    }
  }
  return PlainObject(inv);
}
template <class D> typename MatrixBase<D>::Scalar MatrixBase<D>::determinant() const {
  const Index n = rows();
  Matrix<Scalar, Dynamic, Dynamic> a(*this);
  Scalar det(1);
  for (Index c = 0; c < n; ++c) {
    Index p = c; Scalar best = std::abs(a.ref(c, c));
    for (Index r = c + 1; r < n; ++r) if (std::abs(a.ref(r, c)) > best) { best = std::abs(a.ref(r, c)); p = r; }
    if (best == Scalar(0)) return Scalar(0);
    if (p != c) { for (Index j = 0; j < n; ++j) std::swap(a.ref(p, j), a.ref(c, j)); det = -det; }
    det *= a.ref(c, c);
    for (Index r = c + 1; r < n; ++r) { const Scalar f = a.ref(r, c) / a.ref(c, c); for (Index j = c; j < n; ++j) a.ref(r, j) -= f * a.ref(c, j); }
  }
  return det;
}

enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };

template <class M, int UpLo = 1> class LLT {
  M l_;
  ComputationInfo info_ = Success;

 public:
  typedef typename M::Scalar Scalar;
  LLT() {}
  template <class O> explicit LLT(const MatrixBase<O> &a) { compute(a); }
  template <class O> LLT &compute(const MatrixBase<O> &a) {
    const Index n = a.rows();
    l_ = M(a); l_.setZero();
    for (Index j = 0; j < n; ++j) {
      Scalar s = a.ref(j, j);
      for (Index k = 0; k < j; ++k) s -= l_.ref(j, k) * l_.ref(j, k);
      if (!(s > Scalar(0))) info_ = NumericalIssue;
      const Scalar d = std::sqrt(s);
      l_.ref(j, j) = d;
      for (Index i = j + 1; i < n; ++i) {
        Scalar t = a.ref(i, j);
        for (Index k = 0; k < j; ++k) t -= l_.ref(i, k) * l_.ref(j, k);
        l_.ref(i, j) = t / d;
      }
    }
    return *this;
  }
  const M &matrixL() const { return l_; }
  M matrixU() const { return M(l_.transpose()); }
  ComputationInfo info() const { return info_; }
  template <class O> typename MatrixBase<O>::PlainObject solve(const MatrixBase<O> &b) const {
    typename MatrixBase<O>::PlainObject x(b);
    const Index n = l_.rows();
    for (Index c = 0; c < x.cols(); ++c) {
      for (Index i = 0; i < n; ++i) { Scalar s = x.ref(i, c); for (Index k = 0; k < i; ++k) s -= l_.ref(i, k) * x.ref(k, c); x.ref(i, c) = s / l_.ref(i, i); }
      for (Index i = n - 1; i >= 0; --i) { Scalar s = x.ref(i, c); for (Index k = i + 1; k < n; ++k) s -= l_.ref(k, i) * x.ref(k, c); x.ref(i, c) = s / l_.ref(i, i); }
    }
    return x;
  }
};

// Symmetric eigen-decomposition: Householder tridiagonalisation + implicit QL (the classical tred2/tqli scheme),
// eigenvalues ascending, eigenvectors in the columns.
template <class M> class SelfAdjointEigenSolver {
  typedef typename M::Scalar Scalar;
  Matrix<Scalar, M::RowsAtCompileTime, 1> w_;
  M v_;
  ComputationInfo info_ = Success;

 public:
  SelfAdjointEigenSolver() {}
  template <class O> explicit SelfAdjointEigenSolver(const MatrixBase<O> &a) { compute(a); }
  template <class O> SelfAdjointEigenSolver &compute(const MatrixBase<O> &A) {
    const Index n = A.rows();
    std::vector<Scalar> z((size_t)(n * n)), d((size_t)n), e((size_t)n);
    auto a = [&](Index i, Index j) -> Scalar & { return z[(size_t)(i * n + j)]; };
    for (Index i = 0; i < n; ++i) for (Index j = 0; j < n; ++j) a(i, j) = (j <= i) ? A.ref(i, j) : A.ref(j, i);  // lower triangle is referenced
    // tred2
    for (Index i = n - 1; i > 0; --i) {
      Index l = i - 1; Scalar h(0), scale(0);
      if (l > 0) {
        for (Index k = 0; k <= l; ++k) scale += std::abs(a(i, k));
        if (scale == Scalar(0)) e[i] = a(i, l);
        else {
          for (Index k = 0; k <= l; ++k) { a(i, k) /= scale; h += a(i, k) * a(i, k); }
          Scalar f = a(i, l);
          Scalar g = (f >= Scalar(0) ? -std::sqrt(h) : std::sqrt(h));
          e[i] = scale * g; h -= f * g; a(i, l) = f - g; f = Scalar(0);
          for (Index j = 0; j <= l; ++j) {
            a(j, i) = a(i, j) / h; g = Scalar(0);
            for (Index k = 0; k <= j; ++k) g += a(j, k) * a(i, k);
            for (Index k = j + 1; k <= l; ++k) g += a(k, j) * a(i, k);
            e[j] = g / h; f += e[j] * a(i, j);
          }
          const Scalar hh = f / (h + h);
          for (Index j = 0; j <= l; ++j) {
            f = a(i, j); e[j] = g = e[j] - hh * f;
            for (Index k = 0; k <= j; ++k) a(j, k) -= (f * e[k] + g * a(i, k));
          }
        }
      } else e[i] = a(i, l);
      d[i] = h;
    }
    if (n > 0) { d[0] = Scalar(0); e[0] = Scalar(0); }
    for (Index i = 0; i < n; ++i) {
      Index l = i - 1;
      if (d[i] != Scalar(0)) {
        for (Index j = 0; j <= l; ++j) {
          Scalar g(0);
          for (Index k = 0; k <= l; ++k) g += a(i, k) * a(k, j);
          for (Index k = 0; k <= l; ++k) a(k, j) -= g * a(k, i);
        }
      }
      d[i] = a(i, i); a(i, i) = Scalar(1);
      for (Index j = 0; j <= l; ++j) a(j, i) = a(i, j) = Scalar(0);
    }
    // tqli
    for (Index i = 1; i < n; ++i) e[i - 1] = e[i];
    if (n > 0) e[n - 1] = Scalar(0);
    for (Index l = 0; l < n; ++l) {
      int iter = 0; Index m;
      do {
        for (m = l; m < n - 1; ++m) {
          const Scalar dd = std::abs(d[m]) + std::abs(d[m + 1]);
          if (std::abs(e[m]) <= std::numeric_limits<Scalar>::epsilon() * dd) break;
        }
        if (m != l) {
          if (iter++ == 120) { info_ = NoConvergence; break; }
          Scalar g = (d[l + 1] - d[l]) / (Scalar(2) * e[l]);
          Scalar r = std::hypot(g, Scalar(1));
          g = d[m] - d[l] + e[l] / (g + (g >= Scalar(0) ? std::abs(r) : -std::abs(r)));
          Scalar s(1), c(1), p(0);
          Index i;
          for (i = m - 1; i >= l; --i) {
            Scalar f = s * e[i], b = c * e[i];
            e[i + 1] = (r = std::hypot(f, g));
            if (r == Scalar(0)) { d[i + 1] -= p; e[m] = Scalar(0); break; }
            s = f / r; c = g / r; g = d[i + 1] - p;
            r = (d[i] - g) * s + Scalar(2) * c * b;
            d[i + 1] = g + (p = s * r); g = c * r - b;
            for (Index k = 0; k < n; ++k) { f = a(k, i + 1); a(k, i + 1) = s * a(k, i) + c * f; a(k, i) = c * a(k, i) - s * f; }
          }
          if (r == Scalar(0) && i >= l) continue;
          d[l] -= p; e[l] = g; e[m] = Scalar(0);
        }
      } while (m != l);
    }
    std::vector<Index> order((size_t)n);
    for (Index i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](Index p, Index q) { return d[(size_t)p] < d[(size_t)q]; });
    w_.resize(n); v_.resize(n, n);
    for (Index j = 0; j < n; ++j) {
      w_(j) = d[(size_t)order[(size_t)j]];
      for (Index i = 0; i < n; ++i) v_.ref(i, j) = a(i, order[(size_t)j]);
    }
    return *this;
  }
  const Matrix<Scalar, M::RowsAtCompileTime, 1> &eigenvalues() const { return w_; }
  const M &eigenvectors() const { return v_; }
  ComputationInfo info() const { return info_; }
};

enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };
// Singular values (descending) and right singular vectors by one-sided Jacobi (Hestenes) on the columns of A.
template <class M> class JacobiSVD {
  typedef typename M::Scalar Scalar;
  Matrix<Scalar, Dynamic, Dynamic> v_;
  Matrix<Scalar, Dynamic, 1> s_;

 public:
  JacobiSVD() {}
  template <class O> explicit JacobiSVD(const MatrixBase<O> &a, unsigned = 0) { compute(a); }
  template <class O> JacobiSVD &compute(const MatrixBase<O> &A, unsigned = 0) {
    const Index m = A.rows(), n = A.cols();
    Matrix<Scalar, Dynamic, Dynamic> U(A), V = Matrix<Scalar, Dynamic, Dynamic>::Identity(n, n);
    for (int sweep = 0; sweep < 60; ++sweep) {
      Scalar off(0);
      for (Index p = 0; p < n - 1; ++p)
        for (Index q = p + 1; q < n; ++q) {
          Scalar al(0), be(0), ga(0);
          for (Index i = 0; i < m; ++i) { al += U.ref(i, p) * U.ref(i, p); be += U.ref(i, q) * U.ref(i, q); ga += U.ref(i, p) * U.ref(i, q); }
          if (ga == Scalar(0)) continue;
          off = std::max(off, std::abs(ga) / std::sqrt(al * be + std::numeric_limits<Scalar>::min()));
          const Scalar zeta = (be - al) / (Scalar(2) * ga);
          const Scalar t = (zeta >= Scalar(0) ? Scalar(1) : Scalar(-1)) / (std::abs(zeta) + std::sqrt(Scalar(1) + zeta * zeta));
          const Scalar c = Scalar(1) / std::sqrt(Scalar(1) + t * t), sn = c * t;
          for (Index i = 0; i < m; ++i) { const Scalar up = U.ref(i, p), uq = U.ref(i, q); U.ref(i, p) = c * up - sn * uq; U.ref(i, q) = sn * up + c * uq; }
          for (Index i = 0; i < n; ++i) { const Scalar vp = V.ref(i, p), vq = V.ref(i, q); V.ref(i, p) = c * vp - sn * vq; V.ref(i, q) = sn * vp + c * vq; }
        }
      if (off < Scalar(1e-15)) break;
    }
    std::vector<Scalar> sv((size_t)n);
    std::vector<Index> order((size_t)n);
    for (Index j = 0; j < n; ++j) { Scalar s2(0); for (Index i = 0; i < m; ++i) s2 += U.ref(i, j) * U.ref(i, j); sv[(size_t)j] = std::sqrt(s2); order[(size_t)j] = j; }
    std::sort(order.begin(), order.end(), [&](Index a, Index b) { return sv[(size_t)a] > sv[(size_t)b]; });
    v_.resize(n, n); s_.resize(n);
    for (Index j = 0; j < n; ++j) { s_(j) = sv[(size_t)order[(size_t)j]]; for (Index i = 0; i < n; ++i) v_.ref(i, j) = V.ref(i, order[(size_t)j]); }
    return *this;
  }
  const Matrix<Scalar, Dynamic, Dynamic> &matrixV() const { return v_; }
  const Matrix<Scalar, Dynamic, 1> &singularValues() const { return s_; }
};
template <class D> template <class Dummy> auto MatrixBase<D>::jacobiSvd(unsigned flags) const { return JacobiSVD<PlainObject>(*this, flags); }

// ---------------------------------------------------------------------------------------------------- Quaternion
namespace internal {
template <class T, int O> struct traits<Quaternion<T, O>> { typedef T Scalar; };
template <class T, int O, int MO, class S> struct traits<Map<Quaternion<T, O>, MO, S>> { typedef T Scalar; };
template <class T, int O, int MO, class S> struct traits<Map<const Quaternion<T, O>, MO, S>> { typedef T Scalar; };
}  // namespace internal

template <class Derived> class QuaternionBase {
 public:
  typedef typename internal::traits<Derived>::Scalar Scalar;
  typedef Matrix<Scalar, 3, 1> Vector3;
  typedef Matrix<Scalar, 3, 3> Matrix3;
  Derived &derived() { return *static_cast<Derived *>(this); }
  const Derived &derived() const { return *static_cast<const Derived *>(this); }
  Scalar *c() const { return derived().qptr(); }   // x y z w
  Scalar &x() { return c()[0]; }
  Scalar &y() { return c()[1]; }
  Scalar &z() { return c()[2]; }
  Scalar &w() { return c()[3]; }
  const Scalar &x() const { return c()[0]; }
  const Scalar &y() const { return c()[1]; }
  const Scalar &z() const { return c()[2]; }
  const Scalar &w() const { return c()[3]; }
  View<Scalar, 3, 1> vec() const { return View<Scalar, 3, 1>(c(), 3, 1, 1, 3); }
  View<Scalar, 4, 1> coeffs() const { return View<Scalar, 4, 1>(c(), 4, 1, 1, 4); }
  template <class O> Derived &operator=(const QuaternionBase<O> &o) { Scalar t[4] = {o.x(), o.y(), o.z(), o.w()}; for (int i = 0; i < 4; ++i) c()[i] = t[i]; return derived(); }
  template <class O> Derived &operator=(const MatrixBase<O> &m) { set_from_rotation(m); return derived(); }
  Derived &setIdentity() { x() = y() = z() = Scalar(0); w() = Scalar(1); return derived(); }
  Scalar squaredNorm() const { return x() * x() + y() * y() + z() * z() + w() * w(); }
  Scalar norm() const { return std::sqrt(squaredNorm()); }
  void normalize() { const Scalar n = norm(); for (int i = 0; i < 4; ++i) c()[i] /= n; }
  Quaternion<Scalar> normalized() const { const Scalar n = norm(); return Quaternion<Scalar>(w() / n, x() / n, y() / n, z() / n); }
  Quaternion<Scalar> conjugate() const { return Quaternion<Scalar>(w(), -x(), -y(), -z()); }
  Quaternion<Scalar> inverse() const {
    const Scalar n2 = squaredNorm();
    if (n2 > Scalar(0)) return Quaternion<Scalar>(w() / n2, -x() / n2, -y() / n2, -z() / n2);
    return Quaternion<Scalar>(Scalar(0), Scalar(0), Scalar(0), Scalar(0));
  }
  template <class O> Scalar dot(const QuaternionBase<O> &o) const { return x() * o.x() + y() * o.y() + z() * o.z() + w() * o.w(); }
  template <class O> Scalar angularDistance(const QuaternionBase<O> &o) const {
    Quaternion<Scalar> d = (*this) * o.conjugate();
    return Scalar(2) * std::atan2(d.vec().norm(), std::abs(d.w()));
  }
  template <class O> Quaternion<Scalar> operator*(const QuaternionBase<O> &b) const {
    const QuaternionBase &a = *this;
    return Quaternion<Scalar>(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                              a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                              a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                              a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
  }
  template <class O> Derived &operator*=(const QuaternionBase<O> &b) { Quaternion<Scalar> t = (*this) * b; return (*this) = t; }
  template <class O> Vector3 operator*(const MatrixBase<O> &v) const { return _transformVector(v); }
  template <class O> Vector3 _transformVector(const MatrixBase<O> &v) const {
    Vector3 vv(v);
    Vector3 uv = vec().cross(vv);
    uv += uv;
    return vv + w() * uv + vec().cross(uv);
  }
  Matrix3 toRotationMatrix() const {
    Matrix3 r;
    const Scalar tx = Scalar(2) * x(), ty = Scalar(2) * y(), tz = Scalar(2) * z();
    const Scalar twx = tx * w(), twy = ty * w(), twz = tz * w();
    const Scalar txx = tx * x(), txy = ty * x(), txz = tz * x();
    const Scalar tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
    r(0, 0) = Scalar(1) - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = Scalar(1) - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = Scalar(1) - (txx + tyy);
    return r;
  }
  Matrix3 matrix() const { return toRotationMatrix(); }
  template <class O> void set_from_rotation(const MatrixBase<O> &m) {
    if (m.rows() == 4 && m.cols() == 1) { for (int i = 0; i < 4; ++i) c()[i] = m.ref(i, 0); return; }
    assert(m.rows() == 3 && m.cols() == 3);
    Scalar t = m.ref(0, 0) + m.ref(1, 1) + m.ref(2, 2);
    if (t > Scalar(0)) {
      t = std::sqrt(t + Scalar(1));
      w() = Scalar(0.5) * t; t = Scalar(0.5) / t;
      x() = (m.ref(2, 1) - m.ref(1, 2)) * t; y() = (m.ref(0, 2) - m.ref(2, 0)) * t; z() = (m.ref(1, 0) - m.ref(0, 1)) * t;
    } else {
      int i = 0;
      if (m.ref(1, 1) > m.ref(0, 0)) i = 1;
      if (m.ref(2, 2) > m.ref(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m.ref(i, i) - m.ref(j, j) - m.ref(k, k) + Scalar(1));
      c()[i] = Scalar(0.5) * t; t = Scalar(0.5) / t;
      w() = (m.ref(k, j) - m.ref(j, k)) * t;
      c()[j] = (m.ref(j, i) + m.ref(i, j)) * t;
      c()[k] = (m.ref(k, i) + m.ref(i, k)) * t;
    }
  }
  template <class O> Quaternion<Scalar> slerp(const Scalar &t, const QuaternionBase<O> &o) const {
    Scalar d = dot(o), ad = std::abs(d), s0, s1;
    if (ad >= Scalar(1) - std::numeric_limits<Scalar>::epsilon()) { s0 = Scalar(1) - t; s1 = t; }
    else { const Scalar th = std::acos(ad), st = std::sin(th); s0 = std::sin((Scalar(1) - t) * th) / st; s1 = std::sin(t * th) / st; }
    if (d < Scalar(0)) s1 = -s1;
    return Quaternion<Scalar>(s0 * w() + s1 * o.w(), s0 * x() + s1 * o.x(), s0 * y() + s1 * o.y(), s0 * z() + s1 * o.z());
  }
};

template <class T, int Opt> class Quaternion : public QuaternionBase<Quaternion<T, Opt>> {
  T q_[4] = {T(0), T(0), T(0), T(0)};

 public:
  typedef T Scalar;
  T *qptr() const { return const_cast<T *>(q_); }
  Quaternion() {}
  Quaternion(const T &w, const T &x, const T &y, const T &z) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
  explicit Quaternion(const T *p) { for (int i = 0; i < 4; ++i) q_[i] = p[i]; }
  Quaternion(const Quaternion &) = default;
  Quaternion &operator=(const Quaternion &) = default;
  template <class O> Quaternion(const QuaternionBase<O> &o) { q_[0] = o.x(); q_[1] = o.y(); q_[2] = o.z(); q_[3] = o.w(); }
  template <class O> explicit Quaternion(const MatrixBase<O> &m) { this->set_from_rotation(m); }
  template <class O> Quaternion &operator=(const QuaternionBase<O> &o) { QuaternionBase<Quaternion>::operator=(o); return *this; }
  template <class O> Quaternion &operator=(const MatrixBase<O> &m) { this->set_from_rotation(m); return *this; }
  static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
  template <class A, class B> static Quaternion FromTwoVectors(const MatrixBase<A> &a, const MatrixBase<B> &b) {
    Matrix<T, 3, 1> v0 = a.normalized(), v1 = b.normalized();
    T c = v1.dot(v0);
    if (c < T(-1) + std::numeric_limits<T>::epsilon()) {  // opposite vectors: any orthogonal axis
      Matrix<T, 3, 1> axis = v0.cross(Matrix<T, 3, 1>(T(1), T(0), T(0)));
      if (axis.norm() < T(1e-6)) axis = v0.cross(Matrix<T, 3, 1>(T(0), T(1), T(0)));
      axis.normalize();
      return Quaternion(T(0), axis(0), axis(1), axis(2));
    }
    Matrix<T, 3, 1> axis = v0.cross(v1);
    const T s = std::sqrt((T(1) + c) * T(2)), invs = T(1) / s;
    return Quaternion(s * T(0.5), axis(0) * invs, axis(1) * invs, axis(2) * invs);
  }
  template <class NewT> Quaternion<NewT> cast() const { return Quaternion<NewT>(NewT(this->w()), NewT(this->x()), NewT(this->y()), NewT(this->z())); }
};
template <class T, int O, int MO, class S> class Map<Quaternion<T, O>, MO, S> : public QuaternionBase<Map<Quaternion<T, O>, MO, S>> {
  T *p_;

 public:
  typedef T Scalar;
  T *qptr() const { return p_; }
  explicit Map(T *p) : p_(p) {}
  Map(const Map &) = default;
  Map &operator=(const Map &o) { for (int i = 0; i < 4; ++i) p_[i] = o.p_[i]; return *this; }
  template <class OO> Map &operator=(const QuaternionBase<OO> &o) { QuaternionBase<Map>::operator=(o); return *this; }
};
template <class T, int O, int MO, class S> class Map<const Quaternion<T, O>, MO, S> : public QuaternionBase<Map<const Quaternion<T, O>, MO, S>> {
  const T *p_;

 public:
  typedef T Scalar;
  T *qptr() const { return const_cast<T *>(p_); }
  explicit Map(const T *p) : p_(p) {}
};
template <class D> std::ostream &operator<<(std::ostream &os, const QuaternionBase<D> &q) { return os << q.x() << "i + " << q.y() << "j + " << q.z() << "k + " << q.w(); }

template <class T> class AngleAxis {
  T angle_; Matrix<T, 3, 1> axis_;

 public:
  template <class O> AngleAxis(const T &a, const MatrixBase<O> &ax) : angle_(a), axis_(ax) {}
  Matrix<T, 3, 3> toRotationMatrix() const {
    const T c = std::cos(angle_), s = std::sin(angle_);
    Matrix<T, 3, 1> sa = axis_ * s, ca = axis_ * (T(1) - c);
    Matrix<T, 3, 3> r;
    T tmp = ca.x() * axis_.y(); r(0, 1) = tmp - sa.z(); r(1, 0) = tmp + sa.z();
    tmp = ca.x() * axis_.z(); r(0, 2) = tmp + sa.y(); r(2, 0) = tmp - sa.y();
    tmp = ca.y() * axis_.z(); r(1, 2) = tmp - sa.x(); r(2, 1) = tmp + sa.x();
    r(0, 0) = ca.x() * axis_.x() + c; r(1, 1) = ca.y() * axis_.y() + c; r(2, 2) = ca.z() * axis_.z() + c;
    return r;
  }
};
typedef AngleAxis<double> AngleAxisd;

template <class T> struct Triplet {
  int r_, c_; T v_;
  Triplet(int r = 0, int c = 0, const T &v = T()) : r_(r), c_(c), v_(v) {}
  int row() const { return r_; }
  int col() const { return c_; }
  const T &value() const { return v_; }
};

// ---------------------------------------------------------------------------------------------------- typedefs
#define ME_TYPEDEFS(T, S)                                 \
  typedef Matrix<T, 2, 2> Matrix2##S;                     \
  typedef Matrix<T, 3, 3> Matrix3##S;                     \
  typedef Matrix<T, 4, 4> Matrix4##S;                     \
  typedef Matrix<T, Dynamic, Dynamic> MatrixX##S;         \
  typedef Matrix<T, 2, 1> Vector2##S;                     \
  typedef Matrix<T, 3, 1> Vector3##S;                     \
  typedef Matrix<T, 4, 1> Vector4##S;                     \
  typedef Matrix<T, Dynamic, 1> VectorX##S;               \
  typedef Matrix<T, 1, 2> RowVector2##S;                  \
  typedef Matrix<T, 1, 3> RowVector3##S;                  \
  typedef Matrix<T, 1, 4> RowVector4##S;                  \
  typedef Matrix<T, 1, Dynamic> RowVectorX##S;
ME_TYPEDEFS(double, d)
ME_TYPEDEFS(float, f)
ME_TYPEDEFS(int, i)
#undef ME_TYPEDEFS
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;
template <class T> class aligned_allocator : public std::allocator<T> {
 public:
  template <class U> struct rebind { typedef aligned_allocator<U> other; };
  aligned_allocator() {}
  template <class U> aligned_allocator(const aligned_allocator<U> &) {}
};

}  // namespace Eigen
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
