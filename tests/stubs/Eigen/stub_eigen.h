// TEST SCAFFOLDING — a declaration-level stand-in for the parts of Eigen that the reference-side adapters (adapters/**) and the
// reference headers they include mention.  It exists so that `g++ -fsyntax-only` can type-check the adapters in this
// container, where Eigen is not installed (tests/test_adapters_compile.py).  Nothing here computes anything and nothing in
// the product includes it.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <iosfwd>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen
{
using Index = std::ptrdiff_t;
constexpr int Dynamic = -1;
enum StorageOptions
{
  ColMajor = 0,
  RowMajor = 1
};
enum UpLoType
{
  Lower = 1,
  Upper = 2
};
enum NoChange_t
{
  NoChange
};
template <typename Derived>
struct ArrayOps;

template <typename Scalar, int Rows, int Cols, int Options = ColMajor>
class Matrix
{
public:
  Matrix();
  explicit Matrix(Index n);
  Matrix(Index r, Index c);
  Matrix(Scalar a, Scalar b, Scalar c);
  template <typename Other>
  Matrix(const Other& other);
  template <typename Other>
  Matrix& operator=(const Other& other);
  Index size() const;
  Index rows() const;
  Index cols() const;
  Scalar* data();
  const Scalar* data() const;
  Scalar& operator()(Index i);
  const Scalar& operator()(Index i) const;
  Scalar& operator()(Index i, Index j);
  const Scalar& operator()(Index i, Index j) const;
  Scalar& operator[](Index i);
  const Scalar& operator[](Index i) const;
  Scalar x() const;
  Scalar y() const;
  Scalar z() const;
  void resize(Index n);
  void resize(Index r, Index c);
  void setZero();
  void setZero(Index n);
  void setConstant(Scalar v);
  Scalar sum() const;
  Scalar maxCoeff() const;
  Scalar minCoeff() const;
  Scalar norm() const;
  Matrix transpose() const;
  Matrix cwiseMax(const Matrix& o) const;
  Matrix cwiseMin(const Matrix& o) const;
  static Matrix Constant(Index n, Scalar v);
  static Matrix Zero(Index n);
  static Matrix Zero(Index r, Index c);
  static Matrix Ones(Index n);
  static Matrix Identity();
  Matrix<Scalar, Dynamic, 1> head(Index n) const;
  Matrix<Scalar, Dynamic, 1> tail(Index n) const;
  Matrix<Scalar, Dynamic, 1> segment(Index i, Index n) const;
  Matrix<Scalar, Dynamic, 1> col(Index j) const;
  Matrix<Scalar, 1, Dynamic> row(Index i) const;
  template <int R, int C>
  Matrix<Scalar, R, C> block(Index i, Index j) const;
  ArrayOps<Matrix> array() const;
  Matrix operator*(Scalar s) const;
  Matrix operator+(const Matrix& o) const;
  Matrix operator-(const Matrix& o) const;
  Matrix operator-() const;
};
template <typename Scalar, int Rows, int Cols, int Options>
Matrix<Scalar, Rows, Cols, Options> operator*(Scalar s, const Matrix<Scalar, Rows, Cols, Options>& m);
template <typename Scalar, int Rows, int Cols, int Options>
std::ostream& operator<<(std::ostream& os, const Matrix<Scalar, Rows, Cols, Options>& m);

template <typename Derived>
struct ArrayOps
{
  ArrayOps abs() const;
  ArrayOps operator<(double v) const;
  Derived select(double a, const ArrayOps& b) const;
};

using VectorXd = Matrix<double, Dynamic, 1>;
using VectorXi = Matrix<int, Dynamic, 1>;
using MatrixXd = Matrix<double, Dynamic, Dynamic>;
using MatrixX2d = Matrix<double, Dynamic, 2>;
using Vector3d = Matrix<double, 3, 1>;
using Vector4d = Matrix<double, 4, 1>;
using Matrix3d = Matrix<double, 3, 3>;
using Matrix4d = Matrix<double, 4, 4>;

template <typename T>
class Ref : public T
{
public:
  template <typename Other>
  Ref(const Other& other);
};
template <typename T>
class Ref<const T> : public T
{
public:
  template <typename Other>
  Ref(const Other& other);
};
template <typename T>
class Map : public T
{
public:
  Map(const typename std::remove_const<decltype(*T().data())>::type* p, Index n);
};
template <>
class Map<const VectorXd> : public VectorXd
{
public:
  Map(const double* p, Index n);
};

template <typename Scalar>
class Triplet
{
public:
  Triplet(Index i, Index j, Scalar v);
  Index row() const;
  Index col() const;
  Scalar value() const;
};

template <typename Scalar, int Options = ColMajor, typename StorageIndex = int>
class SparseMatrix
{
public:
  SparseMatrix();
  SparseMatrix(Index r, Index c);
  template <typename Other>
  SparseMatrix(const Other& other);
  template <typename Other>
  SparseMatrix& operator=(const Other& other);
  Index rows() const;
  Index cols() const;
  Index nonZeros() const;
  Index outerSize() const;
  Index innerSize() const;
  void resize(Index r, Index c);
  void conservativeResize(Index r, NoChange_t);
  void conservativeResize(Index r, Index c);
  template <typename Sizes>
  void reserve(const Sizes& sizes);
  Scalar& insert(Index i, Index j);
  Scalar& coeffRef(Index i, Index j);
  Scalar coeff(Index i, Index j) const;
  void makeCompressed();
  bool isCompressed() const;
  template <typename It>
  void setFromTriplets(It begin, It end);
  const Scalar* valuePtr() const;
  const StorageIndex* innerIndexPtr() const;
  const StorageIndex* outerIndexPtr() const;
  struct InnerVectorRef
  {
    Index nonZeros() const;
  };
  InnerVectorRef innerVector(Index k) const;
  template <unsigned Mode>
  SparseMatrix triangularView() const;
  SparseMatrix eval() const;
  SparseMatrix transpose() const;
  class InnerIterator
  {
  public:
    InnerIterator(const SparseMatrix& m, Index outer);
    InnerIterator& operator++();
    explicit operator bool() const;
    Scalar value() const;
    Index row() const;
    Index col() const;
    Index index() const;
  };
};
template <typename Scalar, int Options, typename StorageIndex>
SparseMatrix<Scalar, Options, StorageIndex> operator*(Scalar s, const SparseMatrix<Scalar, Options, StorageIndex>& m);

template <typename Scalar, int Options = ColMajor, typename StorageIndex = int>
class SparseVector
{
public:
  Index size() const;
};

template <typename Scalar, int Dim, int Mode = 0>
class Transform
{
public:
  Transform();
  Matrix<Scalar, 4, 4> matrix() const;
  Matrix<Scalar, 3, 1> translation() const;
  Matrix<Scalar, 3, 3> linear() const;
  Matrix<Scalar, 3, 3> rotation() const;
  Transform inverse() const;
  Transform operator*(const Transform& o) const;
  Matrix<Scalar, 3, 1> operator*(const Matrix<Scalar, 3, 1>& p) const;
  static Transform Identity();
  Scalar operator()(Index i, Index j) const;
};
using Isometry3d = Transform<double, 3, 1>;
}  // namespace Eigen
