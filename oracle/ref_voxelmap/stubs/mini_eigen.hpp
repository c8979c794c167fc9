// ORACLE / TEST INFRASTRUCTURE ONLY -- an Eigen-SHAPED stub: just enough of Eigen's interface (fixed-size dense matrices, eager evaluation) for the
// reference's own source files to compile from where they lie (oracle/Makefile: _ref/libref_voxelmap.so, _ref/libref_meshgeom.so).  It pins the
// reference's LOGIC -- which operands, which formulas, which thresholds, which control flow -- not Eigen's arithmetic: products are plain
// k-ascending sums, EigenSolver / SelfAdjointEigenSolver of a symmetric 3 x 3 are the oracle's cyclic Jacobi (oracle/orc_linalg.hpp).  Eigen itself
// is not in this image; results are compared with the oracle's to rounding, discrete outcomes exactly.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstddef>
#include <vector>
#include <memory>
#include <cstdlib>
#include <ostream>
#include <type_traits>
#include "../../orc_linalg.hpp"

namespace Eigen {
typedef std::ptrdiff_t Index;
template <typename T> struct aligned_allocator : std::allocator<T> { template <typename U> struct rebind { typedef aligned_allocator<U> other; }; };

template <typename T, int R, int C> struct Matrix;
template <typename T> struct real_of { typedef T type; };
template <typename T> struct real_of<std::complex<T>> { typedef T type; };

template <typename T, int R, int C> struct CommaInit {
    Matrix<T, R, C>& m; int k;
    CommaInit(Matrix<T, R, C>& mm, T v) : m(mm), k(0) { m.a[k++] = v; }
    CommaInit& operator,(T v) { if (k < R * C) m.a[k++] = v; return *this; }
};
template <typename T, int R, int C, int BR, int BC> struct BlockRef {   // writable view
    Matrix<T, R, C>& m; int r0, c0;
    Matrix<T, BR, BC> eval() const { Matrix<T, BR, BC> o; for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) o(i, j) = m(r0 + i, c0 + j); return o; }
    template <int SR, int SC> BlockRef& operator=(const Matrix<T, SR, SC>& s) {
        static_assert(SR * SC == BR * BC, "block assignment: size mismatch");
        // same shape, or a vector assigned to a vector-shaped block of the other orientation (Eigen transposes vectors on assignment)
        for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) m(r0 + i, c0 + j) = (SR == BR) ? s.a[i * SC + j] : s.a[j * SC + i];
        return *this;
    }
    // a dynamic-size right-hand side (MatrixXd below): anything that says so with a `dyn_tag`
    template <typename D, typename = typename D::dyn_tag> BlockRef& operator=(const D& s) {
        if (s.rows() != BR || s.cols() != BC) std::abort();
        for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) m(r0 + i, c0 + j) = s(i, j);
        return *this;
    }
    operator Matrix<T, BR, BC>() const { return eval(); }
    template <int K> Matrix<T, BR, K> operator*(const Matrix<T, BC, K>& b) const { return eval() * b; }
    template <int R2, int C2, int K> Matrix<T, BR, K> operator*(const BlockRef<T, R2, C2, BC, K>& b) const { return eval() * b.eval(); }
    template <typename D, typename = typename D::dynvec_tag> Matrix<T, BR, 1> operator*(const D& v) const {   // fixed block * VectorXd
        if (v.size() != BC) std::abort();
        Matrix<T, BR, 1> o;
        for (int i = 0; i < BR; i++) { T s = T(0); for (int k = 0; k < BC; k++) s += m(r0 + i, c0 + k) * v(k); o.a[i] = s; }
        return o;
    }
    typename real_of<T>::type norm() const { return eval().norm(); }
    void setZero() { for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) m(r0 + i, c0 + j) = T(0); }
    // `.block< 3, 3 >( r, c ).diagonal() = v` (IMU_Processing.cpp:536-537, 857-860): a writable view of the block's diagonal
    struct DiagRef { Matrix<T, R, C>& m; int r0, c0; template <int SR, int SC> DiagRef& operator=(const Matrix<T, SR, SC>& s) { static_assert(SR * SC == (BR < BC ? BR : BC), "diagonal assignment: size mismatch"); for (int i = 0; i < SR * SC; i++) m(r0 + i, c0 + i) = s.a[i]; return *this; } };
    DiagRef diagonal() { return DiagRef{m, r0, c0}; }
    // (only inside the reference's `if ( 0 )` degeneracy print, voxel_mapping.cpp:1601-1607: has to compile, never runs)
    Matrix<std::complex<T>, BR, 1> eigenvalues() const { std::abort(); }
};
template <typename T, int R, int C> struct Rowwise { const Matrix<T, R, C>& m; Matrix<T, R, 1> sum() const { Matrix<T, R, 1> o; for (int i = 0; i < R; i++) { T s = T(0); for (int j = 0; j < C; j++) s += m(i, j); o.a[i] = s; } return o; } };

template <typename T, int R, int C> struct Matrix {
    typedef Eigen::Index Index;
    typedef T Scalar;
    T a[R * C];   // row-major
    Matrix() { for (int i = 0; i < R * C; i++) a[i] = T(0); }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "3-vector constructor"); a[0] = x; a[1] = y; a[2] = z; }
    Matrix(T x, T y) { static_assert(R * C == 2, "2-vector constructor"); a[0] = x; a[1] = y; }
    Matrix(T x, T y, T z, T w) { static_assert(R * C == 4, "4-vector constructor"); a[0] = x; a[1] = y; a[2] = z; a[3] = w; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Identity() { Matrix m; for (int i = 0; i < R && i < C; i++) m(i, i) = T(1); return m; }
    T& operator()(int i, int j) { return a[i * C + j]; }
    const T& operator()(int i, int j) const { return a[i * C + j]; }
    T& operator()(int i) { static_assert(R == 1 || C == 1, "vector access"); return a[i]; }
    const T& operator()(int i) const { static_assert(R == 1 || C == 1, "vector access"); return a[i]; }
    T& operator[](int i) { return a[i]; }
    const T& operator[](int i) const { return a[i]; }
    int rows() const { return R; } int cols() const { return C; } int size() const { return R * C; }
    operator T() const { static_assert(R == 1 && C == 1, "only a 1 x 1 matrix converts to its scalar"); return a[0]; }
    CommaInit<T, R, C> operator<<(T v) { return CommaInit<T, R, C>(*this, v); }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o(j, i) = (*this)(i, j); return o; }
    Matrix operator-() const { Matrix o; for (int i = 0; i < R * C; i++) o.a[i] = -a[i]; return o; }
    Matrix operator+(const Matrix& b) const { Matrix o; for (int i = 0; i < R * C; i++) o.a[i] = a[i] + b.a[i]; return o; }
    Matrix operator-(const Matrix& b) const { Matrix o; for (int i = 0; i < R * C; i++) o.a[i] = a[i] - b.a[i]; return o; }
    Matrix& operator+=(const Matrix& b) { for (int i = 0; i < R * C; i++) a[i] += b.a[i]; return *this; }
    Matrix& operator-=(const Matrix& b) { for (int i = 0; i < R * C; i++) a[i] -= b.a[i]; return *this; }
    Matrix& operator*=(T s) { for (int i = 0; i < R * C; i++) a[i] *= s; return *this; }
    Matrix& operator/=(T s) { for (int i = 0; i < R * C; i++) a[i] /= s; return *this; }
    Matrix operator*(T s) const { Matrix o; for (int i = 0; i < R * C; i++) o.a[i] = a[i] * s; return o; }
    Matrix operator/(T s) const { Matrix o; for (int i = 0; i < R * C; i++) o.a[i] = a[i] / s; return o; }
    template <int K> Matrix<T, R, K> operator*(const Matrix<T, C, K>& b) const {
        Matrix<T, R, K> o;
        for (int i = 0; i < R; i++) for (int j = 0; j < K; j++) { T s = T(0); for (int k = 0; k < C; k++) s += (*this)(i, k) * b(k, j); o(i, j) = s; }
        return o;
    }
    template <int R2, int C2, int K> Matrix<T, R, K> operator*(const BlockRef<T, R2, C2, C, K>& b) const { return (*this) * b.eval(); }
    template <int R2, int C2> Matrix operator+(const BlockRef<T, R2, C2, R, C>& b) const { return (*this) + b.eval(); }
    template <int R2, int C2> Matrix operator-(const BlockRef<T, R2, C2, R, C>& b) const { return (*this) - b.eval(); }
    Matrix<T, R, 1> col(Index j) const { Matrix<T, R, 1> o; for (int i = 0; i < R; i++) o.a[i] = (*this)(i, (int)j); return o; }
    BlockRef<T, R, C, 1, C> row(Index i) { return BlockRef<T, R, C, 1, C>{*this, (int)i, 0}; }
    Matrix<T, 1, C> row(Index i) const { Matrix<T, 1, C> o; for (int j = 0; j < C; j++) o.a[j] = (*this)((int)i, j); return o; }
    template <int BR, int BC> BlockRef<T, R, C, BR, BC> block(int r0, int c0) { return BlockRef<T, R, C, BR, BC>{*this, r0, c0}; }
    template <int BR, int BC> Matrix<T, BR, BC> block(int r0, int c0) const { Matrix<T, BR, BC> o; for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) o(i, j) = (*this)(r0 + i, c0 + j); return o; }
    template <int N> BlockRef<T, R, C, N, 1> head() { static_assert(C == 1, "head of a column vector"); return BlockRef<T, R, C, N, 1>{*this, 0, 0}; }
    // block( r0, c0, rows, cols ) with run-time extents (ImMesh_mesh_reconstruction.cpp:100, 199): a view that is assigned from / converts to a fixed matrix
    struct DynBlock {
        Matrix& m; int r0, c0, nr, nc;
        template <int SR, int SC> DynBlock& operator=(const Matrix<T, SR, SC>& s) { if (SR != nr || SC != nc) std::abort(); for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) m(r0 + i, c0 + j) = s(i, j); return *this; }
        template <int BR, int BC> operator Matrix<T, BR, BC>() const { if (BR != nr || BC != nc) std::abort(); Matrix<T, BR, BC> o; for (int i = 0; i < nr; i++) for (int j = 0; j < nc; j++) o(i, j) = m(r0 + i, c0 + j); return o; }
    };
    DynBlock block(int r0, int c0, int nr, int nc) { return DynBlock{*this, r0, c0, nr, nc}; }
    // v.asDiagonal() (IMU_Processing.cpp:858): the dense diagonal matrix -- a product with it adds exact zeros to what Eigen's diagonal product computes
    Matrix<T, R * C, R * C> asDiagonal() const { Matrix<T, R * C, R * C> o; for (int i = 0; i < R * C; i++) for (int j = 0; j < R * C; j++) o(i, j) = i == j ? a[i] : T(0); return o; }
    Matrix<T, (R < C ? R : C), 1> diagonal() const { Matrix<T, (R < C ? R : C), 1> o; for (int i = 0; i < R && i < C; i++) o.a[i] = (*this)(i, i); return o; }
    typename real_of<T>::type squaredNorm() const { typename real_of<T>::type s = 0; for (int i = 0; i < R * C; i++) s += std::norm(a[i]); return s; }
    typename real_of<T>::type norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { const T n = (T)norm(); for (int i = 0; i < R * C; i++) a[i] /= n; }
    Matrix normalized() const { Matrix o = *this; o.normalize(); return o; }
    template <int R2, int C2> T dot(const Matrix<T, R2, C2>& b) const { static_assert(R2 * C2 == R * C, "dot: sizes differ"); T s = T(0); for (int i = 0; i < R * C; i++) s += a[i] * b.a[i]; return s; }
    Matrix cross(const Matrix& b) const { static_assert(R * C == 3, "cross"); Matrix o; o.a[0] = a[1] * b.a[2] - a[2] * b.a[1]; o.a[1] = a[2] * b.a[0] - a[0] * b.a[2]; o.a[2] = a[0] * b.a[1] - a[1] * b.a[0]; return o; }
    T sum() const { T s = T(0); for (int i = 0; i < R * C; i++) s += a[i]; return s; }
    Rowwise<T, R, C> rowwise() const { return Rowwise<T, R, C>{*this}; }
    T minCoeff(Index* idx) const { static_assert(R == 1 || C == 1, "vector"); int b = 0; for (int i = 1; i < R * C; i++) if (a[i] < a[b]) b = i; *idx = b; return a[b]; }   // (first minimum, as Eigen's visitor)
    T maxCoeff(Index* idx) const { static_assert(R == 1 || C == 1, "vector"); int b = 0; for (int i = 1; i < R * C; i++) if (a[i] > a[b]) b = i; *idx = b; return a[b]; }
    Matrix<typename real_of<T>::type, R, C> real() const { Matrix<typename real_of<T>::type, R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = std::real(a[i]); return o; }
    template <typename U> Matrix<U, R, C> cast() const { Matrix<U, R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = (U)a[i]; return o; }
    void setZero() { for (int i = 0; i < R * C; i++) a[i] = T(0); }
    void setIdentity() { setZero(); for (int i = 0; i < R && i < C; i++) (*this)(i, i) = T(1); }
    T trace() const { T s = T(0); for (int i = 0; i < R && i < C; i++) s += (*this)(i, i); return s; }
    // Matrix<double, 18, 18>::inverse() (voxel_mapping.cpp:1588): Eigen's LU is not restated -- Gauss-Jordan with partial pivoting (orc_linalg.hpp) stands in
    Matrix inverse() const { static_assert(R == C, "inverse of a square matrix"); Matrix o; if (!orc::inv_gauss_jordan(a, o.a, R)) std::abort(); return o; }
    T* data() { return a; } const T* data() const { return a; }
};
template <typename T, int R, int C> Matrix<T, R, C> operator*(T s, const Matrix<T, R, C>& m) { return m * s; }
// (float * Matrix<double>: the reference writes `range * direction_hat * N` with a float range)
template <int R, int C> Matrix<double, R, C> operator*(float s, const Matrix<double, R, C>& m) { return m * (double)s; }
template <int R, int C> Matrix<double, R, C> operator*(int s, const Matrix<double, R, C>& m) { return m * (double)s; }

typedef Matrix<double, 3, 1> Vector3d;   typedef Matrix<float, 3, 1> Vector3f;   typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 3> Matrix3d;   typedef Matrix<float, 3, 3> Matrix3f;   typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<std::complex<double>, 3, 3> Matrix3cd;   typedef Matrix<std::complex<double>, 3, 1> Vector3cd;

// general eigen-solver of a 3 x 3 matrix: the reference only ever hands it symmetric (covariance) matrices; Eigen's real-Schur algorithm is not
// restated -- the oracle's cyclic Jacobi stands in (columns = eigenvectors, eigenvalue k at position k, zero imaginary parts)
template <typename M> struct EigenSolver;
template <> struct EigenSolver<Matrix3d> {
    Matrix3cd V; Vector3cd L;
    explicit EigenSolver(const Matrix3d& A) {
        double ev[3], vv[9];
        orc::sym3_eigen_jacobi(A.a, ev, vv);
        for (int i = 0; i < 3; i++) { L.a[i] = ev[i]; for (int j = 0; j < 3; j++) V(i, j) = vv[i * 3 + j]; }
    }
    const Matrix3cd& eigenvectors() const { return V; }
    const Vector3cd& eigenvalues() const { return L; }
};
// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute (mesh_rec_geometry.cpp:199-200): eigenvalues ascending, eigenvector k in column k -- the oracle's
// Jacobi + a stable ascending sort, exactly as orc_mesher.hpp stands in for it
template <typename M> struct SelfAdjointEigenSolver;
template <> struct SelfAdjointEigenSolver<Matrix3d> {
    Matrix3d V; Vector3d L;
    SelfAdjointEigenSolver& compute(const Matrix3d& A) {
        double ev[3], vv[9];
        orc::sym3_eigen_jacobi(A.a, ev, vv);
        int ord[3] = {0, 1, 2};
        std::stable_sort(ord, ord + 3, [&](int x, int y) { return ev[x] < ev[y]; });
        for (int k = 0; k < 3; k++) { L.a[k] = ev[ord[k]]; for (int i = 0; i < 3; i++) V(i, k) = vv[i * 3 + ord[k]]; }
        return *this;
    }
    const Matrix3d& eigenvectors() const { return V; }
    const Vector3d& eigenvalues() const { return L; }
};

// the handful of dynamic-size operations the compiled excerpts perform: delaunay_triangulation (mesh_rec_geometry.cpp:174-213) on its n x 3 point
// matrix; lio_state_estimation (voxel_mapping.cpp:1487-1587) on Hsub (M x 6), Hsub_T_R_inv (6 x M), R_inv / meas_vec (M)
inline std::vector<double>& last_dyn_matvec() { static thread_local std::vector<double> v; return v; }   // test tap: the newest MatrixXd * VectorXd (HTz)
struct VectorXd {
    typedef void dynvec_tag;
    std::vector<double> a;
    VectorXd() {}
    explicit VectorXd(int n) : a((size_t)n, 0.0) {}
    int size() const { return (int)a.size(); } int rows() const { return (int)a.size(); } int cols() const { return 1; }
    void setZero() { std::fill(a.begin(), a.end(), 0.0); }
    double& operator()(int i) { return a[(size_t)i]; } double operator()(int i) const { return a[(size_t)i]; }
    double& operator[](int i) { return a[(size_t)i]; } double operator[](int i) const { return a[(size_t)i]; }
};
struct MatrixXd;
struct MatrixXdT { const MatrixXd& m; };
struct MatrixXd {
    typedef void dyn_tag;
    int r = 0, c = 0; std::vector<double> a;
    MatrixXd() {}
    MatrixXd(int rr, int cc) { resize(rr, cc); }
    void resize(int rr, int cc) { r = rr; c = cc; a.assign((size_t)rr * cc, 0.0); }
    int rows() const { return r; } int cols() const { return c; }
    double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    // `row(i) << a, b, ...` / `col(j) << a, b, ...` (voxel_mapping.cpp:1564-1566): values in order, floats widen to double
    struct CommaDyn { MatrixXd& m; int i, j, di, dj; CommaDyn& operator,(double v) { m(i, j) = v; i += di; j += dj; return *this; } };
    struct RowRef {
        MatrixXd& m; int i;
        RowRef& operator=(const Matrix<double, 3, 1>& v) { for (int j = 0; j < 3; j++) m(i, j) = v.a[j]; return *this; }
        double dot(const Matrix<double, 3, 1>& v) const { double s = 0; for (int j = 0; j < 3; j++) s += m(i, j) * v.a[j]; return s; }
        CommaDyn operator<<(double v) { m(i, 0) = v; return CommaDyn{m, i, 1, 0, 1}; }
    };
    struct ColRef { MatrixXd& m; int j; CommaDyn operator<<(double v) { m(0, j) = v; return CommaDyn{m, 1, j, 1, 0}; } };
    RowRef row(int i) { return RowRef{*this, i}; }
    ColRef col(int j) { return ColRef{*this, j}; }
    struct Colwise { const MatrixXd& m; Matrix<double, 1, 3> mean() const { Matrix<double, 1, 3> o; for (int j = 0; j < 3; j++) { double s = 0; for (int i = 0; i < m.r; i++) s += m(i, j); o.a[j] = s / (double)m.r; } return o; } };
    Colwise colwise() const { return Colwise{*this}; }
    struct RowwiseX { const MatrixXd& m; MatrixXd operator-(const Matrix<double, 1, 3>& v) const { MatrixXd o; o.resize(m.r, m.c); for (int i = 0; i < m.r; i++) for (int j = 0; j < m.c; j++) o(i, j) = m(i, j) - v.a[j]; return o; } };
    RowwiseX rowwise() const { return RowwiseX{*this}; }
    MatrixXdT transpose() const { return MatrixXdT{*this}; }
    MatrixXd operator/(double s) const { MatrixXd o = *this; for (double& x : o.a) x /= s; return o; }
    template <int R, int C> operator Matrix<double, R, C>() const { Matrix<double, R, C> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o(i, j) = (*this)(i, j); return o; }
};
inline MatrixXd operator*(const MatrixXdT& at, const MatrixXd& b) {   // (A^T B)(p, q) = sum over the rows, in row order
    MatrixXd o; o.resize(at.m.c, b.c);
    for (int p = 0; p < at.m.c; p++) for (int q = 0; q < b.c; q++) { double s = 0; for (int i = 0; i < b.r; i++) s += at.m(i, p) * b(i, q); o(p, q) = s; }
    return o;
}
inline MatrixXd operator*(const MatrixXd& x, const MatrixXd& y) {     // inner index ascending: for (6 x M)(M x 6) that is the sum over the matches in list order
    if (x.c != y.r) std::abort();
    MatrixXd o; o.resize(x.r, y.c);
    for (int p = 0; p < x.r; p++) for (int q = 0; q < y.c; q++) { double s = 0; for (int k = 0; k < x.c; k++) s += x(p, k) * y(k, q); o(p, q) = s; }
    return o;
}
inline VectorXd operator*(const MatrixXd& x, const VectorXd& v) {
    if (x.c != v.size()) std::abort();
    VectorXd o(x.r);
    for (int p = 0; p < x.r; p++) { double s = 0; for (int k = 0; k < x.c; k++) s += x(p, k) * v(k); o(p) = s; }
    last_dyn_matvec() = o.a;
    return o;
}
template <typename T, int R, int C> std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) { for (int i = 0; i < R; i++) { for (int j = 0; j < C; j++) os << m(i, j) << ' '; if (R > 1) os << '\n'; } return os; }
// Eigen::Quaterniond( rotation matrix ) (voxel_mapping.cpp:1246, ImMesh_mesh_reconstruction.cpp:416): carried along, never read by the compiled excerpts
struct Quaterniond {
    Matrix3d R; double q[4] = {0, 0, 0, 1};   // x, y, z, w
    Quaterniond() {}
    Quaterniond(const Matrix3d& m) : R(m) {   // (the usual trace / largest-diagonal conversion)
        const double t = m(0, 0) + m(1, 1) + m(2, 2);
        if (t > 0) { double s = std::sqrt(t + 1.0); q[3] = 0.5 * s; s = 0.5 / s; q[0] = (m(2, 1) - m(1, 2)) * s; q[1] = (m(0, 2) - m(2, 0)) * s; q[2] = (m(1, 0) - m(0, 1)) * s; }
        else {
            int i = 0; if (m(1, 1) > m(0, 0)) i = 1; if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            double s = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0); q[i] = 0.5 * s; s = 0.5 / s;
            q[3] = (m(k, j) - m(j, k)) * s; q[j] = (m(j, i) + m(i, j)) * s; q[k] = (m(k, i) + m(i, k)) * s;
        }
    }
    Matrix<double, 4, 1> coeffs() const { return Matrix<double, 4, 1>(q[0], q[1], q[2], q[3]); }
};
}  // namespace Eigen
