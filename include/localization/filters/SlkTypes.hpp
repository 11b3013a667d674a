/**\file SlkTypes.hpp
 * Small dense types used by the GPU-backed facade of localization::Usckf / localization::Msckf
 * when Eigen / MTK are not on the include path.  They follow Eigen's / MTK's conventions (column-major
 * storage, data()/rows()/cols()/operator()(i,j), comma initialiser, quaternion coefficient order x,y,z,w,
 * quaternion * vector = rotation, SO3::exp / log / boxplus / boxminus as in MTK::SO3), so that the model
 * functions of the reference's own tests (test/MsckfUnitTest.cpp:33-47, test/UsckfUnitTest.cpp:34-86)
 * compile against them unchanged and a caller compiled against Eigen can pass Eigen::Matrix /
 * Eigen::Quaternion objects to the same templated entry points (only data(), rows(), cols(), size() are
 * used on matrix arguments).
 *
 * The manifold arithmetic here is the HOST side of the interface (State::set / boxplus / boxminus /
 * getVectorizedState, the wrapper operators, model functors of the Tier-B path); the filters' own
 * boxplus / boxminus work runs on the GPU (csrc/slk_math.hpp) with the same definitions.
 */
#ifndef _SLK_TYPES_HPP_
#define _SLK_TYPES_HPP_

#include <cassert>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <type_traits>
#include <utility>
#include <vector>

namespace localization
{
namespace slk
{
    /** after one element of MTK's text form: blanks and an optional comma (never touches a stream that is at its end) */
    inline void skip_separator(std::istream &is)
    {
        if (!is.good()) return;
        is >> std::ws;
        if (is.good() && is.peek() == ',') is.ignore(1);
    }

    /** `m << a, b, c;` -- Eigen's comma initialiser, filling in storage order of a vector / row-major order of a matrix. */
    template <class T>
    class CommaInit
    {
        T &t_;
        int i_;
    public:
        CommaInit(T &t, double first) : t_(t), i_(0) { t_.comma_set(i_++, first); }
        CommaInit &operator,(double v) { t_.comma_set(i_++, v); return *this; }
    };

    /** Column-major dynamic matrix (Eigen::MatrixXd stand-in). */
    class Matrix
    {
    protected:
        int r_, c_;
        std::vector<double> d_;
    public:
        Matrix() : r_(0), c_(0) {}
        Matrix(int r, int c) : r_(r), c_(c), d_((std::size_t)r * c, 0.0) {}
        /** from / to any other column-major matrix type with data() / rows() / cols() -- an Eigen::Matrix<double, 12, 12> of
         *  a caller, say: what the reference hands out as SingleStateCovariance etc. converts into the caller's own type
         *  (resized first if that type has resize(rows, cols); a fixed-size type must match) */
        template <class M, class = typename std::enable_if<!std::is_base_of<Matrix, M>::value, decltype((void)std::declval<const M &>().data(),
                  (void)std::declval<const M &>().rows(), (void)std::declval<const M &>().cols())>::type>
        Matrix(const M &o) : r_((int)o.rows()), c_((int)o.cols()), d_(o.data(), o.data() + (std::size_t)o.rows() * o.cols()) {}
        template <class M, class = typename std::enable_if<!std::is_base_of<Matrix, M>::value && !std::is_arithmetic<M>::value,
                  decltype((void)std::declval<M &>().data(), (void)std::declval<const M &>().rows(), (void)std::declval<const M &>().cols())>::type>
        operator M() const
        {
            M out;
            resize_if_possible(out, r_, c_, 0);
            assert((int)out.rows() == r_ && (int)out.cols() == c_);
            std::copy(d_.begin(), d_.end(), out.data());
            return out;
        }
    private:
        template <class M> static auto resize_if_possible(M &m, int r, int c, int) -> decltype(m.resize(r, c), void()) { m.resize(r, c); }
        template <class M> static void resize_if_possible(M &, int, int, long) {}
    public:
        static Matrix Zero(int r, int c) { return Matrix(r, c); }
        static Matrix Identity(int r, int c) { Matrix m(r, c); for (int i = 0; i < r && i < c; ++i) m(i, i) = 1.0; return m; }
        void resize(int r, int c) { r_ = r; c_ = c; d_.assign((std::size_t)r * c, 0.0); }
        void setZero() { d_.assign(d_.size(), 0.0); }
        void setIdentity() { setZero(); for (int i = 0; i < r_ && i < c_; ++i) (*this)(i, i) = 1.0; }
        int rows() const { return r_; }
        int cols() const { return c_; }
        int size() const { return r_ * c_; }
        double *data() { return d_.data(); }
        const double *data() const { return d_.data(); }
        double &operator()(int i, int j) { return d_[(std::size_t)j * r_ + i]; }
        double operator()(int i, int j) const { return d_[(std::size_t)j * r_ + i]; }
        double &operator()(int i) { return d_[i]; }
        double operator()(int i) const { return d_[i]; }
        double &operator[](int i) { return d_[i]; }
        double operator[](int i) const { return d_[i]; }
        void comma_set(int i, double v) { (*this)(i / c_, i % c_) = v; }       // row by row, like Eigen
        CommaInit<Matrix> operator<<(double v) { return CommaInit<Matrix>(*this, v); }
        Matrix operator*(double s) const { Matrix m(*this); for (double &v : m.d_) v *= s; return m; }
        Matrix operator/(double s) const { Matrix m(*this); for (double &v : m.d_) v /= s; return m; }
        Matrix operator+(const Matrix &o) const { assert(r_ == o.r_ && c_ == o.c_); Matrix m(*this); for (std::size_t i = 0; i < d_.size(); ++i) m.d_[i] += o.d_[i]; return m; }
        Matrix operator-(const Matrix &o) const { assert(r_ == o.r_ && c_ == o.c_); Matrix m(*this); for (std::size_t i = 0; i < d_.size(); ++i) m.d_[i] -= o.d_[i]; return m; }
        Matrix operator-() const { Matrix m(*this); for (double &v : m.d_) v = -v; return m; }
        Matrix operator*(const Matrix &o) const
        {
            assert(c_ == o.r_);
            Matrix m(r_, o.c_);
            for (int j = 0; j < o.c_; ++j) for (int k = 0; k < c_; ++k) { const double b = o(k, j); for (int i = 0; i < r_; ++i) m(i, j) += (*this)(i, k) * b; }
            return m;
        }
        Matrix transpose() const { Matrix m(c_, r_); for (int j = 0; j < c_; ++j) for (int i = 0; i < r_; ++i) m(j, i) = (*this)(i, j); return m; }
        Matrix block(int i0, int j0, int nr, int nc) const
        {
            Matrix m(nr, nc);
            for (int j = 0; j < nc; ++j) for (int i = 0; i < nr; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
            return m;
        }
        void setBlock(int i0, int j0, const Matrix &b) { for (int j = 0; j < b.cols(); ++j) for (int i = 0; i < b.rows(); ++i) (*this)(i0 + i, j0 + j) = b(i, j); }
        double norm() const { double s = 0; for (double v : d_) s += v * v; return std::sqrt(s); }
        double maxAbs() const { double s = 0; for (double v : d_) s = std::fabs(v) > s ? std::fabs(v) : s; return s; }
        /** Eigen's isZero(prec): every |coefficient| <= prec (compared against 1) */
        bool isZero(double prec = 1e-12) const { for (double v : d_) if (!(std::fabs(v) <= prec)) return false; return true; }
        const Matrix &matrix() const { return *this; }
        friend std::ostream &operator<<(std::ostream &os, const Matrix &m)
        {
            for (int i = 0; i < m.r_; ++i) { for (int j = 0; j < m.c_; ++j) os << (j ? " " : "") << m(i, j); if (i + 1 < m.r_) os << "\n"; }
            return os;
        }
    };
    inline Matrix operator*(double s, const Matrix &m) { return m * s; }

    /** Dynamic column vector (Eigen::VectorXd / MTK::vect<Dynamic> stand-in). */
    class Vector : public Matrix
    {
    public:
        Vector() {}
        explicit Vector(int n) : Matrix(n, 1) {}
        Vector(int n, int) : Matrix(n, 1) {}
        Vector(const Matrix &m) : Matrix(m) { assert(m.cols() == 1 || m.size() == 0); }
        static Vector Zero(int n) { return Vector(n); }
        void resize(int n) { Matrix::resize(n, 1); }
        void resize(int n, int) { Matrix::resize(n, 1); }
        void comma_set(int i, double v) { d_[i] = v; }
        CommaInit<Vector> operator<<(double v) { return CommaInit<Vector>(*this, v); }
        Vector operator+(const Vector &o) const { return Vector(Matrix::operator+(o)); }
        Vector operator-(const Vector &o) const { return Vector(Matrix::operator-(o)); }
        Vector operator-() const { return Vector(Matrix::operator-()); }
        Vector operator*(double s) const { return Vector(Matrix::operator*(s)); }
        Vector segment(int i0, int n) const { Vector v(n); for (int i = 0; i < n; ++i) v[i] = d_[i0 + i]; return v; }
        /** MTK::vect text form (State.hpp:202-210 streams its members with these): the elements, each followed by a blank */
        friend std::ostream &operator<<(std::ostream &os, const Vector &v) { for (int i = 0; i < v.size(); ++i) os << v[i] << " "; return os; }
        /** MTK::vect operator>>: optional ( [ { around the elements, optional commas.  Bracketed input resizes the vector to
         *  what is read (MTK requires the brackets for dynamic vectors); bare input fills the current size. */
        friend std::istream &operator>>(std::istream &is, Vector &v)
        {
            char term = 0;
            is >> std::ws;
            switch (is.peek()) { case '(': term = ')'; is.ignore(1); break; case '[': term = ']'; is.ignore(1); break; case '{': term = '}'; is.ignore(1); break; default: break; }
            if (term) {
                std::vector<double> vals;
                for (;;) {
                    is >> std::ws;
                    if (!is || is.peek() == term) break;
                    double x; is >> x; if (!is) break;
                    vals.push_back(x);
                    skip_separator(is);
                }
                char x; is >> x; if (x != term) is.setstate(std::ios::failbit);
                v.resize((int)vals.size());
                for (std::size_t i = 0; i < vals.size(); ++i) v[(int)i] = vals[i];
            } else {
                for (int i = 0; i < v.size(); ++i) { is >> v[i]; skip_separator(is); }
            }
            return is;
        }
    };
    inline Vector operator*(double s, const Vector &v) { return v * s; }

    /** Fixed 3-vector (Eigen::Vector3d / MTK::vect<3> stand-in). */
    struct Vec3
    {
        double v[3];
        Vec3() { v[0] = v[1] = v[2] = 0.0; }
        Vec3(double x, double y, double z) { v[0] = x; v[1] = y; v[2] = z; }
        Vec3(const Vector &o) { assert(o.size() == 3); v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; }
        static Vec3 Zero() { return Vec3(); }
        double &operator[](int i) { return v[i]; }
        double operator[](int i) const { return v[i]; }
        double &operator()(int i) { return v[i]; }
        double operator()(int i) const { return v[i]; }
        double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
        double *data() { return v; }
        const double *data() const { return v; }
        int size() const { return 3; }
        int rows() const { return 3; }
        int cols() const { return 1; }
        void comma_set(int i, double x) { v[i] = x; }
        CommaInit<Vec3> operator<<(double x) { return CommaInit<Vec3>(*this, x); }
        Vec3 operator+(const Vec3 &o) const { return Vec3(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
        Vec3 operator-(const Vec3 &o) const { return Vec3(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
        Vec3 operator-() const { return Vec3(-v[0], -v[1], -v[2]); }
        Vec3 operator*(double s) const { return Vec3(v[0] * s, v[1] * s, v[2] * s); }
        Vec3 operator/(double s) const { return Vec3(v[0] / s, v[1] / s, v[2] / s); }
        Vec3 &operator+=(const Vec3 &o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
        double dot(const Vec3 &o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
        Vec3 cross(const Vec3 &o) const { return Vec3(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]); }
        double squaredNorm() const { return dot(*this); }
        double norm() const { return std::sqrt(squaredNorm()); }
        operator Vector() const { Vector o(3); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; return o; }
        /** MTK::vect boxplus / boxminus: x += s * v, x - y */
        void boxplus(const double *d, double scale = 1) { for (int i = 0; i < 3; ++i) v[i] += scale * d[i]; }
        void boxplus(const Vec3 &d, double scale = 1) { boxplus(d.v, scale); }
        void boxminus(double *res, const Vec3 &o) const { for (int i = 0; i < 3; ++i) res[i] = v[i] - o.v[i]; }
        enum { DOF = 3 };
        typedef double scalar;
        friend std::ostream &operator<<(std::ostream &os, const Vec3 &a) { return os << a.v[0] << " " << a.v[1] << " " << a.v[2] << " "; }
        /** MTK::vect operator>>: an optional ( [ { around the elements, optional commas between them */
        friend std::istream &operator>>(std::istream &is, Vec3 &a)
        {
            char term = 0;
            is >> std::ws;
            switch (is.peek()) { case '(': term = ')'; is.ignore(1); break; case '[': term = ']'; is.ignore(1); break; case '{': term = '}'; is.ignore(1); break; default: break; }
            for (int i = 0; i < 3; ++i) { is >> a.v[i]; skip_separator(is); }
            if (term) { char x; is >> x; if (x != term) is.setstate(std::ios::failbit); }
            return is;
        }
    };
    inline Vec3 operator*(double s, const Vec3 &a) { return a * s; }

    /** 3x3 matrix, row access m(i, j) (Eigen::Matrix3d stand-in for rotation matrices). */
    struct Mat3
    {
        double m[3][3];
        Mat3() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = 0.0; }
        static Mat3 Identity() { Mat3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1.0; return r; }
        double &operator()(int i, int j) { return m[i][j]; }
        double operator()(int i, int j) const { return m[i][j]; }
        Vec3 operator*(const Vec3 &x) const { return Vec3(m[0][0] * x[0] + m[0][1] * x[1] + m[0][2] * x[2], m[1][0] * x[0] + m[1][1] * x[1] + m[1][2] * x[2], m[2][0] * x[0] + m[2][1] * x[1] + m[2][2] * x[2]); }
        Mat3 operator*(const Mat3 &o) const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) r.m[i][j] += m[i][k] * o.m[k][j]; return r; }
        Mat3 transpose() const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
    };

    /** Unit quaternion, coefficient order (x, y, z, w) like Eigen::Quaternion::coeffs(), with the SO(3) manifold
     *  interface of MTK::SO3<double> (third-party of the reference; restated from its published algorithm, see
     *  SURVEY.md Appendix A): exp(v, s) = (sinc-form sin(|v|s/2)/|v| * v, cos(|v|s/2)), log(q) = 2 atan(|vec|/w)/|vec| * vec
     *  (|vec| clamped to 1e-11), boxplus q <- q * exp(v, s), boxminus log(other^-1 * q).  Used by the reference at
     *  src/filters/State.hpp:82, 90, 96, 128, 166-200 and by its test models (quaternion * vector). */
    struct Quaternion
    {
        double c[4];
        Quaternion() { c[0] = c[1] = c[2] = 0.0; c[3] = 1.0; }
        Quaternion(double w, double x, double y, double z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }   // Eigen ctor order
        static Quaternion Identity() { return Quaternion(); }
        double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
        double &x() { return c[0]; } double &y() { return c[1]; } double &z() { return c[2]; } double &w() { return c[3]; }
        const double *coeffs() const { return c; }
        double *coeffs() { return c; }
        Vec3 vec() const { return Vec3(c[0], c[1], c[2]); }
        Quaternion operator*(const Quaternion &b) const
        {
            Quaternion o;
            o.c[3] = w() * b.w() - x() * b.x() - y() * b.y() - z() * b.z();
            o.c[0] = w() * b.x() + x() * b.w() + y() * b.z() - z() * b.y();
            o.c[1] = w() * b.y() + y() * b.w() + z() * b.x() - x() * b.z();
            o.c[2] = w() * b.z() + z() * b.w() + x() * b.y() - y() * b.x();
            return o;
        }
        /** Eigen QuaternionBase::_transformVector: v + w (2 u x v) + u x (2 u x v) */
        Vec3 operator*(const Vec3 &v) const
        {
            const Vec3 u = vec();
            const Vec3 uv = u.cross(v) * 2.0;
            return v + uv * w() + u.cross(uv);
        }
        Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
        Quaternion inverse() const { const double n2 = x() * x() + y() * y() + z() * z() + w() * w(); return Quaternion(w() / n2, -x() / n2, -y() / n2, -z() / n2); }
        void normalize() { const double n = std::sqrt(x() * x() + y() * y() + z() * z() + w() * w()); for (int i = 0; i < 4; ++i) c[i] /= n; }
        Mat3 toRotationMatrix() const
        {
            const double tx = 2 * x(), ty = 2 * y(), tz = 2 * z();
            const double twx = tx * w(), twy = ty * w(), twz = tz * w(), txx = tx * x(), txy = ty * x(), txz = tz * x();
            const double tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
            Mat3 r;
            r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
            r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
            r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
            return r;
        }
        /** MTK::SO3::exp(vec, scale) */
        template <class V3>
        static Quaternion exp(const V3 &v, double scale = 1.0)
        {
            const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            const double x = 0.25 * scale * scale * n2;           // (theta / 2)^2
            double cs, sn;                                        // MTK cos_sinc_sqrt: cos(sqrt x), sin(sqrt x) / sqrt x,
            if (x >= 1.220703125e-4) {                            // Taylor series (3 terms) below eps^(1/4)
                const double sx = std::sqrt(x); cs = std::cos(sx); sn = std::sin(sx) / sx;
            } else {
                static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
                cs = 1.; sn = 1.;
                double term = -1 / 2. * x;
                for (int i = 0; i < 3; ++i) { cs += term; term *= inv[2 * i]; sn += term; term *= -inv[2 * i + 1] * x; }
            }
            const double k = sn * 0.5 * scale;
            return Quaternion(cs, k * v[0], k * v[1], k * v[2]);
        }
        /** MTK::SO3::log(q): atan (not atan2), so q and -q give the same rotation vector */
        static Vec3 log(const Quaternion &q)
        {
            double nv = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
            if (nv < 1e-11) nv = 1e-11;
            const double s = 2.0 / nv * std::atan(nv / q.w());
            return Vec3(s * q.x(), s * q.y(), s * q.z());
        }
        void boxplus(const double *v, double scale = 1) { *this = (*this) * exp(v, scale); }
        void boxplus(const Vec3 &v, double scale = 1) { boxplus(v.v, scale); }
        void boxminus(double *res, const Quaternion &o) const { const Vec3 r = log(o.conjugate() * (*this)); res[0] = r[0]; res[1] = r[1]; res[2] = r[2]; }
        enum { DOF = 3 };
        typedef double scalar;
        /** MTK::SO3 text form: the four coefficients x y z w; reading normalises (MTK reads a vect<4> and normalises it) */
        friend std::ostream &operator<<(std::ostream &os, const Quaternion &q) { return os << q.c[0] << " " << q.c[1] << " " << q.c[2] << " " << q.c[3] << " "; }
        friend std::istream &operator>>(std::istream &is, Quaternion &q)
        {
            char term = 0;
            is >> std::ws;
            switch (is.peek()) { case '(': term = ')'; is.ignore(1); break; case '[': term = ']'; is.ignore(1); break; case '{': term = '}'; is.ignore(1); break; default: break; }
            for (int i = 0; i < 4; ++i) { is >> q.c[i]; skip_separator(is); }
            if (term) { char x; is >> x; if (x != term) is.setstate(std::ios::failbit); }
            if (is) q.normalize();
            return is;
        }
    };

    /** Rigid transform (Eigen::Affine3d stand-in as the reference's models use it: UsckfUnitTest.cpp:71-72, 79). */
    struct Affine3
    {
        Mat3 R;
        Vec3 t;
        Affine3() : R(Mat3::Identity()) {}
        Affine3(const Quaternion &q) : R(q.toRotationMatrix()) {}
        static Affine3 Identity() { return Affine3(); }
        Vec3 &translation() { return t; }
        const Vec3 &translation() const { return t; }
        const Mat3 &linear() const { return R; }
        Mat3 &linear() { return R; }
        Vec3 operator*(const Vec3 &x) const { return R * x + t; }
        Affine3 operator*(const Affine3 &o) const { Affine3 a; a.R = R * o.R; a.t = R * o.t + t; return a; }
    };
} // namespace slk
} // namespace localization

#endif
