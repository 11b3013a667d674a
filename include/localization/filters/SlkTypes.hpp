/**\file SlkTypes.hpp
 * Small dense types used by the GPU-backed facade of localization::Usckf / localization::Msckf
 * when Eigen / MTK are not on the include path.  They follow Eigen's conventions (column-major
 * storage, data()/rows()/cols()/operator()(i,j), quaternion coefficient order x,y,z,w), so that
 * a caller compiled against Eigen can pass Eigen::Matrix / Eigen::Quaternion objects to the same
 * templated entry points (only data(), rows(), cols(), size() are used on matrix arguments).
 */
#ifndef _SLK_TYPES_HPP_
#define _SLK_TYPES_HPP_

#include <cassert>
#include <cmath>
#include <cstddef>
#include <vector>

namespace localization
{
namespace slk
{
    /** Column-major dynamic matrix (Eigen::MatrixXd stand-in). */
    class Matrix
    {
        int r_, c_;
        std::vector<double> d_;
    public:
        Matrix() : r_(0), c_(0) {}
        Matrix(int r, int c) : r_(r), c_(c), d_((std::size_t)r * c, 0.0) {}
        static Matrix Zero(int r, int c) { return Matrix(r, c); }
        static Matrix Identity(int r, int c) { Matrix m(r, c); for (int i = 0; i < r && i < c; ++i) m(i, i) = 1.0; return m; }
        void resize(int r, int c) { r_ = r; c_ = c; d_.assign((std::size_t)r * c, 0.0); }
        void setZero() { d_.assign(d_.size(), 0.0); }
        int rows() const { return r_; }
        int cols() const { return c_; }
        int size() const { return r_ * c_; }
        double *data() { return d_.data(); }
        const double *data() const { return d_.data(); }
        double &operator()(int i, int j) { return d_[(std::size_t)j * r_ + i]; }
        double operator()(int i, int j) const { return d_[(std::size_t)j * r_ + i]; }
        double &operator[](int i) { return d_[i]; }
        double operator[](int i) const { return d_[i]; }
        Matrix operator*(double s) const { Matrix m(*this); for (double &v : m.d_) v *= s; return m; }
        Matrix block(int i0, int j0, int nr, int nc) const
        {
            Matrix m(nr, nc);
            for (int j = 0; j < nc; ++j) for (int i = 0; i < nr; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
            return m;
        }
    };
    inline Matrix operator*(double s, const Matrix &m) { return m * s; }

    /** Dynamic column vector. */
    class Vector : public Matrix
    {
    public:
        Vector() {}
        explicit Vector(int n) : Matrix(n, 1) {}
        void resize(int n) { Matrix::resize(n, 1); }
        void resize(int n, int) { Matrix::resize(n, 1); }
    };

    struct Vec3
    {
        double v[3];
        Vec3() { v[0] = v[1] = v[2] = 0.0; }
        Vec3(double x, double y, double z) { v[0] = x; v[1] = y; v[2] = z; }
        double &operator[](int i) { return v[i]; }
        double operator[](int i) const { return v[i]; }
        enum { DOF = 3 };
    };

    /** Unit quaternion, coefficient order (x, y, z, w) like Eigen::Quaternion::coeffs().
     *  exp follows MTK::SO3::exp (src/filters/State.hpp:179 in the reference). */
    struct Quaternion
    {
        double c[4];
        Quaternion() { c[0] = c[1] = c[2] = 0.0; c[3] = 1.0; }
        Quaternion(double w, double x, double y, double z) { c[0] = x; c[1] = y; c[2] = z; c[3] = w; }   // Eigen ctor order
        double x() const { return c[0]; } double y() const { return c[1]; } double z() const { return c[2]; } double w() const { return c[3]; }
        const double *coeffs() const { return c; }
        double *coeffs() { return c; }
        Quaternion operator*(const Quaternion &b) const
        {
            Quaternion o;
            o.c[3] = w() * b.w() - x() * b.x() - y() * b.y() - z() * b.z();
            o.c[0] = w() * b.x() + x() * b.w() + y() * b.z() - z() * b.y();
            o.c[1] = w() * b.y() + y() * b.w() + z() * b.x() - x() * b.z();
            o.c[2] = w() * b.z() + z() * b.w() + x() * b.y() - y() * b.x();
            return o;
        }
        static Quaternion exp(const Vec3 &v, double scale = 1.0)
        {
            double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) * scale, h = 0.5 * th;
            double k = (th > 1e-12) ? std::sin(h) / th * scale : 0.5 * scale;
            return Quaternion(std::cos(h), k * v[0], k * v[1], k * v[2]);
        }
        enum { DOF = 3 };
    };
} // namespace slk
} // namespace localization

#endif
