/**\file SlkBackend.hpp
 * Glue between the header facade (Usckf.hpp / Msckf.hpp) and the C ABI (include/slk.h):
 * RAII handle, registered-model tag types (Tier A) and the host-functor path (Tier B).
 */
#ifndef _SLK_BACKEND_HPP_
#define _SLK_BACKEND_HPP_

#include <cmath>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../slk.h"
#include "State.hpp"

namespace localization
{
namespace slk
{
    inline void check(int rc, const char *what)
    {
        if (rc != SLK_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + slk_last_error());
    }

    /** Owns one slk_filter (a batch of B filters; B = 1 is the reference object). */
    class Handle
    {
        slk_filter *h_;
    public:
        Handle() : h_(0) {}
        ~Handle() { reset(); }
        Handle(const Handle &) = delete;
        Handle &operator=(const Handle &) = delete;
        void reset() { if (h_) slk_destroy(h_); h_ = 0; }
        void create(int kind, int batch, int n_clones, int nfk, int nfkl, int device = 0, void *stream = 0)
        {
            reset();
            slk_config cfg = {kind, batch, device, n_clones, nfk, nfkl, stream};
            check(slk_create(&cfg, &h_), "slk_create");
        }
        slk_filter *get() const { return h_; }
        int N() const { return slk_dof(h_); }
        int Nq() const { return slk_storage(h_); }
        int B() const { return slk_batch(h_); }
    };

    /** Registered process models (run on the GPU).  Any other callable takes the Tier-B path. */
    struct ConstVelocityModel      /* test/UsckfUnitTest.cpp:34-49 */
    {
        double u[7];
        ConstVelocityModel(const Vec3 &velocity, const Vec3 &angular_velocity, double dt)
        { for (int i = 0; i < 3; ++i) { u[i] = velocity[i]; u[3 + i] = angular_velocity[i]; } u[6] = dt; }
    };
    struct DeltaPoseModel          /* test/MsckfUnitTest.cpp:33-47 */
    {
        double u[13];
        DeltaPoseModel(const Vec3 &dpos, const Quaternion &dq, const Vec3 &velocity, const Vec3 &angular_velocity)
        {
            for (int i = 0; i < 3; ++i) { u[i] = dpos[i]; u[7 + i] = velocity[i]; u[10 + i] = angular_velocity[i]; }
            for (int i = 0; i < 4; ++i) u[3 + i] = dq.coeffs()[i];
        }
    };
    struct DeadReckonModel         /* src/core/DeadReckon.hpp:129-239 (delta pose from two velocity samples) feeding the delta-pose model */
    {
        double u[13];
        DeadReckonModel(double dt, const Vec3 &velocity, const Vec3 &angular_velocity,
                        const Vec3 &prev_velocity, const Vec3 &prev_angular_velocity)
        {
            u[0] = dt;
            for (int i = 0; i < 3; ++i) { u[1 + i] = velocity[i]; u[4 + i] = angular_velocity[i]; u[7 + i] = prev_velocity[i]; u[10 + i] = prev_angular_velocity[i]; }
        }
    };
    /** Registered measurement models. */
    struct VoRelativeModel {};     /* test/UsckfUnitTest.cpp:62-86 */
    struct FeatureProjectionModel  /* m/2 landmarks seen as normalised image points from pose indices */
    {
        std::vector<double> params;   /* (x, y, z, pose index) per feature */
        void add(double x, double y, double z, int pose) { params.push_back(x); params.push_back(y); params.push_back(z); params.push_back(pose); }
    };
    struct PosePositionModel { double pose; explicit PosePositionModel(int p) : pose(p) {} };

    template <class T> struct is_registered_process { enum { value = 0 }; };
    template <> struct is_registered_process<ConstVelocityModel> { enum { value = 1 }; };
    template <> struct is_registered_process<DeltaPoseModel> { enum { value = 1 }; };
    template <> struct is_registered_process<DeadReckonModel> { enum { value = 1 }; };

    /** Noise arguments come as a matrix (anything with data()) or, in the reference's general overloads, as a nullary
     *  functor returning one (boost::bind(ukfom::id<Cov>, Q): Msckf.hpp:94, :148; Usckf.hpp:164, :282). */
    template <class T> inline auto noise_matrix(const T &q, int) -> decltype(q.data(), q) { return q; }
    template <class T> inline auto noise_matrix(const T &q, long) -> decltype(q()) { return q(); }

    /** has data() and rows(): a matrix argument (H, R), as opposed to a significance test or a noise functor */
    template <class T> struct is_matrix_like
    {
        template <class U> static char test(decltype(std::declval<const U &>().data()) *, decltype(std::declval<const U &>().rows()) *);
        template <class U> static long test(...);
        enum { value = sizeof(test<T>(0, 0)) == sizeof(char) };
    };

    /** Significance tests (`mt`): what the caller passed as the last argument of update().
     *  0 = accept everything, 1 = the library's chi-square gate (runs inside the kernel), 2 = any other callable
     *  (evaluated on the host between two launches). */
    inline int gate_kind(bool g) { return g ? 1 : 0; }
    inline int gate_kind(int g) { return g ? 1 : 0; }
    template <class S> inline int gate_kind(bool (*fn)(const S &, const int), bool (*builtin)(const S &, const int)) { return fn == builtin ? 1 : 2; }

    /** dense inverse (Gauss-Jordan, partial pivoting) of a small host matrix -- the reference inverts the EKF information
     *  matrix with Eigen's PartialPivLU (Msckf.hpp:765-766); host side of the custom-`mt` EKF path only */
    inline Matrix inverse(const Matrix &A)
    {
        const int n = A.rows();
        Matrix M(A), I = Matrix::Identity(n, n);
        for (int k = 0; k < n; ++k) {
            int p = k;
            for (int i = k + 1; i < n; ++i) if (std::fabs(M(i, k)) > std::fabs(M(p, k))) p = i;
            if (p != k) for (int j = 0; j < n; ++j) { std::swap(M(k, j), M(p, j)); std::swap(I(k, j), I(p, j)); }
            const double d = M(k, k);
            for (int j = 0; j < n; ++j) { M(k, j) /= d; I(k, j) /= d; }
            for (int i = 0; i < n; ++i) {
                if (i == k) continue;
                const double f = M(i, k);
                if (f == 0.0) continue;
                for (int j = 0; j < n; ++j) { M(i, j) -= f * M(k, j); I(i, j) -= f * I(k, j); }
            }
        }
        return I;
    }

    /** column-major copy of anything with data()/rows()/cols() */
    template <class M>
    inline std::vector<double> dense(const M &m) { return std::vector<double>(m.data(), m.data() + (std::size_t)m.rows() * m.cols()); }
} // namespace slk
} // namespace localization
#endif
