// Test scaffolding only: the names the reference's test sources spell (Eigen::Vector3d, Eigen::Affine3d,
// Eigen::Dynamic, MTK::vect, MTK::setDiagonal, localization::D2R) bound to the facade's own dependency-free types
// (include/localization/filters/SlkTypes.hpp), so that model functions copied from test/MsckfUnitTest.cpp:33-47 and
// test/UsckfUnitTest.cpp:34-86 compile UNCHANGED against the GPU-backed facade.  A client that has Eigen / MTK on its
// include path passes those types instead; nothing here is part of the product.
#ifndef TESTS_EIGEN_MTK_NAMES_HPP
#define TESTS_EIGEN_MTK_NAMES_HPP

#include <cmath>

#include <localization/filters/MtkWrap.hpp>
#include <localization/filters/State.hpp>

namespace Eigen
{
    enum { Dynamic = -1 };
    typedef localization::slk::Vec3 Vector3d;
    typedef localization::slk::Affine3 Affine3d;
    typedef localization::slk::Matrix MatrixXd;
    typedef localization::slk::Vector VectorXd;
}

namespace MTK
{
    template <int D, class S> struct vect_of { typedef localization::slk::Vector type; };
    template <class S> struct vect_of<3, S> { typedef localization::slk::Vec3 type; };
    template <int D, class S> using vect = typename vect_of<D, S>::type;
    // MTK::setDiagonal(cov, &State::member, value): the member's tangent offset comes from SubManifold<T, idx> in the
    // reference (State.hpp:141-144: pos 0, orient 3, velo 6, angvelo 9)
    inline int start_idx(localization::vec3 localization::State::*m)
    {
        return m == &localization::State::pos ? 0 : (m == &localization::State::velo ? 6 : 9);
    }
    inline int start_idx(localization::SO3 localization::State::*) { return 3; }
    template <class Mat, class T>
    inline void setDiagonal(Mat &cov, T localization::State::*m, double v)
    {
        const int o = start_idx(m);
        for (int i = 0; i < 3; ++i) cov(o + i, o + i) = v;
    }
}

namespace localization
{
    static const double D2R = M_PI / 180.00;   /** src/Configuration.hpp:36 */
    static const double R2D = 180.00 / M_PI;   /** src/Configuration.hpp:37 */
}
#endif
