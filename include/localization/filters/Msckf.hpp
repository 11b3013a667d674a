/**\file Msckf.hpp
 * GPU-backed localization::Msckf: same class template, typedefs and method names as the
 * reference's src/filters/Msckf.hpp; the numerics run in libslk_hip.so (include/slk.h).
 *
 *   predict(f, Q)         Msckf.hpp:89-189   -> slk_predict (registered f) or the sigma-point path
 *   update(z, h, R[, mt]) Msckf.hpp:196-277  -> slk_update  (registered h) or the sigma-point path
 *   muSingleState / setPkSingleState / getPkSingleState / muState / getPk / setPk  :351-395
 *
 * `f` / `h` may be a registered model object (slk::DeltaPoseModel, slk::FeatureProjectionModel, ...)
 * -- evaluated on the GPU -- or ANY callable like the reference's boost::bind functors
 * (test/MsckfUnitTest.cpp:200-205): then the sigma points are drawn on the GPU, mapped by the
 * callable on the host and the step is finished on the GPU (same results).
 * The EKF overloads of the reference (:284-349) are not part of the sigma-point hot path and are
 * not provided here.
 */
#ifndef _MSCKF_HPP_
#define _MSCKF_HPP_

#include <algorithm>
#include <vector>

#include "SlkBackend.hpp"

namespace localization
{
    template <typename _MultiState, typename _SingleState>
    class Msckf
    {
        typedef Msckf self;
    public:
        enum { DOF_SINGLE_STATE = _MultiState::SingleState::DOF };
        enum { SENSOR_DOF = _MultiState::SENSOR_DOF };
        typedef typename _MultiState::scalar_type ScalarType;
        typedef slk::Vector VectorizedSingleState;
        typedef slk::Matrix SingleStateCovariance;
        typedef std::vector<_SingleState> SingleStateSigma;
        typedef slk::Vector VectorizedMultiState;
        typedef slk::Matrix MultiStateCovariance;
        typedef std::vector<_MultiState> MultiStateSigma;

    private:
        mutable _MultiState mu_state;       /** host mirror of the mean (refreshed lazily) **/
        mutable MultiStateCovariance Pk;    /** host mirror of the covariance **/
        mutable bool mean_stale, cov_stale;
        slk::Handle h;
        unsigned int last_outliers;

        void upload()
        {
            std::vector<double> m(h.Nq());
            slk_store(mu_state, m.data());
            slk::check(slk_set_state(h.get(), m.data(), Pk.data(), SLK_HOST), "slk_set_state");
            mean_stale = cov_stale = false;
        }
        void pull_mean() const
        {
            if (!mean_stale) return;
            std::vector<double> m(h.Nq());
            slk::check(slk_get_state(h.get(), m.data(), 0, SLK_HOST), "slk_get_state");
            slk_load(mu_state, m.data());
            mean_stale = false;
        }
        void pull_cov() const
        {
            if (!cov_stale) return;
            Pk.resize(h.N(), h.N());
            slk::check(slk_get_state(h.get(), 0, Pk.data(), SLK_HOST), "slk_get_state");
            cov_stale = false;
        }

    public:
        /**@brief Constructor (Msckf.hpp:80-85) */
        template <class Cov>
        Msckf(const _MultiState &state, const Cov &P0) : mu_state(state), mean_stale(false), cov_stale(false), last_outliers(0)
        {
            Pk.resize(P0.rows(), P0.cols());
            std::copy(P0.data(), P0.data() + (std::size_t)P0.rows() * P0.cols(), Pk.data());
            h.create(SLK_MSCKF, 1, (int)state.sensorsk.size(), 0, 0);
            upload();
        }

        /**@brief Filter prediction step, registered process model on the GPU (Msckf.hpp:89-95) */
        template <class Cov>
        void predict(const slk::DeltaPoseModel &f, const Cov &Q)
        {
            slk::check(slk_predict(h.get(), SLK_PM_DELTA_POSE, f.u, 0, Q.data(), 0, SLK_HOST), "slk_predict");
            mean_stale = cov_stale = true;
        }
        template <class Cov>
        void predict(const slk::ConstVelocityModel &f, const Cov &Q)
        {
            slk::check(slk_predict(h.get(), SLK_PM_CONST_VELOCITY, f.u, 0, Q.data(), 0, SLK_HOST), "slk_predict");
            mean_stale = cov_stale = true;
        }
        /** dead reckoning fused into the prediction (src/core/DeadReckon.hpp:129-239 -> the delta-pose model) */
        template <class Cov>
        void predict(const slk::DeadReckonModel &f, const Cov &Q)
        {
            slk::check(slk_predict(h.get(), SLK_PM_DEAD_RECKON, f.u, 0, Q.data(), 0, SLK_HOST), "slk_predict");
            mean_stale = cov_stale = true;
        }
        /**@brief Filter prediction step with an arbitrary process model functor f: _SingleState -> _SingleState */
        template <typename _ProcessModel, class Cov>
        void predict(_ProcessModel f, const Cov &Q)
        {
            std::vector<double> X(25 * 13), Y(25 * 13);
            slk::check(slk_predict_sigma_points(h.get(), X.data(), SLK_HOST), "slk_predict_sigma_points");
            for (int i = 0; i < 25; ++i) {                       // std::transform(X, X, f), Msckf.hpp:125
                _SingleState x, y;
                slk_load(x, &X[13 * i]);
                y = f(x);
                slk_store(y, &Y[13 * i]);
            }
            slk::check(slk_predict_from_sigma(h.get(), Y.data(), Q.data(), 0, SLK_HOST), "slk_predict_from_sigma");
            mean_stale = cov_stale = true;
        }

        /**@brief UKF update with a registered measurement model (Msckf.hpp:196-213); returns the outlier count (:276) */
        template <typename _Measurement, class Cov>
        unsigned int update(const _Measurement &z, const slk::FeatureProjectionModel &hmodel, const Cov &R, bool gate = true)
        {
            slk::check(slk_update(h.get(), SLK_MM_FEATURE_PROJ, hmodel.params.data(), 0, z.data(), (int)z.size(), R.data(), 0,
                                  gate ? 1 : 0, SLK_HOST), "slk_update");
            return finish_update();
        }
        template <typename _Measurement, class Cov>
        unsigned int update(const _Measurement &z, const slk::PosePositionModel &hmodel, const Cov &R, bool gate = true)
        {
            slk::check(slk_update(h.get(), SLK_MM_POSE_POSITION, &hmodel.pose, 0, z.data(), (int)z.size(), R.data(), 0,
                                  gate ? 1 : 0, SLK_HOST), "slk_update");
            return finish_update();
        }
        /**@brief UKF update with an arbitrary measurement functor h: _MultiState -> vector (Msckf.hpp:220-277) */
        template <typename _Measurement, typename _MeasurementModel, class Cov>
        unsigned int update(const _Measurement &z, _MeasurementModel hfun, const Cov &R, bool gate = true)
        {
            const int N = h.N(), Nq = h.Nq(), S = 2 * N + 1, m = (int)z.size();
            std::vector<double> X((std::size_t)S * Nq), Z((std::size_t)S * m);
            slk::check(slk_update_sigma_points(h.get(), X.data(), SLK_HOST), "slk_update_sigma_points");
            _MultiState x(mu_state);
            for (int i = 0; i < S; ++i) {                        // std::transform(X, Z, h), Msckf.hpp:231-232
                slk_load(x, &X[(std::size_t)i * Nq]);
                const _Measurement zi = hfun(x);
                for (int r = 0; r < m; ++r) Z[(std::size_t)i * m + r] = zi[r];
            }
            slk::check(slk_update_from_sigma(h.get(), Z.data(), z.data(), m, R.data(), 0, gate ? 1 : 0, SLK_HOST),
                       "slk_update_from_sigma");
            return finish_update();
        }

        /**@brief EKF update (Msckf.hpp:284-349): h(mu_state, H) returns the expected measurement and fills the
         * Jacobian H (m x N) at the current mean, exactly like the reference's functor (:310); the gate, the Householder
         * compression (reduceDimension) and the gain run on the GPU.  Returns the outlier count (:348). */
        template <typename _Measurement, typename _MeasurementModel, class Jac, class Cov>
        unsigned int update(const _Measurement &z, _MeasurementModel hfun, Jac &H, const Cov &R, bool gate = true)
        {
            pull_mean();
            const _Measurement mean_z = hfun(mu_state, H);
            const int m = (int)z.size();
            slk::check(slk_update_ekf(h.get(), z.data(), mean_z.data(), H.data(), m, R.data(), 0, gate ? 1 : 0, SLK_HOST),
                       "slk_update_ekf");
            return finish_update();
        }

        void muSingleState(const _SingleState &state)            // Msckf.hpp:351-354
        {
            pull_mean(); pull_cov();
            mu_state.statek = state;
            upload();
        }
        _SingleState muSingleState() { pull_mean(); return mu_state.statek; }          // :356-361
        template <class Cov>
        void setPkSingleState(const Cov &Pk_i)                   // :363-366
        {
            pull_mean(); pull_cov();
            for (int j = 0; j < 12; ++j) for (int i = 0; i < 12; ++i) Pk(i, j) = Pk_i.data()[i + 12 * j];
            upload();
        }
        SingleStateCovariance getPkSingleState() { pull_cov(); return Pk.block(0, 0, 12, 12); }   // :368-374
        const _MultiState &muState() const { pull_mean(); return mu_state; }            // :376-379
        const MultiStateCovariance &getPk() const { pull_cov(); return Pk; }            // :386-389
        /** Replace mean and covariance together (the reference lets callers push/pop clones through the
         *  non-const muState() and then call setPk, :381-395); resizes the device batch if the window changed. */
        template <class Cov>
        void setState(const _MultiState &state, const Cov &Pk_i)
        {
            mu_state = state;
            Pk.resize(Pk_i.rows(), Pk_i.cols());
            std::copy(Pk_i.data(), Pk_i.data() + (std::size_t)Pk_i.rows() * Pk_i.cols(), Pk.data());
            if ((int)state.sensorsk.size() * 6 + 12 != h.N())
                slk::check(slk_msckf_resize(h.get(), (int)state.sensorsk.size()), "slk_msckf_resize");
            upload();
        }
        template <class Cov>
        void setPk(const Cov &Pk_i) { pull_mean(); setState(mu_state, Pk_i); }          // :391-395

        /** per-filter numerical status bits of include/slk.h (the reference reports nothing) */
        int status() { int s = 0; slk::check(slk_get_status(h.get(), &s, SLK_HOST), "slk_get_status"); return s; }

        /** chi-square gate of the reference (:844-905), kept for callers that use it directly */
        template <typename _ScalarType>
        static bool accept_mahalanobis_distance(const _ScalarType &mahalanobis2, const int dof)
        {
            static const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
            return (dof >= 1 && dof <= 9) ? (mahalanobis2 < thr[dof]) : false;
        }

    private:
        unsigned int finish_update()
        {
            slk::check(slk_get_outliers(h.get(), &last_outliers, SLK_HOST), "slk_get_outliers");
            mean_stale = cov_stale = true;
            return last_outliers;
        }
    };
} // namespace localization

#endif // __MSCKF_HPP_
