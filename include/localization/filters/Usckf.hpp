/**\file Usckf.hpp
 * GPU-backed localization::Usckf: same class template, enum CloningMode, typedefs and method
 * names as the reference's src/filters/Usckf.hpp; the numerics run in libslk_hip.so.
 *
 *   Usckf(state, P0) :83-86,  Usckf(single_state, P0_single) :90-103 (places + clones twice)
 *   predict(f, Q) :107-111, predict(f, QFn) :113-244 (Q: matrix or nullary functor returning one)
 *   update(z, h, R) :246-258, update(z, h, RFn, mt) :260-308 (mt: whole-vector significance test bool(d2); an int is the
 *   chi-square dof of the library's own gate, 0 = ukfom::accept_any_mahalanobis_distance, the reference default :249)
 *   updateEKF :310-319 (a stub in the reference too)   checkSigmaPoints :769-789
 *   setMeasurement   :322-389   cloning :391-433
 *   setSingleState / muSingleState / setPkSingleState / PkSingleState / muState / PkAugmentedState :435-526
 *
 * Deviations from the reference, all documented in DESIGN.md: the unconditional std::cout in
 * update() (:298) is dropped; a non-positive Cholesky pivot is reported through status() and the
 * call leaves the filter unchanged instead of continuing with Eigen's half-factored matrix.
 */
#ifndef _USCKF_HPP_
#define _USCKF_HPP_

#include <algorithm>
#include <cassert>
#include <iostream>
#include <vector>

#include "SlkBackend.hpp"

namespace localization
{
    /** Different cloning mechanism (Usckf.hpp:37-42) **/
    enum CloningMode
    {
        STATEK = 1,
        STATEK_L = 2,
        STATEK_I = 3
    };

    template <typename _AugmentedState, typename _SingleState>
    class Usckf
    {
        typedef Usckf self;
    public:
        enum { DOF_AUGMENTED_STATE = _AugmentedState::DOF };
        enum { DOF_SINGLE_STATE = _SingleState::DOF };
        typedef typename _AugmentedState::scalar_type ScalarType;
        typedef slk::Vector VectorizedSingleState;
        typedef slk::Matrix SingleStateCovariance;
        typedef std::vector<_SingleState> SingleStateSigma;
        typedef slk::Vector VectorizedAugmentedState;
        typedef slk::Matrix AugmentedStateCovariance;
        typedef std::vector<_AugmentedState> AugmentedStateSigma;
        typedef slk::Matrix MultiStateCovariance;

    private:
        mutable _AugmentedState mu_state;
        mutable AugmentedStateCovariance Pk;
        mutable bool stale;
        slk::Handle h;
        int nfk, nfkl;

        void upload()
        {
            std::vector<double> m(h.Nq());
            slk_store(mu_state, m.data());
            slk::check(slk_set_state(h.get(), m.data(), Pk.data(), SLK_HOST), "slk_set_state");
            stale = false;
        }
        void pull() const
        {
            if (!stale) return;
            std::vector<double> m(h.Nq());
            Pk.resize(h.N(), h.N());
            slk::check(slk_get_state(h.get(), m.data(), Pk.data(), SLK_HOST), "slk_get_state");
            slk_load(mu_state, m.data(), nfk, nfkl);
            stale = false;
        }

    public:
        /**@brief Constructor (Usckf.hpp:83-86) */
        template <class Cov>
        Usckf(const _AugmentedState &state, const Cov &P0) : mu_state(state), stale(false)
        {
            nfk = state.featuresk.size(); nfkl = state.featuresk_l.size();
            Pk.resize(P0.rows(), P0.cols());
            std::copy(P0.data(), P0.data() + (std::size_t)P0.rows() * P0.cols(), Pk.data());
            h.create(SLK_USCKF, 1, 0, nfk, nfkl);
            upload();
        }
        /**@brief Constructor from the current single state (Usckf.hpp:90-103) */
        template <class Cov>
        Usckf(const _SingleState &single_state, const Cov &P0_single, int /*disambiguate*/ = 0) : stale(false), nfk(0), nfkl(0)
        {
            mu_state.statek_i = single_state;
            Pk.resize(36, 36);
            for (int j = 0; j < 12; ++j) for (int i = 0; i < 12; ++i) Pk(24 + i, 24 + j) = P0_single.data()[i + 12 * j];
            h.create(SLK_USCKF, 1, 0, 0, 0);
            upload();
            this->cloning(STATEK_I);        /** Clone the state k+l = k+i **/
            this->cloning(STATEK_L);        /** Clone the state k = k+l **/
        }

        template <class Cov>
        void predict(const slk::ConstVelocityModel &f, const Cov &Q)
        {
            const auto &Qm = slk::noise_matrix(Q, 0);
            slk::check(slk_predict(h.get(), SLK_PM_CONST_VELOCITY, f.u, 0, Qm.data(), 0, SLK_HOST), "slk_predict");
            stale = true;
        }
        template <class Cov>
        void predict(const slk::DeltaPoseModel &f, const Cov &Q)
        {
            const auto &Qm = slk::noise_matrix(Q, 0);
            slk::check(slk_predict(h.get(), SLK_PM_DELTA_POSE, f.u, 0, Qm.data(), 0, SLK_HOST), "slk_predict");
            stale = true;
        }
        template <class Cov>
        void predict(const slk::DeadReckonModel &f, const Cov &Q)
        {
            const auto &Qm = slk::noise_matrix(Q, 0);
            slk::check(slk_predict(h.get(), SLK_PM_DEAD_RECKON, f.u, 0, Qm.data(), 0, SLK_HOST), "slk_predict");
            stale = true;
        }
        /**@brief predict with an arbitrary process model functor (the reference's boost::bind form, UsckfUnitTest.cpp:246) */
        template <typename _ProcessModel, class Cov>
        void predict(_ProcessModel f, const Cov &Q)
        {
            std::vector<double> X(25 * 13), Y(25 * 13);
            slk::check(slk_predict_sigma_points(h.get(), X.data(), SLK_HOST), "slk_predict_sigma_points");
            for (int i = 0; i < 25; ++i) {
                _SingleState x, y;
                slk_load(x, &X[13 * i]);
                y = f(x);
                slk_store(y, &Y[13 * i]);
            }
            const auto &Qm = slk::noise_matrix(Q, 0);
            slk::check(slk_predict_from_sigma(h.get(), Y.data(), Qm.data(), 0, SLK_HOST), "slk_predict_from_sigma");
            stale = true;
        }

        /**@brief UKF update (Usckf.hpp:246-258): accept_any_mahalanobis_distance */
        template <typename _Measurement, typename _MeasurementModel, typename _MeasurementNoiseCovariance>
        void update(const _Measurement &z, _MeasurementModel hfun, const _MeasurementNoiseCovariance &R)
        {
            update(z, hfun, R, 0);
        }
        /**@brief UKF update with a significance test (Usckf.hpp:260-308), registered relative-transform model
         * (UsckfUnitTest.cpp:62-86).  mt: int = chi-square dof of the library's gate (0 = accept any), or any callable
         * bool(mahalanobis2) evaluated here on the innovation / covariance the GPU computed (:292-294). */
        template <typename _Measurement, class Cov, typename _SignificanceTest>
        void update(const _Measurement &z, const slk::VoRelativeModel &, const Cov &R, _SignificanceTest mt)
        {
            run_update(z, SLK_MM_VO_RELATIVE, 0, 0, slk::noise_matrix(R, 0), mt);
        }
        /**@brief UKF update with an arbitrary measurement functor h: _AugmentedState -> vector (:260-308) */
        template <typename _Measurement, typename _MeasurementModel, class Cov, typename _SignificanceTest>
        void update(const _Measurement &z, _MeasurementModel hfun, const Cov &R, _SignificanceTest mt)
        {
            const int N = h.N(), Nq = h.Nq(), S = 2 * N + 1, m = (int)z.size();
            std::vector<double> X((std::size_t)S * Nq), Z((std::size_t)S * m);
            slk::check(slk_update_sigma_points(h.get(), X.data(), SLK_HOST), "slk_update_sigma_points");
            _AugmentedState x;
            for (int i = 0; i < S; ++i) {                        // std::transform(X, Z, h), Usckf.hpp:277-278
                slk_load(x, &X[(std::size_t)i * Nq], nfk, nfkl);
                const _Measurement zi = hfun(x);
                for (int r = 0; r < m; ++r) Z[(std::size_t)i * m + r] = zi[r];
            }
            run_update(z, SLK_MODEL_EXTERNAL, 0, &Z, slk::noise_matrix(R, 0), mt);
        }

        /**@brief updateEKF (Usckf.hpp:310-319): a stub in the reference (it sizes a vector, prints and returns);
         * kept so that callers link, does nothing */
        template <typename _Measurement, typename _MeasurementModelMatrix, typename _MeasurementNoiseCovariance>
        void updateEKF(const _Measurement &, _MeasurementModelMatrix &, _MeasurementNoiseCovariance &) {}

        /**@brief checkSigmaPoints (Usckf.hpp:769-789): the sigma points of (mu_state, Pk) are drawn on the GPU
         * (slk_update_sigma_points); their manifold mean and covariance are folded here with the host-side state
         * operators.  Returns whether cov == Pk (1e-6) and mean == mu_state (1e-12), reports the two errors. */
        bool checkSigmaPoints(double &cov_err, double &mean_err)
        {
            pull();
            const int N = h.N(), Nq = h.Nq(), S = 2 * N + 1;
            std::vector<double> X((std::size_t)S * Nq);
            slk::check(slk_update_sigma_points(h.get(), X.data(), SLK_HOST), "slk_update_sigma_points");
            std::vector<_AugmentedState> sig(S);
            for (int i = 0; i < S; ++i) slk_load(sig[i], &X[(std::size_t)i * Nq], nfk, nfkl);
            _AugmentedState ref = sig[0];                        // meanSigmaPoints, :601-627
            int it = 0;
            double norm;
            do {
                slk::Vector md(N);
                for (int i = 0; i < S; ++i) {
                    _AugmentedState d = sig[i] - ref;
                    const slk::Vector dv = d.getVectorizedState();
                    for (int t = 0; t < N; ++t) md[t] += dv[t];
                }
                for (int t = 0; t < N; ++t) md[t] /= (double)S;
                norm = md.norm();
                _AugmentedState delta;
                delta.set(md, nfk, nfkl);
                ref = ref + delta;
            } while (norm > 1e-6 && ++it < 10000);
            slk::Matrix C(N, N);                                 // covSigmaPoints, :654-670
            for (int i = 0; i < S; ++i) {
                _AugmentedState d = sig[i] - ref;
                const slk::Vector dv = d.getVectorizedState();
                for (int c = 0; c < N; ++c) for (int r = 0; r < N; ++r) C(r, c) += dv[r] * dv[c];
            }
            cov_err = (C * 0.5 - Pk).maxAbs();
            _AugmentedState dm = mu_state - ref;
            mean_err = dm.getVectorizedState().norm();
            return cov_err <= 1e-6 && mean_err <= 1e-12;
        }
        void checkSigmaPoints()
        {
            double cov_err = 0, mean_err = 0;
            const bool ok = checkSigmaPoints(cov_err, mean_err);
            if (cov_err > 1e-6) std::cerr << "checkSigmaPoints: max |Pktest - Pk| = " << cov_err << "\n";
            assert(ok);
            (void)ok;
        }

        /**@brief setMeasurement (Usckf.hpp:322-389) */
        template <typename _Measurement, typename _MeasurementNoiseCovariance>
        void setMeasurement(CloningMode mode, _Measurement &z_k_i, _MeasurementNoiseCovariance R)
        {
            assert(z_k_i.size() == R.rows());
            assert(z_k_i.size() == R.cols());
            slk::check(slk_usckf_set_measurement(h.get(), (int)mode, z_k_i.data(), (int)z_k_i.size(), R.data(), SLK_HOST),
                       "slk_usckf_set_measurement");
            if (mode == STATEK) nfk = (int)z_k_i.size();
            else if (mode == STATEK_L) nfkl = (int)z_k_i.size();
            stale = true;
        }

        void cloning(int mode)                                   // Usckf.hpp:391-433
        {
            slk::check(slk_usckf_cloning(h.get(), mode), "slk_usckf_cloning");
            stale = true;
        }

        void setSingleState(const _SingleState &state, int order = STATEK_I)   // :435-455 (STATEK writes statek_l, as in the reference)
        {
            pull();
            switch (order) {
            case STATEK_I: mu_state.statek_i = state; break;
            case STATEK_L: mu_state.statek_l = state; break;
            case STATEK: mu_state.statek_l = state; break;
            default: break;
            }
            upload();
        }
        _SingleState muSingleState(int state = STATEK_I)         // :457-478
        {
            pull();
            switch (state) {
            case STATEK_L: return mu_state.statek_l;
            case STATEK: return mu_state.statek;
            default: return mu_state.statek_i;
            }
        }
        template <class Cov>
        void setPkSingleState(const Cov &Pk_i, int order = STATEK_I)   // :480-491
        {
            if (order != STATEK_I) return;
            pull();
            for (int j = 0; j < 12; ++j) for (int i = 0; i < 12; ++i) Pk(24 + i, 24 + j) = Pk_i.data()[i + 12 * j];
            upload();
        }
        SingleStateCovariance PkSingleState(int state = STATEK_I)       // :493-516
        {
            pull();
            int o = (state == STATEK_L) ? 12 : (state == STATEK) ? 0 : 24;
            return Pk.block(o, o, 12, 12);
        }
        const _AugmentedState &muState() const { pull(); return mu_state; }                 // :518-521
        const AugmentedStateCovariance &PkAugmentedState() const { pull(); return Pk; }     // :523-526

        int status() { int s = 0; slk::check(slk_get_status(h.get(), &s, SLK_HOST), "slk_get_status"); return s; }

        template <typename _ScalarType>
        static bool accept_mahalanobis_distance(const _ScalarType &mahalanobis2, const int dof)   // :794-855
        {
            static const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
            return (dof >= 1 && dof <= 9) ? (mahalanobis2 < thr[dof]) : false;
        }

    private:
        /** library gate: chi-square dof (0 = accept any) */
        template <typename _Measurement, class Cov>
        void run_update(const _Measurement &z, int model, const double *params, const std::vector<double> *Z, const Cov &R, int gate_dof)
        {
            const int m = (int)z.size();
            if (Z) slk::check(slk_update_from_sigma(h.get(), Z->data(), z.data(), m, R.data(), 0, gate_dof, SLK_HOST), "slk_update_from_sigma");
            else slk::check(slk_update(h.get(), model, params, 0, z.data(), m, R.data(), 0, gate_dof, SLK_HOST), "slk_update");
            stale = true;
        }
        /** any other significance test bool(mahalanobis2): evaluated on the host between two launches (:292-294) */
        template <typename _Measurement, class Cov, typename _SignificanceTest>
        void run_update(const _Measurement &z, int model, const double *params, const std::vector<double> *Z, const Cov &R, _SignificanceTest mt)
        {
            const int m = (int)z.size();
            std::vector<double> SI((std::size_t)m * m + m);
            slk::check(slk_update_innovation(h.get(), model, params, 0, Z ? Z->data() : 0, z.data(), m, R.data(), 0, SI.data(), SLK_HOST),
                       "slk_update_innovation");
            slk::Matrix S(m, m);
            slk::Vector innov(m);
            std::copy(SI.begin(), SI.begin() + (std::size_t)m * m, S.data());
            std::copy(SI.begin() + (std::size_t)m * m, SI.end(), innov.data());
            const slk::Matrix d2 = innov.transpose() * (slk::inverse(S) * innov);      // :292
            if (mt(ScalarType(d2[0]))) run_update(z, model, params, Z, R, 0);
        }
    };
} // namespace localization

#endif // __USCKF_HPP_
