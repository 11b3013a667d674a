/**\file Msckf.hpp
 * GPU-backed localization::Msckf: same class template, typedefs, method names and overload set as the
 * reference's src/filters/Msckf.hpp; the numerics run in libslk_hip.so (include/slk.h).
 *
 *   predict(f, Q) :89-95, predict(f, QFn, Nk) :97-189          -> slk_predict / sigma-point path
 *   update(z, h, R) :196-200, update(z, h, R) (matrix) :207-213, update(z, h, R, mt) :220-277
 *                                                              -> slk_update / slk_update_from_sigma /
 *                                                                 slk_update_innovation + slk_update_selected
 *   EKF update(z, h, H, R) :284-290, update(z, h, H, R, mt) :297-349 -> slk_update_ekf
 *   muSingleState / setPkSingleState / getPkSingleState / muState (const and non-const) / getPk / setPk :351-395
 *   checkSigmaPoints :819-839 -> slk_check_sigma_points;  accept_mahalanobis_distance :844-905
 *
 * `f` / `h` may be a registered model object (slk::DeltaPoseModel, slk::FeatureProjectionModel, ...)
 * -- evaluated on the GPU -- or ANY callable like the reference's boost::bind functors
 * (test/MsckfUnitTest.cpp:200-205): then the sigma points are drawn on the GPU, mapped by the
 * callable on the host and the step is finished on the GPU (same results).
 * `Q` / `R` may be matrices or nullary functors returning one (boost::bind(ukfom::id<Cov>, Q), :94).
 * `mt` may be the class's own accept_mahalanobis_distance (runs inside the kernel), a bool (true = that
 * gate, false = accept all) or any callable bool(d2, dof): then the innovation and its covariance come
 * back from the GPU, the reference's removeOutliers loop (:723-754) runs here with the callable and the
 * surviving rows go back to the GPU.
 *
 * Host mirror: mu_state / Pk are kept lazily in sync with the device (the reference hands out references,
 * :376-389); the non-const muState() marks the mirror as edited, so that clone push / pop followed by
 * setPk() (:381-395) reaches the device before the next filter call.
 * EKF update: needs m >= N rows and an even m (reduceDimension compresses 2-row blocks to N rows, :791-816;
 * the reference reads out of range otherwise) -- std::runtime_error instead of undefined behaviour.
 */
#ifndef _MSCKF_HPP_
#define _MSCKF_HPP_

#include <algorithm>
#include <cassert>
#include <iostream>
#include <stdexcept>
#include <type_traits>
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
        mutable _MultiState mu_state;       /** host mirror of the mean **/
        mutable MultiStateCovariance Pk;    /** host mirror of the covariance **/
        mutable bool mean_stale, cov_stale; /** the device holds something newer than the mirror **/
        mutable bool mean_dirty, cov_dirty; /** the mirror holds something newer than the device **/
        mutable slk::Handle h;
        unsigned int last_outliers;
        int host_status = 0;                /** status bits decided on the host (EKF update behind a caller-side test) **/

        /** push host-side edits (muState(), muSingleState(state), setPk, setPkSingleState) to the device */
        void sync_device() const
        {
            if (!mean_dirty && !cov_dirty) return;
            const int k = (int)mu_state.sensorsk.size(), N = 12 + 6 * k;
            if (N != h.N()) {
                // the window changed through muState().sensorsk (Msckf.hpp:381-395): needs the matching covariance
                if (Pk.rows() != N || Pk.cols() != N || cov_stale)
                    throw std::logic_error("Msckf: muState() changed the number of sensor poses; call setPk() with the matching covariance");
                slk::check(slk_msckf_resize(h.get(), k), "slk_msckf_resize");
                mean_dirty = cov_dirty = true;
            } else if (cov_dirty && (Pk.rows() != N || Pk.cols() != N)) {
                throw std::logic_error("Msckf: setPk() with a covariance that does not match the number of sensor poses");
            }
            std::vector<double> m;
            if (mean_dirty) { m.resize(h.Nq()); slk_store(mu_state, m.data()); }
            slk::check(slk_set_state(h.get(), mean_dirty ? m.data() : 0, cov_dirty ? Pk.data() : 0, SLK_HOST), "slk_set_state");
            mean_dirty = cov_dirty = false;
        }
        void pull_mean() const
        {
            if (!mean_stale) return;
            std::vector<double> m(h.Nq());
            slk::check(slk_get_state(h.get(), m.data(), 0, SLK_HOST), "slk_get_state");
            mu_state.sensorsk.resize((h.N() - 12) / 6);
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
        void device_changed() { mean_stale = cov_stale = true; }
        unsigned int finish_update()
        {
            slk::check(slk_get_outliers(h.get(), &last_outliers, SLK_HOST), "slk_get_outliers");
            device_changed();
            return last_outliers;
        }

        /** registered models -> (id, parameter pointer) for the C ABI; anything else is a host functor */
        static int model_id(const slk::FeatureProjectionModel &) { return SLK_MM_FEATURE_PROJ; }
        static int model_id(const slk::PosePositionModel &) { return SLK_MM_POSE_POSITION; }
        static const double *model_params(const slk::FeatureProjectionModel &m) { return m.params.data(); }
        static const double *model_params(const slk::PosePositionModel &m) { return &m.pose; }

        /** Z = h(X) over the sigma points of the full state on the host (std::transform(X, Z, h), Msckf.hpp:231-232) */
        template <typename _Measurement, typename _MeasurementModel>
        std::vector<double> map_sigma_points(_MeasurementModel &hfun, int m)
        {
            const int N = h.N(), Nq = h.Nq(), S = 2 * N + 1;
            std::vector<double> X((std::size_t)S * Nq), Z((std::size_t)S * m);
            slk::check(slk_update_sigma_points(h.get(), X.data(), SLK_HOST), "slk_update_sigma_points");
            pull_mean();
            _MultiState x(mu_state);
            for (int i = 0; i < S; ++i) {
                slk_load(x, &X[(std::size_t)i * Nq]);
                const _Measurement zi = hfun(x);
                assert((int)zi.size() == m);
                for (int r = 0; r < m; ++r) Z[(std::size_t)i * m + r] = zi[r];
            }
            return Z;
        }

        /** removeOutliers (Msckf.hpp:723-754) with an arbitrary significance test on the innovation / covariance the
         *  GPU computed: returns { surviving rows, outliers, row indices } for slk_update_selected.  The second erase
         *  of a rejected block acts on the already shifted arrays, as in the reference (:741-744). */
        template <typename _SignificanceTest>
        static std::vector<int> select_rows(const std::vector<double> &SI, int m, _SignificanceTest &mt)
        {
            const double *S = SI.data(), *innov = SI.data() + (std::size_t)m * m;
            std::vector<int> idx(m);
            for (int r = 0; r < m; ++r) idx[r] = r;
            int cnt = m, nout = 0, i = 0;
            while (i < cnt / 2) {
                const int p = idx[2 * i], q = idx[2 * i + 1];
                const double s00 = S[p + m * p], s01 = S[p + m * q], s10 = S[q + m * p], s11 = S[q + m * q];
                const double det = s00 * s11 - s01 * s10, r0 = innov[p], r1 = innov[q];
                const ScalarType d2 = (r0 * (s11 * r0 - s01 * r1) + r1 * (s00 * r1 - s10 * r0)) / det;
                if (!mt(d2, 2)) {
                    for (int rep = 0; rep < 2; ++rep) {                    // removeRow semantics, :688-697
                        const int pos = 2 * i + rep, numRows = cnt - 1;
                        if (pos < numRows) for (int w = pos; w < numRows; ++w) idx[w] = idx[w + 1];
                        cnt = numRows;
                    }
                    ++nout;
                } else {
                    ++i;
                }
            }
            std::vector<int> rs(m + 2, 0);
            rs[0] = cnt; rs[1] = nout;
            for (int r = 0; r < cnt; ++r) rs[2 + r] = idx[r];
            return rs;
        }

        /** the common tail of every UKF update overload: model = registered id or SLK_MODEL_EXTERNAL with Z */
        template <typename _Measurement, class Cov, typename _SignificanceTest>
        unsigned int run_update(const _Measurement &z, int model, const double *params, const std::vector<double> *Z,
                                const Cov &R, int kind, _SignificanceTest &mt)
        {
            const int m = (int)z.size();
            const double *Zp = Z ? Z->data() : 0;
            if (kind < 2) {
                if (Zp) slk::check(slk_update_from_sigma(h.get(), Zp, z.data(), m, R.data(), 0, kind, SLK_HOST), "slk_update_from_sigma");
                else slk::check(slk_update(h.get(), model, params, 0, z.data(), m, R.data(), 0, kind, SLK_HOST), "slk_update");
                return finish_update();
            }
            std::vector<double> SI((std::size_t)m * m + m);
            slk::check(slk_update_innovation(h.get(), model, params, 0, Zp, z.data(), m, R.data(), 0, SI.data(), SLK_HOST),
                       "slk_update_innovation");
            const std::vector<int> rs = select_rows(SI, m, mt);
            slk::check(slk_update_selected(h.get(), model, params, 0, Zp, z.data(), m, R.data(), 0, rs.data(), SLK_HOST),
                       "slk_update_selected");
            return finish_update();
        }
        struct NoTest { bool operator()(const ScalarType &, int) const { return true; } };

        template <class T> static int kind_of(const T &) { return 2; }
        static int kind_of(const bool &g) { return g ? 1 : 0; }
        static int kind_of(bool (*const &fn)(const ScalarType &, const int)) { return fn == &self::template accept_mahalanobis_distance<ScalarType> ? 1 : 2; }
        template <class T> static bool call_mt(T &mt, const ScalarType &d2, int dof) { return mt(d2, dof); }
        static bool call_mt(bool &, const ScalarType &, int) { return true; }
        template <class T> struct MtCall { T &mt; bool operator()(const ScalarType &d2, int dof) { return call_mt(mt, d2, dof); } };

    public:
        /**@brief Constructor (Msckf.hpp:80-85) */
        template <class Cov>
        Msckf(const _MultiState &state, const Cov &P0)
            : mu_state(state), mean_stale(false), cov_stale(false), mean_dirty(true), cov_dirty(true), last_outliers(0)
        {
            Pk.resize(P0.rows(), P0.cols());
            std::copy(P0.data(), P0.data() + (std::size_t)P0.rows() * P0.cols(), Pk.data());
            h.create(SLK_MSCKF, 1, (int)state.sensorsk.size(), 0, 0);
            sync_device();
        }

        /**@brief Filter prediction step, registered process model on the GPU (Msckf.hpp:89-95) */
        template <class QArg>
        void predict(const slk::DeltaPoseModel &f, const QArg &Q) { predict_registered(SLK_PM_DELTA_POSE, f.u, Q); }
        template <class QArg>
        void predict(const slk::ConstVelocityModel &f, const QArg &Q) { predict_registered(SLK_PM_CONST_VELOCITY, f.u, Q); }
        /** dead reckoning fused into the prediction (src/core/DeadReckon.hpp:129-239 -> the delta-pose model) */
        template <class QArg>
        void predict(const slk::DeadReckonModel &f, const QArg &Q) { predict_registered(SLK_PM_DEAD_RECKON, f.u, Q); }
        /**@brief Filter prediction step with an arbitrary process model functor f: _SingleState -> _SingleState; Q is
         * a matrix (Msckf.hpp:89-95) or a nullary functor returning one (:97-98) */
        template <typename _ProcessModel, class QArg>
        void predict(_ProcessModel f, const QArg &Q)
        {
            sync_device();
            const auto &Qm = slk::noise_matrix(Q, 0);
            std::vector<double> X(25 * 13), Y(25 * 13);
            slk::check(slk_predict_sigma_points(h.get(), X.data(), SLK_HOST), "slk_predict_sigma_points");
            for (int i = 0; i < 25; ++i) {                       // std::transform(X, X, f), Msckf.hpp:125
                _SingleState x, y;
                slk_load(x, &X[13 * i]);
                y = f(x);
                slk_store(y, &Y[13 * i]);
            }
            slk::check(slk_predict_from_sigma(h.get(), Y.data(), Qm.data(), 0, SLK_HOST), "slk_predict_from_sigma");
            device_changed();
        }
        /**@brief predict(f, Q, Nk) (Msckf.hpp:97-98): the null-space matrix is not used by the reference either
         * (its only use, :171-182, is commented out) */
        template <typename _ProcessModel, typename _ProcessNoiseCovariance, typename _NullSpaceMatrix>
        void predict(_ProcessModel f, _ProcessNoiseCovariance Q, _NullSpaceMatrix /*Nk*/) { predict(f, Q); }

        /**@brief UKF update (Msckf.hpp:196-213): default significance test = accept_mahalanobis_distance; returns the
         * outlier count (:276).  h = registered model or any callable _MultiState -> vector. */
        template <typename _Measurement, typename _MeasurementModel, typename _MeasurementNoiseCovariance>
        unsigned int update(const _Measurement &z, _MeasurementModel hfun, const _MeasurementNoiseCovariance &R)
        {
            return update(z, hfun, R, true);
        }
        /**@brief UKF update with a significance test (Msckf.hpp:220-277), registered measurement model */
        template <typename _Measurement, class Cov, typename _SignificanceTest>
        typename std::enable_if<!slk::is_matrix_like<_SignificanceTest>::value, unsigned int>::type
        update(const _Measurement &z, const slk::FeatureProjectionModel &hmodel, const Cov &R, _SignificanceTest mt)
        {
            sync_device();
            MtCall<_SignificanceTest> call = {mt};
            return run_update(z, model_id(hmodel), model_params(hmodel), 0, slk::noise_matrix(R, 0), kind_of(mt), call);
        }
        template <typename _Measurement, class Cov, typename _SignificanceTest>
        typename std::enable_if<!slk::is_matrix_like<_SignificanceTest>::value, unsigned int>::type
        update(const _Measurement &z, const slk::PosePositionModel &hmodel, const Cov &R, _SignificanceTest mt)
        {
            sync_device();
            MtCall<_SignificanceTest> call = {mt};
            return run_update(z, model_id(hmodel), model_params(hmodel), 0, slk::noise_matrix(R, 0), kind_of(mt), call);
        }
        /**@brief UKF update with a significance test (Msckf.hpp:220-277), arbitrary measurement functor */
        template <typename _Measurement, typename _MeasurementModel, class Cov, typename _SignificanceTest>
        typename std::enable_if<!slk::is_matrix_like<_SignificanceTest>::value, unsigned int>::type
        update(const _Measurement &z, _MeasurementModel hfun, const Cov &R, _SignificanceTest mt)
        {
            sync_device();
            const std::vector<double> Z = map_sigma_points<_Measurement>(hfun, (int)z.size());
            MtCall<_SignificanceTest> call = {mt};
            return run_update(z, SLK_MODEL_EXTERNAL, 0, &Z, slk::noise_matrix(R, 0), kind_of(mt), call);
        }

        /**@brief EKF update (Msckf.hpp:284-290): h(mu_state, H) returns the expected measurement and fills the
         * Jacobian H (m x N) at the current mean, exactly like the reference's functor (:310); the gate, the Householder
         * compression (reduceDimension) and the gain run on the GPU.  Returns the outlier count (:348). */
        template <typename _Measurement, typename _MeasurementModel, class Jac, class Cov>
        typename std::enable_if<slk::is_matrix_like<Jac>::value && slk::is_matrix_like<Cov>::value, unsigned int>::type
        update(const _Measurement &z, _MeasurementModel hfun, Jac &H, Cov &R)
        {
            return update(z, hfun, H, R, true);
        }
        /**@brief EKF update with a significance test (Msckf.hpp:297-349).  A custom test runs the reference's loop
         * (:756-789) here on the information matrix (H P H^T + R)^-1, erases the rejected rows of z, h(mu), H and R
         * and hands the rest to the GPU ungated -- which is what the reference's removeOutliers leaves behind. */
        template <typename _Measurement, typename _MeasurementModel, class Jac, class Cov, typename _SignificanceTest>
        unsigned int update(const _Measurement &z, _MeasurementModel hfun, Jac &H, Cov &R, _SignificanceTest mt)
        {
            sync_device();
            pull_mean();
            const _Measurement mean_z = hfun(mu_state, H);
            const int m = (int)z.size(), N = h.N(), kind = kind_of(mt);
            if (kind < 2) {
                slk::check(slk_update_ekf(h.get(), z.data(), mean_z.data(), H.data(), m, R.data(), 0, kind, SLK_HOST), "slk_update_ekf");
                return finish_update();
            }
            pull_cov();
            slk::Matrix Hm(m, N), Rm(m, m);
            std::copy(H.data(), H.data() + (std::size_t)m * N, Hm.data());
            std::copy(R.data(), R.data() + (std::size_t)m * m, Rm.data());
            const slk::Matrix info = slk::inverse(Hm * Pk * Hm.transpose() + Rm);
            std::vector<int> idx(m);
            for (int r = 0; r < m; ++r) idx[r] = r;
            std::vector<double> innov(m);
            for (int r = 0; r < m; ++r) innov[r] = z[r] - mean_z[r];
            int cnt = m, i = 0;
            unsigned int nout = 0;
            while (i < cnt / 2) {
                // the information matrix is NOT shrunk (:772): block i of it, rows of the shrunk innovation
                const double a = info(2 * i, 2 * i), b = info(2 * i, 2 * i + 1), c = info(2 * i + 1, 2 * i), d = info(2 * i + 1, 2 * i + 1);
                const double r0 = innov[idx[2 * i]], r1 = innov[idx[2 * i + 1]];
                const ScalarType d2 = r0 * (a * r0 + b * r1) + r1 * (c * r0 + d * r1);
                if (!call_mt(mt, d2, 2)) {
                    for (int rep = 0; rep < 2; ++rep) {
                        const int pos = 2 * i + rep, numRows = cnt - 1;
                        if (pos < numRows) for (int w = pos; w < numRows; ++w) idx[w] = idx[w + 1];
                        cnt = numRows;
                    }
                    ++nout;
                } else {
                    ++i;
                }
            }
            last_outliers = nout;
            if (cnt == 0) return nout;                       // every block rejected: nothing is applied (Msckf.hpp:320)
            if (cnt < N) {
                // fewer rows than states survive: the reference's reduceDimension would index R.block(0, 0, N, N) out of
                // range (:806).  Same outcome as with the built-in gate: the update is skipped, SLK_ST_EKF_ROWS is reported
                host_status |= SLK_ST_EKF_ROWS;
                return nout;
            }
            std::vector<double> zs(cnt), zm(cnt), Hs((std::size_t)cnt * N), Rs((std::size_t)cnt * cnt);
            for (int r = 0; r < cnt; ++r) {
                zs[r] = z[idx[r]]; zm[r] = mean_z[idx[r]];
                for (int j = 0; j < N; ++j) Hs[r + (std::size_t)j * cnt] = Hm(idx[r], j);
                for (int j = 0; j < cnt; ++j) Rs[r + (std::size_t)j * cnt] = Rm(idx[r], idx[j]);
            }
            slk::check(slk_update_ekf(h.get(), zs.data(), zm.data(), Hs.data(), cnt, Rs.data(), 0, 0, SLK_HOST), "slk_update_ekf");
            device_changed();
            last_outliers = nout;
            return nout;
        }

        void muSingleState(const _SingleState &state)            // Msckf.hpp:351-354
        {
            pull_mean();
            mu_state.statek = state;
            mean_dirty = true;
        }
        _SingleState muSingleState() { pull_mean(); return mu_state.statek; }          // :356-361
        template <class Cov>
        void setPkSingleState(const Cov &Pk_i)                   // :363-366
        {
            pull_cov();
            for (int j = 0; j < 12; ++j) for (int i = 0; i < 12; ++i) Pk(i, j) = Pk_i.data()[i + 12 * j];
            cov_dirty = true;
        }
        SingleStateCovariance getPkSingleState() { pull_cov(); return Pk.block(0, 0, 12, 12); }   // :368-374
        const _MultiState &muState() const { pull_mean(); return mu_state; }            // :376-379
        /** non-const access (:381-384): callers push / pop sensor poses through it and then call setPk (:391-395);
         *  the edit reaches the device before the next filter call */
        _MultiState &muState() { pull_mean(); mean_dirty = true; return mu_state; }
        const MultiStateCovariance &getPk() const { pull_cov(); return Pk; }            // :386-389
        template <class Cov>
        void setPk(const Cov &Pk_i)                                                     // :391-395
        {
            Pk.resize(Pk_i.rows(), Pk_i.cols());
            std::copy(Pk_i.data(), Pk_i.data() + (std::size_t)Pk_i.rows() * Pk_i.cols(), Pk.data());
            cov_stale = false;
            cov_dirty = true;
        }
        /** Replace mean and covariance together (push / pop of clones plus setPk in one call) */
        template <class Cov>
        void setState(const _MultiState &state, const Cov &Pk_i)
        {
            mu_state = state;
            mean_stale = false;
            mean_dirty = true;
            setPk(Pk_i);
            sync_device();
        }

        /**@brief checkSigmaPoints (Msckf.hpp:819-839) on the device: the covariance of the sigma points of
         * (mu_state, Pk) must reproduce Pk (1e-6) and their mean mu_state.  Asserts like the reference; the overload
         * with arguments reports the two errors instead. */
        void checkSigmaPoints()
        {
            double cov_err = 0, mean_err = 0;
            const bool ok = checkSigmaPoints(cov_err, mean_err);
            if (cov_err > 1e-6) { pull_cov(); std::cerr << "checkSigmaPoints: max |Pktest - Pk| = " << cov_err << "\n\n" << Pk << "\n"; }
            if (mean_err > 1e-12) std::cout << "norm:" << (mean_err > 0. ? ">" : "=") << std::endl;
            assert(ok);
            (void)ok;
        }
        bool checkSigmaPoints(double &cov_err, double &mean_err)
        {
            sync_device();
            slk::check(slk_check_sigma_points(h.get(), &cov_err, &mean_err, SLK_HOST), "slk_check_sigma_points");
            return cov_err <= 1e-6 && mean_err <= 1e-12;
        }

        /** per-filter numerical status bits of include/slk.h (the reference reports nothing) */
        int status() { int s = 0; slk::check(slk_get_status(h.get(), &s, SLK_HOST), "slk_get_status"); return s | host_status; }

        /** chi-square gate of the reference (:844-905) */
        template <typename _ScalarType>
        static bool accept_mahalanobis_distance(const _ScalarType &mahalanobis2, const int dof)
        {
            static const double thr[10] = {0, 3.84, 5.99, 7.81, 9.49, 11.07, 12.59, 14.07, 15.51, 16.92};
            if (dof >= 1 && dof <= 9) return mahalanobis2 < thr[dof];
            std::cerr << "mahalanobis distance not implemented for dof " << dof << std::endl;   // :899-901
            return false;
        }

    private:
        template <class QArg>
        void predict_registered(int model, const double *u, const QArg &Q)
        {
            sync_device();
            const auto &Qm = slk::noise_matrix(Q, 0);
            slk::check(slk_predict(h.get(), model, u, 0, Qm.data(), 0, SLK_HOST), "slk_predict");
            device_changed();
        }
    };
} // namespace localization

#endif // __MSCKF_HPP_
