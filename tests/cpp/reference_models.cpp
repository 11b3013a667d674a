// GPU: the reference's OWN model functions -- copied unchanged from test/MsckfUnitTest.cpp:32-47 and
// test/UsckfUnitTest.cpp:34-49, :62-86 -- driven through the GPU-backed facade with the reference's call forms
// (boost::bind -> std::bind): predict(f, Q), predict(f, QFn, Nk), update(z, h, R), update(z, h, R, mt) with the class's
// own accept_mahalanobis_distance and with an arbitrary callable, checkSigmaPoints(), the non-const muState() +
// setPk() window edit.  Prints "name rows cols v0 v1 ..." lines (column-major) for tests/test_gpu_facade.py.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <functional>

#include <localization/filters/Msckf.hpp>
#include <localization/filters/Usckf.hpp>
#include "eigen_mtk_names.hpp"

using namespace localization;

/** Wrap the states (test/MsckfUnitTest.cpp:24-28, test/UsckfUnitTest.cpp:26-30) **/
typedef localization::MtkWrap<localization::State> WSingleState;
typedef localization::MtkDynamicWrap< localization::MultiState<localization::State, localization::SensorState> > WMultiState;
typedef localization::Msckf<WMultiState, WSingleState> MultiStateFilter;
typedef ::MTK::vect<Eigen::Dynamic, double> MeasurementType;
typedef localization::MtkMultiStateWrap<localization::AugmentedState<Eigen::Dynamic> > WAugmentedState;
typedef localization::Usckf<WAugmentedState, WSingleState > StateFilterDynamic;

// ---- verbatim from test/MsckfUnitTest.cpp:32-47 ------------------------------------------------------------
/** Process model when accumulating delta poses **/
WSingleState processModel (const WSingleState &state,  const Eigen::Vector3d &delta_position, const localization::SO3 &delta_orientation,
                            const Eigen::Vector3d &velocity, const Eigen::Vector3d &angular_velocity)
{
    WSingleState s2; /** Propagated state */

    /** Apply Rotation **/
    s2.orient = state.orient * delta_orientation;
    s2.angvelo = angular_velocity;

    /** Apply Translation **/
    s2.pos = state.pos + (s2.orient * delta_position);
    s2.velo = velocity;

    return s2;
};

// ---- verbatim from test/UsckfUnitTest.cpp:34-60 -------------------------------------------------------------
WSingleState processModel (const WSingleState &state,  const Eigen::Vector3d &velocity, const Eigen::Vector3d &angular_velocity, double dt)
{
    WSingleState s2; /** Propagated state */

    /** Apply Rotation **/
    Eigen::Vector3d scaled_axis = angular_velocity * dt;
    localization::SO3 rot = localization::SO3::exp (scaled_axis);
    s2.orient = state.orient * rot ;
    s2.angvelo = angular_velocity;

    /** Apply Translation **/
    s2.velo = velocity;
    s2.pos = state.pos + state.velo * dt;

    return s2;
};

StateFilterDynamic::SingleStateCovariance processNoiseCov (double dt)
{
    StateFilterDynamic::SingleStateCovariance cov = StateFilterDynamic::SingleStateCovariance::Zero(12, 12);   // (fixed-size Zero() in the reference)
    MTK::setDiagonal (cov, &WSingleState::pos, 0.1 * dt);
    MTK::setDiagonal (cov, &WSingleState::orient, 0.1 * dt);
    MTK::setDiagonal (cov, &WSingleState::velo,  0.1 * dt);
    MTK::setDiagonal (cov, &WSingleState::angvelo,  0.1 * dt);

    return cov ;
};

// ---- verbatim from test/UsckfUnitTest.cpp:62-86 -------------------------------------------------------------
localization::AugmentedState<Eigen::Dynamic>::MeasurementType measurementModelVO (const WAugmentedState &wastate)
{
    WSingleState delta_state, statek, statek_i; /** Propagated state */
    localization::AugmentedState<Eigen::Dynamic>::MeasurementType z_hat;
    z_hat = wastate.featuresk;
    statek = wastate.statek;
    statek_i = wastate.statek_i;

    delta_state = statek - statek_i;
    Eigen::Affine3d delta_transform (delta_state.orient);
    delta_transform.translation() = delta_state.pos;

    for (register unsigned int i = 0; i < z_hat.size(); i+=3)
    {
        Eigen::Vector3d coord;
        coord<<wastate.featuresk[i], wastate.featuresk[i+1], wastate.featuresk[i+2];
        coord = delta_transform * coord;
        z_hat[i] = coord[0];
        z_hat[i+1] = coord[1];
        z_hat[i+2] = coord[2];
    }
//    std::cout<<"z_hat "<<z_hat<<"\n";

    return z_hat;
};
// ---------------------------------------------------------------------------------------------------------------

typedef WSingleState (*DeltaPoseFn)(const WSingleState &, const Eigen::Vector3d &, const localization::SO3 &, const Eigen::Vector3d &, const Eigen::Vector3d &);
typedef WSingleState (*ConstVelFn)(const WSingleState &, const Eigen::Vector3d &, const Eigen::Vector3d &, double);

static void dump(const char *name, const slk::Matrix &m)
{
    std::printf("%s %d %d", name, m.rows(), m.cols());
    for (int i = 0; i < m.size(); ++i) std::printf(" %.17g", m.data()[i]);
    std::printf("\n");
}
template <class S>
static void dump_mean(const char *name, const S &s, int nq)
{
    std::vector<double> v(nq);
    slk_store(s, v.data());
    std::printf("%s %d 1", name, nq);
    for (int i = 0; i < nq; ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}
static void dump_scalar(const char *name, double v) { std::printf("%s 1 1 %.17g\n", name, v); }

struct CountingChi2        // an arbitrary significance test: same decision as the library's gate, counts its calls
{
    int *calls;
    bool operator()(const double &d2, int dof) const { ++*calls; return dof == 2 && d2 < 5.99; }
};

static void msckf(int variant)
{
    using namespace std::placeholders;
    const unsigned int number_sensor_poses = 4;                                   // MsckfUnitTest.cpp:154
    WMultiState statek_0;
    statek_0.sensorsk.resize(number_sensor_poses);
    const int N = WSingleState::DOF + WMultiState::SENSOR_DOF * number_sensor_poses;
    slk::Matrix Pk_0 = 0.025 * slk::Matrix::Identity(N, N);                       // square SPD Pk_0
    Eigen::Vector3d position, velocity, angular_velocity;
    position << 0.1, 0.1, 0.1;                                                    // :164
    localization::SO3 orientation = localization::SO3::exp(Eigen::Vector3d(0, 0, 1.0 * D2R)) * localization::SO3::exp(Eigen::Vector3d(0, 1.0 * D2R, 0))
                                    * localization::SO3::exp(Eigen::Vector3d(1.0 * D2R, 0, 0));   // :165-168
    velocity << 0.1, 0.1, 0.1;
    angular_velocity << 0.1, 0.1, 0.1;
    typedef MultiStateFilter::SingleStateCovariance SingleStateCovariance;
    SingleStateCovariance cov_process = 0.01 * SingleStateCovariance::Identity(12, 12);           // :173-177
    MultiStateFilter filter(statek_0, Pk_0);
    char nm[96];
    const char *tag = variant == 0 ? "a" : (variant == 1 ? "b" : "c");
    for (int i = 0; i < 2; ++i) {                                                                 // :196-206
        auto f = std::bind(static_cast<DeltaPoseFn>(processModel), _1, position, orientation, velocity, angular_velocity);
        if (variant == 0) filter.predict(f, cov_process);                                         // predict(f, Q), Msckf.hpp:89-95
        else filter.predict(f, [&]() { return cov_process; }, slk::Matrix(12, 4));                // predict(f, QFn, Nk), :97-98
        std::snprintf(nm, sizeof nm, "ref_msckf_%s_pred%d_mean", tag, i); dump_mean(nm, filter.muState(), 13 + 7 * 4);
        std::snprintf(nm, sizeof nm, "ref_msckf_%s_pred%d_P", tag, i); dump(nm, filter.getPk());
    }
    double ce = 0, me = 0;
    const bool ok = filter.checkSigmaPoints(ce, me);                                              // :819-839
    filter.checkSigmaPoints();                                                                    // the asserting form
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_check_ok", tag); dump_scalar(nm, ok);
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_check_cov", tag); dump_scalar(nm, ce);
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_check_mean", tag); dump_scalar(nm, me);
    // update on the golden scenario's features (tests/golden/make_golden.py), third one a gross outlier
    const int nf = 4;
    slk::FeatureProjectionModel h;
    slk::Vector z(2 * nf);
    for (int j = 0; j < nf; ++j) {
        h.add(0.5 * (j - 1.5), 0.3 * (1.5 - j), 5.0 + j, (j % 4) + 1);
        z[2 * j] = 0.1 * (j - 1.0) * 0.5;
        z[2 * j + 1] = 0.05 * (j + 0.5) * 0.5;
    }
    z[4] += 3.0;
    slk::Matrix R = 0.01 * slk::Matrix::Identity(2 * nf, 2 * nf);
    unsigned int outliers;
    int calls = 0;
    if (variant == 0) outliers = filter.update(z, h, R);                                          // update(z, h, R), :196-200
    else if (variant == 1) outliers = filter.update(z, h, R, MultiStateFilter::accept_mahalanobis_distance<double>);   // :220-223
    else { CountingChi2 mt = {&calls}; outliers = filter.update(z, h, R, mt); }                   // arbitrary mt
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_upd_mean", tag); dump_mean(nm, filter.muState(), 13 + 7 * 4);
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_upd_P", tag); dump(nm, filter.getPk());
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_outliers", tag); dump_scalar(nm, outliers);
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_mt_calls", tag); dump_scalar(nm, calls);
    std::snprintf(nm, sizeof nm, "ref_msckf_%s_status", tag); dump_scalar(nm, filter.status());
}

// Msckf.hpp:381-395: push a sensor pose through the non-const muState(), then setPk with the matching covariance
static void msckf_window_edit()
{
    using namespace std::placeholders;
    WMultiState s0;
    s0.sensorsk.resize(1);
    s0.statek.pos << 1, 2, 3;
    slk::Matrix P0 = 0.02 * slk::Matrix::Identity(18, 18);
    MultiStateFilter filter(s0, P0);
    Eigen::Vector3d dp(0.1, 0.0, 0.05), v(0.1, 0.1, 0.1), w(0.0, 0.0, 0.1);
    localization::SO3 dq = localization::SO3::exp(Eigen::Vector3d(0, 0, 0.02));
    slk::Matrix Q = 0.01 * slk::Matrix::Identity(12, 12);
    auto f = std::bind(static_cast<DeltaPoseFn>(processModel), _1, dp, dq, v, w);
    filter.predict(f, Q);
    // window edit
    localization::SensorState clone(filter.muState().statek.pos, filter.muState().statek.orient);
    filter.muState().sensorsk.push_back(clone);
    slk::Matrix Pold = filter.getPk(), Pnew(24, 24);
    Pnew.setBlock(0, 0, Pold);
    for (int i = 0; i < 6; ++i) Pnew(18 + i, 18 + i) = 0.03;
    filter.setPk(Pnew);
    filter.predict(f, Q);
    dump_mean("ref_window_mean", filter.muState(), 13 + 7 * 2);
    dump("ref_window_P", filter.getPk());
    // the same from a fresh filter that starts with the edited state
    WMultiState s1 = filter.muState();               // (after the second predict: rebuild the pre-predict state instead)
    (void)s1;
    MultiStateFilter g(s0, P0);
    g.predict(f, Q);
    WMultiState s2 = g.muState();
    s2.sensorsk.push_back(localization::SensorState(s2.statek.pos, s2.statek.orient));
    MultiStateFilter g2(s2, Pnew);
    g2.predict(f, Q);
    dump_mean("ref_window_fresh_mean", g2.muState(), 13 + 7 * 2);
    dump("ref_window_fresh_P", g2.getPk());
}

// Usckf on a well-posed (SPD) state of the unit-test shape: N = 36 + 3 + 9; closed-form pattern the Python test rebuilds
static void usckf(int variant)
{
    using namespace std::placeholders;
    const int nfk = 3, nfkl = 9, N = 48;
    WAugmentedState x0;
    State *st[3] = {&x0.statek, &x0.statek_l, &x0.statek_i};
    for (int b = 0; b < 3; ++b) {
        st[b]->pos << 0.5 + 0.1 * b, -0.3 + 0.05 * b, 1.0 - 0.2 * b;
        st[b]->orient = localization::SO3::exp(Eigen::Vector3d(0.05 * (b + 1), -0.04 * b, 0.03 + 0.02 * b));
        st[b]->velo << 0.3, -0.1 * b, 0.2;
        st[b]->angvelo << 0.01 * b, 0.02, -0.01;
    }
    x0.featuresk.resize(nfk); x0.featuresk_l.resize(nfkl);
    for (int i = 0; i < nfk; ++i) x0.featuresk[i] = 2.0 + 0.5 * i;
    for (int i = 0; i < nfkl; ++i) x0.featuresk_l[i] = 1.0 + 0.25 * i;
    slk::Matrix A(N, N), P(N, N);
    for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) A(i, j) = 0.007 * (((i * 7 + j * 13) % 11) - 5.0) / 5.0;
    P = A * A.transpose();
    for (int i = 0; i < N; ++i) P(i, i) += 0.0025;
    StateFilterDynamic filter(x0, P);
    const double dt = 0.01;
    Eigen::Vector3d velo, angular_velo;
    velo << 1.0, 0.1, -0.2;
    angular_velo << (10.00*localization::D2R), (-5.00*localization::D2R), (8.00*localization::D2R);
    StateFilterDynamic::SingleStateCovariance myCov = processNoiseCov(dt);
    slk::Vector z(3);
    z << 2.05, 2.45, 3.1;
    slk::Matrix R = 0.01 * slk::Matrix::Identity(3, 3);
    int calls = 0;
    const char *tag = variant == 0 ? "model" : (variant == 1 ? "functor" : "mt");
    char nm[96];
    for (int i = 0; i < 2; ++i) {
        if (variant == 0) {
            filter.predict(slk::ConstVelocityModel(velo, angular_velo, dt), myCov);
            filter.update(z, slk::VoRelativeModel(), R);
        } else if (variant == 1) {
            filter.predict(std::bind(static_cast<ConstVelFn>(processModel), _1 , velo , angular_velo, dt), myCov);   // UsckfUnitTest.cpp:246
            filter.update(z, std::bind(measurementModelVO, _1), R);                                                   // :284
        } else {
            filter.predict(std::bind(static_cast<ConstVelFn>(processModel), _1 , velo , angular_velo, dt), [&]() { return myCov; });   // predict(f, QFn), Usckf.hpp:113-114
            filter.update(z, std::bind(measurementModelVO, _1), [&]() { return R; },
                          [&](const double &d2) { ++calls; return d2 < 1e9; });                                       // update(z, h, RFn, mt), :260-263
        }
    }
    double ce = 0, me = 0;
    const bool ok = filter.checkSigmaPoints(ce, me);                                                                 // Usckf.hpp:769-789
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_mean", tag); dump_mean(nm, filter.muState(), 39 + nfk + nfkl);
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_P", tag); dump(nm, filter.PkAugmentedState());
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_status", tag); dump_scalar(nm, filter.status());
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_check_ok", tag); dump_scalar(nm, ok);
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_check_cov", tag); dump_scalar(nm, ce);
    std::snprintf(nm, sizeof nm, "ref_usckf_%s_mt_calls", tag); dump_scalar(nm, calls);
    if (variant == 2) {       // a rejecting test leaves the filter as it is (:294: `if (mt(mahalanobis2))`)
        slk::Matrix before = filter.PkAugmentedState();
        filter.update(z, std::bind(measurementModelVO, _1), R, [](const double &) { return false; });
        dump_scalar("ref_usckf_rejected_unchanged", (filter.PkAugmentedState() - before).maxAbs() == 0.0);
    }
}

int main()
{
    for (int v = 0; v < 3; ++v) msckf(v);
    msckf_window_edit();
    for (int v = 0; v < 3; ++v) usckf(v);
    return 0;
}
