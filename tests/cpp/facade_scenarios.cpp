// The reference's unit-test scenarios written against the GPU-backed header facade, the way a
// Rock/orogen task would use the library:
//   * MSCKF         test/MsckfUnitTest.cpp:151-206 (2 x predict with the delta-pose model, then an update)
//   * USCKF_DYNAMIC test/UsckfUnitTest.cpp:175-248 (ctor, 3 x setMeasurement, 2 x predict)
// once with registered models (GPU) and once with opaque functors (the boost::bind form).
// Prints "name rows cols v0 v1 ..." lines (column-major) that tests/test_gpu_facade.py checks
// against the golden fixtures.
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include <localization/filters/Msckf.hpp>
#include <localization/filters/Usckf.hpp>
#include <localization/filters/MtkWrap.hpp>
#include <localization/filters/State.hpp>

using namespace localization;

typedef MtkWrap<State> WSingleState;
typedef MtkDynamicWrap<MultiState<State, SensorState> > WMultiState;
typedef Msckf<WMultiState, WSingleState> MultiStateFilter;
typedef MtkMultiStateWrap<AugmentedState<-1> > WAugmentedState;
typedef Usckf<WAugmentedState, WSingleState> StateFilterDynamic;

static const double D2R = M_PI / 180.0;

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

// the reference's process model as an opaque functor (test/MsckfUnitTest.cpp:33-47)
struct DeltaPoseFunctor
{
    slk::Vec3 dp, v, w; slk::Quaternion dq;
    WSingleState operator()(const WSingleState &state) const
    {
        WSingleState s2;
        s2.orient = state.orient * dq;
        s2.angvelo = w;
        const slk::Quaternion &q = s2.orient;       // q * v as Eigen's _transformVector
        double ux = q.y() * dp[2] - q.z() * dp[1], uy = q.z() * dp[0] - q.x() * dp[2], uz = q.x() * dp[1] - q.y() * dp[0];
        ux += ux; uy += uy; uz += uz;
        s2.pos[0] = state.pos[0] + (dp[0] + q.w() * ux + (q.y() * uz - q.z() * uy));
        s2.pos[1] = state.pos[1] + (dp[1] + q.w() * uy + (q.z() * ux - q.x() * uz));
        s2.pos[2] = state.pos[2] + (dp[2] + q.w() * uz + (q.x() * uy - q.y() * ux));
        s2.velo = v;
        return s2;
    }
};

static int msckf_scenario(int k, bool functor)
{
    WMultiState statek_0;
    statek_0.sensorsk.resize(k);
    const int N = 12 + 6 * k;
    slk::Matrix Pk_0 = 0.025 * slk::Matrix::Identity(N, N);
    slk::Vec3 dpos(0.1, 0.1, 0.1), vel(0.1, 0.1, 0.1), angvel(0.1, 0.1, 0.1);
    slk::Quaternion dq = slk::Quaternion::exp(slk::Vec3(0, 0, D2R)) * slk::Quaternion::exp(slk::Vec3(0, D2R, 0))
                         * slk::Quaternion::exp(slk::Vec3(D2R, 0, 0));          // MsckfUnitTest.cpp:165-168
    slk::Matrix cov_process = 0.01 * slk::Matrix::Identity(12, 12);              // :173-177
    MultiStateFilter filter(statek_0, Pk_0);                                     // :179-180
    for (int i = 0; i < 2; ++i) {                                                // :196-206
        if (functor) { DeltaPoseFunctor f; f.dp = dpos; f.v = vel; f.w = angvel; f.dq = dq; filter.predict(f, cov_process); }
        else filter.predict(slk::DeltaPoseModel(dpos, dq, vel, angvel), cov_process);
        char nm[64];
        std::snprintf(nm, sizeof nm, "msckf_k%d_%s_pred%d_mean", k, functor ? "functor" : "model", i);
        dump_mean(nm, filter.muState(), 13 + 7 * k);
        std::snprintf(nm, sizeof nm, "msckf_k%d_%s_pred%d_P", k, functor ? "functor" : "model", i);
        dump(nm, filter.getPk());
    }
    // update with the registered feature-projection model on the golden scenario's features
    const int nf = k > 0 ? 4 : 1;
    slk::FeatureProjectionModel h;
    slk::Vector z(2 * nf);
    for (int j = 0; j < nf; ++j) {
        h.add(0.5 * (j - 1.5), 0.3 * (1.5 - j), 5.0 + j, k ? (j % k) + 1 : 0);
        z[2 * j] = 0.1 * (j - 1.0) * 0.5;
        z[2 * j + 1] = 0.05 * (j + 0.5) * 0.5;
    }
    slk::Matrix R = 0.01 * slk::Matrix::Identity(2 * nf, 2 * nf);
    unsigned int outliers = filter.update(z, h, R);
    char nm[64];
    std::snprintf(nm, sizeof nm, "msckf_k%d_%s_upd_mean", k, functor ? "functor" : "model");
    dump_mean(nm, filter.muState(), 13 + 7 * k);
    std::snprintf(nm, sizeof nm, "msckf_k%d_%s_upd_P", k, functor ? "functor" : "model");
    dump(nm, filter.getPk());
    std::printf("msckf_k%d_%s_outliers 1 1 %u\n", k, functor ? "functor" : "model", outliers);
    return filter.status();
}

static int usckf_scenario()
{
    WSingleState state_single;
    const double dt = 0.01;                                                       // UsckfUnitTest.cpp:182
    slk::Matrix P0_single = 0.0025 * slk::Matrix::Identity(12, 12);               // :186
    StateFilterDynamic filter(state_single, P0_single);                           // :191
    dump("usckf_ctor_P", filter.PkAugmentedState());
    slk::Vector featuresVO(3), featuresICP(9);
    for (int i = 0; i < 3; ++i) featuresVO[i] = 3.34;
    for (int i = 0; i < 9; ++i) featuresICP[i] = 1.34;
    slk::Matrix featuresVOCov = 0.008 * slk::Matrix::Identity(3, 3), featuresICPCov = 0.008 * slk::Matrix::Identity(9, 9);
    filter.setMeasurement(STATEK, featuresVO, featuresVOCov);                     // :210
    filter.setMeasurement(STATEK_L, featuresICP, featuresICPCov);                 // :216
    for (int i = 0; i < 3; ++i) featuresVO[i] = 3.35;
    featuresVOCov = 0.05 * slk::Matrix::Identity(3, 3);
    filter.setMeasurement(STATEK, featuresVO, featuresVOCov);                     // :225
    dump("usckf_setm2_P", filter.PkAugmentedState());
    dump_mean("usckf_setm2_mean", filter.muState(), 39 + 12);
    slk::Matrix myCov = (0.1 * dt) * slk::Matrix::Identity(12, 12);               // :51-60
    for (int i = 0; i < 2; ++i) {                                                 // :239-248
        slk::Vec3 velo(100.0, 0.0, 0.0), angular_velo(100.0 * D2R, 100.0 * D2R, 100.0 * D2R);
        filter.predict(slk::ConstVelocityModel(velo, angular_velo, dt), myCov);
        char nm[64];
        std::snprintf(nm, sizeof nm, "usckf_pred%d_P", i);
        dump(nm, filter.PkAugmentedState());
        std::snprintf(nm, sizeof nm, "usckf_pred%d_mean", i);
        dump_mean(nm, filter.muState(), 39 + 12);
    }
    // literal update(): the cloned covariance is indefinite -> reported, not applied
    slk::Vector measurementVO(3);
    measurementVO[0] = 2.33; measurementVO[1] = 3.35; measurementVO[2] = 3.35;    // :268
    slk::Matrix measurementNoiseVO = 0.01 * slk::Matrix::Identity(3, 3);          // :280-283
    filter.update(measurementVO, slk::VoRelativeModel(), measurementNoiseVO);     // :284
    std::printf("usckf_literal_update_status 1 1 %d\n", filter.status());
    return 0;
}

// dead reckoning fused into predict (src/core/DeadReckon.hpp:129-239 -> the delta-pose model): the tag type
// through the facade, input row 0 and 1 of tests/golden/dead_reckon.npz are passed on the command line by the test
static void dead_reckon_scenario(const double *u)
{
    WMultiState statek_0;
    statek_0.sensorsk.resize(1);
    slk::Matrix Pk_0 = 0.025 * slk::Matrix::Identity(18, 18);
    slk::Matrix cov_process = 0.01 * slk::Matrix::Identity(12, 12);
    MultiStateFilter filter(statek_0, Pk_0);
    filter.predict(slk::DeadReckonModel(u[0], slk::Vec3(u[1], u[2], u[3]), slk::Vec3(u[4], u[5], u[6]),
                                        slk::Vec3(u[7], u[8], u[9]), slk::Vec3(u[10], u[11], u[12])), cov_process);
    dump_mean("dead_reckon_mean", filter.muState(), 20);
    dump("dead_reckon_P", filter.getPk());
}

// EKF update through the facade (Msckf.hpp:284-349): the functor has the reference's h(mu_state, H) form; the
// Jacobian and the expected measurement are closed-form patterns the Python test rebuilds
struct PatternEkfModel
{
    int m, N;
    slk::Vector operator()(const WMultiState &, slk::Matrix &H) const
    {
        slk::Vector zm(m);
        for (int i = 0; i < m; ++i) {
            zm[i] = std::cos(0.3 * i);
            for (int j = 0; j < N; ++j) H(i, j) = (j >= 6 && j < 12) ? 0.0 : std::sin(0.37 * i + 1.3 * j) + ((i % N) == j ? 2.0 : 0.0);
        }
        return zm;
    }
};

static void ekf_scenario()
{
    const int k = 1, N = 18, m = 24;
    WMultiState statek_0;
    statek_0.sensorsk.resize(k);
    slk::Matrix Pk_0 = 0.025 * slk::Matrix::Identity(N, N);
    MultiStateFilter filter(statek_0, Pk_0);
    slk::Matrix H(m, N), R = 0.04 * slk::Matrix::Identity(m, m);
    slk::Vector z(m);
    for (int i = 0; i < m; ++i) z[i] = std::cos(0.3 * i) + 0.1 * std::sin(1.0 * i);
    z[6] += 30.0;                                              // one outlier block
    PatternEkfModel hm; hm.m = m; hm.N = N;
    unsigned outliers = filter.update(z, hm, H, R);
    dump_mean("ekf_mean", filter.muState(), 20);
    dump("ekf_P", filter.getPk());
    std::printf("ekf_outliers 1 1 %u\n", outliers);
}

int main(int argc, char **argv)
{
    int st = 0;
    if (argc == 14) {
        double u[13];
        for (int i = 0; i < 13; ++i) u[i] = std::atof(argv[1 + i]);
        dead_reckon_scenario(u);
    }
    for (int k : {0, 4, 8}) st |= msckf_scenario(k, false);
    st |= msckf_scenario(4, true);
    usckf_scenario();
    ekf_scenario();
    std::printf("msckf_status 1 1 %d\n", st);
    return 0;
}
