// A caller that holds its matrices and vectors in types of its OWN -- fixed-size, column-major, nothing to do with
// slk::Matrix / slk::Vector -- the way a Rock task holds Eigen::Matrix<double, 12, 12> / Eigen::VectorXd: every
// matrix / vector argument of the facade classes (constructor, predict's Q, update's z and R, the EKF's H, setPk,
// setPkSingleState, setMeasurement) is a template on "anything column-major with data() / rows() / cols()" (vectors:
// data() / size() / operator[]), and what the classes hand out (getPk, getPkSingleState, PkAugmentedState) converts into
// the caller's type.  The same scenario runs once on these foreign types and once on the facade's own: the printed
// results must be identical (tests/test_gpu_facade.py).
#include <cstdio>
#include <cmath>

#include <localization/filters/Msckf.hpp>
#include <localization/filters/Usckf.hpp>
#include <localization/filters/MtkWrap.hpp>
#include <localization/filters/State.hpp>

namespace fx {   // the caller's own linear algebra: no relation to the slk:: types
template <int R, int C> struct Mat {
    double a[R * C];
    Mat() { for (int i = 0; i < R * C; ++i) a[i] = 0.0; }
    static Mat Identity(double s) { Mat m; for (int i = 0; i < R && i < C; ++i) m(i, i) = s; return m; }
    int rows() const { return R; }
    int cols() const { return C; }
    double *data() { return a; }
    const double *data() const { return a; }
    double &operator()(int i, int j) { return a[i + R * j]; }
    double operator()(int i, int j) const { return a[i + R * j]; }
};
template <int M> struct Vec {
    double a[M];
    Vec() { for (int i = 0; i < M; ++i) a[i] = 0.0; }
    int size() const { return M; }
    int rows() const { return M; }
    int cols() const { return 1; }
    double *data() { return a; }
    const double *data() const { return a; }
    double &operator[](int i) { return a[i]; }
    double operator[](int i) const { return a[i]; }
};
}

using namespace localization;
typedef MtkWrap<State> WSingleState;
typedef MtkDynamicWrap<MultiState<State, SensorState> > WMultiState;
typedef Msckf<WMultiState, WSingleState> MultiStateFilter;
typedef MtkMultiStateWrap<AugmentedState<-1> > WAugmentedState;
typedef Usckf<WAugmentedState, WSingleState> StateFilterDynamic;
static const double D2R = M_PI / 180.0;

template <class M> static void dump(const char *name, const M &m)
{
    std::printf("%s %d %d", name, (int)m.rows(), (int)m.cols());
    for (int i = 0; i < (int)m.rows() * (int)m.cols(); ++i) std::printf(" %.17g", m.data()[i]);
    std::printf("\n");
}
template <class S> static void dump_mean(const char *name, const S &s, int nq)
{
    std::vector<double> v(nq);
    slk_store(s, v.data());
    std::printf("%s %d 1", name, nq);
    for (int i = 0; i < nq; ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}

// the reference's h(mu_state, H) functor form of the EKF update (Msckf.hpp:310), Jacobian in the caller's type
template <class Jac, class Meas> struct PatternEkfModel {
    Meas operator()(const WMultiState &, Jac &H) const
    {
        Meas zm;
        for (int i = 0; i < (int)H.rows(); ++i) {
            zm[i] = std::cos(0.3 * i);
            for (int j = 0; j < (int)H.cols(); ++j) H(i, j) = (j >= 6 && j < 12) ? 0.0 : std::sin(0.37 * i + 1.3 * j) + ((i % (int)H.cols()) == j ? 2.0 : 0.0);
        }
        return zm;
    }
};

// K clones, N = 12 + 6 K; CovN / Cov12 / Cov8 / Vec8 are the matrix / vector types of this run
template <int K, class CovN, class Cov12, class Cov8, class Vec8>
static void msckf_run(const char *tag)
{
    const int N = 12 + 6 * K;
    WMultiState statek_0;
    statek_0.sensorsk.resize(K);
    CovN Pk_0;
    Cov12 Q;
    Cov8 R;
    for (int i = 0; i < N; ++i) Pk_0(i, i) = 0.025;
    for (int i = 0; i < 12; ++i) Q(i, i) = 0.01;
    for (int i = 0; i < 8; ++i) R(i, i) = 0.01;
    MultiStateFilter filter(statek_0, Pk_0);                                     // MsckfUnitTest.cpp:179-180
    slk::Vec3 dpos(0.1, 0.1, 0.1), vel(0.1, 0.1, 0.1), angvel(0.1, 0.1, 0.1);
    slk::Quaternion dq = slk::Quaternion::exp(slk::Vec3(0, 0, D2R)) * slk::Quaternion::exp(slk::Vec3(0, D2R, 0))
                         * slk::Quaternion::exp(slk::Vec3(D2R, 0, 0));
    for (int i = 0; i < 2; ++i) filter.predict(slk::DeltaPoseModel(dpos, dq, vel, angvel), Q);
    slk::FeatureProjectionModel h;
    Vec8 z;
    for (int j = 0; j < 4; ++j) {
        h.add(0.5 * (j - 1.5), 0.3 * (1.5 - j), 5.0 + j, (j % K) + 1);
        z[2 * j] = 0.1 * (j - 1.0) * 0.5;
        z[2 * j + 1] = 0.05 * (j + 0.5) * 0.5;
    }
    const unsigned int outliers = filter.update(z, h, R);
    char nm[96];
    std::snprintf(nm, sizeof nm, "%s_upd_mean", tag); dump_mean(nm, filter.muState(), 13 + 7 * K);
    CovN Pout = filter.getPk();                                                  // hands out slk::Matrix: converts
    std::snprintf(nm, sizeof nm, "%s_upd_P", tag); dump(nm, Pout);
    Cov12 P12 = filter.getPkSingleState();                                       // Msckf.hpp:368-374
    std::snprintf(nm, sizeof nm, "%s_P12", tag); dump(nm, P12);
    std::printf("%s_outliers 1 1 %u\n", tag, outliers);
    // setPkSingleState / setPk with the caller's types (:363-366, :391-395), then another predict
    for (int i = 0; i < 12; ++i) P12(i, i) += 0.001;
    filter.setPkSingleState(P12);
    filter.predict(slk::DeltaPoseModel(dpos, dq, vel, angvel), Q);
    Pout = filter.getPk();
    for (int i = 0; i < N; ++i) Pout(i, i) += 0.002;
    filter.setPk(Pout);
    filter.predict(slk::DeltaPoseModel(dpos, dq, vel, angvel), Q);
    std::snprintf(nm, sizeof nm, "%s_final_P", tag); dump(nm, CovN(filter.getPk()));
    std::printf("%s_status 1 1 %d\n", tag, filter.status());
}

template <class Cov18, class Jac, class Cov24, class Vec24>
static void ekf_run(const char *tag)
{
    WMultiState statek_0;
    statek_0.sensorsk.resize(1);
    Cov18 Pk_0;
    for (int i = 0; i < 18; ++i) Pk_0(i, i) = 0.025;
    MultiStateFilter filter(statek_0, Pk_0);
    Jac H;
    Cov24 R;
    Vec24 z;
    for (int i = 0; i < 24; ++i) { R(i, i) = 0.04; z[i] = std::cos(0.3 * i) + 0.1 * std::sin(1.0 * i); }
    z[6] += 30.0;
    PatternEkfModel<Jac, Vec24> hm;
    const unsigned int outliers = filter.update(z, hm, H, R);                    // Msckf.hpp:284-290
    char nm[96];
    std::snprintf(nm, sizeof nm, "%s_ekf_mean", tag); dump_mean(nm, filter.muState(), 20);
    std::snprintf(nm, sizeof nm, "%s_ekf_P", tag); dump(nm, Cov18(filter.getPk()));
    std::printf("%s_ekf_outliers 1 1 %u\n", tag, outliers);
    // a caller-side significance test (Msckf.hpp:297-349) that rejects every block: nothing is applied (:320); one that
    // leaves fewer rows than states: skipped and reported like the built-in gate does (the reference reads out of range)
    {
        MultiStateFilter f2(statek_0, Pk_0);
        struct RejectAll { bool operator()(const double &, int) const { return false; } } none;
        const unsigned int o2 = f2.update(z, hm, H, R, none);
        std::snprintf(nm, sizeof nm, "%s_ekf_none_P", tag); dump(nm, Cov18(f2.getPk()));
        std::printf("%s_ekf_none_outliers 1 1 %u\n%s_ekf_none_status 1 1 %d\n", tag, o2, tag, f2.status());
        struct RejectSome { mutable int calls; bool operator()(const double &, int) const { return calls++ >= 4; } } some = {0};
        const unsigned int o3 = f2.update(z, hm, H, R, some);
        std::snprintf(nm, sizeof nm, "%s_ekf_some_P", tag); dump(nm, Cov18(f2.getPk()));
        std::printf("%s_ekf_some_outliers 1 1 %u\n%s_ekf_some_status 1 1 %d\n", tag, o3, tag, f2.status());
    }
}

template <class Cov12, class Vec3T, class Cov3, class Vec9T, class Cov9>
static void usckf_run(const char *tag)
{
    WSingleState state_single;
    const double dt = 0.01;
    Cov12 P0, Q;
    Cov3 R3, Rm;
    Cov9 R9;
    Vec3T f3, zm;
    Vec9T f9;
    for (int i = 0; i < 12; ++i) { P0(i, i) = 0.0025; Q(i, i) = 0.1 * dt; }
    for (int i = 0; i < 3; ++i) { R3(i, i) = 0.008; Rm(i, i) = 0.01; f3[i] = 3.34; }
    for (int i = 0; i < 9; ++i) { R9(i, i) = 0.008; f9[i] = 1.34; }
    StateFilterDynamic filter(state_single, P0);                                  // UsckfUnitTest.cpp:191
    filter.setMeasurement(STATEK, f3, R3);                                        // :210
    filter.setMeasurement(STATEK_L, f9, R9);                                      // :216
    slk::Vec3 velo(100.0, 0.0, 0.0), angular_velo(100.0 * D2R, 100.0 * D2R, 100.0 * D2R);
    for (int i = 0; i < 2; ++i) filter.predict(slk::ConstVelocityModel(velo, angular_velo, dt), Q);
    char nm[96];
    std::snprintf(nm, sizeof nm, "%s_usckf_mean", tag); dump_mean(nm, filter.muState(), 39 + 12);
    std::snprintf(nm, sizeof nm, "%s_usckf_P", tag); dump(nm, slk::Matrix(filter.PkAugmentedState()));
    Cov12 Pi = filter.PkSingleState(STATEK_I);                                    // Usckf.hpp:493-516
    std::snprintf(nm, sizeof nm, "%s_usckf_PkI", tag); dump(nm, Pi);
    zm[0] = 2.33; zm[1] = 3.35; zm[2] = 3.35;
    filter.update(zm, slk::VoRelativeModel(), Rm);                                // (indefinite cloned covariance: reported)
    std::printf("%s_usckf_status 1 1 %d\n", tag, filter.status());
}

// fixed-size stand-ins for the facade's dynamic types in the second run
struct SlkMat : slk::Matrix { SlkMat() {} SlkMat(const slk::Matrix &m) : slk::Matrix(m) {} };
template <int R, int C> struct SlkMatRC : slk::Matrix { SlkMatRC() : slk::Matrix(R, C) {} SlkMatRC(const slk::Matrix &m) : slk::Matrix(m) {} };
template <int M> struct SlkVecM : slk::Vector { SlkVecM() : slk::Vector(M) {} };

int main()
{
    msckf_run<8, fx::Mat<60, 60>, fx::Mat<12, 12>, fx::Mat<8, 8>, fx::Vec<8> >("foreign_k8");
    msckf_run<8, SlkMatRC<60, 60>, SlkMatRC<12, 12>, SlkMatRC<8, 8>, SlkVecM<8> >("own_k8");
    msckf_run<2, fx::Mat<24, 24>, fx::Mat<12, 12>, fx::Mat<8, 8>, fx::Vec<8> >("foreign_k2");
    msckf_run<2, SlkMatRC<24, 24>, SlkMatRC<12, 12>, SlkMatRC<8, 8>, SlkVecM<8> >("own_k2");
    ekf_run<fx::Mat<18, 18>, fx::Mat<24, 18>, fx::Mat<24, 24>, fx::Vec<24> >("foreign");
    ekf_run<SlkMatRC<18, 18>, SlkMatRC<24, 18>, SlkMatRC<24, 24>, SlkVecM<24> >("own");
    usckf_run<fx::Mat<12, 12>, fx::Vec<3>, fx::Mat<3, 3>, fx::Vec<9>, fx::Mat<9, 9> >("foreign");
    usckf_run<SlkMatRC<12, 12>, SlkVecM<3>, SlkMatRC<3, 3>, SlkVecM<9>, SlkMatRC<9, 9> >("own");
    return 0;
}
