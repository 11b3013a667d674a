// CPU-only: the host-side manifold interface of the facade (State.hpp / MtkWrap.hpp) against the assertions the
// reference's own tests hold -- test/MsckfUnitTest.cpp STATES :50-76 and OPERATIONS :78-123 (BOOST_CHECKs at :61, :62,
// :66, :71, :110, :113) -- plus operator<< / operator>> round trips (State.hpp:202-210, 298-306, 483-507, 636-646) and
// the verbatim model functions of both test files evaluated at a few states against closed forms.
// Exit code 0 = every check held; each failure prints a line.
#include <cstdio>
#include <sstream>
#include <vector>

#include "eigen_mtk_names.hpp"

/** Wrap the Multi State (test/MsckfUnitTest.cpp:24-28) **/
typedef localization::MtkWrap<localization::State> WSingleState;
typedef localization::MtkDynamicWrap< localization::MultiState<localization::State, localization::SensorState> > WMultiState;
typedef ::MTK::vect<Eigen::Dynamic, double> MeasurementType;
/** test/UsckfUnitTest.cpp:26-28 **/
typedef localization::MtkMultiStateWrap<localization::AugmentedState<Eigen::Dynamic> > WAugmentedState;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

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

// ---- verbatim from test/UsckfUnitTest.cpp:34-49 -------------------------------------------------------------
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

static void states()                         // test/MsckfUnitTest.cpp:50-76
{
    WMultiState mstate;
    CHECK(mstate.getDOF() == (unsigned)mstate.getVectorizedState().size());   // :61
    CHECK(mstate == mstate);                                                  // :62
    WMultiState mstatebis;
    mstatebis.set(mstate.getVectorizedState());
    CHECK(mstate == mstatebis);                                               // :66
    typedef localization::MtkDynamicWrap< localization::MultiState<localization::ReducedState, localization::SensorState> > WReducedMultiState;
    WReducedMultiState rmstate;
    CHECK(rmstate.DOF == 6);                                                  // :71
    // the same with clones (the reference's test has none)
    WMultiState m4;
    m4.sensorsk.resize(4);
    CHECK(m4.getDOF() == 36 && m4.getVectorizedState().size() == 36);
}

static void operations(int k)                // test/MsckfUnitTest.cpp:78-123
{
    WMultiState mstate, mstatebis;
    mstate.sensorsk.resize(k); mstatebis.sensorsk.resize(k);
    mstatebis.statek.pos<< 1, 2.0, -3.00;

    Eigen::Vector3d euler; /** In euler angles **/
    euler[2] = 1.00 * localization::D2R;
    euler[1] = 1.00 * localization::D2R;
    euler[0] = 1.00 * localization::D2R;

    mstatebis.statek.orient.boxplus(euler);

    localization::SensorState sstate(mstatebis.statek.pos, mstatebis.statek.orient);
    for (std::vector<localization::SensorState>::iterator it = mstatebis.sensorsk.begin();
                    it != mstatebis.sensorsk.end(); ++it)
    {
        it->set(sstate.getVectorizedState());
    }

    /** Operation with states **/
    WMultiState sumstate, resstate;
    sumstate.sensorsk.resize(k); resstate.sensorsk.resize(k);
    WMultiState::vectorized_type vresstate, deltastate;
    deltastate.resize(sumstate.getDOF(), 1);
    vresstate.resize(sumstate.getDOF(), 1);

    vresstate = mstate - mstatebis;
    deltastate = vresstate;
    resstate.set(vresstate);
    sumstate = mstate + deltastate;
    CHECK(resstate == sumstate);                                              // :110
    deltastate = -vresstate;
    sumstate = mstate + deltastate;
    CHECK(mstatebis == sumstate);                                             // :113
    CHECK(mstate != mstatebis);
}

template <class S>
static bool roundtrip(const S &a, S &b)
{
    std::stringstream ss;
    ss.precision(17);
    ss << a;
    ss >> b;
    return !ss.fail();
}

static void text_io()
{
    // State / SensorState / MultiState / AugmentedState through operator<< and operator>> (the reference's only
    // checkpoint wire format): what is read back equals what was written
    WSingleState s;
    s.pos << 1.5, -2.25, 3.125;
    s.velo << 0.1, 0.2, -0.3;
    s.angvelo << -0.01, 0.02, 0.03;
    s.orient.boxplus(Eigen::Vector3d(0.3, -0.2, 0.5));
    WSingleState s2;
    CHECK(roundtrip(s, s2));
    CHECK(s == s2);
    WMultiState m, m2;
    m.statek = s;
    m.sensorsk.resize(3); m2.sensorsk.resize(3);
    for (int c = 0; c < 3; ++c) { m.sensorsk[c].pos << 0.5 * c, 1.0 - c, 2.0; m.sensorsk[c].orient.boxplus(Eigen::Vector3d(0.1 * c, 0.2, -0.1)); }
    CHECK(roundtrip(m, m2));
    CHECK(m == m2);
    WAugmentedState a, a2;
    a.statek = s; a.statek_i = s2;
    a.statek_l.pos << 9, 8, 7;
    a.featuresk.resize(3); a.featuresk << 3.34, 3.35, 3.36;
    a.featuresk_l.resize(6); a.featuresk_l << 1, 2, 3, 4, 5, 6;
    CHECK(roundtrip(a, a2));
    CHECK(a2.featuresk.size() == 3 && a2.featuresk_l.size() == 6);
    CHECK(a == a2);
    // MTK input forms: brackets and commas
    std::stringstream ss("(1, 2, 3) [0 0 0 1] 4 5 6 {7,8,9}");
    WSingleState s3;
    ss >> s3;
    CHECK(!ss.fail() && s3.pos[1] == 2 && s3.orient.w() == 1 && s3.velo[2] == 6 && s3.angvelo[0] == 7);
}

static void models()
{
    // the reference's model functions on the host types against closed forms
    WSingleState s;
    s.pos << 1, 2, 3;
    s.velo << 0.5, 0, -0.5;
    s.orient.boxplus(Eigen::Vector3d(0, 0, M_PI / 2));                       // 90 deg about z
    Eigen::Vector3d dp(1, 0, 0), v(0.1, 0.2, 0.3), w(0, 0, 0.2);
    localization::SO3 dq = localization::SO3::exp(Eigen::Vector3d(0, 0, M_PI / 2));
    WSingleState a = processModel(s, dp, dq, v, w);                          // orientation 180 deg: x -> -x
    CHECK(std::fabs(a.pos[0] - 0.0) < 1e-15 && std::fabs(a.pos[1] - 2.0) < 1e-15 && a.velo[1] == 0.2 && a.angvelo[2] == 0.2);
    WSingleState b = processModel(s, v, w, 0.5);
    CHECK(std::fabs(b.pos[0] - 1.25) < 1e-15 && std::fabs(b.pos[2] - 2.75) < 1e-15 && b.velo[0] == 0.1);
    Eigen::Vector3d r = localization::SO3::log(s.orient.conjugate() * b.orient);
    CHECK(std::fabs(r[2] - 0.1) < 1e-15 && std::fabs(r[0]) < 1e-15);
    WAugmentedState x;
    x.statek = s;                                                            // statek - statek_i = (pos diff, rotation diff)
    x.statek_i.pos << 0, 2, 3;
    x.featuresk.resize(3); x.featuresk << 1, 1, 1;
    MeasurementType z = measurementModelVO(x);                               // R(90 deg z) * (1,1,1) + (1,0,0)
    CHECK(std::fabs(z[0] - 0.0) < 1e-15 && std::fabs(z[1] - 1.0) < 1e-15 && std::fabs(z[2] - 1.0) < 1e-15);
}

int main()
{
    states();
    operations(0);          // the reference's case
    operations(3);
    text_io();
    models();
    if (failures) std::printf("%d check(s) failed\n", failures);
    else std::printf("all manifold identities hold\n");
    return failures ? 1 : 0;
}
