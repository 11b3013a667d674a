/**\file State.hpp
 * Filter state types of the GPU-backed facade: same names, members, DOF constants and tangent
 * ordering as the reference's src/filters/State.hpp (State :137-240, SensorState :242-334,
 * MultiState :336-527, AugmentedState :529-669), without the MTK dependency.  The manifold
 * arithmetic itself (boxplus / boxminus / exp / log) runs on the GPU inside the filters; these
 * structs are the host-side value types plus their (de)serialisation to the C-ABI storage layout
 * (include/slk.h): State = pos[3] quat[4:x,y,z,w] velo[3] angvelo[3], SensorState = pos[3] quat[4].
 *
 * A build that keeps the reference's own MTK-based State.hpp only has to provide the two
 * `slk_store` / `slk_load` overloads for its types.
 */
#ifndef _STATE_HPP_
#define _STATE_HPP_

#include <vector>

#include "SlkTypes.hpp"

namespace localization
{
    typedef slk::Vec3 vec3;
    typedef slk::Quaternion SO3;

    struct State
    {
        typedef State self;
        vec3 pos; SO3 orient; vec3 velo; vec3 angvelo;
        enum { DOF = 12 };                                   // State.hpp:146-149
        enum { STORAGE = 13 };
        enum VectorizedMode { EULER_ANGLES = 0, ANGLE_AXIS = 1 };
        typedef double scalar;
        typedef slk::Vector vectorized_type;
        State(const vec3 &pos = vec3(), const SO3 &orient = SO3(), const vec3 &velo = vec3(), const vec3 &angvelo = vec3())
            : pos(pos), orient(orient), velo(velo), angvelo(angvelo) {}
    };

    struct SensorState
    {
        typedef SensorState self;
        vec3 pos; SO3 orient;
        enum { DOF = 6 };                                    // State.hpp:249-252
        enum { STORAGE = 7 };
        enum VectorizedMode { EULER_ANGLES = 0, ANGLE_AXIS = 1 };
        typedef double scalar;
        typedef slk::Vector vectorized_type;
        SensorState(const vec3 &pos = vec3(), const SO3 &orient = SO3()) : pos(pos), orient(orient) {}
    };

    template <class _State, class _SensorState>
    struct MultiState
    {
        typedef MultiState self;
        _State statek;                                       // State.hpp:341
        std::vector<_SensorState> sensorsk;                  // State.hpp:342
        enum { SENSOR_DOF = _SensorState::DOF };
        enum { DOF = _State::DOF + 0 };
        enum VectorizedMode { EULER_ANGLES = 0, ANGLE_AXIS = 1 };
        typedef double scalar;
        typedef slk::Vector vectorized_type;
        typedef _State SingleState;
        MultiState(const _State &statek = _State(), const std::vector<_SensorState> &sensorsk = std::vector<_SensorState>())
            : statek(statek), sensorsk(sensorsk) {}
        unsigned int getDOF() const { return _State::DOF + (SENSOR_DOF * sensorsk.size()); }   // State.hpp:373-376
    };

    template <int _MeasurementDimension>
    struct AugmentedState
    {
        typedef AugmentedState self;
        typedef slk::Vector MeasurementType;
        State statek, statek_l, statek_i;                    // State.hpp:536-538
        MeasurementType featuresk, featuresk_l;              // State.hpp:539-540
        enum { DOF = State::DOF + State::DOF + State::DOF + 0 };
        enum VectorizedMode { EULER_ANGLES = 0, ANGLE_AXIS = 1 };
        typedef double scalar;
        typedef slk::Vector vectorized_type;
        AugmentedState(const State &statek = State(), const State &statek_l = State(), const State &statek_i = State(),
                       const MeasurementType &featuresk = MeasurementType(), const MeasurementType &featuresk_l = MeasurementType())
            : statek(statek), statek_l(statek_l), statek_i(statek_i), featuresk(featuresk), featuresk_l(featuresk_l) {}
        unsigned int getDOF() const { return DOF + featuresk.size() + featuresk_l.size(); }   // State.hpp:590-593
    };

    // ---- C-ABI storage (de)serialisation ------------------------------------------------
    inline void slk_store(const State &s, double *o)
    {
        for (int i = 0; i < 3; ++i) { o[i] = s.pos[i]; o[7 + i] = s.velo[i]; o[10 + i] = s.angvelo[i]; }
        for (int i = 0; i < 4; ++i) o[3 + i] = s.orient.coeffs()[i];
    }
    inline void slk_load(State &s, const double *o)
    {
        for (int i = 0; i < 3; ++i) { s.pos[i] = o[i]; s.velo[i] = o[7 + i]; s.angvelo[i] = o[10 + i]; }
        for (int i = 0; i < 4; ++i) s.orient.coeffs()[i] = o[3 + i];
    }
    inline void slk_store(const SensorState &s, double *o)
    {
        for (int i = 0; i < 3; ++i) o[i] = s.pos[i];
        for (int i = 0; i < 4; ++i) o[3 + i] = s.orient.coeffs()[i];
    }
    inline void slk_load(SensorState &s, const double *o)
    {
        for (int i = 0; i < 3; ++i) s.pos[i] = o[i];
        for (int i = 0; i < 4; ++i) s.orient.coeffs()[i] = o[3 + i];
    }
    template <class S, class C>
    inline void slk_store(const MultiState<S, C> &m, double *o)
    {
        slk_store(static_cast<const S &>(m.statek), o);
        for (std::size_t c = 0; c < m.sensorsk.size(); ++c) slk_store(static_cast<const C &>(m.sensorsk[c]), o + 13 + 7 * c);
    }
    template <class S, class C>
    inline void slk_load(MultiState<S, C> &m, const double *o)
    {
        slk_load(static_cast<S &>(m.statek), o);
        for (std::size_t c = 0; c < m.sensorsk.size(); ++c) slk_load(static_cast<C &>(m.sensorsk[c]), o + 13 + 7 * c);
    }
    template <int M>
    inline void slk_store(const AugmentedState<M> &a, double *o)
    {
        slk_store(a.statek, o); slk_store(a.statek_l, o + 13); slk_store(a.statek_i, o + 26);
        for (int i = 0; i < a.featuresk.size(); ++i) o[39 + i] = a.featuresk[i];
        for (int i = 0; i < a.featuresk_l.size(); ++i) o[39 + a.featuresk.size() + i] = a.featuresk_l[i];
    }
    template <int M>
    inline void slk_load(AugmentedState<M> &a, const double *o, int nfk, int nfkl)
    {
        slk_load(a.statek, o); slk_load(a.statek_l, o + 13); slk_load(a.statek_i, o + 26);
        a.featuresk.resize(nfk); a.featuresk_l.resize(nfkl);
        for (int i = 0; i < nfk; ++i) a.featuresk[i] = o[39 + i];
        for (int i = 0; i < nfkl; ++i) a.featuresk_l[i] = o[39 + nfk + i];
    }
}

#endif /** end of _STATE_HPP_ */
