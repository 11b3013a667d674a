/**\file State.hpp
 * Filter state types of the GPU-backed facade: same names, members, DOF constants, tangent ordering
 * and host-side manifold interface -- set() / boxplus() / boxminus() / getVectorizedState() /
 * operator<< / operator>> -- as the reference's src/filters/State.hpp (ReducedState :39-135,
 * State :137-240, SensorState :242-334, MultiState :336-527, AugmentedState :529-669), without the MTK /
 * Eigen dependency (slk::Vec3 / slk::Quaternion / slk::Vector stand in for MTK::vect / MTK::SO3 /
 * Eigen vectors, SlkTypes.hpp).
 *
 * Inside the filters the manifold arithmetic runs on the GPU; the methods here are what CLIENT code
 * calls on state objects (the reference's test models do: test/UsckfUnitTest.cpp:40-41, :69,
 * test/MsckfUnitTest.cpp:39-43, :61-113), plus the (de)serialisation to the C-ABI storage layout
 * (include/slk.h): State = pos[3] quat[4:x,y,z,w] velo[3] angvelo[3], SensorState = pos[3] quat[4].
 * The stream operators are the reference's only checkpoint wire format (State.hpp:202-210, 298-306,
 * 483-507, 636-646): whitespace-separated numbers, quaternions as x y z w.
 *
 * A build that keeps the reference's own MTK-based State.hpp only has to provide the two
 * `slk_store` / `slk_load` overloads for its types.
 */
#ifndef _STATE_HPP_
#define _STATE_HPP_

#include <iostream>
#include <vector>

#include "SlkTypes.hpp"

namespace localization
{
    typedef slk::Vec3 vec3;
    typedef slk::Quaternion SO3;

    namespace slk
    {
        /** Euler-angle forms of set() / getVectorizedState() (State.hpp:171-176, :224-229):
         *  orient = Rz(a[2]) * Ry(a[1]) * Rx(a[0]); back: eulerAngles(2,1,0) of the rotation matrix. */
        inline Quaternion euler_zyx(const double *a)
        {
            const double v0[3] = {a[0], 0, 0}, v1[3] = {0, a[1], 0}, v2[3] = {0, 0, a[2]};
            return Quaternion::exp(v2) * Quaternion::exp(v1) * Quaternion::exp(v0);
        }
        inline void to_euler_zyx(const Quaternion &q, double *a)
        {
            // Eigen MatrixBase::eulerAngles(2, 1, 0): first angle in [0, pi], then the other two
            const Mat3 R = q.toRotationMatrix();
            double yaw = std::atan2(R(1, 0), R(0, 0));
            const double c2 = std::sqrt(R(2, 2) * R(2, 2) + R(2, 1) * R(2, 1));
            double pitch;
            if (yaw < 0.0) { yaw += M_PI; pitch = std::atan2(-R(2, 0), -c2); }
            else pitch = std::atan2(-R(2, 0), c2);
            const double s1 = std::sin(yaw), c1 = std::cos(yaw);
            const double roll = std::atan2(s1 * R(0, 2) - c1 * R(1, 2), c1 * R(1, 1) - s1 * R(0, 1));
            a[2] = yaw; a[1] = pitch; a[0] = roll;
        }
        template <class Q>
        inline void set_orient(Q &orient, const double *axis_angle, int type)
        {
            if (type == 0) orient = euler_zyx(axis_angle);
            else orient = Quaternion::exp(axis_angle, 1);           // State.hpp:179
        }
        template <class Q>
        inline void get_orient(const Q &orient, double *o, int type)
        {
            if (type == 0) to_euler_zyx(orient, o);
            else { const Vec3 r = Quaternion::log(orient); o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; }   // State.hpp:231
        }
    }

    /** State.hpp:39-135 */
    struct ReducedState
    {
        typedef ReducedState self;
        vec3 pos; SO3 orient; vec3 velo;
        enum { DOF = 6 };                                    // pos + orient (State.hpp:48-51)
        enum VectorizedMode { EULER_ANGLES = 0, ANGLE_AXIS = 1 };
        typedef double scalar;
        typedef slk::Vector vectorized_type;
        ReducedState(const vec3 &pos = vec3(), const SO3 &orient = SO3(), const vec3 &velo = vec3()) : pos(pos), orient(orient), velo(velo) {}
        void set(const vectorized_type &v, const VectorizedMode type = ANGLE_AXIS)
        {
            for (int i = 0; i < 3; ++i) pos[i] = v[i];
            slk::set_orient(orient, v.data() + 3, type);
        }
        void boxplus(const double *vec, scalar scale = 1) { pos.boxplus(vec, scale); orient.boxplus(vec + 3, scale); }
        void boxminus(double *res, const ReducedState &oth) const { pos.boxminus(res, oth.pos); orient.boxminus(res + 3, oth.orient); }
        friend std::ostream &operator<<(std::ostream &os, const ReducedState &v) { return os << v.pos << " " << v.orient << " "; }
        friend std::istream &operator>>(std::istream &is, ReducedState &v) { return is >> v.pos >> v.orient; }
        vectorized_type getVectorizedState(const VectorizedMode type = ANGLE_AXIS) const
        {
            vectorized_type v(DOF);
            for (int i = 0; i < 3; ++i) v[i] = pos[i];
            slk::get_orient(orient, v.data() + 3, type);
            return v;
        }
    };

    /** State.hpp:137-240 */
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

        /** set the State from a vectorized State (State.hpp:166-184) */
        void set(const vectorized_type &v, const VectorizedMode type = ANGLE_AXIS)
        {
            assert(v.size() == DOF);
            for (int i = 0; i < 3; ++i) { pos[i] = v[i]; velo[i] = v[6 + i]; angvelo[i] = v[9 + i]; }
            slk::set_orient(orient, v.data() + 3, type);
        }
        /** State.hpp:186-192 */
        void boxplus(const double *vec, scalar scale = 1)
        {
            pos.boxplus(vec, scale); orient.boxplus(vec + 3, scale); velo.boxplus(vec + 6, scale); angvelo.boxplus(vec + 9, scale);
        }
        void boxplus(const vectorized_type &vec, scalar scale = 1) { assert(vec.size() == DOF); boxplus(vec.data(), scale); }
        /** State.hpp:194-200 */
        void boxminus(double *res, const State &oth) const
        {
            pos.boxminus(res, oth.pos); orient.boxminus(res + 3, oth.orient); velo.boxminus(res + 6, oth.velo); angvelo.boxminus(res + 9, oth.angvelo);
        }
        /** State.hpp:202-210 */
        friend std::ostream &operator<<(std::ostream &os, const State &v) { return os << v.pos << " " << " " << v.orient << " " << v.velo << " " << v.angvelo << " "; }
        friend std::istream &operator>>(std::istream &is, State &v) { return is >> v.pos >> v.orient >> v.velo >> v.angvelo; }
        /** State.hpp:215-239 */
        vectorized_type getVectorizedState(const VectorizedMode type = ANGLE_AXIS) const
        {
            vectorized_type v(DOF);
            for (int i = 0; i < 3; ++i) { v[i] = pos[i]; v[6 + i] = velo[i]; v[9 + i] = angvelo[i]; }
            slk::get_orient(orient, v.data() + 3, type);
            return v;
        }
    };

    /** State.hpp:242-334 */
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
        void set(const vectorized_type &v, const VectorizedMode type = ANGLE_AXIS)       // State.hpp:268-284
        {
            assert(v.size() == DOF);
            for (int i = 0; i < 3; ++i) pos[i] = v[i];
            slk::set_orient(orient, v.data() + 3, type);
        }
        void boxplus(const double *vec, scalar scale = 1) { pos.boxplus(vec, scale); orient.boxplus(vec + 3, scale); }   // :286-290
        void boxplus(const vectorized_type &vec, scalar scale = 1) { assert(vec.size() == DOF); boxplus(vec.data(), scale); }
        void boxminus(double *res, const SensorState &oth) const { pos.boxminus(res, oth.pos); orient.boxminus(res + 3, oth.orient); }   // :292-296
        friend std::ostream &operator<<(std::ostream &os, const SensorState &v) { return os << v.pos << " " << " " << v.orient << " "; }    // :298-301
        friend std::istream &operator>>(std::istream &is, SensorState &v) { return is >> v.pos >> v.orient; }                              // :303-306
        vectorized_type getVectorizedState(const VectorizedMode type = ANGLE_AXIS) const   // :311-333
        {
            vectorized_type v(DOF);
            for (int i = 0; i < 3; ++i) v[i] = pos[i];
            slk::get_orient(orient, v.data() + 3, type);
            return v;
        }
    };

    /** State.hpp:336-527 */
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

        /** State.hpp:380-400 */
        void set(const vectorized_type &v, const VectorizedMode type = ANGLE_AXIS)
        {
            assert(v.size() == (int)getDOF());
            statek.set(v.segment(0, _State::DOF), typename _State::VectorizedMode(type));
            for (std::size_t c = 0; c < sensorsk.size(); ++c)
                sensorsk[c].set(v.segment(_State::DOF + (int)c * _SensorState::DOF, _SensorState::DOF), typename _SensorState::VectorizedMode(type));
        }
        /** State.hpp:402-416: boxplus with the vectorized form of another multi state */
        void boxplus(MultiState &state, scalar scale = 1)
        {
            statek.boxplus(state.statek.getVectorizedState().data(), scale);
            for (std::size_t c = 0; c < state.sensorsk.size() && c < sensorsk.size(); ++c)
                sensorsk[c].boxplus(state.sensorsk[c].getVectorizedState().data(), scale);
        }
        /** State.hpp:418-434 */
        void boxplus(const vectorized_type &vec, scalar scale = 1)
        {
            if (vec.size() == (int)this->getDOF()) {
                statek.boxplus(vec.data(), scale);
                for (std::size_t c = 0; c < sensorsk.size(); ++c) sensorsk[c].boxplus(vec.data() + _State::DOF + c * _SensorState::DOF, scale);
            }
        }
        /** State.hpp:460-481 */
        void boxminus(vectorized_type *res, const MultiState &oth) const
        {
            assert(res->size() == (int)getDOF() && oth.sensorsk.size() == sensorsk.size());
            statek.boxminus(res->data(), oth.statek);
            for (std::size_t c = 0; c < sensorsk.size(); ++c) sensorsk[c].boxminus(res->data() + _State::DOF + c * _SensorState::DOF, oth.sensorsk[c]);
        }
        /** State.hpp:483-495 */
        friend std::ostream &operator<<(std::ostream &os, const MultiState &v)
        {
            os << "\n" << v.statek << "\n";
            for (typename std::vector<_SensorState>::const_iterator it = v.sensorsk.begin(); it != v.sensorsk.end(); ++it) os << *it << "\n";
            return os;
        }
        /** State.hpp:497-507: reads as many sensor poses as the object already holds */
        friend std::istream &operator>>(std::istream &is, MultiState &v)
        {
            is >> v.statek;
            for (typename std::vector<_SensorState>::iterator it = v.sensorsk.begin(); it != v.sensorsk.end(); ++it) is >> *it;
            return is;
        }
        /** State.hpp:509-526 (the reference sizes the result with the static DOF -- an out-of-range write as soon as
         *  there is a clone; sized with getDOF() here) */
        vectorized_type getVectorizedState(const VectorizedMode type = ANGLE_AXIS) const
        {
            vectorized_type v((int)getDOF());
            const vectorized_type s = statek.getVectorizedState(static_cast<typename _State::VectorizedMode>(type));
            for (int i = 0; i < _State::DOF; ++i) v[i] = s[i];
            for (std::size_t c = 0; c < sensorsk.size(); ++c) {
                const vectorized_type q = sensorsk[c].getVectorizedState(static_cast<typename _SensorState::VectorizedMode>(type));
                for (int i = 0; i < _SensorState::DOF; ++i) v[_State::DOF + (int)c * _SensorState::DOF + i] = q[i];
            }
            return v;
        }
    };

    /** State.hpp:529-669 */
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
            : statek(statek), statek_l(statek_l), statek_i(statek_i), featuresk(featuresk), featuresk_l(featuresk_l)
        {
            if (_MeasurementDimension > 0) {                 // MTK::vect<D> has D entries from the start
                if (this->featuresk.size() == 0) this->featuresk.resize(_MeasurementDimension);
                if (this->featuresk_l.size() == 0) this->featuresk_l.resize(_MeasurementDimension);
            }
        }
        /** State.hpp:556-577 */
        void set(const vectorized_type &v, const std::size_t size_featuresk, const std::size_t size_featuresk_l, const VectorizedMode type = ANGLE_AXIS)
        {
            assert(v.size() == (int)(DOF + size_featuresk + size_featuresk_l));
            statek.set(v.segment(0, State::DOF), State::VectorizedMode(type));
            statek_l.set(v.segment(State::DOF, State::DOF), State::VectorizedMode(type));
            statek_i.set(v.segment(2 * State::DOF, State::DOF), State::VectorizedMode(type));
            featuresk = v.segment(3 * State::DOF, (int)size_featuresk);
            featuresk_l = v.segment(3 * State::DOF + (int)size_featuresk, (int)size_featuresk_l);
        }
        unsigned int getDOF() const { return DOF + featuresk.size() + featuresk_l.size(); }   // State.hpp:590-593
        /** State.hpp:595-611 */
        void boxplus(AugmentedState &state, scalar scale = 1)
        {
            statek.boxplus(state.statek.getVectorizedState().data(), scale);
            statek_l.boxplus(state.statek_l.getVectorizedState().data(), scale);
            statek_i.boxplus(state.statek_i.getVectorizedState().data(), scale);
            featuresk = featuresk + state.featuresk;
            featuresk_l = featuresk_l + state.featuresk_l;
        }
        /** State.hpp:613-634: the result is a state OBJECT whose members hold the differences */
        void boxminus(AugmentedState &res, const AugmentedState &oth) const
        {
            State::vectorized_type d(State::DOF);
            statek.boxminus(d.data(), oth.statek); res.statek.set(d);
            statek_l.boxminus(d.data(), oth.statek_l); res.statek_l.set(d);
            statek_i.boxminus(d.data(), oth.statek_i); res.statek_i.set(d);
            res.featuresk = res.featuresk - oth.featuresk;
            res.featuresk_l = res.featuresk_l - oth.featuresk_l;
        }
        /** State.hpp:636-646 (dynamic feature vectors are written in [ ] so that they can be read back, MTK requires
         *  the brackets on input) */
        friend std::ostream &operator<<(std::ostream &os, const AugmentedState &v)
        {
            os << "\n" << v.statek << "\n" << v.statek_l << "\n" << v.statek_i << "\n[ " << v.featuresk << "]\n[ " << v.featuresk_l << "]\n";
            return os;
        }
        friend std::istream &operator>>(std::istream &is, AugmentedState &v)
        {
            return is >> v.statek >> v.statek_l >> v.statek_i >> v.featuresk >> v.featuresk_l;
        }
        /** State.hpp:648-668 */
        vectorized_type getVectorizedState(const VectorizedMode type = ANGLE_AXIS) const
        {
            vectorized_type v((int)getDOF());
            const State *st[3] = {&statek, &statek_l, &statek_i};
            for (int b = 0; b < 3; ++b) {
                const vectorized_type s = st[b]->getVectorizedState(static_cast<State::VectorizedMode>(type));
                for (int i = 0; i < State::DOF; ++i) v[b * State::DOF + i] = s[i];
            }
            for (int i = 0; i < featuresk.size(); ++i) v[3 * State::DOF + i] = featuresk[i];
            for (int i = 0; i < featuresk_l.size(); ++i) v[3 * State::DOF + featuresk.size() + i] = featuresk_l[i];
            return v;
        }
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
