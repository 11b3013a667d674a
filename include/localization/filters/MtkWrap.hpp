/**\file MtkWrap.hpp
 * Wrapper templates of the reference (src/filters/MtkWrap.hpp: MtkWrap :51-117, MtkDynamicWrap
 * :123-244, MtkMultiStateWrap :250-326): operator+ = boxplus, operator- = boxminus, == / != through
 * `isZero(1e-12)` of the difference, assignment from a vectorized state.  These are the HOST-side
 * operators client code (model functors, tests) applies to state objects; inside the filters the same
 * operations run on the GPU.
 */
#ifndef _MTKWRAP_HPP_
#define _MTKWRAP_HPP_

#include <cassert>

#include "State.hpp"

namespace localization
{
    /** MtkWrap.hpp:51-117: static DOF, vectorized_type of DOF entries */
    template <class M>
    struct MtkWrap : public M
    {
        typedef MtkWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        enum { DOF = M::DOF };
        typedef slk::Vector vectorized_type;
        MtkWrap(const M &m = M()) : M(m) {}

        self &operator=(const vectorized_type &vstate) { this->set(vstate); return *this; }                  // :65-69
        self &operator+=(const vectorized_type &delta_state)                                                 // :75-80
        {
            assert(delta_state.size() == DOF);
            M::boxplus(delta_state.data());
            return *this;
        }
        const self operator+(const vectorized_type &delta_state) const { self result = *this; result += delta_state; return result; }   // :82-88
        const vectorized_type operator-(const self &other) const                                             // :93-100
        {
            vectorized_type result(DOF);
            M::boxminus(result.data(), other);
            return result;
        }
        bool operator==(const self &other) const { vectorized_type diff = (*this) - other; return diff.isZero(1e-12); }   // :102-106
        bool operator!=(const self &other) const { return !(*this == other); }
    };

    /** MtkWrap.hpp:123-244: run-time DOF (MultiState), difference is a vector */
    template <class M>
    struct MtkDynamicWrap : public M
    {
        typedef MtkDynamicWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        typedef slk::Vector vectorized_type;
        MtkDynamicWrap(const M &m = M()) : M(m) {}

        self &operator=(const vectorized_type &vstate) { this->set(vstate); return *this; }                  // :139-143
        self &operator=(const self &state) { this->statek = state.statek; this->sensorsk = state.sensorsk; return *this; }   // :148-153
        MtkDynamicWrap(const self &state) : M(state) {}
        /** manifold operator (+) with another multi state taken as a delta (:160-175) */
        self &operator+=(const self &delta_state)
        {
            assert(delta_state.getDOF() == M::getDOF());
            M state; state = delta_state;
            M::boxplus(state);
            return *this;
        }
        const self operator+(const self &delta_state) const { self result = *this; result += delta_state; return result; }
        /** manifold operator (+) with a vector (:181-195) */
        self &operator+=(const vectorized_type &delta_state)
        {
            assert(delta_state.size() == (int)M::getDOF());
            M::boxplus(delta_state);
            return *this;
        }
        const self operator+(const vectorized_type &delta_state) const { self result = *this; result += delta_state; return result; }
        /** manifold operator (-) (:218-228) */
        const vectorized_type operator-(const self &other) const
        {
            vectorized_type result;
            result.resize(M::getDOF(), 1);
            assert(result.size() == (int)other.getDOF());
            M::boxminus(&result, other);
            return result;
        }
        bool operator==(const self &other) const { vectorized_type diff = (*this) - other; return diff.isZero(1e-12); }   // :231-235
        bool operator!=(const self &other) const { return !(*this == other); }
    };

    /** MtkWrap.hpp:250-326: AugmentedState; the difference is a state OBJECT (:297-310).  The reference's
     *  operator== (:313-317) assigns that object to a vector and would not compile if instantiated (SURVEY.md
     *  Appendix B.9); here it compares the vectorized difference, which is what it means. */
    template <class M>
    struct MtkMultiStateWrap : public M
    {
        typedef MtkMultiStateWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        typedef slk::Vector vectorized_type;
        MtkMultiStateWrap(const M &m = M()) : M(m) {}
        MtkMultiStateWrap(const self &state) : M(state) {}

        self &operator=(const self &state)                                                                   // :264-272
        {
            this->statek = state.statek; this->statek_l = state.statek_l; this->statek_i = state.statek_i;
            this->featuresk = state.featuresk; this->featuresk_l = state.featuresk_l;
            return *this;
        }
        self &operator+=(const self &delta_state)                                                            // :278-285
        {
            assert(delta_state.getDOF() == M::getDOF());
            M state; state = delta_state;
            M::boxplus(state);
            return *this;
        }
        const self operator+(const self &delta_state) const { self result = *this; result += delta_state; return result; }
        const self operator-(const self &other) const                                                        // :297-310
        {
            M result;
            assert(other.getDOF() == M::getDOF());
            result.featuresk = this->featuresk;
            result.featuresk_l = this->featuresk_l;
            M::boxminus(result, other);
            return result;
        }
        bool operator==(const self &other) const
        {
            self diff = (*this) - other;
            return diff.getVectorizedState().isZero(1e-12);
        }
        bool operator!=(const self &other) const { return !(*this == other); }
    };
}
#endif /* _MTKWRAP_HPP_ */
