/**\file MtkWrap.hpp
 * Wrapper templates of the reference (src/filters/MtkWrap.hpp: MtkWrap :51-117, MtkDynamicWrap
 * :123-244, MtkMultiStateWrap :250-326).  In the reference they add operator+ / operator- / ==
 * on top of boxplus / boxminus, executed on the host inside the filter loops.  In the GPU-backed
 * facade those operators are evaluated on the device by the kernels, so the wrappers only carry
 * the type names and typedefs client code spells out.
 */
#ifndef _MTKWRAP_HPP_
#define _MTKWRAP_HPP_

#include "State.hpp"

namespace localization
{
    template <class M>
    struct MtkWrap : public M
    {
        typedef MtkWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        enum { DOF = M::DOF };
        typedef slk::Vector vectorized_type;
        MtkWrap(const M &m = M()) : M(m) {}
    };

    template <class M>
    struct MtkDynamicWrap : public M
    {
        typedef MtkDynamicWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        typedef slk::Vector vectorized_type;
        MtkDynamicWrap(const M &m = M()) : M(m) {}
    };

    template <class M>
    struct MtkMultiStateWrap : public M
    {
        typedef MtkMultiStateWrap<M> self;
        typedef typename M::scalar scalar_type;
        typedef typename M::VectorizedMode VectorizedMode;
        typedef slk::Vector vectorized_type;
        MtkMultiStateWrap(const M &m = M()) : M(m) {}
    };
}
#endif /* _MTKWRAP_HPP_ */
