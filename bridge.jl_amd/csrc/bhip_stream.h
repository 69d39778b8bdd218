// bhip_stream.h -- scalar-unit operand streaming for the path-per-lane kernels at 4 <= d <= 8 (MLinPro<D, bhip_cptr_t>, the
// (nu, H) guide and auxiliary parts of the coefficient row): d x d matrices are uniform over the wave and belong in scalar
// registers, but five of them per step do not fit the file -- they are read in blocks, each block pinned behind the
// arithmetic of the one before it.  (Also part of the source embedded for hipRTC user models.)
#pragma once

namespace bhip {

typedef const __attribute__((address_space(4))) double *bhip_cptr_t;
// "from here on": the scalar loads through `q` cannot be issued before `dep` has been computed -- they are pinned behind the
// phase of the step that produced it, so that the phases' operands follow one another through the scalar registers
BHIP_DEV void bhip_after(bhip_cptr_t &q, double dep)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(q) : "v"(dep));
#else
    (void)dep;
#endif
}
BHIP_DEV void bhip_after(const double *&, double) {}

// o = M v for a column-major d x d matrix read through `Mp`, 4 <= d <= 8: accumulated COLUMN by column (each o[i] still sums
// over j in increasing order) in blocks of at most 16 matrix entries, every block pinned behind the one before it -- an
// 8 x 8 matrix is 128 scalar registers, more than the file has
template <int D, class PTR>
BHIP_DEV void matvec_streamed(PTR Mp, const double *v, double *o)
{
    constexpr int CB = 16 / D > 0 ? 16 / D : 1;
#pragma unroll
    for (int j0 = 0; j0 < D; j0 += CB) {
        PTR q = Mp;
        if (j0 > 0) bhip_after(q, o[D - 1]);
#pragma unroll
        for (int j = j0; j < (j0 + CB < D ? j0 + CB : D); j++)
#pragma unroll
            for (int i = 0; i < D; i++) o[i] = j == 0 ? q[i] * v[0] : __builtin_fma(q[i + D * j], v[j], o[i]);
    }
}

// o = init + M v, the same blocks: the accumulators START from a path-independent vector read through `ip` (the regrouped step of
// the LinPro family at 4 <= d <= 12, bhip_path_kernel.h GUIDE_QF)
template <int D, class PTR>
BHIP_DEV void matvec_streamed_init(PTR Mp, PTR ip, const double *v, double *o)
{
    constexpr int CB = 16 / D > 0 ? 16 / D : 1;
#pragma unroll
    for (int j0 = 0; j0 < D; j0 += CB) {
        PTR q = Mp, qi = ip;
        if (j0 > 0) bhip_after(q, o[D - 1]);
#pragma unroll
        for (int j = j0; j < (j0 + CB < D ? j0 + CB : D); j++)
#pragma unroll
            for (int i = 0; i < D; i++) o[i] = __builtin_fma(q[i + D * j], v[j], j == 0 ? qi[i] : o[i]);
    }
}
// o += M v
template <int D, class PTR>
BHIP_DEV void matvec_streamed_acc(PTR Mp, const double *v, double *o)
{
    constexpr int CB = 16 / D > 0 ? 16 / D : 1;
#pragma unroll
    for (int j0 = 0; j0 < D; j0 += CB) {
        PTR q = Mp;
        bhip_after(q, o[D - 1]);
#pragma unroll
        for (int j = j0; j < (j0 + CB < D ? j0 + CB : D); j++)
#pragma unroll
            for (int i = 0; i < D; i++) o[i] = __builtin_fma(q[i + D * j], v[j], o[i]);
    }
}

}  // namespace bhip
