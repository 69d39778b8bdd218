// bhip_inst.hip -- instantiates the fused path kernel for ONE target model per translation unit
// (compiled once per model with -DBHIP_INST=<n> so the build parallelises).
#include <hip/hip_runtime.h>
// BHIP_FUSED (second compilation of the d <= 3 models, -ffp-contract=fast): the same kernels with a*b + c contracted to one
// fused multiply-add wherever the compiler finds it -- BHIP_OPT_FUSED_ARITHMETIC, results to a stated tolerance instead of bit
// for bit -- in a namespace of their own (the macro renames the namespace of every header included below).
#ifdef BHIP_FUSED
#define bhip bhip_fused
#endif
#include "bhip_path_kernel.h"
#include "bhip_chain_kernel.h"
#include "bhip_pc_kernel.h"
#include "bhip_guide_kernel.h"

namespace bhip {
#if BHIP_INST == 0
launch_fn get_launch_ou(int gk, int mo, int noise, int fl) { return get_launch<MOU>(gk, mo, noise, fl); }
#elif BHIP_INST == 1
launch_fn get_launch_linpro1(int gk, int mo, int noise, int fl) { return get_launch<MLinPro<1>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_linpro1(int noise, int fl) { return get_launch_ppr<MLinPro<1>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_linpro1(int mo) { return get_guide_launch<MLinPro<1>>(mo); }
#endif
#elif BHIP_INST == 2
launch_fn get_launch_linpro2(int gk, int mo, int noise, int fl) { return get_launch<MLinPro<2>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_linpro2(int noise, int fl) { return get_launch_ppr<MLinPro<2>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_linpro2(int mo) { return get_guide_launch<MLinPro<2>>(mo); }
#endif
#elif BHIP_INST == 3
launch_fn get_launch_linpro3(int gk, int mo, int noise, int fl) { return get_launch<MLinPro<3>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_linpro3(int noise, int fl) { return get_launch_ppr<MLinPro<3>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_linpro3(int mo) { return get_guide_launch<MLinPro<3>>(mo); }
#endif
#elif BHIP_INST == 4
launch_fn get_launch_fhn(int gk, int mo, int noise, int fl) { return get_launch<MFHN>(gk, mo, noise, fl); }
#elif BHIP_INST == 5
launch_fn get_launch_nclar(int gk, int mo, int noise, int fl) { return get_launch<MNCLAR>(gk, mo, noise, fl); }
#elif BHIP_INST == 6
launch_fn get_launch_intdiff(int gk, int mo, int noise, int fl) { return get_launch<MIntDiff>(gk, mo, noise, fl); }
#elif BHIP_INST == 7
launch_fn get_launch_lorenz(int gk, int mo, int noise, int fl) { return get_launch<MLorenz>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_lorenz(int noise, int fl) { return get_launch_ppr<MLorenz>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_lorenz(int mo) { return get_guide_launch<MLorenz>(mo); }
#endif
#elif BHIP_INST == 8
launch_fn get_launch_fhn2(int gk, int mo, int noise, int fl) { return get_launch<MFHN2>(gk, mo, noise, fl); }
#elif BHIP_INST == 9
launch_fn get_launch_pendulum(int gk, int mo, int noise, int fl) { return get_launch<MPendulum>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_pendulum(int noise, int fl) { return get_launch_ppr<MPendulum>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_pendulum(int mo) { return get_guide_launch<MPendulum>(mo); }
#endif
#elif BHIP_INST == 10
launch_fn get_launch_wiener1(int gk, int mo, int noise, int fl) { return get_launch<MWiener<1>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_wiener1(int noise, int fl) { return get_launch_ppr<MWiener<1>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_wiener1(int mo) { return get_guide_launch<MWiener<1>>(mo); }
#endif
launch_fn get_launch_wiener2(int gk, int mo, int noise, int fl) { return get_launch<MWiener<2>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_wiener2(int noise, int fl) { return get_launch_ppr<MWiener<2>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_wiener2(int mo) { return get_guide_launch<MWiener<2>>(mo); }
#endif
launch_fn get_launch_wiener3(int gk, int mo, int noise, int fl) { return get_launch<MWiener<3>>(gk, mo, noise, fl); }
#ifndef BHIP_FUSED
launch_fn get_launch_ppr_wiener3(int noise, int fl) { return get_launch_ppr<MWiener<3>>(noise, fl); }
#endif
#ifndef BHIP_FUSED
guide_launch_fn get_guide_launch_wiener3(int mo) { return get_guide_launch<MWiener<3>>(mo); }
#endif
#elif BHIP_INST == 11
launch_fn get_launch_mid4(int gk, int noise, int fl) { return get_launch_mid<MLinPro<4, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 12
launch_fn get_launch_mid5(int gk, int noise, int fl) { return get_launch_mid<MLinPro<5, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 13
launch_fn get_launch_mid6(int gk, int noise, int fl) { return get_launch_mid<MLinPro<6, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 14
launch_fn get_launch_mid7(int gk, int noise, int fl) { return get_launch_mid<MLinPro<7, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 15
launch_fn get_launch_mid8(int gk, int noise, int fl) { return get_launch_mid<MLinPro<8, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 16
launch_fn get_launch_mid9(int gk, int noise, int fl) { return get_launch_mid<MLinPro<9, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 17
launch_fn get_launch_mid10(int gk, int noise, int fl) { return get_launch_mid<MLinPro<10, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 18
launch_fn get_launch_mid11(int gk, int noise, int fl) { return get_launch_mid<MLinPro<11, bhip_cptr_t>>(gk, noise, fl); }
#elif BHIP_INST == 19
launch_fn get_launch_mid12(int gk, int noise, int fl) { return get_launch_mid<MLinPro<12, bhip_cptr_t>>(gk, noise, fl); }
#else
#error "BHIP_INST must be 0..19"
#endif
}  // namespace bhip
