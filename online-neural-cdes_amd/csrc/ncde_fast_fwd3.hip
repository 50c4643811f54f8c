// ncde_fwd_fast_bf3<.., NLT = 3, ..> of both register-resident shape sets: see ncde_fast_fwd3.h.  The kernel template is taken from
// ncde_fast.hip as it is (NCDE_FAST_KERNELS_ONLY leaves out its host part).
#define NCDE_FAST_KERNELS_ONLY
#include "ncde_fast.hip"
#include "ncde_fast_fwd3.h"

// (still inside the anonymous namespace ncde_fast.hip opened; its closing brace sits in the part left out)
template <int H, int HH, int C, int HP>
NcdeFastFwd3Kernel fwd3_pick(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_fwd_fast_bf3<H, HH, C, 4, I, M, 0, 3, HP>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}
}  // namespace

NcdeFastFwd3Kernel ncde_fast_fwd3(int hidden, int interp, int method, int hp) {
    if (hidden == 32) return hp ? fwd3_pick<32, 32, 20, 1>(interp, method) : fwd3_pick<32, 32, 20, 0>(interp, method);
    if (hidden == 64) return hp ? fwd3_pick<64, 64, 4, 1>(interp, method) : fwd3_pick<64, 64, 4, 0>(interp, method);
    return nullptr;
}
