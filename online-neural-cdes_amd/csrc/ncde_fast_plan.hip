// PLAN = 1 instantiations of the shape-specialised kernels (ncde_fast.hip): the general time axis of cdeint -- any increasing output
// times, any step_size, user knot grids (torchdiffeq/_impl/solvers.py:78-119, 166-172; adjoint.py:116-133) -- used to run on the
// batch-tiled / generic families only: rk4 with step_size 0.5 at cfg2's shape cost 5 x the default-axis step for 2 x the work
// (VERDICT round 3, item 6).  The kernel templates are taken from ncde_fast.hip as they are (NCDE_FAST_KERNELS_ONLY).
#define NCDE_FAST_KERNELS_ONLY
#include "ncde_fast.hip"
#include "ncde_fast_plan.h"

// (still inside the anonymous namespace ncde_fast.hip opened)
template <int H, int HH, int C, int HP>
NcdeFastPlanKernel plan_fwd_pick(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_fwd_fast_bf3<H, HH, C, 4, I, M, 0, 0, HP, 1>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}
template <int HP>
NcdeFastPlanKernel plan_adj3_pick(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_adj_fast3<3, 20, I, M, 0, 0, HP, 1>;
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

NcdeFastPlanKernel ncde_fast_plan_fwd(int shape, int interp, int method, int hp) {
    if (shape == 0) return hp ? plan_fwd_pick<32, 32, 20, 1>(interp, method) : plan_fwd_pick<32, 32, 20, 0>(interp, method);
    if (shape == 1) return hp ? plan_fwd_pick<64, 64, 4, 1>(interp, method) : plan_fwd_pick<64, 64, 4, 0>(interp, method);
    return nullptr;
}
size_t ncde_fast_plan_adj3_lds(int n_layers, int interp, int method, int hp) {
    if (n_layers != 3) return (size_t)-1;
    const int S = method == NCDE_RK4_38 ? 4 : (method == NCDE_MIDPOINT ? 2 : 1);
    return adj3_lds_bytes<3, 20>(interp, hp, S);
}
NcdeFastPlanKernel ncde_fast_plan_adj3(int n_layers, int interp, int method, int hp) {
    if (n_layers != 3 || (hp != 0 && hp != 2)) return nullptr;
    if (ncde_fast_plan_adj3_lds(n_layers, interp, method, hp) > (size_t)kLdsLimit) return nullptr;
    return hp == 2 ? plan_adj3_pick<2>(interp, method) : plan_adj3_pick<0>(interp, method);
}
