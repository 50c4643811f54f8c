// Small channel counts on the register-resident kernel set (round 6): (H, HH) = (32, 32) with C = NCDE_FAST_C in
// {4, 8, 12}.  A model with few channels -- CharacterTrajectories has 4, and the reference's DEFAULT widths are hidden_hidden_dim = 15,
// num_layers = 3 (src/ncde/ncde.py:47-48) -- used to be zero-padded to the C = 20 instantiations and paid for 10 output tiles per wave
// where 2 carry data: (4, 32, 15, 3) cost what (20, 32, 32, 3) costs, a tenth of the flops (profiles/r05_shape_sweep_perf.txt).
// The kernel templates are ncde_fast.hip's own (NCDE_FAST_KERNELS_ONLY); this file is compiled once per channel count (Makefile).
#ifndef NCDE_FAST_C
#error "compile with -DNCDE_FAST_C=4 | 8 | 12"
#endif
#include "ncde_fast_c.h"
#define NCDE_FAST_KERNELS_ONLY
#include "ncde_fast.hip"

// (still inside the anonymous namespace ncde_fast.hip opened; its closing brace sits in the part left out)
constexpr int FC = NCDE_FAST_C;
template <int HP>
NcdeFastCKernel c_fwd_pick(int interp, int method) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return ncde_fwd_fast_bf3<32, 32, FC, 4, I, M, 0, 0, HP>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
#undef NCDE_PICK
    return nullptr;
}
template <int NL, int DISC>
NcdeFastCKernel c_adj_pick(int interp, int method, int hp) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return hp == 2 ? ncde_adj_fast3<NL, FC, I, M, 0, DISC, 2> : ncde_adj_fast3<NL, FC, I, M, 0, DISC, 0>;
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

#define NCDE_CAT_(a, b, c) a##b##c
#define NCDE_CAT(a, b, c) NCDE_CAT_(a, b, c)
#define NCDE_CFN(name) NCDE_CAT(ncde_fast_c, NCDE_FAST_C, name)

NcdeFastCKernel NCDE_CFN(_fwd)(int interp, int method, int hp) { return hp ? c_fwd_pick<1>(interp, method) : c_fwd_pick<0>(interp, method); }
size_t NCDE_CFN(_adj_lds)(int n_layers, int interp, int hp) {
    switch (n_layers) {
        case 1: return adj3_lds_bytes<1, FC>(interp, hp);
        case 2: return adj3_lds_bytes<2, FC>(interp, hp);
        case 3: return adj3_lds_bytes<3, FC>(interp, hp);
        case 4: return adj3_lds_bytes<4, FC>(interp, hp);
        default: return (size_t)-1;
    }
}
NcdeFastCKernel NCDE_CFN(_adj)(int n_layers, int interp, int method, int hp, bool discrete) {
    if (hp != 0 && hp != 2) return nullptr;
    if (NCDE_CFN(_adj_lds)(n_layers, interp, hp) > (size_t)kLdsLimit) return nullptr;
    switch (n_layers) {
        case 1: return discrete ? c_adj_pick<1, 1>(interp, method, hp) : c_adj_pick<1, 0>(interp, method, hp);
        case 2: return discrete ? c_adj_pick<2, 1>(interp, method, hp) : c_adj_pick<2, 0>(interp, method, hp);
        case 3: return discrete ? c_adj_pick<3, 1>(interp, method, hp) : c_adj_pick<3, 0>(interp, method, hp);
        case 4: return discrete ? c_adj_pick<4, 1>(interp, method, hp) : c_adj_pick<4, 0>(interp, method, hp);
        default: return nullptr;
    }
}
