// ncde_adj_fast3 (ncde_fast.hip: the specialised chain + gradient-wave adjoint of H = HH = 32, C = 20) for the layer counts besides
// BASELINE's nl = 3: the reference's hyper-parameter range is num_layers in [1, 4] (experiments/configurations/configurations.json5:36)
// and a model one layer away used to fall to the batch-tiled family (2.3 - 3 x the time: profiles/r04_shape_sweep_perf.txt).
// The kernel template is taken from ncde_fast.hip as it is (NCDE_FAST_KERNELS_ONLY leaves out its host part).
#define NCDE_FAST_KERNELS_ONLY
#include "ncde_fast.hip"
#include "ncde_fast_nl.h"

// (still inside the anonymous namespace ncde_fast.hip opened; its closing brace sits in the part left out)
template <int NL, int DISC>
NcdeFastNlKernel nl_pick(int interp, int method, int hp) {
#define NCDE_PICK(I, M) \
    if (interp == I && method == M) return hp == 2 ? ncde_adj_fast3<NL, 20, I, M, 0, DISC, 2> : ncde_adj_fast3<NL, 20, I, M, 0, DISC, 0>;
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_RK4_38)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_MIDPOINT)
    NCDE_PICK(NCDE_INTERP_LINEAR, NCDE_EULER)
    if constexpr (NL < 4) {
        NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_RK4_38)
        NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_MIDPOINT)
        NCDE_PICK(NCDE_INTERP_CUBIC, NCDE_EULER)
    }
#undef NCDE_PICK
    return nullptr;
}
}  // namespace

NcdeFastNlKernel ncde_fast_adj3_nl(int n_layers, int interp, int method, int hp, bool discrete) {
    if (hp != 0 && hp != 2) return nullptr;
    if (ncde_fast_adj3_nl_lds(n_layers, interp, hp) > (size_t)kLdsLimit) return nullptr;
    switch (n_layers) {
        case 1: return discrete ? nl_pick<1, 1>(interp, method, hp) : nl_pick<1, 0>(interp, method, hp);
        case 2: return discrete ? nl_pick<2, 1>(interp, method, hp) : nl_pick<2, 0>(interp, method, hp);
        case 4: return discrete ? nl_pick<4, 1>(interp, method, hp) : nl_pick<4, 0>(interp, method, hp);
        default: return nullptr;
    }
}
size_t ncde_fast_adj3_nl_lds(int n_layers, int interp, int hp) {
    switch (n_layers) {
        case 1: return adj3_lds_bytes<1, 20>(interp, hp);
        case 2: return adj3_lds_bytes<2, 20>(interp, hp);
        case 4: return adj3_lds_bytes<4, 20>(interp, hp);
        default: return (size_t)-1;
    }
}
const char* ncde_fast_adj3_nl_name(int n_layers, int hp, bool discrete) {
    static const char* kNames[3][2][2] = {
        {{"ncde_adj_fast3<H32,HH32,C20,NL1,chain+grad,bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL1,chain+grad,bf16x3,discrete>"},
         {"ncde_adj_fast3<H32,HH32,C20,NL1,chain+grad,fwd-side fp16x2 + bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL1,chain+grad,fwd-side fp16x2 + bf16x3,discrete>"}},
        {{"ncde_adj_fast3<H32,HH32,C20,NL2,chain+grad,bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL2,chain+grad,bf16x3,discrete>"},
         {"ncde_adj_fast3<H32,HH32,C20,NL2,chain+grad,fwd-side fp16x2 + bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL2,chain+grad,fwd-side fp16x2 + bf16x3,discrete>"}},
        {{"ncde_adj_fast3<H32,HH32,C20,NL4,chain+grad,bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL4,chain+grad,bf16x3,discrete>"},
         {"ncde_adj_fast3<H32,HH32,C20,NL4,chain+grad,fwd-side fp16x2 + bf16x3>", "ncde_adj_fast3<H32,HH32,C20,NL4,chain+grad,fwd-side fp16x2 + bf16x3,discrete>"}}};
    const int i = n_layers == 1 ? 0 : (n_layers == 2 ? 1 : 2);
    return kNames[i][hp == 2 ? 1 : 0][discrete ? 1 : 0];
}
