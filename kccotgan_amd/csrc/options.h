// Process-wide options of the library (include/kccot.h: kccot_set_option documents each one).  Kernel dispatch reads
// them through opt(): one relaxed atomic load, no environment lookup on any call path.
#pragma once

namespace kccot {

enum Option {
    OPT_GRAM_F32 = 0,            // "gram_f32"
    OPT_APPLY_F32,               // "apply_f32"
    OPT_COST_TILED,              // "cost_tiled"
    OPT_COST_TILE256,            // "cost_tile256"
    OPT_COST_BLOCKED,            // "cost_blocked"
    OPT_APPLY_M256,              // "apply_m256"
    OPT_APPLY_ONE_LAUNCH,        // "apply_one_launch"
    OPT_APPLY_Q256,              // "apply_q256"
    OPT_SK_SHORTCUT,             // "sinkhorn_shortcut"
    OPT_SK_FUSED,                // "sinkhorn_fused"
    OPT_SK_FUSED_MAX_N,          // "sinkhorn_fused_max_n"
    OPT_SK_LPR,                  // "sinkhorn_lanes_per_line"
    OPT_SK_COOP,                 // "sinkhorn_coop"
    OPT_SK_COOP_XCD,             // "sinkhorn_coop_xcd"
    OPT_SK_COOP_MAX_WG,          // "sinkhorn_coop_max_wg"
    OPT_SMOOTH_STREAM,           // "smooth_stream"
    OPT_SMOOTH_GENERIC,          // "smooth_generic"
    OPT_SMOOTH_FUSED_TW,         // "smooth_fused_tw"
    OPT_SMOOTH_BWD_FOLD,         // "smooth_bwd_fold"
    OPT_SMOOTH_FUSED3,           // "smooth_fused3"
    OPT_COUNT
};

int opt(Option o);

}  // namespace kccot
