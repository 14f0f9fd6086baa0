/* ABI version of include/mmsa.h (its own header so that the library's sources can include it without the declarations).
 * Bumped whenever an entry point changes its arguments or their meaning.  101 (round 5): round 4 added arguments to mmsa_gemm_split3,
 * mmsa_convnext_mlp_fused and the attention entries and removed mmsa_gemm_next_extras / mmsa_debug_*_flavour / mmsa_dwconv7_ln without bumping it.
 * 102 (round 6): mmsa_convnext_mlp_fused takes clamp_max; new entry mmsa_msda_fused_planes. */
#ifndef MMSA_VERSION_H
#define MMSA_VERSION_H
#define MMSA_ABI_VERSION 102
#endif
