// experiment.hpp -- guard for the timing-only ablation switches used by profiles/exp/ (NOT product behaviour).
//
// TFHE_ABL_NOKEY / NOLDS / NOFFT / TPB_DPP / KM_NOBARRIER / ROUND_LSB remove, replace or falsify parts of the kernels so that the cost of what
// is left can be measured (profiles/exp/logs/r2h_ab_energy_decomposition.log); the results of such a build are
// WRONG BY CONSTRUCTION.  They can only be switched on together with -DTFHE_EXPERIMENT (what
// profiles/exp/build_variants.sh passes); such a library reports itself as "hip-gfx950-EXPERIMENT" through
// tfhe_hip_name() and rs_tfhe_amd refuses to load it unless TFHE_HIP_ALLOW_EXPERIMENT=1 is set.
#pragma once
#if (defined(TFHE_ABL_NOKEY) && TFHE_ABL_NOKEY) || (defined(TFHE_ABL_NOLDS) && TFHE_ABL_NOLDS) || \
    (defined(TFHE_ABL_NOFFT) && TFHE_ABL_NOFFT) || (defined(TFHE_ABL_TPB_DPP) && TFHE_ABL_TPB_DPP) || \
    (defined(TFHE_ABL_KM_NOBARRIER) && TFHE_ABL_KM_NOBARRIER) || (defined(TFHE_ABL_LAT) && TFHE_ABL_LAT) || \
    (defined(TFHE_ABL_SL) && TFHE_ABL_SL) || (defined(TFHE_ABL_ROUND_LSB) && TFHE_ABL_ROUND_LSB)
#ifndef TFHE_EXPERIMENT
#error "TFHE_ABL_* are timing-only experiment switches (results wrong by construction): build with -DTFHE_EXPERIMENT (profiles/exp/build_variants.sh)"
#endif
#define TFHE_ABLATED 1
#else
#define TFHE_ABLATED 0
#endif
#ifndef TFHE_ABL_NOKEY
#define TFHE_ABL_NOKEY 0
#endif
#ifndef TFHE_ABL_NOLDS
#define TFHE_ABL_NOLDS 0
#endif
#ifndef TFHE_ABL_NOFFT
#define TFHE_ABL_NOFFT 0
#endif
#ifndef TFHE_ABL_TPB_DPP
#define TFHE_ABL_TPB_DPP 0
#endif
#ifndef TFHE_ABL_LAT  // latency kernels. wide: bit 0 no exchange of partial products, bit 1 no rotated reads / digits, bit 2 no update;
                      // wide2: 8 no forward FFT, 16 no inverse FFT, 32 no MAC phase, 64 no key loads in the loop, 128 no digit preparation
#define TFHE_ABL_LAT 0
#endif
#ifndef TFHE_ABL_ROUND_LSB  // mutation, not timing: round_product<false> returns one LSB too much on ~1/1024 of the words, so that
                            // the parity suite can be shown to notice (the l = 1 exact-regime tests must FAIL on this build)
#define TFHE_ABL_ROUND_LSB 0
#endif
#ifndef TFHE_ABL_KM_NOBARRIER  // matrix-core key switch without its per-step barrier (races: wrong results)
#define TFHE_ABL_KM_NOBARRIER 0
#endif
#ifndef TFHE_ABL_SL  // column-sliced key switch: bit 0 no key DMA, 1 no barrier, 2 (first form only) no a_bar restage after the first,
                     // 3 no LDS row reads (the additions take registers instead), 4 no additions (rows read and dropped)
#define TFHE_ABL_SL 0
#endif
