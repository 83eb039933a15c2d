// Row-owner engine, 16-rows-per-wave variant ("x3b"): the same chain, arithmetic, weight-stream ring and parameter blob as
// rowowner.hpp, re-tiled so that TWO waves share each SIMD.
//
// WHY.  The 32-row kernel needs ~350 registers per wave (planes 128 + accumulators 128 + ...), i.e. one wave per SIMD, and
// at one wave per SIMD every non-MFMA instruction is exposed: elimination runs (tools/x3_probe.hip) put its MFMA + VALU
// floor at 0.50 of the fp16/3 bound with all memory traffic removed - row scaling, the fp16 plane split, LayerNorm and the
// accumulator moves of ten phase transitions run while the matrix pipe idles.  Here a wave owns 16 rows on
// v_mfma_f32_16x16x32_f16 (same FLOP rate per cycle as 32x32x16): planes 64 + accumulators 64 registers, ~210 in all, so a
// workgroup is 8 waves = 128 rows and the two waves of a SIMD overlap one's VALU / LDS latency with the other's MFMAs.
// The price: an A fragment (1 KB) now feeds a 16-cycle MFMA instead of a 32-cycle one, so LDS fragment traffic doubles
// (~66 % of the LDS read rate at full MFMA speed); fragment sets, chunks, ring and parameter blob are unchanged in size.
//
// GEOMETRY.  lane l: row q = l & 15, group g = l >> 4.  State x[16] (f32x4 tiles): x[T][r] = feature 16 T + 4 g + r.
//   A operand: lane (p, g) holds A[p][k = 8 g + j];  B: lane (q, g) holds B[k = 8 g + j][q];  D: D[p = 4 g + r][q].
//   A k-step is 32 features = tiles 2 ks and 2 ks + 1: B element j = x[2 ks + (j >> 2)][j & 3], i.e. k position 8 g + j is
//   feature 32 ks + 16 (j >> 2) + 4 g + (j & 3); the host packs the A fragments with the same permutation
//   (amdrec/weights.py x3b_frags).  Stream orders mirror rowowner.hpp with 16-feature output tiles:
//     gemm256 : for ks < 8: for tile pair tp < 8: {A_h(2tp, ks), A_l(2tp, ks), A_h(2tp+1, ks), A_l(2tp+1, ks)}
//     ffn step t: for u < 8: stage 1 {W_1 tiles 2t, 2t+1 at ks = u}, then stage 2 {W_2 tiles 2u, 2u+1 at k-step t-1}
//     heads, per task and hidden tile (32): stage 1 as above (8 groups), stage 2 {W_2 tiles 0,1}, {tiles 2,3} at k-step t
#pragma once
#include "rowowner.hpp"

#define AMDREC_X3B_NAMESPACE x3b
#define AMDREC_X3B_WAVES 8
#include "rowowner16_impl.hpp"
#undef AMDREC_X3B_NAMESPACE
#undef AMDREC_X3B_WAVES

#define AMDREC_X3B_NAMESPACE x3b4
#define AMDREC_X3B_WAVES 4
#include "rowowner16_impl.hpp"
#undef AMDREC_X3B_NAMESPACE
#undef AMDREC_X3B_WAVES
