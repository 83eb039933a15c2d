// "Row-owner" fp16x3 engine of the TransformerRanker forward (transformer_ranker.py:332-380, eval mode) for gfx950.
//
// WHAT.  One kernel runs the whole chain  gather/x0 -> n x { x = LN(x + W_ov x + b) ; x = LN(x + W_2 relu(W_1 x + b_1)
// + b_2) } -> 3 x cross -> 3 heads -> logits  for 128 candidate rows per workgroup (4 waves x 32 rows), with every
// activation living in REGISTERS between the GEMMs: nothing but the input rows and the logits touches HBM.
//
// ARITHMETIC ("x3": fp32 in, fp32 out, fp32-level error on the 16-bit matrix pipe).  Every fp32 operand is multiplied
// by a power of two (exact) and split into two fp16 planes  v = h + l + e,  h = RN16(v), l = RN16(v - h),
// |e| <= 2^-22 |v|  (fp16 has 11 significant bits; the second rounding is taken of an exact fp32 difference).  A product
// a*b is evaluated as  ah*bh + ah*bl + al*bh  by three v_mfma_f32_32x32x16_f16 (an fp16 x fp16 product is exact in
// fp32; the MFMA accumulates in fp32); the dropped al*bl term is <= 2^-22 |ab|.  Split error measured on the host
// against float64 (tools/split_accuracy.py): rms 7.6e-8 of a K = 256 dot product whose fp32 fma-chain evaluation
// itself is off by rms 2.9e-7 - the same level as the round-1 six-product bf16 split (6.0e-8), at half the MFMAs.
//  * scaling: weights carry one power of two per matrix (host, max |w| -> [2^12, 2^13)); activations one power of two
//    per ROW, from the row's own max |x| (-> [2^12, 2^13)), recomputed in registers before each GEMM; a hidden tile
//    (FFN, heads) one power of two per row from the bound |relu(w_j . x + b_j)| <= ||w_j||_2 ||x||_2 + |b_j| with
//    ||x||_2 <= 16 max|x| (-> below 2^14).  fp16 overflow (65504) is therefore impossible for finite inputs; elements
//    more than 2^15 below the row maximum lose RELATIVE precision only (absolute error <= 2^-25 of a scaled unit:
//    2^-37 of the row maximum).
//
// HOW (cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's operand").  C[p][q] = sum_k A[p][k] B[k][q]
// with A = weights (p = output feature), B = activations (q = row): in the accumulator a lane owns ROW q = lane & 31 and
// its registers hold features p = 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of tile i.  The next GEMM sums over exactly
// that feature index, so registers 8 s .. 8 s + 7 of tile i, converted to fp16, ARE the B fragment of k-step 2 i + s:
// no LDS, no lane movement.  The k order inside such a step is permuted (element j of lane half h is feature
// 16 ks + 8 (j >> 2) + 4 h + (j & 3)); the host packs the weight fragments with the same permutation.
//  * a wave keeps: the planes of its rows (16 k-steps x 2 planes x 4 VGPRs = 128), the accumulators of all 256 output
//    features (8 tiles x 16 = 128), one hidden tile (16) and its planes (16): ~350 of the 512 registers a wave may use
//    at one wave per SIMD (256 threads per workgroup, one workgroup per CU);
//  * the residual is folded into the accumulator's initial value ((x + b) * scale), so x itself is dead during a GEMM;
//  * LayerNorm, the cross product and the heads' final dot are in-register (row statistics: in-lane + one lane-half
//    exchange); parameters are read with scalar loads (uniform addresses, constant address space: they do not touch
//    the vector-memory counter the ring relies on).
//  * WEIGHT STREAM: the host packs all weights of the chain as ONE linear stream of 1 KB "fragment sets" (64 lanes x
//    16 B = the A operand of one MFMA k-step of one 32-feature tile and plane, already in lane order) in exactly the
//    order the kernel consumes them.  The stream flows through an LDS ring of 8 x 16 KB chunks filled by LDS-DMA
//    (global_load_lds, 1 KB per wave instruction, source and destination both linear: no swizzle needed, a fragment
//    read is a conflict-free ds_read_b128 at base + lane * 16).  Five chunks are in flight; a chunk is certified
//    (counted s_waitcnt vmcnt + one s_barrier per 16 KB) one chunk ahead of its first read, and fragments are
//    double-buffered in registers one micro-step (4 fragment sets = 6 MFMAs) ahead of their MFMAs.
#pragma once
#include "common.hpp"

namespace amdrec {
namespace x3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int FRAG_BYTES = 1024;                 // one fragment set: 64 lanes x 8 fp16
constexpr int CHUNK_FRAGS = 16;
constexpr int CHUNK_BYTES = CHUNK_FRAGS * FRAG_BYTES;
#ifndef AMDREC_X3_NBUF
#define AMDREC_X3_NBUF 7
#endif
constexpr int NBUF = AMDREC_X3_NBUF;             // ring chunks (7 x 16 KB = 112 KB; 6, 8 and 9 measured the same)
constexpr int PARAM_FLOATS = 11264;              // LDS parameter area behind the ring: 44 KB (the reference architecture needs 41.5)
constexpr int RING_BYTES = NBUF * CHUNK_BYTES;
constexpr int DEPTH = NBUF - 3;                  // chunks in flight beyond the certified one (NBUF >= DEPTH + 3)
constexpr int ROWS_PER_WAVE = 32, WAVES = 4, ROWS_PER_WG = ROWS_PER_WAVE * WAVES;
constexpr int TARGET_EXP = 12;                   // scaled row / matrix maxima lie in [2^12, 2^13)

enum PhaseType { PH_ATTN_LN = 0, PH_FFN_LN = 1, PH_CROSS = 2, PH_HEADS = 3 };

struct Phase {
    int type;
    int n_steps;            // FFN: d_ff / 32 hidden tiles; HEADS: head_h1 / 32 hidden tiles per task
    int n_tasks;            // HEADS
    int pad_;
    // offsets (in floats) into the parameter blob, which is DMA'd into LDS once per workgroup (see Program::params)
    int b1;                 // ATTN/CROSS: bias [256]; FFN: b_1 [d_ff]; HEADS: stacked b_1 [n_tasks * head_h1]
    int b2;                 // FFN: b_2 [256]
    int gamma;              // LayerNorm weight / bias [256] (ATTN, FFN)
    int beta;
    float sw1, sw2;         // power-of-two scales of the packed weight planes (W_ov / W_1 / W_c / head W_1; W_2 / head W_2)
    float hn, hb;           // FFN / HEADS hidden bound: |relu(w_j . x + b_j)| <= hn * (2^13 / row scale) + hb, with
                            // hn = 16 max_j ||w_j||_2 (||x||_2 <= 16 max|x| over 256 features), hb = max_j |b_j|
    float ln_eps;
    float pad2_;
};
constexpr int MAX_PHASES = 20;          // 8 encoder layers x 2 + 3 cross + heads; the whole Program travels as a kernel argument
struct Program {
    int n_phases;
    int total_chunks;                   // length of the weight stream in 16 KB chunks
    const unsigned char* stream;        // packed fragment sets
    // All biases / LayerNorm weights / head vectors of the chain as ONE float blob (amdrec/weights.py pack_x3_params:
    // per layer b_ov, gamma1, beta1, b_1, b_2, gamma2, beta2; per cross layer its bias; heads b_1, then per task b_2, w_3,
    // b_3 padded to 4).  The kernel copies it into LDS behind the ring at start; a lane then reads the 4 parameters of
    // its features with ONE ds_read_b128 (lanes of a half share the address: a broadcast) - scalar loads needed a select
    // per element for the lane half and so many SGPRs that ~600 of them spilled at every phase transition.
    const float* params;
    int n_params;                       // floats, multiple of 1024 (padded), <= PARAM_FLOATS
    int hb2[4];                         // HEADS, per task: offsets of b_2 [64], w_3 [64], b_3 [4]
    int hw3[4];
    int hb3[4];
    Phase ph[MAX_PHASES];
};

// input rows: either a dense fp32 matrix X[rows][256] (the projection GEMM's output) or the cached form
// x0[r] = ad_proj_cache[ad row of r] + U[user of r]  (layers.hip proj_gather_kernel: same addends, same order)
struct Input {
    const float* X;            // dense [rows][ldx] or nullptr
    long long ldx;
    const float* cache;        // [n_cache][ldc]
    long long ldc, n_cache;
    const long long* rowmap;   // candidate -> cache row (may be nullptr: identity)
    const float* U;            // [n_users][256]
    long long row_base;        // global index of row 0 of this launch (for the user index)
    int rowdiv;
};

// ---- parameters: the 4 values of features f0 + 8 g + 4 h + {0..3} of array `off` = one ds_read_b128 from the LDS copy
// of the blob (`pb` = LDS address of the blob + 16 h bytes, per lane; every lane of a half reads the same address)
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ f32x4 param4(lds_cfloat* pb, int off, int f0, int g) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(pb + off + f0 + 8 * g);
}

// ---- the ring --------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte;

// Elimination switches for tools/x3_probe.hip (0 in the product): 1 = no weight DMA and no wait for it, 2 = no
// per-chunk barrier, 4 = no fragment reads from LDS (stale registers), 8 = no hidden-tile conversion (stale planes).
#ifndef AMDREC_X3_DBG
#define AMDREC_X3_DBG 0
#endif
constexpr int DBG = AMDREC_X3_DBG;

struct Ring {
    const unsigned char* gsrc;   // stream + wave * 4 KB + lane * 16 (per lane)
    lds_byte* lds_dma;           // ring + wave * 4 KB (wave-uniform: the DMA adds lane * 16 itself)
    lds_byte* lds_rd;            // ring + lane * 16 (fragment reads)
    int issued;                  // chunks whose DMA has been issued
    int total;                   // chunks in the stream
    uint32_t rpos;               // byte position (mod RING_BYTES) of the next fragment set to read
    int rfrags;                  // fragment sets read so far

    __device__ __forceinline__ void issue() {
        if (DBG & 1) { ++issued; return; }
        const int c = issued < total ? issued : total - 1;        // past the end: harmless re-load of the last chunk into a free slot
        const unsigned char* src = gsrc + (size_t)c * CHUNK_BYTES;
        lds_byte* dst = lds_dma + (uint32_t)(issued % NBUF) * CHUNK_BYTES;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * FRAG_BYTES),
                                             (__attribute__((address_space(3))) void*)(dst + u * FRAG_BYTES), 16, 0, 0);
        ++issued;
    }
    // Certify the chunk after the one whose first fragments are about to be read: my share of it has landed (all
    // younger DMAs may still be in flight), everyone's share after the barrier; then refill the slot of the chunk
    // two behind (every wave has issued all MFMAs that consumed it: it is past that chunk's last fragment read).
    __device__ __forceinline__ void certify_next() {
        if (!(DBG & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (DEPTH - 1)) : "memory");
        if (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        issue();
    }
    __device__ __forceinline__ void start(const unsigned char* stream, int total_chunks, lds_byte* lds, int wave, int lane) {
        gsrc = stream + wave * 4 * FRAG_BYTES + lane * 16;
        lds_dma = lds + wave * 4 * FRAG_BYTES;
        lds_rd = lds + lane * 16;
        issued = 0;
        total = total_chunks;
        rpos = 0;
        rfrags = 0;
#pragma unroll
        for (int c = 0; c < DEPTH + 1; ++c) issue();               // chunks 0 .. DEPTH
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DEPTH) : "memory");   // chunk 0 (mine)
        __builtin_amdgcn_s_barrier();
    }
    // read the next N fragment sets (N = 2 or 4; a group never straddles a chunk: chunk = 16 sets, groups are aligned)
    template <int N>
    __device__ __forceinline__ void read(f16x8 (&f)[N]) {
        if ((rfrags & (CHUNK_FRAGS - 1)) == 0) certify_next();     // first read of a chunk: certify the one after it
        const lds_byte* a = lds_rd + rpos;
        if ((DBG & 4) && rfrags != 0) {
#pragma unroll
            for (int u = 0; u < N; ++u) asm volatile("" : "+v"(f[u]));   // whatever the registers hold
        } else {
#pragma unroll
            for (int u = 0; u < N; ++u)
                f[u] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(a + u * FRAG_BYTES);
        }
        rpos += N * FRAG_BYTES;
        if (rpos >= RING_BYTES) rpos -= RING_BYTES;
        rfrags += N;
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

__device__ __forceinline__ f32x16 mfma(const f16x8& a, const f16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// acc += A * B with A = (ah, al) and B = (bh, bl): ah*bl, al*bh first (small terms), then ah*bh
__device__ __forceinline__ void mac3(f32x16& acc, const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl) {
    acc = mfma(ah, bl, acc);
    acc = mfma(al, bh, acc);
    acc = mfma(ah, bh, acc);
}

// ---- per-row power-of-two scale from the row's max |x| ------------------------------------------------------------
// returns s = 2^(TARGET_EXP - floor(log2 max)) and inv = 1 / s (both exact powers of two); max == 0 or denormal -> clamped
__device__ __forceinline__ void row_scale(const f32x16 (&x)[8], float& s, float& inv) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, __builtin_fabsf(x[i][r]));      // NaN is dropped by fmaxf: it re-enters through the planes
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);
    // rows below 2^-40 (incl. all-zero rows) are scaled as if their maximum were 2^-40: (x + bias) * s * 2^sw must stay
    // finite in fp32, and their elements are far below any bias anyway
    eb = eb < 87 ? 87 : (eb > 250 ? 250 : eb);
    s = __uint_as_float((uint32_t)(127 + TARGET_EXP + 127 - eb) << 23);
    inv = __uint_as_float((uint32_t)(eb - TARGET_EXP) << 23);
}
// power of two sh with bound * sh in [2^13, 2^14)  (bound > 0 finite; tiny bounds clamped)
__device__ __forceinline__ float hidden_scale(float bound) {
    int eb = (int)((__float_as_uint(bound) >> 23) & 0xffu);
    eb = eb < 40 ? 40 : (eb > 250 ? 250 : eb);
    return __uint_as_float((uint32_t)(127 + 13 + 127 - eb) << 23);
}

// fp16 planes of the 8 registers r0 .. r0 + 7 of one tile, scaled by s (a power of two): the B fragment of one k-step
__device__ __forceinline__ void split8(const f32x16& t, int r0, float s, f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = t[r0 + j] * s;
        const _Float16 hh = (_Float16)v;
        h[j] = hh;
        l[j] = (_Float16)(v - (float)hh);
    }
}
__device__ __forceinline__ void split_rows(const f32x16 (&x)[8], float s, f16x8 (&xh)[16], f16x8 (&xl)[16]) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) split8(x[i], 8 * sp, s, xh[2 * i + sp], xl[2 * i + sp]);
}

// ---- 256 x 256 GEMM over the wave's 32 rows: acc[i] += W[32 i .. 32 i + 31][:] . x ----------------------------------
// stream order: for ks: for i: {A_h(i, ks), A_l(i, ks)}; a micro-step = two tiles of one k-step (4 fragment sets, 6 MFMAs)
__device__ __forceinline__ void gemm256(Ring& ring, const f16x8 (&xh)[16], const f16x8 (&xl)[16], f32x16 (&acc)[8]) {
    f16x8 cur[4], nxt[4];
    ring.read<4>(cur);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
            const bool last = ks == 15 && ip == 3;
            if (!last) ring.read<4>(nxt);
            acc[2 * ip] = mfma(cur[0], xl[ks], acc[2 * ip]);
            acc[2 * ip + 1] = mfma(cur[2], xl[ks], acc[2 * ip + 1]);
            acc[2 * ip] = mfma(cur[1], xh[ks], acc[2 * ip]);
            acc[2 * ip + 1] = mfma(cur[3], xh[ks], acc[2 * ip + 1]);
            acc[2 * ip] = mfma(cur[0], xh[ks], acc[2 * ip]);
            acc[2 * ip + 1] = mfma(cur[2], xh[ks], acc[2 * ip + 1]);
            if (!last) {
#pragma unroll
                for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
            }
        }
}

// ---- in-register LayerNorm over the 256 features of each row (transformer_ranker.py:149, :153; two-pass) -----------
__device__ __forceinline__ void layer_norm(f32x16 (&y)[8], lds_cfloat* pb, int gamma, int beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += y[i][r];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = y[i][r] - mean;
            q += d * d;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ga = param4(pb, gamma, 32 * i, g), be = param4(pb, beta, 32 * i, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][4 * g + e] = (y[i][4 * g + e] - mean) * rstd * ga[e] + be[e];
        }
}

// Planes of the rows AND the accumulators' initial value, tile by tile: after tile i both have been produced x[i] is dead,
// so the three register sets (x, planes, accumulators: 128 each) are never all live (measured: the unfused order spilled
// ~150 VGPRs per phase transition into scratch, 3 GB of HBM traffic per launch).  acc[i] = (x[i] * WITH_X + bias) * scale.
template <bool WITH_X>
__device__ __forceinline__ void prepare(const f32x16 (&x)[8], float s, lds_cfloat* pb, int bias, float scale,
                                        f16x8 (&xh)[16], f16x8 (&xl)[16], f32x16 (&acc)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) split8(x[i], 8 * sp, s, xh[2 * i + sp], xl[2 * i + sp]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = param4(pb, bias, 32 * i, g);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][4 * g + e] = ((WITH_X ? x[i][4 * g + e] : 0.f) + b[e]) * scale;
        }
    }
}

// ---- x = LN(x + W x + b) ---------------------------------------------------------------------------------------
__device__ __forceinline__ void phase_attn_ln(Ring& ring, const Phase& P, f32x16 (&x)[8], lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[16], xl[16];
    f32x16 acc[8];
    prepare<true>(x, s, pb, P.b1, s * P.sw1, xh, xl, acc);
    gemm256(ring, xh, xl, acc);
    const float un = inv / P.sw1;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[i][r] = acc[i][r] * un;
    layer_norm(x, pb, P.gamma, P.beta, P.ln_eps);
}

// ---- hidden tile -> planes: H = relu(acc1) * c (c = hidden scale / (2^sw1 * row scale)), clamped below the fp16 maximum
__device__ __forceinline__ void hidden_planes(const f32x16& a1, float c, f16x8 (&hh)[2], f16x8 (&hl)[2]) {
    if (DBG & 8) {
        asm volatile("" : "+v"(hh[0]), "+v"(hh[1]), "+v"(hl[0]), "+v"(hl[1]) : "v"(a1));
        return;
    }
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = fminf(fmaxf(a1[r], 0.f) * c, 60000.f);
    split8(t, 0, 1.0f, hh[0], hl[0]);
    split8(t, 8, 1.0f, hh[1], hl[1]);
}
__device__ __forceinline__ void init_tile(f32x16& a, lds_cfloat* pb, int bias, int f0, float scale) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = param4(pb, bias, f0, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * g + e] = b[e] * scale;
    }
}

// One FFN step: stage 1 of hidden tile t (acc1 += W_1[tile t] . x, 16 k-steps) interleaved with stage 2 of hidden tile
// t - 1 (acc2[i] += W_2[tile i][k-steps 2(t-1), 2(t-1)+1] . H(t-1)).  Stream order per micro-step u = 0..15:
// S1: {A1_h(t, u), A1_l(t, u)}   S2: {A2_h(i = u & 7, ks = 2 (t-1) + (u >> 3)), A2_l(..)}
template <bool S1, bool S2>
__device__ __forceinline__ void ffn_step(Ring& ring, const f16x8 (&xh)[16], const f16x8 (&xl)[16], f32x16& acc1,
                                         f32x16 (&acc2)[8], const f16x8 (&hh)[2], const f16x8 (&hl)[2]) {
    constexpr int N = (S1 ? 2 : 0) + (S2 ? 2 : 0);
    f16x8 cur[N], nxt[N];
    ring.read<N>(cur);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u < 15) ring.read<N>(nxt);
        constexpr int o2 = S1 ? 2 : 0;
        if constexpr (S1 && S2) {
            acc1 = mfma(cur[0], xl[u], acc1);
            acc2[u & 7] = mfma(cur[o2], hl[u >> 3], acc2[u & 7]);
            acc1 = mfma(cur[1], xh[u], acc1);
            acc2[u & 7] = mfma(cur[o2 + 1], hh[u >> 3], acc2[u & 7]);
            acc1 = mfma(cur[0], xh[u], acc1);
            acc2[u & 7] = mfma(cur[o2], hh[u >> 3], acc2[u & 7]);
        } else if constexpr (S1) {
            mac3(acc1, cur[0], cur[1], xh[u], xl[u]);
        } else {
            mac3(acc2[u & 7], cur[0], cur[1], hh[u >> 3], hl[u >> 3]);
        }
        if (u < 15) {
#pragma unroll
            for (int v = 0; v < N; ++v) cur[v] = nxt[v];
        }
    }
}

// ---- x = LN(x + W_2 relu(W_1 x + b_1) + b_2) -----------------------------------------------------------------------
__device__ __forceinline__ void phase_ffn_ln(Ring& ring, const Phase& P, f32x16 (&x)[8], lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[16], xl[16];
    const float sh = hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));   // max|x| < 2^13 / s
    f32x16 acc2[8];
    prepare<true>(x, s, pb, P.b2, P.sw2 * sh, xh, xl, acc2);  // planes; residual + b_2 in stage 2's scaled domain
    const float b1s = s * P.sw1;                              // stage 1 accumulates (W_1 2^sw1)(x s)
    const float c1 = sh * inv / P.sw1;                        // acc1 -> H sh
    f32x16 acc1;
    f16x8 hh[2], hl[2];
    init_tile(acc1, pb, P.b1, 0, b1s);
    ffn_step<true, false>(ring, xh, xl, acc1, acc2, hh, hl);
    for (int t = 1; t < P.n_steps; ++t) {
        hidden_planes(acc1, c1, hh, hl);
        init_tile(acc1, pb, P.b1, 32 * t, b1s);
        ffn_step<true, true>(ring, xh, xl, acc1, acc2, hh, hl);
    }
    hidden_planes(acc1, c1, hh, hl);
    ffn_step<false, true>(ring, xh, xl, acc1, acc2, hh, hl);
    const float un = 1.0f / (P.sw2 * sh);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[i][r] = acc2[i][r] * un;
    layer_norm(x, pb, P.gamma, P.beta, P.ln_eps);
}

// ---- row I/O in accumulator layout: lane (q, h) moves the 16-byte groups [32 i + 8 g + 4 h, +4) of row q -------------
__device__ __forceinline__ void load_rows(f32x16 (&x)[8], const float* row_ptr, int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row_ptr + 32 * i + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[i][4 * g + e] = v[e];
        }
}
__device__ __forceinline__ void add_rows(f32x16 (&x)[8], const float* row_ptr, int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row_ptr + 32 * i + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[i][4 * g + e] += v[e];
        }
}
__device__ __forceinline__ void store_rows(const f32x16 (&x)[8], float* row_ptr, int h) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(row_ptr + 32 * i + 8 * g + 4 * h) =
                f32x4{x[i][4 * g], x[i][4 * g + 1], x[i][4 * g + 2], x[i][4 * g + 3]};
}

// ---- xl = x0 * (W xl + b) + xl   (FeatureInteractionLayer :199-202); x0 is re-read from `x0_row` (scratch, L2) -------
__device__ __forceinline__ void phase_cross(Ring& ring, const Phase& P, f32x16 (&xl_)[8], const float* x0_row, int h,
                                            lds_cfloat* pb) {
    float s, inv;
    row_scale(xl_, s, inv);
    f16x8 xh[16], xl[16];
    f32x16 acc[8];
    prepare<false>(xl_, s, pb, P.b1, s * P.sw1, xh, xl, acc);
    gemm256(ring, xh, xl, acc);
    const float un = inv / P.sw1;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x0_row + 32 * i + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) xl_[i][4 * g + e] = v0[e] * (acc[i][4 * g + e] * un) + xl_[i][4 * g + e];
        }
}

// ---- heads: per task  logit = w3 . relu(W_2 relu(W_1 x + b_1) + b_2) + b3   (transformer_ranker.py:277-305, :375-378)
// stream per task: for t in hidden tiles: stage 1 {A1_h(t, u), A1_l(t, u)} u = 0..15 (32 sets), then stage 2
// {A2_h(i, s), A2_l(i, s)} for s = 0, 1, i = 0, 1 (8 sets): 40 sets per hidden tile, 8 tiles = 320 sets = 20 chunks.
__device__ __forceinline__ void phase_heads(Ring& ring, const Program& G, const Phase& P, const f32x16 (&x)[8], float* out,
                                            long long ld_out, long long row, bool row_ok, int h, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[16], xl[16];
    split_rows(x, s, xh, xl);
    const float sh = hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));
    const float b1s = s * P.sw1, c1 = sh * inv / P.sw1, un2 = 1.0f / (P.sw2 * sh);
    for (int task = 0; task < P.n_tasks; ++task) {
        f32x16 acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) init_tile(acc2[i], pb, G.hb2[task], 32 * i, P.sw2 * sh);
        for (int t = 0; t < P.n_steps; ++t) {
            f32x16 acc1;
            init_tile(acc1, pb, P.b1 + task * P.n_steps * 32, 32 * t, b1s);
            f16x8 cur[2], nxt[2];
            ring.read<2>(cur);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (u < 15) ring.read<2>(nxt);
                mac3(acc1, cur[0], cur[1], xh[u], xl[u]);
                if (u < 15) { cur[0] = nxt[0]; cur[1] = nxt[1]; }
            }
            f16x8 hh[2], hl[2];
            hidden_planes(acc1, c1, hh, hl);
            f16x8 a2[4];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                ring.read<4>(a2);
                mac3(acc2[0], a2[0], a2[1], hh[sp], hl[sp]);
                mac3(acc2[1], a2[2], a2[3], hh[sp], hl[sp]);
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = param4(pb, G.hw3[task], 32 * i, g);
#pragma unroll
                for (int e = 0; e < 4; ++e) dot += fmaxf(acc2[i][4 * g + e] * un2, 0.f) * w[e];
            }
        dot += __shfl_xor(dot, 32, 64);
        if (h == 0 && row_ok) out[(long long)task * ld_out + row] = dot + pb[G.hb3[task] - 4 * h];   // b_3 (h == 0 here)
    }
}

// ---- the kernel: input rows (dense X or cached-projection gather) -> all phases of the chain -> logits ---------------
__global__ __launch_bounds__(256, 1) void ranker_x3_kernel(Program G, Input in, long long rows, float* scratch,
                                                           float* x_out, long long ld_xout, float* logits,
                                                           long long ld_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, q = lane & 31;
    const long long row = (long long)blockIdx.x * ROWS_PER_WG + wave * ROWS_PER_WAVE + q;
    const bool row_ok = row < rows;
    const long long rowc = row_ok ? row : rows - 1;                 // clamped: branch-free loads, stores are guarded

    // parameter blob -> LDS behind the ring (plain LDS-DMA, 4 KB per pass of the workgroup), visible after the ring's first barrier
    lds_byte* pbase = (lds_byte*)smem + RING_BYTES;
    for (int o = 0; o < G.n_params * 4; o += 4096)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(
                                             reinterpret_cast<const unsigned char*>(G.params) + o + tid * 16),
                                         (__attribute__((address_space(3))) void*)(pbase + o + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_cfloat* pb = reinterpret_cast<lds_cfloat*>(pbase) + 4 * h;   // lane half h reads features + 4 h

    Ring ring;
    ring.start(G.stream, G.total_chunks, (lds_byte*)smem, wave, lane);

    f32x16 x[8];
    if (in.X != nullptr) {
        load_rows(x, in.X + rowc * in.ldx, h);
    } else {
        const long long gr = in.row_base + rowc;
        long long a = in.rowmap ? in.rowmap[gr] : gr;
        a = a < 0 ? 0 : (a >= in.n_cache ? in.n_cache - 1 : a);     // clamped like the gather loader (reported separately)
        load_rows(x, in.cache + a * in.ldc, h);
        add_rows(x, in.U + (gr / in.rowdiv) * 256, h);              // cache row + the user's half (same order as proj_gather)
    }
    float* x0_row = scratch + row * 256;                            // this lane's own row (scratch is padded to whole workgroups)
    bool x0_saved = false;
    for (int p = 0; p < G.n_phases; ++p) {
        const Phase& P = G.ph[p];
        const int type = __builtin_amdgcn_readfirstlane(P.type);
        if (type == PH_ATTN_LN) {
            phase_attn_ln(ring, P, x, pb);
        } else if (type == PH_FFN_LN) {
            phase_ffn_ln(ring, P, x, pb);
        } else if (type == PH_CROSS) {
            if (!x0_saved) {                                        // x0 = the encoder output, kept for all cross layers
                store_rows(x, x0_row, h);
                x0_saved = true;
            }
            phase_cross(ring, P, x, x0_row, h, pb);
        } else {
            phase_heads(ring, G, P, x, logits, ld_logits, row, row_ok, h, pb);
        }
    }
    if (x_out != nullptr && row_ok) store_rows(x, x_out + row * ld_xout, h);
    ring.drain();                                                   // no LDS-DMA may land after the workgroup's LDS is released
}

}  // namespace x3
}  // namespace amdrec
