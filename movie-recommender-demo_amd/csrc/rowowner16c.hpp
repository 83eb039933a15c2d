// Row-owner engine, COLUMN-SPLIT variant ("x3c") for single requests and other small passes (<= 4096 rows).
//
// WHY.  In the 16-row kernels (rowowner16.hpp) a wave carries its 16 rows through the whole chain alone: 212 MFLOP of fp16
// products per wave = 87 us of one SIMD's matrix pipe at best, 200 us in practice (LDS fragment reads, waits) - and that is
// the latency of ONE request's 500 candidates whatever the chip's other 990 SIMDs do (profiles/r03_trace_b1.log: 200 of the
// 417 us).  Here a WORKGROUP of four waves owns 16 rows and the waves split every GEMM's OUTPUT features: wave w computes
// tiles 4 w .. 4 w + 3 of a 256-wide layer, one hidden tile pair of four in the FFN / heads.  A wave's MFMA work falls to a
// quarter (22 us), the 500 rows spread over 32 CUs instead of 8, and the bound becomes the rate at which ONE CU can pull the
// 8.8 MB weight stream through LDS-DMA: 108 GB/s = 82 us (tools/dma_rate_probe.hip, profiles/r03_dma_rate_probe.log).
//
// SAME ARITHMETIC, BIT FOR BIT.  Every output element sees the MFMAs of the 16-row kernel in the same order (k-steps
// ascending, the three plane products in group6's order), the row scale, the plane split, LayerNorm, the hidden-tile
// conversion and the heads' final dot are the 16-row kernel's own functions on the same values; only WHO computes an
// element changes.  tests/test_x3_gpu.py compares the two kernels with torch.equal.
//
// HOW.  lane (q = l & 15, g = l >> 4) as in rowowner16.hpp; all four waves hold the same 16 rows.
//  * a GEMM's accumulators are split by output tile; afterwards each wave writes its four accumulator tiles to a 16 KB LDS
//    exchange area, one barrier, every wave reads all sixteen and runs the row-wise epilogue (LayerNorm / cross / row
//    scale / plane split) REPLICATED: ~600 vector instructions per phase and wave, no cross-wave reductions, and every
//    wave has the full row as the next GEMM's B operand.
//  * FFN / heads: hidden tiles in SUPER-STEPS of four steps (one per wave, K = 256 each); a wave converts its hidden tile
//    to fp16 planes and publishes them in LDS (double-buffered, 2 KB per step); stage 2 of super-step T - 1 (its four
//    k-steps, ascending, every wave on its own output tiles) interleaves with stage 1 of super-step T as in the 16-row
//    kernel.  The heads' 64 outputs per task are two tile pairs: waves 0 and 1 (waves 2, 3 multiply zero fragments).
//  * WEIGHT STREAM (amdrec/weights.py x3c_stream_*): the same fragment sets, ordered so that EVERY 16 KB chunk holds one
//    group (4 fragment sets = 6 MFMAs) per wave, group w for wave w.  Ring of 4 chunks: at the barrier that ends chunk
//    c every wave has chunk c + 1's group in registers already (read during chunk c's MFMAs), chunk c + 2 is certified
//    (counted vmcnt + barrier), c + 3, c + 4 are in flight and the new DMA (c + 5) refills chunk c + 1's slot: 48 KB in
//    flight, which gives the full DMA rate (see the probe).  With ONE wave per SIMD every instruction of the chunk loop is
//    exposed (a wave issues at most one instruction per ~4 cycles: the first version's ~45 instructions per chunk - ring
//    arithmetic, register moves - ran at 375 cycles per chunk against the stream's 315), so the ring position is a
//    compile-time constant everywhere: every phase is a multiple of 4 chunks long, the chunk loops are unrolled, a
//    chunk's DMA / LDS addresses are immediates and the stream offset one scalar add (reads past the stream's end are
//    dropped by the buffer descriptor's bounds check, no clamp).
//    LDS: ring 64 KB + parameters 44 KB + exchange 16 KB + hidden planes 16 KB = 140 KB.
#pragma once
#include "rowowner16.hpp"

namespace amdrec {
namespace x3c {

using x3::CHUNK_BYTES;
using x3::f16x8;
using x3::FRAG_BYTES;
using x3::Input;
using x3::lds_byte;
using x3::lds_cfloat;
using x3::PARAM_FLOATS;
using x3::Phase;
using x3::Program;
using x3b::add_rows;
using x3b::hidden_planes;
using x3b::init_pair;
using x3b::load_rows;
using x3b::param4;
using x3b::reduce_sum4;
using x3b::row_scale;
using x3b::split8;
using x3b::store_rows;

// Elimination / stamp switches for tools/x3c_time.py (0 in the product): 1 = no weight DMA and no wait for it, 2 = no MFMAs,
// 4 = no fragment reads, 16 = cycle stamps (whole kernel, chunk-closing waits + barriers) into the logits buffer's tail.
#ifndef AMDREC_X3C_DBG
#define AMDREC_X3C_DBG 0
#endif
constexpr int CDBG = AMDREC_X3C_DBG;
constexpr int WAVES = 4, ROWS_PER_WG = 16;
constexpr int NBUF = 4;                       // ring chunks (a power of two; every phase is a multiple of 4 chunks long)
constexpr int AHEAD = NBUF - 1;               // chunks in flight beyond the certified one (see Ring::advance)
constexpr int PIECES = 4;                     // 1 KB DMA instructions per wave and chunk
constexpr int RING_BYTES = NBUF * CHUNK_BYTES;
constexpr int EX_BYTES = 16 * 64 * 16;        // accumulator exchange: 16 tiles x 64 lanes x f32x4
constexpr int HID_BYTES = 2 * 4 * 2 * FRAG_BYTES;   // two super-steps x four hidden steps x {h, l} planes
constexpr int LDS_BYTES = RING_BYTES + PARAM_FLOATS * 4 + EX_BYTES + HID_BYTES;
constexpr long long MAX_ROWS = 256ll * ROWS_PER_WG;  // one workgroup per CU

struct Ring {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff, so;            // lane * 16; stream offset of this wave's pieces of the next chunk to request
    lds_byte* lds_dma;            // ring + wave * 4 KB (wave-uniform)
    const lds_byte* lds_rd;       // ring + wave * 4 KB + lane * 16: this wave's group of a chunk
    unsigned long long t_wait, t_bar;   // CDBG & 16

    // request piece U (1 KB) of the next chunk of the stream into ring slot `slot` (a constant after unrolling); the chunk
    // counter moves with the last piece.  Past the stream's end the descriptor's bounds check drops the loads.
    template <int U>
    __device__ __forceinline__ void issue_piece(int slot) {
        if (CDBG & 1) return;
        lds_byte* dst = lds_dma + slot * CHUNK_BYTES;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, so, U * FRAG_BYTES, 0);
        if (U == PIECES - 1) so += CHUNK_BYTES;
    }
    __device__ __forceinline__ void issue(int slot) {
        issue_piece<0>(slot); issue_piece<1>(slot); issue_piece<2>(slot); issue_piece<3>(slot);
    }
    __device__ __forceinline__ void read_group(int slot, f16x8 (&f)[4]) const {
        const lds_byte* a = lds_rd + slot * CHUNK_BYTES;
        if (CDBG & 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(f[u]));
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            f[u] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(a + u * FRAG_BYTES);
    }
    // chunks 0 .. NBUF requested, 0 and 1 landed, chunk 0's group in `cur`
    __device__ __forceinline__ void start(const unsigned char* stream, int total_chunks, lds_byte* lds, int wave, int lane,
                                          f16x8 (&cur)[4]) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(stream), 0, (uint32_t)total_chunks * CHUNK_BYTES,
                                                 0x00020000);
        voff = (uint32_t)lane * 16;
        so = (uint32_t)wave * PIECES * FRAG_BYTES;
        lds_dma = lds + wave * PIECES * FRAG_BYTES;
        lds_rd = lds + wave * 4 * FRAG_BYTES + lane * 16;
        t_wait = t_bar = 0;
#pragma unroll
        for (int c = 0; c < NBUF; ++c) issue(c);
        if (!(CDBG & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (NBUF - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        read_group(0, cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // chunk 0's slot is free: every wave holds its group
    }
    // the NEXT chunk's group (certified by the barrier that opened this chunk); pos = this chunk's position mod NBUF
    __device__ __forceinline__ void prefetch(int pos, f16x8 (&nxt)[4]) const { read_group((pos + 1) & (NBUF - 1), nxt); }
    // End of chunk c: this wave's pieces of chunk c + 2 have landed and its LDS accesses (the prefetch of chunk c + 1's group,
    // exchange / hidden-plane traffic) have completed; barrier.  Chunk c + 1's slot - every wave holds its group in
    // registers now - is free: chunk c + 1's body refills it with chunk c + 1 + NBUF (between its MFMAs, see chunk()).
    __device__ __forceinline__ void advance() {
        if (CDBG & 16) {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(PIECES * (AHEAD - 1)) : "memory");
            const unsigned long long b = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            t_wait += b - a;
            t_bar += __builtin_amdgcn_s_memtime() - b;
        } else {
            if (CDBG & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(PIECES * (AHEAD - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// One chunk of a GEMM: this wave's group against one B k-step.  pos = the chunk's position mod NBUF, a constant at every
// call once the loops are unrolled (each phase / super-step starts at a multiple of NBUF), and moves on.
// The four DMA instructions that refill this chunk's own slot (its group is in `cur`) go out BETWEEN the MFMAs: with one
// wave per SIMD a DMA instruction that waits for the memory pipeline blocks the wave's issue, and the stream runs at
// the pipeline's limit - issued in a block after the barrier they added their whole wait to the MFMAs' time (118 us
// per launch, 71 without DMA, 114 without MFMAs); two MFMAs ahead of each keep the matrix pipe busy under the wait.
__device__ __forceinline__ void chunk(Ring& ring, f16x8 (&cur)[4], int& pos, const f16x8& bh, const f16x8& bl, f32x4& c0, f32x4& c1) {
    f16x8 nxt[4];
    ring.prefetch(pos, nxt);
    __builtin_amdgcn_sched_barrier(0);        // the reads stay AHEAD of the MFMAs (the scheduler sinks them to their first use)
    if (CDBG & 2) {
        asm volatile("" : "+v"(c0), "+v"(c1) : "v"(cur[0]), "v"(cur[1]), "v"(cur[2]), "v"(cur[3]), "v"(bh), "v"(bl));
        ring.issue(pos);
    } else {                                  // group6's products in group6's order
        c0 = x3b::mfma(cur[0], bl, c0);
        c1 = x3b::mfma(cur[2], bl, c1);
        __builtin_amdgcn_sched_barrier(0);
        ring.template issue_piece<0>(pos);
        __builtin_amdgcn_sched_barrier(0);
        c0 = x3b::mfma(cur[1], bh, c0);
        c1 = x3b::mfma(cur[3], bh, c1);
        __builtin_amdgcn_sched_barrier(0);
        ring.template issue_piece<1>(pos);
        __builtin_amdgcn_sched_barrier(0);
        c0 = x3b::mfma(cur[0], bh, c0);
        c1 = x3b::mfma(cur[2], bh, c1);
        __builtin_amdgcn_sched_barrier(0);
        ring.template issue_piece<2>(pos);
        ring.template issue_piece<3>(pos);
    }
    __builtin_amdgcn_sched_barrier(0);        // ... and everything in front of the chunk's closing wait (MFMAs would sink past it)
    ring.advance();
    pos = (pos + 1) & (NBUF - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
}

// tiles 4 w .. 4 w + 3 of a full row state.  w is wave-uniform; written as selects on values: an if-chain over x[i], x[4 + i],
// ... is folded by the compiler into the indexed load x[4 w + i], which puts the whole row state into scratch memory
__device__ __forceinline__ void own4(const f32x4 (&x)[16], int w, f32x4 (&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 a = x[i], b = x[4 + i], c = x[8 + i], d = x[12 + i];
        o[i] = w == 0 ? a : (w == 1 ? b : (w == 2 ? c : d));
    }
}

struct Exchange {
    lds_byte* ex;                 // exchange area + lane * 16
    lds_byte* hid;                // hidden planes + lane * 16
    __device__ __forceinline__ void put(int tile, const f32x4& v) const {
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(ex + tile * 1024) = v;
    }
    __device__ __forceinline__ f32x4 get(int tile) const {
        return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(ex + tile * 1024);
    }
    // publish this wave's four tiles, collect all sixteen (the area's previous readers are at least one chunk barrier behind)
    __device__ __forceinline__ void all_gather(const f32x4 (&own)[4], int w, f32x4 (&all)[16]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) put(4 * w + i, own[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < 16; ++t) all[t] = get(t);
    }
    // this wave's LDS writes are complete and every wave has passed: what was put before is readable
    __device__ __forceinline__ void publish() const {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __device__ __forceinline__ void put_hidden(int buf, int step, const f16x8& hh, const f16x8& hl) const {
        lds_byte* a = hid + ((buf * 4 + step) * 2) * FRAG_BYTES;
        *reinterpret_cast<__attribute__((address_space(3))) f16x8*>(a) = hh;
        *reinterpret_cast<__attribute__((address_space(3))) f16x8*>(a + FRAG_BYTES) = hl;
    }
    __device__ __forceinline__ void get_hidden(int buf, int step, f16x8& hh, f16x8& hl) const {
        const lds_byte* a = hid + ((buf * 4 + step) * 2) * FRAG_BYTES;
        hh = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(a);
        hl = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(a + FRAG_BYTES);
    }
};

// planes of the full row; this wave's accumulators start at (own tiles + bias) * scale (WITH_X) or bias * scale
template <bool WITH_X>
__device__ __forceinline__ void prepare(const f32x4 (&x)[16], const f32x4 (&xo)[4], float s, lds_cfloat* pb, int bias, float scale,
                                        int w, f16x8 (&xh)[8], f16x8 (&xl)[8], f32x4 (&acc)[4]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) split8(x[2 * ks], x[2 * ks + 1], s, xh[ks], xl[ks]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 b = param4(pb, bias, 4 * w + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] = ((WITH_X ? xo[i][r] : 0.f) + b[r]) * scale;
    }
}

// x3b::layer_norm_scaled on the gathered accumulators, plus this wave's own tiles of the result from its own accumulators
// (the same expressions on the same values: xo[i] == y[4 w + i] bit for bit, without 192 selects to pick them out of y)
__device__ __forceinline__ void layer_norm_own(const f32x4 (&all)[16], const f32x4 (&acc)[4], float un, f32x4 (&y)[16],
                                               f32x4 (&xo)[4], int w, lds_cfloat* pb, int gamma, int beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += all[t][r];
    const float mean = reduce_sum4(s) * un * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = __builtin_fmaf(all[t][r], un, -mean);
            y[t][r] = d;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(reduce_sum4(q) * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 ga = param4(pb, gamma, t), be = param4(pb, beta, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[t][r] = y[t][r] * rstd * ga[r] + be[r];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 ga = param4(pb, gamma, 4 * w + i), be = param4(pb, beta, 4 * w + i);
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[i][r] = __builtin_fmaf(acc[i][r], un, -mean) * rstd * ga[r] + be[r];
    }
}

// this wave's quarter of a 256 x 256 GEMM: 16 chunks = 8 k-steps x 2 tile pairs
__device__ __forceinline__ void gemm256(Ring& ring, f16x8 (&cur)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8], f32x4 (&acc)[4]) {
    int pos = 0;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        chunk(ring, cur, pos, xh[ks], xl[ks], acc[0], acc[1]);
        chunk(ring, cur, pos, xh[ks], xl[ks], acc[2], acc[3]);
    }
}

__device__ __forceinline__ void phase_attn_ln(Ring& ring, f16x8 (&cur)[4], const Exchange& E, const Phase& P, f32x4 (&x)[16],
                                              f32x4 (&xo)[4], int w, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
    f32x4 acc[4];
    prepare<true>(x, xo, s, pb, P.b1, s * P.sw1, w, xh, xl, acc);
    gemm256(ring, cur, xh, xl, acc);
    f32x4 all[16];
    E.all_gather(acc, w, all);
    layer_norm_own(all, acc, inv / P.sw1, x, xo, w, pb, P.gamma, P.beta, P.ln_eps);
}

__device__ __forceinline__ void phase_ffn_ln(Ring& ring, f16x8 (&cur)[4], const Exchange& E, const Phase& P, f32x4 (&x)[16],
                                             f32x4 (&xo)[4], int w, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
    const float sh = x3::hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));
    f32x4 acc2[4];
    prepare<true>(x, xo, s, pb, P.b2, P.sw2 * sh, w, xh, xl, acc2);
    const float b1s = s * P.sw1, c1 = sh * inv / P.sw1, lim1 = 60000.f / c1;
    const int S = P.n_steps >> 2;                                // super-steps of four hidden steps (x3_build: n_steps % 4 == 0)
    f32x4 a10, a11;
    f16x8 hh, hl, ph, pl;
    // super-step 0: stage 1 only (8 chunks)
    init_pair(a10, a11, pb, P.b1, 2 * w, b1s);
    {
        int pos = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) chunk(ring, cur, pos, xh[i], xl[i], a10, a11);
    }
    hidden_planes(a10, a11, c1, lim1, ph, pl);
    E.put_hidden(0, w, ph, pl);
    // super-steps 1 .. S - 1: stage 1 of T interleaved with stage 2 of T - 1 (16 chunks)
    for (int T = 1; T < S; ++T) {
        init_pair(a10, a11, pb, P.b1, 2 * (4 * T + w), b1s);
        int pos = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // the planes of stage 2's k-step are requested ahead of the stage-1 chunk whose closing wait covers them; the
            // FIRST ones behind it: that chunk's barrier separates them from the other waves' put_hidden
            if ((i & 1) == 0 && i > 0) E.get_hidden((T - 1) & 1, i >> 1, hh, hl);
            chunk(ring, cur, pos, xh[i], xl[i], a10, a11);                                    // stage 1, k-step i
            if (i == 0) E.get_hidden((T - 1) & 1, 0, hh, hl);
            chunk(ring, cur, pos, hh, hl, acc2[2 * (i & 1)], acc2[2 * (i & 1) + 1]);          // stage 2 of super-step T - 1
        }
        hidden_planes(a10, a11, c1, lim1, ph, pl);
        E.put_hidden(T & 1, w, ph, pl);
    }
    // stage 2 of the last super-step (8 chunks); no stage-1 chunk (= barrier) ahead of the first planes' read
    E.publish();
    {
        int pos = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i & 1) == 0) E.get_hidden((S - 1) & 1, i >> 1, hh, hl);
            chunk(ring, cur, pos, hh, hl, acc2[2 * (i & 1)], acc2[2 * (i & 1) + 1]);
        }
    }
    f32x4 all[16];
    E.all_gather(acc2, w, all);
    layer_norm_own(all, acc2, 1.0f / (P.sw2 * sh), x, xo, w, pb, P.gamma, P.beta, P.ln_eps);
}

// xl <- x0 * (xl W + b) + xl; x0o / xo = this wave's tiles of the trunk's output / of xl
__device__ __forceinline__ void phase_cross(Ring& ring, f16x8 (&cur)[4], const Exchange& E, const Phase& P, f32x4 (&x)[16],
                                            f32x4 (&xo)[4], const f32x4 (&x0o)[4], int w, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
    f32x4 acc[4];
    prepare<false>(x, xo, s, pb, P.b1, s * P.sw1, w, xh, xl, acc);
    gemm256(ring, cur, xh, xl, acc);
    const float un = inv / P.sw1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) xo[i][r] = x0o[i][r] * (acc[i][r] * un) + xo[i][r];
    E.all_gather(xo, w, x);
}

__device__ __forceinline__ void phase_heads(Ring& ring, f16x8 (&cur)[4], const Exchange& E, const Program& G, const Phase& P,
                                            const f32x4 (&x)[16], float* out, long long ld_out, long long row, bool store,
                                            int w, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) split8(x[2 * ks], x[2 * ks + 1], s, xh[ks], xl[ks]);
    const float sh = x3::hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));
    const float b1s = s * P.sw1, c1 = sh * inv / P.sw1, un2 = 1.0f / (P.sw2 * sh), lim1 = 60000.f / c1;
    const int T = P.n_steps, S = (P.n_tasks * T) >> 2;          // hidden steps per task / super-steps in all (T % 4 == 0)
    const int pr = w & 1;                                        // this wave's pair of the task's 64 outputs (waves 2, 3: zero fragments)
    f32x4 a2[2], a10, a11;
    f16x8 hh0, hl0, hh1, hl1;
    int task = 0, done = 0;                                      // hidden steps of `task` whose stage 2 has run
    init_pair(a2[0], a2[1], pb, G.hb2[0], 2 * pr, P.sw2 * sh);
    // a task's 64 outputs are complete once the stage 2 of all its hidden steps has run: waves 0, 1 publish them
    auto stage2_done = [&]() {
        done += 4;
        if (done != T) return;
        if (w < 2) { E.put(2 * w, a2[0]); E.put(2 * w + 1, a2[1]); }
        E.publish();
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 o = E.get(t);
            const f32x4 wv = param4(pb, G.hw3[task], t);
#pragma unroll
            for (int r = 0; r < 4; ++r) dot += fmaxf(o[r] * un2, 0.f) * wv[r];
        }
        dot = reduce_sum4(dot);
        if (store) out[(long long)task * ld_out + row] = dot + pb[G.hb3[task]];               // store: wave 0, g == 0, row valid
        done = 0;
        if (++task < P.n_tasks) init_pair(a2[0], a2[1], pb, G.hb2[task], 2 * pr, P.sw2 * sh);
    };
    f16x8 ph, pl;
    // super-step 0: stage 1 only (8 chunks)
    init_pair(a10, a11, pb, P.b1, 2 * w, b1s);                                                // stacked b_1
    {
        int pos = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) chunk(ring, cur, pos, xh[i], xl[i], a10, a11);
    }
    hidden_planes(a10, a11, c1, lim1, ph, pl);
    E.put_hidden(0, w, ph, pl);
    // super-steps 1 .. S - 1: 8 stage-1 chunks with the 4 stage-2 chunks of super-step ss - 1 behind chunks 3 and 7
    for (int ss = 1; ss < S; ++ss) {
        init_pair(a10, a11, pb, P.b1, 2 * (4 * ss + w), b1s);
        int pos = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool s2 = (i & 3) == 3;                        // two stage-2 chunks (k-steps 2 (i >> 2), + 1) behind this one
            if (s2) {                                            // (the stage-1 chunks 0 .. 2 are the barrier behind put_hidden)
                E.get_hidden((ss - 1) & 1, 2 * (i >> 2), hh0, hl0);
                E.get_hidden((ss - 1) & 1, 2 * (i >> 2) + 1, hh1, hl1);
            }
            chunk(ring, cur, pos, xh[i], xl[i], a10, a11);
            if (s2) {
                chunk(ring, cur, pos, hh0, hl0, a2[0], a2[1]);
                chunk(ring, cur, pos, hh1, hl1, a2[0], a2[1]);
            }
        }
        hidden_planes(a10, a11, c1, lim1, ph, pl);
        E.put_hidden(ss & 1, w, ph, pl);
        stage2_done();
    }
    // stage 2 of the last super-step (4 chunks)
    E.publish();
    {
        int pos = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            E.get_hidden((S - 1) & 1, 2 * h, hh0, hl0);
            E.get_hidden((S - 1) & 1, 2 * h + 1, hh1, hl1);
            chunk(ring, cur, pos, hh0, hl0, a2[0], a2[1]);
            chunk(ring, cur, pos, hh1, hl1, a2[0], a2[1]);
        }
    }
    stage2_done();
}

// PREFETCH workgroups.  The kernel's bound is the latency-limited rate of ONE workgroup's 48 KB of DMA in flight, so it matters
// where the stream comes from: 118 us per launch when the 8.8 MB sit in the memory-side cache (back-to-back launches),
// 165 us in the serving pipeline, where the search's 1 GB corpus scan has just pushed them out to HBM
// (profiles/r03_trace_b1_colsplit.log).  The pass occupies a few dozen of the 256 CUs, so the launch carries PREFETCH_WGS
// extra workgroups that do nothing but read one slice of the stream each (9 x 16 B per thread, all in flight at once) and
// exit: the stream is on its way into the cache hierarchy within the first microseconds of the kernel.
constexpr int PREFETCH_WGS = 64;

__global__ __launch_bounds__(64 * WAVES) void ranker_x3c_kernel(Program G, Input in, long long rows, int n_row_wgs, float* x_out,
                                                                 long long ld_xout, float* logits, long long ld_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_pre = (int)gridDim.x - n_row_wgs;                 // the FIRST workgroups of the grid: dispatched first even
    if ((int)blockIdx.x < n_pre) {                                // when the row workgroups fill every CU (one per CU: LDS)
        const int j = (int)blockIdx.x;
        const long long n16 = (long long)G.total_chunks * (CHUNK_BYTES / 16);          // 16-byte words of the stream
        const long long per = (n16 + n_pre - 1) / n_pre;
        const f32x4* src = reinterpret_cast<const f32x4*>(G.stream);
        float sink = 0.f;
        for (long long i = j * per + threadIdx.x; i < (j + 1) * per && i < n16; i += 64 * WAVES) sink += src[i][0];
        asm volatile("" ::"v"(sink));
        return;
    }
    const int wg = (int)blockIdx.x - n_pre;                       // row workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, q = lane & 15;

    lds_byte* pbase = (lds_byte*)smem + RING_BYTES;
    for (int o = 0; o + wave * 1024 < G.n_params * 4; o += WAVES * 1024)  // WAVES x 1 KB per pass; n_params % 1024 == 0
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(
                                             reinterpret_cast<const unsigned char*>(G.params) + o + tid * 16),
                                         (__attribute__((address_space(3))) void*)(pbase + o + wave * 1024), 16, 0, 0);
    lds_cfloat* pb = reinterpret_cast<lds_cfloat*>(pbase) + 4 * g;
    Exchange E;
    E.ex = pbase + PARAM_FLOATS * 4 + lane * 16;
    E.hid = pbase + PARAM_FLOATS * 4 + EX_BYTES + lane * 16;

    const long long row = (long long)wg * ROWS_PER_WG + q;
    const bool row_ok = row < rows;
    const long long rowc = row_ok ? row : rows - 1;
    f32x4 x[16];                                                  // every wave loads the workgroup's 16 rows
    if (in.X != nullptr) {
        load_rows(x, in.X + rowc * in.ldx, g);
    } else {
        const long long gr = in.row_base + rowc;
        long long a = in.rowmap ? in.rowmap[gr] : gr;
        a = a < 0 ? 0 : (a >= in.n_cache ? in.n_cache - 1 : a);
        load_rows(x, in.cache + a * in.ldc, g);
        add_rows(x, in.U + (gr / in.rowdiv) * 256, g);
    }
    const unsigned long long t_begin = (CDBG & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
    Ring ring;
    f16x8 cur[4];
    ring.start(G.stream, G.total_chunks, (lds_byte*)smem, wave, lane, cur);   // its vmcnt wait covers the blob and the rows too

    f32x4 xo[4];                                                  // this wave's tiles of x, carried beside the full row
    own4(x, wave, xo);
    int p = 0;
    for (; p < G.n_phases; ++p) {
        const Phase& P = G.ph[p];
        const int type = __builtin_amdgcn_readfirstlane(P.type);
        if (type == x3::PH_ATTN_LN) phase_attn_ln(ring, cur, E, P, x, xo, wave, pb);
        else if (type == x3::PH_FFN_LN) phase_ffn_ln(ring, cur, E, P, x, xo, wave, pb);
        else break;
    }
    if (p < G.n_phases && __builtin_amdgcn_readfirstlane(G.ph[p].type) == x3::PH_CROSS) {
        f32x4 x0o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x0o[i] = xo[i];
        for (; p < G.n_phases && __builtin_amdgcn_readfirstlane(G.ph[p].type) == x3::PH_CROSS; ++p)
            phase_cross(ring, cur, E, G.ph[p], x, xo, x0o, wave, pb);
    }
    if (p < G.n_phases) phase_heads(ring, cur, E, G, G.ph[p], x, logits, ld_logits, row, row_ok && wave == 0 && g == 0, wave, pb);
    if (x_out != nullptr && row_ok && wave == 0) store_rows(x, x_out + row * ld_xout, g);
    ring.drain();
    if ((CDBG & 16) && lane == 0) {           // diagnostic build: s_memtime ticks (100 MHz) into the unused tail of the logits buffer
        float* dbg = logits + 3 * ld_logits + ((long long)wg * WAVES + wave) * 4;
        dbg[0] = (float)(__builtin_amdgcn_s_memtime() - t_begin);
        dbg[1] = (float)ring.t_wait;
        dbg[2] = (float)ring.t_bar;
        dbg[3] = 0.f;
    }
}

}  // namespace x3c
}  // namespace amdrec
