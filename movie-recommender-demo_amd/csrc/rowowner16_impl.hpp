// Body of the 16-rows-per-wave row-owner kernel (see rowowner16.hpp for the design): included once per workgroup shape
// with AMDREC_X3B_NAMESPACE / AMDREC_X3B_WAVES set - x3b: 8 waves = 128 rows per workgroup (two waves per SIMD, the
// throughput shape), x3b4: 4 waves = 64 rows (one wave per SIMD: a pass of <= 16384 rows spreads over twice the CUs and a
// wave has its SIMD to itself - 0.20 instead of 0.28 ms for one request's 500 rows, tools/x3_probe.hip).  No include guard.

// (a step without stage 2 is passed the not-yet-written hidden planes by reference and never reads them)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized-const-reference"
namespace amdrec {
namespace AMDREC_X3B_NAMESPACE {

using x3::CHUNK_BYTES;
using x3::CHUNK_FRAGS;
using x3::DBG;
using x3::f16x8;
using x3::FRAG_BYTES;
using x3::Input;
using x3::lds_byte;
using x3::lds_cfloat;
using x3::NBUF;
using x3::PARAM_FLOATS;
using x3::Phase;
using x3::Program;
using x3::RING_BYTES;
using x3::TARGET_EXP;
using x3::DEPTH;

#if !defined(AMDREC_X3B_WAVES) || !defined(AMDREC_X3B_NAMESPACE)
#error "include rowowner16.hpp, not this file"
#endif
constexpr int WAVES = AMDREC_X3B_WAVES, ROWS_PER_WAVE = 16, ROWS_PER_WG = WAVES * ROWS_PER_WAVE;
constexpr int DMA_PER_WAVE = CHUNK_FRAGS / WAVES;     // 2 fragment sets per wave and chunk

// Round-3 switches (bits of AMDREC_X3B_OPT; same-box A/Bs with tools/x3_probe.hip + tools/x3b_ab.sh, logs profiles/r03_x3b_*).
// DEFAULT 71 = 1 | 2 | 4 | 64: 3.09 -> 2.96 ms per 256 000-row launch on the same box.
//   1  hidden tile: relu + clamp as ONE v_med3_f32 in the unscaled domain, the scale folded into the plane split
//      (v_fma_mixlo/hi_f16 pairs): 3 vector instructions per element instead of 8 (bit-identical)
//   2  LayerNorm: the power-of-two unscale of the accumulators folded into the mean / deviation passes, the deviation
//      kept in place: 5 instructions per element instead of 7 (bit-identical barring fp32 denormals)
//   4  weight DMA by buffer_load ... lds with an SGPR chunk offset: no per-lane 64-bit address add, M0 written once per
//      chunk (the instruction's immediate offset advances source and LDS destination together)
//  64  the chunk's DMA pieces are issued one group AFTER the barrier (behind the first group's reads and MFMAs), so the
//      matrix pipe has work queued while both waves of a SIMD sit in the memory-instruction issue
//   (1 | 2 cut the non-MFMA vector instructions of an FFN step from 86 to 50 and move the time by 1 %; 64 alone moves
//    nothing, 4 | 64 gives 3 %, all four 4.5 %: the kernel is not vector-issue-bound.)
// Measured and NOT adopted (kept switchable where the code is small):
//  16  static priority 1 for waves 4-7: +-0
//  32  the next chunk's first fragment reads BEFORE the chunk's vmcnt wait + barrier: -2 % alone, +1.5 % on top of 64
// 128  ONE barrier per TWO chunks (needs 64): +1 % - barriers are not what the waves wait for
// 1024 (round 4) s_setprio 1 around the six MFMAs of every group, 0 after them: +0.8 % (profiles/r04_x3_setprio_ab.log)
// 512  stagger (MI355X_MICROARCH "two waves that run the SAME program with one barrier per block"): waves 4-7 take the
//      chunk's barrier in front of group 2 instead of group 0, as a second compile-time instantiation of the chain (RingT<2>):
//      +2 % SLOWER, like its run-time-branch form (removed); also removed after losing their A/Bs: fragment prefetch
//      carried across step / phase boundaries (16 more live registers: 19 -> 51 spilled, +2.5 %) and a third fragment
//      buffer for a two-group read-ahead (+2.5 %, and no faster even with DMA and barrier compiled out: the fragment
//      reads cost 0.5 ms of the launch by their LDS -> VGPR traffic, not by exposed latency); an FFN step's first group
//      requested at the end of the previous step, ahead of the hidden-tile conversion (+0.5 %: 12 more spilled registers).
#ifndef AMDREC_X3B_OPT
#define AMDREC_X3B_OPT 71
#endif
constexpr int OPT = AMDREC_X3B_OPT;
#ifndef AMDREC_X3B_DMA_G0            // OPT & 64: groups (1..3) after whose reads DMA pieces 0 and 1 are issued
#define AMDREC_X3B_DMA_G0 1
#endif
#ifndef AMDREC_X3B_DMA_G1
#define AMDREC_X3B_DMA_G1 AMDREC_X3B_DMA_G0
#endif
constexpr int DMA_G0 = AMDREC_X3B_DMA_G0, DMA_G1 = AMDREC_X3B_DMA_G1;

// LAG_: OPT & 512 (compile-time stagger): the group (0 or 2) of a chunk in front of which this wave half takes the chunk's
// barrier - waves 0-3 run RingT<0>, waves 4-7 RingT<2>, two instantiations of the whole chain (no run-time branch splits
// the unrolled GEMM blocks)
template <int LAG_>
struct RingT {
    const unsigned char* gsrc;
    lds_byte* lds_dma;
    lds_byte* lds_rd;
    int issued, total;
    int slot;                    // ring slot of the chunk being read
    __amdgpu_buffer_rsrc_t rsrc; // OPT & 4: the whole stream as a raw buffer
    uint32_t voff, soff0;        // OPT & 4: lane * 16; wave * DMA_PER_WAVE * FRAG_BYTES
    const lds_byte* cbase;       // its address for this lane (lds_rd + slot * CHUNK_BYTES)
    int gdyn;                    // group within the chunk, for read4_dyn only
    unsigned long long t_wait, t_bar, t_dma;     // DBG & 16 (diagnostic build only): cycles in the DMA wait / barrier / DMA issue
    unsigned long long t_lds, t_cal;             // DBG & 32: cycles waiting for a group's fragments; stamp-pair calibration
    // DBG & 32: wait for the fragments of the group about to be multiplied, timed
    __device__ __forceinline__ void timed_landed(f16x8 (&f)[4]) {
        if (!(DBG & 32)) return;
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
        const unsigned long long b = __builtin_amdgcn_s_memtime();
        t_lds += b - a;
    }

    template <int U>
    __device__ __forceinline__ void dma_pieces(lds_byte* dst, uint32_t so) {
        if constexpr (U < DMA_PER_WAVE) {
            // the 12-bit immediate covers four pieces; beyond that the SGPR offset and the LDS base move
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (U >> 2) * 4 * FRAG_BYTES),
                                                     16, voff, so + (U >> 2) * 4 * FRAG_BYTES, (U & 3) * FRAG_BYTES, 0);
            dma_pieces<U + 1>(dst, so);
        }
    }
    // piece U of the chunk `issued` (pieces may be issued in different groups: the chunk counter moves with the last one)
    template <int U>
    __device__ __forceinline__ void issue_piece() {
        if (DBG & 1) { if (U == DMA_PER_WAVE - 1) ++issued; return; }
        const int c = issued < total ? issued : total - 1;
        lds_byte* dst = lds_dma + (uint32_t)(issued % NBUF) * CHUNK_BYTES;
        if (OPT & 4) {
            const uint32_t so = __builtin_amdgcn_readfirstlane(soff0 + (uint32_t)c * CHUNK_BYTES);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (U >> 2) * 4 * FRAG_BYTES),
                                                     16, voff, so + (U >> 2) * 4 * FRAG_BYTES, (U & 3) * FRAG_BYTES, 0);
        } else {
            const unsigned char* src = gsrc + (size_t)c * CHUNK_BYTES;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + U * FRAG_BYTES),
                                             (__attribute__((address_space(3))) void*)(dst + U * FRAG_BYTES), 16, 0, 0);
        }
        if (U == DMA_PER_WAVE - 1) ++issued;
    }
    template <int U0>
    __device__ __forceinline__ void issue_pieces_from() {
        if constexpr (U0 < DMA_PER_WAVE) {
            issue_piece<U0>();
            issue_pieces_from<U0 + 1>();
        }
    }
    template <int G>
    __device__ __forceinline__ void issue_at_group() {
        if (!(OPT & 64)) return;
        if constexpr (DMA_PER_WAVE == 2) {
            if (G == DMA_G0) issue_piece<0>();
            if (G == DMA_G1) issue_piece<1>();
        } else {
            if (G == DMA_G0) issue_pieces_from<0>();
        }
    }
    __device__ __forceinline__ void issue() {
        if (DBG & 1) { ++issued; return; }
        const int c = issued < total ? issued : total - 1;
        lds_byte* dst = lds_dma + (uint32_t)(issued % NBUF) * CHUNK_BYTES;
        if (OPT & 4) {
            const uint32_t so = __builtin_amdgcn_readfirstlane(soff0 + (uint32_t)c * CHUNK_BYTES);
            dma_pieces<0>(dst, so);
        } else {
            const unsigned char* src = gsrc + (size_t)c * CHUNK_BYTES;
#pragma unroll
            for (int u = 0; u < DMA_PER_WAVE; ++u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + u * FRAG_BYTES),
                                                 (__attribute__((address_space(3))) void*)(dst + u * FRAG_BYTES), 16, 0, 0);
        }
        ++issued;
    }
    __device__ __forceinline__ void certify_next() {
        if (DBG & 16) {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (DEPTH - 1)) : "memory");
            const unsigned long long b = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long c = __builtin_amdgcn_s_memtime();
            if (!(OPT & 64)) issue();
            const unsigned long long d = __builtin_amdgcn_s_memtime();
            t_wait += b - a; t_bar += c - b; t_dma += d - c;
            return;
        }
        if (!(DBG & 1)) {
            // OPT & 512 with the delayed DMA issue (64): a half whose barrier sits behind the issue group has already
            // issued the pieces of one more chunk when it waits
            constexpr int AHEAD = ((OPT & 512) != 0 && (OPT & 64) != 0) ? int(DMA_G0 < LAG_) + int(DMA_G1 < LAG_) : 0;
            if (OPT & 128) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (DEPTH - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (DEPTH - 1) + AHEAD) : "memory");
        }
        if (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        if (!(OPT & 64)) issue();
    }
    __device__ __forceinline__ void start(const unsigned char* stream, int total_chunks, lds_byte* lds, int wave, int lane) {
        gsrc = stream + wave * DMA_PER_WAVE * FRAG_BYTES + lane * 16;
        if (OPT & 4) {
            rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(stream), 0,
                                                     (uint32_t)total_chunks * CHUNK_BYTES, 0x00020000);
            voff = (uint32_t)lane * 16;
            soff0 = (uint32_t)wave * DMA_PER_WAVE * FRAG_BYTES;
        }
        lds_dma = lds + wave * DMA_PER_WAVE * FRAG_BYTES;
        lds_rd = lds + lane * 16;
        issued = 0;
        total = total_chunks;
        slot = -1;
        cbase = lds_rd;
        gdyn = 0;
        t_wait = t_bar = t_dma = t_lds = 0;
        t_cal = 0;
        if (DBG & 32) {
            const unsigned long long a = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long b = __builtin_amdgcn_s_memtime();
            t_cal = b - a;
        }
#pragma unroll
        for (int c = 0; c < DEPTH + 1; ++c) issue();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * DEPTH) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    // first read of a chunk: certify the one after it, move to its slot
    __device__ __forceinline__ void next_chunk() {
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        cbase = lds_rd + (uint32_t)slot * CHUNK_BYTES;
    }
    // OPT & 512: the chunk's barrier (certify the next chunk, refill a free slot) in front of group LAG_ of every chunk:
    // group 0 for waves 0-3, group 2 for waves 4-7.  Barrier j of either half: chunk j + 1 certified (it is read after
    // the barrier by both), chunk j + DEPTH + 1 issued into the slot of chunk j - 2, which both halves have left.
    template <int G>
    __device__ __forceinline__ void staggered_barrier() {
        if ((OPT & 512) && G == LAG_) certify_next();
    }
    // Fragment group G (0..3, a compile-time constant) of the current chunk: a chunk is 4 groups of 4 fragment sets, every
    // GEMM / FFN step starts on a chunk boundary and its loops are unrolled, so the position inside the chunk is known
    // at compile time: the four reads are one base register + immediate offsets, and the ring bookkeeping (slot
    // wrap-around, barrier, DMA) runs once per chunk instead of the per-group address arithmetic and boundary test.
    template <int G8>                       // position in a PAIR of chunks (0..7); the group in its chunk is G8 & 3
    __device__ __forceinline__ void read4(f16x8 (&f)[4]) {
        constexpr int G = G8 & 3;
        constexpr bool BAR = (OPT & 512) ? false : ((OPT & 128) ? G8 == 0 : G == 0);      // this group opens a barrier interval
        static_assert(G8 >= 0 && G8 < 8 && CHUNK_FRAGS == 16, "four groups of four fragment sets per chunk");
        static_assert(!(OPT & 128) || ((OPT & 64) && DEPTH == 4), "OPT 128 needs OPT 64, no stagger, DEPTH 4");
        staggered_barrier<G>();
        if (BAR && !(OPT & 32)) certify_next();
        if (G == 0) next_chunk();
        // DBG & 64 (diagnostic build only): fragment reads for ONE group in four - the LDS -> VGPR traffic a kernel would
        // have in which a fragment set feeds four row tiles (the hybrid row-tile-owner / column-split proposal)
        if ((DBG & 4) || ((DBG & 64) && G != 0) || ((DBG & 128) && (G & 1))) {       // 128: one group in TWO (pairs of waves sharing)
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(f[u]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                f[u] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(cbase + (4 * G + u) * FRAG_BYTES);
        }
        if (BAR && (OPT & 32)) certify_next();
        issue_at_group<G>();
    }
    // the same with the group index at run time (heads: 10 groups per hidden tile); align() before the first use
    __device__ __forceinline__ void align() { gdyn = 0; }
    __device__ __forceinline__ void read4_dyn(f16x8 (&f)[4]) {
        const int g = gdyn & 3;
        const bool bar = (OPT & 512) ? false : ((OPT & 128) ? gdyn == 0 : g == 0);
        if ((OPT & 512) && g == LAG_) certify_next();
        if (bar && !(OPT & 32)) certify_next();
        if (g == 0) next_chunk();
        const lds_byte* a = cbase + (uint32_t)g * (4 * FRAG_BYTES);
        if ((DBG & 4) || ((DBG & 64) && g != 0) || ((DBG & 128) && (g & 1))) {
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(f[u]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                f[u] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>(a + u * FRAG_BYTES);
        }
        if (bar && (OPT & 32)) certify_next();
        if (OPT & 64) {
            if constexpr (DMA_PER_WAVE == 2) {
                if (g == DMA_G0) issue_piece<0>();
                if (g == DMA_G1) issue_piece<1>();
            } else {
                if (g == DMA_G0) issue_pieces_from<0>();
            }
        }
        gdyn = (gdyn + 1) & 7;
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

__device__ __forceinline__ f32x4 mfma(const f16x8& a, const f16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// one group of 4 fragment sets {Ah(t0), Al(t0), Ah(t1), Al(t1)} against one B k-step (bh, bl): 6 MFMAs, two accumulators interleaved
__device__ __forceinline__ void group6(const f16x8 (&a)[4], const f16x8& bh, const f16x8& bl, f32x4& c0, f32x4& c1) {
    if (OPT & 1024) __builtin_amdgcn_s_setprio(1);       // (round 4 A/B: issue priority raised for the six MFMAs of a group)
    c0 = mfma(a[0], bl, c0);
    c1 = mfma(a[2], bl, c1);
    c0 = mfma(a[1], bh, c0);
    c1 = mfma(a[3], bh, c1);
    c0 = mfma(a[0], bh, c0);
    c1 = mfma(a[2], bh, c1);
    if (OPT & 1024) __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ float reduce_max4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float reduce_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void row_scale(const f32x4 (&x)[16], float& s, float& inv) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m = fmaxf(m, __builtin_fabsf(x[t][r]));
    m = reduce_max4(m);
    int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);
    eb = eb < 87 ? 87 : (eb > 250 ? 250 : eb);                  // see rowowner.hpp row_scale
    s = __uint_as_float((uint32_t)(127 + TARGET_EXP + 127 - eb) << 23);
    inv = __uint_as_float((uint32_t)(eb - TARGET_EXP) << 23);
}

// planes of one k-step from two adjacent tiles (elements 0..3 from `a`, 4..7 from `b`), scaled by s
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, float s, f16x8& h, f16x8& l) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = (j < 4 ? a[j & 3] : b[j & 3]) * s;
        const _Float16 hh = (_Float16)v;
        h[j] = hh;
        l[j] = (_Float16)(v - (float)hh);
    }
}

__device__ __forceinline__ f32x4 param4(lds_cfloat* pb, int off, int tile) {      // features 16 tile + 4 g + {0..3} (pb carries 4 g)
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(pb + off + 16 * tile);
}

// planes + initial accumulators, tile pair by tile pair (x[2ks], x[2ks+1] die as they are consumed)
template <bool WITH_X>
__device__ __forceinline__ void prepare(const f32x4 (&x)[16], float s, lds_cfloat* pb, int bias, float scale,
                                        f16x8 (&xh)[8], f16x8 (&xl)[8], f32x4 (&acc)[16]) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        split8(x[2 * ks], x[2 * ks + 1], s, xh[ks], xl[ks]);
#pragma unroll
        for (int t = 2 * ks; t < 2 * ks + 2; ++t) {
            const f32x4 b = param4(pb, bias, t);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = ((WITH_X ? x[t][r] : 0.f) + b[r]) * scale;
        }
    }
}

// groups I .. 63 of one 256 x 256 GEMM (group I = k-step I / 8, tile pair I % 8); `cur` holds group I's fragments
template <class RingX, int I>
__device__ __forceinline__ void gemm256_groups(RingX& ring, f16x8 (&cur)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8],
                                               f32x4 (&acc)[16]) {
    constexpr int ks = I >> 3, tp = I & 7;
    f16x8 nxt[4];
    if constexpr (I < 63) ring.template read4<(I + 1) & 7>(nxt);
    ring.timed_landed(cur);
    group6(cur, xh[ks], xl[ks], acc[2 * tp], acc[2 * tp + 1]);
    if constexpr (I < 63) gemm256_groups<RingX, I + 1>(ring, nxt, xh, xl, acc);
}
template <class RingX, int I0>
__device__ __forceinline__ void gemm256_from(RingX& ring, const f16x8 (&xh)[8], const f16x8 (&xl)[8], f32x4 (&acc)[16]) {
    f16x8 cur[4];
    ring.template read4<0>(cur);
    gemm256_groups<RingX, I0>(ring, cur, xh, xl, acc);
}

template <class RingX>
__device__ __forceinline__ void gemm256(RingX& ring, const f16x8 (&xh)[8], const f16x8 (&xl)[8], f32x4 (&acc)[16]) {
    gemm256_from<RingX, 0>(ring, xh, xl, acc);
}

__device__ __forceinline__ void layer_norm(f32x4 (&y)[16], lds_cfloat* pb, int gamma, int beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += y[t][r];
    const float mean = reduce_sum4(s) * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = y[t][r] - mean;
            q += d * d;
        }
    const float rstd = 1.0f / sqrtf(reduce_sum4(q) * (1.0f / 256.0f) + eps);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 ga = param4(pb, gamma, t), be = param4(pb, beta, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[t][r] = (y[t][r] - mean) * rstd * ga[r] + be[r];
    }
}

// LayerNorm of y = acc * un (un a power of two: acc * un is exact, and so is sum(acc) * un == sum(acc * un) barring fp32
// denormals): the unscale rides in the mean and in the deviation's fma, the deviation is kept in place
__device__ __forceinline__ void layer_norm_scaled(const f32x4 (&acc)[16], float un, f32x4 (&y)[16], lds_cfloat* pb, int gamma,
                                                  int beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[t][r];
    const float mean = reduce_sum4(s) * un * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = __builtin_fmaf(acc[t][r], un, -mean);
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
}

template <class RingX>
__device__ __forceinline__ void phase_attn_ln(RingX& ring, const Phase& P, f32x4 (&x)[16], lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
    f32x4 acc[16];
    prepare<true>(x, s, pb, P.b1, s * P.sw1, xh, xl, acc);
    gemm256(ring, xh, xl, acc);
    const float un = inv / P.sw1;
    if (OPT & 2) {
        layer_norm_scaled(acc, un, x, pb, P.gamma, P.beta, P.ln_eps);
        return;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = acc[t][r] * un;
    layer_norm(x, pb, P.gamma, P.beta, P.ln_eps);
}

// hidden tile (two 16-feature accumulators = one k-step of stage 2) -> planes
// `lim` = 60000 / c (exact: c is a power of two)
__device__ __forceinline__ void hidden_planes(const f32x4& a0, const f32x4& a1, float c, float lim, f16x8& hh, f16x8& hl) {
    if (DBG & 8) {
        asm volatile("" : "+v"(hh), "+v"(hl) : "v"(a0), "v"(a1));
        return;
    }
    f32x4 t0, t1;
    if (OPT & 1) {
        // min(max(a, 0) * c, 60000) == med3(a, 0, 60000 / c) * c for a power-of-two c: relu and clamp are one instruction
        // in the unscaled domain and the scale rides in the split's fma_mix instructions
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t0[r] = __builtin_amdgcn_fmed3f(a0[r], 0.f, lim);
            t1[r] = __builtin_amdgcn_fmed3f(a1[r], 0.f, lim);
        }
        split8(t0, t1, c, hh, hl);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t0[r] = fminf(fmaxf(a0[r], 0.f) * c, 60000.f);
        t1[r] = fminf(fmaxf(a1[r], 0.f) * c, 60000.f);
    }
    split8(t0, t1, 1.0f, hh, hl);
}
__device__ __forceinline__ void init_pair(f32x4& a0, f32x4& a1, lds_cfloat* pb, int bias, int tile0, float scale) {
    const f32x4 b0 = param4(pb, bias, tile0), b1 = param4(pb, bias, tile0 + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a0[r] = b0[r] * scale;
        a1[r] = b1[r] * scale;
    }
}

template <class RingX, bool S1, bool S2, int GI>
__device__ __forceinline__ void ffn_groups(RingX& ring, f16x8 (&cur)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8],
                                           f32x4& a10, f32x4& a11, f32x4 (&acc2)[16], const f16x8& hh, const f16x8& hl) {
    constexpr int NG = (S1 ? 8 : 0) + (S2 ? 8 : 0);          // groups in this step (a multiple of 4: whole chunks)
    constexpr bool is1 = S1 && (!S2 || (GI & 1) == 0);
    constexpr int u = (S1 && S2) ? GI >> 1 : GI;
    f16x8 nxt[4];
    if constexpr (GI < NG - 1) ring.template read4<(GI + 1) & 7>(nxt);
    ring.timed_landed(cur);
    if constexpr (is1) group6(cur, xh[u], xl[u], a10, a11);
    else group6(cur, hh, hl, acc2[2 * u], acc2[2 * u + 1]);
    if constexpr (GI < NG - 1) ffn_groups<RingX, S1, S2, GI + 1>(ring, nxt, xh, xl, a10, a11, acc2, hh, hl);
}

// One FFN step: 8 x { stage-1 group (W_1 tiles 2t, 2t+1 at ks = u), stage-2 group (W_2 tiles 2u, 2u+1 at k-step t-1) }
template <class RingX, bool S1, bool S2>
__device__ __forceinline__ void ffn_step(RingX& ring, const f16x8 (&xh)[8], const f16x8 (&xl)[8], f32x4& a10, f32x4& a11,
                                         f32x4 (&acc2)[16], const f16x8& hh, const f16x8& hl) {
    f16x8 cur[4];
    ring.template read4<0>(cur);
    ffn_groups<RingX, S1, S2, 0>(ring, cur, xh, xl, a10, a11, acc2, hh, hl);
}

template <class RingX>
__device__ __forceinline__ void phase_ffn_ln(RingX& ring, const Phase& P, f32x4 (&x)[16], lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
    const float sh = x3::hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));
    f32x4 acc2[16];
    prepare<true>(x, s, pb, P.b2, P.sw2 * sh, xh, xl, acc2);
    const float b1s = s * P.sw1, c1 = sh * inv / P.sw1, lim1 = 60000.f / c1;
    f32x4 a10, a11;
    f16x8 hh, hl;
    init_pair(a10, a11, pb, P.b1, 0, b1s);
    ffn_step<RingX, true, false>(ring, xh, xl, a10, a11, acc2, hh, hl);
    for (int t = 1; t < P.n_steps; ++t) {
        hidden_planes(a10, a11, c1, lim1, hh, hl);
        init_pair(a10, a11, pb, P.b1, 2 * t, b1s);
        ffn_step<RingX, true, true>(ring, xh, xl, a10, a11, acc2, hh, hl);
    }
    hidden_planes(a10, a11, c1, lim1, hh, hl);
    ffn_step<RingX, false, true>(ring, xh, xl, a10, a11, acc2, hh, hl);
    const float un = 1.0f / (P.sw2 * sh);
    if (OPT & 2) {
        layer_norm_scaled(acc2, un, x, pb, P.gamma, P.beta, P.ln_eps);
        return;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] = acc2[t][r] * un;
    layer_norm(x, pb, P.gamma, P.beta, P.ln_eps);
}

// row I/O: lane (q, g) moves the 16-byte groups [16 T + 4 g, +4) of row q (64 contiguous bytes per row and instruction)
__device__ __forceinline__ void load_rows(f32x4 (&x)[16], const float* row_ptr, int g) {
#pragma unroll
    for (int t = 0; t < 16; ++t) x[t] = *reinterpret_cast<const f32x4*>(row_ptr + 16 * t + 4 * g);
}
__device__ __forceinline__ void add_rows(f32x4 (&x)[16], const float* row_ptr, int g) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row_ptr + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[t][r] += v[r];
    }
}
__device__ __forceinline__ void store_rows(const f32x4 (&x)[16], float* row_ptr, int g) {
#pragma unroll
    for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(row_ptr + 16 * t + 4 * g) = x[t];
}

// one QUARTER of a cross layer's 256 x 256 GEMM: output tiles 4 Q4 .. 4 Q4 + 3 (two tile pairs), 16 groups = 4 chunks,
// k-step major (amdrec/weights.py x3b_stream_cross); `cur` holds group I's fragments
template <class RingX, int I>
__device__ __forceinline__ void cross_quarter_groups(RingX& ring, f16x8 (&cur)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8],
                                                     f32x4 (&acc)[4]) {
    constexpr int ks = I >> 1, pr = I & 1;
    f16x8 nxt[4];
    if constexpr (I < 15) ring.template read4<(I + 1) & 7>(nxt);
    ring.timed_landed(cur);
    group6(cur, xh[ks], xl[ks], acc[2 * pr], acc[2 * pr + 1]);
    if constexpr (I < 15) cross_quarter_groups<RingX, I + 1>(ring, nxt, xh, xl, acc);
}

// xl <- x0 * (xl W + b) + xl with x0 IN REGISTERS (round 2 wrote the trunk's output to HBM once and read it back in each of
// the three cross layers: 1.05 of the launch's 1.99 GB of HBM traffic).  x0 (64 registers) + the residual xl (64) + the
// planes of xl (64) leave room for a 16-register accumulator, so the GEMM runs in four quarters of the output features;
// a quarter's epilogue updates its own four tiles of xl in place - the planes were taken from the old xl, and a tile's
// residual is its own old value.  Same MFMAs in the same order per output element as the undivided GEMM: bit-identical.
template <class RingX>
__device__ __forceinline__ void phase_cross(RingX& ring, const Phase& P, f32x4 (&xl_)[16], const f32x4 (&x0)[16], lds_cfloat* pb) {
    float s, inv;
    row_scale(xl_, s, inv);
    f16x8 xh[8], xl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) split8(xl_[2 * ks], xl_[2 * ks + 1], s, xh[ks], xl[ks]);
    const float bs = s * P.sw1, un = inv / P.sw1;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 b = param4(pb, P.b1, 4 * q4 + i);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = b[r] * bs;
        }
        f16x8 cur[4];
        ring.template read4<0>(cur);
        cross_quarter_groups<RingX, 0>(ring, cur, xh, xl, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) xl_[4 * q4 + i][r] = x0[4 * q4 + i][r] * (acc[i][r] * un) + xl_[4 * q4 + i][r];
    }
}

// One step of the heads, software-pipelined like the FFN (round 3; the heads used to finish a hidden tile's stage 1, convert
// it and only then run its stage 2: a dependency bubble per tile and two fragment reads with nothing to hide behind - 5.45 us
// per chunk against the FFN's 4.55): the stage-1 groups of hidden tile tt (8: W_1 tiles 2tt, 2tt+1 at ks = u) with the two
// stage-2 groups of tile tt - 1 (W_2 tile pairs 0 and 1 of that tile's task, at k-step = its index in the task) riding behind
// u = 3 and u = 7.  Stream order: amdrec/weights.py x3b_stream_heads.  P0 = position of the step's first group in its
// chunk (0 or 2: a full step is 10 groups = 2.5 chunks).
template <class RingX, bool S1, bool S2, int P0, int GI>
__device__ __forceinline__ void heads_groups(RingX& ring, f16x8 (&cur)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8],
                                             f32x4& a10, f32x4& a11, f32x4 (&acc2)[4], const f16x8& hh, const f16x8& hl) {
    static_assert(!(OPT & 128), "the heads' chunk positions are tracked modulo one chunk");
    constexpr int NG = (S1 ? 8 : 0) + (S2 ? 2 : 0);
    // (S1 && S2): u0 u1 u2 u3 p0 u4 u5 u6 u7 p1
    constexpr bool is2 = S2 && (!S1 || GI == 4 || GI == 9);
    constexpr int u = !S1 ? 0 : (S2 ? (GI < 4 ? GI : GI - 1) : GI);
    constexpr int pr = !S1 ? GI : (GI == 9 ? 1 : 0);
    f16x8 nxt[4];
    if constexpr (GI < NG - 1) ring.template read4<(P0 + GI + 1) & 3>(nxt);
    ring.timed_landed(cur);
    if constexpr (is2) group6(cur, hh, hl, acc2[2 * pr], acc2[2 * pr + 1]);
    else group6(cur, xh[u], xl[u], a10, a11);
    if constexpr (GI < NG - 1) heads_groups<RingX, S1, S2, P0, GI + 1>(ring, nxt, xh, xl, a10, a11, acc2, hh, hl);
}
template <class RingX, bool S1, bool S2, int P0>
__device__ __forceinline__ void heads_step(RingX& ring, const f16x8 (&xh)[8], const f16x8 (&xl)[8], f32x4& a10, f32x4& a11,
                                           f32x4 (&acc2)[4], const f16x8& hh, const f16x8& hl) {
    f16x8 cur[4];
    ring.template read4<P0>(cur);
    heads_groups<RingX, S1, S2, P0, 0>(ring, cur, xh, xl, a10, a11, acc2, hh, hl);
}

template <class RingX>
__device__ __forceinline__ void phase_heads(RingX& ring, const Program& G, const Phase& P, const f32x4 (&x)[16], float* out,
                                            long long ld_out, long long row, bool row_ok, int g, lds_cfloat* pb) {
    float s, inv;
    row_scale(x, s, inv);
    f16x8 xh[8], xl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) split8(x[2 * ks], x[2 * ks + 1], s, xh[ks], xl[ks]);
    const float sh = x3::hidden_scale(fmaf(P.hn * 8192.0f, inv, P.hb));
    const float b1s = s * P.sw1, c1 = sh * inv / P.sw1, un2 = 1.0f / (P.sw2 * sh), lim1 = 60000.f / c1;
    const int T = P.n_steps, NT = P.n_tasks * T;               // hidden tiles per task / in all (NT even: x3_build)
    f32x4 acc2[4], a10, a11;
    f16x8 hh, hl;
    init_pair(acc2[0], acc2[1], pb, G.hb2[0], 0, P.sw2 * sh);
    init_pair(acc2[2], acc2[3], pb, G.hb2[0], 2, P.sw2 * sh);
    init_pair(a10, a11, pb, P.b1, 0, b1s);
    heads_step<RingX, true, false, 0>(ring, xh, xl, a10, a11, acc2, hh, hl);                   // tile 0: stage 1 only
    int task = 0, t_in = 0;                                    // task / index of the tile whose stage 2 runs in step tt
    for (int tt = 1; tt <= NT; ++tt) {
        hidden_planes(a10, a11, c1, lim1, hh, hl);             // tile tt - 1
        if (tt < NT) {
            init_pair(a10, a11, pb, P.b1, 2 * tt, b1s);        // stacked b_1: tile tt of all tasks' hidden units
            if (tt & 1) heads_step<RingX, true, true, 0>(ring, xh, xl, a10, a11, acc2, hh, hl);
            else heads_step<RingX, true, true, 2>(ring, xh, xl, a10, a11, acc2, hh, hl);
        } else {
            heads_step<RingX, false, true, 2>(ring, xh, xl, a10, a11, acc2, hh, hl);           // NT even: the last step starts at 2
        }
        if (++t_in == T) {                                     // that was the task's last hidden tile: its 64 outputs are complete
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 w = param4(pb, G.hw3[task], t);
#pragma unroll
                for (int r = 0; r < 4; ++r) dot += fmaxf(acc2[t][r] * un2, 0.f) * w[r];
            }
            dot = reduce_sum4(dot);
            if (g == 0 && row_ok) out[(long long)task * ld_out + row] = dot + pb[G.hb3[task]];  // g == 0: pb carries no offset
            t_in = 0;
            if (++task < P.n_tasks) {
                init_pair(acc2[0], acc2[1], pb, G.hb2[task], 0, P.sw2 * sh);
                init_pair(acc2[2], acc2[3], pb, G.hb2[task], 2, P.sw2 * sh);
            }
        }
    }
}

// everything after the parameter blob is in LDS: ring start, row load, the phases, logits
template <class RingX>
__device__ __forceinline__ void run_chain(const Program& G, const Input& in, long long rows, float* scratch, float* x_out,
                                          long long ld_xout, float* logits, long long ld_logits, unsigned char* smem, int wave,
                                          int lane, lds_cfloat* pb) {
    const int g = lane >> 4, q = lane & 15;
    const long long row = (long long)blockIdx.x * ROWS_PER_WG + wave * ROWS_PER_WAVE + q;
    const bool row_ok = row < rows;
    const long long rowc = row_ok ? row : rows - 1;
    RingX ring;
    const unsigned long long t_begin = (DBG & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
    ring.start(G.stream, G.total_chunks, (lds_byte*)smem, wave, lane);

    f32x4 x[16];
    if (in.X != nullptr) {
        load_rows(x, in.X + rowc * in.ldx, g);
    } else {
        const long long gr = in.row_base + rowc;
        long long a = in.rowmap ? in.rowmap[gr] : gr;
        a = a < 0 ? 0 : (a >= in.n_cache ? in.n_cache - 1 : a);
        load_rows(x, in.cache + a * in.ldc, g);
        add_rows(x, in.U + (gr / in.rowdiv) * 256, g);
    }
    // The chain is [encoder phases] [cross phases] [heads] (ranker_x3.hip x3_build), walked as three loops so that x0 - the
    // trunk's output, 64 registers - is live only across the cross layers.
    int p = 0;
    for (; p < G.n_phases; ++p) {
        const Phase& P = G.ph[p];
        const int type = __builtin_amdgcn_readfirstlane(P.type);
        if (type == x3::PH_ATTN_LN) phase_attn_ln(ring, P, x, pb);
        else if (type == x3::PH_FFN_LN) phase_ffn_ln(ring, P, x, pb);
        else break;
    }
    if (p < G.n_phases && __builtin_amdgcn_readfirstlane(G.ph[p].type) == x3::PH_CROSS) {
        f32x4 x0[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x0[t] = x[t];
        for (; p < G.n_phases && __builtin_amdgcn_readfirstlane(G.ph[p].type) == x3::PH_CROSS; ++p)
            phase_cross(ring, G.ph[p], x, x0, pb);
    }
    if (p < G.n_phases) phase_heads(ring, G, G.ph[p], x, logits, ld_logits, row, row_ok, g, pb);
    if (x_out != nullptr && row_ok) store_rows(x, x_out + row * ld_xout, g);
    ring.drain();
    if ((DBG & 16) && lane == 0) {            // diagnostic build: cycle stamps into the unused tail of the logits buffer
        float* dbg = logits + 3 * ld_logits + ((long long)blockIdx.x * WAVES + wave) * 4;
        dbg[0] = (float)(__builtin_amdgcn_s_memtime() - t_begin);
        dbg[1] = (DBG & 32) ? (float)ring.t_lds : (float)ring.t_wait;
        dbg[2] = (DBG & 32) ? (float)ring.t_cal : (float)ring.t_bar;
        dbg[3] = (float)ring.t_dma;
    }
}

// (the 4-wave shape could take 512 registers per wave; bounded to 256 like the 8-wave shape it ran 3 % FASTER: no AGPR traffic)
__global__ __launch_bounds__(64 * WAVES, 2) void ranker_x3b_kernel(Program G, Input in, long long rows, float* scratch,
                                                            float* x_out, long long ld_xout, float* logits,
                                                            long long ld_logits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;

    lds_byte* pbase = (lds_byte*)smem + RING_BYTES;
    for (int o = 0; o + wave * 1024 < G.n_params * 4; o += WAVES * 1024)  // WAVES x 1 KB per pass; n_params % 1024 == 0
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(
                                             reinterpret_cast<const unsigned char*>(G.params) + o + tid * 16),
                                         (__attribute__((address_space(3))) void*)(pbase + o + wave * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_cfloat* pb = reinterpret_cast<lds_cfloat*>(pbase) + 4 * g;

    if ((OPT & 16) && wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
    if ((OPT & 512) && wave >= WAVES / 2)
        run_chain<RingT<2>>(G, in, rows, scratch, x_out, ld_xout, logits, ld_logits, smem, wave, lane, pb);
    else
        run_chain<RingT<0>>(G, in, rows, scratch, x_out, ld_xout, logits, ld_logits, smem, wave, lane, pb);
}

}  // namespace AMDREC_X3B_NAMESPACE
}  // namespace amdrec
#pragma clang diagnostic pop
