// Block-wide selection helpers shared by the flat and IVF search kernels (LDS histograms for radix
// select, bitonic sort of 64-bit keys, result write-out).
#pragma once
#include "common.hpp"

namespace amdrec {

// ---- block-wide helpers -------------------------------------------------------------
// Find, scanning bins from the top, the bin that holds the r-th largest element.
// hist[NB] in LDS; returns bin and updates r to the rank inside that bin.  All threads call.
template <int NB, int NT>
__device__ inline int find_bin_desc(const int* hist, int& r, int* scratch /*[NT+2]*/) {
    // Thread t owns bins [t*PER, (t+1)*PER).  A suffix scan over the threads' sums (wave shuffles + one LDS round
    // over the wave totals) gives each thread the count of keys in bins above its own; exactly one thread's range
    // holds the r-th largest key.  (A serial scan by one lane cost ~25 us per call at NT = 512.)
    constexpr int PER = NB / NT;
    constexpr int NW = NT / 64;
    static_assert(NB % NT == 0 && NT % 64 == 0, "bins must split evenly over whole waves");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int local = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) local += hist[tid * PER + i];
    int incl = local;                               // inclusive suffix sum within the wave (lanes >= mine)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += v;
    }
    if (lane == 0) scratch[wv] = incl;              // wave total
    if (tid == 0) scratch[NT] = -1;
    __syncthreads();
    int above = incl - local;                       // keys in higher bins of this wave ...
#pragma unroll
    for (int x = 0; x < NW; ++x)
        if (x > wv) above += scratch[x];            // ... and of the higher waves
    if (above < r && r <= above + local) {          // the r-th largest lies in my bins (exactly one thread)
        int bin = -1;
#pragma unroll
        for (int i = PER - 1; i >= 0; --i) {
            const int h = hist[tid * PER + i];
            if (bin < 0) {
                if (above + h >= r) bin = tid * PER + i;
                else above += h;
            }
        }
        scratch[NT] = bin;
        scratch[NT + 1] = r - above;
    }
    __syncthreads();
    const int bin = scratch[NT];
    r = scratch[NT + 1];
    __syncthreads();
    return bin;
}

// Bitonic sort of P (power of two) 64-bit keys in LDS, descending.  All threads call; blockDim.x must be a
// multiple of 64.  The network is indexed by compare-exchange PAIR (pair p -> elements i, i | st with the st
// bit spliced out of p), so every lane does useful work.  A pair with st < 32 lies inside the 64 consecutive
// elements owned by one wave's pair group, and a wave's LDS operations execute in order, so those stages need
// only a wave-level fence; block barriers are paid for st >= 32 and at the end of each outer stage.
__device__ inline void bitonic_desc(unsigned long long* buf, int P) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int half = P >> 1;
    for (int sz = 2; sz <= P; sz <<= 1) {
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int p = tid; p < half; p += nt) {
                const int i = ((p & ~(st - 1)) << 1) | (p & (st - 1));
                const int j = i | st;
                const unsigned long long a = buf[i], b = buf[j];
                const bool desc = (i & sz) == 0;
                if (desc ? (a < b) : (a > b)) { buf[i] = b; buf[j] = a; }
            }
            if (st >= 32 || st == 1) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// Descending sort of m <= 512 R keys in LDS by a 512-thread block (R = 2 or 4 keys per lane), keys UNIQUE and non-zero
// (0 = padding): `src`[0 .. m) -> `dst`[0 .. m) (two different arrays of >= 512 R keys; `src` is overwritten; positions of
// `dst` that receive no key keep their contents: clear them first if zeros can occur among the m).  Each of the 8 waves
// sorts a run of 64 R keys in registers (a bitonic network on shuffles: 28 stages at R = 2, 36 at R = 4), the runs go back to
// `src`, and every key finds its final position as its index in its own run + the number of larger keys in each other run
// (a binary search per run: the runs are sorted and the keys unique).  ~4 us at 1024 keys against ~17 for the all-LDS
// bitonic network (66 stages, most of them behind a block barrier) - the single-query search's finalize was the sort.
template <int R>
__device__ inline void sort_desc_runs(unsigned long long* src, unsigned long long* dst, int m) {
    static_assert(R == 2 || R == 4, "two or four keys per lane");
    constexpr int RUN = 64 * R, STEPS = R == 2 ? 8 : 9;      // binary search over 0 .. RUN
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned long long key[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = RUN * w + 64 * r + lane;
        key[r] = i < m ? src[i] : 0ull;
    }
    __syncthreads();                                        // every key is in registers: `src` may be overwritten
#pragma unroll
    for (int k = 2; k <= RUN; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {                                   // the partner is another register of the same lane
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int pr = r ^ (j >> 6);
                    if (pr < r) continue;                    // each pair once, from its lower element
                    const bool desc = ((64 * r) & k) == 0;
                    const unsigned long long a = key[r], b = key[pr];
                    const bool swap = desc ? (a < b) : (a > b);
                    key[r] = swap ? b : a;
                    key[pr] = swap ? a : b;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = 64 * r + lane;
                    const unsigned long long other = __shfl_xor(key[r], j, 64);
                    const bool lower = (lane & j) == 0;          // i < partner
                    const bool desc = (i & k) == 0;              // this k-block ends descending
                    const bool take_max = lower == desc;
                    const bool other_greater = other > key[r];
                    key[r] = (take_max == other_greater) ? other : key[r];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) src[RUN * w + 64 * r + lane] = key[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (key[r] == 0ull) continue;
        int rank = 64 * r + lane;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            if (v == w) continue;
            const unsigned long long* run = src + RUN * v;
            int lo = 0, hi = RUN;
#pragma unroll
            for (int it = 0; it < STEPS; ++it) {             // keys of the run greater than mine: 0 .. RUN (zeros are smaller than any key)
                const int mid = (lo + hi) >> 1;
                if (lo < hi && run[mid] > key[r]) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        dst[rank] = key[r];
    }
    __syncthreads();
}

// |bf16 inner product - fp32 inner product| of a query against any corpus row (derivation: search.hip, "Error bound of the
// bf16 pass"): qn = ||q||, dqn = ||bf16(q) - q||, M / D = largest row norm / largest row rounding-error norm of the corpus.
__device__ __forceinline__ float eps_bound(float qn, float dqn, float M, float D, int d) {
    return (dqn * (M + D) + qn * D) * 1.0001f + (float)d * 1.2e-7f * qn * (M + D);
}

__device__ inline void write_result(const unsigned long long* buf, int have, int k, long long q, float* outD,
                                    long long* outI, long long pos_offset) {
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        unsigned long long key = (i < have) ? buf[i] : 0ull;
        bool valid = key != 0ull;
        outD[q * k + i] = valid ? key_score(key) : -INFINITY;
        outI[q * k + i] = valid ? (long long)key_pos(key) + pos_offset : -1ll;
    }
}


}  // namespace amdrec
