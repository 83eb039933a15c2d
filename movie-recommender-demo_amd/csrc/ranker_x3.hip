// The fp16x3 row-owner engine of amdrec_ranker_forward (kernel machinery and numerics: rowowner.hpp).
//   ranker_x3_kernel : input rows (dense X or cached-projection gather) -> all phases of the chain -> logits
//   amdrec_ranker_x3_prefix : debugging / test entry: run the first n phases on a dense X and return the rows
#include "rowowner.hpp"
#include "rowowner16.hpp"
#include "rowowner16c.hpp"
#include "../../include/amdrec.h"

using namespace amdrec;

// Parameter blob layout (floats; amdrec/weights.py pack_x3_params writes exactly this): per encoder layer
// [b_ov 256 | gamma1 256 | beta1 256 | b_1 d_ff | b_2 256 | gamma2 256 | beta2 256], per cross layer [bias 256], heads
// [b_1 n_tasks*h1] then per task [b_2 64 | w_3 64 | b_3 4]; padded to a multiple of 1024.
static long long x3_param_floats(const amdrec_ranker_params* p) {
    const long long n = (long long)p->n_layers * (6 * 256 + p->d_ff) + 256ll * p->n_cross + (long long)p->n_tasks * (p->head_h1 + 132);
    return (n + 1023) / 1024 * 1024;
}

// eligibility of the engine for these parameters (the reference architecture: d_model 256, 64-wide head layer 2)
static bool x3_eligible(const amdrec_ranker_params* p) {
    if (p->x3.stream == nullptr || p->x3.chunks <= 0 || p->x3.params == nullptr) return false;
    if (p->x3.variant != 16 && p->x3.variant != 32) return false;
    if (x3_param_floats(p) > x3::PARAM_FLOATS) return false;      // the parameter blob must fit its LDS area
    if (p->d_model != 256 || p->d_ff % 32 != 0 || p->head_h1 % 32 != 0 || p->head_h2 != 64) return false;
    if (2 * p->n_layers + p->n_cross + 1 > x3::MAX_PHASES || p->n_tasks > 4) return false;
    for (int l = 0; l < p->n_layers; ++l)
        if (p->layers[l].w_v != nullptr) return false;              // needs the pre-multiplied W_ov form
    return true;
}

// the column-split stream (rowowner16c.hpp) is present and the architecture fits its super-steps of four hidden tiles
static bool x3c_available(const amdrec_ranker_params* p) {
    return p->x3.variant == 16 && p->x3.stream_cs != nullptr && p->x3.chunks_cs > 0 && p->d_ff % 128 == 0 &&
           p->head_h1 % 128 == 0;
}

// n_phases < 0: the whole chain; cs: the column-split kernel's stream
static int x3_build(const amdrec_ranker_params* p, int n_phases, x3::Program& G, bool cs = false) {
    memset(&G, 0, sizeof(G));
    int n = 0, o = 0;                                               // o: running offset into the parameter blob (floats)
    for (int l = 0; l < p->n_layers; ++l) {
        x3::Phase& A = G.ph[n++];
        A.type = x3::PH_ATTN_LN; A.b1 = o; A.gamma = o + 256; A.beta = o + 512; A.sw1 = p->x3.sw_ov[l];
        A.sw2 = 1.f; A.ln_eps = p->ln_eps;
        o += 768;
        x3::Phase& F = G.ph[n++];
        F.type = x3::PH_FFN_LN; F.n_steps = p->d_ff / 32; F.b1 = o; F.b2 = o + p->d_ff; F.gamma = o + p->d_ff + 256;
        F.beta = o + p->d_ff + 512;
        F.sw1 = p->x3.sw_1[l]; F.sw2 = p->x3.sw_2[l]; F.hn = p->x3.hn[l]; F.hb = p->x3.hb[l]; F.ln_eps = p->ln_eps;
        o += p->d_ff + 768;
    }
    for (int c = 0; c < p->n_cross; ++c) {
        x3::Phase& C = G.ph[n++];
        C.type = x3::PH_CROSS; C.b1 = o; C.sw1 = p->x3.sw_cross[c]; C.sw2 = 1.f;
        o += 256;
    }
    x3::Phase& H = G.ph[n++];
    H.type = x3::PH_HEADS; H.n_steps = p->head_h1 / 32; H.n_tasks = p->n_tasks; H.b1 = o;
    H.sw1 = p->x3.sw_h1; H.sw2 = p->x3.sw_h2; H.hn = p->x3.hn_head; H.hb = p->x3.hb_head;
    o += p->n_tasks * p->head_h1;
    for (int t = 0; t < p->n_tasks; ++t) { G.hb2[t] = o; G.hw3[t] = o + 64; G.hb3[t] = o + 128; o += 132; }
    REQUIRE(p->x3.n_params == x3_param_floats(p), "x3: parameter blob has %lld floats, the architecture needs %lld",
            (long long)p->x3.n_params, x3_param_floats(p));
    G.params = p->x3.params;
    G.n_params = (int)p->x3.n_params;
    // chunks consumed by a prefix of the chain (the ring only needs to know where the stream ends)
    const long long per_layer = 16 + 4ll * (p->d_ff / 32);          // W_ov: 16 chunks; FFN: 64 fragment sets per hidden tile
    // heads: 8 stage-1 + 2 stage-2 groups per hidden tile; column-split: per 4 hidden tiles 8 + 4 chunks (stage 2 half empty)
    const long long hidden_tiles = (long long)p->n_tasks * (p->head_h1 / 32);
    const long long heads = cs ? hidden_tiles * 3 : hidden_tiles * 40 / 16;
    REQUIRE(cs || hidden_tiles * 40 % 16 == 0, "x3: head stream is not a whole number of chunks");
    const long long total = p->n_layers * per_layer + 16ll * p->n_cross + heads;
    const long long have = cs ? p->x3.chunks_cs : p->x3.chunks;
    REQUIRE(total == have, "x3: stream length %lld chunks does not match the architecture (%lld)", have, total);
    G.n_phases = n_phases < 0 || n_phases > n ? n : n_phases;
    G.total_chunks = (int)total;
    G.stream = reinterpret_cast<const unsigned char*>(cs ? p->x3.stream_cs : p->x3.stream);
    return AMDREC_OK;
}

constexpr long long X3B4_MAX_ROWS = 256ll * 64;      // one 64-row workgroup per CU

static size_t x3_scratch_bytes(long long rows) {
    return (size_t)((rows + x3::ROWS_PER_WG - 1) / x3::ROWS_PER_WG) * x3::ROWS_PER_WG * 256 * 4;
}

static int x3_launch(const x3::Program& G, const x3::Input& in, long long rows, float* scratch, float* x_out,
                     long long ld_xout, float* logits, long long ld_logits, hipStream_t st, int variant) {
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(x3::ranker_x3_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, x3::RING_BYTES + x3::PARAM_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(x3b::ranker_x3b_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, x3::RING_BYTES + x3::PARAM_FLOATS * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(x3b4::ranker_x3b_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, x3::RING_BYTES + x3::PARAM_FLOATS * 4));
        attr_done.mark();
    }
    if (variant == 160) {                              // column-split: 16 rows per workgroup (the caller built G on that stream)
        static PerDeviceOnce attr_cs;
        if (attr_cs.pending()) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(x3c::ranker_x3c_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, x3c::LDS_BYTES));
            attr_cs.mark();
        }
        double w = 0;
        for (int i = 0; i < G.n_phases; ++i) {
            const x3::Phase& P = G.ph[i];
            if (P.type == x3::PH_ATTN_LN || P.type == x3::PH_CROSS) w += 256.0 * 256.0;
            else if (P.type == x3::PH_FFN_LN) w += 2.0 * 256.0 * 32.0 * P.n_steps;
            else w += (double)P.n_tasks * (256.0 * 32.0 * P.n_steps + 64.0 * 32.0 * P.n_steps + 64.0);
        }
        ProfScope prof("ranker_colsplit16_x3", 2.0 * (double)rows * w, (double)rows * (1024.0 + 12.0), st);
        const int n_pre = x3c::PREFETCH_WGS;           // (A/B against 0: profiles/r03_x3c_prefetch_ab.log)
        const int n_row_wgs = (int)((rows + x3c::ROWS_PER_WG - 1) / x3c::ROWS_PER_WG);
        hipLaunchKernelGGL(x3c::ranker_x3c_kernel, dim3((unsigned)(n_row_wgs + (n_pre > 0 ? n_pre : 0))), dim3(64 * x3c::WAVES),
                           x3c::LDS_BYTES, st, G, in, rows, n_row_wgs, x_out, ld_xout, logits, ld_logits);
        HIP_TRY(hipGetLastError());
        return AMDREC_OK;
    }
    // 16-row variant, small passes: 64-row workgroups of four waves (one per SIMD) - a pass that fits the chip once in that
    // shape (<= 256 CUs x 64 rows) finishes in ~0.7 of the 128-row shape's time (one request's 500 rows: 0.20 against
    // 0.28 ms, profiles/r03_x3b_waves.log); beyond it the 128-row shape's two waves per SIMD win
    const bool small = variant == 16 && rows <= X3B4_MAX_ROWS;
    const int rows_wg = small ? x3b4::ROWS_PER_WG : x3::ROWS_PER_WG;
    const unsigned grid = (unsigned)((rows + rows_wg - 1) / rows_wg);
    {
        // algorithmic FLOPs: 2 * rows * sum over the phases' weight elements (bench.py prices them against bf16 MFMA / 3)
        double w = 0;
        for (int i = 0; i < G.n_phases; ++i) {
            const x3::Phase& P = G.ph[i];
            if (P.type == x3::PH_ATTN_LN || P.type == x3::PH_CROSS) w += 256.0 * 256.0;
            else if (P.type == x3::PH_FFN_LN) w += 2.0 * 256.0 * 32.0 * P.n_steps;
            else w += (double)P.n_tasks * (256.0 * 32.0 * P.n_steps + 64.0 * 32.0 * P.n_steps + 64.0);
        }
        ProfScope prof(small ? "ranker_rowowner16_64_x3" : (variant == 16 ? "ranker_rowowner16_128_x3" : "ranker_rowowner_128_x3"),
                       2.0 * (double)rows * w, (double)rows * (1024.0 + 12.0), st);
        if (small)
            hipLaunchKernelGGL(x3b4::ranker_x3b_kernel, dim3(grid), dim3(64 * x3b4::WAVES), x3::RING_BYTES + x3::PARAM_FLOATS * 4, st,
                               G, in, rows, scratch, x_out, ld_xout, logits, ld_logits);
        else if (variant == 16)
            hipLaunchKernelGGL(x3b::ranker_x3b_kernel, dim3(grid), dim3(512), x3::RING_BYTES + x3::PARAM_FLOATS * 4, st, G, in,
                               rows, scratch, x_out, ld_xout, logits, ld_logits);
        else
            hipLaunchKernelGGL(x3::ranker_x3_kernel, dim3(grid), dim3(256), x3::RING_BYTES + x3::PARAM_FLOATS * 4, st, G, in,
                               rows, scratch, x_out, ld_xout, logits, ld_logits);
    }
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

// used by amdrec_ranker_forward (layers.hip)
namespace amdrec {
bool ranker_x3_wanted(const amdrec_ranker_params* p, long long rows) {
    const long long min_rows = p->x3.min_rows > 0 ? p->x3.min_rows : 8193;
    return rows >= min_rows && x3_eligible(p);
}
size_t ranker_x3_scratch_bytes(long long rows) { return x3_scratch_bytes(rows); }
int ranker_x3_run(const amdrec_ranker_params* p, const float* X, long long ldx, const float* U, const long long* rowmap,
                  long long row_base, int rowdiv, long long n_cache, long long rows, float* scratch, float* logits,
                  long long ld_logits, hipStream_t st) {
    // one request's pass (<= 4096 rows by default): the column-split kernel, 16 rows per workgroup
    const long long cs_rows = p->x3.cs_max_rows > 0 ? p->x3.cs_max_rows : (p->x3.cs_max_rows < 0 ? 0 : x3c::MAX_ROWS);
    const bool cs = x3c_available(p) && rows <= cs_rows;
    x3::Program G;
    int rc = x3_build(p, -1, G, cs);
    if (rc) return rc;
    x3::Input in{};
    if (X != nullptr) {
        in.X = X; in.ldx = ldx;
    } else {
        in.cache = p->ad_proj_cache; in.ldc = p->ld_ad_proj_cache; in.n_cache = n_cache; in.rowmap = rowmap; in.U = U;
        in.row_base = row_base; in.rowdiv = rowdiv;
    }
    return x3_launch(G, in, rows, scratch, nullptr, 0, logits, ld_logits, st, cs ? 160 : (int)p->x3.variant);
}
}  // namespace amdrec

extern "C" int amdrec_ranker_x3_prefix(const amdrec_ranker_params* p, const float* X, int64_t ldx, int64_t rows,
                                       int n_phases, float* x_out, int64_t ld_out, float* logits, int64_t ld_logits,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    REQUIRE(p != nullptr && X != nullptr, "null pointer");
    REQUIRE(x3_eligible(p), "these parameters are not eligible for the fp16x3 engine (or carry no x3 stream)");
    if (rows <= 0) return AMDREC_OK;
    REQUIRE(ldx >= 256 && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0, "bad X layout");
    REQUIRE(x_out == nullptr || (ld_out >= 256 && ld_out % 4 == 0 && ((uintptr_t)x_out % 16) == 0), "bad x_out layout");
    const long long cs_rows = p->x3.cs_max_rows > 0 ? p->x3.cs_max_rows : (p->x3.cs_max_rows < 0 ? 0 : x3c::MAX_ROWS);
    const bool cs = x3c_available(p) && rows <= cs_rows;
    x3::Program G;
    int rc = x3_build(p, n_phases, G, cs);
    if (rc) return rc;
    const bool heads = G.n_phases == 2 * p->n_layers + p->n_cross + 1;
    REQUIRE(!heads || (logits != nullptr && ld_logits >= rows), "the full chain needs a logits buffer");
    const size_t need = x3_scratch_bytes(rows);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    x3::Input in{};
    in.X = X; in.ldx = ldx;
    return x3_launch(G, in, rows, reinterpret_cast<float*>(workspace), x_out, ld_out, logits, ld_logits,
                     reinterpret_cast<hipStream_t>(stream), cs ? 160 : (int)p->x3.variant);
}
