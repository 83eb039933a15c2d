// Host-side driver of the C ABI for the AddressSanitizer / UBSan build:
// every entry point's argument validation, workspace arithmetic and error plumbing is exercised WITHOUT a GPU (each
// call below returns before its first HIP call, or is pure host arithmetic).  Built and run by
// tests/test_sanitizers.py against lib/libamdrec_san.so (host code compiled with -fsanitize=address,undefined).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/amdrec.h"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("FAIL %s:%d %s (last error: %s)\n", __FILE__, __LINE__, #cond, amdrec_last_error()); ++fails; } } while (0)

int main() {
    EXPECT(amdrec_abi_version() == AMDREC_ABI_VERSION);
    EXPECT(amdrec_last_error() != nullptr);
    alignas(256) static char fake[4096];                 // stands in for "a non-null, aligned pointer"; never dereferenced
    float* fp = reinterpret_cast<float*>(fake);
    int64_t* ip = reinterpret_cast<int64_t*>(fake);

    // ---- flat search: workspace arithmetic over the whole accepted range, validation of every argument ----
    size_t prev = 0;
    for (int64_t nq : {0ll, 1ll, 31ll, 512ll, 4096ll, 70000ll, (1ll << 24) - 1})
        for (int64_t n : {0ll, 1ll, 8192ll, 8193ll, 1000000ll, (1ll << 31) - 2048})
            for (int k : {1, 10, 500, 2048}) {
                size_t b = 0, bm = 0;
                EXPECT(amdrec_flat_search_workspace(nq, n, k, &b) == 0 && b % 256 == 0);
                EXPECT(amdrec_flat_search_mixed_workspace(nq, n, k, 256, &bm) == 0 && bm >= b);
                prev = b;
            }
    (void)prev;
    size_t b = 0;
    EXPECT(amdrec_flat_search_workspace(1, 1, 0, &b) < 0 && strstr(amdrec_last_error(), "k=0") != nullptr);
    EXPECT(amdrec_flat_search_workspace(1, 1, 2049, &b) < 0);
    EXPECT(amdrec_flat_search_workspace(-1, 1, 5, &b) < 0);
    EXPECT(amdrec_flat_search_workspace(1, 1, 5, nullptr) < 0);
    EXPECT(amdrec_flat_search_mixed_workspace(1, 1, 5, 12, &b) < 0);           // dim % 8
    EXPECT(amdrec_flat_search(fp, 10, 256, 255, fp, 1, 256, 5, 0, fp, ip, fake, 4096, nullptr, nullptr) < 0);   // dim % 4
    EXPECT(amdrec_flat_search(fp, 10, 256, 256, fp, 1, 128, 5, 0, fp, ip, fake, 4096, nullptr, nullptr) < 0);   // ld_queries < dim
    EXPECT(amdrec_flat_search(fp, 1ll << 31, 256, 256, fp, 1, 256, 5, 0, fp, ip, fake, 4096, nullptr, nullptr) < 0);
    EXPECT(amdrec_flat_search(fp, 10, 256, 256, nullptr, 1, 256, 5, 0, fp, ip, fake, 4096, nullptr, nullptr) < 0);
    EXPECT(amdrec_flat_search(fp, 10, 256, 256, fp, 1, 256, 5, 0, fp, ip, fake, 16, nullptr, nullptr) == -3);    // workspace too small
    EXPECT(amdrec_flat_search(fp, 10, 256, 256, fp, 0, 256, 5, 0, fp, ip, nullptr, 0, nullptr, nullptr) == 0);  // nq == 0: no-op
    EXPECT(amdrec_flat_search_mixed(fp, 10, 256, 256, nullptr, 256, fp, fp, 1, 256, 5, 0, fp, ip, fake, 4096, nullptr, nullptr) < 0);
    EXPECT(amdrec_bf16_rows(fp, 5, 256, 6, reinterpret_cast<uint16_t*>(fake), 256, nullptr, nullptr) < 0);
    EXPECT(amdrec_bf16_rows(fp, 0, 256, 256, reinterpret_cast<uint16_t*>(fake), 256, nullptr, nullptr) == 0);
    EXPECT(amdrec_topk_merge(fp, reinterpret_cast<int32_t*>(fake), 40, 4096, 0, 1, 500, fp, ip, nullptr) < 0);  // n_lists*k > 16384
    EXPECT(amdrec_topk_merge(fp, reinterpret_cast<int32_t*>(fake), 8, 4095, 0, 1, 500, fp, ip, nullptr) < 0);   // stride % 4
    EXPECT(amdrec_topk_merge(fp, reinterpret_cast<int32_t*>(fake), 8, 4096, 0, 0, 500, fp, ip, nullptr) == 0);
    EXPECT(amdrec_ivf_scan(fp, 256, 255, ip, ip, fp, 1, 256, ip, ip, 4, reinterpret_cast<uint64_t*>(fake), 16, 0, nullptr) < 0);
    EXPECT(amdrec_ivf_scan(fp, 256, 256, ip, ip, fp, 70000, 256, ip, ip, 4, reinterpret_cast<uint64_t*>(fake), 16, 0, nullptr) < 0);
    EXPECT(amdrec_ivf_select(reinterpret_cast<uint64_t*>(fake), 16, ip, 1, 0, fp, ip, nullptr) < 0);
    // bf16-prefilter forms of the grouped scan (ABI v10): dim must be a multiple of 8, nothing to do without queries / tiles
    EXPECT(amdrec_ivf_filter_bounds(fp, 4, 256, 12, reinterpret_cast<uint16_t*>(fake), 256, fp, fp, 1, fp, nullptr) < 0);
    EXPECT(amdrec_ivf_filter_bounds(fp, 0, 256, 256, reinterpret_cast<uint16_t*>(fake), 256, fp, fp, 1, fp, nullptr) == 0);
    EXPECT(amdrec_ivf_filter_bounds(fp, 4, 256, 256, nullptr, 256, fp, fp, 1, fp, nullptr) < 0);
    EXPECT(amdrec_ivf_scan_grouped_mixed(fp, 256, reinterpret_cast<uint16_t*>(fake), 256, 256, ip, ip, 16, 100, fp, 256,
                                         reinterpret_cast<uint16_t*>(fake), 256, ip, ip, 0, 64, ip, reinterpret_cast<uint64_t*>(fake),
                                         16, 0, fp, 1, fp, ip, nullptr) == 0);                                     // no tiles
    EXPECT(amdrec_ivf_scan_grouped_mixed(fp, 256, reinterpret_cast<uint16_t*>(fake), 256, 256, ip, ip, 16, 100, fp, 256,
                                         reinterpret_cast<uint16_t*>(fake), 256, ip, ip, 8, 48, ip, reinterpret_cast<uint64_t*>(fake),
                                         16, 0, fp, 1, fp, ip, nullptr) < 0);                                      // qtile 48
    EXPECT(amdrec_ivf_scan_grouped_mixed(fp, 256, nullptr, 256, 256, ip, ip, 16, 100, fp, 256,
                                         reinterpret_cast<uint16_t*>(fake), 256, ip, ip, 8, 64, ip, reinterpret_cast<uint64_t*>(fake),
                                         16, 0, fp, 1, fp, ip, nullptr) < 0);                                      // null shadow
    EXPECT(amdrec_l2_normalize(fp, 256, fp, 256, 0, 256, nullptr) == 0);
    EXPECT(amdrec_remap_ids(ip, ip, 10, ip, 0, nullptr) == 0);
    EXPECT(amdrec_prep_numerical(fp, fp, fp, fp, 0, 13, nullptr) == 0);
    EXPECT(amdrec_select_topk(fp, 500, 3, 5, ip, 1, 500, 10, ip, fp, nullptr, nullptr) < 0);                    // rank_task out of range

    // ---- towers ----
    amdrec_tower_params tp;
    memset(&tp, 0, sizeof(tp));
    EXPECT(amdrec_tower_workspace(&tp, 10, &b) < 0);                             // n_feat == 0
    tp.n_feat = 6; tp.emb_dim = 16; tp.n_num = 13; tp.n_layers = 3;
    tp.dims[0] = 109; tp.dims[1] = 512; tp.dims[2] = 256; tp.dims[3] = 256;
    tp.ldw[0] = 128; tp.ldw[1] = 512; tp.ldw[2] = 256;
    tp.tables = fp; tp.table_off = reinterpret_cast<int32_t*>(fake); tp.cards = reinterpret_cast<int32_t*>(fake);
    for (int l = 0; l < 3; ++l) { tp.w[l] = fp; tp.b[l] = fp; }
    for (int64_t rows : {0ll, 1ll, 511ll, 262144ll, 300001ll, 10000000ll}) {
        EXPECT(amdrec_tower_workspace(&tp, rows, &b) == 0 && b % 256 == 0);
        EXPECT(amdrec_tower_forward(&tp, ip, fp, rows, fp, 256, nullptr, fake, 0, nullptr) == (rows ? -3 : 0));   // workspace too small
    }
    tp.emb_dim = 12;
    EXPECT(amdrec_tower_workspace(&tp, 10, &b) < 0);
    tp.emb_dim = 16; tp.dims[0] = 110;
    EXPECT(amdrec_tower_workspace(&tp, 10, &b) < 0);
    tp.dims[0] = 109; tp.ldw[0] = 100;
    EXPECT(amdrec_tower_workspace(&tp, 10, &b) < 0);
    tp.ldw[0] = 128; tp.dims[3] = 512;
    EXPECT(amdrec_tower_workspace(&tp, 10, &b) < 0);                             // output_dim > 256

    // ---- ranker (reference architecture) ----
    amdrec_ranker_params rp;
    memset(&rp, 0, sizeof(rp));
    EXPECT(amdrec_ranker_workspace(&rp, 10, &b) < 0);
    rp.n_user_feat = 6; rp.n_ad_feat = 20; rp.emb_dim = 32; rp.n_num = 13; rp.d_model = 256; rp.d_ff = 1024;
    rp.n_layers = 3; rp.n_cross = 3; rp.n_tasks = 3; rp.head_h1 = 256; rp.head_h2 = 64; rp.ln_eps = 1e-5f;
    rp.tables = fp; rp.table_off = reinterpret_cast<int32_t*>(fake); rp.cards = reinterpret_cast<int32_t*>(fake);
    rp.w_proj = fp; rp.b_proj = fp;
    size_t last = 0;
    for (int64_t rows : {0ll, 1ll, 127ll, 128ll, 8192ll, 262144ll, 262145ll, 300001ll, 2560000ll}) {
        EXPECT(amdrec_ranker_workspace(&rp, rows, &b) == 0 && b % 256 == 0 && b >= last);
        last = rows <= 262144 ? b : last;
        if (rows) EXPECT(amdrec_ranker_forward(&rp, ip, fp, 1, ip, nullptr, rows, fp, rows, nullptr, rows, rows, fake, 64, nullptr) == -3);
    }
    EXPECT(amdrec_ranker_forward(&rp, ip, fp, 0, ip, nullptr, 5, fp, 5, nullptr, 5, 5, fake, 4096, nullptr) < 0);   // user_rowdiv 0
    EXPECT(amdrec_ranker_forward(&rp, ip, fp, 1, ip, nullptr, 5, fp, 4, nullptr, 5, 5, fake, 4096, nullptr) < 0);   // ld_logits < rows
    EXPECT(amdrec_ranker_project_ads(&rp, ip, 10, fp, 256, fake, 4096, nullptr) < 0);                              // no split projection
    // fp16x3 engine: eligibility and stream-length checks are host logic
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0);               // no stream
    rp.x3.stream = fake; rp.x3.chunks = 5; rp.x3.variant = 32;
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0);               // no parameter blob
    rp.x3.params = fp; rp.x3.n_params = 7;
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0 &&
           strstr(amdrec_last_error(), "parameter blob") != nullptr);                                            // wrong blob size
    rp.x3.n_params = 10240;                                       // 3 * (1536 + 1024) + 3 * 256 + 3 * (256 + 132) = 9612 -> padded
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0 &&
           strstr(amdrec_last_error(), "chunks") != nullptr);                                                    // wrong stream length
    rp.x3.chunks = 3 * (16 + 128) + 48 + 60;
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 250, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0);               // ldx < 256
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) == -3);             // workspace too small
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 0, -1, fp, 256, fp, 10, fake, 4096, nullptr) == 0);               // rows == 0
    rp.d_model = 128;
    EXPECT(amdrec_ranker_x3_prefix(&rp, fp, 256, 10, -1, fp, 256, fp, 10, fake, 4096, nullptr) < 0);               // not eligible
    rp.d_model = 300;
    EXPECT(amdrec_ranker_workspace(&rp, 10, &b) < 0);                                                            // d_model > 256

    // ---- profiling plumbing without any launch ----
    amdrec_profile_entry pe[4];
    int n = -1;
    EXPECT(amdrec_profile_enable(1) == 0 && amdrec_profile_report(pe, 4, &n) == 0 && n == 0);
    EXPECT(amdrec_profile_report(nullptr, 0, &n) == 0 && amdrec_profile_report(pe, 4, nullptr) < 0);
    EXPECT(amdrec_profile_enable(0) == 0);
    printf(fails ? "abi_san_driver: %d FAILED\n" : "abi_san_driver: all host-side checks passed\n", fails);
    return fails ? 1 : 0;
}
