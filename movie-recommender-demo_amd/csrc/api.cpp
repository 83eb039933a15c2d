// libamdrec: error plumbing and version entry points.
#include <stdarg.h>
#include <string.h>

#include "../../include/amdrec.h"
#include "common.hpp"

namespace amdrec {
thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace amdrec

// ---- profiling ------------------------------------------------------------------------
#include <mutex>
#include <string>
#include <vector>
namespace amdrec {
std::atomic<bool> g_prof_on{false};  // read without the lock on every entry (relaxed is enough: a scope that misses a toggle is untimed)
namespace {
struct Rec { int tag; hipEvent_t a, b; };
struct Tag { std::string name; long long launches; double ms, flops, bytes; };
std::vector<Rec> g_recs;
std::vector<Tag> g_tags;
std::vector<hipEvent_t> g_pool;
std::string g_only;              // non-empty: only tags with this prefix are timed
std::mutex g_mu;
unsigned long long g_gen = 0;    // bumped (under g_mu) whenever the records' events go back to the pool
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
ProfScope::ProfScope(const char* tag, double flops, double bytes, hipStream_t s) : end(nullptr), st(s), gen(0) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_only.empty() && strncmp(tag, g_only.c_str(), g_only.size()) != 0) return;
    int t = -1;
    for (size_t i = 0; i < g_tags.size(); ++i) if (g_tags[i].name == tag) { t = (int)i; break; }
    if (t < 0) { g_tags.push_back(Tag{tag, 0, 0, 0, 0}); t = (int)g_tags.size() - 1; }
    g_tags[t].launches++; g_tags[t].flops += flops; g_tags[t].bytes += bytes;
    Rec r{t, get_event(), get_event()};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    end = r.b;
    gen = g_gen;
}
ProfScope::~ProfScope() {
    if (end == nullptr) return;
    // Events are pooled, never destroyed, so `end` is always a valid handle - but a reset / report from another thread
    // between this scope's two ends has returned it to the pool, and get_event() may have handed it to a new scope:
    // re-recording it would corrupt that scope's timing.  The generation says whether it is still ours.
    std::lock_guard<std::mutex> lk(g_mu);
    if (gen != g_gen) return;
    (void)hipEventRecord(end, st);
}
}  // namespace amdrec

extern "C" int amdrec_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(amdrec::g_mu);
    for (auto& r : amdrec::g_recs) { amdrec::g_pool.push_back(r.a); amdrec::g_pool.push_back(r.b); }
    amdrec::g_recs.clear();
    amdrec::g_tags.clear();
    ++amdrec::g_gen;
    amdrec::g_prof_on = on != 0;
    return AMDREC_OK;
}

extern "C" int amdrec_profile_only(const char* tag_prefix) {
    std::lock_guard<std::mutex> lk(amdrec::g_mu);
    amdrec::g_only = tag_prefix ? tag_prefix : "";
    return AMDREC_OK;
}

extern "C" int amdrec_profile_report(amdrec_profile_entry* out, int max_entries, int* n) {
    REQUIRE(n != nullptr && (out != nullptr || max_entries == 0), "null pointer");
    std::lock_guard<std::mutex> lk(amdrec::g_mu);
    for (auto& r : amdrec::g_recs) {
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        amdrec::g_tags[r.tag].ms += ms;
        amdrec::g_pool.push_back(r.a);
        amdrec::g_pool.push_back(r.b);
    }
    amdrec::g_recs.clear();
    ++amdrec::g_gen;
    int cnt = 0;
    for (auto& t : amdrec::g_tags) {
        if (cnt >= max_entries) break;
        amdrec_profile_entry& e = out[cnt++];
        snprintf(e.name, sizeof(e.name), "%s", t.name.c_str());
        e.launches = t.launches; e.total_ms = t.ms; e.flops = t.flops; e.bytes = t.bytes;
    }
    *n = cnt;
    return AMDREC_OK;
}

extern "C" int amdrec_abi_version(void) { return AMDREC_ABI_VERSION; }
extern "C" const char* amdrec_last_error(void) { return amdrec::g_err; }
