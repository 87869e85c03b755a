// Library-level entry points: version, the thread-local error string and the process-wide option table.
#include "common.h"
#include "options.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace kccot {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- options (include/kccot.h: kccot_set_option) ----------------------------------------------------------------------
// One atomic int per option: a call reads the ones it needs with relaxed loads (no environment lookups on any call path,
// safe against a concurrent kccot_set_option from another thread: the call sees the old or the new value).
struct OptDesc { const char* name; int def, lo, hi; };
static const OptDesc g_desc[OPT_COUNT] = {
    {"gram_f32", 0, 0, 1},
    {"apply_f32", 0, 0, 1},
    {"cost_tiled", 1, 0, 1},
    {"cost_tile256", 1, 0, 1},
    {"cost_blocked", 1, 0, 1},
    {"apply_m256", 1, 0, 1},
    {"apply_one_launch", 1, 0, 1},
    {"apply_q256", 1, 0, 1},
    {"sinkhorn_shortcut", 1, 0, 1},
    {"sinkhorn_fused", 1, 0, 1},
    {"sinkhorn_fused_max_n", 64, 1, 128},
    {"sinkhorn_lanes_per_line", 0, 0, 16},
    {"sinkhorn_coop", 1, 0, 1},
    {"sinkhorn_coop_xcd", 1, 0, 2},
    {"sinkhorn_coop_max_wg", 0, 0, 1 << 20},
    {"smooth_stream", 1, 0, 1},
    {"smooth_generic", 0, 0, 1},
    {"smooth_fused_tw", 1, 0, 1},
    {"smooth_bwd_fold", 1, 0, 2},
    {"smooth_fused3", 1, 0, 2},
};
static std::atomic<int> g_val[OPT_COUNT];
static std::once_flag g_once;

static int find_option(const char* name, size_t len) {
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strlen(g_desc[i].name) == len && strncmp(g_desc[i].name, name, len) == 0) return i;
    return -1;
}

static void init_options() {
    for (int i = 0; i < OPT_COUNT; ++i) g_val[i].store(g_desc[i].def, std::memory_order_relaxed);
    // KCCOT_OPTIONS="name=value,name=value": initial values for command-line tools (bench.py, tools/), read ONCE at
    // the first use of the library.  Unknown names and out-of-range values are ignored here; kccot_set_option rejects them.
    const char* e = getenv("KCCOT_OPTIONS");
    while (e && *e) {
        const char* end = strchr(e, ',');
        const size_t len = end ? (size_t)(end - e) : strlen(e);
        const char* eq = (const char*)memchr(e, '=', len);
        if (eq) {
            const int i = find_option(e, (size_t)(eq - e));
            const int v = atoi(eq + 1);
            if (i >= 0 && v >= g_desc[i].lo && v <= g_desc[i].hi) g_val[i].store(v, std::memory_order_relaxed);
        }
        e = end ? end + 1 : nullptr;
    }
}

int opt(Option o) {
    std::call_once(g_once, init_options);
    return g_val[o].load(std::memory_order_relaxed);
}
}  // namespace kccot

using namespace kccot;

extern "C" int kccot_version(void) { return KCCOT_VERSION; }
extern "C" const char* kccot_last_error(void) { return kccot::g_err; }

extern "C" int kccot_set_option(const char* name, int value) {
    if (!name) return fail(KCCOT_EINVAL, "set_option: null name");
    std::call_once(g_once, init_options);
    const int i = find_option(name, strlen(name));
    if (i < 0) return fail(KCCOT_EINVAL, "set_option: unknown option '%s'", name);
    if (value < g_desc[i].lo || value > g_desc[i].hi)
        return fail(KCCOT_EINVAL, "set_option: %s = %d is outside [%d, %d]", name, value, g_desc[i].lo, g_desc[i].hi);
    g_val[i].store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int kccot_get_option(const char* name, int* value) {
    if (!name || !value) return fail(KCCOT_EINVAL, "get_option: null pointer");
    std::call_once(g_once, init_options);
    const int i = find_option(name, strlen(name));
    if (i < 0) return fail(KCCOT_EINVAL, "get_option: unknown option '%s'", name);
    *value = g_val[i].load(std::memory_order_relaxed);
    return 0;
}

extern "C" int kccot_option_count(void) { return OPT_COUNT; }
extern "C" const char* kccot_option_name(int index) { return (index >= 0 && index < OPT_COUNT) ? g_desc[index].name : nullptr; }
