// Library-wide entry points: version, thread-local error string, tuning options.
#include <string.h>

#include "common.h"

namespace sad {

static thread_local char g_err[512] = "";
// Tuning knobs (A/B switches for measurements; every default is 0).  Relaxed atomics: a knob may be
// flipped by one thread while another launches — each launch reads a consistent int, never a torn
// one; the per-call `geometry` field of sad_mlp_args is the race-free way to steer one call.
static std::atomic<int> g_opt[OPT_COUNT] = {};

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int get_option(int which) { return (which >= 0 && which < OPT_COUNT) ? g_opt[which].load(std::memory_order_relaxed) : 0; }

}  // namespace sad

SAD_API int sad_version(void) { return SAD_ABI_VERSION; }

SAD_API const char *sad_last_error(void) { return sad::g_err; }

SAD_API int sad_set_option(const char *key, int value) {
    SAD_REQUIRE(key, "sad_set_option: NULL key");
    if (!strcmp(key, "fps_dpp")) { sad::g_opt[sad::OPT_FPS_DPP].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_rw")) { sad::g_opt[sad::OPT_MLP_RW].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_budget_kb")) { sad::g_opt[sad::OPT_MLP_BUDGET_KB].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_force")) { sad::g_opt[sad::OPT_MLP_FORCE].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_dedup_f")) { sad::g_opt[sad::OPT_MLP_DEDUP_F].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_nodedup")) { sad::g_opt[sad::OPT_MLP_NODEDUP].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_static")) { sad::g_opt[sad::OPT_MLP_STATIC].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_noxcd")) { sad::g_opt[sad::OPT_MLP_NOXCD].store(value, std::memory_order_relaxed); return SAD_OK; }   // 1 = plain chunk order
    if (!strcmp(key, "mlp_dyn_slots")) { sad::g_opt[sad::OPT_MLP_DYN_SLOTS].store(value, std::memory_order_relaxed); return SAD_OK; }   // workgroups per CU of the global-packing grid
    // test knobs of the cooperative chain kernel's item queues (csrc/mlp_coop.hip, include/sad_amd.h)
    if (!strcmp(key, "mlp_steal_after")) { sad::g_opt[sad::OPT_MLP_STEAL_AFTER].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_layer_queue")) { sad::g_opt[sad::OPT_MLP_LAYER_QUEUE].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_rows_form")) { sad::g_opt[sad::OPT_MLP_ROWS_FORM].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "mlp_check_inuse")) { sad::g_opt[sad::OPT_MLP_CHECK_INUSE].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "group_variant")) { sad::g_opt[sad::OPT_GROUP_VARIANT].store(value, std::memory_order_relaxed); return SAD_OK; }   // 1 = L2-gather kernel only
    if (!strcmp(key, "bq_variant")) { sad::g_opt[sad::OPT_BQ_VARIANT].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "fps_threads")) { sad::g_opt[sad::OPT_FPS_THREADS].store(value, std::memory_order_relaxed); return SAD_OK; }
    if (!strcmp(key, "fps_variant")) { sad::g_opt[sad::OPT_FPS_VARIANT].store(value, std::memory_order_relaxed); return SAD_OK; }
    return sad::fail(SAD_EINVAL, "sad_set_option: unknown key '%s'", key);
}
