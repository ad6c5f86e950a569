// tuning.cpp -- the table behind tuning.hpp: key names <-> fields, and the one-time read of ACVM_TUNING.
#include "tuning.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace acvm {
namespace {

struct Key { const char *name; int64_t Tuning::*field; };
const Key KEYS[] = {
    {"scale", &Tuning::scale}, {"relax", &Tuning::relax}, {"pedersen_bundle", &Tuning::pedersen_bundle}, {"pedersen_bundle_waves", &Tuning::pedersen_bundle_waves}, {"pairs", &Tuning::pairs}, {"chains", &Tuning::chains}, {"max_tails", &Tuning::max_tails},
    {"inv_epoch", &Tuning::inv_epoch}, {"inv_latency", &Tuning::inv_latency}, {"inv_chunk", &Tuning::inv_chunk}, {"byte_plane", &Tuning::byte_plane}, {"heavy_epoch", &Tuning::heavy_epoch}, {"heavy_latency", &Tuning::heavy_latency},
    {"pedersen_latency", &Tuning::pedersen_latency}, {"pedersen_epoch", &Tuning::pedersen_epoch}, {"digest_epoch", &Tuning::digest_epoch}, {"range_fuse", &Tuning::range_fuse},
    {"range_merge", &Tuning::range_merge}, {"hash_chain", &Tuning::hash_chain}, {"brillig_inline", &Tuning::brillig_inline}, {"sl_lane", &Tuning::sl_lane},
    {"pedersen_waves", &Tuning::pedersen_waves}, {"pedersen_prio", &Tuning::pedersen_prio}, {"light_fuse", &Tuning::light_fuse}, {"plan_validate", &Tuning::plan_validate}, {"brillig_mem_cells", &Tuning::brillig_mem_cells}, {"overlap", &Tuning::overlap}, {"heavy_streams", &Tuning::heavy_streams}, {"heavy_only_streams", &Tuning::heavy_only_streams},
    {"fc_relevel", &Tuning::fc_relevel}, {"exact_async", &Tuning::exact_async}, {"brillig_steps_log2", &Tuning::brillig_steps_log2},
    {"brillig_steps_max_log2", &Tuning::brillig_steps_max_log2}, {"brillig_call_depth", &Tuning::brillig_call_depth},
    {"brillig_call_depth_max", &Tuning::brillig_call_depth_max}, {"brillig_mem_max_log2", &Tuning::brillig_mem_max_log2},
    {"pedersen_window_bits", &Tuning::pedersen_window_bits}, {"win16", &Tuning::win16}, {"tables_keep", &Tuning::tables_keep},
};
constexpr unsigned N_KEYS = sizeof KEYS / sizeof KEYS[0];

Tuning g_tuning;
std::once_flag g_env_once;

void read_env() {
    const char *e = getenv("ACVM_TUNING");
    if (!e) return;
    std::string s(e);
    size_t at = 0;
    while (at < s.size()) {
        size_t end = s.find(',', at);
        if (end == std::string::npos) end = s.size();
        const std::string item = s.substr(at, end - at);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            const std::string k = item.substr(0, eq);
            for (const Key &key : KEYS)
                if (k == key.name) g_tuning.*key.field = strtoll(item.c_str() + eq + 1, nullptr, 10);
        }
        at = end + 1;
    }
}

}  // namespace

Tuning &tuning() {
    std::call_once(g_env_once, read_env);
    return g_tuning;
}
bool tuning_set(const char *key, int64_t value) {
    if (!key) return false;
    Tuning &t = tuning();
    // the window table exists for ONE window width (grumpkin_host.hpp GRUMPKIN_PEDW_BITS): any other nonzero value would silently get that one
    if (!strcmp(key, "pedersen_window_bits") && value != 0 && value != 24) return false;
    for (const Key &k : KEYS)
        if (!strcmp(key, k.name)) { t.*k.field = value; return true; }
    return false;
}
bool tuning_get(const char *key, int64_t *value) {
    if (!key || !value) return false;
    const Tuning &t = tuning();
    for (const Key &k : KEYS)
        if (!strcmp(key, k.name)) { *value = t.*k.field; return true; }
    return false;
}
const char *tuning_key(unsigned index) { return index < N_KEYS ? KEYS[index].name : nullptr; }

}  // namespace acvm
