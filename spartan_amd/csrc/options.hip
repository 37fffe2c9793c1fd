// spartan_amd: option table (options.hpp) — parsing, tiers, the process-wide defaults and the SPARTAN_OPTIONS hook.
#include "internal.hpp"

#include <mutex>
#include <string>

const SpOptDesc kOptDesc[OPT_COUNT] = {
#define X(id, key, def, lo, hi, tier, doc) {key, def, lo, hi, tier, doc},
    SP_OPTION_TABLE(X)
#undef X
};

static std::mutex g_opt_mu;
static SpOptions g_defaults;
static bool g_defaults_ready = false;

static int opt_find(const char* key) {
  for (int i = 0; i < OPT_COUNT; i++)
    if (strcmp(kOptDesc[i].key, key) == 0) return i;
  return -1;
}
static int32_t opt_apply(SpOptions& o, const char* key, const char* value, bool quiet) {
  if (!key || !value) return SP_EINVAL;
  int i = opt_find(key);
  if (i < 0) {
    if (!quiet) fprintf(stderr, "spartan_hip: unknown option \"%s\"\n", key);
    return SP_EINVAL;
  }
  char* end = nullptr;
  long long v = strtoll(value, &end, 10);
  if (end == value || *end != '\0' || v < kOptDesc[i].lo || v > kOptDesc[i].hi) {
    if (!quiet) fprintf(stderr, "spartan_hip: option %s = \"%s\" is outside %lld..%lld\n", key, value, kOptDesc[i].lo, kOptDesc[i].hi);
    return SP_EINVAL;
  }
  // values inside the range that no kernel supports: 1..5-bit LDS-form tables (a sub-table must be whole KB pieces of LDS-DMA: ADVICE r5 — 5
  // bits gave silently wrong commitments), queue-form workgroups that are not 1, 2 or 3 wavefronts per SIMD
  if ((i == OPT_MSM_LDS_BITS && v != 0 && v < 6) || (i == OPT_MSM_WINDOWS && v != 0 && v < 17) || (i == OPT_MSM_WBITS && v != 0 && v < 4) || ((i == OPT_MSM_Q_WAVES || i == OPT_MSM_Q_BG_WAVES) && v != 4 && v != 8 && v != 12)) {
    if (!quiet) fprintf(stderr, "spartan_hip: option %s = %lld is not a supported value\n", key, v);
    return SP_EINVAL;
  }
  if (kOptDesc[i].tier > 0 && !o.v[OPT_TESTING_UNLOCK]) {
    if (!quiet) fprintf(stderr, "spartan_hip: option %s is an A/B / test switch: set testing.unlock = 1 first\n", key);
    return SP_EINVAL;
  }
  o.v[i] = v;
  return SP_OK;
}
static void defaults_init_locked() {
  if (g_defaults_ready) return;
  for (int i = 0; i < OPT_COUNT; i++) g_defaults.v[i] = kOptDesc[i].def;
  if (const char* e = getenv("SPARTAN_OPTIONS")) {  // the library's one environment hook: "key=value,key=value" (options.hpp)
    std::string s(e);
    size_t p = 0;
    while (p < s.size()) {
      size_t q = s.find_first_of(",; ", p);
      if (q == std::string::npos) q = s.size();
      std::string kv = s.substr(p, q - p);
      p = q + 1;
      if (kv.empty()) continue;
      size_t eq = kv.find('=');
      if (eq == std::string::npos || opt_apply(g_defaults, kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str(), false) != SP_OK) {
        fprintf(stderr, "spartan_hip: SPARTAN_OPTIONS entry \"%s\" refused\n", kv.c_str());
        abort();  // a misspelt switch in an A/B script must not silently measure the default
      }
    }
  }
  g_defaults_ready = true;
}
SpOptions sp_default_options() {  // a copy, taken under the lock: sp_ctx_set_option(NULL, ...) may run on another thread
  std::lock_guard<std::mutex> lk(g_opt_mu);
  defaults_init_locked();
  return g_defaults;
}
void ctx_options_changed(sp_ctx* c, int which);  // core.hip: derived state (background workgroups, ...)

extern "C" {
int32_t sp_ctx_set_option(sp_ctx* c, const char* key, const char* value) {
  if (!c) {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    defaults_init_locked();
    return opt_apply(g_defaults, key, value, false);
  }
  {  // options that are process-wide by nature live in the process-wide table only: set on a context they would be accepted and ignored
    const int i = opt_find(key ? key : "");
    if (i == OPT_HOST_KECCAK || i == OPT_HOST_PROOF_GATE || i == OPT_HOST_CALLSTATS || i == OPT_HOST_PIN_THREAD) {
      fprintf(stderr, "spartan_hip: option %s is process-wide: set it with a NULL context (before the first proof)\n", key);
      return SP_EINVAL;
    }
  }
  int32_t rc = opt_apply(c->opt, key, value, false);
  if (rc == SP_OK) ctx_options_changed(c, opt_find(key));
  return rc;
}
int32_t sp_ctx_get_option(const sp_ctx* c, const char* key, int64_t* value) {
  if (!key || !value) return SP_EINVAL;
  int i = opt_find(key);
  if (i < 0) return SP_EINVAL;
  *value = c ? c->opt.v[i] : sp_default_options().v[i];
  return SP_OK;
}
int32_t sp_ctx_copy_options(sp_ctx* dst, const sp_ctx* src) {
  if (!dst || !src) return SP_EINVAL;
  dst->opt = src->opt;
  ctx_options_changed(dst, -1);
  return SP_OK;
}
int32_t sp_option_describe(int index, const char** key, int64_t* def, int64_t* lo, int64_t* hi, int* tier, const char** doc) {
  if (index < 0 || index >= OPT_COUNT) return SP_EINVAL;
  if (key) *key = kOptDesc[index].key;
  if (def) *def = kOptDesc[index].def;
  if (lo) *lo = kOptDesc[index].lo;
  if (hi) *hi = kOptDesc[index].hi;
  if (tier) *tier = kOptDesc[index].tier;
  if (doc) *doc = kOptDesc[index].doc;
  return SP_OK;
}
}
