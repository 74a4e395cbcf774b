// config.hip -- the table behind M1_CFG (common.h) and m1_config_set / m1_config_get / m1_config_unset (include/m1hip.h).
#include "common.h"
#include <mutex>
#include <string.h>

#define M1_CFG_MAX 160
static M1CfgEntry g_cfg[M1_CFG_MAX];
static int g_ncfg = 0;
static std::mutex g_cfg_mu;

static void cfg_refresh(M1CfgEntry* e) { e->v = e->has_ovr ? e->ovr : (e->env_set ? e->env : e->def); }
static M1CfgEntry* cfg_find_or_add(const char* name) {      // (caller holds the lock)
    for (int i = 0; i < g_ncfg; ++i) if (!strcmp(g_cfg[i].name, name)) return &g_cfg[i];
    if (g_ncfg >= M1_CFG_MAX || strlen(name) >= sizeof(g_cfg[0].name)) return nullptr;
    M1CfgEntry* e = &g_cfg[g_ncfg++];
    memset(e, 0, sizeof(*e));
    strcpy(e->name, name);
    const char* ev = getenv(name);
    if (ev) { e->env = atoi(ev); e->env_set = 1; }
    cfg_refresh(e);
    return e;
}
M1CfgEntry* m1_cfg_entry(const char* name, int def) {
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    static M1CfgEntry overflow;                               // (table full: the default, not settable)
    M1CfgEntry* e = cfg_find_or_add(name);
    if (!e) { overflow.def = overflow.v = def; return &overflow; }
    if (!e->def_known) { e->def = def; e->def_known = 1; cfg_refresh(e); }
    return e;
}
extern "C" int m1_config_set(const char* name, int value) {
    if (!name || strncmp(name, "M1_", 3)) return M1_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    M1CfgEntry* e = cfg_find_or_add(name);
    if (!e) return M1_ERR_BAD_ARG;
    e->ovr = value; e->has_ovr = 1; cfg_refresh(e);
    return M1_OK;
}
extern "C" int m1_config_unset(const char* name) {
    if (!name) return M1_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    for (int i = 0; i < g_ncfg; ++i) if (!strcmp(g_cfg[i].name, name)) { g_cfg[i].has_ovr = 0; cfg_refresh(&g_cfg[i]); return M1_OK; }
    return M1_OK;
}
// *value = the switch's current value; M1_ERR_UNSUPPORTED when no launch has consulted it yet and nothing was set
extern "C" int m1_config_get(const char* name, int* value) {
    if (!name || !value) return M1_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_cfg_mu);
    for (int i = 0; i < g_ncfg; ++i)
        if (!strcmp(g_cfg[i].name, name)) {
            if (!g_cfg[i].def_known && !g_cfg[i].has_ovr && !g_cfg[i].env_set) return M1_ERR_UNSUPPORTED;
            *value = g_cfg[i].v; return M1_OK;
        }
    return M1_ERR_UNSUPPORTED;
}
