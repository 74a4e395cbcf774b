// prof.hip -- opt-in per-kernel-family timing with hipEvents recorded on the launch stream.
// Off by default (zero overhead: one predictable branch per entry point).  Not for use under stream capture.
#include "common.h"
#include <string.h>
#include <vector>
#include <mutex>
#include <stdarg.h>
#include <string>

namespace {
struct Rec { char name[48]; double flops, bytes; long long launches; };
struct Pair { hipEvent_t a, b; int slot; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<Pair> g_pairs;
size_t g_used = 0;
std::mutex g_mu;
}

M1ProfScope::M1ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : slot(-1), s(stream) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    int idx = -1;
    for (size_t i = 0; i < g_recs.size(); ++i) if (!strcmp(g_recs[i].name, name)) { idx = (int)i; break; }
    if (idx < 0) { Rec r{}; strncpy(r.name, name, 47); g_recs.push_back(r); idx = (int)g_recs.size() - 1; }
    g_recs[idx].flops += flops; g_recs[idx].bytes += bytes; g_recs[idx].launches += 1;
    if (g_used == g_pairs.size()) {
        Pair p{}; 
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
        g_pairs.push_back(p);
    }
    slot = (int)g_used++;
    g_pairs[slot].slot = idx;
    (void)hipEventRecord(g_pairs[slot].a, s);
}
M1ProfScope::~M1ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pairs[slot].b, s);
}

extern "C" int m1_prof_enable(int on) { std::lock_guard<std::mutex> lk(g_mu); g_on = on != 0; return M1_OK; }
extern "C" int m1_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.clear(); g_used = 0;
    return M1_OK;
}
extern "C" int m1_prof_read(m1_prof_rec_t* out, int max_n) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<double> ms(g_recs.size(), 0.0);
    for (size_t i = 0; i < g_used; ++i) {
        float t = 0.f;
        if (hipEventSynchronize(g_pairs[i].b) != hipSuccess) continue;
        if (hipEventElapsedTime(&t, g_pairs[i].a, g_pairs[i].b) == hipSuccess) ms[g_pairs[i].slot] += t;
    }
    int n = 0;
    for (size_t i = 0; i < g_recs.size() && n < max_n; ++i, ++n) {
        memset(&out[n], 0, sizeof(out[n]));
        strncpy(out[n].name, g_recs[i].name, 47);
        out[n].total_ms = ms[i]; out[n].flops = g_recs[i].flops; out[n].bytes = g_recs[i].bytes; out[n].launches = g_recs[i].launches;
    }
    return n;
}

// ---- kernel-choice log: off until m1_debug_kernels(1) is called; names are appended, comma-separated, up to 1 MB
namespace { bool g_klog_on = false; std::string g_klog; std::mutex g_klog_mu; }
void m1_note_kernel(const char* fmt, ...) {
    if (!g_klog_on) return;
    char buf[160];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> lk(g_klog_mu);
    if (g_klog.size() < (1u << 20)) { if (!g_klog.empty()) g_klog += ','; g_klog += buf; }
}
extern "C" const char* m1_debug_kernels(int mode) {
    static std::string out;
    std::lock_guard<std::mutex> lk(g_klog_mu);
    out = g_klog;
    if (mode >= 0) { g_klog.clear(); g_klog_on = mode != 0; }
    return out.c_str();
}

// ---- debug: LDS canary.  Blocks fill 31 KB of static LDS (the footprint of thin_fwd_kernel) with a pattern and re-read it `spins`
// times; *bad counts words that changed.  Run next to another kernel inside a graph (tools/dbg/stress_lds.py) it shows whether that
// kernel writes LDS outside its own allocation (LDS-DMA is not bounds-checked against the workgroup's allocation).
__global__ void __launch_bounds__(256) m1_lds_canary_kernel(unsigned* __restrict__ bad, int spins) {
    __shared__ unsigned buf[7936];
    for (int i = threadIdx.x; i < 7936; i += 256) buf[i] = (unsigned)i * 2654435761u ^ blockIdx.x;
    __syncthreads();
    unsigned nb = 0;
    for (int s = 0; s < spins; ++s) {
        for (int i = threadIdx.x; i < 7936; i += 256) nb += ((volatile unsigned*)buf)[i] != ((unsigned)i * 2654435761u ^ blockIdx.x);
        __builtin_amdgcn_s_sleep(8);
    }
    if (nb) atomicAdd(bad, nb);
}
extern "C" int m1_debug_lds_canary(unsigned* bad, int blocks, int spins, void* stream) {
    hipLaunchKernelGGL(m1_lds_canary_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bad, spins);
    return m1_check_launch();
}

// M1_DEBUG_SKIP=name[,name...] (measurement aid, results are garbage): the named entry points return M1_OK without launching anything --
// what a family of kernels costs the REPLAYED step, next to what its kernels cost alone (tools/dbg/skip_families.sh)
#include <string.h>
#include <stdlib.h>
bool m1_debug_skip(const char* name) {
    static const char* env = getenv("M1_DEBUG_SKIP");
    if (!env || !*env) return false;
    const size_t n = strlen(name);
    for (const char* p = env; *p;) {
        const char* e = strchr(p, ','); const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == n && !strncmp(p, name, n)) return true;
        p += len; if (*p == ',') ++p;
    }
    return false;
}
