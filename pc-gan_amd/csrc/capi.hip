// Status / diagnostics part of the C-ABI (include/pcgan_hip.h).
#include "common.h"
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace pcgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pcgan

// ---- kernel timer ---------------------------------------------------------------------------------------------------------------
namespace pcgan {
namespace {
struct TimerKind {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
};
std::mutex g_timer_mu;
std::atomic<int> g_timer_on{0};
TimerKind g_timer[TIMER_KINDS];
}  // namespace

TimerScope::TimerScope(int kind_, hipStream_t st_) : kind(kind_), slot(-1), st(st_) {
    if (kind < 0 || kind >= TIMER_KINDS || !g_timer_on.load(std::memory_order_relaxed)) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;     // never inside a graph capture
    std::lock_guard<std::mutex> lk(g_timer_mu);
    TimerKind& t = g_timer[kind];
    if (t.used >= (int)t.start.size()) return;
    slot = t.used++;
    (void)hipEventRecord(t.start[slot], st);
}
TimerScope::~TimerScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    (void)hipEventRecord(g_timer[kind].stop[slot], st);
}
}  // namespace pcgan

extern "C" int pcgan_timer_enable(int capacity) {
    using namespace pcgan;
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_on.store(0);
    for (int k = 0; k < TIMER_KINDS; ++k) {
        TimerKind& t = g_timer[k];
        for (hipEvent_t e : t.start) (void)hipEventDestroy(e);
        for (hipEvent_t e : t.stop) (void)hipEventDestroy(e);
        t.start.clear();
        t.stop.clear();
        t.used = 0;
        for (int i = 0; i < capacity; ++i) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                set_error("timer_enable: hipEventCreate failed");
                return 1;
            }
            t.start.push_back(a);
            t.stop.push_back(b);
        }
    }
    g_timer_on.store(capacity > 0 ? 1 : 0);
    return 0;
}

extern "C" int pcgan_timer_read(int kind, float* ms, int cap) {
    using namespace pcgan;
    if (kind < 0 || kind >= TIMER_KINDS || (!ms && cap > 0)) {
        set_error("timer_read: bad kind / null output");
        return -1;
    }
    std::lock_guard<std::mutex> lk(g_timer_mu);
    TimerKind& t = g_timer[kind];
    int n = 0;
    for (int i = 0; i < t.used && n < cap; ++i) {
        if (hipEventSynchronize(t.stop[i]) != hipSuccess) continue;
        float v = 0.f;
        if (hipEventElapsedTime(&v, t.start[i], t.stop[i]) == hipSuccess) ms[n++] = v;
    }
    t.used = 0;
    return n;
}

namespace pcgan {
static std::atomic<unsigned*> g_nonfinite{nullptr};
unsigned* nonfinite_counter() { return g_nonfinite.load(std::memory_order_relaxed); }
}  // namespace pcgan
extern "C" int pcgan_set_nonfinite_counter(unsigned int* dev_word) {
    pcgan::g_nonfinite.store(dev_word);
    return 0;
}

namespace pcgan {
static const char* const g_opt_names[OPT_COUNT] = {"bsplit_halo", "wgrad_gen", "wgrad_padcopy", "wgrad_cw", "hgemm_bf16", "wgd_look", "wgrad_direct", "hgemm_tile", "hgemm_ks", "wgrad_rowring"};
static std::atomic<int> g_opt[OPT_COUNT] = {{1}, {1}, {0}, {0}, {1}, {3}, {0}, {0}, {0}, {1}};
int option(int id) { return (id >= 0 && id < OPT_COUNT) ? g_opt[id].load(std::memory_order_relaxed) : 0; }
static int option_index(const char* key) {
    for (int i = 0; key && i < OPT_COUNT; ++i)
        if (strcmp(key, g_opt_names[i]) == 0) return i;
    return -1;
}
}  // namespace pcgan
extern "C" int pcgan_set_option(const char* key, int value) {
    using namespace pcgan;
    const int i = option_index(key);
    PCGAN_CHECK(i >= 0, "set_option: unknown option '%s' (bsplit_halo, wgrad_gen, wgrad_padcopy, wgrad_cw, hgemm_bf16, wgd_look, wgrad_direct, hgemm_tile, hgemm_ks, wgrad_rowring)", key ? key : "(null)");
    PCGAN_CHECK(i != OPT_WGRAD_CW || value == 0 || value == 128 || value == 256, "set_option: wgrad_cw takes 0 (default), 128 or 256, got %d", value);
    PCGAN_CHECK(i != OPT_HGEMM_TILE || value == 0 || value == 128128 || value == 128064 || value == 64128 || value == 64064,
                "set_option: hgemm_tile takes 0 (heuristic) or BM * 1000 + BP with BM, BP in {64, 128}, got %d", value);
    PCGAN_CHECK(i != OPT_HGEMM_KS || (value >= 0 && value <= 8), "set_option: hgemm_ks takes 0 (heuristic) .. 8, got %d", value);
    g_opt[i].store(value);
    return 0;
}
extern "C" int pcgan_get_option(const char* key, int* value) {
    using namespace pcgan;
    const int i = option_index(key);
    PCGAN_CHECK(i >= 0 && value, "get_option: unknown option or null output");
    *value = g_opt[i].load();
    return 0;
}

extern "C" const char* pcgan_last_error(void) { return pcgan::g_err; }
extern "C" int pcgan_version(void) { return 100; }

extern "C" int pcgan_device_info(int* cu_count, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, 0);
    if (e != hipSuccess) {
        pcgan::set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return 1;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}
