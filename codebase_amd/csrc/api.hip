// libmarlhip.so: version / error text / device probe.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace marl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct KTimer {
    static constexpr int MAXP = 8192;
    hipEvent_t a[MAXP], b[MAXP];
    int created = 0, n = 0;
    bool open = false;
};
static KTimer g_timers[TIMER_COUNT];
static bool g_timing = false;

void timing_begin(int id, hipStream_t st) {
    if (!g_timing) return;
    KTimer& t = g_timers[id];
    if (t.n >= KTimer::MAXP) return;
    if (t.n >= t.created) {
        if (hipEventCreate(&t.a[t.created]) != hipSuccess || hipEventCreate(&t.b[t.created]) != hipSuccess) return;
        ++t.created;
    }
    (void)hipEventRecord(t.a[t.n], st);
    t.open = true;
}
void timing_end(int id, hipStream_t st) {
    if (!g_timing) return;
    KTimer& t = g_timers[id];
    if (!t.open) return;
    (void)hipEventRecord(t.b[t.n], st);
    ++t.n;
    t.open = false;
}

struct PackScratch {
    int dev;
    hipStream_t st;
    float* buf;
    size_t cap;
};
static PackScratch g_scratch[16];
static int g_nscratch = 0;

float* collect_pack_scratch(size_t bytes, hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    PackScratch* e = nullptr;
    for (int i = 0; i < g_nscratch; ++i)
        if (g_scratch[i].dev == dev && g_scratch[i].st == st) e = &g_scratch[i];
    if (e == nullptr) {
        if (g_nscratch == 16) return nullptr;
        e = &g_scratch[g_nscratch++];
        e->dev = dev; e->st = st; e->buf = nullptr; e->cap = 0;
    }
    if (bytes > e->cap) {
        if (e->buf != nullptr) {
            (void)hipStreamSynchronize(st);  // a kernel of this stream may still read the old buffer
            (void)hipFree(e->buf);
        }
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        if (hipMalloc(reinterpret_cast<void**>(&e->buf), want) != hipSuccess) {
            e->buf = nullptr; e->cap = 0;
            return nullptr;
        }
        e->cap = want;
    }
    return e->buf;
}
}  // namespace marl

extern "C" int marlhip_version(void) { return MARLHIP_VERSION; }
extern "C" const char* marlhip_last_error(void) { return marl::g_err; }
extern "C" int marlhip_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

extern "C" int marlhip_timing_enable(int on) {
    marl::g_timing = on != 0;
    for (int i = 0; i < marl::TIMER_COUNT; ++i) { marl::g_timers[i].n = 0; marl::g_timers[i].open = false; }
    return 0;
}
extern "C" int marlhip_timing_read(int id, int64_t* launches, double* total_ms) {
    if (id < 0 || id >= marl::TIMER_COUNT || !launches || !total_ms) { marl::set_error("timing_read: bad argument"); return -1; }
    marl::KTimer& t = marl::g_timers[id];
    double sum = 0.0;
    for (int i = 0; i < t.n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(t.b[i]) != hipSuccess || hipEventElapsedTime(&ms, t.a[i], t.b[i]) != hipSuccess) {
            marl::set_error("timing_read: event query failed");
            return -2;
        }
        sum += ms;
    }
    *launches = t.n;
    *total_ms = sum;
    t.n = 0;
    return 0;
}
