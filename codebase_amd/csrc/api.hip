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

static thread_local char* g_scratch_base = nullptr;
static thread_local int64_t g_scratch_bytes = 0;

void scratch_bind(void* base, int64_t bytes) {
    g_scratch_base = static_cast<char*>(base);
    g_scratch_bytes = base != nullptr ? bytes : 0;
}

static thread_local AcGhost g_ac_ghost = {};
const AcGhost& ac_ghost_current() { return g_ac_ghost; }
void ac_ghost_bind(const AcGhost* g) { g_ac_ghost = g != nullptr ? *g : AcGhost{}; }

static thread_local AcKeep g_ac_keep = {};
const AcKeep& ac_keep_current() { return g_ac_keep; }
void ac_keep_bind(const AcKeep* k) { g_ac_keep = k != nullptr ? *k : AcKeep{}; }

int64_t scratch_avail() {  // bytes collect_pack_scratch can hand out in this call
    if (g_scratch_base == nullptr) return 0;
    char* b = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(g_scratch_base) + 15) & ~(uintptr_t)15);
    const int64_t avail = g_scratch_bytes - (b - g_scratch_base);
    return avail > 0 ? avail : 0;
}

float* collect_pack_scratch(size_t bytes, hipStream_t) {
    char* b = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(g_scratch_base) + 15) & ~(uintptr_t)15);
    const int64_t avail = g_scratch_base != nullptr ? g_scratch_bytes - (b - g_scratch_base) : 0;
    if (g_scratch_base == nullptr || (int64_t)bytes > avail) {
        set_error("workspace: %zu bytes of weight-pack scratch needed, the caller's workspace holds %lld (marlhip_forward_workspace_bytes)",
                  bytes + 16, (long long)(avail > 0 ? avail : 0));
        return nullptr;
    }
    return reinterpret_cast<float*>(b);
}
}  // namespace marl

extern "C" int marlhip_version(void) { return MARLHIP_VERSION; }
extern "C" const char* marlhip_last_error(void) { return marl::g_err; }
extern "C" int marlhip_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

extern "C" int64_t marlhip_forward_workspace_bytes(const marlhip_net_shape* s) {
    if (s == nullptr || s->n_agents < 1 || s->n_agents > 16 || s->hidden < 16 || s->hidden % 16 != 0 || s->obs_dim < 1) {
        marl::set_error("forward_workspace_bytes: bad net shape");
        return -1;
    }
    // the largest forward pack any entry point builds for this shape: MLP or recurrent layout (mlp.h / gru.h), own or concatenated
    // (centralised critic) observations; two pack sets (paired recurrent passes) of every agent
    const int64_t P = s->n_agents, H = s->hidden, MT = H / 16;
    const int64_t Dmax = (int64_t)s->obs_dim * P, KS1 = (Dmax + 15) / 16 * 4;
    const int64_t mlp = MT * KS1 * 64 + MT * MT * 256 + 2 * H + 16 + MT * 256;
    const int64_t gru = MT * KS1 * 64 + 6 * MT * MT * 256 + MT * 256 + 7 * H + 16;
    return 2 * P * (mlp > gru ? mlp : gru) * 4 + 256;
}

extern "C" int marlhip_stream_create_cu_share(int32_t percent, int32_t pattern, void** stream_out) {
    using namespace marl;
    MARL_REQUIRE(stream_out != nullptr && percent >= 1 && percent <= 100 && (pattern == 0 || pattern == 1),
                 "stream_create_cu_share: percent %d (1..100), pattern %d (0 / 1)", percent, pattern);
    int dev = 0, cus = 0;
    MARL_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0,
                 "stream_create_cu_share: no device");
    const int want = (cus * percent + 99) / 100;
    uint32_t mask[32] = {0};
    MARL_REQUIRE(cus <= 32 * 32, "stream_create_cu_share: %d compute units", cus);
    int set = 0;
    for (int i = 0; i < cus && set < want; ++i) {
        const bool take = pattern == 0 || (i & 1) == 0 || want > (cus + 1) / 2;  // every other unit first; above one half every unit is a candidate
        if (take) {
            mask[i >> 5] |= 1u << (i & 31);
            ++set;
        }
    }
    hipStream_t st = nullptr;
    MARL_REQUIRE(hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask) == hipSuccess, "stream_create_cu_share: hipExtStreamCreateWithCUMask failed");
    *stream_out = st;
    return 0;
}

extern "C" int marlhip_stream_destroy(void* stream) {
    using namespace marl;
    MARL_REQUIRE(stream != nullptr, "stream_destroy: NULL stream");
    MARL_REQUIRE(hipStreamDestroy((hipStream_t)stream) == hipSuccess, "stream_destroy: hipStreamDestroy failed");
    return 0;
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
