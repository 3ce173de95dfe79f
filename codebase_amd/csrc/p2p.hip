// In-library gradient exchange for the data-parallel learner: a one-shot all-reduce (SUM) over peer-mapped buffers, ONE kernel per
// update, no host hop, no collective library launch.  SURVEY.md 8(e): the path's only exchange is the flat gradient (44.6 KB at
// 2 agents x 64-64, 1 MB at most), latency-bound on xGMI - a ring or tree of a collective library pays a launch plus several link
// latencies per update; with every GPU of the node one xGMI hop away, each rank instead PUBLISHES its gradient in its own memory and
// READS the other G - 1 directly.
//
// Per rank: one uncached (fine-grained) device allocation [flags | slot 0 | slot 1], exported as a hipIpcMemHandle and mapped by
// every peer (marlhip_p2p_create / _connect; the handles travel through whatever the host side has - torch.distributed's object
// all-gather in codebase_amd/parallel.py).  marlhip_p2p_allreduce(ctx, grad, n, stream) - which has the marlhip_exchange_fn
// signature, so marlhip_idqn_update_n_dist takes it directly as its callback - launches one kernel in which block b
//   1. copies its chunk of `grad` into this rank's slot (epoch parity), fences, and raises this rank's flag of chunk b to the epoch;
//   2. waits until every peer's flag of chunk b has reached the epoch (system-scope loads; bounded: a peer that never arrives
//      sets the error word instead of hanging the GPU);
//   3. sums chunk b over the ranks IN RANK ORDER, from the slots (its own included), into `grad`.
// Every rank forms the same sums in the same order: replicas stay bitwise identical, as with the collective.  Two slots suffice: a
// rank can enter update u + 1 (other slot) while a peer still reads its slot of update u, but cannot finish the wait of u + 1 before
// that peer has entered u + 1 itself, i.e. has left u.  Nothing here exists in the reference (one process, no collective).
#include "common.h"

#include "p2p.h"

namespace marl {

// grad: this rank's gradient in, the sum over the ranks out.  flags layout: [parity][chunk], error word at index 2 * max_chunks.
// A workgroup takes the chunks blockIdx.x, blockIdx.x + gridDim.x, ...: it PUBLISHES all of them first, then waits for and sums one after
// the other - so at most gridDim.x workgroups ever sit spinning on a peer (marlhip_p2p_allreduce caps the grid: P2P_MAX_WGS).  With one
// workgroup per chunk a 1 MB gradient put a spinning workgroup on every compute unit; on a device shared by two ranks (the test rigs) the
// peer's learner kernels - one workgroup per CU, the whole register file - then could not start anywhere, and the exchange they precede
// never arrived: both ranks ran into the timeout (gpurun r6W: 4 of 7 runs on one box).  Every element is still summed in rank order from
// the same published values: the same bits as before.
static __global__ __launch_bounds__(256) void p2p_allreduce_kernel(float* __restrict__ grad, int64_t n, P2pPeers peers, int rank, int world,
                                                                   int64_t slot_floats, int max_chunks, uint32_t epoch, long long timeout_ticks,
                                                                   int chunks) {
    const int par = (int)(epoch & 1u);
    float* mine = const_cast<float*>(peers.slot[rank]) + (int64_t)par * slot_floats;
    uint32_t* my_flags = const_cast<uint32_t*>(peers.flags[rank]);
    for (int c = blockIdx.x; c < chunks; c += gridDim.x) {  // publish
        const int64_t i0 = (int64_t)c * P2P_CHUNK + 4 * threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) __hip_atomic_store(mine + i0 + k, grad[i0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __syncthreads();
        // flags are per 64 floats: this chunk's first one stands for it
        if (threadIdx.x == 0) __hip_atomic_store(my_flags + par * max_chunks + c * (P2P_CHUNK / 64), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __shared__ int s_late;
    for (int c = blockIdx.x; c < chunks; c += gridDim.x) {  // wait for the peers' chunk c (one lane per peer polls; epochs only grow), sum
        const int b = c * (P2P_CHUNK / 64);
        const int64_t i0 = (int64_t)c * P2P_CHUNK + 4 * threadIdx.x;
        if (threadIdx.x == 0) s_late = 0;
        __syncthreads();
        const bool broken = __hip_atomic_load(my_flags + 2 * max_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;  // see p2p_wave_sum
        if (broken && threadIdx.x == 0) s_late = 1;
        if (!broken && threadIdx.x < world && threadIdx.x != rank) {
            const uint32_t* f = peers.flags[threadIdx.x] + par * max_chunks + b;
            const long long t0 = wall_clock64();
            while ((int32_t)(p2p_ld_sys(f) - epoch) < 0) {
                if (wall_clock64() - t0 > timeout_ticks) {
                    s_late = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (s_late) {  // a peer never published: leave the local gradient, raise the error word (read by marlhip_p2p_status)
            if (threadIdx.x == 0) __hip_atomic_store(my_flags + 2 * max_chunks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {  // rank order on every rank: the same float sums everywhere
            const float* src = peers.slot[r] + (int64_t)par * slot_floats;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < n) {
                    const float x = r == rank ? grad[i0 + k] : __hip_atomic_load(src + i0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    acc[k] = r == 0 ? x : acc[k] + x;
                }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) grad[i0 + k] = acc[k];
        __syncthreads();  // s_late is reset for the next chunk
    }
}

// The exchange in the launch geometry of the learner's fused reduce (dqn_reduce_p2p_kernel): one 256-thread workgroup per 64 values, its
// wave 0 runs p2p_wave_sum, the other waves leave at once.  marlhip_p2p_allreduce_wave64: what the set-up's self-test runs at the real
// gradient size, so that the flags-per-64-floats protocol and the residency pattern of the fused launch (count / 64 workgroups, each
// spinning on its peer's same-index workgroup) have crossed the links before a training run depends on them.
static __global__ __launch_bounds__(256) void p2p_allreduce_wave64_kernel(float* __restrict__ grad, int64_t n, P2pPeers peers, int rank, int world,
                                                                          int64_t slot_floats, int max_chunks, uint32_t epoch, long long timeout_ticks) {
    if (threadIdx.x >= 64) return;
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool in = i < n;
    bool late;
    const float v = in ? grad[i] : 0.f;
    const float s = p2p_wave_sum(peers, rank, world, slot_floats, max_chunks, epoch, (int)blockIdx.x, i, in, v, timeout_ticks, late);
    if (in) grad[i] = s;
}

// wall_clock64 counts at 100 MHz.  The wait is bounded so that a dead peer cannot hang the GPU; the bound is DIAGNOSTIC (a collective
// library would block for ever): 5 minutes by default, MARLHIP_P2P_TIMEOUT_MS overrides; a malformed or non-positive value keeps the default.
long long p2p_timeout_ticks() {  // read per exchange (a getenv, ~0.1 us next to a launch): the set-up's self-test runs under a short bound of its own
    double ms = 300000.0;
    const char* v = getenv("MARLHIP_P2P_TIMEOUT_MS");
    if (v != nullptr && *v) {
        char* end = nullptr;
        const double x = strtod(v, &end);
        if (end != v && x > 0.0 && x < 8.64e7) {
            ms = x;
        } else {
            static bool warned = false;
            if (!warned) fprintf(stderr, "[marlhip] MARLHIP_P2P_TIMEOUT_MS=\"%s\" is not a positive number of milliseconds; keeping %.0f ms\n", v, ms);
            warned = true;
        }
    }
    return (long long)(ms * 1e5);
}

// marlhip_idqn_update_n_dist recognises this exchange by its address and folds it into the learner's reduce launch
bool p2p_is_builtin(marlhip_exchange_fn fn) { return fn == &marlhip_p2p_allreduce; }

}  // namespace marl

using namespace marl;

extern "C" int marlhip_p2p_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// One allocation per rank: [flags | slot 0 | slot 1], uncached so that stores and the system-scope loads of the peers meet in
// memory, not in a cache.  `handle_out` receives the hipIpcMemHandle (marlhip_p2p_handle_bytes() bytes) to hand to the peers.
extern "C" int marlhip_p2p_create(int32_t rank, int32_t world, int64_t max_floats, void** state_out, void* handle_out) {
    MARL_REQUIRE(state_out && handle_out, "p2p_create: NULL pointer");
    MARL_REQUIRE(world >= 1 && world <= P2P_MAX_WORLD && rank >= 0 && rank < world, "p2p_create: rank %d / world %d (at most %d ranks)", rank, world,
                 P2P_MAX_WORLD);
    MARL_REQUIRE(max_floats > 0, "p2p_create: max_floats must be > 0");
    const int64_t slot = (max_floats + P2P_CHUNK - 1) / P2P_CHUNK * P2P_CHUNK;
    MARL_REQUIRE(slot <= (int64_t)1 << 28, "p2p_create: %lld floats per exchange is beyond what this buffer layout is meant for", (long long)max_floats);
    const int max_chunks = (int)(slot / 64);  // one flag per 64 floats (the learner's reduce launch exchanges 64 values per workgroup)
    P2pState* st = new P2pState();
    st->rank = rank; st->world = world; st->max_floats = slot; st->max_chunks = max_chunks; st->epoch = 0; st->connected = false;
    st->flag_bytes = ((size_t)(2 * (int64_t)max_chunks + 1) * 4 + 4095) & ~(size_t)4095;
    for (int r = 0; r < P2P_MAX_WORLD; ++r) { st->mapped[r] = nullptr; st->peers.slot[r] = nullptr; st->peers.flags[r] = nullptr; }
    const size_t bytes = st->flag_bytes + 2 * (size_t)slot * sizeof(float);
    hipError_t e = hipExtMallocWithFlags(&st->local, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        set_error("p2p_create: hipExtMallocWithFlags(%zu bytes, uncached): %s", bytes, hipGetErrorString(e));
        delete st;
        return -2;
    }
    // flags, error word and slots must read zero before any peer can see them: the memset runs on the null stream, the exchanges on the
    // caller's (possibly non-blocking) stream, so wait for it here - before the handle leaves this process
    e = hipMemset(st->local, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("p2p_create: zeroing the exchange buffer: %s", hipGetErrorString(e));
        (void)hipFree(st->local);
        delete st;
        return -2;
    }
    e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle_out), st->local);
    if (e != hipSuccess) {
        set_error("p2p_create: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be exported)", hipGetErrorString(e));
        (void)hipFree(st->local);
        delete st;
        return -2;
    }
    *state_out = st;
    return 0;
}

// handles: world x marlhip_p2p_handle_bytes() bytes, rank-major (this rank's own entry is ignored).  Every rank calls this after all
// ranks have created theirs; the peers' allocations are mapped into this process (peer access is enabled on demand).
extern "C" int marlhip_p2p_connect(void* state, const void* handles) {
    P2pState* st = static_cast<P2pState*>(state);
    MARL_REQUIRE(st && handles, "p2p_connect: NULL pointer");
    const hipIpcMemHandle_t* h = static_cast<const hipIpcMemHandle_t*>(handles);
    for (int r = 0; r < st->world; ++r) {
        void* base = st->local;
        if (r != st->rank) {
            const hipError_t e = hipIpcOpenMemHandle(&base, h[r], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                set_error("p2p_connect: hipIpcOpenMemHandle of rank %d's buffer: %s", r, hipGetErrorString(e));
                return -2;
            }
            st->mapped[r] = base;
        }
        st->peers.flags[r] = reinterpret_cast<const uint32_t*>(base);
        st->peers.slot[r] = reinterpret_cast<const float*>(static_cast<const char*>(base) + st->flag_bytes);
    }
    st->connected = true;
    return 0;
}

// marlhip_exchange_fn: ctx = the state of marlhip_p2p_create.  Leaves the SUM over the ranks in grad; enqueues one kernel on
// `stream`, never synchronises.  Every rank must call it the same number of times with the same count (the epoch is the call count).
extern "C" int marlhip_p2p_allreduce(void* ctx, float* grad, int64_t count, void* stream) {
    P2pState* st = static_cast<P2pState*>(ctx);
    MARL_REQUIRE(st && grad, "p2p_allreduce: NULL pointer");
    MARL_REQUIRE(st->connected, "p2p_allreduce: marlhip_p2p_connect has not run");
    MARL_REQUIRE(count > 0 && count <= st->max_floats, "p2p_allreduce: %lld floats, the buffers hold %lld", (long long)count, (long long)st->max_floats);
    st->epoch += 1;
    const int chunks = (int)((count + P2P_CHUNK - 1) / P2P_CHUNK);
    const long long timeout_ticks = p2p_timeout_ticks();
    // at most P2P_MAX_WGS workgroups wait on a peer at a time (see the kernel); MARLHIP_P2P_MAX_WGS overrides (diagnostics: a large value
    // restores one workgroup per chunk)
    int cap = 32;
    if (const char* v = getenv("MARLHIP_P2P_MAX_WGS")) {
        const int x = atoi(v);
        if (x >= 1) cap = x;
    }
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(chunks < cap ? chunks : cap), dim3(256), 0, (hipStream_t)stream, grad, count, st->peers, st->rank, st->world,
                       st->max_floats, st->max_chunks, st->epoch, timeout_ticks, chunks);
    MARL_CHECK_LAUNCH("p2p_allreduce_kernel");
    return 0;
}

// The same exchange, launched as the learner's fused reduce launches it (one workgroup per 64 values; see the kernel).  Shares the epoch
// counter with marlhip_p2p_allreduce: every rank must issue the same sequence of calls of either kind.
extern "C" int marlhip_p2p_allreduce_wave64(void* ctx, float* grad, int64_t count, void* stream) {
    P2pState* st = static_cast<P2pState*>(ctx);
    MARL_REQUIRE(st && grad, "p2p_allreduce_wave64: NULL pointer");
    MARL_REQUIRE(st->connected, "p2p_allreduce_wave64: marlhip_p2p_connect has not run");
    MARL_REQUIRE(count > 0 && count <= st->max_floats, "p2p_allreduce_wave64: %lld floats, the buffers hold %lld", (long long)count, (long long)st->max_floats);
    st->epoch += 1;
    hipLaunchKernelGGL(p2p_allreduce_wave64_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, (hipStream_t)stream, grad, count, st->peers,
                       st->rank, st->world, st->max_floats, st->max_chunks, st->epoch, p2p_timeout_ticks());
    MARL_CHECK_LAUNCH("p2p_allreduce_wave64_kernel");
    return 0;
}

// 0 = every exchange so far saw all its peers; 1 = some wait ran into the timeout (the sums of that call are not valid).
// Synchronises with the device (a 4-byte read of the error word): for tests and end-of-run checks, not for the hot loop.
extern "C" int marlhip_p2p_status(void* state) {
    P2pState* st = static_cast<P2pState*>(state);
    MARL_REQUIRE(st, "p2p_status: NULL pointer");
    uint32_t err = 0;
    const hipError_t e = hipMemcpy(&err, static_cast<const uint32_t*>(st->local) + 2 * st->max_chunks, sizeof(err), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        set_error("p2p_status: %s", hipGetErrorString(e));
        return -2;
    }
    return err ? 1 : 0;
}

extern "C" int marlhip_p2p_destroy(void* state) {
    P2pState* st = static_cast<P2pState*>(state);
    if (st == nullptr) return 0;
    for (int r = 0; r < st->world; ++r)
        if (st->mapped[r] != nullptr) (void)hipIpcCloseMemHandle(st->mapped[r]);
    if (st->local != nullptr) (void)hipFree(st->local);
    delete st;
    return 0;
}
