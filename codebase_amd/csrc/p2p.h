// Peer-mapped buffers of the in-library gradient exchange (p2p.hip) and the device side of one exchange step, shared with the learner's
// reduce launch (dqn_update_kernels.h: the exchange fused into it).
#pragma once
#include "common.h"

namespace marl {

constexpr int P2P_MAX_WORLD = 16;
constexpr int P2P_CHUNK = 1024;  // floats per block of the stand-alone kernel (256 threads x float4)

struct P2pPeers {
    const float* slot[P2P_MAX_WORLD];      // peer r's data area (both slots)
    const uint32_t* flags[P2P_MAX_WORLD];  // peer r's flags
};

struct P2pState {
    int rank, world;
    int64_t max_floats;   // floats per slot (a multiple of P2P_CHUNK)
    int max_chunks;       // flags per parity: one per 64 floats of a slot
    size_t flag_bytes;    // [2 parities][max_chunks] uint32 epochs, then the error word; padded to 4 KB
    void* local;                 // this rank's allocation
    void* mapped[P2P_MAX_WORLD];  // peers' allocations as mapped here (nullptr for self)
    P2pPeers peers;
    uint32_t epoch;
    bool connected;
};


__device__ __forceinline__ uint32_t p2p_ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// The exchange for ONE wave holding 64 consecutive gradient values (element i0 + lane; `in` = inside the buffer): publish them in
// this rank's slot, raise flag `chunk64` of the epoch's parity, wait for the peers' flag, return the sum over the ranks in rank order.
// `late` comes back true on every lane when a peer did not arrive within the timeout (the caller keeps its local value).
__device__ __forceinline__ float p2p_wave_sum(const P2pPeers& peers, int rank, int world, int64_t slot_floats, int max_chunks, uint32_t epoch,
                                              int chunk64, int64_t i, bool in, float v, long long timeout_ticks, bool& late) {
    const int lane = threadIdx.x & 63, par = (int)(epoch & 1u);
    float* mine = const_cast<float*>(peers.slot[rank]) + (int64_t)par * slot_floats;
    uint32_t* my_flags = const_cast<uint32_t*>(peers.flags[rank]);
    if (in) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(my_flags + par * max_chunks + chunk64, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    bool mine_late = false;
    // once an exchange has timed out the run is broken anyway (marlhip_p2p_status reports it): later calls still publish, so that
    // healthy peers do not wait for this rank, but no longer wait themselves - a dead peer costs ONE timeout, not one per update
    const bool broken = __hip_atomic_load(my_flags + 2 * max_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (broken) mine_late = true;
    if (!broken && lane < world && lane != rank) {
        const uint32_t* f = peers.flags[lane] + par * max_chunks + chunk64;
        const long long t0 = wall_clock64();
        while ((int32_t)(p2p_ld_sys(f) - epoch) < 0) {
            if (wall_clock64() - t0 > timeout_ticks) {
                mine_late = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    late = __any(mine_late);
    if (late) {
        if (lane == 0) __hip_atomic_store(my_flags + 2 * max_chunks, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return v;
    }
    float acc = 0.f;
    for (int r = 0; r < world; ++r) {
        const float* src = peers.slot[r] + (int64_t)par * slot_floats;
        const float x = (r == rank || !in) ? v : __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc = r == 0 ? x : acc + x;
    }
    return acc;
}

long long p2p_timeout_ticks();

}  // namespace marl
