// trsim_resident.hip — the resident worker: a consumer-paced env step without a kernel launch per step.
//
// What it replaces: Car.start calls GymInterface.step once per tick with that tick's controls (reference core/car.py:45-53,
// components/gyminterface.py:66-76).  With one launch per call every step pays a kernel boundary and the re-staging of
// 124 KB of LDS tables (~3 us of a 13.5 us step at 1024 envs).  In resident mode (trs_set_step_mode) the first trs_step
// launches trs_worker_kernel, which stays on the GPU: tables staged once, env state in LDS, one workgroup per CU exactly as
// in trs_step_kernel (8 raster waves + 4 physics waves).  Every later trs_step only POSTS: the host writes the step's control
// pointers into a ring in pinned host memory and bumps a counter; nothing else crosses the API.
//
//   host  --(pinned mailbox: posted, close, ring[8])-->  dispatcher (workgroup 0, physics wave 0; polls over PCIe)
//         --(device word + ring copy, sc1)-->  one leader wave per workgroup  --(LDS word)-->  the other 11 waves
//   physics team: per step, per env: controls (system-scope loads) -> env_advance -> state written through to HBM (system
//         scope), camera parameters into an LDS ring kCamDepth deep; it runs ahead of the raster team as far as posts and
//         the ring allow (back-pressure: a slot is reused once all raster waves have read it)
//   raster team: per step: the uniform rows of ALL the workgroup's envs first (they need no pose), then per env the rows
//         that see the track; every store write-through (sc0 sc1)
//   completion: every wave drains its stores (s_waitcnt vmcnt(0)) and arrives on an LDS counter; the workgroup's last wave
//         arrives on a device counter sharded by blockIdx % 8, the last shard on a top counter, and the last of all stores
//         done[step % 8] = step + 1 into the mailbox (system scope).  One writer per flag and step, so the host never sees
//         a counter run backwards.
//   exit: only the dispatcher decides (close requested, no post for idle_us, or lifetime spent): it stores `exited`, drains,
//         reads `posted` one last time, publishes EXIT | posted — every workgroup finishes the steps below that count and
//         leaves.  A post that raced with the exit is seen by the host (`exited` set after its post) and the host relaunches
//         from `consumed`.  Every spin in the kernel is bounded (safety timeout -> abort bit -> every wave leaves).
// Hand-off rules follow /opt/skills/guides/cdna_hip_programming.md Guideline 16: shared words are agent/system-scope
// accesses, payload stores are write-through and drained by every storing wave before the wave that signals does so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/trsim.h"
#include "../../include/trsim_spec.h"
#include "trsim_device.hpp"
#include "trsim_env.hpp"
#include "trsim_internal.hpp"

#define TRS_EXPORT extern "C" __attribute__((visibility("default")))

namespace trsim {

constexpr int kSlots = 8;       // posts in flight: ring entries, arrival counters, done flags
constexpr int kCamDepth = 4;    // steps the physics team may run ahead of the raster team

struct WEntry {                 // one posted step (64 B)
    const float* steer; const float* thr; const float* brk; const uint8_t* reset;
    uint32_t synth, pad0; uint64_t pad1[3];
};
static_assert(sizeof(WEntry) == 64, "one entry per 64-B line");

struct Mailbox {                // pinned host memory the device reads and writes over PCIe
    alignas(64) uint64_t posted;            // host -> device: steps [0, posted) have been posted (absolute step indices)
    uint32_t close, pad0;                   // host -> device: leave once everything posted is done
    alignas(64) uint64_t exited;            // device -> host: the dispatcher has decided to leave
    uint64_t consumed;                      //   ... and every step below this index is processed by the time the kernel ends
    uint64_t error;                         //   non-zero: a bounded wait gave up (code << 32 | block)
    alignas(64) uint64_t done[kSlots];      // device -> host: done[s % 8] = s + 1 when step s is complete in memory
    alignas(64) WEntry ring[kSlots];        // host -> device
};

struct DevCtl {                 // device memory; touched only by sc1 accesses and atomics
    alignas(64) unsigned long long word;    // posted count | kExitBit | kAbortBit, republished by the dispatcher
    alignas(64) WEntry ring[kSlots];
    alignas(64) unsigned arrive[kSlots][8][16];   // [slot][blockIdx % 8]: one 64-B line each
    alignas(64) unsigned top[kSlots][16];
};

struct WParams {
    PParams ph;
    RParams ra;
    uint8_t *img0, *img1;
    float *dep0, *dep1;
    Mailbox* mb;
    DevCtl* dc;
    unsigned long long start;                               // first step index this launch processes
    unsigned long long idle_ticks, life_ticks, safety_ticks;   // wall_clock64() ticks (100 MHz)
    int lds_off_phys, lds_off_ctl, n_blocks;
};

struct Resident {
    bool enabled = false, running = false;
    Mailbox* mb = nullptr;
    DevCtl* dc = nullptr;
    hipStream_t sC = nullptr;            // copies while the worker owns the handle's stream
    uint64_t base = 0;                   // steps [base, step_count) were handed to the worker since the last quiesce
    uint64_t seen_done = 0;              // every step below this index has been observed complete
    unsigned idle_us = 2000;
    unsigned char* hctl = nullptr;       // pinned staging for host-array controls: [kSlots] x (3 float[n] + uint8[n])
    size_t hctl_slot = 0;
    int lds_bytes = 0, lds_off_ctl = 0;
};

}  // namespace trsim

namespace {

using namespace trsim;
using u64 = unsigned long long;

constexpr u64 kExitBit = 1ull << 63, kAbortBit = 1ull << 62, kCountMask = kAbortBit - 1;
constexpr int kWaves = kBlock / 64;

// ---- scoped accesses ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 sys_load64(const void* p) { return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned sys_load32(const void* p) { return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store64(void* p, u64 v) { __hip_atomic_store((u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 agent_load64(const void* p) { return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store64(void* p, u64 v) { __hip_atomic_store((u64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 lds_load64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_load32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store32(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void drain_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// a control value from an array the host named in a post: a GLOBAL (never flat) system-scope load, so that neither this CU's L1
// nor this XCD's L2 can answer with what an earlier step read from the same address
template <typename T>
__device__ __forceinline__ T sys_load_val(const T* p)
{
    return __hip_atomic_load((const __attribute__((address_space(1))) T*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// what a workgroup shares in LDS (behind the tables)
struct WLds {
    u64* word;          // posted count | flags as last seen by this workgroup's leader
    int* arrive;        // [kSlots] waves that have finished step s (stores drained)
    int* pprog;         // [epw] steps the physics team has finished per env (relative to the launch's first step)
    int* rread;         // [epw] raster-wave reads of camera slots per env (back-pressure)
    float4* lcam;       // [kCamDepth][epw]
    float* lst;         // [epw][16] env state
};

__device__ __forceinline__ WLds wlds_of(unsigned char* base, int epw)
{
    WLds l;
    l.word = reinterpret_cast<u64*>(base);
    l.arrive = reinterpret_cast<int*>(base + 16);
    l.pprog = reinterpret_cast<int*>(base + 64);
    l.rread = l.pprog + epw;
    const size_t cam_off = (64 + (size_t)epw * 8 + 15) & ~(size_t)15;
    l.lcam = reinterpret_cast<float4*>(base + cam_off);
    l.lst = reinterpret_cast<float*>(base + cam_off + (size_t)kCamDepth * epw * 16);
    return l;
}

__host__ __device__ inline size_t wlds_bytes(int epw)
{
    const size_t cam_off = (64 + (size_t)epw * 8 + 15) & ~(size_t)15;
    return cam_off + (size_t)kCamDepth * epw * 16 + (size_t)epw * 64;
}

// a bounded wait gave up: tell the host, every workgroup (device word) and this workgroup (LDS word)
__device__ __forceinline__ void worker_abort(const WParams& wp, const WLds& l, unsigned code)
{
    sys_store64(&wp.mb->error, ((u64)code << 32) | (u64)blockIdx.x);
    __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Leader { u64 known, t_last, t_start; };

// dispatcher (workgroup 0's leader): copy newly posted entries host -> device, then republish the count
__device__ __forceinline__ void dispatcher_publish(const WParams& wp, const WLds& l, Leader& L, u64 hp, bool leaving, int lane)
{
    for (u64 q = L.known; q < hp; ++q) {
        const u64* src = reinterpret_cast<const u64*>(&wp.mb->ring[q & (kSlots - 1)]);
        u64* dst = reinterpret_cast<u64*>(&wp.dc->ring[q & (kSlots - 1)]);
        if (lane < 8) agent_store64(dst + lane, sys_load64(src + lane));
    }
    drain_vmem();
    const u64 w = hp | (leaving ? kExitBit : 0ull);
    if (lane == 0) { agent_store64(&wp.dc->word, w); lds_store64(l.word, w); }
    drain_vmem();
    L.known = hp;
}

__device__ __forceinline__ void dispatcher_poll(const WParams& wp, const WLds& l, Leader& L, int lane)
{
    const u64 now = (u64)wall_clock64();
    const u64 hp = sys_load64(&wp.mb->posted);
    if (hp > L.known) { dispatcher_publish(wp, l, L, hp, false, lane); L.t_last = now; return; }
    const unsigned hc = sys_load32(&wp.mb->close);
    if (hc != 0u || now - L.t_last > wp.idle_ticks || now - L.t_start > wp.life_ticks) {
        if (lane == 0) sys_store64(&wp.mb->exited, 1ull);
        drain_vmem();                                        // `exited` is in host memory before the last look at `posted`
        const u64 hp2 = sys_load64(&wp.mb->posted);
        if (lane == 0) sys_store64(&wp.mb->consumed, hp2);
        dispatcher_publish(wp, l, L, hp2, true, lane);
        return;
    }
    __builtin_amdgcn_s_sleep(8);
}

// 1 = step s is posted (go), 0 = leave.  Leaders refresh the workgroup's LDS word; the others only read it.
__device__ __forceinline__ int wait_posted(const WParams& wp, const WLds& l, bool leader, Leader& L, u64 s, int lane)
{
    u64 t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        const u64 w = lds_load64(l.word);
        if (w & kAbortBit) return 0;
        if ((w & kCountMask) > s) return 1;
        if (w & kExitBit) return 0;
        if (leader) {
            if (blockIdx.x == 0) { dispatcher_poll(wp, l, L, lane); continue; }
            const u64 g = agent_load64(&wp.dc->word);
            if (g != w) { if (lane == 0) lds_store64(l.word, g); drain_lds(); continue; }
        }
        __builtin_amdgcn_s_sleep(4);
        if ((spins & 255u) == 255u) {
            const u64 now = (u64)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > wp.safety_ticks) { worker_abort(wp, l, 1u); return 0; }
        }
    }
}

// bounded wait on an LDS counter; 0 = gave up / aborted
__device__ __forceinline__ int wait_lds_ge(const WParams& wp, const WLds& l, const int* ctr, int want, unsigned code)
{
    u64 t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        if (lds_load32(ctr) >= want) return 1;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (lds_load64(l.word) & kAbortBit) return 0;
            const u64 now = (u64)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > wp.safety_ticks) { worker_abort(wp, l, code); return 0; }
        }
    }
}

// this wave has finished step s: drain its stores, arrive; the last wave of the workgroup arrives for the workgroup, the last
// workgroup of a shard for the shard, the last shard tells the host
__device__ __forceinline__ void wave_arrive(const WParams& wp, const WLds& l, u64 s, int lane)
{
    drain_vmem();
    if (lane != 0) return;
    const int slot = (int)(s & (kSlots - 1));
    const int old = __hip_atomic_fetch_add(&l.arrive[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old != kWaves - 1) return;
    lds_store32(&l.arrive[slot], 0);
    const int shard = (int)(blockIdx.x & 7u), nshards = min(8, wp.n_blocks);
    const unsigned members = (unsigned)((wp.n_blocks - shard + 7) / 8);
    unsigned* a = &wp.dc->arrive[slot][shard][0];
    if (__hip_atomic_fetch_add(a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != members - 1u) return;
    __hip_atomic_store(a, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* t = &wp.dc->top[slot][0];
    if (__hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)nshards - 1u) return;
    __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_vmem();                                            // the counters are back at zero before the host can post step s + 8
    sys_store64(&wp.mb->done[slot], s + 1);
}

template <bool DEPTH>
__global__ __launch_bounds__(kBlock) void trs_worker_kernel(const WParams wp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool raster_team = tid < kRasterThreads;
    const RParams& p = wp.ra;
    const int epw = p.envs_per_wg;
    const WLds l = wlds_of(smem + wp.lds_off_ctl, epw);
    const int e_begin = blockIdx.x * epw;
    const int n_loc = min(e_begin + epw, p.n_envs) - e_begin;
    const int pw = wave - kRasterThreads / 64;
    if ((unsigned)(uintptr_t)smem != 0u) {                   // the map addressing assumes LDS offset 0
        if (tid == 0) {
            atomicAdd(&p.stats[2], 1ull);
            sys_store64(&wp.mb->error, (9ull << 32) | (u64)blockIdx.x);
            __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) { sys_store64(&wp.mb->consumed, wp.start); sys_store64(&wp.mb->exited, 1ull); }
        }
        return;
    }
    // ---- once per launch: tables by LDS-DMA, control block, env state ----
    if (raster_team) stage_lds_dma(p.blob, p.blob_bytes, 0u, wave, kRasterThreads / 64, lane);
    else stage_lds_dma(wp.ph.blob, wp.ph.blob_bytes, (unsigned)wp.lds_off_phys, pw, kPhysWaves, lane);
    if (tid == 0) lds_store64(l.word, wp.start);
    if (tid < kSlots) l.arrive[tid] = 0;
    for (int j = tid; j < 2 * epw; j += kBlock) l.pprog[j] = 0;          // pprog | rread
    if (!raster_team)
        for (int j = pw; j < n_loc; j += kPhysWaves) {
            EnvRegs st;
            env_load(wp.ph, e_begin + j, st);
            if (lane == 0) {
                float* q = l.lst + (size_t)j * 16;
                q[0] = st.x; q[1] = st.y; q[2] = st.z; q[3] = st.yaw; q[4] = st.v; q[5] = st.sf; q[6] = st.epr;
                q[7] = __int_as_float(st.seg); q[8] = __int_as_float(st.epl); q[9] = __int_as_float(st.done); q[10] = __int_as_float(st.pend);
            }
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    if (!raster_team) {
        // ---- physics team ----
        Leader L{wp.start, (u64)wall_clock64(), (u64)wall_clock64()};
        const bool leader = pw == 0;
        const unsigned char* const lphys = smem + wp.lds_off_phys;
        const PParams& P = wp.ph;
        for (u64 s = wp.start;; ++s) {
            const int r = (int)(s - wp.start);
            if (!wait_posted(wp, l, leader, L, s, lane)) return;
            const u64* en = reinterpret_cast<const u64*>(&wp.dc->ring[s & (kSlots - 1)]);
            const u64 ev = lane < 5 ? agent_load64(en + lane) : 0ull;
            auto bcast = [&](int k) -> u64 {
                return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(ev >> 32), k) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)ev, k);
            };
            const float* const c_st = reinterpret_cast<const float*>(bcast(0));
            const float* const c_th = reinterpret_cast<const float*>(bcast(1));
            const float* const c_br = reinterpret_cast<const float*>(bcast(2));
            const uint8_t* const c_rs = reinterpret_cast<const uint8_t*>(bcast(3));
            const int synth = (int)(unsigned)bcast(4);
            for (int j = pw; j < n_loc; j += kPhysWaves) {
                const int e = e_begin + j;
                // back-pressure: camera slot r % kCamDepth is free once every raster wave has read step r - kCamDepth of this env
                if (r >= kCamDepth && !wait_lds_ge(wp, l, &l.rread[j], (r - kCamDepth + 1) * (kRasterThreads / 64), 2u)) return;
                float steer = 0.f, thr = 0.f, brk = 0.f;
                uint8_t rin = 0;
                if (!synth) {
                    steer = sys_load_val(&c_st[e]); thr = sys_load_val(&c_th[e]);
                    if (c_br) brk = sys_load_val(&c_br[e]);
                    if (c_rs) rin = sys_load_val(&c_rs[e]);
                }
                float* const q = l.lst + (size_t)j * 16;
                EnvRegs st;
                st.x = q[0]; st.y = q[1]; st.z = q[2]; st.yaw = q[3]; st.v = q[4]; st.sf = q[5]; st.epr = q[6];
                st.seg = __float_as_int(q[7]); st.epl = __float_as_int(q[8]); st.done = __float_as_int(q[9]); st.pend = __float_as_int(q[10]);
                st.speed = 0.f; st.cte = 0.f;
                StepOut o;
                env_advance<true>(P, lphys, e, st, (uint32_t)s, synth, steer, thr, brk, rin, lane, o);
                if (lane == 0) {
                    q[0] = st.x; q[1] = st.y; q[2] = st.z; q[3] = st.yaw; q[4] = st.v; q[5] = st.sf; q[6] = st.epr;
                    q[7] = __int_as_float(st.seg); q[8] = __int_as_float(st.epl); q[9] = __int_as_float(st.done); q[10] = __int_as_float(st.pend);
                    l.lcam[(size_t)(r & (kCamDepth - 1)) * epw + j] = o.cam;
                    store_out<true>(&P.x[e], st.x); store_out<true>(&P.y[e], st.y); store_out<true>(&P.z[e], st.z);
                    store_out<true>(&P.yaw[e], st.yaw); store_out<true>(&P.v[e], st.v); store_out<true>(&P.speed[e], st.speed);
                    store_out<true>(&P.cte[e], st.cte); store_out<true>(&P.seg_idx[e], (int32_t)st.seg);
                    store_out<true>(&P.done[e], (uint8_t)st.done); store_out<true>(&P.ep_return[e], st.epr);
                    store_out<true>(&P.ep_len[e], (int32_t)st.epl); store_out<true>(&P.steer_filt[e], st.sf);
                    store_out<true>(&P.pending[e], (uint8_t)0);
                    if (o.is_done) atomicAdd(&P.stats[0], 1ull);
                    if (o.do_reset) atomicAdd(&P.stats[1], 1ull);
                    drain_lds();                              // state and camera parameters are in LDS before the counter moves
                    lds_store32(&l.pprog[j], r + 1);
                }
            }
            wave_arrive(wp, l, s, lane);
        }
    }

    // ---- raster team ----
    const RasterThread rth = raster_thread(p, smem, tid);
    Leader none{0, 0, 0};
    for (u64 s = wp.start;; ++s) {
        const int r = (int)(s - wp.start);
        if (!wait_posted(wp, l, false, none, s, lane)) return;
        uint8_t* const img = (s & 1ull) ? wp.img1 : wp.img0;
        float* const dep = (s & 1ull) ? wp.dep1 : wp.dep0;
        for (int j = 0; j < n_loc; ++j)                      // rows that need no pose: every env's first, while the physics team integrates
            raster_uniform_rows<DEPTH>(p, rth, frame_desc<DEPTH>(p, img, dep, e_begin + j));
        for (int j = 0; j < n_loc; ++j) {
            if (!wait_lds_ge(wp, l, &l.pprog[j], r + 1, 3u)) return;
            const float4 cam = l.lcam[(size_t)(r & (kCamDepth - 1)) * epw + j];
            asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(cam.x), "v"(cam.y), "v"(cam.z), "v"(cam.w) : "memory");
            if (lane == 0) __hip_atomic_fetch_add(&l.rread[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            raster_ground_rows<DEPTH>(p, rth, frame_desc<DEPTH>(p, img, dep, e_begin + j), cam);
        }
        wave_arrive(wp, l, s, lane);
    }
}

// zero the device control block and set its word before a launch (stream-ordered in front of the worker)
__global__ void trs_worker_init_kernel(DevCtl* dc, u64 start)
{
    unsigned* w = reinterpret_cast<unsigned*>(dc);
    for (int i = threadIdx.x; i < (int)(sizeof(DevCtl) / 4); i += blockDim.x) w[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) agent_store64(&dc->word, start);
}

// ---- host side ------------------------------------------------------------------------------------------------------

#define RCHK(call)                                                                                \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return trs_internal_fail(TRS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline uint64_t host_load(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void host_store(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

int worker_launch(trs_env* e, uint64_t start)
{
    Resident* R = e->res;
    Mailbox* mb = R->mb;
    host_store(&mb->exited, 0); host_store(&mb->consumed, start); host_store(&mb->error, 0);
    __atomic_store_n(&mb->close, 0u, __ATOMIC_RELEASE);
    WParams wp{};
    wp.ph = e->pp; wp.ph.synth = 0; wp.ph.write_cam = 0; wp.ph.n_steps = 0; wp.ph.step_off = 0; wp.ph.ctl_stride = 0;
    wp.ra = e->rp;
    wp.img0 = e->img[0]; wp.img1 = e->img[1]; wp.dep0 = e->depth[0]; wp.dep1 = e->depth[1];
    wp.mb = mb; wp.dc = R->dc;
    wp.start = start;
    wp.idle_ticks = (unsigned long long)R->idle_us * 100ull;
    wp.life_ticks = 50000000ull;                            // 0.5 s: then the dispatcher leaves at the next lull and the host relaunches
    wp.safety_ticks = 200000000ull;                         // 2 s
    wp.lds_off_phys = e->lds_off_phys; wp.lds_off_ctl = R->lds_off_ctl;
    const int grid = (e->n + e->pp.envs_per_wg - 1) / e->pp.envs_per_wg;
    wp.n_blocks = grid;
    hipLaunchKernelGGL(trs_worker_init_kernel, dim3(1), dim3(256), 0, e->sP, R->dc, (u64)start);
    if (e->rp.depth) hipLaunchKernelGGL(trs_worker_kernel<true>, dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
    else hipLaunchKernelGGL(trs_worker_kernel<false>, dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
    RCHK(hipGetLastError());
    R->running = true;
    return TRS_OK;
}

int worker_error(trs_env* e)
{
    const uint64_t err = host_load(&e->res->mb->error);
    if (!err) return TRS_OK;
    static const char* const what[] = {"", "waiting for a post", "camera ring back-pressure", "waiting for the physics team", "", "", "", "", "",
                                       "dynamic LDS segment not at offset 0"};
    const unsigned code = (unsigned)(err >> 32);
    return trs_internal_fail(TRS_ERR_DEVICE, std::string("resident worker gave up (") + (code < 10 ? what[code] : "?") + ") in workgroup " +
                                                 std::to_string((unsigned)err));
}

// the worker has said it leaves: wait for the kernel, restart it if posts raced with its exit
int handle_exit(trs_env* e)
{
    Resident* R = e->res;
    RCHK(hipStreamSynchronize(e->sP));
    R->running = false;
    int rc = worker_error(e);
    if (rc) return rc;
    const uint64_t consumed = host_load(&R->mb->consumed), posted = host_load(&R->mb->posted);
    if (consumed > R->seen_done) R->seen_done = consumed;   // the kernel has ended: everything it consumed is complete
    if (consumed < posted) return worker_launch(e, consumed);
    return TRS_OK;
}

int wait_done(trs_env* e, uint64_t s)
{
    Resident* R = e->res;
    if (s < R->seen_done) return TRS_OK;
    Mailbox* mb = R->mb;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (host_load(&mb->done[s & (kSlots - 1)]) >= s + 1) { R->seen_done = s + 1; return TRS_OK; }
        if ((spins & 63u) == 63u) {
            if (R->running && host_load(&mb->exited)) {
                int rc = handle_exit(e);
                if (rc) return rc;
                if (s < R->seen_done) return TRS_OK;
            } else if (!R->running) {
                if (host_load(&mb->posted) > R->seen_done) { int rc = worker_launch(e, R->seen_done); if (rc) return rc; }
            }
            if (host_load(&mb->error)) { (void)handle_exit(e); return worker_error(e); }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10))
                return trs_internal_fail(TRS_ERR_DEVICE, "resident worker: step " + std::to_string(s) + " did not complete within 10 s");
        }
        __builtin_ia32_pause();
    }
}

int ensure_resident(trs_env* e)
{
    Resident* R = e->res;
    if (R->mb) return TRS_OK;
    RCHK(hipHostMalloc((void**)&R->mb, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(R->mb, 0, sizeof(Mailbox));
    RCHK(hipMalloc((void**)&R->dc, sizeof(DevCtl)));
    RCHK(hipStreamCreateWithFlags(&R->sC, hipStreamNonBlocking));
    R->hctl_slot = ((size_t)e->n * 13 + 63) & ~(size_t)63;
    RCHK(hipHostMalloc((void**)&R->hctl, R->hctl_slot * kSlots, hipHostMallocMapped | hipHostMallocCoherent));
    return TRS_OK;
}

}  // namespace

namespace trsim {

bool resident_on(const trs_env* e) { return e && e->res && e->res->enabled; }

int resident_post(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, int n, size_t stride)
{
    Resident* R = e->res;
    Mailbox* mb = R->mb;
    for (int k = 0; k < n; ++k) {
        const uint64_t s = e->step_count;
        if (s >= R->base + kSlots) { int rc = wait_done(e, s - kSlots); if (rc) return rc; }   // ring slot, counters and done flag of s % 8 are free
        WEntry en{};
        const size_t off = (size_t)k * stride;
        en.steer = st ? st + off : nullptr; en.thr = th ? th + off : nullptr; en.brk = br ? br + off : nullptr;
        en.reset = k == 0 ? rs : nullptr; en.synth = synth ? 1u : 0u;
        std::memcpy(&mb->ring[s & (kSlots - 1)], &en, sizeof en);
        host_store(&mb->posted, s + 1);
        std::atomic_thread_fence(std::memory_order_seq_cst);     // the post is visible before `exited` is read
        e->step_count = s + 1;
        if (!R->running) { int rc = worker_launch(e, s); if (rc) return rc; }
        else if (host_load(&mb->exited)) { int rc = handle_exit(e); if (rc) return rc; }
    }
    return TRS_OK;
}

int resident_wait(trs_env* e)
{
    Resident* R = e->res;
    if (!R || e->step_count <= R->seen_done || e->step_count <= R->base) return TRS_OK;
    return wait_done(e, e->step_count - 1);
}

int resident_quiesce(trs_env* e)
{
    Resident* R = e->res;
    if (!R || !R->mb) return TRS_OK;
    int rc = TRS_OK;
    for (int guard = 0; R->running && guard < 4; ++guard) {
        __atomic_store_n(&R->mb->close, 1u, __ATOMIC_RELEASE);
        rc = handle_exit(e);                                 // waits for the kernel; relaunches (with `close` cleared) if posts raced
        if (rc) break;
    }
    if (!rc && R->running) rc = trs_internal_fail(TRS_ERR_DEVICE, "resident worker did not leave");
    R->base = R->seen_done = e->step_count;
    return rc;
}

void resident_destroy(trs_env* e)
{
    Resident* R = e->res;
    if (!R) return;
    if (R->running) { __atomic_store_n(&R->mb->close, 1u, __ATOMIC_RELEASE); (void)hipStreamSynchronize(e->sP); }
    if (R->mb) (void)hipHostFree(R->mb);
    if (R->hctl) (void)hipHostFree(R->hctl);
    (void)hipFree(R->dc);
    if (R->sC) (void)hipStreamDestroy(R->sC);
    delete R;
    e->res = nullptr;
}

hipStream_t resident_copy_stream(trs_env* e) { return (e->res && e->res->running) ? e->res->sC : e->sP; }

// controls handed over as host arrays: into this step's slot of the pinned staging buffer, which the device reads over PCIe
int resident_post_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps)
{
    Resident* R = e->res;
    const uint64_t s = e->step_count;
    if (s >= R->base + kSlots) { int rc = wait_done(e, s - kSlots); if (rc) return rc; }     // the staging slot is free as well
    unsigned char* slot = R->hctl + (s & (kSlots - 1)) * R->hctl_slot;
    const size_t n = (size_t)e->n;
    float* f = reinterpret_cast<float*>(slot);
    std::memcpy(f, h_st, n * 4); std::memcpy(f + n, h_th, n * 4);
    if (h_br) std::memcpy(f + 2 * n, h_br, n * 4);
    uint8_t* rsb = slot + n * 12;
    if (h_rs) std::memcpy(rsb, h_rs, n);
    // held controls: every step of the call reads the same slot, so the slot must outlive them — post them one by one and
    // keep the slot until the last is done (n_steps > kSlots would wrap onto it: copy again per step instead)
    for (int k = 0; k < n_steps; ++k) {
        if (k > 0) {
            const uint64_t sk = e->step_count;
            if (sk >= R->base + kSlots) { int rc = wait_done(e, sk - kSlots); if (rc) return rc; }
            unsigned char* sl = R->hctl + (sk & (kSlots - 1)) * R->hctl_slot;
            if (sl != slot) std::memcpy(sl, slot, n * 12);
            slot = sl; f = reinterpret_cast<float*>(slot); rsb = slot + n * 12;
        }
        int rc = resident_post(e, f, f + n, h_br ? f + 2 * n : nullptr, (k == 0 && h_rs) ? rsb : nullptr, 0, 1, 0);
        if (rc) return rc;
    }
    return TRS_OK;
}

}  // namespace trsim

TRS_EXPORT int trs_set_step_mode(trs_env* e, int mode, int idle_us)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (mode != TRS_STEP_LAUNCH && mode != TRS_STEP_RESIDENT) return trs_internal_fail(TRS_ERR_ARG, "mode must be TRS_STEP_LAUNCH or TRS_STEP_RESIDENT");
    RCHK(hipSetDevice(e->device));
    if (mode == TRS_STEP_LAUNCH) {
        if (!e->res) return TRS_OK;
        int rc = resident_quiesce(e);
        e->res->enabled = false;
        return rc;
    }
    if (!e->cfg.render) return trs_internal_fail(TRS_ERR_STATE, "resident mode needs a camera (cfg.render == 1): the physics-only kernel already keeps K steps in one launch");
    if (!e->track_loaded) return trs_internal_fail(TRS_ERR_STATE, "no track loaded");
    if (!e->res) e->res = new (std::nothrow) Resident();
    if (!e->res) return trs_internal_fail(TRS_ERR_NOMEM, "out of memory");
    int rc = ensure_resident(e);
    if (rc) return rc;
    Resident* R = e->res;
    R->lds_off_ctl = (e->lds_step + 15) & ~15;
    R->lds_bytes = (int)(R->lds_off_ctl + wlds_bytes(e->pp.envs_per_wg) + 16);
    if (R->lds_bytes > 160 * 1024) return trs_internal_fail(TRS_ERR_LIMIT, "too many envs per workgroup for the resident worker's LDS state");
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (idle_us > 0) R->idle_us = (unsigned)std::min(idle_us, 1000000);
    if (!R->enabled) { R->base = R->seen_done = e->step_count; host_store(&R->mb->posted, e->step_count); }
    R->enabled = true;
    return TRS_OK;
}
