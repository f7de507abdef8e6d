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
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <mutex>
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

struct WEntry {                 // one posted step: ONE 64-B line, so the dispatcher learns of a post and gets it in one PCIe read
    uint64_t seq_lo;            // step index + 1: the tag of the line's FIRST 32-byte half, written by the host after steer / thr / brk
    const float* steer; const float* thr; const float* brk;
    const uint8_t* reset;
    uint32_t synth, pad0;
    uint64_t seq;               // step index + 1, written LAST: the tag of the second half.  The line is a valid post for step s iff
                                // seq_lo == seq == s + 1 — should the device's 64-byte read ever be served as two 32-byte requests, a
                                // stale half cannot pair with a fresh one (each half carries its own tag, written after its payload)
    uint64_t pad1;
};
static_assert(sizeof(WEntry) == 64, "one entry per 64-B line");
constexpr int kTagLo = 0, kTagHi = 6;   // u64 word indices of the two tags within a line
static_assert(offsetof(WEntry, seq_lo) == 8 * kTagLo && offsetof(WEntry, seq) == 8 * kTagHi && offsetof(WEntry, reset) == 32, "tag placement");

struct Mailbox {                // pinned host memory the device reads and writes over PCIe
    alignas(64) uint64_t close;             // host -> device: leave once everything posted is done
    uint64_t posted;                        // host bookkeeping: steps [0, posted) have been posted (the device reads the entries' tags)
    alignas(64) uint64_t exited;            // device -> host: the dispatcher has decided to leave (kExitNormal), or it found the launch NOT co-resident
                                            //   (kExitNotCoresident: another worker — of another process — holds part of the CUs; nothing was consumed)
    uint64_t consumed;                      //   ... and every step below this index is processed by the time the kernel ends
    uint64_t error;                         //   non-zero: a bounded wait gave up (code << 32 | block)
    uint64_t started;                       //   1 = every workgroup of this launch has reported in: the launch is co-resident and serves posts
    alignas(64) uint64_t done[kSlots];      // device -> host: done[s % 8] = s + 1 when step s is complete in memory
    alignas(64) WEntry ring[kSlots];        // host -> device
};

struct DevCtl {                 // device memory; touched only by sc1 accesses and atomics
    alignas(64) unsigned long long word;    // posted count | kExitBit | kAbortBit, republished by the dispatcher
    alignas(64) WEntry ring[kSlots];
    alignas(64) unsigned arrive[kSlots][8][16];   // [slot][blockIdx % 8]: one 64-B line each
    alignas(64) unsigned top[kSlots][16];
    alignas(64) unsigned checkin[16];       // workgroups of this launch that are on a CU with their tables staged (the dispatcher waits for n_blocks)
    alignas(64) unsigned decision[16];      // 0 = open, kGo = the whole grid is on the GPU: posts are served, kNoGo = it is not: every workgroup leaves (set ONCE, by compare-and-swap)
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
    unsigned long long checkin_ticks;                       // how long the dispatcher waits for every workgroup of the launch to report in
    int lds_off_phys, lds_off_ctl, n_blocks;
    int lds_off_hill;                                       // a track with elevation: two per-env row tables (hill_table_bytes each) + the raster team's barrier counter
    FParams fp;                                             // DYN instantiation: ImgPreprocessing with dynamic brightness behind the rasteriser (trs_set_frame_filter)
};
#ifndef TRS_RESIDENT_DIAG
#define TRS_RESIDENT_DIAG 0   /* timing-only diagnostic builds (-DTRS_RESIDENT_DIAG=bits), never shipped, like TRS_ABLATE: 1 = arrive without the counted wait (WRONG completion flags), 2 = no telemetry stores, 4 = clock probe of one raster wave into stats[40..44] */
#endif
constexpr int kDiag = TRS_RESIDENT_DIAG;

constexpr unsigned kDefaultLifeUs = 50000;
#ifndef TRS_PHYS_PRIO_MAX_ENVS
#define TRS_PHYS_PRIO_MAX_ENVS 1   /* envs per workgroup up to which a physics wave integrates at raised priority (0 = never: A/B builds) */
#endif
constexpr unsigned kRetryMs0 = 100;

struct Resident {
    bool enabled = false, running = false;
    bool broken = false;                 // a worker gave up (bounded wait): some workgroups may have taken a step others did not — the env
                                         // state is undefined until the track is loaded again; every resident call fails meanwhile
    Mailbox* mb = nullptr;
    DevCtl* dc = nullptr;
    hipStream_t sC = nullptr;            // copies while the worker owns the handle's stream
    uint64_t base = 0;                   // steps [base, step_count) were handed to the worker since the last quiesce
    uint64_t seen_done = 0;              // every step below this index has been observed complete
    bool launched = false;               // steps were LAUNCHED on the handle's stream since the last wait (trs_step_pilot in resident mode): no
                                         // completion flag will ever be written for them — the stream is what to wait for
    bool fell_back = false;              // a launch of the worker was found not co-resident (another process's worker on the GPU): the handle went
                                         // back to TRS_STEP_LAUNCH by itself (trs_last_error() says so); trs_set_step_mode selects resident mode again
    bool dc_ready = false; uint64_t dc_ready_for = 0;   // the device control block has been zeroed and set up for a launch that starts at this step (handle_exit does it for the NEXT launch)
    std::chrono::steady_clock::time_point t_launch{};   // when the running worker was launched (the host gives up on a launch that never reports in)
    std::chrono::steady_clock::time_point t_fallback{}; // when the handle went back to launches; resident mode is tried again retry_ms later
    unsigned retry_ms = kRetryMs0;                      //   ... doubling up to 2 s while the GPU stays shared
    unsigned idle_us = 2000;
    unsigned life_us = kDefaultLifeUs;   // a worker leaves after this long whatever happens and the next post (or the one that raced) starts a new one
                                         // (trs_resident_debug_lifetime: tests force many generations).  50 ms since round 5 (0.5 s before): the gap
                                         // between two generations is where ANOTHER process's kernels get CUs — its launches, or its own worker,
                                         // which then holds the GPU for its 50 ms: processes that share a GPU take turns at the reference's tick
                                         // rate (20 Hz, car_templates/manage.py:38) or better, and a worker restart (~35 us) per 50 ms costs 0.1 %
    unsigned char* hctl = nullptr;       // pinned staging for host-array controls: [kSlots] x (3 float[n] + uint8[n])
    size_t hctl_slot = 0;
    int lds_bytes = 0, lds_off_ctl = 0, lds_off_dyn = 0, lds_off_hill = 0;
    long long pw_capacity_envs = 0;      // physics-only handles: envs whose workgroups the GPU holds at once (worker_fits)
    int pw_capacity_lds = -1;            //   ... asked for this LDS need
};

}  // namespace trsim

namespace {

using namespace trsim;
using u64 = unsigned long long;

constexpr u64 kExitBit = 1ull << 63, kAbortBit = 1ull << 62, kCountMask = kAbortBit - 1;
constexpr u64 kExitNormal = 1ull, kExitNotCoresident = 2ull;     // Mailbox::exited
constexpr u64 kCloseLeave = 1ull, kCloseCancel = 2ull;            // Mailbox::close: leave once everything posted is done / the host has given up on this launch
constexpr int kFellBack = trsim::kResidentFellBack;               // internal return code (> 0: not an error): the handle has just gone back to launch mode

// ---- one resident worker per GPU (per process) -----------------------------------------------------------------------
// A worker needs EVERY workgroup of its grid on a CU at once (the completion flags count all of them, and a workgroup that is there never
// leaves while it waits for posts), one slot with most of the LDS per CU.  Two workers on one GPU would each hold a part of the CUs and wait
// for the rest (seen in round 4: gpurun_out/r04_full_2.log, both ranks "resident worker gave up" after the 2 s safety).  So the library keeps
// one owner per device: worker_launch asks the other handle's worker to leave first (close -> it finishes what was posted and exits -> the
// kernel has ended), which costs a worker start per alternation, never idle_us and never the safety.  The lock also makes the host side of a
// handle's resident state safe against that eviction coming from another handle's thread.  What the lock cannot see is another PROCESS: that
// case is caught on the device (dispatcher_checkin) and by the host's launch deadline (wait_done), and ends in launch mode, not in a hang.
constexpr int kMaxDevices = 64;
struct DeviceSlot { std::recursive_mutex mu; trs_env* owner = nullptr; };
DeviceSlot g_dev[kMaxDevices];
struct DevLock {
    std::unique_lock<std::recursive_mutex> lk;
    explicit DevLock(const trs_env* e) : lk(g_dev[(unsigned)e->device % kMaxDevices].mu) {}
};
inline DeviceSlot& slot_of(const trs_env* e) { return g_dev[(unsigned)e->device % kMaxDevices]; }

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
// wait until at most n of this wave's vector-memory operations are outstanding (n wave-uniform; the instruction takes an
// immediate).  A raster wave issues only stores in its steady state and stores are acknowledged in issue order, so behind M
// newer stores vmcnt(M) says "everything older has reached memory" without waiting for the newer ones.
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
    switch (__builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > 63 ? 63 : n))) {
#define W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23) W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45) W(46) W(47) W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61) W(62) W(63)
#undef W
    }
}

template <typename T>
__device__ __forceinline__ T sys_load_val(const T* p)
{
    return __hip_atomic_load((const __attribute__((address_space(1))) T*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// what a workgroup shares in LDS (behind the tables)
constexpr int kSlotWords = 20;  // one hand-off slot: camera parameters (4) | x y z yaw v speed cte seg epr epl sf last_return (12) | done | view pitch (a track with elevation) | 2 spare
struct WLds {
    u64* word;          // posted count | flags as last seen by this workgroup's leader
    u64* fwd;           // steps this workgroup has completely passed on (the forwarder's count; the dispatcher's idle clock reads it)
    int* arrive;        // [kSlots] raster waves whose stores of step s are in memory
    int* pprog;         // [epw] steps the physics team has finished per env (relative to the launch's first step)
    int* rread;         // [epw] raster-wave reads of hand-off slots per env (back-pressure)
    float* slot;        // [kCamDepth][epw][kSlotWords] physics -> raster: the pose to render AND the telemetry to write out
    float* lst;         // [epw][16] env state carried from step to step
};

__host__ __device__ inline size_t wlds_slot_off(int epw) { return (64 + (size_t)epw * 8 + 15) & ~(size_t)15; }

__device__ __forceinline__ WLds wlds_of(unsigned char* base, int epw)
{
    WLds l;
    l.word = reinterpret_cast<u64*>(base);
    l.fwd = reinterpret_cast<u64*>(base + 8);
    l.arrive = reinterpret_cast<int*>(base + 16);
    l.pprog = reinterpret_cast<int*>(base + 64);
    l.rread = l.pprog + epw;
    l.slot = reinterpret_cast<float*>(base + wlds_slot_off(epw));
    l.lst = l.slot + (size_t)kCamDepth * epw * kSlotWords;
    return l;
}

__host__ __device__ inline size_t wlds_bytes(int epw)
{
    return wlds_slot_off(epw) + (size_t)kCamDepth * epw * kSlotWords * 4 + (size_t)epw * 64;
}

// a bounded wait gave up: tell the host, every workgroup (device word) and this workgroup (LDS word)
__device__ __forceinline__ void worker_abort(const WParams& wp, const WLds& l, unsigned code)
{
    sys_store64(&wp.mb->error, ((u64)code << 32) | (u64)blockIdx.x);
    __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct Leader { u64 known, t_last, t_start; };

// dispatcher = physics wave 0 of workgroup 0, and nothing else (that workgroup's envs go to its other three physics waves): a
// poll is a PCIe round trip of several microseconds, which must not sit in front of any env's integration.  One poll = two
// wave instructions in flight together: the 64 lanes read the WHOLE ring (8 lines of 64 B; lane k = word k % 8 of slot k / 8),
// then the close word.  Every slot whose tag continues the sequence is a post: the lines go on to the device ring (sc1),
// then the count is republished (device word for the other workgroups' leaders, LDS word for this workgroup).
__device__ __forceinline__ u64 lane_u64(u64 v, int k)
{
    return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), k) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)v, k);
}

__device__ __forceinline__ int dispatcher_take(const WParams& wp, const WLds& l, Leader& L, int lane, bool& close_req)
{
    const u64 v = sys_load64(reinterpret_cast<const u64*>(&wp.mb->ring[0]) + lane);
    const u64 c = sys_load64(&wp.mb->close);
    close_req = c != 0ull;
    int fresh = 0;
    for (; fresh < kSlots; ++fresh) {
        const int slot = (int)((L.known + (u64)fresh) & (kSlots - 1));
        const int hi = slot * 8 + kTagHi, lo = slot * 8 + kTagLo;   // both halves of the line carry the tag
        const u64 tag = ((u64)(unsigned)__shfl((int)(v >> 32), hi, 64) << 32) | (u64)(unsigned)__shfl((int)v, hi, 64);
        const u64 tag_lo = ((u64)(unsigned)__shfl((int)(v >> 32), lo, 64) << 32) | (u64)(unsigned)__shfl((int)v, lo, 64);
        if (tag != L.known + (u64)fresh + 1 || tag_lo != tag) break;
    }
    if (fresh == 0) return 0;
    const int rel = (int)(((u64)(lane >> 3) - L.known) & (kSlots - 1));   // this lane's slot is the rel-th behind `known`
    if (rel < fresh) agent_store64(reinterpret_cast<u64*>(&wp.dc->ring[0]) + lane, v);
    drain_vmem();                                            // the entries are in the device ring before the count says so
    L.known += (u64)fresh;
    if (lane == 0) {                                         // max, not store: an abort bit (1 << 62: above every count) stays set, in the
        __hip_atomic_fetch_max(&wp.dc->word, L.known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // device word another workgroup OR'ed it into
        __hip_atomic_fetch_max(l.word, L.known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // and in the LDS word a wave of THIS workgroup did
    }
    return fresh;
}

// The dispatcher's whole life.  It leaves — and with it, after the steps below the final count, every wave of the launch — when
// the host asks (close), when nothing was posted for idle_us although everything published is rendered, or when the launch's
// lifetime is spent (the host starts a new worker at its next post).
// the EXIT bit goes out once, with the FINAL count: every workgroup finishes the steps below it and leaves
__device__ __forceinline__ void dispatcher_publish_exit(const WParams& wp, const WLds& l, u64 count, int lane)
{
    if (lane == 0) {
        const u64 w = count | kExitBit;
        // EXIT (1 << 63) outranks ABORT (1 << 62) in a max: an abort that arrived just before is carried over by hand
        const u64 old_g = __hip_atomic_fetch_max(&wp.dc->word, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old_g & kAbortBit) __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 old_l = __hip_atomic_fetch_max(l.word, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((old_l | old_g) & kAbortBit) __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    drain_vmem();
}

// a workgroup is on its CU with its tables staged (called once per workgroup, behind the prologue's barrier)
__device__ __forceinline__ void worker_checkin(const WParams& wp)
{
    __hip_atomic_fetch_add(&wp.dc->checkin[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Before the first post is taken: is the WHOLE grid on the GPU?  A step completes when every workgroup has arrived for it, and the workgroups
// that are on a CU stay there while they wait for posts — a launch of which only a part found room (the rest of the CUs' LDS is held by the
// worker of another process: this process's own handles hand the GPU over in worker_launch) would never complete a step and never make room
// for its own missing workgroups (round 4, gpurun_out/r04_full_2.log: two ranks on one GPU, both "gave up (waiting for a post)" after the 2 s
// safety — in workgroups 212 and 38, whose dispatchers in workgroup 0 had found no CU at all: the XCDs place their shares of a grid
// independently).  The workgroups of a launch that has the GPU to itself report in within microseconds of each other (dispatch + ~124 KB of
// LDS-DMA).  One word decides, once, by compare-and-swap: the dispatcher sets kGo when all n_blocks have reported in; the dispatcher after
// checkin_ticks (2 ms) — or ANY workgroup's leader after twice that, should workgroup 0 be among the missing — sets kNoGo.  On kNoGo nothing
// has been consumed: the winner tells the host (consumed = start, exited = kExitNotCoresident) and every workgroup — those on the GPU now and
// the ones that get a CU later — leaves without taking a step.  The host then runs the posted steps by launches (fall_back_to_launches).  The
// dispatcher takes the same exit when the host has given up on the launch first (close = kCloseCancel: the kernel sat in the queue behind
// another process's worker for 250 ms).
constexpr unsigned kGo = 1u, kNoGo = 2u;

__device__ __forceinline__ unsigned decision_load(const WParams& wp) { return __hip_atomic_load(&wp.dc->decision[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// try to close the decision with `want`; returns what the decision IS afterwards (wave-uniform: lane 0 swaps, the wave reads it back)
__device__ __forceinline__ unsigned decision_close(const WParams& wp, unsigned want, int lane, bool& won)
{
    unsigned seen = 0u;
    if (lane == 0) {
        unsigned expected = 0u;
        const bool ok = __hip_atomic_compare_exchange_strong(&wp.dc->decision[0], &expected, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = ok ? (want | 0x100u) : expected;
    }
    seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
    won = (seen & 0x100u) != 0u;
    return seen & 0xffu;
}

// the launch is not co-resident: tell the host (the winner of the decision only) and this workgroup
__device__ __forceinline__ void leave_not_coresident(const WParams& wp, const WLds& l, int lane, bool won)
{
    if (won) {
        if (lane == 0) sys_store64(&wp.mb->consumed, wp.start);
        drain_vmem();                                        // `consumed` is in host memory before `exited` says the launch is over
        if (lane == 0) sys_store64(&wp.mb->exited, kExitNotCoresident);
    }
    dispatcher_publish_exit(wp, l, wp.start, lane);          // EXIT | start: device word (leaders that are past their own check) and this workgroup's LDS word
}

// 1 = serve posts, 0 = leave.
__device__ __forceinline__ int dispatcher_checkin(const WParams& wp, const WLds& l, int lane)
{
    const u64 t0 = (u64)wall_clock64();
    bool cancel = sys_load64(&wp.mb->close) == kCloseCancel, won = false;
    for (unsigned spins = 0; !cancel; ++spins) {
        if (decision_load(wp) == kNoGo) { leave_not_coresident(wp, l, lane, false); return 0; }
        if (__hip_atomic_load(&wp.dc->checkin[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)wp.n_blocks) {
            if (decision_close(wp, kGo, lane, won) != kGo) { leave_not_coresident(wp, l, lane, false); return 0; }
            if (lane == 0) sys_store64(&wp.mb->started, 1ull);
            return 1;
        }
        if (agent_load64(&wp.dc->word) & kAbortBit) {        // a workgroup refused to run (LDS segment not at offset 0): its error word is set
            if (lane == 0) __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return 0;
        }
        if ((u64)wall_clock64() - t0 > wp.checkin_ticks) break;
        if ((spins & 63u) == 63u) cancel = sys_load64(&wp.mb->close) == kCloseCancel;   // (a PCIe read: sparse)
        __builtin_amdgcn_s_sleep(4);
    }
    const unsigned d = decision_close(wp, kNoGo, lane, won);
    if (d == kGo) { if (lane == 0) sys_store64(&wp.mb->started, 1ull); return 1; }   // (cannot happen: only this wave sets kGo)
    leave_not_coresident(wp, l, lane, won);
    return 0;
}

// A leader (one wave per workgroup other than workgroup 0) waits for the decision before it serves its workgroup.  1 = go, 0 = leave.
__device__ __forceinline__ int leader_wait_decision(const WParams& wp, const WLds& l, int lane)
{
    const u64 t0 = (u64)wall_clock64();
    for (;;) {
        unsigned d = decision_load(wp);
        if (d == 0u && (u64)wall_clock64() - t0 > 2ull * wp.checkin_ticks) {
            bool won = false;
            d = decision_close(wp, kNoGo, lane, won);        // the dispatcher itself is missing
            if (d == kNoGo) { leave_not_coresident(wp, l, lane, won); return 0; }
        }
        if (d == kGo) return 1;
        if (d == kNoGo) { leave_not_coresident(wp, l, lane, false); return 0; }
        if (agent_load64(&wp.dc->word) & kAbortBit) {
            if (lane == 0) __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return 0;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

__device__ __forceinline__ void dispatcher_run(const WParams& wp, const WLds& l, int lane)
{
    if (!dispatcher_checkin(wp, l, lane)) return;
    Leader L{wp.start, (u64)wall_clock64(), (u64)wall_clock64()};
    for (;;) {
        if ((lds_load64(l.word) | agent_load64(&wp.dc->word)) & kAbortBit) {   // some wave gave up: this workgroup learns of it here
            if (lane == 0) __hip_atomic_fetch_or(l.word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
        const u64 now = (u64)wall_clock64();
        bool close_req = false;
        const int fresh = dispatcher_take(wp, l, L, lane, close_req);
        const bool busy = (u64)lds_load64(l.fwd) < L.known;           // this workgroup has not finished everything published
        if (fresh || busy) L.t_last = now;
        const bool leave = (!fresh && (close_req || now - L.t_last > wp.idle_ticks)) || now - L.t_start > wp.life_ticks;
        if (!leave) { if (!fresh) __builtin_amdgcn_s_sleep(4); continue; }      // (a poll is a PCIe round trip anyway; 16 until late round 4: ~0.25 us more until a post is seen)
        if (lane == 0) sys_store64(&wp.mb->exited, kExitNormal);
        drain_vmem();                                        // `exited` is in host memory before the last look at the ring
        (void)dispatcher_take(wp, l, L, lane, close_req);    // whatever was posted before that look is still served by this launch
        if (lane == 0) sys_store64(&wp.mb->consumed, L.known);
        dispatcher_publish_exit(wp, l, L.known, lane);
        return;
    }
}

// A raster wave whose stores of step s are in memory says so in LDS — and nothing more: a returning global atomic would make
// it wait for ALL its outstanding stores, the newer steps' too.
__device__ __forceinline__ void raster_arrive(const WLds& l, u64 s, int lane)
{
    if (lane == 0) __hip_atomic_fetch_add(&l.arrive[s & (kSlots - 1)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The forwarder (the workgroup's last physics wave: it never stores to global memory, so it has nothing to drain) watches the
// LDS counters in its idle moments.  All eight raster waves in: it arrives for the workgroup on a device counter sharded by
// blockIdx % 8; the last workgroup of a shard arrives on the top counter; the last shard tells the host.
__device__ __forceinline__ void forward_arrivals(const WParams& wp, const WLds& l, u64& fwd, int lane, int want = kRasterThreads / 64)
{
    const int slot = (int)(fwd & (kSlots - 1));
    if (lds_load32(&l.arrive[slot]) != want) return;
    if (lane == 0) {
        lds_store32(&l.arrive[slot], 0);
        const int shard = (int)(blockIdx.x & 7u), nshards = min(8, wp.n_blocks);
        const unsigned members = (unsigned)((wp.n_blocks - shard + 7) / 8);
        // The counters only ever count up (zeroed by trs_worker_init_kernel in front of every launch): step fwd is the uses-th use of its slot in
        // this launch, so the last arrival of a shard sees members * uses - 1 and the last shard nshards * uses - 1 (modulo 2^32 like the counters).
        // Until late round 4 the last arrival reset the counter and drained that store in front of the done flag: a device round trip on the
        // consumer's critical path of every lock-step tick.
        const unsigned uses = (unsigned)((fwd - wp.start) / (u64)kSlots) + 1u;
        unsigned* a = &wp.dc->arrive[slot][shard][0];
        if (__hip_atomic_fetch_add(a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members * uses - 1u) {
            unsigned* t = &wp.dc->top[slot][0];
            if (__hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nshards * uses - 1u)
                sys_store64(&wp.mb->done[slot], fwd + 1);
        }
    }
    fwd += 1;
    if (lane == 0) lds_store64(l.fwd, fwd);
}

// What a waiting physics wave does on the side: the leader keeps the workgroup's LDS word fresh (workgroup 0's also talks to
// the host), the forwarder passes completed steps on.
struct Duties { bool leader, forwarder; u64 fwd; int want = kRasterThreads / 64; };   // want: arrivals that complete a step in this workgroup

// 1 = step s is posted (go), 0 = leave.
__device__ __forceinline__ int wait_posted(const WParams& wp, const WLds& l, Duties& D, u64 s, int lane)
{
    u64 t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        if (D.forwarder) forward_arrivals(wp, l, D.fwd, lane, D.want);
        const u64 w = lds_load64(l.word);
        if (w & kAbortBit) return 0;
        if ((w & kCountMask) > s) return 1;
        if (w & kExitBit) return 0;
        if (D.leader) {                                      // (workgroup 0 has no leader: its dispatcher writes the LDS word itself)
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
__device__ __forceinline__ int wait_lds_ge(const WParams& wp, const WLds& l, Duties* D, const int* ctr, int want, unsigned code, int lane)
{
    u64 t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        if (lds_load32(ctr) >= want) return 1;
        if (D && D->forwarder) forward_arrivals(wp, l, D->fwd, lane, D->want);
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (lds_load64(l.word) & kAbortBit) return 0;
            const u64 now = (u64)wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > wp.safety_ticks) { worker_abort(wp, l, code); return 0; }
        }
    }
}

template <bool DEPTH, bool DYN, bool HILLS = false>      // HILLS: a track with elevation (its own instantiations, see raster_hill_frame)
__global__ __launch_bounds__(kBlock) void trs_worker_kernel(const WParams wp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool raster_team = tid < kRasterThreads;
    const RParams& p = wp.ra;
    const PParams& P = wp.ph;
    const int epw = p.envs_per_wg;
    const WLds l = wlds_of(smem + wp.lds_off_ctl, epw);
    const int e_begin = blockIdx.x * epw;
    const int n_loc = min(e_begin + epw, p.n_envs) - e_begin;
    const int pw = wave - kRasterThreads / 64;
    if ((unsigned)(uintptr_t)smem != 0u) {                   // the map addressing assumes LDS offset 0
        if (tid == 0) {
            atomicAdd(&p.stats[2], 1ull);
            sys_store64(p.fault, 1ull);
            sys_store64(&wp.mb->error, (9ull << 32) | (u64)blockIdx.x);
            __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) { sys_store64(&wp.mb->consumed, wp.start); sys_store64(&wp.mb->exited, 1ull); }
        }
        return;
    }
    // ---- once per launch: tables by LDS-DMA, control block, env state ----
    if (raster_team) stage_lds_dma(p.blob, p.blob_bytes, 0u, wave, kRasterThreads / 64, lane);
    else stage_lds_dma(P.blob, P.blob_bytes, (unsigned)wp.lds_off_phys, pw, kPhysWaves, lane);
    if (tid == 0) { lds_store64(l.word, wp.start); lds_store64(l.fwd, wp.start); }
    if (tid < kSlots) l.arrive[tid] = 0;
    for (int j = tid; j < 2 * epw; j += kBlock) l.pprog[j] = 0;          // pprog | rread
    int* const hbar = reinterpret_cast<int*>(smem + wp.lds_off_hill + hill_batch(p.H) * hill_table_bytes(p.H));   // (a track with elevation: the row tables' team-barrier counter)
    if (HILLS && tid == 0) *hbar = 0;
    if constexpr (DYN) {
        if (tid < 32) reinterpret_cast<int*>(smem + wp.fp.lds_off + kDynBatch * p.H * 16)[tid] = 0;   // channel sums, team-barrier counter
        dyn_stage_tables(smem, wp.fp, p.H, tid, kBlock, reinterpret_cast<const uint32_t*>(p.blob + p.off_pal));                   // the filter's tables, once per launch: the raster waves' steady state issues no loads
    }
    if (!raster_team)
        for (int j = pw; j < n_loc; j += kPhysWaves) {
            EnvRegs st;
            env_load(P, e_begin + j, st);
            const float lr = coherent_load(&P.last_return[e_begin + j]);
            if (lane == 0) {
                float* q = l.lst + (size_t)j * 16;
                q[0] = st.x; q[1] = st.y; q[2] = st.z; q[3] = st.yaw; q[4] = st.v; q[5] = st.sf; q[6] = st.epr;
                q[7] = __int_as_float(st.seg); q[8] = __int_as_float(st.epl); q[9] = __int_as_float(st.done); q[10] = __int_as_float(st.pend);
                q[11] = lr;
            }
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) worker_checkin(wp);

    if (!raster_team) {
        // ---- physics team: LDS in, LDS out.  It reads its controls (system-scope loads) and hands the new pose AND the step's
        // telemetry to the raster team through the slot ring; it never stores to global memory, so no step of it ever waits for
        // a store acknowledgement and it can run ahead of the raster team as far as posts and the ring allow. ----
        const int first = blockIdx.x == 0 ? 1 : 0, nphys = kPhysWaves - first;   // workgroup 0: wave 0 is the dispatcher, three waves share the envs
        if (pw < first) { dispatcher_run(wp, l, lane); return; }
        Duties D{blockIdx.x != 0 && pw == 0, pw == kPhysWaves - 1, wp.start};
        if (D.leader && !leader_wait_decision(wp, l, lane)) return;   // (the other waves wait for a post: the LDS word says EXIT when the launch is called off)
        const unsigned char* const lphys = smem + wp.lds_off_phys;
        u64 s = wp.start;
        for (;; ++s) {
            const int r = (int)(s - wp.start);
            if (!wait_posted(wp, l, D, s, lane)) break;
            const u64* en = reinterpret_cast<const u64*>(&wp.dc->ring[s & (kSlots - 1)]);
            const u64 ev = lane < 6 ? agent_load64(en + lane) : 0ull;   // words 1..5 of the entry (WEntry): steer, thr, brk, reset, synth
            const float* const c_st = reinterpret_cast<const float*>(lane_u64(ev, 1));
            const float* const c_th = reinterpret_cast<const float*>(lane_u64(ev, 2));
            const float* const c_br = reinterpret_cast<const float*>(lane_u64(ev, 3));
            const uint8_t* const c_rs = reinterpret_cast<const uint8_t*>(lane_u64(ev, 4));
            const int synth = (int)(unsigned)lane_u64(ev, 5);
            for (int j = pw - first; j < n_loc; j += nphys) {
                const int e = e_begin + j;
                // back-pressure: slot r % kCamDepth is free once every raster wave has read step r - kCamDepth of this env
                if (r >= kCamDepth && !wait_lds_ge(wp, l, &D, &l.rread[j], (r - kCamDepth + 1) * (kRasterThreads / 64), 2u, lane)) return;
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
                float lr = q[11];
                const float epr_before = st.epr;
                StepOut o;
                // With ONE env per workgroup (256 envs or fewer per GPU) a step is as long as that env's integration beside the raster waves of its SIMD: there the
                // integration goes in front of them (256 envs 3.74 -> 3.52 us per step, 128 envs the same).  With two envs per workgroup it changes nothing (512 envs:
                // 4.97 against 5.01 us), with four the physics team runs ahead anyway and its priority only displaces raster waves (1024 envs: -5 %; profiles/r05_physics_chain.txt).
                if (n_loc <= TRS_PHYS_PRIO_MAX_ENVS) __builtin_amdgcn_s_setprio(3);
                env_advance<true, false>(P, lphys, e, st, (uint32_t)s, synth, steer, thr, brk, rin, lane, o,
                                         HILLS ? reinterpret_cast<const trsim::HillBlock*>(p.blob + trsim::hill_block_offset(p.blob_bytes)) : nullptr);
                if (n_loc <= TRS_PHYS_PRIO_MAX_ENVS) __builtin_amdgcn_s_setprio(0);
                if (o.do_reset) lr = epr_before;
                if (lane == 0) {
                    q[0] = st.x; q[1] = st.y; q[2] = st.z; q[3] = st.yaw; q[4] = st.v; q[5] = st.sf; q[6] = st.epr;
                    q[7] = __int_as_float(st.seg); q[8] = __int_as_float(st.epl); q[9] = __int_as_float(st.done); q[10] = __int_as_float(st.pend);
                    q[11] = lr;
                    float* const sl = l.slot + ((size_t)(r & (kCamDepth - 1)) * epw + j) * kSlotWords;
                    *reinterpret_cast<float4*>(sl) = o.cam;
                    sl[4] = st.x; sl[5] = st.y; sl[6] = st.z; sl[7] = st.yaw; sl[8] = st.v; sl[9] = st.speed; sl[10] = st.cte;
                    sl[11] = __int_as_float(st.seg); sl[12] = st.epr; sl[13] = __int_as_float(st.epl); sl[14] = st.sf; sl[15] = lr;
                    sl[16] = __int_as_float(st.done);
                    if constexpr (HILLS) sl[17] = o.pitch;
                    if (o.is_done) atomicAdd(&P.stats[0], 1ull);
                    if (o.do_reset) atomicAdd(&P.stats[1], 1ull);
                    drain_lds();                              // the slot is in LDS before the counter moves
                    lds_store32(&l.pprog[j], r + 1);
                }
            }
        }
        // leaving: the forwarder stays until every step this launch took (s of them) has been passed on
        if (D.forwarder) {
            u64 t0 = 0;
            for (unsigned spins = 0; D.fwd < s; ++spins) {
                forward_arrivals(wp, l, D.fwd, lane);
                if (lds_load64(l.word) & kAbortBit) break;
                __builtin_amdgcn_s_sleep(2);
                if ((spins & 1023u) == 1023u) {
                    const u64 now = (u64)wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > wp.safety_ticks) { worker_abort(wp, l, 4u); break; }
                }
            }
        }
        return;
    }

    // ---- raster team ----
    // A wave's stores of step s must be in memory before it arrives for s, and a store takes microseconds to be acknowledged —
    // about as long as a whole step of a small shard.  A wave that waited for its acknowledgements at the end of every step
    // would issue nothing meanwhile (all eight waves drain together: the CU's store path runs dry once per step).  So arrivals
    // LAG: with K steps of lag the wave arrives for step s - K after issuing step s's uniform rows, behind a counted wait
    // (vmcnt(N), N = the store instructions it has issued since the end of step s - K; a raster wave issues only stores and
    // they are acknowledged in order) — what it waits for was issued K steps ago.  K = 2, or 3 for very short steps, and never
    // more than the posts queued behind this step allow: a consumer that keeps two steps in
    // flight gets step s's completion after step s + 1's uniform rows, a lock-step consumer at once (drain, arrive).
    const RasterThread rth = raster_thread(p, smem, tid);
    int nu = 0, ng = 0;
    for (int v = rth.vstart; v < p.uni_rows; v += p.rows_per_pass) ++nu;
    for (int v = rth.vground; v < p.H; v += p.rows_per_pass) ++ng;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nu = max(nu, __shfl_xor(nu, off, 64)); ng = max(ng, __shfl_xor(ng, off, 64)); }   // instructions the WAVE issues per env
    nu *= DEPTH ? 2 : 1; ng *= DEPTH ? 2 : 1;
    int own = 0;                                              // envs whose telemetry this wave writes out (2 store instructions each)
    for (int j = wave; j < n_loc; j += kRasterThreads / 64) ++own;
    const int nuni = nu * n_loc, nstep = (nu + ng) * n_loc + 2 * own;
    // two steps of lag always: what the wave then waits for was issued a whole step ago (a count beyond 63 is clamped, which only
    // asks for what the 6-bit counter enforces anyway); three where a step is so short that even that is younger than a store's
    // round trip (small shards: 2 * nstep + nuni still fits the counter)
    const int lag = (2 * nstep + nuni <= 63) ? 3 : 2;
    // this lane's telemetry array (lanes 0..11: 4-byte arrays in slot-word order, lanes 12, 13: the byte arrays done, pending)
    unsigned char* optr = nullptr;
    {
        unsigned char* const tab[14] = {(unsigned char*)P.x, (unsigned char*)P.y, (unsigned char*)P.z, (unsigned char*)P.yaw, (unsigned char*)P.v,
                                        (unsigned char*)P.speed, (unsigned char*)P.cte, (unsigned char*)P.seg_idx, (unsigned char*)P.ep_return,
                                        (unsigned char*)P.ep_len, (unsigned char*)P.steer_filt, (unsigned char*)P.last_return,
                                        (unsigned char*)P.done, (unsigned char*)P.pending};
#pragma unroll
        for (int k = 0; k < 14; ++k) optr = lane == k ? tab[k] : optr;
    }
    Duties none{false, false, 0};
    u64 owed = wp.start;                                      // oldest step this wave has not arrived for
    // diagnostics (diag bit 4): where one raster wave (workgroup 7, wave 0) spends its clocks, into stats[40..45]
    const bool probe = (kDiag & 4) && blockIdx.x == 7 && wave == 0;
    auto now_clk = [&]() -> u64 { u64 t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; };
    u64 t_post = 0, t_pose = 0, t_vm = 0, t_all = 0, t_mark = probe ? now_clk() : 0;
    for (u64 s = wp.start;; ++s) {
        const int r = (int)(s - wp.start);
        if (owed < s && (lds_load64(l.word) & kCountMask) <= s) {   // nothing further posted: do not make the host wait
            drain_vmem();
            for (; owed < s; ++owed) raster_arrive(l, owed, lane);
        }
        const u64 tp0 = probe ? now_clk() : 0;
        if (!wait_posted(wp, l, none, s, lane)) {
            drain_vmem();
            for (; owed < s; ++owed) raster_arrive(l, owed, lane);
            if (probe && lane == 0) {
                t_all = now_clk() - t_mark;
                atomicAdd(&p.stats[40], t_post); atomicAdd(&p.stats[41], t_pose); atomicAdd(&p.stats[42], t_vm); atomicAdd(&p.stats[43], t_all);
                atomicAdd(&p.stats[44], (u64)r);
            }
            return;
        }
        if (probe) t_post += now_clk() - tp0;
        uint8_t* const img = (s & 1ull) ? wp.img1 : wp.img0;
        float* const dep = (s & 1ull) ? wp.dep1 : wp.dep0;
        // Order of the rows.  With more steps queued behind this one the physics team is ahead and nothing waits for a pose: the
        // frame of an env is written as ONE contiguous stream (its uniform rows, then its ground rows), env by env — two sweeps
        // over all envs cost 6 % of the HBM rate.  With nothing queued (a lock-step consumer is waiting for this very frame) the
        // uniform rows of ALL envs go first: they need no pose and are on their way while the physics team integrates.
        const u64 ahead = (lds_load64(l.word) & kCountMask) - 1 - s;
        const u64 keep = ahead + 1 < (u64)lag ? ahead + 1 : (u64)lag;       // the lag shrinks with the queue: a consumer that waits gets its flag early
        if constexpr (DYN) {
            // Dynamic-brightness frame filter: the envs of a step go through raster_dyn_batch four at a time (classify the brightness rows,
            // team-reduce the channel sums, filter the palettes, shade), exactly as trs_step_kernel's DYN instantiation does — the same
            // instructions on the same tables: bit-identical frames.  Arrivals owed are settled at the step's start: everything issued
            // since the end of step `owed` is then (s - owed - 1) whole steps.
            while (s - owed >= keep) {
                wait_vmcnt_le((int)(s - owed - 1) * nstep);
                raster_arrive(l, owed++, lane);
            }
            const int nbatch = (n_loc + kDynBatch - 1) / kDynBatch;
            for (int b0 = 0; b0 < n_loc; b0 += kDynBatch) {
                float4 cams[kDynBatch];
                unsigned tel = 0; int mine_j = -1;
#pragma unroll
                for (int bi = 0; bi < kDynBatch; ++bi) {
                    const int j = b0 + bi;
                    cams[bi] = make_float4(0.f, 0.f, 0.f, 1.f);
                    if (j < n_loc) {
                        if (!wait_lds_ge(wp, l, nullptr, &l.pprog[j], r + 1, 3u, lane)) return;
                        const float* const sl = l.slot + ((size_t)(r & (kCamDepth - 1)) * epw + j) * kSlotWords;
                        cams[bi] = *reinterpret_cast<const float4*>(sl);
                        if ((j % (kRasterThreads / 64)) == wave) { mine_j = j; tel = __float_as_uint(sl[4 + min(lane, 12)]); }
                        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(cams[bi].x), "v"(cams[bi].y), "v"(cams[bi].z), "v"(cams[bi].w), "v"(tel) : "memory");
                        if (lane == 0) __hip_atomic_fetch_add(&l.rread[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                u64 t0_bar = 0;
                // the team barriers inside are bounded like every other spin of the worker: abort bit of this workgroup, then the safety deadline
                auto bail = [&](bool first) -> bool {
                    if (lds_load64(l.word) & kAbortBit) return true;
                    const u64 now = (u64)wall_clock64();
                    if (first) { t0_bar = now; return false; }
                    if (now - t0_bar > wp.safety_ticks) { worker_abort(wp, l, 5u); return true; }
                    return false;
                };
                if (!raster_dyn_batch<DEPTH>(p, wp.fp, rth, smem, cams, min(kDynBatch, n_loc - b0), img, dep, e_begin + b0, r * nbatch + b0 / kDynBatch, tid, lane, bail)) return;
                if (mine_j >= 0 && !(kDiag & 2)) {                    // the step's telemetry of this wave's env of the batch (a wave owns at most one of four)
                    const size_t e = (size_t)(e_begin + mine_j);
                    if (lane < 12) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(uintptr_t)optr + e, tel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else if (lane < 14) __hip_atomic_store((__attribute__((address_space(1))) unsigned char*)(uintptr_t)optr + e, (unsigned char)(lane == 12 ? tel : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            continue;
        }
        if constexpr (HILLS) {
            // A track with elevation: the envs of a step go through in batches (trsim_device.hpp, hill_batch_build): poses, view pitches and telemetry of the batch from
            // the hand-off slots, the batch's row tables between two team barriers — bounded like every other wait of the worker (abort bit, safety deadline) — then
            // the row loop per env on its table.  Arrivals owed are settled at the step's start, as in the dynamic-brightness path: everything issued since the end of
            // step `owed` is then whole steps.
            while (s - owed >= keep) {
                wait_vmcnt_le((int)(s - owed - 1) * nstep);
                raster_arrive(l, owed++, lane);
            }
            const int HB = hill_batch(p.H);
            const int nbatch = (n_loc + HB - 1) / HB;
            for (int b0 = 0; b0 < n_loc; b0 += HB) {
                const int nb = min(HB, n_loc - b0);
                float4 cams[kHillBatchMax]; float Pv[kHillBatchMax];
                unsigned tel = 0; int mine_j = -1;
#pragma unroll
                for (int bi = 0; bi < kHillBatchMax; ++bi) {
                    cams[bi] = make_float4(0.f, 0.f, 0.f, 1.f); Pv[bi] = 0.f;
                    const int j = b0 + bi;
                    if (bi < nb) {
                        if (!wait_lds_ge(wp, l, nullptr, &l.pprog[j], r + 1, 3u, lane)) return;
                        const float* const sl = l.slot + ((size_t)(r & (kCamDepth - 1)) * epw + j) * kSlotWords;
                        cams[bi] = *reinterpret_cast<const float4*>(sl);
                        Pv[bi] = sl[17];
                        if ((j % (kRasterThreads / 64)) == wave) { mine_j = j; tel = __float_as_uint(sl[4 + min(lane, 12)]); }
                        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(cams[bi].x), "v"(cams[bi].y), "v"(cams[bi].z), "v"(cams[bi].w), "v"(tel), "v"(Pv[bi]) : "memory");
                        if (lane == 0) __hip_atomic_fetch_add(&l.rread[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                u64 t0_bar = 0;
                auto bail = [&](bool first) -> bool {
                    if (lds_load64(l.word) & kAbortBit) return true;
                    const u64 now = (u64)wall_clock64();
                    if (first) { t0_bar = now; return false; }
                    if (now - t0_bar > wp.safety_ticks) { worker_abort(wp, l, 5u); return true; }
                    return false;
                };
                const int done_before = (r * nbatch + b0 / HB) * 2 * (kRasterThreads / 64);
                if (!hill_batch_build(p, smem, (unsigned)wp.lds_off_hill, Pv, nb, hbar, done_before, tid, lane, bail)) return;
#pragma unroll
                for (int bi = 0; bi < kHillBatchMax; ++bi)
                    if (bi < nb)
                        raster_hill_frame<DEPTH>(p, rth, smem, (unsigned)wp.lds_off_hill, bi, hbar, done_before, frame_desc<DEPTH>(p, img, dep, e_begin + b0 + bi), cams[bi]);
                if (mine_j >= 0 && !(kDiag & 2)) {                    // the step's telemetry of this wave's env of the batch (a wave owns at most one of a batch of <= 4 <= 8)
                    const size_t e = (size_t)(e_begin + mine_j);
                    if (lane < 12) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(uintptr_t)optr + e, tel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else if (lane < 14) __hip_atomic_store((__attribute__((address_space(1))) unsigned char*)(uintptr_t)optr + e, (unsigned char)(lane == 12 ? tel : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            continue;
        }
        const bool sweep = ahead == 0;
        if (sweep)
            for (int j = 0; j < n_loc; ++j) raster_uniform_rows<DEPTH>(p, rth, frame_desc<DEPTH>(p, img, dep, e_begin + j));
        for (int j = 0; j < n_loc; ++j) {
            const FrameDesc fd = frame_desc<DEPTH>(p, img, dep, e_begin + j);
            if (!sweep) raster_uniform_rows<DEPTH>(p, rth, fd);
            if (j == 0)                                       // arrivals owed: everything this wave has issued since the end of step `owed`
                while (s - owed >= keep) {                    // is (s - owed - 1) whole steps + this step's uniform rows so far
                    const u64 tv0 = probe ? now_clk() : 0;
                    if (!(kDiag & 1)) wait_vmcnt_le((int)(s - owed - 1) * nstep + (sweep ? nuni : nu));
                    if (probe) t_vm += now_clk() - tv0;
                    raster_arrive(l, owed++, lane);
                }
            const u64 tq0 = probe ? now_clk() : 0;
            if (!wait_lds_ge(wp, l, nullptr, &l.pprog[j], r + 1, 3u, lane)) return;
            if (probe) t_pose += now_clk() - tq0;
            const float* const sl = l.slot + ((size_t)(r & (kCamDepth - 1)) * epw + j) * kSlotWords;
            const float4 cam = *reinterpret_cast<const float4*>(sl);
            const bool mine = (j % (kRasterThreads / 64)) == wave;
            unsigned tel = 0;
            if (mine) tel = __float_as_uint(sl[4 + min(lane, 12)]);      // lanes 0..11 their word, lane 12 `done`
            asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(cam.x), "v"(cam.y), "v"(cam.z), "v"(cam.w), "v"(tel) : "memory");
            if (lane == 0) __hip_atomic_fetch_add(&l.rread[j], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            raster_ground_rows<DEPTH>(p, rth, fd, cam);
            if (mine && !(kDiag & 2)) {                     // the step's telemetry of env j: two wave instructions, written through
                const size_t e = (size_t)(e_begin + j);
                if (lane < 12) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(uintptr_t)optr + e, tel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else if (lane < 14) __hip_atomic_store((__attribute__((address_space(1))) unsigned char*)(uintptr_t)optr + e, (unsigned char)(lane == 12 ? tel : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---- physics-only envs (cfg.render == 0, BASELINE configs[1]): the consumer-paced step without a launch per step -------------------
// The same mailbox, device ring, arrival counters and exit protocol as trs_worker_kernel, on the geometry of trs_physics_kernel: one env
// per wave, four envs per workgroup, the 38 KB track image staged once.  A workgroup has two more waves that own no env:
//   wave 4, the service wave: leader (keeps the workgroup's LDS word fresh from the device word) and forwarder (passes whole-workgroup
//           arrivals on: sharded device counter -> top counter -> done flag in the mailbox);
//   wave 5, workgroup 0 only: the dispatcher (dispatcher_run, unchanged: PCIe polls must not sit in front of any env's integration).
// A physics wave keeps its env in registers for the whole launch.  Per step: wait for the post, read the entry from the device ring — a
// load, and vector-memory operations complete in order on this chip, so once it has returned every store of the previous step has been
// acknowledged: the wave arrives for step s - 1 here, without a drain (the lagged arrival of the raster waves, with a lag of one) — then
// the controls (system-scope loads), env_advance (the same inlined routine as every other step path: bit-identical results) and the
// step's telemetry, written through (14 arrays, one lane each: two wave instructions).  With nothing queued behind the step (a
// lock-step consumer is waiting for it) the wave drains and arrives at once.  With step s + 1 already posted its inputs are requested in
// front of step s's arithmetic (the control loads, system scope, are the slowest part of a step's critical path).
constexpr int kPwEnvs = 4;                          // envs (= physics waves) per workgroup
constexpr int kPwBlock = 64 * (kPwEnvs + 2);

__global__ __launch_bounds__(kPwBlock) void trs_physics_worker_kernel(const WParams wp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PParams& P = wp.ph;
    const WLds l = wlds_of(smem + wp.lds_off_ctl, kPwEnvs);
    const int e_begin = blockIdx.x * kPwEnvs;
    const int n_loc = min(e_begin + kPwEnvs, P.n_envs) - e_begin;
    if ((unsigned)(uintptr_t)smem != 0u) {                   // the track image is addressed from LDS offset 0
        if (tid == 0) {
            atomicAdd(&P.stats[2], 1ull);
            sys_store64(P.fault, 1ull);
            sys_store64(&wp.mb->error, (9ull << 32) | (u64)blockIdx.x);
            __hip_atomic_fetch_or(&wp.dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) { sys_store64(&wp.mb->consumed, wp.start); sys_store64(&wp.mb->exited, 1ull); }
        }
        return;
    }
    stage_lds_dma(P.blob, P.blob_bytes, 0u, wave, kPwBlock / 64, lane);
    if (tid == 0) { lds_store64(l.word, wp.start); lds_store64(l.fwd, wp.start); }
    if (tid < kSlots) l.arrive[tid] = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) worker_checkin(wp);

    if (wave == kPwEnvs + 1) {                               // the dispatcher: workgroup 0 only
        if (blockIdx.x == 0) dispatcher_run(wp, l, lane);
        return;
    }
    if (wave == kPwEnvs) {                                   // the service wave: leader + forwarder, until the launch's last step has been passed on
        Duties D{blockIdx.x != 0, true, wp.start, n_loc};
        if (D.leader && !leader_wait_decision(wp, l, lane)) return;
        u64 s = wp.start;
        while (wait_posted(wp, l, D, s, lane)) ++s;          // returns at once for a step that is posted: s runs up to the published count, then the wait does the duties
        const u64 w = lds_load64(l.word);
        if (w & kAbortBit) return;
        const u64 last = w & kCountMask;                     // EXIT | final count: every physics wave finishes the steps below it
        u64 t0 = 0;
        for (unsigned spins = 0; D.fwd < last; ++spins) {
            forward_arrivals(wp, l, D.fwd, lane, n_loc);
            if (lds_load64(l.word) & kAbortBit) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 1023u) == 1023u) {
                const u64 now = (u64)wall_clock64();
                if (t0 == 0) t0 = now;
                else if (now - t0 > wp.safety_ticks) { worker_abort(wp, l, 4u); break; }
            }
        }
        return;
    }
    if (wave >= n_loc) return;                               // the last workgroup may own fewer than four envs

    const int e = e_begin + wave;
    EnvRegs st;
    env_load(P, e, st);
    float lr = coherent_load(&P.last_return[e]);
    unsigned char* optr = nullptr;                           // this lane's telemetry array (lanes 0..11: 4-byte arrays, lanes 12, 13: done, pending)
    {
        unsigned char* const tab[14] = {(unsigned char*)P.x, (unsigned char*)P.y, (unsigned char*)P.z, (unsigned char*)P.yaw, (unsigned char*)P.v,
                                        (unsigned char*)P.speed, (unsigned char*)P.cte, (unsigned char*)P.seg_idx, (unsigned char*)P.ep_return,
                                        (unsigned char*)P.ep_len, (unsigned char*)P.steer_filt, (unsigned char*)P.last_return,
                                        (unsigned char*)P.done, (unsigned char*)P.pending};
#pragma unroll
        for (int k = 0; k < 14; ++k) optr = lane == k ? tab[k] : optr;
    }
    Duties none{false, false, 0};
    u64 owed = wp.start;                                     // oldest step this wave has not arrived for
    // One step's inputs come through two dependent memory round trips: the step's entry in the device ring (control POINTERS, ~0.7 us), then
    // this env's controls (system-scope loads, ~1 us).  Both leave the critical path when the consumer keeps steps queued: while step s
    // integrates (~2 us of dependent arithmetic), the controls of step s + 1 and the entry of step s + 2 are already on their way
    // (a tick posted on its own: 3.9 us with neither, 3.0 with the controls, see profiles/r04_sweep.txt for both).
    struct Ctl { float steer, thr, brk; uint8_t rin; int synth; };
    auto entry_load = [&](u64 s) -> u64 {                    // lanes 1..5: steer, thr, brk, reset, synth of step s (WEntry words)
        const u64* en = reinterpret_cast<const u64*>(&wp.dc->ring[s & (kSlots - 1)]);
        return lane < 6 ? agent_load64(en + lane) : 0ull;
    };
    auto ctl_load = [&](u64 ev) -> Ctl {                     // reading the entry's lanes waits for its load; the control loads are only issued here
        const float* const c_st = reinterpret_cast<const float*>(lane_u64(ev, 1));
        const float* const c_th = reinterpret_cast<const float*>(lane_u64(ev, 2));
        const float* const c_br = reinterpret_cast<const float*>(lane_u64(ev, 3));
        const uint8_t* const c_rs = reinterpret_cast<const uint8_t*>(lane_u64(ev, 4));
        Ctl in{0.f, 0.f, 0.f, 0, (int)(unsigned)lane_u64(ev, 5)};
        asm volatile("" :: "s"(c_st), "s"(c_th), "s"(in.synth) : "memory");   // the entry HAS returned here (and with it every older store of this wave)
        if (!in.synth) {
            in.steer = sys_load_val(&c_st[e]); in.thr = sys_load_val(&c_th[e]);
            if (c_br) in.brk = sys_load_val(&c_br[e]);
            if (c_rs) in.rin = sys_load_val(&c_rs[e]);
        }
        return in;
    };
    bool have_ctl = false, have_ent = false;                 // the controls of step s / the entry of step s + 1 were requested by an earlier iteration
    Ctl nxt{};
    u64 ent_next = 0ull;
    u64 clean_below = wp.start;                              // every store of the steps below this index has been acknowledged
    for (u64 s = wp.start;; ++s) {
        if (owed < s && (lds_load64(l.word) & kCountMask) <= s) {   // nothing further posted: the consumer may be waiting for step s - 1
            drain_vmem();
            for (; owed < s; ++owed) raster_arrive(l, owed, lane);
            clean_below = s;
        }
        Ctl in;
        if (have_ctl) {
            in = nxt;
        } else {
            if (!wait_posted(wp, l, none, s, lane)) {
                drain_vmem();
                for (; owed < s; ++owed) raster_arrive(l, owed, lane);
                return;
            }
            in = ctl_load(entry_load(s));                    // (an entry in flight implies controls in flight: have_ent is false here)
            drain_vmem();                                    // loads return in order behind every older store of this wave: steps < s are in memory
            clean_below = s;
        }
        const u64 count = lds_load64(l.word) & kCountMask;
        // the controls of step s + 1: their entry was requested one iteration ago (behind the stores of step s - 2: once it is read, the steps
        // below s - 1 are in memory) or is read now (behind the stores of step s - 1)
        have_ctl = count > s + 1;
        if (have_ctl) {
            const bool was_prefetched = have_ent;
            nxt = ctl_load(was_prefetched ? ent_next : entry_load(s + 1));
            const u64 c = was_prefetched ? s - 1 : s;
            clean_below = c > clean_below ? c : clean_below;
        }
        // the entry of step s + 2, not waited for
        have_ent = have_ctl && count > s + 2;
        if (have_ent) ent_next = entry_load(s + 2);
        for (; owed < clean_below; ++owed) raster_arrive(l, owed, lane);   // lagged arrivals: no drain of their own
        const float epr_before = st.epr;
        StepOut o;
        env_advance<true, false, true>(P, smem, e, st, (uint32_t)s, in.synth, in.steer, in.thr, in.brk, in.rin, lane, o);   // (the select forms: see spec_sincos_sel)
        if (o.do_reset) lr = epr_before;
        if (lane == 0) {
            if (o.is_done) atomicAdd(&P.stats[0], 1ull);
            if (o.do_reset) atomicAdd(&P.stats[1], 1ull);
        }
        unsigned tel = 0;
        {
            const unsigned vals[13] = {__float_as_uint(st.x), __float_as_uint(st.y), __float_as_uint(st.z), __float_as_uint(st.yaw), __float_as_uint(st.v),
                                       __float_as_uint(st.speed), __float_as_uint(st.cte), (unsigned)st.seg, __float_as_uint(st.epr), (unsigned)st.epl,
                                       __float_as_uint(st.sf), __float_as_uint(lr), (unsigned)st.done};
#pragma unroll
            for (int k = 0; k < 13; ++k) tel = lane == k ? vals[k] : tel;
        }
        if (lane < 12) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)(uintptr_t)optr + e, tel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (lane < 14) __hip_atomic_store((__attribute__((address_space(1))) unsigned char*)(uintptr_t)optr + e, (unsigned char)(lane == 12 ? tel : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

// the worker's LDS need with the track that is loaded NOW (a larger track may have been loaded since resident mode was selected)
// and the frame filter that is set NOW (the dynamic-brightness filter adds the per-env palettes of a batch of four envs)
int worker_fits(trs_env* e)
{
    Resident* R = e->res;
    if (!e->cfg.render) {                                    // physics-only envs: the track image + the control block (trs_physics_worker_kernel)
        R->lds_off_ctl = (e->lds_p + 15) & ~15;
        R->lds_bytes = (int)(R->lds_off_ctl + wlds_bytes(kPwEnvs) + 16);
        if (R->lds_bytes > 160 * 1024) return trs_internal_fail(TRS_ERR_LIMIT, "track image too large for the resident physics worker");
        // The grid is ceil(n / 4) workgroups and ALL of them must be on the GPU at once (arrival counters, done flags: see the note on
        // DeviceSlot).  What the GPU holds at once is what the occupancy query says for this block size and LDS need, per CU (ADVICE r04:
        // beyond it the workgroups that found no room never arrive — a step would complete only when the others idle out).
        if (R->pw_capacity_lds != R->lds_bytes) {            // (asked once per LDS need: this runs in front of every worker start)
            int per_cu = 0;
            RCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(trs_physics_worker_kernel), kPwBlock, (size_t)R->lds_bytes));
            R->pw_capacity_envs = (long long)per_cu * (long long)std::max(e->cu_count, 1) * kPwEnvs;
            R->pw_capacity_lds = R->lds_bytes;
        }
        const long long capacity = R->pw_capacity_envs / kPwEnvs;    // 0 = the runtime gave no answer: the device-side check-in decides
        const long long grid = ((long long)e->n + kPwEnvs - 1) / kPwEnvs;
        if (capacity > 0 && grid > capacity)
            return trs_internal_fail(TRS_ERR_LIMIT, "resident mode: " + std::to_string(e->n) + " physics-only envs need " + std::to_string(grid) +
                                                        " co-resident workgroups, this GPU holds " + std::to_string(capacity) + " (" + std::to_string(capacity * kPwEnvs) +
                                                        " envs): use TRS_STEP_LAUNCH with several steps per launch for shards of this size");
        return TRS_OK;
    }
    R->lds_off_ctl = (e->lds_step + 15) & ~15;              // behind the tables
    R->lds_bytes = (int)(R->lds_off_ctl + wlds_bytes(e->pp.envs_per_wg) + 16);
    if (e->has_frame_filter && e->filter_dynamic) { R->lds_off_dyn = (R->lds_bytes + 15) & ~15; R->lds_bytes = R->lds_off_dyn + dyn_lds_bytes(e->H); }
    R->lds_off_hill = (R->lds_bytes + 15) & ~15;
    if (e->hilly) R->lds_bytes = R->lds_off_hill + hill_lds_bytes(e->H);   // a track with elevation: the per-env row tables (trsim_device.hpp, hill_rows_build)
    if (R->lds_bytes > 160 * 1024)
        return trs_internal_fail(TRS_ERR_LIMIT, e->hilly ? "the resident worker's LDS state and the per-env row tables of a track with elevation do not fit beside this track's tables: use TRS_STEP_LAUNCH"
                                                            : "too many envs per workgroup for the resident worker's LDS state");
    return TRS_OK;
}

int evict(trs_env* other);

int worker_launch(trs_env* e, uint64_t start)
{
    Resident* R = e->res;
    Mailbox* mb = R->mb;
    { int rc = worker_fits(e); if (rc) return rc; }
    {   // one worker per GPU: the worker of another handle of this process leaves first (the caller holds the device's lock)
        DeviceSlot& D = slot_of(e);
        trs_env* const other = D.owner;
        if (other && other != e && other->res && other->res->running) (void)evict(other);   // (a worker that gave up is gone all the same: its handle reports it)
        D.owner = e;
    }
    host_store(&mb->exited, 0); host_store(&mb->consumed, start); host_store(&mb->error, 0);
    host_store(&mb->close, 0); host_store(&mb->started, 0);
    WParams wp{};
    wp.ph = e->pp; wp.ph.synth = 0; wp.ph.write_cam = 0; wp.ph.n_steps = 0; wp.ph.step_off = 0; wp.ph.ctl_stride = 0;
    wp.ra = e->rp;
    wp.img0 = e->img[0]; wp.img1 = e->img[1]; wp.dep0 = e->depth[0]; wp.dep1 = e->depth[1];
    e->uniform_ok[0] = e->uniform_ok[1] = false;            // (launch_step's bookkeeping of which buffer holds whole frames of the current palette: not kept across a worker)
    wp.mb = mb; wp.dc = R->dc;
    wp.start = start;
    wp.idle_ticks = (unsigned long long)R->idle_us * 100ull;
    wp.life_ticks = (unsigned long long)R->life_us * 100ull;   // 50 ms by default: then the dispatcher leaves and the host starts a new worker at its next post
    wp.safety_ticks = 200000000ull;                         // 2 s
    wp.checkin_ticks = 200000ull;                           // 2 ms: every workgroup of a launch that has the GPU to itself reports in within tens of microseconds
    wp.lds_off_phys = e->lds_off_phys; wp.lds_off_ctl = R->lds_off_ctl; wp.lds_off_hill = R->lds_off_hill;
    const int grid = e->cfg.render ? (e->n + e->pp.envs_per_wg - 1) / e->pp.envs_per_wg : (e->n + kPwEnvs - 1) / kPwEnvs;
    wp.n_blocks = grid;
    // the control block is set up by a one-workgroup kernel in front of the worker — unless the previous worker's normal exit already did that for exactly this start
    // (handle_exit): the set-up then sits in the stream's idle time instead of in front of every launch (~4 us per worker start)
    if (!(R->dc_ready && R->dc_ready_for == start)) hipLaunchKernelGGL(trs_worker_init_kernel, dim3(1), dim3(256), 0, e->sP, R->dc, (u64)start);
    R->dc_ready = false;
    if (!e->cfg.render) {
        hipLaunchKernelGGL(trs_physics_worker_kernel, dim3(grid), dim3(kPwBlock), R->lds_bytes, e->sP, wp);
        RCHK(hipGetLastError());
        R->running = true;
        R->t_launch = std::chrono::steady_clock::now();
        return TRS_OK;
    }
    const bool dyn = e->has_frame_filter && e->filter_dynamic;
    if (dyn) {                                              // the fields trs_step_kernel's DYN instantiation gets (trsim_hip.hip, launch_step)
        const trs_pre_config& c = e->frame_filter;
        wp.fp.baseline = c.brightness_baseline; wp.fp.contrast = c.contrast_ratio; wp.fp.offset = c.contrast_offset;
        wp.fp.color = c.color_filter_enabled; wp.fp.n_filters = c.n_filters;
        for (int k = 0; k < 4; ++k) {
            wp.fp.lo[k] = c.hsv_lo[k][0] | (c.hsv_lo[k][1] << 8) | (c.hsv_lo[k][2] << 16);
            wp.fp.hi[k] = c.hsv_hi[k][0] | (c.hsv_hi[k][1] << 8) | (c.hsv_hi[k][2] << 16);
            wp.fp.dst_ch[k] = c.dst_channel[k];
        }
        wp.fp.w0 = std::min(40, e->H); wp.fp.w1 = std::min(119, e->H);     // img[40:119] (img_preprocessing.py:88)
        wp.fp.tabs = e->dyn_tab;
        wp.fp.lds_off = R->lds_off_dyn;
    }
    if (e->hilly) {                                       // a track with elevation (no frame filters there)
        if (e->rp.depth) hipLaunchKernelGGL((trs_worker_kernel<true, false, true>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
        else hipLaunchKernelGGL((trs_worker_kernel<false, false, true>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
        RCHK(hipGetLastError());
        R->running = true;
        R->t_launch = std::chrono::steady_clock::now();
        return TRS_OK;
    }
    if (e->rp.depth) { if (dyn) hipLaunchKernelGGL((trs_worker_kernel<true, true>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
                       else hipLaunchKernelGGL((trs_worker_kernel<true, false>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp); }
    else { if (dyn) hipLaunchKernelGGL((trs_worker_kernel<false, true>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp);
           else hipLaunchKernelGGL((trs_worker_kernel<false, false>), dim3(grid), dim3(kBlock), R->lds_bytes, e->sP, wp); }
    RCHK(hipGetLastError());
    R->running = true;
    R->t_launch = std::chrono::steady_clock::now();
    return TRS_OK;
}

int worker_error(trs_env* e)
{
    const uint64_t err = host_load(&e->res->mb->error);
    if (!err) return TRS_OK;
    e->res->broken = true;
    static const char* const what[] = {"", "waiting for a post", "camera ring back-pressure", "waiting for the physics team", "forwarding the last arrivals",
                                       "team barrier of the dynamic-brightness batch", "", "abort injected by trs_resident_debug_abort (test hook)", "",
                                       "dynamic LDS segment not at offset 0"};
    const unsigned code = (unsigned)(err >> 32);
    return trs_internal_fail(TRS_ERR_DEVICE, std::string("resident worker gave up (") + (code < 10 ? what[code] : "?") + ") in workgroup " +
                                                 std::to_string((unsigned)err));
}

// The launch was not co-resident (dispatcher_checkin) or never reported in (wait_done's deadline): the GPU is shared with the worker of another
// process.  The handle goes back to TRS_STEP_LAUNCH — launches need no co-residency, they make progress whenever CUs come free — and the steps
// that were posted but not consumed run as launches now, with the control pointers of their posts (the ring is host memory; pinned staging
// slots of trs_step_host are device-readable).  Returns kFellBack (> 0): not an error, the posted steps will complete on the handle's stream.
int fall_back_to_launches(trs_env* e, uint64_t consumed, const char* why)
{
    Resident* R = e->res;
    const uint64_t posted = host_load(&R->mb->posted);
    R->enabled = false;
    R->fell_back = true;
    R->t_fallback = std::chrono::steady_clock::now();
    if (consumed > R->seen_done) R->seen_done = consumed;
    for (uint64_t s = consumed; s < posted; ++s) {
        const WEntry& en = R->mb->ring[s & (kSlots - 1)];
        if (en.seq != s + 1 || en.seq_lo != s + 1)
            return trs_internal_fail(TRS_ERR_DEVICE, "resident worker: the post of step " + std::to_string(s) + " is no longer in the ring");
        int rc = trs_internal_replay_launch(e, en.steer, en.thr, en.brk, en.reset, (int)en.synth, s);
        if (rc) return rc;
    }
    R->launched = true;                                      // these steps have no completion flag: the stream is what to wait for
    R->base = R->seen_done = e->step_count;
    for (int k = 0; k < kSlots; ++k) { host_store(&R->mb->ring[k].seq, 0); host_store(&R->mb->ring[k].seq_lo, 0); host_store(&R->mb->done[k], 0); }
    trs_internal_note(std::string("resident worker: ") + why + " - another process's worker holds CUs of this GPU; the handle has gone back to TRS_STEP_LAUNCH "
                      "(resident mode is tried again by itself after " + std::to_string(R->retry_ms) + " ms; trs_get_step_mode tells)");
    return kFellBack;
}

// the worker has said it leaves (or has been told to): wait for the kernel, restart it if posts raced with its exit.
// relaunch = false (eviction by another handle): posts that raced stay in the ring, the handle's next call starts a worker from seen_done.
int handle_exit(trs_env* e, bool relaunch = true)
{
    Resident* R = e->res;
    RCHK(hipStreamSynchronize(e->sP));
    R->running = false;
    int rc = worker_error(e);
    if (rc) return rc;
    const uint64_t consumed = host_load(&R->mb->consumed), posted = host_load(&R->mb->posted);
    if (host_load(&R->mb->exited) == kExitNotCoresident)
        return fall_back_to_launches(e, consumed, "the launch did not get every CU within 2 ms");
    if (consumed > R->seen_done) R->seen_done = consumed;   // the kernel has ended: everything it consumed is complete
    if (host_load(&R->mb->started)) R->retry_ms = kRetryMs0;   // this launch had the GPU: the sharing that caused an earlier fallback is over
    if (consumed < posted && relaunch) return worker_launch(e, consumed);
    if (consumed == posted) {                               // nothing left to serve: the next launch of this handle starts here — its control block is set up now
        hipLaunchKernelGGL(trs_worker_init_kernel, dim3(1), dim3(256), 0, e->sP, R->dc, (u64)consumed);
        if (hipGetLastError() == hipSuccess) { R->dc_ready = true; R->dc_ready_for = consumed; }
    }
    return TRS_OK;
}

// another handle of this process wants the GPU for its worker: this one finishes what was posted and leaves
int evict(trs_env* other)
{
    Resident* R = other->res;
    int rc = TRS_OK;
    if (R->running) {
        host_store(&R->mb->close, kCloseLeave);
        rc = handle_exit(other, false);                      // (kFellBack: it left for launch mode — the GPU is free of it just the same)
    }
    return rc < 0 ? rc : TRS_OK;
}

// the host has waited too long for a launch to report in: the kernel sits in the queue behind something that does not leave (another
// process's worker).  Cancel it (the dispatcher leaves at its first look, or serves what it had already begun) and go back to launches.
int give_up_on_launch(trs_env* e)
{
    Resident* R = e->res;
    host_store(&R->mb->close, kCloseCancel);
    RCHK(hipStreamSynchronize(e->sP));                       // bounded by the other worker's lifetime (50 ms) or idle time
    R->running = false;
    int rc = worker_error(e);
    if (rc) return rc;
    return fall_back_to_launches(e, host_load(&R->mb->consumed), "the launch had not started after 250 ms");
}

int wait_done(trs_env* e, uint64_t s)
{
    Resident* R = e->res;
    if (s < R->seen_done) return TRS_OK;
    Mailbox* mb = R->mb;
    const auto t0 = std::chrono::steady_clock::now();
    auto fell_back = [&]() -> int {                           // the posted steps went onto the stream as launches: wait for the stream
        RCHK(hipStreamSynchronize(e->sP));
        R->launched = false;
        R->base = R->seen_done = e->step_count;
        return TRS_OK;
    };
    for (unsigned spins = 0;; ++spins) {
        if (host_load(&mb->done[s & (kSlots - 1)]) >= s + 1) { R->seen_done = s + 1; return TRS_OK; }
        if ((spins & 63u) == 63u) {
            if (R->running && host_load(&mb->exited)) {
                int rc = handle_exit(e);
                if (rc == kFellBack) return fell_back();
                if (rc) return rc;
                if (s < R->seen_done) return TRS_OK;
            } else if (!R->running) {
                if (host_load(&mb->posted) > R->seen_done) { int rc = worker_launch(e, R->seen_done); if (rc) return rc; }
            } else if ((spins & 4095u) == 4095u && !host_load(&mb->started) &&
                       std::chrono::steady_clock::now() - R->t_launch > std::chrono::milliseconds(250)) {
                int rc = give_up_on_launch(e);
                if (rc == kFellBack) return fell_back();
                if (rc) return rc;
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
    // The copy stream must not share a hardware queue with the handle's stream: a copy queued behind the worker kernel on the same queue would
    // wait until the worker leaves (seen in round 4: 100 ms = idle_us per trs_fetch_outputs in a process that had created many streams, where
    // the runtime's least-used-queue choice put both streams on one queue).  The runtime keeps a separate pool of hardware queues per stream
    // priority, so the copy stream takes the HIGHEST priority and the handle's stream (default priority) can never alias it.
    {
        int least = 0, greatest = 0;
        RCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        RCHK(hipStreamCreateWithPriority(&R->sC, hipStreamNonBlocking, greatest));
    }
    R->hctl_slot = ((size_t)e->n * 13 + 63) & ~(size_t)63;
    RCHK(hipHostMalloc((void**)&R->hctl, R->hctl_slot * kSlots, hipHostMallocMapped | hipHostMallocCoherent));
    return TRS_OK;
}

}  // namespace

namespace trsim {

bool resident_on(const trs_env* e) { return e && e->res && e->res->enabled; }

// A handle that went back to launches because the GPU was shared with another process's worker tries resident mode again by itself:
// called in front of every step call.  The step that follows starts a worker; if the GPU is still shared the launch is called off
// again within milliseconds (dispatcher_checkin) and the interval doubles (100 ms .. 2 s).
void resident_retry(trs_env* e)
{
    Resident* R = e ? e->res : nullptr;
    if (!R || R->enabled || !R->fell_back || R->broken || !R->mb) return;
    DevLock lock(e);
    if (std::chrono::steady_clock::now() - R->t_fallback < std::chrono::milliseconds(R->retry_ms)) return;
    if (worker_fits(e) != TRS_OK) return;
    R->retry_ms = std::min(R->retry_ms * 2u, 2000u);
    R->base = R->seen_done = e->step_count;
    host_store(&R->mb->posted, e->step_count);
    R->launched = false;
    R->enabled = true;
    R->fell_back = false;
}
bool resident_running(const trs_env* e) { return e && e->res && e->res->running; }
void resident_clear_fault(trs_env* e) { if (e && e->res) e->res->broken = false; }
bool resident_fits_dynamic_filter(const trs_env* e)
{
    const int off_ctl = (e->lds_step + 15) & ~15;
    const int base = (int)(off_ctl + wlds_bytes(e->pp.envs_per_wg) + 16);
    return ((base + 15) & ~15) + dyn_lds_bytes(e->H) <= 160 * 1024;
}

int resident_post(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, int n, size_t stride, int* n_done)
{
    DevLock lock(e);
    Resident* R = e->res;
    Mailbox* mb = R->mb;
    *n_done = 0;
    if (R->broken) return trs_internal_fail(TRS_ERR_DEVICE, "a resident worker gave up earlier: the env state is undefined, load the track again (trs_load_track)");
    for (int k = 0; k < n; ++k) {
        const uint64_t s = e->step_count;
        if (!R->running) {
            if (host_load(&mb->posted) > R->seen_done && host_load(&mb->posted) == s) {
                // posts that raced with the worker's exit while another handle took the GPU over (evict): they are still in the ring
                int rc = worker_launch(e, R->seen_done);
                if (rc) return rc;
            } else { R->base = R->seen_done = s; R->launched = false; }   // no worker: nothing is in flight that a flag will report (the step counter may have moved or restarted since; launched steps are ahead of the worker on the same stream)
        }
        if (s >= R->base + kSlots) {                                  // ring slot, counters and done flag of s % 8 are free
            int rc = wait_done(e, s - kSlots);
            if (rc) return rc;
            if (!R->enabled) { *n_done = k; return kFellBack; }       // the handle went back to launch mode: the caller launches steps k.. itself
        }
        WEntry en{};
        const size_t off = (size_t)k * stride;
        en.steer = st ? st + off : nullptr; en.thr = th ? th + off : nullptr; en.brk = br ? br + off : nullptr;
        en.reset = k == 0 ? rs : nullptr; en.synth = synth ? 1u : 0u;
        if (!R->running) { int rc = worker_fits(e); if (rc) return rc; }   // nothing is published for a worker that could not be launched
        WEntry* slot = &mb->ring[s & (kSlots - 1)];
        slot->steer = en.steer; slot->thr = en.thr; slot->brk = en.brk;
        host_store(&slot->seq_lo, s + 1);                         // first half: payload, then its tag (x86 keeps the store order)
        slot->reset = en.reset; slot->synth = en.synth;
        host_store(&slot->seq, s + 1);                            // second half likewise; this tag last: the line is now a valid post
        host_store(&mb->posted, s + 1);
        std::atomic_thread_fence(std::memory_order_seq_cst);     // the post is visible before `exited` is read
        e->step_count = s + 1;
        if (!R->running) { int rc = worker_launch(e, s); if (rc) return rc; }
        else if (host_load(&mb->exited)) {
            int rc = handle_exit(e);
            if (rc == kFellBack) { *n_done = k + 1; return kFellBack; }   // (step k was in the ring: it has been launched with the rest)
            if (rc) return rc;
        }
    }
    *n_done = n;
    return TRS_OK;
}

int resident_wait(trs_env* e)
{
    Resident* R = e->res;
    if (!R) return TRS_OK;
    DevLock lock(e);
    if (R->launched && !R->running) {                        // the newest steps went through launches (resident_note_launch): wait for the stream
        RCHK(hipStreamSynchronize(e->sP));
        R->launched = false;
        R->base = R->seen_done = e->step_count;
        return TRS_OK;
    }
    if (e->step_count <= R->seen_done || e->step_count <= R->base) return TRS_OK;
    return wait_done(e, e->step_count - 1);
}

// a step was launched on the handle's stream although resident mode is selected (the pilot loop: its kernels need the LDS a worker
// would hold).  The caller has quiesced the worker; the step has no post and gets no completion flag.
void resident_note_launch(trs_env* e)
{
    Resident* R = e->res;
    if (!R) return;
    DevLock lock(e);
    R->launched = true;
    R->base = R->seen_done = e->step_count;
}

int resident_quiesce(trs_env* e)
{
    Resident* R = e->res;
    if (!R || !R->mb) return TRS_OK;
    DevLock lock(e);
    int rc = TRS_OK;
    if (!R->running && R->enabled && !R->broken && host_load(&R->mb->posted) > R->seen_done && host_load(&R->mb->posted) == e->step_count)
        rc = worker_launch(e, R->seen_done);                 // posts an eviction left in the ring (see resident_post): they complete here
    for (int guard = 0; !rc && R->running && guard < 4; ++guard) {
        host_store(&R->mb->close, kCloseLeave);
        rc = handle_exit(e);                                 // waits for the kernel; relaunches (with `close` cleared) if posts raced
        if (rc == kFellBack) { rc = TRS_OK; break; }         // the posted steps are launches on the handle's stream now
    }
    if (!rc && R->running) rc = trs_internal_fail(TRS_ERR_DEVICE, "resident worker did not leave");
    R->base = R->seen_done = e->step_count;
    if (!R->running)                                         // tags and flags of the past must not match a step index that comes round again
        for (int k = 0; k < kSlots; ++k) { host_store(&R->mb->ring[k].seq, 0); host_store(&R->mb->ring[k].seq_lo, 0); host_store(&R->mb->done[k], 0); }
    return rc;
}

void resident_destroy(trs_env* e)
{
    Resident* R = e->res;
    if (!R) return;
    DevLock lock(e);
    if (slot_of(e).owner == e) slot_of(e).owner = nullptr;
    if (R->running) { host_store(&R->mb->close, kCloseLeave); (void)hipStreamSynchronize(e->sP); }
    if (R->mb) (void)hipHostFree(R->mb);
    if (R->hctl) (void)hipHostFree(R->hctl);
    (void)hipFree(R->dc);
    if (R->sC) (void)hipStreamDestroy(R->sC);
    delete R;
    e->res = nullptr;
}

hipStream_t resident_copy_stream(trs_env* e) { return (e->res && e->res->running) ? e->res->sC : e->sP; }

// controls handed over as host arrays: into this step's slot of the pinned staging buffer, which the device reads over PCIe
int resident_post_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps, int* n_done)
{
    DevLock lock(e);
    Resident* R = e->res;
    *n_done = 0;
    const uint64_t s = e->step_count;
    if (!R->running && !(host_load(&R->mb->posted) > R->seen_done && host_load(&R->mb->posted) == s)) R->base = R->seen_done = s;
    if (s >= R->base + kSlots) {                             // the staging slot is free as well
        int rc = wait_done(e, s - kSlots);
        if (rc) return rc;
        if (!R->enabled) return kFellBack;
    }
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
            if (sk >= R->base + kSlots) {
                int rc = wait_done(e, sk - kSlots);
                if (rc) return rc;
                if (!R->enabled) { *n_done = k; return kFellBack; }
            }
            unsigned char* sl = R->hctl + (sk & (kSlots - 1)) * R->hctl_slot;
            if (sl != slot) std::memcpy(sl, slot, n * 12);
            slot = sl; f = reinterpret_cast<float*>(slot); rsb = slot + n * 12;
        }
        int one = 0;
        int rc = resident_post(e, f, f + n, h_br ? f + 2 * n : nullptr, (k == 0 && h_rs) ? rsb : nullptr, 0, 1, 0, &one);
        if (rc == kFellBack) { *n_done = k + one; return kFellBack; }
        if (rc) return rc;
    }
    *n_done = n_steps;
    return TRS_OK;
}

}  // namespace trsim

#ifdef TRS_TEST_HOOKS   /* csrc/libtrsim_testhooks.so only (include/trsim.h) */
TRS_EXPORT int trs_resident_debug_lifetime(trs_env* e, int life_us)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (!e->res) return trs_internal_fail(TRS_ERR_STATE, "resident mode has not been selected on this handle");
    DevLock lock(e);
    e->res->life_us = life_us > 0 ? (unsigned)std::min(life_us, 10000000) : kDefaultLifeUs;
    return TRS_OK;
}

// test hook: what a wave does when one of its bounded waits gives up, done from outside — the abort bit in the device word and the
// error word in the mailbox, by a one-thread kernel on the side stream (the worker owns the handle's stream)
__global__ void trs_worker_debug_abort_kernel(Mailbox* mb, DevCtl* dc)
{
    sys_store64(&mb->error, (7ull << 32));
    __hip_atomic_fetch_or(&dc->word, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

TRS_EXPORT int trs_resident_debug_abort(trs_env* e)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    DevLock lock(e);
    if (!e->res || !e->res->running) return trs_internal_fail(TRS_ERR_STATE, "no resident worker is running on this handle");
    RCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(trs_worker_debug_abort_kernel, dim3(1), dim3(1), 0, e->res->sC, e->res->mb, e->res->dc);
    RCHK(hipGetLastError());
    RCHK(hipStreamSynchronize(e->res->sC));
    return TRS_OK;
}
#endif

TRS_EXPORT int trs_get_step_mode(trs_env* e, int* mode, int* fell_back)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    DevLock lock(e);
    if (mode) *mode = (e->res && e->res->enabled) ? TRS_STEP_RESIDENT : TRS_STEP_LAUNCH;
    if (fell_back) *fell_back = (e->res && e->res->fell_back) ? 1 : 0;
    return TRS_OK;
}

TRS_EXPORT int trs_set_step_mode(trs_env* e, int mode, int idle_us)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (mode != TRS_STEP_LAUNCH && mode != TRS_STEP_RESIDENT) return trs_internal_fail(TRS_ERR_ARG, "mode must be TRS_STEP_LAUNCH or TRS_STEP_RESIDENT");
    RCHK(hipSetDevice(e->device));
    DevLock lock(e);
    if (mode == TRS_STEP_LAUNCH) {
        if (!e->res) return TRS_OK;
        int rc = resident_quiesce(e);
        e->res->enabled = false;
        return rc;
    }
    if (!e->track_loaded) return trs_internal_fail(TRS_ERR_STATE, "no track loaded");
    if (!e->res) e->res = new (std::nothrow) Resident();
    if (!e->res) return trs_internal_fail(TRS_ERR_NOMEM, "out of memory");
    int rc = ensure_resident(e);
    if (rc) return rc;
    Resident* R = e->res;
    { int rf = worker_fits(e); if (rf) return rf; }
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_worker_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_physics_worker_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (idle_us > 0) R->idle_us = (unsigned)std::min(idle_us, 1000000);
    if (!R->enabled) { R->base = R->seen_done = e->step_count; host_store(&R->mb->posted, e->step_count); }
    R->enabled = true;
    R->fell_back = false;
    R->retry_ms = kRetryMs0;
    return TRS_OK;
}
