// trsim_hip.hip — libtrsim.so: the gfx950 (MI355X, CDNA4) kernels and the C ABI of include/trsim.h.
//
//   trs_step_kernel     camera on: ONE launch per env step on ONE stream.  768-thread workgroups (12 wave64) own a
//     contiguous range of envs and are wave-specialised:
//       physics team (waves 8..11, one wave per env, no workgroup barrier): SoA state load, bicycle-model step
//         (include/trsim_spec.h), binary64 L1 scan of the LDS-resident track with a wave64 DPP argmin
//         (= reference LocationTracker.__find_closest, components/track_data_process.py:89-101), y / cte / done /
//         reward, state store, one float4 of camera parameters per env into a small ring;
//       raster team (waves 0..7): LDS holds the packed 2-bit surface-class map (LDS offset 0, odd row pitch), the
//         per-row camera table and the per-row fogged palette; a thread owns a 4-pixel column group and walks rows:
//         1 packed-fp32 fma per pixel, saturating convert + min (= floor + clamp), mad_u24 addressing, one LDS map
//         read, bit-field extract, one LDS palette read; 4 pixels -> 12 bytes via v_perm, so one wave-instruction
//         stores 768 contiguous bytes (6 full 128-B lines).  Bound: HBM writes (57,600 B per env-step at 120x160).
//     Inside a multi-step call the raster team renders step t-1 while the physics team computes step t; the host
//     opens the call with a physics-only launch and closes it with a raster-only launch (the lag never leaves the
//     call).  A single-step call is one launch: the raster team writes the rows that need no pose (sky, far ground)
//     and then waits per env for the physics team's progress counter.  All LDS staging is LDS-DMA before one barrier.
//   trs_physics_kernel  camera off (BASELINE config 2): the same wave-per-env routine, 4 envs per 256-thread
//     workgroup, K steps per launch, no synchronisation after the one-time staging of the track into LDS.
//   trs_locate_kernel   batched LocationTracker for arbitrary binary64 query points.
//
// Rejected on measurements: all-wave fused phases; two kernels on three streams + hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (spec rule R1: no implicit FMA contraction).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/trsim.h"
#include "../../include/trsim_spec.h"
#include "trsim_device.hpp"
#include "trsim_env.hpp"
#include "trsim_internal.hpp"
#include "trsim_tables.hpp"


#define TRS_EXPORT extern "C" __attribute__((visibility("default")))

namespace {



#ifndef TRS_STAMPS
#define TRS_STAMPS 0   /* 1 = diagnostic build: s_memtime stamps per phase into stats[8..] (scripts/stamps.sh), never shipped */
#endif
#if TRS_STAMPS
#define STAMP(slot)                                                                                         \
    do {                                                                                                    \
        if (blockIdx.x == 7 && (threadIdx.x == 0 || threadIdx.x == kRasterStampThread)) {                                                   \
            unsigned long long _t;                                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                      \
            STAMP_STATS[8 + (threadIdx.x ? 24 : 0) + (slot)] = _t;                                                      \
        }                                                                                                   \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif


#undef STAMP_STATS
#define STAMP_STATS sp.ra.stats
// ---------------------------------------------------------------------------------------------
// Fused, wave-specialised step kernel (camera on).  One launch per env step, one stream.
//   raster team  = waves 0..7  (512 threads): rasterises the frame of camera-ring slot `r_slot`
//   physics team = waves 8..11 (one wave per env, no workgroup barriers): advances the envs one step and
//                  writes their camera parameters to ring slot `p_slot`
//   seq = 1  (single-step call): raster waits for this launch's physics and renders the SAME step
//   seq = 0  (inside a multi-step call): raster renders the PREVIOUS step while physics computes the next one;
//            the host issues a physics-only first launch and a raster-only last launch, so the lag never
//            leaves the call.

struct SParams {
    PParams ph;                         // physics side (blob = px|py|pz|tan image; cam = ring base)
    RParams ra;                         // raster side
    uint8_t* img0; uint8_t* img1;       // frame of absolute step s goes to img[s & 1]
    float* dep0; float* dep1;           // z-depth frames, same double buffering (NULL unless cfg.depth)
    int n_phys;                         // physics steps this launch advances (0 = raster-only flush)
    int r_first, r_last;                // raster renders launch-local steps r_first..r_last; -1 = the step before this
                                        // launch (camera parameters from the global ring, written by the previous launch)
    unsigned step_base;                 // absolute index of this launch's physics step 0
    int lds_off_phys, lds_off_cam, lds_off_prog, cam_stride;   // LDS: physics image, float4 lcam[n_phys][cam_stride], int pprog[cam_stride]
    int skip_uniform;                   // 1: the target frame buffer already holds this palette's uniform rows (sky, beyond the far plane: they depend on neither the
                                        // pose nor the step) of every env — an earlier step wrote them and nothing has touched them since: only the rows that see the
                                        // track are written.  Set by the closed pilot loop only (trs_internal_step_launch); every other step path writes whole frames.
    FParams fp;
};

// Physics-only envs (BASELINE config 2): the same barrier-free wave-per-env routine, 4 envs per 256-thread workgroup,
// K steps per launch.  The track image is staged once per launch; nothing is synchronised after that.
constexpr int kPhysBlock = 256;

__global__ __launch_bounds__(kPhysBlock) void trs_physics_kernel(const PParams p)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the track image goes global -> LDS by LDS-DMA (1 KB per wave instruction, no registers, no ds_write pass)
    stage_lds_dma(p.blob, p.blob_bytes, (unsigned)(uintptr_t)smem, wave, kPhysBlock / 64, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4* const lsink = reinterpret_cast<float4*>(smem + p.off_scratch);          // per-wave sinks for the camera hand-off slots
    int* const psink = reinterpret_cast<int*>(smem + p.off_scratch + (kPhysBlock / 64) * 16);
    __syncthreads();
    const int e = blockIdx.x * (kPhysBlock / 64) + wave;
    if (e >= p.n_envs) return;
    EnvRegs st;
    env_load(p, e, st);
    for (int k = 0; k < p.n_steps; ++k)
        env_step<true>(p, smem, e, st, p.step_off + (uint32_t)k, k, p.cam, &lsink[wave], &psink[wave], lane);   // (no raster wave beside this one: the select forms, see spec_sincos_sel)
    env_store(p, e, st, lane);
}


template <bool DEPTH, bool DYN, bool HILLS = false>      // HILLS: a track with elevation (its own instantiations: the flat kernels' loops do not change for it)
__global__ __launch_bounds__(kBlock) void trs_step_kernel(const SParams sp)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool raster_team = tid < kRasterThreads;
    const RParams& p = sp.ra;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;        // LDS byte address of the dynamic segment
    if (lds0 != 0u) {                                       // the map addressing below assumes LDS offset 0 (no static __shared__ in this kernel:
        if (tid == 0 && blockIdx.x == 0) {                  // tests/test_build_lint.py checks .group_segment_fixed_size == 0); refusing is LOUD:
            atomicAdd(&p.stats[2], 1ull);                   // every later synchronisation of the handle fails (trsim::check_fault)
            __hip_atomic_store(p.fault, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    STAMP(0);
    const int e_begin = blockIdx.x * p.envs_per_wg;
    const int e_end = min(e_begin + p.envs_per_wg, p.n_envs);
    float4* const lcam = reinterpret_cast<float4*>(smem + sp.lds_off_cam);     // [n_phys][cam_stride]
    int* const pprog = reinterpret_cast<int*>(smem + sp.lds_off_prog);         // [cam_stride] physics steps finished per env
    const bool rendering = sp.r_last >= sp.r_first;
    for (int j = tid; j < sp.cam_stride; j += kBlock) pprog[j] = 0;
    // tracks with elevation (HILLS instantiations only; the host's launch_step lays the same regions out): behind the progress counters float lpitch[n_phys + 1][cam_stride]
    // (view pitch per step and env), then the batch's row tables and the raster team's barrier counter; the block of constants sits behind the raster image in memory
    const int lds_off_pitch = sp.lds_off_prog + sp.cam_stride * 4 + 16;
    const int lds_off_hill = (lds_off_pitch + (max(sp.n_phys, 1) + 1) * sp.cam_stride * 4 + 15) & ~15;
    float* const lpitch = reinterpret_cast<float*>(smem + lds_off_pitch);
    int* const hbar = reinterpret_cast<int*>(smem + lds_off_hill + hill_batch(p.H) * hill_table_bytes(p.H));
    const trsim::HillBlock* const hill = HILLS ? reinterpret_cast<const trsim::HillBlock*>(p.blob + trsim::hill_block_offset(p.blob_bytes)) : nullptr;
    if (HILLS && tid == 0) *hbar = 0;
    if constexpr (DYN) {
        if (tid < 32) reinterpret_cast<int*>(smem + sp.fp.lds_off + 4 * p.H * 16)[tid] = 0;   // esum[2][4][3], dbar
        dyn_stage_tables(smem, sp.fp, p.H, tid, kBlock, reinterpret_cast<const uint32_t*>(p.blob + p.off_pal));                   // OpenCV's reciprocals + the in-range byte masks (behind the prologue's barrier)
    }
    // ---- prologue: everything is staged global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction moves
    // 64 lanes x 16 B = 1 KB, lane-linear, no registers and no ds_write pass), all requests are in flight together and
    // one workgroup barrier closes the stage.  Raster waves: the class map + row tables (one linear image at LDS offset
    // 0); physics waves: the track image.  The previous step's poses ride along through a register. ----
    const int pw = wave - kRasterThreads / 64;                                // physics wave index (< 0 for the raster team)
    float4* const lcam_prev = lcam + max(sp.n_phys, 1) * sp.cam_stride;       // [cam_stride] poses of step_base - 1
    {
        const bool has_c = raster_team && rendering && sp.r_first < 0 && tid < e_end - e_begin;
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* const cam_prev = sp.ph.cam + (size_t)((sp.step_base - 1u) & (kRing - 1)) * sp.ph.n_envs;
        if (has_c) cv = cam_prev[e_begin + tid];
        if (raster_team) {
            if (rendering) stage_lds_dma(p.blob, p.blob_bytes, 0u, wave, kRasterThreads / 64, lane);
        } else if (sp.n_phys > 0) {
            stage_lds_dma(sp.ph.blob, sp.ph.blob_bytes, (unsigned)sp.lds_off_phys, pw, kPhysWaves, lane);
        }
        if (has_c) lcam_prev[tid] = cv;
        if (raster_team && rendering && sp.r_first < 0)
            for (int j = tid + kRasterThreads; j < e_end - e_begin; j += kRasterThreads) lcam_prev[j] = cam_prev[e_begin + j];
        if (HILLS && raster_team && rendering && sp.r_first < 0) {             // ... and its view pitch (a track with elevation)
            const float* const pitch_prev = hill->cam_pitch + (size_t)((sp.step_base - 1u) & (kRing - 1)) * sp.ph.n_envs;
            for (int j = tid; j < e_end - e_begin; j += kRasterThreads) lpitch[max(sp.n_phys, 1) * sp.cam_stride + j] = pitch_prev[e_begin + j];
        }
    }
    STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's DMA pieces have landed
    __syncthreads();
    STAMP(2);

    // ---- physics team: one wave per env, no workgroup synchronisation; runs up to n_phys steps ahead of the raster ----
    if (!raster_team) {
        if (sp.n_phys <= 0) return;
        float4* const ring = sp.ph.cam;
        const unsigned char* const lphys = smem + sp.lds_off_phys;
        if (p.envs_per_wg <= kPhysWaves) {
            // at most one env per physics wave: its state lives in registers for all the steps of this launch
            const int e = e_begin + pw;
            if (e < e_end) {
                const int j = e - e_begin;
                EnvRegs st;
                env_load(sp.ph, e, st);
                for (int k = 0; k < sp.n_phys; ++k) {
                    const uint32_t t = sp.step_base + (uint32_t)k;
                    env_step(sp.ph, lphys, e, st, t, k, ring + (size_t)(t & (kRing - 1)) * sp.ph.n_envs, &lcam[k * sp.cam_stride + j], &pprog[j], lane,
                             hill, HILLS ? &lpitch[k * sp.cam_stride + j] : nullptr);
                }
                env_store(sp.ph, e, st, lane);
            }
        } else {
            // several envs per wave: step-major order keeps every env's raster fed; state goes through L2 between steps
            for (int k = 0; k < sp.n_phys; ++k) {
                const uint32_t t = sp.step_base + (uint32_t)k;
                float4* const cam_out = ring + (size_t)(t & (kRing - 1)) * sp.ph.n_envs;
                for (int e = e_begin + pw; e < e_end; e += kPhysWaves) {
                    const int j = e - e_begin;
                    EnvRegs st;
                    env_load(sp.ph, e, st);
                    env_step(sp.ph, lphys, e, st, t, k, cam_out, &lcam[k * sp.cam_stride + j], &pprog[j], lane,
                             hill, HILLS ? &lpitch[k * sp.cam_stride + j] : nullptr);
                    env_store(sp.ph, e, st, lane);
                }
                if (sp.n_phys > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores have reached L2 before the next step reloads them
            }
        }
        STAMP(3);
        return;
    }
    if (!rendering) return;

    // ---- raster team ----
    // A thread owns one 4-pixel column group (u0 fixed) and walks image rows, so the pixel-centre offsets uf are
    // loop constants.  The first `uni_rows` rows (sky, ground beyond the far plane) have four equal class colours: they
    // need neither the class map nor the camera pose and are written FIRST, while the map is still on its way and (in a
    // single-step call) while the physics team integrates.  Per pixel of the other rows: 1 packed fma (gx,gz), 2
    // saturating converts + 2 min (= floor + clamp), 3 address ops, 1 LDS map read, shift + bit-field extract, 1 palette
    // address op, 1 LDS palette read.
    const RasterThread rth = raster_thread(p, smem, tid);
    [[maybe_unused]] const f2v* const lrow = rth.lrow;
    [[maybe_unused]] const float* const lrowdepth = rth.lrowdepth;
    [[maybe_unused]] const unsigned gwm1 = rth.gwm1, ghm1 = rth.ghm1, pitch = rth.pitch;
    [[maybe_unused]] const int cg = rth.cg, vstart = rth.vstart, col_off = rth.col_off;
    [[maybe_unused]] const f2v ufa = rth.ufa, ufb = rth.ufb, ufc = rth.ufc, ufd = rth.ufd;
    [[maybe_unused]] const size_t row_bytes = (size_t)rth.row_bytes;
    for (int sidx = sp.r_first; sidx <= sp.r_last; ++sidx) {
    const unsigned abs_step = sp.step_base + (unsigned)sidx;                  // sidx = -1: the step before this launch
    uint8_t* const img = (abs_step & 1u) ? sp.img1 : sp.img0;
    float* const dep = (abs_step & 1u) ? sp.dep1 : sp.dep0;
    // (Writing the uniform rows of ALL the workgroup's envs first — so that more bytes are in flight while a single-step call
    // waits for its poses — was measured in round 2: the single-step call did not get shorter (16.7 us both ways) and the
    // pipelined launch got 6 % LONGER (13.5 -> 14.4 us at 1024 envs: the frames are then written in two sweeps over all envs
    // instead of one contiguous 57.6 KB stream per env).  Env by env it is.)
    for (int e = e_begin; e < e_end; ++e) {
        if constexpr (HILLS) {
            // ---- a track with elevation: the envs go through in batches — poses and view pitches of the batch, its row tables between two team barriers
            // (trsim_device.hpp, hill_batch_build), then the same row loop on each env's table
            const int HB = hill_batch(p.H);
            if ((e - e_begin) % HB != 0) continue;                            // the batch leader's iteration does the work
            const int nbatch = (e_end - e_begin + HB - 1) / HB;
            const int it = (sidx - sp.r_first) * nbatch + (e - e_begin) / HB; // batches so far (the same in every wave)
            const int nb = min(HB, e_end - e);
            float4 cams[kHillBatchMax]; float Pv[kHillBatchMax];
#pragma unroll
            for (int bi = 0; bi < kHillBatchMax; ++bi) {
                cams[bi] = make_float4(0.f, 0.f, 0.f, 1.f); Pv[bi] = 0.f;
                if (bi < nb) {
                    const int j = e - e_begin + bi;
                    if (sidx < 0) { cams[bi] = lcam_prev[j]; Pv[bi] = lpitch[max(sp.n_phys, 1) * sp.cam_stride + j]; }
                    else {
                        while (__hip_atomic_load(&pprog[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < sidx + 1) __builtin_amdgcn_s_sleep(2);
                        cams[bi] = lcam[sidx * sp.cam_stride + j]; Pv[bi] = lpitch[sidx * sp.cam_stride + j];
                    }
                }
            }
            auto never = [](bool) { return false; };                          // (a launch has no abort: every raster wave arrives)
            (void)hill_batch_build(p, smem, (unsigned)lds_off_hill, Pv, nb, hbar, it * 2 * (kRasterThreads / 64), tid, lane, never);
#pragma unroll
            for (int bi = 0; bi < kHillBatchMax; ++bi)
                if (bi < nb)
                    raster_hill_frame<DEPTH>(p, rth, smem, (unsigned)lds_off_hill, bi, hbar, it * 2 * (kRasterThreads / 64), frame_desc<DEPTH>(p, img, dep, e + bi), cams[bi]);
            continue;
        }
        if constexpr (DYN) {
            // ---- dynamic brightness behind the rasteriser: the frame's own mean over rows [w0, w1) only needs the class of
            // every pixel there, so, for up to kDynBatch envs at a time: (A) classify those rows once (classes kept in
            // registers), sum the RAW colours per channel, reduce over the team; (B) every thread filters entries of the
            // per-env palettes with each frame's delta; (C) shade all rows from those palettes.  No extra pass over HBM, each
            // pixel classified once, two team barriers per batch.
            if ((e - e_begin) % kDynBatch != 0) continue;                     // the batch leader does the work
            const int nbatch = (e_end - e_begin + kDynBatch - 1) / kDynBatch;
            const int it = (sidx - sp.r_first) * nbatch + (e - e_begin) / kDynBatch;   // batches so far (same in every wave)
            float4 cams[kDynBatch];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
                const int eb = e + bi;
                cams[bi] = make_float4(0.f, 0.f, 0.f, 1.f);
                if (eb < e_end) {
                    const int j = eb - e_begin;
                    if (sidx < 0) cams[bi] = lcam_prev[j];
                    else {
                        while (__hip_atomic_load(&pprog[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < sidx + 1) __builtin_amdgcn_s_sleep(2);
                        cams[bi] = lcam[sidx * sp.cam_stride + j];
                    }
                }
            }
            (void)raster_dyn_batch<DEPTH>(p, sp.fp, rth, smem, cams, min(kDynBatch, e_end - e), img, dep, e, it, tid, lane, [](bool) { return false; });   // a launch has no abort: every raster wave arrives
            continue;
        }
        // rows with four equal class colours need no map lookup and no pose: they are written first, and in a single-step
        // call while the physics team still integrates
        const FrameDesc fd = frame_desc<DEPTH>(p, img, dep, e);
        if (!sp.skip_uniform) raster_uniform_rows<DEPTH>(p, rth, fd);
        // -- rows that see the track
        float4 cam;
        const int j = e - e_begin;
        if (sidx < 0) {
            cam = lcam_prev[j];                                               // written by the previous launch, staged in the prologue
        } else {                                                              // wait until the physics team has finished this step of env j
            while (__hip_atomic_load(&pprog[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < sidx + 1) __builtin_amdgcn_s_sleep(2);
            cam = lcam[sidx * sp.cam_stride + j];
        }
        raster_ground_rows<DEPTH>(p, rth, fd, cam);
    }
    }
    STAMP(5);
}

#undef STAMP_STATS
// ---------------------------------------------------------------------------------------------
// Image path (SURVEY rows a10, a11 colour masks, a13): ImgPreprocessing.__process without the Canny layer
// (components/img_preprocessing.py:37-74,81-102) and the pilot's float32/255 normalisation
// (components/keras_pilot.py:49-55).  One 256-thread workgroup per frame; a lane handles 4 pixels (12 B).
//   pass 1: exact integer channel sums over rows 40..118 (cv2.mean, :88), wave reduce -> LDS -> delta (binary64)
//   pass 2: binary32 trim in numpy's operation order (:92-99), OpenCV 8-bit RGB->HSV + inRange masks (:65-74),
//           masks written over their destination channels (:57-63); the second read of the frame hits L2
// Bound: HBM, 2 x H*W*3 bytes per frame (one read, one write).
struct PreParams {
    const uint8_t* src; uint8_t* dst;
    const int* hsv_tab;                 // [512]: sdiv[256] | hdiv[256] (OpenCV fixed-point reciprocal tables)
    int n_img, H, W, gpr, gpe, r0, r1;  // 4-pixel groups per row / per frame; brightness rows [r0, r1)
    int dynamic, color, n_filters;
    float contrast, offset;
    double baseline;
    unsigned lo[4], hi[4];              // packed h | s<<8 | v<<16
    int dst_ch[4];
    int edge, edge_low, edge_high, edge_ch;   // Canny layer (edge kernel only)
    int off_mag, off_map, off_tab;      // edge kernel: offsets of the gradient magnitudes / edge map behind the trimmed frame; LDS offset of the tables
    unsigned char* scratch;             // edge kernel, frames too large for LDS: per-workgroup work arrays in global memory (L2 resident)
    size_t scratch_stride;
};

// four pixels (r, g, b, x) -> the 12 bytes of their group (the rasteriser's byte shuffles)
__device__ __forceinline__ u3v pack_rgb4(unsigned P0, unsigned P1, unsigned P2, unsigned P3)
{
    return u3v{__builtin_amdgcn_perm(P1, P0, 0x04020100u), __builtin_amdgcn_perm(P2, P1, 0x05040201u), __builtin_amdgcn_perm(P3, P2, 0x06050402u)};
}

__device__ __forceinline__ unsigned sum_bytes(unsigned w, unsigned mask, unsigned acc) { return __builtin_amdgcn_sad_u8(w & mask, 0u, acc); }

constexpr int kPreGpt = 19;   // 4-pixel groups a thread of trs_preprocess_kernel can hold between its two passes (dynamic brightness): 4,800 groups on 256 threads, 19,200 on 1024

// REGS: the instantiation that may hold a frame in registers between its two passes (dynamic brightness, see below and trs_preprocess's dispatch); the other
// one keeps 42 registers per thread - the occupancy the single-pass variants and the mask arithmetic live on (with the frame in registers: trim 21.5 -> 22.4 us
// per 1024 frames, trim + HSV masks 29.5 -> 30.2, and + 11 % on 1024-thread workgroups with masks; profiles/r04_image_path_regs.txt).
template <bool REGS>
__global__ __launch_bounds__(1024) void trs_preprocess_kernel(const PreParams p)   // 256 threads per frame at 120x160, 1024 for frames of 8,192+ pixel groups
{
    __shared__ int s_tab[512];
    __shared__ unsigned s_part[16][3];
    // per-value tables replace per-pixel arithmetic (the masks variant was VALU bound at ~80 integer ops per pixel):
    // s_trim[x] = the trim of byte value x for this frame's delta; s_rng[c][x] = bit f set when value x of component c
    // (h, s, v) lies inside the range of the filter that owns channel f's mask, as a BYTE mask -> the AND of three lookups is a pixel's masks (mask_pixel)
    __shared__ unsigned s_trim[256];
    __shared__ unsigned s_rng[3][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const size_t frame_bytes = (size_t)p.gpe * 12;
    for (int i = tid; i < 512; i += nthreads) s_tab[i] = p.hsv_tab[i];
    unsigned sel = 0;                                                       // the channels that carry a mask
    for (int i = tid; i < 768; i += nthreads) (&s_rng[0][0])[i] = range_byte_entry(p.lo, p.hi, p.dst_ch, p.n_filters, i, nullptr);
    (void)range_byte_entry(p.lo, p.hi, p.dst_ch, p.n_filters, 0, &sel);
    auto trim_table = [&](float deltaf) {                                   // this frame's trim of every byte value, in numpy's operation order (:92-99)
        if (tid < 256) {
            float x = (float)tid;
            if (p.dynamic) x = x + deltaf;
            x = x - p.offset;
            x = x * p.contrast;
            x = x + p.offset;
            x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
            s_trim[tid] = (unsigned)(int)x;
        }
    };
    // Without dynamic brightness (the reference's default, config.py) nothing depends on the frame's own mean: the table is made once and
    // the frames stream through one pass, no channel sums and no barrier per frame (round 3).
    if (!p.dynamic) trim_table(0.0f);
    __syncthreads();
    for (int img = blockIdx.x; img < p.n_img; img += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src) + (size_t)img * frame_bytes, 0, (int)frame_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dst + (size_t)img * frame_bytes, 0, (int)frame_bytes, 0x00020000);
        // Dynamic brightness needs the frame's own mean before its first pixel can be trimmed.  Where the frame fits the workgroup's registers (at most
        // kPreGpt groups per thread: 120x160 on 256 threads, 240x320 on 1024) it is read from memory ONCE - every load in flight together - and both the
        // channel sums and the second pass work on the registers (round 4, late: the sums pass used to read the brightness rows from memory and the
        // second pass the whole frame again: 6 of this variant's 36 us per 1024 frames).  Larger frames keep the two passes over memory.
        const bool in_regs = REGS && p.dynamic && (p.gpe + nthreads - 1) / nthreads <= kPreGpt;
        u3v R[REGS ? kPreGpt : 1];
        if constexpr (REGS) {
            if (in_regs) {
#pragma unroll
                for (int k = 0; k < kPreGpt; ++k) R[k] = __builtin_amdgcn_raw_buffer_load_b96(rs, (tid + k * nthreads) * 12, 0, 0);   // past the frame: zeros (the descriptor's bounds)
            }
        }
        if (p.dynamic) {
        // ---- pass 1: channel sums over the brightness rows ----
        unsigned sr = 0, sg = 0, sb = 0;
        auto add_group = [&](const u3v w) {
            // bytes: w.x = R0 G0 B0 R1 | w.y = G1 B1 R2 G2 | w.z = B2 R3 G3 B3
            sr = sum_bytes(w.x, 0xFF0000FFu, sr); sr = sum_bytes(w.y, 0x00FF0000u, sr); sr = sum_bytes(w.z, 0x0000FF00u, sr);
            sg = sum_bytes(w.x, 0x0000FF00u, sg); sg = sum_bytes(w.y, 0xFF0000FFu, sg); sg = sum_bytes(w.z, 0x00FF0000u, sg);
            sb = sum_bytes(w.x, 0x00FF0000u, sb); sb = sum_bytes(w.y, 0x0000FF00u, sb); sb = sum_bytes(w.z, 0xFF0000FFu, sb);
        };
        if (in_regs) {
            if constexpr (REGS) {
#pragma unroll
                for (int k = 0; k < kPreGpt; ++k) {
                    const int g = tid + k * nthreads;
                    if (g >= p.r0 * p.gpr && g < p.r1 * p.gpr) add_group(R[k]);
                }
            }
        } else {
            for (int g = p.r0 * p.gpr + tid; g < p.r1 * p.gpr; g += nthreads) add_group(__builtin_amdgcn_raw_buffer_load_b96(rs, g * 12, 0, 0));
        }
        sr = wave_sum_dpp(sr); sg = wave_sum_dpp(sg); sb = wave_sum_dpp(sb);   // (DPP: totals in lane 63)
        if (lane == 63) { s_part[wave][0] = sr; s_part[wave][1] = sg; s_part[wave][2] = sb; }
        __syncthreads();
        if (tid < 256) {                                                    // every thread of the table evaluates the same binary64 expression itself (exact integer
            const double cnt = (double)(p.r1 - p.r0) * (double)p.W;         // totals, any order): no serial pass by one thread, no barrier for the delta (round 3)
            double cur = 0.0;
            for (int ch = 0; ch < 3; ++ch) {
                unsigned long long tot = 0;
                for (int w = 0; w < nwaves; ++w) tot += s_part[w][ch];
                cur = cur + (cnt > 0 ? (double)tot / cnt : 0.0);
            }
            cur = cur + 0.0;
            trim_table((float)((p.baseline - cur) / 3));
        }
        __syncthreads();
        }
        // ---- pass 2: trim, masks, merge ----
        // (p.color is tested once per 4-pixel group, not per pixel: the four pixels' chains of dependent table lookups — trim, OpenCV's
        // reciprocal tables, the range bits — then interleave instead of running one behind the other)
        auto out_group = [&](const u3v w, int g) {
            // bytes: w.x = R0 G0 B0 R1 | w.y = G1 B1 R2 G2 | w.z = B2 R3 G3 B3 -> four trimmed pixels (r, g, b, 0)
            auto tr = [&](unsigned word, int k) -> unsigned { return s_trim[(word >> (8 * k)) & 255u]; };
            const unsigned t00 = tr(w.x, 0), t01 = tr(w.x, 1), t02 = tr(w.x, 2), t10 = tr(w.x, 3), t11 = tr(w.y, 0), t12 = tr(w.y, 1);
            const unsigned t20 = tr(w.y, 2), t21 = tr(w.y, 3), t22 = tr(w.z, 0), t30 = tr(w.z, 1), t31 = tr(w.z, 2), t32 = tr(w.z, 3);
            unsigned P0, P1, P2, P3;
            if (p.color) {                                                  // the components go to the masks as they come out of the trim table (round 4: not packed and unpacked again)
                P0 = mask_pixel_rgb((int)t00, (int)t01, (int)t02, s_tab, &s_rng[0][0], sel); P1 = mask_pixel_rgb((int)t10, (int)t11, (int)t12, s_tab, &s_rng[0][0], sel);
                P2 = mask_pixel_rgb((int)t20, (int)t21, (int)t22, s_tab, &s_rng[0][0], sel); P3 = mask_pixel_rgb((int)t30, (int)t31, (int)t32, s_tab, &s_rng[0][0], sel);
            } else {
                P0 = t00 | (t01 << 8) | (t02 << 16); P1 = t10 | (t11 << 8) | (t12 << 16);
                P2 = t20 | (t21 << 8) | (t22 << 16); P3 = t30 | (t31 << 8) | (t32 << 16);
            }
            const u3v out = pack_rgb4(P0, P1, P2, P3);
            __builtin_amdgcn_raw_buffer_store_b96(out, rd, g * 12, 0, 0);                         // (a group past the frame: dropped by the descriptor's bounds)
        };
        if (in_regs) {
            if constexpr (REGS) {
#pragma unroll
                for (int k = 0; k < kPreGpt; ++k) {
                    const int g = tid + k * nthreads;
                    if (g < p.gpe) out_group(R[k], g);
                }
            }
        } else {
            for (int g = tid; g < p.gpe; g += nthreads) out_group(__builtin_amdgcn_raw_buffer_load_b96(rs, g * 12, 0, 0), g);
        }
        if (p.dynamic) __syncthreads();   // s_part / s_trim are rewritten for the next frame of this workgroup
    }
}

// ImgPreprocessing with the Canny edge layer (components/img_preprocessing.py:37-54,76-79): cv2.Canny(img, a, b) on the trimmed
// 3-channel frame, OpenCV's algorithm (see oracle/trsim_oracle.c canny_u8c3 for the statement).  One 1024-thread workgroup per
// frame; LDS holds the trimmed frame (H*W*3 B), the gradient magnitudes with a zero border ((H+2) x (W+8) int16) and the
// edge map (H*W B: direction class, then 0 = weak / 1 = no / 2 = edge).  Hysteresis = repeated 8-neighbour sweeps until
// a block-wide OR reports no change.  Frames up to ~26,000 pixels (LDS).
// Round 3 (counters first, profiles/r03_image_path.txt: the kernel is bound by instruction ISSUE — its SIMDs issue ~100 % of
// the launch, 212 vector + 85 scalar instructions per pixel — not by LDS (11 % busy) or memory): the instruction count per
// pixel was cut, phase by phase (timing-only builds -DTRS_EDGE_ABLATE: Sobel 60 of 154 us, output 44):
//   * trim: one table of this frame's trim of every byte value (256 threads compute it once) instead of 9 operations per byte;
//   * Sobel: in binary32 — every value is an integer below 2^24, so the arithmetic is exact and |x| is a free source modifier —
//     on a thread that walks DOWN its 4-pixel column group (one new row of 18 values per output row instead of three), with the
//     separable form (column sums t + 2m + b and differences b - t shared by the 4 pixels); the direction class from the same
//     exact comparisons (|ys| 2^15 < |xs| 13573 etc.: all products below 2^24, the 67.5-degree test as
//     fma(|xs|, -2^16, |ys| 2^15) > |xs| 13573, whose left side is 2^15 (|ys| - 2 |xs|));
//   * output: the in-range tests of all (<= 4) filters from three table lookups (as trs_preprocess_kernel does).
#ifndef TRS_EDGE_ABLATE
#define TRS_EDGE_ABLATE 0   /* timing-only diagnostic builds of trs_preprocess_edge_kernel, never shipped (wrong results): 1 = no Sobel phase, 3 = no hysteresis, 4 = no output phase, 6 = no suppression compare (every pixel "no edge") */
#endif
constexpr int kEdgeBlock = 1024;          // 16 waves per frame (512 until round 2: the phases are latency chains of LDS reads, twice the waves hide twice as much)
constexpr int kEdgeGpt = 5;              // 4-pixel groups per thread held in registers between the two passes over a frame: 5120 groups = 20,480 pixels (120x160: 4800 groups)
constexpr int kEdgeTables = 512 * 4 + 3 * (kEdgeBlock / 64) * 4 + 16 + 2 * 256 * 4 + 3 * 256 * 4;   // s_tab | s_part | s_delta | s_trim[2] | s_rng

__device__ __forceinline__ bool has_zero_byte(unsigned w) { return ((w - 0x01010101u) & ~w & 0x80808080u) != 0u; }

// Six pixels x three channels of trimmed-frame row `row` around the 4-pixel group at column x0 (pixels x0 - 1 .. x0 + 4, replicated
// at the frame's left / right edge) as binary32, from five aligned dword reads and one v_cvt_f32_ubyteN per value.  x0 % 4 == 0.
__device__ __forceinline__ void edge_row_f(const unsigned char* simg, int W, int row, int x0, float (&v)[3][6])
{
    const unsigned* base = reinterpret_cast<const unsigned*>(simg + ((size_t)row * W + x0) * 3);
    unsigned d[5];
    d[0] = x0 ? base[-1] : 0u;
    d[1] = base[0]; d[2] = base[1]; d[3] = base[2]; d[4] = base[3];           // base[3] past the row's last group is read but not used (see below)
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int bi = 1 + 3 * j + c;                                     // byte of the 20-byte window that starts 4 bytes in front of the group
            v[c][j] = (float)((d[bi >> 2] >> (8 * (bi & 3))) & 255u);
        }
    if (x0 == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][0] = v[c][1];
    }
    if (x0 + 4 >= W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][5] = v[c][4];
    }
}

// One output row of a 4-pixel group from its three input rows (t above, m the row itself, b below): per pixel the channel with the
// largest |dx| + |dy| (first on ties), its norm (-> mag, 4 x int16 = one 8-byte store) and its direction class (-> map).
__device__ __forceinline__ void edge_sobel_row(const float (&t)[3][6], const float (&m)[3][6], const float (&b)[3][6], short* mrow, unsigned char* maprow)
{
    // channel by channel (12 live column values instead of 36): column sums t + 2 m + b and differences b - t shared by the 4
    // pixels, then per pixel the running best (a strictly larger norm replaces it: the FIRST channel that reaches the maximum wins)
    float bn[4], xs[4], ys[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sv[6], dv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { sv[j] = __builtin_fmaf(2.0f, m[c][j], t[c][j]) + b[c][j]; dv[j] = b[c][j] - t[c][j]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = q + 1;
            const float dx = sv[j + 1] - sv[j - 1];
            const float dy = dv[j - 1] + __builtin_fmaf(2.0f, dv[j], dv[j + 1]);
            const float nr = __builtin_fabsf(dx) + __builtin_fabsf(dy);
            if (c == 0 || nr > bn[q]) { bn[q] = nr; xs[q] = dx; ys[q] = dy; }
        }
    }
    unsigned cls4 = 0u, mg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float ax = __builtin_fabsf(xs[q]), tg22x = ax * 13573.0f, ay = __builtin_fabsf(ys[q]) * 32768.0f;
        const float over = __builtin_fmaf(ax, -65536.0f, ay);                 // ay - (ax << 16), exact
        const unsigned cls = ay < tg22x ? 0u : (over > tg22x ? 1u : (xs[q] * ys[q] < 0.0f ? 3u : 2u));
        mg[q] = (unsigned)(int)bn[q];
        cls4 |= cls << (8 * q);
    }
    *reinterpret_cast<uint2*>(mrow) = make_uint2(mg[0] | (mg[1] << 16), mg[2] | (mg[3] << 16));
    *reinterpret_cast<unsigned*>(maprow) = cls4;
}

// SCRATCH = false: the three whole-frame work arrays live in LDS (frames up to ~26,000 pixels).  SCRATCH = true: they live in
// a per-workgroup global scratch that stays in L2 (any frame size, e.g. config 5's 240x320); only the tables are in
// LDS.  Same code, same results; __syncthreads() orders the workgroup's global accesses between the phases.
template <bool SCRATCH>
__global__ __launch_bounds__(kEdgeBlock) void trs_preprocess_edge_kernel(const PreParams p)
{
    unsigned char* const work = SCRATCH ? p.scratch + (size_t)blockIdx.x * p.scratch_stride : smem;
    unsigned char* const simg = work;
    short* const mag = reinterpret_cast<short*>(work + p.off_mag);           // pixel (x, y) at mag[(y + 1) * MP + x + 4]: a group's 4 values are 8-byte aligned
    unsigned char* const map = work + p.off_map;
    int* const s_tab = reinterpret_cast<int*>(smem + p.off_tab);
    unsigned* const s_part = reinterpret_cast<unsigned*>(s_tab + 512);       // [2][3] channel sums of the brightness rows, by frame parity: the waves ADD their totals (LDS atomics;
                                                                             // round 4: the delta phase's 16-lane 64-bit shuffle reductions are gone; < 2^24 per channel)
    float* const s_delta = reinterpret_cast<float*>(s_part + 3 * (kEdgeBlock / 64));
    unsigned* const s_trim2 = reinterpret_cast<unsigned*>(s_delta + 4);       // [2][256] the trim of every byte value, by frame parity (the next frame's table is made during this frame's output phase)
    unsigned* const s_rng = s_trim2 + 512;                                   // [3][256] byte ch = 0xFF: value x of component c (h, s, v) lies inside the range of the filter that owns channel ch's mask (mask_pixel)
    const int tid = threadIdx.x, lane = tid & 63;
    // Every phase takes its thread index through `fresh`: an empty asm the compiler cannot see through, so that what a phase derives from
    // the index (row / column splits, addresses) is computed where it is used.  Left alone, hipcc hoisted those values of ALL phases in
    // front of the frame loop and kept them alive across it: 65 spilled registers, 240 bytes of scratch per lane = 63 MB written at
    // the start of a launch and re-read at every phase (the first frame's first phase took 40 k clocks against 3.7 k for the others).
    auto fresh = [](int v) -> int { asm volatile("" : "+v"(v)); return v; };
    const int H = p.H, W = p.W, MP = W + 8, npx = H * W;
    const size_t frame_bytes = (size_t)p.gpe * 12;
    const int chunk_groups = kEdgeBlock * kEdgeGpt, nchunks = (p.gpe + chunk_groups - 1) / chunk_groups;
    const bool one_chunk = nchunks == 1;
    auto fetch = [&](const __amdgpu_buffer_rsrc_t& r, int c, u3v (&R)[kEdgeGpt]) {
        const int t = fresh(tid);
#pragma unroll
        for (int k = 0; k < kEdgeGpt; ++k) R[k] = __builtin_amdgcn_raw_buffer_load_b96(r, (c * chunk_groups + k * kEdgeBlock + t) * 12, 0, 0);   // past the frame: zeros (buffer bounds)
    };
    u3v R[kEdgeGpt];                                                          // this thread's groups of the frame (see below)
    if (one_chunk && (int)blockIdx.x < p.n_img) {                            // the workgroup's FIRST frame is requested before anything else: its memory latency runs under the table set-up below
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src) + (size_t)blockIdx.x * frame_bytes, 0, (int)frame_bytes, 0x00020000);
        fetch(r0, 0, R);
    }
    for (int i = tid; i < 512; i += kEdgeBlock) s_tab[i] = p.hsv_tab[i];
    unsigned sel = 0;                                                        // the channels that carry a mask (a later filter on a channel replaces an earlier one, :57-63)
    for (int i = tid; i < 768; i += kEdgeBlock) s_rng[i] = range_byte_entry(p.lo, p.hi, p.dst_ch, p.n_filters, i, nullptr);
    (void)range_byte_entry(p.lo, p.hi, p.dst_ch, p.n_filters, 0, &sel);
    // Sobel work items: (4-pixel column group, chunk of rows); the whole block works at once when the frame has <= 1024 / gpr chunks
    const int nchunk = max(1, kEdgeBlock / p.gpr), rows_per = (H + nchunk - 1) / nchunk;
#ifdef TRS_EDGE_STAMPS   /* diagnostic build: shader clocks per phase of workgroup 7, summed over its frames, printed at the end */
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define EDGE_STAMP(k) do { if (blockIdx.x == 7 && tid == 0) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); if ((k) > 0) ph[k] += tn - tprev; tprev = tn; } } while (0)
#else
#define EDGE_STAMP(k) do { } while (0)
#endif
    const int chunk_groups0 = kEdgeBlock * kEdgeGpt;
    auto chunk_sums = [&](const u3v (&R)[kEdgeGpt], int c, unsigned& sr, unsigned& sg, unsigned& sb) {   // this thread's groups of chunk c inside the brightness rows
        const int t = fresh(tid);
#pragma unroll
        for (int k = 0; k < kEdgeGpt; ++k) {
            const int g = c * chunk_groups0 + k * kEdgeBlock + t;
            if (g >= p.r0 * p.gpr && g < p.r1 * p.gpr) {
                const u3v w = R[k];
                sr = sum_bytes(w.x, 0xFF0000FFu, sr); sr = sum_bytes(w.y, 0x00FF0000u, sr); sr = sum_bytes(w.z, 0x0000FF00u, sr);
                sg = sum_bytes(w.x, 0x0000FF00u, sg); sg = sum_bytes(w.y, 0xFF0000FFu, sg); sg = sum_bytes(w.z, 0x00FF0000u, sg);
                sb = sum_bytes(w.x, 0x00FF0000u, sb); sb = sum_bytes(w.y, 0x0000FF00u, sb); sb = sum_bytes(w.z, 0xFF0000FFu, sb);
            }
        }
    };
    auto publish_sums = [&](unsigned sr, unsigned sg, unsigned sb, int par) {  // wave totals -> s_part[par] (read by the delta phase behind a barrier)
        sr = wave_sum_dpp(sr); sg = wave_sum_dpp(sg); sb = wave_sum_dpp(sb);   // (DPP: the wave's totals end in lane 63; the ds_bpermute shuffles were six dependent LDS round trips per frame)
        if (lane == 63) { atomicAdd(&s_part[par * 3], sr); atomicAdd(&s_part[par * 3 + 1], sg); atomicAdd(&s_part[par * 3 + 2], sb); }
    };
    auto make_trim_table = [&](int par_of_sums, unsigned* table) {           // one thread per byte value (tid < 256): delta from the frame's channel totals, then the trim
        const int t = fresh(tid);
        const double cnt = (double)(p.r1 - p.r0) * (double)W;
        double cur = 0.0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const unsigned tot = s_part[par_of_sums * 3 + ch];               // (a uniform LDS read: exact integer totals, any order of the adds)
            cur = cur + (cnt > 0 ? (double)tot / cnt : 0.0);
        }
        cur = cur + 0.0;
        const float deltaf = (float)((p.baseline - cur) / 3), off = p.offset, con = p.contrast;
        float x = (float)t;
        if (p.dynamic) x = x + deltaf;
        x = x - off; x = x * con; x = x + off;
        x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
        table[t] = (unsigned)(int)x;
    };
    bool table_ready = false;                                                 // uniform: s_trim2[par] already holds this frame's table (made during the previous frame's output phase)
    if (tid < 6) s_part[tid] = 0u;                                            // (behind the first barrier below before anyone adds)
    int par = 0;                                                              // parity of the current frame of this workgroup
    bool sums_ready = false;                                                  // uniform: s_part already holds this frame's sums
    for (int img = blockIdx.x; img < p.n_img; img += gridDim.x) {
        EDGE_STAMP(0);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src) + (size_t)img * frame_bytes, 0, (int)frame_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dst + (size_t)img * frame_bytes, 0, (int)frame_bytes, 0x00020000);
        // magnitudes outside the image are 0: the rows above / below and the columns left / right of it (the interior is overwritten)
        for (int i = fresh(tid); i < 2 * MP + 2 * H; i += kEdgeBlock) {
            int idx;
            if (i < MP) idx = i;
            else if (i < 2 * MP) idx = (H + 1) * MP + (i - MP);
            else { const int k = i - 2 * MP, y = k >> 1; idx = (y + 1) * MP + ((k & 1) ? W + 4 : 3); }
            mag[idx] = 0;
        }
        // ---- the frame: kEdgeGpt groups per thread in registers, every load issued before the first is used (a thread that loaded and used
        // its groups one after the other paid one memory round trip per group: the two passes over the frame were 29 % of the kernel).
        // A frame of up to 1024 x kEdgeGpt groups (every frame whose work arrays fit LDS) is read from memory ONCE — both passes work
        // on the registers — and the next frame's loads are issued as soon as the registers are free, in front of the Sobel phase.
        // Larger frames take the passes in chunks of 1024 x kEdgeGpt groups (the second pass reads L2). ----
        if (!one_chunk) fetch(rs, 0, R);                                    // (one-chunk frames: the first was requested at the kernel's start, the others during the previous frame)
        // ---- channel sums over the brightness rows -> delta (as trs_preprocess_kernel) ----
        // (one-chunk frames after the first: the sums were taken from the prefetched registers in the middle of the previous frame, see below -
        // this phase and its barrier were 12 % of the kernel, most of it waves waiting for each other right after the previous frame's last phase)
        if (!sums_ready) {
            unsigned sr = 0, sg = 0, sb = 0;
            for (int c = 0; c < nchunks; ++c) {
                if (c > 0) fetch(rs, c, R);
                chunk_sums(R, c, sr, sg, sb);
            }
            __syncthreads();                                                  // (the zeroing of s_part[par] is complete: kernel start, or the previous frame's delta phase)
            publish_sums(sr, sg, sb, par);
            __syncthreads();
        }
        EDGE_STAMP(1);
        // every thread of the table reads the three totals (the waves added theirs with LDS atomics: exact integers, any order) and evaluates the same
        // binary64 expression: no serial pass by one thread, no barrier for the delta, no reduction in this phase (round 3 reduced 16 partial sums per
        // channel here with 64-bit shuffles)
        unsigned* const s_trim = s_trim2 + par * 256;
        if (tid < 3) s_part[(par ^ 1) * 3 + tid] = 0u;                       // the NEXT frame's sums start from zero (their adders are behind the barriers below; that half was last read for the previous frame's table)
        if (!table_ready) {
            if (tid < 256) make_trim_table(par, s_trim);                     // this frame's trim of every byte value, in numpy's operation order (:92-99)
            __syncthreads();
        }
        EDGE_STAMP(2);
        // ---- trimmed frame -> LDS ----
        for (int c = 0, t = fresh(tid); c < nchunks; ++c) {
            if (!one_chunk) fetch(rs, c, R);
#pragma unroll
            for (int k = 0; k < kEdgeGpt; ++k) {
                const int g = c * chunk_groups + k * kEdgeBlock + t;
                if (g < p.gpe) {
                    const unsigned src[3] = {R[k].x, R[k].y, R[k].z};
                    unsigned out[3];
#pragma unroll
                    for (int k3 = 0; k3 < 3; ++k3) {
                        const unsigned v = src[k3];
                        out[k3] = s_trim[v & 255u] | (s_trim[(v >> 8) & 255u] << 8) | (s_trim[(v >> 16) & 255u] << 16) | (s_trim[v >> 24] << 24);
                    }
                    unsigned* d = reinterpret_cast<unsigned*>(simg + (size_t)g * 12);
                    d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
                }
            }
        }
        if (one_chunk && img + (int)gridDim.x < p.n_img) {                 // the next frame of this workgroup: on its way during the phases below
            const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.src) + (size_t)(img + gridDim.x) * frame_bytes, 0, (int)frame_bytes, 0x00020000);
            fetch(rn, 0, R);
        }
        __syncthreads();
        EDGE_STAMP(3);
        // ---- Sobel per channel, the channel with the largest |dx| + |dy| wins (first on ties) ----
        // mag <- the winner's norm, map <- its gradient direction class for the non-maximum suppression (OpenCV's fixed-point
        // tangents: 0 = compare left / right, 1 = up / down, 2 / 3 = the two diagonals), so that the suppression needs no second Sobel
        for (int item = fresh(tid); item < (TRS_EDGE_ABLATE == 1 ? 0 : p.gpr * nchunk); item += kEdgeBlock) {
            const int rc = item / p.gpr, cg = item - rc * p.gpr, x0 = cg * 4;
            const int y0 = rc * rows_per, y1 = min(H, y0 + rows_per);
            if (y0 >= y1) continue;
            float ra[3][6], rb[3][6], rc3[3][6];                              // three rows in rotating roles: no register copies between output rows
            edge_row_f(simg, W, y0 > 0 ? y0 - 1 : 0, x0, ra);
            edge_row_f(simg, W, y0, x0, rb);
            for (int y = y0; y < y1; y += 3) {
                edge_row_f(simg, W, y + 1 < H ? y + 1 : H - 1, x0, rc3);
                edge_sobel_row(ra, rb, rc3, mag + (y + 1) * MP + x0 + 4, map + (size_t)y * W + x0);
                if (y + 1 < y1) {
                    edge_row_f(simg, W, y + 2 < H ? y + 2 : H - 1, x0, ra);
                    edge_sobel_row(rb, rc3, ra, mag + (y + 2) * MP + x0 + 4, map + (size_t)(y + 1) * W + x0);
                }
                if (y + 2 < y1) {
                    edge_row_f(simg, W, y + 3 < H ? y + 3 : H - 1, x0, rb);
                    edge_sobel_row(rc3, ra, rb, mag + (y + 3) * MP + x0 + 4, map + (size_t)(y + 2) * W + x0);
                }
            }
        }
        __syncthreads();
        EDGE_STAMP(4);
        // the NEXT frame's channel sums, from the registers its groups were prefetched into in front of the Sobel phase (they have arrived;
        // s_part is not read again in this frame): the next frame starts with its delta, no sums phase and no barrier for it
        sums_ready = false;
        if (one_chunk && img + (int)gridDim.x < p.n_img) {
            unsigned sr = 0, sg = 0, sb = 0;
            chunk_sums(R, 0, sr, sg, sb);
            publish_sums(sr, sg, sb, par ^ 1);
            sums_ready = true;
        }
        // ---- non-maximum suppression + double threshold: map <- 0 = weak / 1 = no / 2 = edge ----
        int weak_here = 0;                                                   // this thread wrote a weak pixel (0) in the phase below
        // Branch-free, two pixels per instruction (hipcc turned the per-pixel choice of neighbours into divergent branches with LDS reads
        // inside them: 133 instructions per pixel).  A comparison a < b of two magnitudes (0 .. 2040) is the sign bit of the 16-bit
        // difference a - b; the magnitudes arrive packed two per dword, so v_pk_sub_i16 compares a PAIR of pixels with their
        // neighbours, the four direction classes' tests are combined on those sign bits (bits 15 and 31; the others carry garbage
        // and are masked at the end) and the pixel's own class picks one with two bit-field selects.
        {
            typedef short s2v __attribute__((ext_vector_type(2)));
            auto sub2 = [](unsigned a, unsigned b) -> unsigned { return __builtin_bit_cast(unsigned, (s2v)(__builtin_bit_cast(s2v, a) - __builtin_bit_cast(s2v, b))); };
            auto bsel = [](unsigned mask, unsigned a, unsigned b) -> unsigned { return (mask & a) | (~mask & b); };   // v_bfi_b32
            const int lo_c = min(max(p.edge_low, -1), 32767), hi_c = min(max(p.edge_high, -1), 32767);   // magnitudes are <= 2040: any larger threshold behaves like 32767
            const unsigned low2 = (unsigned)(lo_c & 0xFFFF) * 0x10001u, high2 = (unsigned)(hi_c & 0xFFFF) * 0x10001u;
            // (row, column group) of this thread's groups without a division per group: g advances by 1024 = dy rows + dc column groups
            const int dy = kEdgeBlock / p.gpr, dc = kEdgeBlock - dy * p.gpr;
            int g = fresh(tid), y = g / p.gpr, cg = g - y * p.gpr;
            for (; g < p.gpe; g += kEdgeBlock, y += dy, cg += dc) {
                if (cg >= p.gpr) { cg -= p.gpr; ++y; }
                const int x0 = cg * 4;
                unsigned ctr[3][2], lft[3][2], rgt[3][2];                    // per row: the pair itself, its left and its right neighbours
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const short* mr = mag + (y + rr) * MP + x0;              // shorts x0 .. x0 + 9 hold columns x0 - 4 .. x0 + 5
                    const uint2 a = *reinterpret_cast<const uint2*>(mr), b = *reinterpret_cast<const uint2*>(mr + 4);
                    const unsigned c = *reinterpret_cast<const unsigned*>(mr + 8);
                    ctr[rr][0] = b.x; ctr[rr][1] = b.y;                       // columns (x0, x0 + 1), (x0 + 2, x0 + 3)
                    lft[rr][0] = __builtin_amdgcn_alignbit(b.x, a.y, 16);     // (x0 - 1, x0)
                    lft[rr][1] = __builtin_amdgcn_alignbit(b.y, b.x, 16);     // (x0 + 1, x0 + 2)
                    rgt[rr][0] = lft[rr][1];
                    rgt[rr][1] = __builtin_amdgcn_alignbit(c, b.y, 16);       // (x0 + 3, x0 + 4)
                }
                const unsigned cls4 = *reinterpret_cast<const unsigned*>(map + (size_t)g * 4);
                unsigned out4 = 0u;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const unsigned M = ctr[1][pr];
                    const unsigned t0 = sub2(lft[1][pr], M) & ~sub2(M, rgt[1][pr]);     // class 0: m > left  && m >= right
                    const unsigned t1 = sub2(ctr[0][pr], M) & ~sub2(M, ctr[2][pr]);     // class 1: m > up    && m >= down
                    const unsigned t2 = sub2(lft[0][pr], M) & sub2(rgt[2][pr], M);      // class 2: m > up-left  && m > down-right
                    const unsigned t3 = sub2(rgt[0][pr], M) & sub2(lft[2][pr], M);      // class 3: m > up-right && m > down-left
                    const unsigned c2 = (cls4 >> (16 * pr)) & 0xFFFFu;                  // the pair's classes: byte 0, byte 1
                    const unsigned b0 = (c2 << 15) | (c2 << 23), b1 = (c2 << 14) | (c2 << 22);   // class bit 0 / bit 1 of the two pixels at bits 15 and 31
                    const unsigned sel = bsel(b1, bsel(b0, t3, t2), bsel(b0, t1, t0));
                    const unsigned ismax = TRS_EDGE_ABLATE == 6 ? 0u : (sel & sub2(low2, M));   // ... && m > low
                    const unsigned strong = ismax & sub2(high2, M);                              // ... && m > high
                    const unsigned h = ((~ismax >> 15) & 0x00010001u) | ((strong >> 14) & 0x00020002u);   // per half: 1 = no, 0 = weak, 2 = edge
                    out4 |= ((h & 0xFFu) | ((h >> 8) & 0xFF00u)) << (16 * pr);
                }
                *reinterpret_cast<unsigned*>(map + (size_t)g * 4) = out4;
                weak_here |= has_zero_byte(out4) ? 1 : 0;                    // (a group past the frame's end does not come here)
            }
        }
        // the barrier behind the suppression phase is a vote: a frame without a single weak pixel (0) has no hysteresis to run - no sweep, no barrier of its own
        const bool frame_has_weak = __syncthreads_or(weak_here) != 0;
        EDGE_STAMP(5);
        // ---- hysteresis: weak pixels 8-connected to an edge become edges.  A thread owns a contiguous run of pixels and walks it
        // forwards, then backwards: a chain along a row closes in one sweep instead of one pixel per sweep (the closure does not
        // depend on the order: only weak -> edge transitions); sweeps repeat until a block-wide OR reports no change ----
        {
            const int strip = 4 * ((p.gpe + kEdgeBlock - 1) / kEdgeBlock), s0 = fresh(tid) * strip, s1 = min(npx, s0 + strip);
            auto visit = [&](int px) -> int {
                const int y = px / W, x = px - y * W;
                bool hit = false;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W && map[yy * W + xx] == 2) hit = true;
                    }
                if (hit) map[px] = 2;
                return hit ? 1 : 0;
            };
            for (int iter = 0; iter < (TRS_EDGE_ABLATE == 3 || !frame_has_weak ? 0 : npx); ++iter) {
                int changed = 0;
                // A strip without a weak pixel (almost every strip) has nothing to do in either direction: its words are read TOGETHER first (one LDS
                // round trip) - the two walks below read them one after the other behind a branch each (ten dependent round trips per sweep).
                bool any_weak = false;
                for (int q = s0; q < s1; q += 4) any_weak |= has_zero_byte(*reinterpret_cast<const unsigned*>(map + q));
                if (any_weak) {
                    for (int q = s0; q < s1; q += 4) {
                        if (!has_zero_byte(*reinterpret_cast<const unsigned*>(map + q))) continue;
                        for (int k = 0; k < 4; ++k) if (map[q + k] == 0) changed |= visit(q + k);
                    }
                    for (int q = s1 - 4; q >= s0; q -= 4) {
                        if (!has_zero_byte(*reinterpret_cast<const unsigned*>(map + q))) continue;
                        for (int k = 3; k >= 0; --k) if (map[q + k] == 0) changed |= visit(q + k);
                    }
                }
                if (!__syncthreads_or(changed)) break;
            }
        }
        EDGE_STAMP(6);
        // ---- colour masks on the trimmed frame, merge, edge layer last (img_preprocessing.py:43-53) ----
        // (the switch p.color is tested once per group, not per pixel: four independent chains of dependent table lookups then
        // interleave instead of running one after the other)
        // The NEXT frame's trim table, by the first four waves (one per SIMD), while the other twelve keep the SIMDs busy with this phase: its channel
        // sums are complete (published in front of the suppression phase, three barriers ago).  Until late round 4 every frame began with this
        // table and a barrier - 256 threads in a chain of binary64 divisions, 768 waiting: 6 % of the kernel by the phase stamps.
        table_ready = sums_ready;
        if (table_ready && tid < 256) make_trim_table(par ^ 1, s_trim2 + (par ^ 1) * 256);
        for (int g = fresh(tid); g < (TRS_EDGE_ABLATE == 4 ? 0 : p.gpe); g += kEdgeBlock) {
            const unsigned* sw = reinterpret_cast<const unsigned*>(simg + (size_t)g * 12);
            const unsigned w0 = sw[0], w1 = sw[1], w2 = sw[2];              // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3, trimmed
            const unsigned e4 = *reinterpret_cast<const unsigned*>(map + (size_t)g * 4);
            unsigned P[4] = {w0, __builtin_amdgcn_alignbyte(w1, w0, 3), __builtin_amdgcn_alignbyte(w2, w1, 2), w2 >> 8};   // pixels as (r, g, b, x)
            if (p.color) {
#pragma unroll
                for (int q = 0; q < 4; ++q) P[q] = mask_pixel(P[q], s_tab, s_rng, sel);
            }
            if (p.edge_ch >= 0 && p.edge_ch <= 2) {                          // the edge layer last (:43-53): 255 where the map says edge (2), else 0
                const unsigned e1 = (e4 >> 1) & 0x01010101u, evb = (e1 << 8) - e1;   // per pixel byte: 0xFF / 0x00
                const unsigned em = 0xFFu << (8 * p.edge_ch);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned ev = (unsigned)__builtin_amdgcn_sbfe((int)evb, 8 * q, 8);   // all ones / zero
                    P[q] = (ev & em) | (P[q] & ~em);
                }
            }
            const u3v out = pack_rgb4(P[0], P[1], P[2], P[3]);
            __builtin_amdgcn_raw_buffer_store_b96(out, rd, g * 12, 0, 0);
        }
        __syncthreads();   // LDS is reused by the next frame of this workgroup
        par ^= 1;
        EDGE_STAMP(7);
    }
#ifdef TRS_EDGE_STAMPS
    if (blockIdx.x == 7 && tid == 0)
        printf("edge phases [clocks, workgroup 7, all its frames]: sums %llu | delta+table %llu | trim %llu | sobel %llu | nms %llu | hysteresis %llu | output %llu\n",
               ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
#endif
}

// float32(img) / 255 (keras_pilot.py:49-50): 4 bytes in, one 16-B store out per lane
__global__ __launch_bounds__(256) void trs_normalize_kernel(const uint32_t* src, float4* dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const uint32_t w = src[i];
        dst[i] = make_float4((float)(w & 255u) / 255.0f, (float)((w >> 8) & 255u) / 255.0f, (float)((w >> 16) & 255u) / 255.0f, (float)(w >> 24) / 255.0f);
    }
}

// DriverAssistance.step for N cars (components/driver_assistance.py:13-31), in place; binary64 like the reference's Python floats
__global__ void trs_driver_assist_kernel(int mode, double k, float* st, float* th, float* br, const float* sp, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double steering = st[i], throttle = th[i], breaking = br[i];
    const double speed = sp[i];
    if (mode == 0 && speed != 0) {
        const double max_steering = k / speed;
        if (steering > max_steering) { steering = max_steering; throttle = -0.1; }
        else if (steering < max_steering * -1) { steering = max_steering * -1; throttle = -0.1; }
    } else if (mode == 1 && steering != 0) {
        const double max_speed = k / steering;
        if (speed > max_speed) { throttle = 0.0; breaking = 0.0; }
    }
    st[i] = (float)steering; th[i] = (float)throttle; br[i] = (float)breaking;
}

// ControlMultiplexer.step for N cars (components/controlmultiplexer.py:24-43); semantics in include/trsim.h.
// Per-car state: int32 st[10] = 8 pending trigger ticks | ring head | last_mode + (steering lock << 8) + (throttle lock << 16)
constexpr int kMuxWords = 10;
constexpr int kMuxNone = INT32_MIN / 2;
struct MuxParams {
    const uint8_t* mode;
    const float *us, *ut, *ub, *as, *at, *ab;
    float *os, *ot, *ob;
    int32_t* state;
    int n, tick;
    int en_t, ticks_t, en_s, ticks_s;
    float val_t, val_s;
};

__global__ void trs_control_mux_kernel(const MuxParams p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    int32_t* st = p.state + (size_t)i * kMuxWords;
    int flags = st[9];
    int last_mode = flags & 255, act_s = (flags >> 8) & 1, act_t = (flags >> 16) & 1;
    // lock-end threads whose sleep elapses at this tick run before the step (:51-54, :67-70)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int trig = st[j];
        if (p.en_t && trig + p.ticks_t == p.tick) act_t = 0;
        if (p.en_s && trig + p.ticks_s == p.tick) act_s = 0;
    }
    const int mode = p.mode[i];
    float s = 0.f, t = 0.f, b = 0.f;
    const bool known = mode <= TRS_MODE_AI;
    if (mode == TRS_MODE_HUMAN) { s = p.us[i]; t = p.ut[i]; b = p.ub[i]; }                    // :26-27
    else if (mode == TRS_MODE_AI_STEERING) { s = p.as[i]; t = p.ut[i]; b = p.ub[i]; }         // :28-29
    else if (mode == TRS_MODE_AI) { s = p.as[i]; t = p.at[i]; b = p.ab[i]; }                  // :30-31
    if (last_mode != TRS_MODE_AI && mode == TRS_MODE_AI && (p.en_t || p.en_s)) {             // :33-35 AI launch detection
        if (p.en_t) act_t = 1;
        if (p.en_s) act_s = 1;
        const int head = st[8];
        st[head & 7] = p.tick;
        st[8] = head + 1;
    }
    if (known) {
        if (act_s) s = p.val_s;                                                               // :37-38
        if (act_t) t = p.val_t;                                                               // :39-40
        p.os[i] = s; p.ot[i] = t; p.ob[i] = b;
    }
    st[9] = (mode & 255) | (act_s << 8) | (act_t << 16);                                      // :42
}

__global__ void trs_control_mux_init_kernel(int32_t* state, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t* st = state + (size_t)i * kMuxWords;
    for (int j = 0; j < 8; ++j) st[j] = kMuxNone;
    st[8] = 0; st[9] = TRS_MODE_HUMAN;
}

// Batched LocationTracker.__find_closest (components/track_data_process.py:89-101): one wave per query,
// track staged in LDS once per workgroup, queries grid-strided.
__global__ __launch_bounds__(kLocBlock) void trs_locate_kernel(const unsigned char* blob, int stage_bytes, const NearParams g,
                                                               const double* q, int nq, int32_t* out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < (stage_bytes >> 4); i += kLocBlock) dst[i] = src[i];
    }
    __syncthreads();
    constexpr int kW = kLocBlock / 64;
    for (int qi = blockIdx.x * kW + wave; qi < nq; qi += gridDim.x * kW) {
        double best;
        int idx;
        wave_nearest(g, smem, q[3 * qi], q[3 * qi + 1], q[3 * qi + 2], lane, best, idx);
        if (lane == 0) out[qi] = idx;
    }
}

// ---------------------------------------------------------------------------------------------
// host side

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return fail(TRS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

namespace {

int grid_of(const trs_env* e) { return (e->n + e->pp.envs_per_wg - 1) / e->pp.envs_per_wg; }

// the handle's stream is idle: a resident worker (trs_set_step_mode) is asked to leave first — it owns the stream while it runs
int sync_all(trs_env* e)
{
    if (e->res) { int rc = trsim::resident_quiesce(e); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(e->sP));
    return trsim::check_fault(e);
}

// anything about to be queued on the handle's stream must not end up behind a resident worker
int quiesce(trs_env* e) { return e->res ? trsim::resident_quiesce(e) : TRS_OK; }

// steps go to the resident worker: resident mode is on (physics-only handles have their own worker kernel since round 4); the dynamic-brightness frame filter has its own worker
// instantiation since round 3 (it used to fall back to launches)
bool resident_steps(const trs_env* e) { return trsim::resident_on(e); }

// one launch of the fused step kernel: physics steps [step_base, step_base + n_phys) and the frames of launch-local
// steps r_first..r_last (-1 = step_base - 1, whose camera parameters the previous launch left in the global ring)
int launch_step(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth,
                int n_phys, int r_first, int r_last, uint64_t step_base, bool keep_uniform = false)
{
    SParams sp;
    sp.ph = e->pp;
    sp.ph.ctl_steer = st; sp.ph.ctl_thr = th; sp.ph.ctl_brk = br; sp.ph.ctl_reset = rs; sp.ph.ctl_stride = e->seq_stride;
    sp.ph.synth = synth; sp.ph.n_steps = n_phys; sp.ph.write_cam = 1; sp.ph.step_off = (uint32_t)step_base;
    sp.ra = e->rp;
    sp.img0 = e->img[0]; sp.img1 = e->img[1];
    sp.dep0 = e->depth[0]; sp.dep1 = e->depth[1];
    sp.n_phys = n_phys; sp.r_first = r_first; sp.r_last = r_last;
    sp.step_base = (unsigned)step_base;
    sp.lds_off_phys = e->lds_off_phys;
    sp.cam_stride = e->pp.envs_per_wg;
    sp.lds_off_cam = e->lds_step;                                                   // ring + counters sit behind the tables
    sp.lds_off_prog = sp.lds_off_cam + (std::max(n_phys, 1) + 1) * sp.cam_stride * 16;   // + one row: poses of the step before the launch
    int lds = sp.lds_off_prog + sp.cam_stride * 4 + 16;                            // + spare counters
    if (e->hilly) {                                                                 // a track with elevation: view pitches behind the counters, the batch's row tables, a counter
        const int off_pitch = lds;                                                  // (= lds_off_prog + cam_stride * 4 + 16: the HILLS kernels compute the same offsets)
        const int off_hill = (off_pitch + (std::max(n_phys, 1) + 1) * sp.cam_stride * 4 + 15) & ~15;
        lds = off_hill + hill_lds_bytes(e->H);
    }
    const bool dyn = e->has_frame_filter && e->filter_dynamic;
    // Which frame buffers hold the CURRENT palette's uniform rows for every env (e->uniform_ok[b]): a launch that renders whole frames into a buffer makes it
    // so; a palette change (upload_palette: track, frame filter) or the dynamic-brightness filter (its uniform rows follow each frame's own mean) undoes it.
    sp.skip_uniform = 0;
    for (int r = r_first; r <= r_last; ++r) {
        const int b = (int)((step_base + (uint64_t)(int64_t)r) & 1u);
        if (keep_uniform && r_first == r_last && e->uniform_ok[b] && !dyn) sp.skip_uniform = 1;
        else e->uniform_ok[b] = !dyn;
    }
    std::memset(&sp.fp, 0, sizeof sp.fp);
    if (dyn) {
        const trs_pre_config& c = e->frame_filter;
        sp.fp.baseline = c.brightness_baseline; sp.fp.contrast = c.contrast_ratio; sp.fp.offset = c.contrast_offset;
        sp.fp.color = c.color_filter_enabled; sp.fp.n_filters = c.n_filters;
        for (int k = 0; k < 4; ++k) {
            sp.fp.lo[k] = c.hsv_lo[k][0] | (c.hsv_lo[k][1] << 8) | (c.hsv_lo[k][2] << 16);
            sp.fp.hi[k] = c.hsv_hi[k][0] | (c.hsv_hi[k][1] << 8) | (c.hsv_hi[k][2] << 16);
            sp.fp.dst_ch[k] = c.dst_channel[k];
        }
        sp.fp.w0 = std::min(40, e->H); sp.fp.w1 = std::min(119, e->H);     // img[40:119] (img_preprocessing.py:88)
        sp.fp.tabs = e->dyn_tab;
        sp.fp.lds_off = (lds + 15) & ~15;
        lds = sp.fp.lds_off + dyn_lds_bytes(e->H);                         // palettes of a batch of 4 envs + channel sums + barrier counter + the mask tables
    }
    const dim3 grid(grid_of(e)), block(kBlock);
    if (e->hilly) {                                         // a track with elevation (its own instantiations; the dynamic-brightness filter is refused there)
        if (e->rp.depth) hipLaunchKernelGGL((trs_step_kernel<true, false, true>), grid, block, lds, e->sP, sp);
        else hipLaunchKernelGGL((trs_step_kernel<false, false, true>), grid, block, lds, e->sP, sp);
    } else if (dyn) {
        if (e->rp.depth) hipLaunchKernelGGL((trs_step_kernel<true, true>), grid, block, lds, e->sP, sp);
        else hipLaunchKernelGGL((trs_step_kernel<false, true>), grid, block, lds, e->sP, sp);
    } else {
#ifndef TRS_SINGLE_VARIANT   /* A/B switch: build without the depth instantiation (HIP guide rule 19: co-compiled variants perturb each other) */
        if (e->rp.depth) hipLaunchKernelGGL((trs_step_kernel<true, false>), grid, block, lds, e->sP, sp);
        else
#endif
        hipLaunchKernelGGL((trs_step_kernel<false, false>), grid, block, lds, e->sP, sp);
    }
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

// n env steps with a camera, K = steps per launch.  n == 1: one launch, the raster team waits for the physics team
// through the LDS progress counters.  Otherwise a software pipeline over launches: a launch advances K physics steps
// and renders the previous launch's last step plus its own steps 0..K-2; a raster-only launch closes the call, so on
// return state and image both belong to step s0+n-1.
int run_camera_steps(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, int n, int per_launch, bool keep_uniform = false)
{
    const uint64_t s0 = e->step_count;
    const bool dyn_filter = e->has_frame_filter && e->filter_dynamic;
    const int kmax = std::max(1, std::min(per_launch, dyn_filter ? e->max_steps_dyn : e->max_steps_per_launch));
    int rc = TRS_OK;
    if (n == 1) {
        rc = launch_step(e, st, th, br, rs, synth, 1, 0, 0, s0, keep_uniform);
    } else {
        for (int done = 0; done < n && !rc;) {
            const int k = std::min(kmax, n - done);
            const size_t co = (size_t)done * (size_t)e->seq_stride;       // a sequence call hands each launch its own slice of controls
            rc = launch_step(e, st ? st + co : st, th ? th + co : th, br ? br + co : br, done == 0 ? rs : nullptr, synth, k, done == 0 ? 0 : -1, k - 2, s0 + done);
            done += k;
        }
        if (!rc) rc = launch_step(e, st, th, br, nullptr, synth, 0, -1, -1, s0 + n);
    }
    if (rc) return rc;
    e->step_count += (uint64_t)n;
    return TRS_OK;
}

// physics-only envs: K steps inside one launch of the physics kernel
int run_physics_steps(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, int n, int per_launch)
{
    for (int done = 0; done < n;) {
        const int now = std::min(per_launch, n - done);
        PParams p = e->pp;
        const size_t co = (size_t)done * (size_t)e->seq_stride;
        p.ctl_steer = st ? st + co : st; p.ctl_thr = th ? th + co : th; p.ctl_brk = br ? br + co : br; p.ctl_reset = done == 0 ? rs : nullptr;
        p.ctl_stride = e->seq_stride;
        p.synth = synth; p.n_steps = now; p.write_cam = 0; p.step_off = (uint32_t)e->step_count;
        hipLaunchKernelGGL(trs_physics_kernel, dim3((e->n + kPhysBlock / 64 - 1) / (kPhysBlock / 64)), dim3(kPhysBlock), e->lds_p, e->sP, p);
        HIPCHK(hipGetLastError());
        e->step_count += (uint64_t)now;
        done += now;
    }
    return TRS_OK;
}

}  // namespace

TRS_EXPORT void trs_default_config(trs_config* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->n_envs = 1; c->img_h = 120; c->img_w = 160; c->render = 1;
    c->seed = TRS_SYNTH_SEED;
    c->dt = TRS_DEF_DT; c->max_steer = TRS_DEF_MAX_STEER; c->inv_wheelbase = TRS_DEF_INV_WHEELBASE;
    c->accel_max = TRS_DEF_ACCEL_MAX; c->drag_lin = TRS_DEF_DRAG_LIN; c->roll_res = TRS_DEF_ROLL_RES;
    c->brake_max = TRS_DEF_BRAKE_MAX; c->v_max = TRS_DEF_V_MAX; c->v_rev_max = TRS_DEF_V_REV_MAX;
    c->offtrack_cte = TRS_DEF_OFFTRACK_CTE; c->offtrack_penalty = TRS_DEF_OFFTRACK_PENALTY; c->cam_fwd = TRS_DEF_CAM_FWD;
    c->road_half = TRS_DEF_ROAD_HALF; c->edge_half = TRS_DEF_EDGE_HALF; c->centre_half = TRS_DEF_CENTRE_HALF;
    c->dash_period = TRS_DEF_DASH_PERIOD; c->dash_on = TRS_DEF_DASH_ON; c->map_margin = TRS_DEF_MAP_MARGIN;
    c->fov_v_deg = TRS_DEF_FOV_V_DEG; c->cam_h = TRS_DEF_CAM_H; c->cam_pitch_deg = TRS_DEF_CAM_PITCH_DEG; c->z_far = TRS_DEF_Z_FAR;
}

TRS_EXPORT int trs_device_count(int* out)
{
    if (!out) return fail(TRS_ERR_ARG, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *out = n;
    return TRS_OK;
}

namespace {
// everything trs_create builds; on a failure the caller destroys the half-built handle
int create_impl(const trs_config* cfg, int device, trs_env* e)
{
    e->cfg = *cfg; e->device = device; e->n = cfg->n_envs; e->H = cfg->img_h; e->W = cfg->img_w;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    e->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&e->sP, hipStreamNonBlocking));
    for (auto& ev : e->ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));

    // one slab for all per-env arrays: 10 float + 2 int32 state arrays, 3 float + 1 byte control arrays, 2 byte flags
    const size_t n = (size_t)e->n, fa = align_up(n * 4, 256), ba = align_up(n, 256);
    const size_t slab_bytes = fa * (10 + 2 + 3) + ba * 3;
    HIPCHK(hipMalloc((void**)&e->slab, slab_bytes));
    HIPCHK(hipMemsetAsync(e->slab, 0, slab_bytes, e->sP));
    unsigned char* c = e->slab;
    auto takef = [&](float*& ptr) { ptr = reinterpret_cast<float*>(c); c += fa; };
    PParams& k = e->pp;
    takef(k.x); takef(k.y); takef(k.z); takef(k.yaw); takef(k.v); takef(k.speed); takef(k.cte);
    takef(k.ep_return); takef(k.last_return); takef(k.steer_filt);
    k.seg_idx = reinterpret_cast<int32_t*>(c); c += fa;
    k.ep_len = reinterpret_cast<int32_t*>(c); c += fa;
    takef(e->ctl_steer); takef(e->ctl_thr); takef(e->ctl_brk);
    k.done = c; c += ba; k.pending = c; c += ba; e->ctl_reset = c; c += ba;
    HIPCHK(hipMalloc((void**)&e->stats, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(e->stats, 0, 64 * sizeof(unsigned long long), e->sP));
    HIPCHK(hipMalloc((void**)&e->cam, (size_t)kRing * n * sizeof(float4)));
    HIPCHK(hipMemsetAsync(e->cam, 0, (size_t)kRing * n * sizeof(float4), e->sP));
    HIPCHK(hipMalloc((void**)&e->cam_pitch, (size_t)kRing * n * sizeof(float)));   // the frames' view pitches beside the camera ring (tracks with elevation)
    HIPCHK(hipMemsetAsync(e->cam_pitch, 0, (size_t)kRing * n * sizeof(float), e->sP));
    k.stats = e->stats; k.cam = e->cam;
    RParams& r = e->rp;
    r.stats = e->stats;
    if (cfg->render) {
        e->img_bytes = n * (size_t)e->H * e->W * 3;
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipMalloc((void**)&e->img[b], e->img_bytes));
            HIPCHK(hipMemsetAsync(e->img[b], 0, e->img_bytes, e->sP));
            if (cfg->depth) {
                HIPCHK(hipMalloc((void**)&e->depth[b], n * (size_t)e->H * e->W * 4));
                HIPCHK(hipMemsetAsync(e->depth[b], 0, n * (size_t)e->H * e->W * 4, e->sP));
            }
        }
    }
    k.n_envs = e->n; k.env_id_base = cfg->env_id_base;
    k.envs_per_wg = (e->n + e->cu_count - 1) / e->cu_count;
    r.n_envs = e->n; r.envs_per_wg = k.envs_per_wg;
    r.H = e->H; r.W = e->W; r.gpr = e->W / 4; r.gpe = r.gpr * e->H; r.depth = (cfg->render && cfg->depth) ? 1 : 0;
    k.dt = cfg->dt; k.max_steer = cfg->max_steer; k.inv_wheelbase = cfg->inv_wheelbase; k.accel_max = cfg->accel_max;
    k.drag_lin = cfg->drag_lin; k.roll_res = cfg->roll_res; k.brake_max = cfg->brake_max; k.v_max = cfg->v_max;
    k.v_rev_max = cfg->v_rev_max; k.offtrack_cte = cfg->offtrack_cte; k.offtrack_penalty = cfg->offtrack_penalty;
    k.cam_fwd = cfg->cam_fwd; k.auto_reset = cfg->auto_reset; k.seed = cfg->seed;
    if (r.gpr > kBlock) return fail(TRS_ERR_LIMIT, "img_w too large: more 4-pixel groups per row than threads per workgroup");
    if (r.gpr > kRasterThreads) return fail(TRS_ERR_LIMIT, "img_w too large: more 4-pixel groups per row than raster threads");
    r.rows_per_pass = kRasterThreads / r.gpr;   // raster threads beyond rows_per_pass * gpr idle (32 of 512 at W = 160)
    HIPCHK(hipHostMalloc((void**)&e->fault, 64, hipHostMallocMapped | hipHostMallocCoherent));   // kernels report a layout fault here (checked at every synchronisation)
    *e->fault = 0ull;
    k.fault = e->fault; r.fault = e->fault;
    HIPCHK(hipStreamSynchronize(e->sP));
    return TRS_OK;
}


}  // namespace

TRS_EXPORT int trs_create(const trs_config* cfg, int device, trs_env** out)
{
    if (!cfg || !out) return fail(TRS_ERR_ARG, "null argument");
    if (cfg->struct_size != sizeof(trs_config)) return fail(TRS_ERR_ARG, "trs_config.struct_size mismatch");
    if (cfg->n_envs < 1 || cfg->img_h < 2 || cfg->img_w < 4 || (cfg->img_w & 3) || cfg->env_id_base < 0)
        return fail(TRS_ERR_ARG, "bad n_envs / image size (img_w must be a multiple of 4)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(TRS_ERR_DEVICE, "no HIP device visible (libtrsim has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TRS_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    trs_env* e = new (std::nothrow) trs_env();
    if (!e) return fail(TRS_ERR_NOMEM, "out of memory");
    e->device = device;
    const int rc = create_impl(cfg, device, e);
    if (rc) {                                                // keep the message of the failure, not of the clean-up
        const std::string why = g_err;
        (void)trs_destroy(e);
        return fail(rc, why);
    }
    *out = e;
    return TRS_OK;
}

TRS_EXPORT int trs_destroy(trs_env* e)
{
    if (!e) return TRS_OK;
    (void)hipSetDevice(e->device);
    trsim::resident_destroy(e);
    if (e->sP) (void)hipStreamSynchronize(e->sP);
    trsim::comm_destroy(e);
    if (e->ev_order) (void)hipEventDestroy(e->ev_order);
    if (e->pilot) { trs_pilot_free(e->pilot); e->pilot = nullptr; }
    (void)hipFree(e->slab); (void)hipFree(e->img[0]); (void)hipFree(e->img[1]); (void)hipFree(e->depth[0]); (void)hipFree(e->depth[1]); (void)hipFree(e->blob_p); (void)hipFree(e->blob_r);
    (void)hipFree(e->tangent); (void)hipFree(e->start_yaw); (void)hipFree(e->cam); (void)hipFree(e->cam_pitch); (void)hipFree(e->dpitch);
    (void)hipFree(e->stats); (void)hipFree(e->loc_q); (void)hipFree(e->loc_out);
    (void)hipFree(e->mux_state); (void)hipFree(e->edge_scratch); (void)hipFree(e->seq_buf); (void)hipFree(e->glue);
    for (void* sc : e->scratch) (void)hipFree(sc);
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->fault) (void)hipHostFree(e->fault);
    (void)hipFree(e->pre); (void)hipFree(e->tmp_in); (void)hipFree(e->tmp_out); (void)hipFree(e->tmp_f); (void)hipFree(e->hsv_tab); (void)hipFree(e->dyn_tab);
    for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->sP) (void)hipStreamDestroy(e->sP);
    delete e;
    return TRS_OK;
}

namespace { int upload_palette(trs_env* e); }

TRS_EXPORT int trs_load_track(trs_env* e, const double* h_xyz, int n_points)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    int rc0 = sync_all(e);
    if (rc0) return rc0;
    // Transactional: everything is built into copies (tables, parameter blocks, layout numbers, device buffers) and
    // committed only after every check, allocation and copy has succeeded — a failed reload leaves the handle with the
    // track it had (or with none), never with new offsets against old blobs.
    std::string err;
    trsim::TrackTables T;
    int rc = trsim::build_tables(e->cfg, h_xyz, n_points, T, err);
    if (rc) return fail(rc, err);
    PParams k = e->pp;
    RParams r = e->rp;
    struct Staged {                                          // device buffers of the new track; freed unless committed
        unsigned char *blob_p = nullptr, *blob_r = nullptr; float *tangent = nullptr, *start_yaw = nullptr, *dpitch = nullptr;
        ~Staged() { (void)hipFree(blob_p); (void)hipFree(blob_r); (void)hipFree(tangent); (void)hipFree(start_yaw); (void)hipFree(dpitch); }
    } nb;
    int n_lds_r = 0, n_lds_p = 0, n_lds_step = 0, n_lds_off_phys = 0, n_pts_bytes = 0, n_max_spl = 1, n_max_dyn = 0;

    // ---- physics LDS image: px | py | pz | tangent (when it fits) + scratch ----
    const size_t pts = align_up((size_t)n_points * 8, 16);
    const size_t scratch = (size_t)(kPhysBlock / 64) * (16 + 4) + 16;     // physics-only kernel: per-wave sinks
    size_t off = 0;
    k.off_py = (int)(off += pts); k.off_pz = (int)(off += pts); off += pts;
    n_pts_bytes = (int)off;
    k.off_tan = (int)off;
    const size_t tan_bytes = align_up((size_t)n_points * 8, 16);
    // ---- raster LDS image: map (rows pitched to an odd number of words) @0 | rowtab | palette ----
    const int pitch_words = T.info.map_words | 1;          // odd pitch: rows of the map start on different LDS banks
    r.map_pitch_b = pitch_words * 4;
    if (r.map_pitch_b >= (1 << 24) || T.info.map_h >= (1 << 24)) return fail(TRS_ERR_LIMIT, "map exceeds the 24-bit multiply of the rasteriser");
    const size_t map_bytes = (size_t)r.map_pitch_b * T.info.map_h;
    size_t roff = align_up(map_bytes, 16);
    r.off_rowtab = (int)roff; roff += align_up((size_t)e->H * 8, 16);
    r.off_pal = (int)roff; roff += (size_t)e->H * 16;
    r.off_depth = (int)roff; roff += align_up((size_t)e->H * 4, 16);
    const int off_sky = (int)roff;
    if (T.hills) roff += align_up((size_t)e->H * 4, 16);     // a track with elevation: the sky colour of every row rides in the raster image (hill_rows_build)
    r.blob_bytes = (int)roff;
    n_lds_r = (int)align_up(roff, 16);
    if ((size_t)r.blob_bytes > (size_t)100 * 1024)
        return fail(TRS_ERR_LIMIT, "map + camera tables exceed the step kernel's LDS staging capacity");
    // tangents ride in LDS when the fused kernel's image (raster tables + points + tangents) still fits a CU's 160 KiB
    n_lds_off_phys = n_lds_r;
    const size_t grid_bytes = align_up(T.grid_start.size() * 2, 16) + align_up(T.grid_pts.size() * 2, 16);
    // (... and leaves room for what the kernels keep behind the tables: the camera hand-off ring of a few steps, and on a track with elevation the two per-env
    // row tables + the view pitches — the mountain track's points + tangents fill the CU to within 150 bytes on their own)
    const size_t behind = (size_t)k.envs_per_wg * 4 + 16 + (size_t)k.envs_per_wg * 20 * 3 + (T.hills ? (size_t)hill_lds_bytes(e->H) + 64 : 0);
    k.tan_in_lds = ((size_t)n_lds_off_phys + off + grid_bytes + tan_bytes + behind <= 160 * 1024) ? 1 : 0;
    if (k.tan_in_lds) off += tan_bytes;
    // nearest-point accelerator tables ride behind the points (uint16 cell starts + point lists)
    k.grid_nx = T.grid_nx; k.grid_nz = T.grid_nz; k.grid_x0 = T.grid_x0; k.grid_z0 = T.grid_z0;
    k.off_gstart = (int)off; off += align_up(T.grid_start.size() * 2, 16);
    k.off_gpts = (int)off; off += align_up(T.grid_pts.size() * 2, 16);
    k.blob_bytes = (int)off;
    k.off_scratch = (int)off;
    n_lds_p = (int)align_up(off + scratch, 16);
    n_lds_step = (int)align_up((size_t)n_lds_off_phys + off, 16);
    if (n_lds_p > 160 * 1024 || (e->cfg.render && n_lds_step > 160 * 1024))
        return fail(TRS_ERR_LIMIT, "track too long for the LDS-resident nearest-point search");
    std::vector<unsigned char> hp(off, 0);
    std::memcpy(hp.data(), T.px.data(), (size_t)n_points * 8);
    std::memcpy(hp.data() + k.off_py, T.py.data(), (size_t)n_points * 8);
    std::memcpy(hp.data() + k.off_pz, T.pz.data(), (size_t)n_points * 8);
    if (k.tan_in_lds) std::memcpy(hp.data() + k.off_tan, T.tangent.data(), (size_t)n_points * 8);
    if (!T.grid_start.empty()) std::memcpy(hp.data() + k.off_gstart, T.grid_start.data(), T.grid_start.size() * 2);
    if (!T.grid_pts.empty()) std::memcpy(hp.data() + k.off_gpts, T.grid_pts.data(), T.grid_pts.size() * 2);

    std::vector<unsigned char> hr(roff, 0);
    for (int row = 0; row < T.info.map_h; ++row)
        std::memcpy(hr.data() + (size_t)row * r.map_pitch_b, T.map.data() + (size_t)row * T.info.map_words, (size_t)T.info.map_words * 4);
    std::memcpy(hr.data() + r.off_rowtab, T.rowtab.data(), (size_t)e->H * 8);
    std::memcpy(hr.data() + r.off_pal, T.palette.data(), (size_t)e->H * 16);
    std::memcpy(hr.data() + r.off_depth, T.rowdepth.data(), (size_t)e->H * 4);
    if (T.hills) std::memcpy(hr.data() + off_sky, T.sky.data(), (size_t)e->H * 4);

    HIPCHK(hipMalloc((void**)&nb.blob_p, off));
    HIPCHK(hipMalloc((void**)&nb.blob_r, trsim::hill_block_offset((int)roff) + sizeof(trsim::HillBlock)));   // (+ the constants of a track with elevation, behind the image)
    HIPCHK(hipMalloc((void**)&nb.tangent, (size_t)n_points * 8));
    HIPCHK(hipMalloc((void**)&nb.start_yaw, (size_t)n_points * 4));
    HIPCHK(hipMemcpy(nb.blob_p, hp.data(), off, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(nb.blob_r, hr.data(), roff, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(nb.tangent, T.tangent.data(), (size_t)n_points * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(nb.start_yaw, T.start_yaw.data(), (size_t)n_points * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&nb.dpitch, (size_t)n_points * 4));
    {   // the view pitch of a frame whose nearest raw track point is idx: pitch_f + dpitch[idx], ONE binary32 addition as the spec has it
        std::vector<float> vp((size_t)n_points);
        for (int i = 0; i < n_points; ++i) vp[i] = T.pitch_f + T.dpitch[i];
        HIPCHK(hipMemcpy(nb.dpitch, vp.data(), (size_t)n_points * 4, hipMemcpyHostToDevice));
    }
    k.blob = nb.blob_p; k.start_yaw = nb.start_yaw; k.tangent_g = nb.tangent;
    r.blob = nb.blob_r;
    k.np = n_points; r.map_w = T.info.map_w; r.map_h = T.info.map_h;
    k.map_x0f = T.map_x0f; k.map_z0f = T.map_z0f; k.inv_cellf = T.inv_cellf;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_physics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, n_lds_p));
    {   // room left in the CU's 160 KiB for the in-launch camera ring: float4 per env per step + one counter per env
        const int epw = k.envs_per_wg;
        // (a track with elevation: + two per-env row tables and their counter, + 4 bytes of view pitch per env and step beside the 16 of the camera parameters)
        const int hill_b = T.hills ? hill_lds_bytes(e->H) + 16 + epw * 4 : 0, per_step = epw * (T.hills ? 20 : 16);
        const int free_b = 160 * 1024 - n_lds_step - epw * 4 - 16 - epw * 16 - hill_b;
        n_max_spl = std::max(1, std::min(16, free_b / per_step));
        const int free_dyn = free_b - (dyn_lds_bytes(e->H) + 32);
        n_max_dyn = free_dyn >= epw * 16 ? std::min(16, free_dyn / (epw * 16)) : 0;
        if (e->cfg.render && free_b < per_step) return fail(TRS_ERR_LIMIT, T.hills ? "no LDS left for the camera hand-off ring and the per-env row tables of a track with elevation"
                                                                                                 : "no LDS left for the camera hand-off ring");
        if (T.hills) n_max_dyn = 0;                              // (the dynamic-brightness filter is not available on a track with elevation)
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifndef TRS_SINGLE_VARIANT
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_step_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_locate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, k.blob_bytes));
    // ---- commit: nothing above has touched the handle ----
    e->track_loaded = false;                                 // (until the state arrays below are in place)
    (void)hipFree(e->blob_p); (void)hipFree(e->blob_r); (void)hipFree(e->tangent); (void)hipFree(e->start_yaw); (void)hipFree(e->dpitch);
    e->blob_p = nb.blob_p; e->blob_r = nb.blob_r; e->tangent = nb.tangent; e->start_yaw = nb.start_yaw; e->dpitch = nb.dpitch;
    {
        trsim::HillBlock& hb = e->hill_host;                 // (host copy: upload_palette fills in the frame filter and sends the block again)
        hb = trsim::HillBlock{};
        hb.vpitch = nb.dpitch; hb.cam_pitch = e->cam_pitch; hb.off_sky = off_sky; hb.far_rgb = T.far_rgb;
        hb.inv_f = T.inv_f; hb.hh = T.hh; hb.cam_h_f = T.cam_h_f; hb.z_far_f = T.z_far_f; hb.inv_zfar_f = T.inv_zfar_f; hb.fog_f = T.fog_f; hb.inv_cell_f = T.inv_cellf;
        e->hilly = T.hills;
        HIPCHK(hipMemcpy(e->blob_r + trsim::hill_block_offset(r.blob_bytes), &hb, sizeof hb, hipMemcpyHostToDevice));
    }
    nb = Staged{};
    e->tab = std::move(T);
    e->pp = k; e->rp = r;
    e->lds_r = n_lds_r; e->lds_p = n_lds_p; e->lds_step = n_lds_step; e->lds_off_phys = n_lds_off_phys; e->pts_bytes = n_pts_bytes;
    e->max_steps_per_launch = n_max_spl; e->max_steps_dyn = n_max_dyn;
    const trsim::TrackTables& TT = e->tab;
    // start poses (host mirror of the reset branch so that telemetry is meaningful before the first step)
    const size_t n = (size_t)e->n;
    std::vector<float> sx(n), sy(n), sz(n), syaw(n);
    std::vector<int32_t> sidx(n);
    for (size_t i = 0; i < n; ++i) {
        const int gid = e->cfg.env_id_base + (int)i;
        const int si = (int)(((long long)TRS_START_STRIDE * gid) % n_points);
        sx[i] = (float)TT.px[si]; sy[i] = (float)TT.py[si]; sz[i] = (float)TT.pz[si]; syaw[i] = TT.start_yaw[si]; sidx[i] = si;
    }
    HIPCHK(hipMemcpy(k.x, sx.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.y, sy.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.z, sz.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.yaw, syaw.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(k.seg_idx, sidx.data(), n * 4, hipMemcpyHostToDevice));
    for (float* a : {k.v, k.speed, k.cte, k.ep_return, k.last_return, k.steer_filt}) HIPCHK(hipMemset(a, 0, n * 4));
    HIPCHK(hipMemset(k.ep_len, 0, n * 4));
    HIPCHK(hipMemset(k.done, 0, n));
    HIPCHK(hipMemset(k.pending, 1, n));
    HIPCHK(hipMemset(e->stats, 0, 64 * sizeof(unsigned long long)));
    e->step_count = 0;
    e->track_loaded = true;
    trsim::resident_clear_fault(e);
    if (e->has_frame_filter && e->filter_dynamic && e->hilly) {   // (the dynamic-brightness filter is not built for tracks with elevation, see trs_set_frame_filter)
        e->has_frame_filter = false; e->filter_dynamic = false;
        (void)upload_palette(e);
        return fail(TRS_ERR_STATE, "this track has elevation: the dynamic-brightness frame filter that was set has been removed (a frame's palette is evaluated per env inside the kernels there; "
                                   "the static filter works, or use trs_preprocess on the rendered frames)");
    }
    if (e->has_frame_filter && e->filter_dynamic && e->max_steps_dyn < 1) {
        e->has_frame_filter = false; e->filter_dynamic = false;
        (void)upload_palette(e);
        return fail(TRS_ERR_LIMIT, "this track leaves no LDS for the dynamic-brightness frame filter that was set: the filter has been removed");
    }
    return upload_palette(e);                 // also sets rp.uni_rows (and applies a frame filter that was set earlier)
}

TRS_EXPORT int trs_reset(trs_env* e, const uint8_t* h_mask)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    HIPCHK(hipSetDevice(e->device));
    int rc = sync_all(e);
    if (rc) return rc;
    if (!h_mask) { HIPCHK(hipMemset(e->pp.pending, 1, (size_t)e->n)); return TRS_OK; }
    std::vector<uint8_t> cur((size_t)e->n);
    HIPCHK(hipMemcpy(cur.data(), e->pp.pending, cur.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->n; ++i) if (h_mask[i]) cur[i] = 1;
    HIPCHK(hipMemcpy(e->pp.pending, cur.data(), cur.size(), hipMemcpyHostToDevice));
    return TRS_OK;
}

TRS_EXPORT int trs_step(trs_env* e, const float* d_st, const float* d_th, const float* d_br, const uint8_t* d_rs, int n_steps)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1) return fail(TRS_ERR_ARG, "n_steps < 1");
    if (!d_st || !d_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    trsim::resident_retry(e);
    if (resident_steps(e)) {
        int done = 0;
        const int rc = trsim::resident_post(e, d_st, d_th, d_br, d_rs, 0, n_steps, 0, &done);
        if (rc != trsim::kResidentFellBack || done == n_steps) return rc < 0 ? rc : TRS_OK;
        n_steps -= done;                                     // the handle went back to launch mode (trs_last_error() says why): the rest of the call by launches
        if (done) d_rs = nullptr;
    }
    { int rq = quiesce(e); if (rq) return rq; }
    // held controls; the reset request applies to the first step only
    return e->cfg.render ? run_camera_steps(e, d_st, d_th, d_br, d_rs, 0, n_steps, n_steps)
                         : run_physics_steps(e, d_st, d_th, d_br, d_rs, 0, n_steps, 1);
}

TRS_EXPORT int trs_step_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1) return fail(TRS_ERR_ARG, "n_steps < 1");
    if (!h_st || !h_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    trsim::resident_retry(e);
    if (resident_steps(e)) {
        int done = 0;
        const int rc = trsim::resident_post_host(e, h_st, h_th, h_br, h_rs, n_steps, &done);
        if (rc != trsim::kResidentFellBack || done == n_steps) return rc < 0 ? rc : TRS_OK;
        n_steps -= done;
        if (done) h_rs = nullptr;
    }
    { int rq = quiesce(e); if (rq) return rq; }
    const size_t n = (size_t)e->n;
    e->h2d_bytes += n * 8 + (h_br ? n * 4 : 0) + (h_rs ? n : 0);
    HIPCHK(hipMemcpyAsync(e->ctl_steer, h_st, n * 4, hipMemcpyHostToDevice, e->sP));
    HIPCHK(hipMemcpyAsync(e->ctl_thr, h_th, n * 4, hipMemcpyHostToDevice, e->sP));
    if (h_br) HIPCHK(hipMemcpyAsync(e->ctl_brk, h_br, n * 4, hipMemcpyHostToDevice, e->sP));
    if (h_rs) HIPCHK(hipMemcpyAsync(e->ctl_reset, h_rs, n, hipMemcpyHostToDevice, e->sP));
    return trs_step(e, e->ctl_steer, e->ctl_thr, h_br ? e->ctl_brk : nullptr, h_rs ? e->ctl_reset : nullptr, n_steps);
}

TRS_EXPORT int trs_step_sequence(trs_env* e, const float* d_st, const float* d_th, const float* d_br, const uint8_t* d_rs, int n_steps, int steps_per_launch)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1 || steps_per_launch < 1) return fail(TRS_ERR_ARG, "n_steps / steps_per_launch < 1");
    if (!d_st || !d_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    trsim::resident_retry(e);
    if (resident_steps(e)) {
        int done = 0;
        const int rc = trsim::resident_post(e, d_st, d_th, d_br, d_rs, 0, n_steps, (size_t)e->n, &done);
        if (rc != trsim::kResidentFellBack || done == n_steps) return rc < 0 ? rc : TRS_OK;
        const size_t co = (size_t)done * (size_t)e->n;
        d_st += co; d_th += co; if (d_br) d_br += co;
        n_steps -= done;
        if (done) d_rs = nullptr;
    }
    { int rq = quiesce(e); if (rq) return rq; }
    e->seq_stride = e->n;
    const int rc = e->cfg.render ? run_camera_steps(e, d_st, d_th, d_br, d_rs, 0, n_steps, steps_per_launch)
                                 : run_physics_steps(e, d_st, d_th, d_br, d_rs, 0, n_steps, steps_per_launch);
    e->seq_stride = 0;
    return rc;
}

TRS_EXPORT int trs_step_sequence_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps, int steps_per_launch)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1 || steps_per_launch < 1) return fail(TRS_ERR_ARG, "n_steps / steps_per_launch < 1");
    if (!h_st || !h_th) return fail(TRS_ERR_ARG, "null controls");
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }            // the sequence is copied through the handle's stream
    const size_t cnt = (size_t)n_steps * (size_t)e->n;
    if (e->seq_cap < cnt) {
        HIPCHK(hipStreamSynchronize(e->sP));
        (void)hipFree(e->seq_buf); e->seq_buf = nullptr; e->seq_cap = 0;
        HIPCHK(hipMalloc((void**)&e->seq_buf, cnt * 3 * sizeof(float)));
        e->seq_cap = cnt;
    }
    float *ds = e->seq_buf, *dt = ds + cnt, *db = dt + cnt;
    HIPCHK(hipMemcpyAsync(ds, h_st, cnt * 4, hipMemcpyHostToDevice, e->sP));
    HIPCHK(hipMemcpyAsync(dt, h_th, cnt * 4, hipMemcpyHostToDevice, e->sP));
    if (h_br) HIPCHK(hipMemcpyAsync(db, h_br, cnt * 4, hipMemcpyHostToDevice, e->sP));
    if (h_rs) HIPCHK(hipMemcpyAsync(e->ctl_reset, h_rs, (size_t)e->n, hipMemcpyHostToDevice, e->sP));
    return trs_step_sequence(e, ds, dt, h_br ? db : nullptr, h_rs ? e->ctl_reset : nullptr, n_steps, steps_per_launch);
}

TRS_EXPORT int trs_step_synthetic(trs_env* e, int n_steps, int steps_per_launch)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (n_steps < 1 || steps_per_launch < 1) return fail(TRS_ERR_ARG, "n_steps / steps_per_launch < 1");
    HIPCHK(hipSetDevice(e->device));
    trsim::resident_retry(e);
    if (resident_steps(e)) {
        int done = 0;
        const int rc = trsim::resident_post(e, nullptr, nullptr, nullptr, nullptr, 1, n_steps, 0, &done);
        if (rc != trsim::kResidentFellBack || done == n_steps) return rc < 0 ? rc : TRS_OK;
        n_steps -= done;
    }
    { int rq = quiesce(e); if (rq) return rq; }
    return e->cfg.render ? run_camera_steps(e, nullptr, nullptr, nullptr, nullptr, 1, n_steps, steps_per_launch)
                         : run_physics_steps(e, nullptr, nullptr, nullptr, nullptr, 1, n_steps, steps_per_launch);
}

TRS_EXPORT int trs_get_state(trs_env* e, trs_state_view* o)
{
    if (!e || !o) return fail(TRS_ERR_ARG, "null argument");
    const PParams& k = e->pp;
    o->n_envs = e->n; o->img_h = e->H; o->img_w = e->W; o->n_points = k.np;
    o->img = e->cfg.render ? e->img[(e->step_count + 1) & 1] : nullptr;   // buffer written by the last step
    o->pos_x = k.x; o->pos_y = k.y; o->pos_z = k.z; o->speed = k.speed; o->cte = k.cte; o->yaw = k.yaw; o->vel = k.v;
    o->seg_idx = k.seg_idx; o->ep_return = k.ep_return; o->last_return = k.last_return; o->ep_len = k.ep_len; o->done = k.done;
    o->step_count = e->step_count;
    o->depth = (e->cfg.render && e->cfg.depth) ? e->depth[(e->step_count + 1) & 1] : nullptr;
    return TRS_OK;
}

TRS_EXPORT int trs_copy_to_host(trs_env* e, int which, void* dst, size_t bytes)
{
    if (!e || !dst) return fail(TRS_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(e->device));
    const PParams& k = e->pp;
    const void* src = nullptr; size_t need = 0; const size_t n = (size_t)e->n;
    bool host_src = false;
    switch (which) {
    case TRS_F_IMG: src = e->cfg.render ? e->img[(e->step_count + 1) & 1] : nullptr; need = e->img_bytes; break;
    case TRS_F_POS_X: src = k.x; need = n * 4; break;
    case TRS_F_POS_Y: src = k.y; need = n * 4; break;
    case TRS_F_POS_Z: src = k.z; need = n * 4; break;
    case TRS_F_SPEED: src = k.speed; need = n * 4; break;
    case TRS_F_CTE: src = k.cte; need = n * 4; break;
    case TRS_F_YAW: src = k.yaw; need = n * 4; break;
    case TRS_F_VEL: src = k.v; need = n * 4; break;
    case TRS_F_SEG_IDX: src = k.seg_idx; need = n * 4; break;
    case TRS_F_EP_RETURN: src = k.ep_return; need = n * 4; break;
    case TRS_F_LAST_RETURN: src = k.last_return; need = n * 4; break;
    case TRS_F_EP_LEN: src = k.ep_len; need = n * 4; break;
    case TRS_F_DONE: src = k.done; need = n; break;
    case TRS_F_STEER_FILT: src = k.steer_filt; need = n * 4; break;
    case TRS_F_CTL_STEER: src = e->ctl_steer; need = n * 4; break;
    case TRS_F_CTL_THR: src = e->ctl_thr; need = n * 4; break;
    case TRS_F_CTL_BRK: src = e->ctl_brk; need = n * 4; break;
    case TRS_F_STATS: src = e->stats; need = 64 * sizeof(unsigned long long); break;
    case TRS_F_DEPTH: src = (e->cfg.render && e->cfg.depth) ? e->depth[(e->step_count + 1) & 1] : nullptr; need = n * e->H * e->W * 4; break;
    case TRS_F_ROWDEPTH: if (e->track_loaded) { src = e->blob_r + e->rp.off_depth; need = (size_t)e->H * 4; } break;
    // the tables live on the host exactly as built; their LDS images on the device are re-laid-out (pitched map)
    case TRS_F_MAP: if (e->track_loaded) { src = e->tab.map.data(); need = e->tab.map.size() * 4; host_src = true; } break;
    case TRS_F_ROWTAB: if (e->track_loaded) { src = e->blob_r + e->rp.off_rowtab; need = (size_t)e->H * 8; } break;
    case TRS_F_PALETTE: if (e->track_loaded) { src = e->blob_r + e->rp.off_pal; need = (size_t)e->H * 16; } break;
    case TRS_F_TANGENT: if (e->track_loaded) { src = e->tangent; need = (size_t)k.np * 8; } break;
    case TRS_F_DPITCH: if (e->track_loaded) { src = e->tab.dpitch.data(); need = e->tab.dpitch.size() * 4; host_src = true; } break;
    default: return fail(TRS_ERR_ARG, "unknown field");
    }
    if (!src) return fail(TRS_ERR_STATE, "field not available");
    if (bytes != need) return fail(TRS_ERR_ARG, "byte count mismatch");
    if (host_src) { std::memcpy(dst, src, need); return TRS_OK; }
    if (resident_steps(e) && which != TRS_F_STATS) {
        // the worker keeps the handle's stream: wait for its posted steps (their bytes are written through to memory before
        // the completion flag), then copy on the side stream
        int rw = trsim::resident_wait(e);
        if (rw) return rw;
        hipStream_t sc = trsim::resident_copy_stream(e);
        HIPCHK(hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, sc));
        HIPCHK(hipStreamSynchronize(sc));
        e->d2h_bytes += need;
        return trsim::check_fault(e);
    }
    int rc = sync_all(e);
    if (rc) return rc;
    HIPCHK(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    e->d2h_bytes += need;
    if (which == TRS_F_STATS && static_cast<unsigned long long*>(dst)[2] != 0)
        return fail(TRS_ERR_DEVICE, "raster kernel found its dynamic LDS segment at a non-zero offset");
    return TRS_OK;
}

TRS_EXPORT int trs_fetch_outputs(trs_env* e, uint8_t* h_img, float* h_x, float* h_y, float* h_z, float* h_speed, float* h_cte,
                                 int32_t* h_seg, uint8_t* h_done)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (h_img && !e->cfg.render) return fail(TRS_ERR_STATE, "the env has no camera (cfg.render == 0)");
    HIPCHK(hipSetDevice(e->device));
    const PParams& k = e->pp;
    const size_t n = (size_t)e->n, img_b = h_img ? e->img_bytes : 0;
    const size_t need = ((img_b + 15) & ~(size_t)15) + 7 * ((n * 4 + 15) & ~(size_t)15) + 16;   // every item starts 16-B aligned
    if (e->pinned_bytes < need) {
        if (e->pinned) { int rq = sync_all(e); if (rq) return rq; (void)hipHostFree(e->pinned); e->pinned = nullptr; e->pinned_bytes = 0; }
        HIPCHK(hipHostMalloc((void**)&e->pinned, need, hipHostMallocDefault));
        e->pinned_bytes = need;
    }
    struct Item { const void* src; void* dst; size_t bytes; };
    const Item items[8] = {
        {h_img ? e->img[(e->step_count + 1) & 1] : nullptr, h_img, img_b},
        {k.x, h_x, n * 4}, {k.y, h_y, n * 4}, {k.z, h_z, n * 4}, {k.speed, h_speed, n * 4}, {k.cte, h_cte, n * 4},
        {k.seg_idx, h_seg, n * 4}, {k.done, h_done, n},
    };
    hipStream_t cs = e->sP;
    if (resident_steps(e)) { int rw = trsim::resident_wait(e); if (rw) return rw; cs = trsim::resident_copy_stream(e); }
    else { int rq = quiesce(e); if (rq) return rq; }
    size_t off = 0;
    for (const Item& it : items) {
        if (!it.dst || !it.bytes) continue;
        HIPCHK(hipMemcpyAsync(e->pinned + off, it.src, it.bytes, hipMemcpyDeviceToHost, cs));
        e->d2h_bytes += it.bytes;
        off += (it.bytes + 15) & ~(size_t)15;
    }
    HIPCHK(hipStreamSynchronize(cs));
    { int rf = trsim::check_fault(e); if (rf) return rf; }
    off = 0;
    for (const Item& it : items) {
        if (!it.dst || !it.bytes) continue;
        std::memcpy(it.dst, e->pinned + off, it.bytes);
        off += (it.bytes + 15) & ~(size_t)15;
    }
    return TRS_OK;
}

TRS_EXPORT int trs_set_pose(trs_env* e, const float* x, const float* y, const float* z, const float* yaw, const float* v)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    HIPCHK(hipSetDevice(e->device));
    int rc = sync_all(e);
    if (rc) return rc;
    const size_t n = (size_t)e->n;
    const PParams& k = e->pp;
    if (x) HIPCHK(hipMemcpy(k.x, x, n * 4, hipMemcpyHostToDevice));
    if (y) HIPCHK(hipMemcpy(k.y, y, n * 4, hipMemcpyHostToDevice));
    if (z) HIPCHK(hipMemcpy(k.z, z, n * 4, hipMemcpyHostToDevice));
    if (yaw) HIPCHK(hipMemcpy(k.yaw, yaw, n * 4, hipMemcpyHostToDevice));
    if (v) HIPCHK(hipMemcpy(k.v, v, n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(k.pending, 0, n));
    HIPCHK(hipMemset(k.done, 0, n));
    return TRS_OK;
}

TRS_EXPORT int trs_locate(trs_env* e, const double* h_xyz, int nq, int32_t* h_idx)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    if (nq < 0 || (nq && (!h_xyz || !h_idx))) return fail(TRS_ERR_ARG, "bad query");
    if (nq == 0) return TRS_OK;
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }
    if (nq > e->loc_cap) {
        HIPCHK(hipStreamSynchronize(e->sP));
        (void)hipFree(e->loc_q); (void)hipFree(e->loc_out); e->loc_q = nullptr; e->loc_out = nullptr; e->loc_cap = 0;
        HIPCHK(hipMalloc((void**)&e->loc_q, (size_t)nq * 24));
        HIPCHK(hipMalloc((void**)&e->loc_out, (size_t)nq * 4));
        e->loc_cap = nq;
    }
    HIPCHK(hipMemcpyAsync(e->loc_q, h_xyz, (size_t)nq * 24, hipMemcpyHostToDevice, e->sP));
    constexpr int kW = kLocBlock / 64;
    int grid = (nq + kW - 1) / kW;
    grid = std::min(grid, e->cu_count * 2);
    const PParams& pk = e->pp;
    const NearParams near{pk.np, pk.off_py, pk.off_pz, pk.off_gstart, pk.off_gpts, pk.grid_nx, pk.grid_nz, pk.grid_x0, pk.grid_z0};
    hipLaunchKernelGGL(trs_locate_kernel, dim3(grid), dim3(kLocBlock), pk.blob_bytes, e->sP,
                       (const unsigned char*)e->blob_p, pk.blob_bytes, near, (const double*)e->loc_q, nq, e->loc_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_idx, e->loc_out, (size_t)nq * 4, hipMemcpyDeviceToHost, e->sP));
    HIPCHK(hipStreamSynchronize(e->sP));
    return TRS_OK;
}


// ---- image path -----------------------------------------------------------------------------------------

namespace {

int check_pre(const trs_pre_config* c)
{
    if (!c || c->struct_size != sizeof(trs_pre_config)) return fail(TRS_ERR_ARG, "trs_pre_config.struct_size mismatch");
    if (c->edge_detection_enabled && (c->edge_dst_channel < 0 || c->edge_dst_channel > 2)) return fail(TRS_ERR_ARG, "edge_dst_channel out of range");
    if (c->n_filters < 0 || c->n_filters > 4) return fail(TRS_ERR_ARG, "n_filters out of range");
    for (int f = 0; f < c->n_filters; ++f)
        if (c->dst_channel[f] < 0 || c->dst_channel[f] > 2) return fail(TRS_ERR_ARG, "dst_channel out of range");
    return TRS_OK;
}

int ensure_hsv_table(trs_env* e)
{
    if (e->hsv_tab) return TRS_OK;
    int tab[512];
    tab[0] = tab[256] = 0;
    for (int i = 1; i < 256; ++i) {          // OpenCV's sdiv_table / hdiv_table180, hsv_shift = 12
        tab[i] = (int)std::lrint((255 << 12) / (1.0 * i));
        tab[256 + i] = (int)std::lrint((180 << 12) / (6.0 * i));
    }
    HIPCHK(hipMalloc((void**)&e->hsv_tab, sizeof tab));
    HIPCHK(hipMemcpy(e->hsv_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    return TRS_OK;
}

// the tables the DYN step kernels stage into LDS (FParams::tabs): OpenCV's reciprocals | the in-range byte masks of THIS filter | sel
int upload_dyn_tables(trs_env* e, const trs_pre_config& c)
{
    int rc = ensure_hsv_table(e);
    if (rc) return rc;
    std::vector<unsigned> t(kDynTabWords, 0u);
    HIPCHK(hipMemcpy(t.data(), e->hsv_tab, 512 * sizeof(int), hipMemcpyDeviceToHost));
    unsigned lo[4], hi[4]; int dc[4];
    for (int k = 0; k < 4; ++k) {
        lo[k] = c.hsv_lo[k][0] | (c.hsv_lo[k][1] << 8) | (c.hsv_lo[k][2] << 16);
        hi[k] = c.hsv_hi[k][0] | (c.hsv_hi[k][1] << 8) | (c.hsv_hi[k][2] << 16);
        dc[k] = c.dst_channel[k];
    }
    unsigned sel = 0;
    for (int i = 0; i < 768; ++i) t[512 + i] = range_byte_entry(lo, hi, dc, c.n_filters, i, &sel);
    t[512 + 768] = sel;
    for (int i = 0; i < 256; ++i) {                                           // class counts of a 4-pixel pack: n0 | n1 << 8 | n2 << 16 | n3 << 24
        unsigned cnt = 0;
        for (int k = 0; k < 4; ++k) cnt += 1u << (8 * ((i >> (2 * k)) & 3));
        t[trsim::kDynCntAt + i] = cnt;
    }
    { int rq = quiesce(e); if (rq) return rq; HIPCHK(hipStreamSynchronize(e->sP)); }   // a running kernel may still be staging the old tables
    if (!e->dyn_tab) HIPCHK(hipMalloc((void**)&e->dyn_tab, kDynTabWords * sizeof(unsigned)));
    HIPCHK(hipMemcpy(e->dyn_tab, t.data(), kDynTabWords * sizeof(unsigned), hipMemcpyHostToDevice));
    return TRS_OK;
}

// ImgPreprocessing.__process of ONE colour (img_preprocessing.py:37-74,92-99 without dynamic brightness and Canny):
// the host twin of trs_preprocess_kernel's per-pixel arithmetic, used to filter the rasteriser's palette.
uint32_t filter_colour(const trs_pre_config& c, uint32_t bgr)
{
    int t[3];
    for (int ch = 0; ch < 3; ++ch) {
        float x = (float)((bgr >> (8 * ch)) & 255u);
        x = x - c.contrast_offset;
        x = x * c.contrast_ratio;
        x = x + c.contrast_offset;
        x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
        t[ch] = (int)x;
    }
    int o[3] = {t[0], t[1], t[2]};
    if (c.color_filter_enabled) {
        const int r = t[0], g = t[1], b = t[2];
        const int v = std::max(r, std::max(g, b)), vmin = std::min(r, std::min(g, b)), diff = v - vmin;
        const int sdiv = v ? (int)std::lrint((255 << 12) / (1.0 * v)) : 0;
        const int hdiv = diff ? (int)std::lrint((180 << 12) / (6.0 * diff)) : 0;
        const int sat = (diff * sdiv + (1 << 11)) >> 12;
        int h = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
        h = (h * hdiv + (1 << 11)) >> 12;
        if (h < 0) h += 180;
        const int hh = std::min(h, 255), ss = std::min(sat, 255);
        for (int f = 0; f < c.n_filters; ++f) {
            const bool in = hh >= c.hsv_lo[f][0] && hh <= c.hsv_hi[f][0] && ss >= c.hsv_lo[f][1] && ss <= c.hsv_hi[f][1] &&
                            v >= c.hsv_lo[f][2] && v <= c.hsv_hi[f][2];
            o[c.dst_channel[f]] = in ? 255 : 0;
        }
    }
    return (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16);
}

int leading_uniform_rows(const std::vector<uint32_t>& pal, int H)
{
    int u = 0;
    while (u < H && pal[4 * u] == pal[4 * u + 1] && pal[4 * u] == pal[4 * u + 2] && pal[4 * u] == pal[4 * u + 3]) ++u;
    return u;
}

// (re)write the palette of the raster LDS image: raw, or filtered when a frame filter is set
int upload_palette(trs_env* e)
{
    if (!e->track_loaded || !e->cfg.render) return TRS_OK;
    std::vector<uint32_t> pal(e->tab.palette);
    if (e->has_frame_filter && !e->filter_dynamic)             // dynamic brightness: the kernel filters a per-env palette itself
        for (auto& c : pal) c = filter_colour(e->frame_filter, c);
    e->rp.uni_rows = e->hilly ? 0 : leading_uniform_rows(pal, e->H);   // (a track with elevation: which rows are sky depends on the env and the frame)
    e->uniform_ok[0] = e->uniform_ok[1] = false;               // (the closed pilot loop's steps skip rows an earlier step wrote: not across a palette change)
    { int rq = sync_all(e); if (rq) return rq; }               // frames in flight keep the palette they were launched with
    HIPCHK(hipMemcpy(e->blob_r + e->rp.off_pal, pal.data(), pal.size() * 4, hipMemcpyHostToDevice));
    if (e->hilly) {
        // a track with elevation: the kernels blend a row's ground colours per env and frame and run the static filter on each (hill_filter_colour); the sky colours and
        // the far colour are constants: filtered here
        const bool flt = e->has_frame_filter && !e->filter_dynamic;
        trsim::HillBlock& hb = e->hill_host;
        std::vector<uint32_t> sky(e->tab.sky);
        hb.far_rgb = e->tab.far_rgb;
        hb.filt = flt ? 1 : 0;
        if (flt) {
            const trs_pre_config& c = e->frame_filter;
            for (auto& v : sky) v = filter_colour(c, v);
            hb.far_rgb = filter_colour(c, hb.far_rgb);
            { int rc = ensure_hsv_table(e); if (rc) return rc; }
            hb.f_color = c.color_filter_enabled; hb.f_nfilters = c.n_filters; hb.f_contrast = c.contrast_ratio; hb.f_offset = c.contrast_offset;
            for (int k = 0; k < 4; ++k) {
                hb.f_lo[k] = (unsigned)(c.hsv_lo[k][0] | (c.hsv_lo[k][1] << 8) | (c.hsv_lo[k][2] << 16));
                hb.f_hi[k] = (unsigned)(c.hsv_hi[k][0] | (c.hsv_hi[k][1] << 8) | (c.hsv_hi[k][2] << 16));
                hb.f_dst[k] = c.dst_channel[k];
            }
            hb.hsv_tab = e->hsv_tab;
        }
        HIPCHK(hipMemcpy(e->blob_r + hb.off_sky, sky.data(), sky.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->blob_r + trsim::hill_block_offset(e->rp.blob_bytes), &hb, sizeof hb, hipMemcpyHostToDevice));
    }
    return TRS_OK;
}

int ensure_tmp(trs_env* e, size_t frames)
{
    if (frames <= e->tmp_cap) return TRS_OK;
    HIPCHK(hipStreamSynchronize(e->sP));
    (void)hipFree(e->tmp_in); (void)hipFree(e->tmp_out); (void)hipFree(e->tmp_f);
    e->tmp_in = e->tmp_out = nullptr; e->tmp_f = nullptr; e->tmp_cap = 0;
    const size_t fb = (size_t)e->H * e->W * 3;
    HIPCHK(hipMalloc((void**)&e->tmp_in, frames * fb));
    HIPCHK(hipMalloc((void**)&e->tmp_out, frames * fb));
    HIPCHK(hipMalloc((void**)&e->tmp_f, frames * fb * sizeof(float)));
    e->tmp_cap = frames;
    return TRS_OK;
}

// device scratch of the *_host control glue, owned by the handle (the N = 1 Car loop calls these every tick)
int ensure_glue(trs_env* e, size_t bytes)
{
    if (bytes <= e->glue_bytes) return TRS_OK;
    int rc = sync_all(e);
    if (rc) return rc;
    (void)hipFree(e->glue); e->glue = nullptr; e->glue_bytes = 0;
    const size_t cap = std::max<size_t>(align_up(bytes, 256), 4096);
    HIPCHK(hipMalloc((void**)&e->glue, cap));
    e->glue_bytes = cap;
    return TRS_OK;
}

const uint8_t* latest_frame(const trs_env* e) { return e->cfg.render ? e->img[(e->step_count + 1) & 1] : nullptr; }

}  // namespace

TRS_EXPORT void trs_default_pre_config(trs_pre_config* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->brightness_baseline = 550.0; c->contrast_ratio = 1.0f; c->contrast_offset = 125.0f;
    c->n_filters = 2;                                       // core/config.py:23-24: white and yellow
    const uint8_t lo[2][3] = {{0, 0, 130}, {25, 180, 155}}, hi[2][3] = {{180, 64, 255}, {43, 255, 255}};
    std::memcpy(c->hsv_lo, lo, sizeof lo); std::memcpy(c->hsv_hi, hi, sizeof hi);
    c->dst_channel[0] = 0; c->dst_channel[1] = 1;
    c->edge_threshold_a = 60; c->edge_threshold_b = 100; c->edge_dst_channel = 2;
}

TRS_EXPORT int trs_set_frame_filter(trs_env* e, const trs_pre_config* c)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    if (!e->cfg.render) return fail(TRS_ERR_STATE, "the env has no camera (cfg.render == 0)");
    if (c) {
        int rc = check_pre(c);
        if (rc) return rc;
        if (c->edge_detection_enabled) return fail(TRS_ERR_ARG, "the Canny layer is a neighbourhood operator: not a palette filter, use trs_preprocess");
        if (c->dynamic_brightness && e->track_loaded && e->hilly)
            return fail(TRS_ERR_STATE, "the loaded track has elevation: a frame's palette is evaluated per env inside the kernels there, and the dynamic-brightness filter behind the "
                                       "rasteriser is not built for that; the static filter works, or use trs_preprocess on the rendered frames");
        if (c->dynamic_brightness) {
            const int rpp = kRasterThreads / (e->W / 4);
            if (rpp < 1 || (79 + rpp - 1) / rpp > 16) return fail(TRS_ERR_LIMIT, "image too wide for the in-kernel dynamic-brightness filter (class bits of the brightness rows live in 4 registers), use trs_preprocess");
            if (e->track_loaded && e->max_steps_dyn < 1) return fail(TRS_ERR_LIMIT, "no LDS left beside this track's tables for the in-kernel dynamic-brightness palette, use trs_preprocess");
            // resident mode: the worker keeps its env state and hand-off ring in LDS as well — refuse HERE, with the reason, what every later
            // trs_step would otherwise refuse as "too many envs per workgroup" 
            if (e->track_loaded && trsim::resident_on(e) && !trsim::resident_fits_dynamic_filter(e))
                return fail(TRS_ERR_LIMIT, "the resident worker's LDS (tables + env state + hand-off ring) leaves no room for the dynamic-brightness palettes of a batch: "
                                           "select TRS_STEP_LAUNCH for this filter, or use trs_preprocess");
        }
        if (c->dynamic_brightness) { HIPCHK(hipSetDevice(e->device)); rc = upload_dyn_tables(e, *c); if (rc) return rc; }
        e->frame_filter = *c; e->has_frame_filter = true; e->filter_dynamic = c->dynamic_brightness != 0;
    } else {
        e->has_frame_filter = false;
    }
    HIPCHK(hipSetDevice(e->device));
    return upload_palette(e);
}

TRS_EXPORT int trs_preprocess(trs_env* e, const trs_pre_config* c, const uint8_t* d_src, uint8_t* d_dst, int n_images, const uint8_t** d_out)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    int rc = check_pre(c);
    if (rc) return rc;
    if (n_images < 0) return fail(TRS_ERR_ARG, "n_images < 0");
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }
    if (!d_src) {
        if (!latest_frame(e) || n_images != e->n) return fail(TRS_ERR_ARG, "latest-frame source needs n_images == n_envs and a camera");
        d_src = latest_frame(e);
    }
    if (!d_dst) {
        if (n_images > e->n) return fail(TRS_ERR_ARG, "own buffer holds n_envs frames");
        if (!e->pre) HIPCHK(hipMalloc((void**)&e->pre, (size_t)e->n * e->H * e->W * 3));
        d_dst = e->pre;
    }
    if (d_out) *d_out = d_dst;
    if (n_images == 0) return TRS_OK;
    rc = ensure_hsv_table(e);
    if (rc) return rc;
    PreParams p{};
    p.src = d_src; p.dst = d_dst; p.hsv_tab = e->hsv_tab;
    p.n_img = n_images; p.H = e->H; p.W = e->W; p.gpr = e->W / 4; p.gpe = p.gpr * e->H;
    p.r0 = std::min(40, e->H); p.r1 = std::min(119, e->H);                 // img[40:119] (img_preprocessing.py:88)
    p.dynamic = c->dynamic_brightness; p.color = c->color_filter_enabled; p.n_filters = c->n_filters;
    p.contrast = c->contrast_ratio; p.offset = c->contrast_offset; p.baseline = c->brightness_baseline;
    for (int f = 0; f < 4; ++f) {
        p.lo[f] = c->hsv_lo[f][0] | (c->hsv_lo[f][1] << 8) | (c->hsv_lo[f][2] << 16);
        p.hi[f] = c->hsv_hi[f][0] | (c->hsv_hi[f][1] << 8) | (c->hsv_hi[f][2] << 16);
        p.dst_ch[f] = c->dst_channel[f];
    }
    if (c->edge_detection_enabled) {
        p.edge = 1; p.edge_ch = c->edge_dst_channel;
        p.edge_low = std::min(c->edge_threshold_a, c->edge_threshold_b);          // cv::Canny swaps the thresholds into order
        p.edge_high = std::max(c->edge_threshold_a, c->edge_threshold_b);
        const size_t npx = (size_t)e->H * e->W;
        p.off_mag = (int)align_up(npx * 3, 16);
        p.off_map = p.off_mag + (int)align_up((size_t)(e->H + 2) * (e->W + 8) * 2, 16);   // rows of W + 8 int16: a 4-pixel group's values are 8-byte aligned
        const size_t work = (size_t)p.off_map + align_up(npx, 16) + 16;
        const int tables = kEdgeTables;
        const int grid = std::min(n_images, e->cu_count);
        if (work + tables <= 160 * 1024) {                                    // whole frame in LDS
            p.off_tab = (int)work;
            const int lds = p.off_tab + tables;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_preprocess_edge_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(trs_preprocess_edge_kernel<false>, dim3(grid), dim3(kEdgeBlock), lds, e->sP, p);
        } else {                                                              // work arrays in a global scratch (stays in L2), tables in LDS
            p.off_tab = 0;
            p.scratch_stride = align_up(work, 256);
            const size_t need = p.scratch_stride * (size_t)grid;
            if (e->edge_scratch_bytes < need) {
                HIPCHK(hipStreamSynchronize(e->sP));
                (void)hipFree(e->edge_scratch); e->edge_scratch = nullptr; e->edge_scratch_bytes = 0;
                HIPCHK(hipMalloc((void**)&e->edge_scratch, need));
                e->edge_scratch_bytes = need;
            }
            p.scratch = e->edge_scratch;
            hipLaunchKernelGGL(trs_preprocess_edge_kernel<true>, dim3(grid), dim3(kEdgeBlock), tables, e->sP, p);
        }
        HIPCHK(hipGetLastError());
        return TRS_OK;
    }
    const int block = p.gpe >= 8192 ? 1024 : 256;                          // large frames: more waves per frame (one workgroup per frame cannot fill the chip otherwise)
    const int grid = std::min(n_images, e->cu_count * (block == 256 ? 8 : 2));
    if (p.dynamic && (!p.color || block == 256)) hipLaunchKernelGGL(trs_preprocess_kernel<true>, dim3(grid), dim3(block), 0, e->sP, p);   // the frame held in registers between the sums and the trim (with the masks' arithmetic only on 256-thread workgroups: on 1024 threads the registers cost more occupancy than the second read)
    else hipLaunchKernelGGL(trs_preprocess_kernel<false>, dim3(grid), dim3(block), 0, e->sP, p);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

TRS_EXPORT int trs_preprocess_host(trs_env* e, const trs_pre_config* c, const uint8_t* h_src, uint8_t* h_dst, int n_images)
{
    if (!e || !h_src || !h_dst || n_images < 0) return fail(TRS_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(e->device));
    if (n_images == 0) return TRS_OK;
    { int rq = quiesce(e); if (rq) return rq; }
    int rc = ensure_tmp(e, (size_t)n_images);
    if (rc) return rc;
    const size_t bytes = (size_t)n_images * e->H * e->W * 3;
    HIPCHK(hipMemcpyAsync(e->tmp_in, h_src, bytes, hipMemcpyHostToDevice, e->sP));
    rc = trs_preprocess(e, c, e->tmp_in, e->tmp_out, n_images, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_dst, e->tmp_out, bytes, hipMemcpyDeviceToHost, e->sP));
    e->d2h_bytes += bytes; e->h2d_bytes += bytes;
    HIPCHK(hipStreamSynchronize(e->sP));
    return TRS_OK;
}

TRS_EXPORT int trs_normalize(trs_env* e, const uint8_t* d_src, float* d_dst, int n_images)
{
    if (!e || !d_dst || n_images < 0) return fail(TRS_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }
    if (!d_src) {
        if (!latest_frame(e) || n_images != e->n) return fail(TRS_ERR_ARG, "latest-frame source needs n_images == n_envs and a camera");
        d_src = latest_frame(e);
    }
    const size_t n4 = (size_t)n_images * e->H * e->W * 3 / 4;            // W % 4 == 0 -> whole dwords
    if (n4 == 0) return TRS_OK;
    const int grid = (int)std::min<size_t>((n4 + 255) / 256, (size_t)e->cu_count * 16);
    hipLaunchKernelGGL(trs_normalize_kernel, dim3(grid), dim3(256), 0, e->sP, reinterpret_cast<const uint32_t*>(d_src), reinterpret_cast<float4*>(d_dst), n4);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

TRS_EXPORT int trs_normalize_host(trs_env* e, const uint8_t* h_src, float* h_dst, int n_images)
{
    if (!e || !h_src || !h_dst || n_images < 0) return fail(TRS_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(e->device));
    if (n_images == 0) return TRS_OK;
    { int rq = quiesce(e); if (rq) return rq; }
    int rc = ensure_tmp(e, (size_t)n_images);
    if (rc) return rc;
    const size_t bytes = (size_t)n_images * e->H * e->W * 3;
    HIPCHK(hipMemcpyAsync(e->tmp_in, h_src, bytes, hipMemcpyHostToDevice, e->sP));
    rc = trs_normalize(e, e->tmp_in, e->tmp_f, n_images);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h_dst, e->tmp_f, bytes * sizeof(float), hipMemcpyDeviceToHost, e->sP));
    e->d2h_bytes += bytes * sizeof(float); e->h2d_bytes += bytes;
    HIPCHK(hipStreamSynchronize(e->sP));
    return TRS_OK;
}

TRS_EXPORT int trs_driver_assist(trs_env* e, int mode, double k, float* d_st, float* d_th, float* d_br, const float* d_sp, int n)
{
    if (!e || !d_st || !d_th || !d_br || n < 0 || (mode != 0 && mode != 1)) return fail(TRS_ERR_ARG, "bad argument (mode 0 = steering, 1 = speed)");
    if (!d_sp) { if (n != e->n) return fail(TRS_ERR_ARG, "the env's own speed needs n == n_envs"); d_sp = e->pp.speed; }
    HIPCHK(hipSetDevice(e->device));
    if (n == 0) return TRS_OK;
    { int rq = quiesce(e); if (rq) return rq; }
    hipLaunchKernelGGL(trs_driver_assist_kernel, dim3((n + 255) / 256), dim3(256), 0, e->sP, mode, k, d_st, d_th, d_br, d_sp, n);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

TRS_EXPORT int trs_driver_assist_host(trs_env* e, int mode, double k, float* h_st, float* h_th, float* h_br, const float* h_sp, int n)
{
    if (!e || !h_st || !h_th || !h_br || !h_sp || n < 0) return fail(TRS_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(e->device));
    if (n == 0) return TRS_OK;
    int rc = quiesce(e);
    if (!rc) rc = ensure_glue(e, (size_t)n * 16);
    if (rc) return rc;
    float* d = e->glue;
    float *ds = d, *dt = d + n, *db = d + 2 * (size_t)n, *dp = d + 3 * (size_t)n;
    hipError_t err = hipSuccess;
    const float* srcs[4] = {h_st, h_th, h_br, h_sp};
    float* dsts[4] = {ds, dt, db, dp};
    for (int a = 0; a < 4 && err == hipSuccess; ++a) err = hipMemcpyAsync(dsts[a], srcs[a], (size_t)n * 4, hipMemcpyHostToDevice, e->sP);
    rc = err == hipSuccess ? trs_driver_assist(e, mode, k, ds, dt, db, dp, n) : fail(TRS_ERR_DEVICE, hipGetErrorString(err));
    float* outs[3] = {h_st, h_th, h_br};
    for (int a = 0; a < 3 && rc == TRS_OK; ++a)
        if (hipMemcpyAsync(outs[a], dsts[a], (size_t)n * 4, hipMemcpyDeviceToHost, e->sP) != hipSuccess) rc = fail(TRS_ERR_DEVICE, "copy back failed");
    if (hipStreamSynchronize(e->sP) != hipSuccess && rc == TRS_OK) rc = fail(TRS_ERR_DEVICE, "stream synchronisation failed");
    return rc;
}

TRS_EXPORT void trs_default_mux_config(trs_mux_config* c)
{
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->throttle_lock_enabled = 0; c->throttle_lock_value = 1.0f; c->throttle_lock_ticks = 100;   // core/config.py:57-59 at 20 Hz
    c->steering_lock_enabled = 0; c->steering_lock_value = 0.0f; c->steering_lock_ticks = 60;    // core/config.py:61-63
}

static int mux_state_ready(trs_env* e)
{
    { int rq = quiesce(e); if (rq) return rq; }
    if (e->mux_state) return TRS_OK;
    HIPCHK(hipMalloc((void**)&e->mux_state, (size_t)e->n * kMuxWords * sizeof(int32_t)));
    hipLaunchKernelGGL(trs_control_mux_init_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->sP, e->mux_state, e->n);
    HIPCHK(hipGetLastError());
    e->mux_tick = 0;
    return TRS_OK;
}

TRS_EXPORT int trs_control_mux_reset(trs_env* e)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }
    if (e->mux_state) { HIPCHK(hipFree(e->mux_state)); e->mux_state = nullptr; }
    return mux_state_ready(e);
}

TRS_EXPORT int trs_control_mux(trs_env* e, const trs_mux_config* c, const uint8_t* d_mode, const float* d_us, const float* d_ut, const float* d_ub,
                               const float* d_as, const float* d_at, const float* d_ab, float* d_os, float* d_ot, float* d_ob, int n)
{
    if (!e || !c || !d_mode || !d_us || !d_ut || !d_ub || !d_as || !d_at || !d_ab || !d_os || !d_ot || !d_ob) return fail(TRS_ERR_ARG, "null argument");
    if (c->struct_size != sizeof(trs_mux_config)) return fail(TRS_ERR_ARG, "trs_mux_config.struct_size mismatch");
    if (n < 0 || n > e->n) return fail(TRS_ERR_ARG, "n must be in [0, n_envs] (the lock state is kept per env)");
    if ((c->throttle_lock_enabled && c->throttle_lock_ticks < 1) || (c->steering_lock_enabled && c->steering_lock_ticks < 1))
        return fail(TRS_ERR_ARG, "lock ticks must be >= 1");
    HIPCHK(hipSetDevice(e->device));
    int rc = mux_state_ready(e);
    if (rc) return rc;
    MuxParams p{};
    p.mode = d_mode; p.us = d_us; p.ut = d_ut; p.ub = d_ub; p.as = d_as; p.at = d_at; p.ab = d_ab; p.os = d_os; p.ot = d_ot; p.ob = d_ob;
    p.state = e->mux_state; p.n = n; p.tick = e->mux_tick;
    p.en_t = c->throttle_lock_enabled != 0; p.ticks_t = c->throttle_lock_ticks; p.val_t = c->throttle_lock_value;
    p.en_s = c->steering_lock_enabled != 0; p.ticks_s = c->steering_lock_ticks; p.val_s = c->steering_lock_value;
    if (n > 0) {
        hipLaunchKernelGGL(trs_control_mux_kernel, dim3((n + 255) / 256), dim3(256), 0, e->sP, p);
        HIPCHK(hipGetLastError());
    }
    e->mux_tick += 1;
    return TRS_OK;
}

TRS_EXPORT int trs_control_mux_host(trs_env* e, const trs_mux_config* c, const uint8_t* h_mode, const float* h_us, const float* h_ut, const float* h_ub,
                                    const float* h_as, const float* h_at, const float* h_ab, float* h_os, float* h_ot, float* h_ob, int n)
{
    if (!e || !h_mode || !h_us || !h_ut || !h_ub || !h_as || !h_at || !h_ab || !h_os || !h_ot || !h_ob || n < 0) return fail(TRS_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(e->device));
    if (n == 0) return trs_control_mux(e, c, h_mode, h_us, h_ut, h_ub, h_as, h_at, h_ab, h_os, h_ot, h_ob, 0);
    const size_t nn = (size_t)n;
    int rc = quiesce(e);
    if (!rc) rc = ensure_glue(e, nn * 4 * 9 + nn);
    if (rc) return rc;
    float* d = e->glue;
    uint8_t* dm = reinterpret_cast<uint8_t*>(d + 9 * nn);
    const float* srcs[9] = {h_us, h_ut, h_ub, h_as, h_at, h_ab, h_os, h_ot, h_ob};
    hipError_t err = hipMemcpyAsync(dm, h_mode, nn, hipMemcpyHostToDevice, e->sP);
    for (int a = 0; a < 9 && err == hipSuccess; ++a) err = hipMemcpyAsync(d + a * nn, srcs[a], nn * 4, hipMemcpyHostToDevice, e->sP);
    rc = err == hipSuccess ? trs_control_mux(e, c, dm, d, d + nn, d + 2 * nn, d + 3 * nn, d + 4 * nn, d + 5 * nn, d + 6 * nn, d + 7 * nn, d + 8 * nn, n)
                           : fail(TRS_ERR_DEVICE, hipGetErrorString(err));
    float* outs[3] = {h_os, h_ot, h_ob};
    for (int a = 0; a < 3 && rc == TRS_OK; ++a)
        if (hipMemcpyAsync(outs[a], d + (6 + a) * nn, nn * 4, hipMemcpyDeviceToHost, e->sP) != hipSuccess) rc = fail(TRS_ERR_DEVICE, "copy back failed");
    if (hipStreamSynchronize(e->sP) != hipSuccess && rc == TRS_OK) rc = fail(TRS_ERR_DEVICE, "stream synchronisation failed");
    return rc;
}

TRS_EXPORT int trs_map_info_get(trs_env* e, trs_map_info* o)
{
    if (!e || !o || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    *o = e->tab.info;
    o->lds_bytes = e->cfg.render ? e->lds_step : e->lds_p;
    return TRS_OK;
}

TRS_EXPORT int trs_sync(trs_env* e)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    if (resident_steps(e)) { int rw = trsim::resident_wait(e); return rw ? rw : trsim::check_fault(e); }   // posted steps complete; the worker stays
    return sync_all(e);
}

TRS_EXPORT int trs_step_wait(trs_env* e, const float* d_st, const float* d_th, const float* d_br, const uint8_t* d_rs, int n_steps)
{
    const int rc = trs_step(e, d_st, d_th, d_br, d_rs, n_steps);
    return rc ? rc : trs_sync(e);
}

TRS_EXPORT int trs_quiesce(trs_env* e)
{
    if (!e) return fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    return quiesce(e);
}

TRS_EXPORT int trs_event_record(trs_env* e, int slot)
{
    if (!e || slot < 0 || slot >= 8) return fail(TRS_ERR_ARG, "bad event slot");
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }            // an event behind a resident worker would wait for the worker to leave anyway
    HIPCHK(hipEventRecord(e->ev[slot], e->sP));
    return TRS_OK;
}

TRS_EXPORT int trs_event_elapsed_ms(trs_env* e, int a, int b, float* ms)
{
    if (!e || !ms || a < 0 || a >= 8 || b < 0 || b >= 8) return fail(TRS_ERR_ARG, "bad event slot");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipEventSynchronize(e->ev[b]));
    HIPCHK(hipEventElapsedTime(ms, e->ev[a], e->ev[b]));
    return TRS_OK;
}

TRS_EXPORT int trs_scratch(trs_env* e, int slot, size_t bytes, void** d_out)
{
    if (!e || !d_out || slot < 0 || slot >= 32) return fail(TRS_ERR_ARG, "bad scratch slot (0..31) / null argument");
    HIPCHK(hipSetDevice(e->device));
    if (bytes > e->scratch_bytes[slot]) {
        int rc = sync_all(e);                                // nothing in flight may still use the old buffer
        if (rc) return rc;
        (void)hipFree(e->scratch[slot]); e->scratch[slot] = nullptr; e->scratch_bytes[slot] = 0;
        const size_t cap = align_up(std::max<size_t>(bytes, 256), 256);
        HIPCHK(hipMalloc(&e->scratch[slot], cap));
        HIPCHK(hipMemset(e->scratch[slot], 0, cap));
        e->scratch_bytes[slot] = cap;
    }
    *d_out = e->scratch[slot];
    return TRS_OK;
}

TRS_EXPORT int trs_upload(trs_env* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e || (bytes && (!d_dst || !h_src))) return fail(TRS_ERR_ARG, "null argument");
    if (!bytes) return TRS_OK;
    HIPCHK(hipSetDevice(e->device));
    { int rq = quiesce(e); if (rq) return rq; }
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->sP));   // pageable source: staged before the call returns
    e->h2d_bytes += bytes;
    return TRS_OK;
}

TRS_EXPORT int trs_counters(trs_env* e, uint64_t out[4])
{
    if (!e || !out) return fail(TRS_ERR_ARG, "null argument");
    out[0] = e->d2h_bytes; out[1] = e->h2d_bytes; out[2] = e->step_count; out[3] = 0;
    return TRS_OK;
}

TRS_EXPORT const char* trs_last_error(void) { return g_err.c_str(); }

// ---- internal accessors for trsim_pilot.hip (not exported) ----------------------------------------------
bool trs_internal_view(trs_env* e, TrsEnvView* v)
{
    if (!e || !v) return false;
    if (quiesce(e)) return false;                            // the pilot's kernels go onto the handle's stream
    v->device = e->device; v->n = e->n; v->H = e->H; v->W = e->W; v->render = e->cfg.render;
    v->stream = e->sP;
    v->latest_frame = (e->cfg.render && e->step_count > 0) ? e->img[(e->step_count + 1) & 1] : nullptr;
    v->speed = e->pp.speed; v->seg_idx = e->pp.seg_idx; v->n_points = e->pp.np;
    v->ctl_steer = e->ctl_steer; v->ctl_thr = e->ctl_thr; v->ctl_brk = e->ctl_brk;
    v->step_count = e->step_count;
    v->stats = e->stats;
    return true;
}
void** trs_internal_pilot_slot(trs_env* e) { return e ? &e->pilot : nullptr; }
const trs_pilot_tuning* trs_internal_pilot_tuning(trs_env* e) { return e && e->has_pilot_tuning ? &e->pilot_tuning : nullptr; }
void trs_internal_set_pilot_tuning(trs_env* e, const trs_pilot_tuning* t)
{
    if (!e) return;
    e->has_pilot_tuning = t != nullptr;
    if (t) e->pilot_tuning = *t;
}
int trs_internal_fail(int code, const std::string& msg) { return fail(code, msg); }
void trs_internal_count(trs_env* e, uint64_t d2h, uint64_t h2d) { if (e) { e->d2h_bytes += d2h; e->h2d_bytes += h2d; } }
// one env step by LAUNCH whatever the handle's step mode: the pilot loop's kernels need the CUs' LDS, which a resident worker
// would hold, and its controls are produced on the handle's stream right in front of the step
int trs_internal_step_launch(trs_env* e, const float* d_st, const float* d_th, const float* d_br)
{
    if (!e || !e->track_loaded) return fail(TRS_ERR_STATE, "no track loaded");
    { int rq = quiesce(e); if (rq) return rq; }
    // (the loop's own steps: the uniform rows of a frame buffer — sky, beyond the far plane — are written by the first step that renders into it and kept after that:
    // 41 % of a frame's bytes at the default camera, 4.8 of the 16.5 us of this step at 1024 x 120x160, ~20 of 48 us at 512 x 240x320 + depth)
    const int rc = e->cfg.render ? run_camera_steps(e, d_st, d_th, d_br, nullptr, 0, 1, 1, true) : run_physics_steps(e, d_st, d_th, d_br, nullptr, 0, 1, 1);
    if (!rc) trsim::resident_note_launch(e);               // resident mode selected: this step has no completion flag, trs_sync / the copies wait for the stream
    return rc;
}
int trs_internal_replay_launch(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, uint64_t step)
{
    if (e->cfg.render) return launch_step(e, st, th, br, rs, synth, 1, 0, 0, step);
    PParams p = e->pp;
    p.ctl_steer = st; p.ctl_thr = th; p.ctl_brk = br; p.ctl_reset = rs; p.ctl_stride = 0;
    p.synth = synth; p.n_steps = 1; p.write_cam = 0; p.step_off = (uint32_t)step;
    hipLaunchKernelGGL(trs_physics_kernel, dim3((e->n + kPhysBlock / 64 - 1) / (kPhysBlock / 64)), dim3(kPhysBlock), e->lds_p, e->sP, p);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}
void trs_internal_note(const std::string& msg) { g_err = msg; }
int trsim::sync_handle(trs_env* e) { return sync_all(e); }
int trsim::quiesce_handle(trs_env* e) { return quiesce(e); }
int trsim::check_fault(trs_env* e)
{
    if (e->fault && __atomic_load_n(e->fault, __ATOMIC_ACQUIRE) != 0ull)
        return fail(TRS_ERR_DEVICE, "a step kernel found its dynamic LDS segment at a non-zero offset and refused to run: frames and state are stale");
    return TRS_OK;
}

#ifdef TRS_DEBUG_PROBES
// Debug build only (scripts/det_probe.py; never part of libtrsim.so): fill the whole LDS of every CU with a pattern on the
// handle's stream, so that a kernel reading LDS bytes it has not (yet) written shows a wrong result instead of silently
// re-reading what its own previous launch left there.
__global__ __launch_bounds__(1024) void trs_debug_poison_kernel(unsigned pattern, int bytes, int spin)
{
    u4v* d = reinterpret_cast<u4v*>(smem);
    for (int i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = (u4v)(pattern);
    __syncthreads();
    const long long t0 = clock64();
    while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);            // stay resident so that every CU gets a workgroup
}

TRS_EXPORT int trs_debug_poison_lds(trs_env* e, unsigned pattern, int bytes)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_debug_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipLaunchKernelGGL(trs_debug_poison_kernel, dim3(512), dim3(1024), bytes, e->sP, pattern, bytes, 20000);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}

// ... and: occupy a slice of every CU's LDS from another stream for `cycles` clocks (touch = 1: keep rewriting that slice), so
// that workgroups launched meanwhile get an LDS allocation that does not start at the CU's LDS address 0.
__global__ __launch_bounds__(64) void trs_debug_spin_kernel(int bytes, long long cycles, int touch)
{
    u4v* d = reinterpret_cast<u4v*>(smem);
    const long long t0 = clock64();
    unsigned k = 0;
    while (clock64() - t0 < cycles) {
        if (touch) for (int i = threadIdx.x; i < bytes / 16; i += 64) d[i] = (u4v)(0xDEAD0000u + k++);
        __builtin_amdgcn_s_sleep(8);
    }
}

TRS_EXPORT int trs_debug_spin(trs_env* e, int wgs, int bytes, long long cycles, int touch)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    static hipStream_t side = nullptr;
    HIPCHK(hipSetDevice(e->device));
    if (!side) HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(trs_debug_spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    hipLaunchKernelGGL(trs_debug_spin_kernel, dim3(wgs), dim3(64), bytes, side, bytes, cycles, touch);
    HIPCHK(hipGetLastError());
    return TRS_OK;
}
#endif
