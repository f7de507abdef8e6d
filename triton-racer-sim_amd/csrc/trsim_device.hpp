// trsim_device.hpp — device-side pieces shared by the step kernels (trsim_hip.hip) and the resident worker
// (trsim_resident.hip): the spec's arithmetic (include/trsim_spec.h), the wave-parallel nearest-point search
// (= reference LocationTracker.__find_closest, components/track_data_process.py:89-104), one env step on registers, and the
// per-thread raster walk.  Everything is force-inlined; both translation units compile their own copy.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/trsim.h"
#include "../../include/trsim_spec.h"

#ifndef TRS_ABLATE
#define TRS_ABLATE 0   /* 0 = product; 1/2 = timing-only diagnostic builds (scripts/ablate.sh), never shipped */
#endif

namespace trsim {

struct PParams {                        // physics kernel
    float *x, *y, *z, *yaw, *v, *speed, *cte, *ep_return, *last_return, *steer_filt;
    int32_t *seg_idx, *ep_len;
    uint8_t *done, *pending;
    const float *ctl_steer, *ctl_thr, *ctl_brk;
    const uint8_t* ctl_reset;
    int ctl_stride;                     // elements between the control arrays of consecutive steps of a launch (0 = the same controls every step)
    float4* cam;                        // [kRing][n_envs] camx, camz, sin, cos (cell units) for the raster kernel
    const unsigned char* blob;          // physics LDS image: px | py | pz | tangent
    const float* start_yaw;             // [np]
    const float* tangent_g;             // [np][2] global copy, used when the table does not fit in LDS
    unsigned long long* stats;          // [0] off-track events, [1] resets, [2] layout faults, [8..] diagnostics
    unsigned long long* fault;          // pinned host word: non-zero = a kernel refused to run (layout fault); checked at every synchronisation
    int n_envs, env_id_base, envs_per_wg, np;
    int off_py, off_pz, off_tan, blob_bytes, off_scratch, tan_in_lds;
    int off_gstart, off_gpts, grid_nx, grid_nz;   // nearest-point accelerator: uint16 cell starts / point lists in the LDS image
    double grid_x0, grid_z0;
    float map_x0f, map_z0f, inv_cellf;
    float dt, max_steer, inv_wheelbase, accel_max, drag_lin, roll_res, brake_max;
    float v_max, v_rev_max, offtrack_cte, offtrack_penalty, cam_fwd;
    int auto_reset, synth, n_steps, write_cam;
    uint32_t step_off;
    unsigned long long seed;
};

// What the kernels need on a track with elevation.  It lives in device memory BEHIND the raster LDS image's bytes (RParams::blob + hill_block_offset): the kernel
// arguments of the flat-track kernels are exactly what they were before tracks had elevation — arguments sit in scalar registers for the whole kernel, and even one
// more pointer shifted the flat single-step launch by 2-4 % (profiles/r05_hills.txt).  Only the HILLS instantiations ever form this address.
struct HillBlock {
    const float* vpitch;                // [np] the view pitch of a frame whose nearest raw track point is idx: (float)pitch + dpitch[idx], one binary32 addition (host)
    float* cam_pitch;                   // [kRing][n_envs] ring of the frames' view pitches beside PParams::cam (launch mode: the next launch's first frame)
    int off_sky;                        // raster LDS image: uint32 sky[H] (the sky colour of every row; filtered by the host when a frame filter is set)
    unsigned far_rgb;                   // (likewise)
    float inv_f, hh, cam_h_f, z_far_f, inv_zfar_f, fog_f, inv_cell_f;
    // the static frame filter (trs_set_frame_filter without dynamic brightness) on a track with elevation: the ground colours of a row are blended per env and
    // frame, so the filter of ONE colour (img_preprocessing.py:37-74,92-99: the host's filter_colour) runs on each of them in hill_row_build
    int filt, f_color, f_nfilters;
    float f_contrast, f_offset;
    unsigned f_lo[4], f_hi[4];          // packed h | s << 8 | v << 16
    int f_dst[4];
    const int* hsv_tab;                 // [512] OpenCV's sdiv | hdiv fixed-point reciprocals (global)
};

struct RParams {                        // raster side of the step kernel
    const unsigned char* blob;          // raster LDS image: map (pitched rows) @0 | rowtab | palette
    unsigned long long* stats;
    unsigned long long* fault;          // see PParams
    int n_envs, envs_per_wg;
    int H, W, gpr, gpe, rows_per_pass;  // gpr/gpe: 4-pixel groups per row / per env
    int map_w, map_h, map_pitch_b;
    int off_rowtab, off_pal, off_depth, blob_bytes;   // off_depth: float rowdepth[H] (z-depth per image row)
    int depth;                          // 1 = also write the binary32 z-depth frame
    int uni_rows;                       // leading image rows whose four class colours are equal (sky, beyond the far plane); 0 on a track with elevation
};
__host__ __device__ inline size_t hill_block_offset(int blob_bytes) { return ((size_t)blob_bytes + 63) & ~(size_t)63; }

struct FParams {                        // ImgPreprocessing with dynamic brightness, evaluated inside the step kernels (their DYN instantiations)
    double baseline;
    float contrast, offset;
    int color, n_filters;
    int lo[4], hi[4], dst_ch[4];
    int w0, w1;                         // brightness window: image rows [w0, w1) = img[40:119] (img_preprocessing.py:88)
    int lds_off;                        // LDS: uint32 penv[4][H][4] | int esum[2][4][3] | int dbar | (128 B) | dyn tables (kDynTabWords)
    const unsigned* tabs;               // global, kDynTabWords: int hsv[512] (OpenCV's sdiv | hdiv fixed-point reciprocals) | rngb[768] (byte masks of the
                                        // in-range tests per component value, range_byte_entry) | sel — staged into LDS once per launch (round 4: the
                                        // resident worker's raster waves then issue NO loads in their steady state)
};
constexpr int kDynTabWords = 512 + 768 + 4 + 256;   // ... | cnt[256]: the class counts of a row's 4-pixel pack (n0 | n1 << 8 | n2 << 16 | n3 << 24; round 4, see raster_dyn_batch phase A)
constexpr int kDynCntAt = 512 + 768 + 4;

}  // namespace trsim

namespace {

#ifndef TRS_RASTER_WAVES
#define TRS_RASTER_WAVES 8    /* raster waves per workgroup: 512 threads = 12 full rows of 40 groups per pass, 10 passes exactly at 120x160, and 12 waves balance over the 4 SIMDs (10 + 5 waves: 72.5 M env-steps/s, 8 + 4: 77.0 M; profiles/r01_step_kernel_waves_ab.txt) */
#endif
#ifndef TRS_PHYS_WAVES
#define TRS_PHYS_WAVES 4      /* physics waves per workgroup: one env per wave at 1024 envs */
#endif
constexpr int kBlock = 64 * (TRS_RASTER_WAVES + TRS_PHYS_WAVES);   // 12 waves
constexpr int kLocBlock = 1024;        // locate kernel: 16 waves = 16 queries in flight per workgroup
constexpr int kRing = 4;               // global camera-parameter ring: the last step of launch i is the first frame of launch i+1
[[maybe_unused]] constexpr int kRasterStampThread = 64 * TRS_RASTER_WAVES;  // diagnostic stamps: wave 0 and the first physics wave
constexpr int kRasterThreads = 64 * TRS_RASTER_WAVES;
constexpr int kPhysWaves = (kBlock - kRasterThreads) / 64;

using trsim::PParams;
using trsim::RParams;
using trsim::FParams;
using trsim::kDynTabWords;
using trsim::kDynCntAt;

// ---------------------------------------------------------------------------------------------
// device pieces of the spec

__device__ __forceinline__ void spec_sincos(float a, float& so, float& co)
{
    const float q = rintf(a * TRS_TWO_OVER_PI);
    float r = fmaf(q, -TRS_PIO2_HI, a);
    r = fmaf(q, -TRS_PIO2_LO, r);
    const float zz = r * r;
    const float ps = fmaf(fmaf(TRS_S0, zz, TRS_S1), zz, TRS_S2);
    const float s = fmaf(r * zz, ps, r);
    const float pc = fmaf(fmaf(TRS_C0, zz, TRS_C1), zz, TRS_C2);
    const float c = fmaf(zz * zz, pc, fmaf(zz, -0.5f, 1.0f));
    const int n = ((int)q) & 3;
    so = (n == 0) ? s : (n == 1) ? c : (n == 2) ? -s : -c;
    co = (n == 0) ? c : (n == 1) ? -s : (n == 2) ? -c : s;
}

// ... the same values with the quadrant as two selects and two sign-bit XORs (a negation IS the sign bit).  Written as a chain of ?: hipcc makes exec-masked
// branches of the quadrant (three s_and_saveexec per call).  A wave that has its SIMD to itself (the physics-only kernels) is faster without them: its step is a chain of
// latencies and every exec write stalls it (2.36 -> 1.86 -> 1.74 us per step, profiles/r05_physics_chain.txt); a physics wave that shares its SIMD with two raster waves is
// faster WITH them — most angles sit in quadrant 0 and the branch skips the work (the same rewrite cost the render kernels 1 % and the hilly step, which calls this per
// image row, 15 %).  So both forms exist, chosen per kernel at compile time.
__device__ __forceinline__ void spec_sincos_sel(float a, float& so, float& co)
{
    const float q = rintf(a * TRS_TWO_OVER_PI);
    float r = fmaf(q, -TRS_PIO2_HI, a);
    r = fmaf(q, -TRS_PIO2_LO, r);
    const float zz = r * r;
    const float ps = fmaf(fmaf(TRS_S0, zz, TRS_S1), zz, TRS_S2);
    const float s = fmaf(r * zz, ps, r);
    const float pc = fmaf(fmaf(TRS_C0, zz, TRS_C1), zz, TRS_C2);
    const float c = fmaf(zz * zz, pc, fmaf(zz, -0.5f, 1.0f));
    const unsigned n = (unsigned)((int)q) & 3u;          // quadrant n: (so, co) = (s, c), (c, -s), (-s, -c), (-c, s)
    const bool odd = (n & 1u) != 0u;
    const float ss = odd ? c : s, cc = odd ? s : c;
    so = __uint_as_float(__float_as_uint(ss) ^ ((n & 2u) << 30));
    co = __uint_as_float(__float_as_uint(cc) ^ (((n + 1u) & 2u) << 30));
}

__device__ __forceinline__ float clampf(float a, float lo, float hi) { return a < lo ? lo : (a > hi ? hi : a); }

// wave64 argmin over (distance, index): smaller distance wins, equal distance -> lower index.  The wave's result ends in lane 63.
// Two reductions through the VALU's DPP path (row_shr 1, 2, 4, 8, then row_bcast15 / row_bcast31; lanes without a source keep their own value):
// the minimum distance first (two dword moves + one v_min_f64 per step; distances are finite and >= 0), then — among the lanes whose own distance IS
// that minimum — the lowest index (one v_min_u32 per step).  Until round 5 one reduction carried the pair: a 64-bit compare, an equality test, an
// index compare and three selects per step, which hipcc laid out as six branchy blocks of ~20 instructions each — a quarter of a physics step's
// instructions on a chain that cannot overlap with anything (the same lexicographic minimum: results are bit-identical).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void min_f64_dpp_step(double& d)
{
    const int lo = __double2loint(d), hi = __double2hiint(d);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double od = __hiloint2double(ohi, olo);
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(od), "v"(d));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void min_u32_dpp_step(unsigned& i)
{
    const unsigned oi = (unsigned)__builtin_amdgcn_update_dpp((int)i, (int)i, CTRL, ROW_MASK, 0xf, false);
    i = oi < i ? oi : i;
}
__device__ __forceinline__ void wave_argmin(double& d, int& i)
{
    const double own = d;
    min_f64_dpp_step<0x111, 0xf>(d);     // row_shr:1
    min_f64_dpp_step<0x112, 0xf>(d);     // row_shr:2
    min_f64_dpp_step<0x114, 0xf>(d);     // row_shr:4
    min_f64_dpp_step<0x118, 0xf>(d);     // row_shr:8   -> lane 15 of each row holds the row's minimum
    min_f64_dpp_step<0x142, 0xa>(d);     // row_bcast:15 into rows 1 and 3
    min_f64_dpp_step<0x143, 0xc>(d);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
    const double dmin = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), 63), __builtin_amdgcn_readlane(__double2loint(d), 63));
    unsigned k = own == dmin ? (unsigned)i : 0xFFFFFFFFu;    // (indices are < 2^16)
    min_u32_dpp_step<0x111, 0xf>(k);
    min_u32_dpp_step<0x112, 0xf>(k);
    min_u32_dpp_step<0x114, 0xf>(k);
    min_u32_dpp_step<0x118, 0xf>(k);
    min_u32_dpp_step<0x142, 0xa>(k);
    min_u32_dpp_step<0x143, 0xc>(k);
    i = (int)k;
}

// wave64 sum of an unsigned per lane by the same DPP steps (lanes without a source add 0); the total ends in lane 63
__device__ __forceinline__ unsigned wave_sum_dpp(unsigned x)
{
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}

__device__ __forceinline__ unsigned cvt_u32_sat(float x)
{
    // v_cvt_u32_f32: truncate toward zero, saturate (negative / NaN -> 0).  For the rasteriser's
    // clamp(floor(g), 0, G-1) this equals min(cvt_u32_sat(g), G-1): g < 0 -> 0, g >= 0 -> trunc == floor.
    unsigned r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ void synth_controls(unsigned long long seed, uint32_t gid, uint32_t step, float& sf, float& steer, float& thr)
{
    unsigned long long z = seed + (((unsigned long long)gid << 32) | (unsigned long long)step) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float us = (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;
    const float ut = (float)(uint32_t)((z >> 16) & 0xFFFFFFu) * 5.9604644775390625e-08f;
    const float raw = us * 2.0f - 1.0f;
    sf = sf + TRS_SYNTH_ALPHA * (raw - sf);
    steer = sf;
    thr = TRS_SYNTH_THR_LO + TRS_SYNTH_THR_SPAN * ut;
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));   // 16-B register tuple (HIP's uint4 struct defeats SROA here)
typedef __attribute__((address_space(3))) const uint32_t* lds_u32p;
typedef unsigned u3v __attribute__((ext_vector_type(3)));

#ifndef TRS_STORE_AUX
#define TRS_STORE_AUX 17   /* cache policy of the image stores: 0 plain, 2 nt, 16 sc1, 17 sc0 sc1 (write-through: the frame streams to HBM while the kernel computes instead of being flushed from L2 at kernel end; +13% at 1024 envs, profiles/r01_store_policy_ab.txt) */
#endif

// one wave advances one env (all lanes compute the same scalars; the track scan is lane-parallel)
template <typename T>
__device__ __forceinline__ T coherent_load(const T* ptr)
{   // vector load that bypasses the per-CU L1 and the scalar cache: inside a K-step launch the value may have been
    // stored by lane 0 of this wave one step earlier
    return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Nearest raw track point of (qx, qy, qz) by one wave: binary64 L1, strict '<' with the lowest index winning ties
// (reference LocationTracker.__find_closest, components/track_data_process.py:89-104).  First the 3x3 block of 4-unit
// cells around the query (typically ~100 of the 1185 points); a block result below one cell size is provably the global
// one (include/trsim_spec.h R3), otherwise every point is scanned.  All lanes get the result.
struct NearParams { int np, off_py, off_pz, off_gstart, off_gpts, nx, nz; double x0, z0; };

template <bool SEL = false>                              // SEL: selects instead of branches (the physics-only kernels, see spec_sincos_sel)
__device__ __forceinline__ void wave_nearest(const NearParams& g, const unsigned char* lphys, double qx, double qy, double qz, int lane,
                                             double& best_out, int& idx_out)
{
    const double* lpx = reinterpret_cast<const double*>(lphys);
    const double* lpy = reinterpret_cast<const double*>(lphys + g.off_py);
    const double* lpz = reinterpret_cast<const double*>(lphys + g.off_pz);
    if (g.nx > 0) {
        const double fx = floor((qx - g.x0) * (1.0 / TRS_NEAR_GRID_CELL)), fz = floor((qz - g.z0) * (1.0 / TRS_NEAR_GRID_CELL));
        if (fx >= -1.0 && fx <= (double)g.nx && fz >= -1.0 && fz <= (double)g.nz) {
            const unsigned short* gstart = reinterpret_cast<const unsigned short*>(lphys + g.off_gstart);
            const unsigned short* gpts = reinterpret_cast<const unsigned short*>(lphys + g.off_gpts);
            const int cx = (int)fx, cz = (int)fz;
            const int x_lo = max(cx - 1, 0), x_hi = min(cx + 1, g.nx - 1);
            double best = TRS_LOST_L1;
            int bi = 0;
            if (x_lo <= x_hi) {
                // The candidates of the (up to) three rows as ONE list: the six bucket bounds are requested together, then a lane takes candidates
                // lane, lane + 64, ... of the concatenation — three LDS round trips (bounds, point index, coordinates) where a loop per row made nine.
                // The minimum over (distance, index) does not depend on the order the candidates are seen in.
                int lo[3], cnt[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int rz = cz - 1 + k;
                    const bool in = rz >= 0 && rz <= g.nz - 1;
                    const int rc = in ? rz : 0;
                    const int a = gstart[rc * g.nx + x_lo], b = gstart[rc * g.nx + x_hi + 1];       // the three cells of a row are contiguous
                    lo[k] = a; cnt[k] = in ? b - a : 0;
                }
                const int c01 = cnt[0] + cnt[1], total = c01 + cnt[2];
                for (int k = lane; k < total; k += 64) {
                    const int i = k < cnt[0] ? lo[0] + k : (k < c01 ? lo[1] + (k - cnt[0]) : lo[2] + (k - c01));
                    const int idx = gpts[i];
                    const double d = (fabs(qx - lpx[idx]) + fabs(qy - lpy[idx])) + fabs(qz - lpz[idx]);
                    if constexpr (SEL) {
                        const bool take = d < best || (d == best && idx < bi);
                        best = take ? d : best; bi = take ? idx : bi;
                    } else {
                        if (d < best || (d == best && idx < bi)) { best = d; bi = idx; }
                    }
                }
            }
            wave_argmin(best, bi);
            const int idx = __builtin_amdgcn_readlane(bi, 63);
            const double bd = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(best), 63), __builtin_amdgcn_readlane(__double2loint(best), 63));
            if (bd < TRS_NEAR_GRID_CELL) { best_out = bd; idx_out = idx; return; }
        }
    }
    double best = TRS_LOST_L1;
    int bi = 0;
    int i = lane;
    for (; i + 64 < g.np; i += 128) {                       // two points per trip: the LDS reads of one overlap the arithmetic of the other
        const int i2 = i + 64;
        const double ax = lpx[i], ay = lpy[i], az = lpz[i];
        const double bx = lpx[i2], by = lpy[i2], bz = lpz[i2];
        const double d1 = (fabs(qx - ax) + fabs(qy - ay)) + fabs(qz - az);
        const double d2 = (fabs(qx - bx) + fabs(qy - by)) + fabs(qz - bz);
        if (d1 < best) { best = d1; bi = i; }
        if (d2 < best) { best = d2; bi = i2; }
    }
    if (i < g.np) {
        const double d1 = (fabs(qx - lpx[i]) + fabs(qy - lpy[i])) + fabs(qz - lpz[i]);
        if (d1 < best) { best = d1; bi = i; }
    }
    wave_argmin(best, bi);
    idx_out = __builtin_amdgcn_readlane(bi, 63);
    best_out = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(best), 63), __builtin_amdgcn_readlane(__double2loint(best), 63));
}

__device__ __forceinline__ NearParams near_of(const PParams& p)
{
    return NearParams{p.np, p.off_py, p.off_pz, p.off_gstart, p.off_gpts, p.grid_nx, p.grid_nz, p.grid_x0, p.grid_z0};
}

// Per-env state held in registers by the wave that owns the env (every lane holds the same values).
struct EnvRegs {
    float x, y, z, yaw, v, sf, epr, speed, cte;
    int seg, epl, done, pend;
};

__device__ __forceinline__ void env_load(const PParams& p, int e, EnvRegs& s)
{
    s.pend = coherent_load(&p.pending[e]); s.done = coherent_load(&p.done[e]);
    s.sf = coherent_load(&p.steer_filt[e]);
    s.x = coherent_load(&p.x[e]); s.y = coherent_load(&p.y[e]); s.z = coherent_load(&p.z[e]);
    s.yaw = coherent_load(&p.yaw[e]); s.v = coherent_load(&p.v[e]);
    s.seg = coherent_load(&p.seg_idx[e]);
    s.epr = coherent_load(&p.ep_return[e]);
    s.epl = coherent_load(&p.ep_len[e]);
    s.speed = 0.f; s.cte = 0.f;
}

__device__ __forceinline__ void env_store(const PParams& p, int e, const EnvRegs& s, int lane)
{
    if (lane == 0) {
        p.x[e] = s.x; p.y[e] = s.y; p.z[e] = s.z; p.yaw[e] = s.yaw; p.v[e] = s.v;
        p.speed[e] = s.speed; p.cte[e] = s.cte; p.seg_idx[e] = s.seg; p.done[e] = (uint8_t)s.done;
        p.ep_return[e] = s.epr; p.ep_len[e] = s.epl; p.steer_filt[e] = s.sf; p.pending[e] = (uint8_t)s.pend;
    }
}
// What one env step hands on besides the new state: the camera parameters of the new pose (map-cell units) and the flags.
struct StepOut { float4 cam; float pitch; int is_done, do_reset; };   // pitch: the frame's view pitch on a track with elevation (0 on a flat one)

template <bool WT, typename T>
__device__ __forceinline__ void store_out(T* ptr, T v)
{   // WT: write-through at system scope (sc0 sc1) — the resident worker's outputs are read by other agents while the kernel
    // is still running, so nothing may sit dirty in this XCD's L2
    if constexpr (WT) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *ptr = v;
}

// One wave advances one env by one step on registers (include/trsim_spec.h, "one env step"): every lane computes the same
// scalars, the track scan is lane-parallel.  Controls are this step's (already fetched; generated here when `synth`); `rin` =
// the user's reset request.  Side effects: last_return[e] on a reset, nothing else.
// STORE_LR: store last_return[e] here on a reset (the resident worker stores it itself, with its other outputs).
template <bool WT, bool STORE_LR = true, bool SEL = false>
__device__ __forceinline__ void env_advance(const PParams& p, const unsigned char* lphys, int e, EnvRegs& s, uint32_t t, int synth,
                                            float steer, float thr, float brk, uint8_t rin, int lane, StepOut& o, const trsim::HillBlock* hill = nullptr)
{   // hill: a track with elevation (the HILLS instantiations of the step kernels pass it; nullptr = flat: a compile-time constant at every other call site)
    const double* lpx = reinterpret_cast<const double*>(lphys);
    const double* lpy = reinterpret_cast<const double*>(lphys + p.off_py);
    const double* lpz = reinterpret_cast<const double*>(lphys + p.off_pz);
    const float2* ltan = reinterpret_cast<const float2*>(lphys + p.off_tan);
    const int gid = p.env_id_base + e;
    float sf = s.sf;
    const int prev_idx = s.seg;
    float epr = s.epr;
    int epl = s.epl;
    const int do_reset = (s.pend != 0) || (rin != 0) || (p.auto_reset && s.done != 0);
    float x1, y0, z1, yaw1, v2, hs, hc;
    if (do_reset) {
        const int si = (int)(((long long)TRS_START_STRIDE * gid) % p.np);
        x1 = (float)lpx[si]; y0 = (float)lpy[si]; z1 = (float)lpz[si];
        yaw1 = p.start_yaw[si]; v2 = 0.0f; sf = 0.0f;
        if constexpr (SEL) spec_sincos_sel(yaw1, hs, hc); else spec_sincos(yaw1, hs, hc);
    } else {
        if (synth) synth_controls(p.seed, (uint32_t)gid, t, sf, steer, thr);
        steer = clampf(steer, -1.0f, 1.0f);
        thr = clampf(thr, -1.0f, 1.0f);
        brk = clampf(brk, 0.0f, 1.0f);
        float sd, cd;
        if constexpr (SEL) spec_sincos_sel(steer * p.max_steer, sd, cd); else spec_sincos(steer * p.max_steer, sd, cd);
        const float tan_d = sd / cd;
        const float a = thr * p.accel_max - p.drag_lin * s.v;
        const float v1 = s.v + a * p.dt;
        const float dv = (p.roll_res + brk * p.brake_max) * p.dt;
        if constexpr (SEL) {                                 // (rolling resistance and brake take speed away but never reverse it)
            const float tp = v1 - dv, tn = v1 + dv;
            const float vp = tp < 0.0f ? 0.0f : tp, vn = tn > 0.0f ? 0.0f : tn;
            v2 = v1 > 0.0f ? vp : (v1 < 0.0f ? vn : 0.0f);
        } else {
            if (v1 > 0.0f) { v2 = v1 - dv; if (v2 < 0.0f) v2 = 0.0f; }
            else if (v1 < 0.0f) { v2 = v1 + dv; if (v2 > 0.0f) v2 = 0.0f; }
            else v2 = 0.0f;
        }
        v2 = clampf(v2, -p.v_rev_max, p.v_max);
        yaw1 = s.yaw + ((v2 * tan_d) * p.inv_wheelbase) * p.dt;
        if constexpr (SEL) {
            yaw1 = yaw1 > TRS_PI ? yaw1 - TRS_TWO_PI : yaw1;
            yaw1 = yaw1 < -TRS_PI ? yaw1 + TRS_TWO_PI : yaw1;
            spec_sincos_sel(yaw1, hs, hc);
        } else {
            if (yaw1 > TRS_PI) yaw1 -= TRS_TWO_PI;
            if (yaw1 < -TRS_PI) yaw1 += TRS_TWO_PI;
            spec_sincos(yaw1, hs, hc);
        }
        x1 = s.x + (v2 * hs) * p.dt;
        z1 = s.z + (v2 * hc) * p.dt;
        y0 = s.y;
    }
    double bestd;
    int idx;
    wave_nearest<SEL>(near_of(p), lphys, (double)x1, (double)y0, (double)z1, lane, bestd, idx);

    const float y1 = (float)lpy[idx];
    const float2 tg = p.tan_in_lds ? ltan[idx] : reinterpret_cast<const float2*>(p.tangent_g)[idx];
    const float cte = (x1 - (float)lpx[idx]) * tg.y - (z1 - (float)lpz[idx]) * tg.x;
    const bool lost = bestd >= TRS_LOST_L1;
    const int is_done = (fabsf(cte) > p.offtrack_cte) || lost;
    if (do_reset) {
        if constexpr (STORE_LR) { if (lane == 0) store_out<WT>(&p.last_return[e], epr); }
        epr = 0.0f; epl = 0;
    } else {
        int d = idx - prev_idx;
        const int half = p.np / 2;
        if (d >= p.np - half) d -= p.np;
        if (d < -half) d += p.np;
        const float reward = (float)d - (is_done ? p.offtrack_penalty : 0.0f);
        epr = epr + reward;
        epl += 1;
    }
    s.x = x1; s.y = y1; s.z = z1; s.yaw = yaw1; s.v = v2; s.sf = sf; s.epr = epr; s.epl = epl;
    s.speed = fabsf(v2); s.cte = cte; s.seg = idx; s.done = is_done; s.pend = 0;
    const float camx = ((x1 + p.cam_fwd * hs) - p.map_x0f) * p.inv_cellf;
    const float camz = ((z1 + p.cam_fwd * hc) - p.map_z0f) * p.inv_cellf;
    o.cam = make_float4(camx, camz, hs, hc);
    o.pitch = hill ? hill->vpitch[idx] : 0.0f;              // (every lane reads the same word: one broadcast load, requested here, needed at the hand-off)
    o.is_done = is_done; o.do_reset = do_reset;
}

// ... inside the step kernels: controls from the launch's arrays, camera parameters to the global ring (the next launch's
// first frame) and to this launch's LDS ring, then the progress counter the raster team waits on.
template <bool SEL = false>
__device__ __forceinline__ void env_step(const PParams& p, const unsigned char* lphys, int e, EnvRegs& s, uint32_t t, int k,
                                         float4* cam_out, float4* lcam_slot, int* pprog_j, int lane, const trsim::HillBlock* hill = nullptr, float* lpitch_slot = nullptr)
{
    const uint8_t rin = (!p.synth && p.ctl_reset && k == 0) ? p.ctl_reset[e] : (uint8_t)0;
    float steer = 0.f, thr = 0.f, brk = 0.f;
    if (!p.synth) {
        const size_t ci = (size_t)k * (size_t)p.ctl_stride + (size_t)e;
        steer = p.ctl_steer[ci]; thr = p.ctl_thr[ci]; brk = p.ctl_brk ? p.ctl_brk[ci] : 0.0f;
    }
    StepOut o;
    env_advance<false, true, SEL>(p, lphys, e, s, t, p.synth, steer, thr, brk, rin, lane, o, hill);
    if (lane == 0) {
        if (p.write_cam) cam_out[e] = o.cam;                // for the next launch (its first frame)
        *lcam_slot = o.cam;                                 // for this launch's raster team
        if (hill) {                                         // a track with elevation: the frame's view pitch rides along (global ring for the next launch, LDS ring for this one)
            if (p.write_cam) hill->cam_pitch[(size_t)(t & (kRing - 1)) * p.n_envs + e] = o.pitch;
            if (lpitch_slot) *lpitch_slot = o.pitch;
        }
        if (o.is_done) atomicAdd(&p.stats[0], 1ull);
        if (o.do_reset) atomicAdd(&p.stats[1], 1ull);
        __hip_atomic_store(pprog_j, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // publish: step k of this env is done
    }
}

// (dx, dz) of a row as two scalar multiplies.  The packed form (v_pk_mul_f32 with the row-table register pair as source AND
// destination) returned +-0 for dx in lanes 48..63 of one row pass now and then, but only while workgroups of other kernels
// shared the CU (scripts/det_probe*.py: 1-3 % of the steps beside a pilot loop on another stream; never alone).  Same
// IEEE products, so results are unchanged.
__device__ __forceinline__ f2v ray_step(f2v kk2, f2v cns)
{
#ifdef TRS_PACKED_RAY_STEP
    return kk2 * cns;
#else
    float dx, dz;
    asm("v_mul_f32 %0, %1, %2" : "=v"(dx) : "v"(kk2.x), "v"(cns.x));
    asm("v_mul_f32 %0, %1, %2" : "=v"(dz) : "v"(kk2.y), "v"(cns.y));
    return f2v{dx, dz};
#endif
}

// ---- the raster walk of one thread ----------------------------------------------------------------------------------
// A thread owns one 4-pixel column group (u0 fixed) and walks image rows `rows_per_pass` apart, so the pixel-centre offsets
// are loop constants.  The first `uni_rows` rows (sky, ground beyond the far plane) have four equal class colours and need
// neither the class map nor the camera pose.  Per pixel of the other rows: 1 packed fma (gx, gz), 2 saturating converts + 2
// min (= floor + clamp), 3 address ops, 1 LDS map read (the map lives at LDS offset 0), shift + bit-field extract, 1 palette
// address op, 1 LDS palette read; 4 pixels -> 12 bytes with v_perm_b32 and ONE buffer_store_dwordx3 per lane (768 contiguous
// bytes = 6 full 128-B lines per wave instruction), write-through.
struct RasterThread {
    const f2v* lrow;
    const float* lrowdepth;
    unsigned pal_off;                   // LDS byte offset of the palette uint32[H][4] the rows are shaded from (the host's, or a frame's own on a track with elevation)
    f2v ufa, ufb, ufc, ufd;
    unsigned gwm1, ghm1, pitch;
    int cg, vstart, vground, col_off, row_bytes;
};

__device__ __forceinline__ RasterThread raster_thread(const RParams& p, const unsigned char* lds, int tid)
{
    RasterThread t;
    t.lrow = reinterpret_cast<const f2v*>(lds + p.off_rowtab);
    t.lrowdepth = reinterpret_cast<const float*>(lds + p.off_depth);
    t.pal_off = (unsigned)p.off_pal;
    const float half_w = (float)(p.W / 2);
    t.gwm1 = (unsigned)(p.map_w - 1); t.ghm1 = (unsigned)(p.map_h - 1);
    t.cg = tid % p.gpr;
    const int r0 = tid / p.gpr;                          // threads with r0 >= rows_per_pass idle (32 of 512 at W = 160: 12 x 40 = 480)
    const float uf0 = (float)(t.cg << 2) + 0.5f - half_w;
    t.ufa = f2v{uf0, uf0}; t.ufb = f2v{uf0 + 1.0f, uf0 + 1.0f}; t.ufc = f2v{uf0 + 2.0f, uf0 + 2.0f}; t.ufd = f2v{uf0 + 3.0f, uf0 + 3.0f};
    t.pitch = (unsigned)p.map_pitch_b;
    t.vstart = r0 < p.rows_per_pass ? r0 : p.H;
    t.vground = t.vstart;                                // this thread's first row that needs the map
    while (t.vground < p.uni_rows) t.vground += p.rows_per_pass;
    t.row_bytes = p.gpr * 12;
    t.col_off = t.cg * 12;
    return t;
}

// the frame (and z-depth frame) of env e as buffer descriptors: stores carry the cache-policy bits TRS_STORE_AUX
struct FrameDesc { __amdgpu_buffer_rsrc_t rgb, dep; };

template <bool DEPTH>
__device__ __forceinline__ FrameDesc frame_desc(const RParams& p, uint8_t* img, float* dep, int e)
{
    FrameDesc f;
    f.rgb = __builtin_amdgcn_make_buffer_rsrc(img + (size_t)e * ((size_t)p.gpe * 12), 0, (int)((size_t)p.gpe * 12), 0x00020000);
    f.dep = f.rgb;
    if constexpr (DEPTH) f.dep = __builtin_amdgcn_make_buffer_rsrc(dep + (size_t)e * ((size_t)p.gpe * 4), 0, (int)((size_t)p.gpe * 16), 0x00020000);
    return f;
}

// rows with four equal class colours: one palette read per 4 pixels, no map lookup, no pose
template <bool DEPTH>
__device__ __forceinline__ void raster_uniform_rows(const RParams& p, const RasterThread& t, const FrameDesc& f, int uni_rows = -1)   // uni_rows: the frame's own count (tracks with elevation)
{
    const int nu = uni_rows < 0 ? p.uni_rows : uni_rows;
    for (int v = t.vstart; v < nu; v += p.rows_per_pass) {
        const uint32_t c = *(lds_u32p)(uintptr_t)(t.pal_off + ((unsigned)v << 4));
        const u3v px3 = {__builtin_amdgcn_perm(c, c, 0x04020100u), __builtin_amdgcn_perm(c, c, 0x05040201u), __builtin_amdgcn_perm(c, c, 0x06050402u)};
        __builtin_amdgcn_raw_buffer_store_b96(px3, f.rgb, t.col_off + v * t.row_bytes, 0, TRS_STORE_AUX);
        if constexpr (DEPTH) {
            const unsigned dz = __float_as_uint(t.lrowdepth[v]);
            const u4v d4 = {dz, dz, dz, dz};
            __builtin_amdgcn_raw_buffer_store_b128(d4, f.dep, (t.cg + v * p.gpr) * 16, 0, TRS_STORE_AUX);
        }
    }
}

// rows that see the track, from the camera parameters (camx, camz, sin, cos) of the env's pose
// UNI_CHECK (tracks with elevation: which rows are sky or beyond the far plane depends on the frame): a row whose table entry has row_k == 0 — exactly the SKY and
// FAR rows — takes its one colour without the four map lookups.
template <bool DEPTH, bool UNI_CHECK = false>
__device__ __forceinline__ void raster_ground_rows(const RParams& p, const RasterThread& t, const FrameDesc& f, const float4 cam)
{
    const f2v sc = {cam.z, cam.w}, cns = {cam.w, -cam.z}, camxz = {cam.x, cam.y};
    f2v rt = t.lrow[t.vground < p.H ? t.vground : 0];
    for (int v = t.vground; v < p.H; v += p.rows_per_pass) {
        const int vn = v + p.rows_per_pass;
        const f2v rtn = t.lrow[vn < p.H ? vn : v];                          // prefetch the next row's table entry
        const unsigned pal_a = t.pal_off + ((unsigned)v << 4);
        const f2v lz2 = {rt.x, rt.x}, kk2 = {rt.y, rt.y};
        f2v a;                                                             // (ax, az)
        if constexpr (UNI_CHECK) {
            // (the HILLS instantiations: hipcc picked the in-place cross-half packed form here — v_pk_fma_f32 v[10:11], v[10:11], .. op_sel_hi:[0,1,1], the one
            // tests/test_build_lint.py screens for (DESIGN.md "Packed FP32") — so the two fused multiply-adds are written apart: the same IEEE results)
            a.x = __builtin_fmaf(rt.x, sc.x, camxz.x);
            a.y = __builtin_fmaf(rt.x, sc.y, camxz.y);
        } else {
            a = __builtin_elementwise_fma(lz2, sc, camxz);
        }
        const f2v d = ray_step(kk2, cns);                                  // (dx, dz) = (k*c, -(k*s))
        auto shade = [&](f2v uf) -> uint32_t {
            const f2v g = __builtin_elementwise_fma(uf, d, a);             // (gx, gz)
            const unsigned ix = min(cvt_u32_sat(g.x), t.gwm1);
            const unsigned iz = min(cvt_u32_sat(g.y), t.ghm1);
            const unsigned xoff = (ix >> 2) & ~3u;                          // byte offset of the map word in its row
            unsigned waddr, paddr;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(waddr) : "v"(iz), "s"(t.pitch), "v"(xoff));
            const uint32_t w = *(lds_u32p)(uintptr_t)waddr;                 // map lives at LDS offset 0 (checked by the kernels)
            const uint32_t cls = __builtin_amdgcn_ubfe(w, ix << 1, 2);      // offset uses bits [4:0] = 2*(ix&15)
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(paddr) : "v"(cls), "v"(pal_a));
            return *(lds_u32p)(uintptr_t)paddr;
        };
#if TRS_ABLATE == 2   /* diagnostic build: stores only */
        const uint32_t c0p = (uint32_t)v, c1p = c0p + 1, c2p = c0p + 2, c3p = c0p + 3; (void)shade;
#else
        uint32_t c0p, c1p, c2p, c3p;
        if (UNI_CHECK && rt.y == 0.0f) c0p = c1p = c2p = c3p = *(lds_u32p)(uintptr_t)pal_a;
        else { c0p = shade(t.ufa); c1p = shade(t.ufb); c2p = shade(t.ufc); c3p = shade(t.ufd); }
#endif
        // 4 x 0x00BBGGRR -> 12 bytes R,G,B,R,G,B,...  (v_perm_b32: selector bytes 0-3 = 2nd operand, 4-7 = 1st)
        const uint32_t w0 = __builtin_amdgcn_perm(c1p, c0p, 0x04020100u);
        const uint32_t w1 = __builtin_amdgcn_perm(c2p, c1p, 0x05040201u);
        const uint32_t w2 = __builtin_amdgcn_perm(c3p, c2p, 0x06050402u);
#if TRS_ABLATE == 1   /* diagnostic build: compute, no stores */
        asm volatile("" :: "v"(w0), "v"(w1), "v"(w2)); (void)f;
#else
        const u3v px3 = {w0, w1, w2};
        __builtin_amdgcn_raw_buffer_store_b96(px3, f.rgb, t.col_off + v * t.row_bytes, 0, TRS_STORE_AUX);
        if constexpr (DEPTH) {   // z-depth is constant along a row of a ground-plane camera: 4 pixels = one 16-B store
            const unsigned dz = __float_as_uint(t.lrowdepth[v]);
            const u4v d4 = {dz, dz, dz, dz};
            __builtin_amdgcn_raw_buffer_store_b128(d4, f.dep, (t.cg + v * p.gpr) * 16, 0, TRS_STORE_AUX);
        }
#endif
        rt = rtn;
    }
}

// ---- tracks with elevation (include/trsim_spec.h, "tracks with elevation") ------------------------------------------------------
// On a hilly track a frame's row tables depend on the env and the step: the view pitch P = pitch + dpitch[nearest track point] (the slope ahead against
// the slope here) comes from the physics wave with the camera parameters, and the raster team evaluates the rows into tables of its own in LDS —
// float2 rowtab[H] | uint32 palette[H][4] | float depth[H], the layout of the host's tables, so the row loops above only get another base
// (raster_use_table).  One thread computes one row: binary32, the spec's operation order (no contraction: the kernels are built with
// -ffp-contract=off; the division is IEEE).  The envs of a workgroup go through in BATCHES of hill_batch(H) (as many whole tables as the 512 raster
// threads fill in one pass, at most 4): team barrier (the previous batch's tables have been read by every wave) - build - team barrier - shade the
// batch's envs.  (First form: one table per env frame behind its own barrier, two tables alternating: 14.7 us per step at 1024 envs x 120x160 against
// 9.45 flat — the waves of the flat kernels never meet; profiles/r05_hills.txt.)
constexpr int kHillRowBytes = 28;
constexpr int kHillBatchMax = 4;
__host__ __device__ inline int hill_table_bytes(int H) { return (kHillRowBytes * H + 15) & ~15; }
__host__ __device__ inline int hill_batch(int H) { const int b = kRasterThreads / (H > 0 ? H : 1); return b < 1 ? 1 : (b > kHillBatchMax ? kHillBatchMax : b); }
__host__ __device__ inline int hill_lds_bytes(int H) { return hill_batch(H) * hill_table_bytes(H) + 48; }   // the batch's tables + the team-barrier counter (4 B) + 12 B spare + int first_ground[2][4]: an env's first row that sees the ground, by batch parity

// ImgPreprocessing.__process of ONE colour without dynamic brightness and Canny (img_preprocessing.py:37-74,92-99): the device twin of the host's filter_colour —
// the same binary32 trim, OpenCV's 8-bit fixed-point HSV with its reciprocal tables, the masks written over their destination channels in filter order.
__device__ __forceinline__ uint32_t hill_filter_colour(const trsim::HillBlock& hb, uint32_t bgr)
{
    int t[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float x = (float)((bgr >> (8 * ch)) & 255u);
        x = x - hb.f_offset;
        x = x * hb.f_contrast;
        x = x + hb.f_offset;
        x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
        t[ch] = (int)x;
    }
    int o0 = t[0], o1 = t[1], o2 = t[2];
    if (hb.f_color) {
        const int r = t[0], g = t[1], b = t[2];
        const int vv = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = vv - vmin;
        const int sat = (diff * hb.hsv_tab[vv] + (1 << 11)) >> 12;
        int h = (vv == r) ? (g - b) : ((vv == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
        h = (h * hb.hsv_tab[256 + diff] + (1 << 11)) >> 12;
        if (h < 0) h += 180;
        const int hh = min(h, 255), ss = min(sat, 255);
#pragma unroll
        for (int f = 0; f < 4; ++f) {                           // (unrolled: constant indices into the block's arrays — a run-time index would put them into scratch memory)
            if (f >= hb.f_nfilters) break;
            const int lo = (int)hb.f_lo[f], hi = (int)hb.f_hi[f];
            const bool in = hh >= (lo & 255) && hh <= (hi & 255) && ss >= ((lo >> 8) & 255) && ss <= ((hi >> 8) & 255) && vv >= ((lo >> 16) & 255) && vv <= ((hi >> 16) & 255);
            const int val = in ? 255 : 0, d = hb.f_dst[f];
            o0 = d == 0 ? val : o0; o1 = d == 1 ? val : o1; o2 = d == 2 ? val : o2;
        }
    }
    return (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16);
}

__device__ __forceinline__ float hill_row_build(const RParams& p, unsigned char* lds, unsigned tab_off, float P, int v)   // returns the row's row_k (0: sky or beyond the far plane)
{
    const trsim::HillBlock hb = *reinterpret_cast<const trsim::HillBlock*>(p.blob + trsim::hill_block_offset(p.blob_bytes));   // (uniform address: scalar loads)
    float sp, cp;
    spec_sincos(P, sp, cp);
    const float yn = (hb.hh - ((float)v + 0.5f)) * hb.inv_f;
    const float dy = yn * cp - sp, dz = yn * sp + cp;
    float lz = 0.0f, kk = 0.0f, dep = hb.z_far_f;
    uint32_t c0, c1, c2, c3;
    if (dy >= -1e-6f) {
        c0 = c1 = c2 = c3 = *(lds_u32p)(uintptr_t)((unsigned)hb.off_sky + ((unsigned)v << 2));
    } else {
        const float t = hb.cam_h_f / (-dy);
        const float zd = t * dz;
        if (zd > hb.z_far_f) {
            c0 = c1 = c2 = c3 = hb.far_rgb;
        } else {
            lz = zd * hb.inv_cell_f; kk = (t * hb.inv_f) * hb.inv_cell_f; dep = zd;
            const float fw = hb.fog_f * (zd * hb.inv_zfar_f), om = 1.0f - fw;
            constexpr int base[4][3] = {TRS_RGB_GRASS, TRS_RGB_ROAD, TRS_RGB_EDGE, TRS_RGB_CENTRE};
            constexpr int fog[3] = TRS_RGB_FOG;
            uint32_t col[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                uint32_t rgb = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float a = (float)base[c][ch] * om, b = (float)fog[ch] * fw;
                    const float sum = a + b;
                    rgb |= (uint32_t)(int)(sum + 0.5f) << (8 * ch);
                }
                col[c] = hb.filt ? hill_filter_colour(hb, rgb) : rgb;
            }
            c0 = col[0]; c1 = col[1]; c2 = col[2]; c3 = col[3];
        }
    }
    unsigned char* const tab = lds + tab_off;
    reinterpret_cast<f2v*>(tab)[v] = f2v{lz, kk};
    reinterpret_cast<u4v*>(tab + 8 * p.H)[v] = u4v{c0, c1, c2, c3};
    reinterpret_cast<float*>(tab + 24 * p.H)[v] = dep;
    return kk;
}

// the row loops read their tables from the frame's own table at tab_off
__device__ __forceinline__ RasterThread raster_use_table(const RasterThread& t, const unsigned char* lds, unsigned tab_off, int H)
{
    RasterThread r = t;
    r.lrow = reinterpret_cast<const f2v*>(lds + tab_off);
    r.pal_off = tab_off + 8u * (unsigned)H;
    r.lrowdepth = reinterpret_cast<const float*>(lds + tab_off + 24u * (unsigned)H);
    return r;
}

// (the raster team's barrier: a counter in LDS, bounded by the caller's `bail` — its comment is with raster_dyn_batch below)
template <typename Bail>
__device__ __forceinline__ bool team_barrier_wait(const int* dbar, int target, Bail& bail)
{
    for (unsigned spins = 0; __hip_atomic_load(dbar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u && bail(spins == 1023u)) return false;
    }
    return true;
}

// The tables of a batch of nb envs (view pitches P[0..nb)), between the batch's two team barriers.  Only the HILLS instantiations of the step kernels contain
// this: as a run-time branch inside the flat kernels' per-env loops the hilly path cost the FLAT-track step 3-6 % on the single-step paths, and as an out-of-line
// call — which gives the whole kernel a stack and the calling convention's register budget — a factor of 2.5 (same-box A/Bs in profiles/r05_hills.txt).
// `done_before` = team barriers this kernel has passed so far x raster waves; returns false when a barrier gave up (resident worker: abort / safety).
template <typename Bail>
__device__ __forceinline__ bool hill_batch_build(const RParams& p, unsigned char* lds, unsigned tab0, const float (&P)[kHillBatchMax], int nb, int* hbar, int done_before,
                                                 int tid, int lane, Bail& bail)
{
    constexpr int nrw = kRasterThreads / 64;
    // first_ground[parity of the batch][env]: the first row of the env's frame that sees the ground (row_k != 0).  The rows above it — sky, beyond the far plane — go
    // through the tight loop of uniform rows (raster_hill_frame), as the host's uni_rows lets the flat kernels do.  Two sets by batch parity: this batch's set is reset
    // here, in front of barrier 1, while a slow wave may still be reading the previous batch's.
    int* const fg = hbar + 4 + 4 * ((done_before / (2 * nrw)) & 1);
    if (tid < kHillBatchMax) fg[tid] = p.H;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of the previous batch's tables have returned (and the reset is in LDS)
    if (lane == 0) __hip_atomic_fetch_add(hbar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!team_barrier_wait(hbar, done_before + nrw, bail)) return false;
    const unsigned tb = (unsigned)hill_table_bytes(p.H);
    for (int row = tid; row < nb * p.H; row += kRasterThreads) {
        const int bi = row / p.H, v = row - bi * p.H;
        const float Pv = bi == 0 ? P[0] : (bi == 1 ? P[1] : (bi == 2 ? P[2] : P[3]));
        if (hill_row_build(p, lds, tab0 + (unsigned)bi * tb, Pv, v) != 0.0f) __hip_atomic_fetch_min(&fg[bi], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(hbar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return team_barrier_wait(hbar, done_before + 2 * nrw, bail);
}

// One env's frame from its table of the current batch (bi = its place in the batch, done_before as in hill_batch_build): the rows above its first ground row as
// uniform rows, the rest through the ground loop (which still takes a row with row_k == 0 as one colour, should one ever sit below a ground row).
template <bool DEPTH>
__device__ __forceinline__ void raster_hill_frame(const RParams& p, const RasterThread& rth, const unsigned char* lds, unsigned tab0, int bi, const int* hbar, int done_before,
                                                  const FrameDesc& f, const float4 cam)
{
    constexpr int nrw = kRasterThreads / 64;
    const int first_ground = hbar[4 + 4 * ((done_before / (2 * nrw)) & 1) + bi];
    RasterThread t = raster_use_table(rth, lds, tab0 + (unsigned)(bi * hill_table_bytes(p.H)), p.H);
    raster_uniform_rows<DEPTH>(p, t, f, first_ground);
    int vg = t.vstart;
    if (vg < first_ground) vg += ((first_ground - vg + p.rows_per_pass - 1) / p.rows_per_pass) * p.rows_per_pass;
    t.vground = vg;
    raster_ground_rows<DEPTH, true>(p, t, f, cam);
}

// The colour masks of one pixel (img_preprocessing.py:57-74; OpenCV's 8-bit RGB -> HSV with its fixed-point reciprocal tables, then inRange).
// P = the TRIMMED pixel, bytes (r, g, b, x).  rngb[c * 256 + x] (range_byte_entry) has byte ch = 0xFF when value x of component c (h, s, v) lies
// inside the range of the filter whose mask goes to channel ch, so the AND of three lookups is the pixel's masks in place; sel has 0xFF in the
// channels that carry a mask (a later filter on the same channel replaces an earlier one, :57-63), the others keep the trimmed value.
// (Round 3: one table lookup chain and one v_bfi per pixel instead of a bit test, a compare and a select per channel.)
// (mask_pixel_rgb: the same for a caller that holds the three trimmed components apart — trs_preprocess_kernel's table lookups — and would otherwise pack
// them only to have them unpacked here: 3 instructions per pixel of ~44)
__device__ __forceinline__ unsigned mask_pixel_rgb(int r, int g, int b, const int* tab, const unsigned* rngb, unsigned sel)
{
    const unsigned P = (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16);
    const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
    const int sat = (__mul24(diff, tab[v]) + (1 << 11)) >> 12;
    int h = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    h = (__mul24(h, tab[256 + diff]) + (1 << 11)) >> 12;
    h = h < 0 ? h + 180 : h;
    const unsigned m = rngb[min(h, 255)] & rngb[256 + min(sat, 255)] & rngb[512 + v];
    return (m & sel) | (P & ~sel);
}
__device__ __forceinline__ unsigned mask_pixel(unsigned P, const int* tab, const unsigned* rngb, unsigned sel)
{
    const int r = (int)(P & 255u), g = (int)((P >> 8) & 255u), b = (int)((P >> 16) & 255u);
    const int v = max(r, max(g, b)), vmin = min(r, min(g, b)), diff = v - vmin;
    const int sat = (__mul24(diff, tab[v]) + (1 << 11)) >> 12;             // 24-bit multiplies: full rate (diff <= 255, the reciprocals < 2^21; 32-bit integer multiplies are quarter rate)
    int h = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    h = (__mul24(h, tab[256 + diff]) + (1 << 11)) >> 12;
    h = h < 0 ? h + 180 : h;
    const unsigned m = rngb[min(h, 255)] & rngb[256 + min(sat, 255)] & rngb[512 + v];
    return (m & sel) | (P & ~sel);
}
// entry i = c * 256 + x of the table above; *sel_out = the channels that carry a mask
__host__ __device__ inline unsigned range_byte_entry(const unsigned (&lo)[4], const unsigned (&hi)[4], const int (&dst_ch)[4], int n_filters, int i, unsigned* sel_out)
{
    const int c = i >> 8, x = i & 255;
    int fsel[3] = {-1, -1, -1};
    for (int f = 0; f < n_filters; ++f) { const int dc = dst_ch[f]; if (dc >= 0 && dc <= 2) fsel[dc] = f; }
    unsigned word = 0, sel = 0;
    for (int ch = 0; ch < 3; ++ch) {
        if (fsel[ch] < 0) continue;
        const int l = (int)((lo[fsel[ch]] >> (8 * c)) & 255u), u = (int)((hi[fsel[ch]] >> (8 * c)) & 255u);
        sel |= 0xFFu << (8 * ch);
        if (x >= l && x <= u) word |= 0xFFu << (8 * ch);
    }
    if (sel_out) *sel_out = sel;
    return word;
}

// ImgPreprocessing.__process of ONE colour with this frame's brightness delta (img_preprocessing.py:37-74,92-99): the
// per-pixel arithmetic of trs_preprocess_kernel (OpenCV's fixed-point reciprocal tables are read from global memory)
__device__ __forceinline__ uint32_t filter_colour_dev(const FParams& f, const unsigned* ltabs, uint32_t bgr, float deltaf)
{
    unsigned P = 0u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float x = (float)((bgr >> (8 * ch)) & 255u);
        x = x + deltaf;
        x = x - f.offset;
        x = x * f.contrast;
        x = x + f.offset;
        x = x < 0.0f ? 0.0f : (x > 255.0f ? 255.0f : x);
        P |= (unsigned)(int)x << (8 * ch);
    }
    // the masks through the LDS tables (the arithmetic of trs_preprocess_kernel's mask_pixel: OpenCV's reciprocals, then three byte-mask lookups)
    if (f.color) P = mask_pixel(P, reinterpret_cast<const int*>(ltabs), ltabs + 512, ltabs[512 + 768]);
    return P;
}



// ---- dynamic brightness behind the rasteriser (trs_set_frame_filter with dynamic_brightness; img_preprocessing.py:88-99): the frame's
// own mean over rows [w0, w1) only needs the class of every pixel there, so, for up to kDynBatch envs at a time: (A) classify those
// rows once (classes kept in registers), sum the RAW colours per channel, reduce over the raster team; (B) every thread filters
// entries of the per-env palettes with each frame's delta; (C) shade all rows from those palettes.  No extra pass over HBM, each pixel
// classified once, two team barriers per batch (an LDS counter: the physics waves of the workgroup take no part).  `it` = batches
// this launch has done before this one (the same in every raster wave): the barrier counter and the parity of the sum slots follow it.
// LDS at f.lds_off: uint32 penv[kDynBatch][H][4] | int esum[2][kDynBatch][3] | int dbar — zeroed (esum, dbar) once per launch.
// Shared by trs_step_kernel<., true> and trs_worker_kernel<., true>: the same instructions, bit-identical frames.
#ifndef TRS_DYN_ABLATE
#define TRS_DYN_ABLATE 0   /* timing-only diagnostic builds (scripts/r04_dyn_ablate.sh), never shipped: 1 no classification in phase A, 2 no filter arithmetic in phase B, 3 no stores in phase C */
#endif
#ifdef TRS_DYN_STAMPS   /* diagnostic build (scripts/dyn_stamps.sh), never shipped: s_memtime counts (shader clocks on this part) of workgroup 7's first raster thread per phase into stats[46..50], batches seen into stats[52] */
#define DYN_STAMP(k) do { if (blockIdx.x == 7 && tid == 0) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); atomicAdd(&p.stats[46 + (k)], tn - dyn_t0); dyn_t0 = tn; } } while (0)
#else
#define DYN_STAMP(k) do { } while (0)
#endif
constexpr int kDynBatch = 4;
__host__ __device__ inline int dyn_lds_bytes(int H) { return kDynBatch * H * 16 + 128 + ((kDynTabWords * 4 + 15) & ~15) + H * 16; }   // ... + rowch[H]: the raw palette by channel
// the tables behind the batch's palettes and sums (staged by dyn_stage_tables before the launch's first barrier)
__device__ __forceinline__ unsigned* dyn_tabs_lds(unsigned char* lds_base, const FParams& f, int H) { return reinterpret_cast<unsigned*>(lds_base + f.lds_off + kDynBatch * H * 16 + 128); }
// rowch[v] = the RAW palette of row v by channel: {R of classes 0..3, G of classes 0..3, B of classes 0..3, 0} as packed bytes (behind the tables)
__device__ __forceinline__ unsigned dyn_rowch_lds(const FParams& f, int H) { return (unsigned)f.lds_off + (unsigned)(kDynBatch * H * 16 + 128 + ((kDynTabWords * 4 + 15) & ~15)); }
__device__ __forceinline__ void dyn_stage_tables(unsigned char* lds_base, const FParams& f, int H, int tid, int nthreads, const uint32_t* pal_global)
{
    unsigned* const dst = dyn_tabs_lds(lds_base, f, H);
    for (int i = tid; i < kDynTabWords; i += nthreads) dst[i] = f.tabs[i];
    uint32_t* const rowch = reinterpret_cast<uint32_t*>(lds_base + dyn_rowch_lds(f, H));
    for (int v = tid; v < H; v += nthreads) {                              // (the palette itself is still on its way into LDS: read it from memory)
        const uint32_t c0 = pal_global[4 * v], c1 = pal_global[4 * v + 1], c2 = pal_global[4 * v + 2], c3 = pal_global[4 * v + 3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            rowch[4 * v + ch] = ((c0 >> (8 * ch)) & 255u) | (((c1 >> (8 * ch)) & 255u) << 8) | (((c2 >> (8 * ch)) & 255u) << 16) | (((c3 >> (8 * ch)) & 255u) << 24);
        rowch[4 * v + 3] = 0u;
    }
}

// The team barrier of raster_dyn_batch: a BOUNDED spin.  Every 1024 polls (~30 us) it asks `bail(first)` whether to give up — the resident worker
// answers from its abort bit and its safety deadline (a wave of the team that left at an aborted wait_posted never arrives here: ADVICE r03); a
// launched kernel has no abort and passes a callable that says no.  false = gave up: the caller leaves the kernel.

template <bool DEPTH, typename Bail>
__device__ __forceinline__ bool raster_dyn_batch(const RParams& p, const FParams& f, const RasterThread& rth, unsigned char* lds_base, const float4 (&cams)[kDynBatch],
                                                 int nb, uint8_t* img, float* dep, int e, int it, int tid, int lane, Bail&& bail)
{
    uint32_t* const penv = reinterpret_cast<uint32_t*>(lds_base + f.lds_off);                          // [kDynBatch][H][4]
#ifdef TRS_DYN_STAMPS
    unsigned long long dyn_t0 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && tid == 0) atomicAdd(&p.stats[52], 1ull);
#endif
    int* const esum = reinterpret_cast<int*>(lds_base + f.lds_off + kDynBatch * p.H * 16);           // [2][kDynBatch][3]
    int* const dbar = esum + 2 * kDynBatch * 3;
    const int par = it & 1;
    const int nrw = kRasterThreads / 64;
    const f2v* const lrow = rth.lrow;
    unsigned cbits[kDynBatch][4];
#pragma unroll
    for (int bi = 0; bi < kDynBatch; ++bi)
#pragma unroll
        for (int k = 0; k < 4; ++k) cbits[bi][k] = 0u;
    // The classes of this thread's 4 pixels of a row (2 bits each) for TWO envs, in stages: eight map addresses, eight reads in flight, eight extractions.  The scheduling fences keep
    // the stages apart: left alone (and short of registers in this kernel) hipcc issued one read, waited for it, issued the next - sixteen
    // dependent LDS round trips per row of the batch where this takes two.
    auto classify_pair = [&](const f2v rt, const float4& cam0, const float4& cam1, unsigned& pack0, unsigned& pack1) {
        const f2v lz2 = {rt.x, rt.x}, kk2 = {rt.y, rt.y};
        const f2v uf[4] = {rth.ufa, rth.ufb, rth.ufc, rth.ufd};
        unsigned wa[8], sh[8];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const float4& cam = e2 ? cam1 : cam0;
            const f2v sc = {cam.z, cam.w}, cns = {cam.w, -cam.z}, camxz = {cam.x, cam.y};
            const f2v a = __builtin_elementwise_fma(lz2, sc, camxz);
            const f2v d = ray_step(kk2, cns);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f2v g = __builtin_elementwise_fma(uf[q], d, a);
                const unsigned ix = min(cvt_u32_sat(g.x), rth.gwm1);
                const unsigned iz = min(cvt_u32_sat(g.y), rth.ghm1);
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(wa[4 * e2 + q]) : "v"(iz), "s"(rth.pitch), "v"((ix >> 2) & ~3u));
                sh[4 * e2 + q] = ix << 1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = *(lds_u32p)(uintptr_t)wa[i];
        __builtin_amdgcn_sched_barrier(0);
        unsigned c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_ubfe(w[i], sh[i], 2);
        pack0 = c[0] | (c[1] << 2) | (c[2] << 4) | (c[3] << 6);
        pack1 = c[4] | (c[5] << 2) | (c[6] << 4) | (c[7] << 6);
    };
    // (A) rows outside, envs inside: the (up to) four envs' lookups of one row are independent chains (row table -> map
    // -> palette are three dependent LDS round trips per row, and two waves per SIMD cannot hide them one env at a time)
    // The channel sums of a row: sum over the classes k of n_k x (raw colour of class k in this row).  The pack of four 2-bit classes indexes a
    // 256-entry table of packed counts (n0 .. n3 as bytes), the row's raw palette sits in LDS by channel (rowch: four R bytes, four G bytes, four
    // B bytes — one 16-byte read per row, shared by the batch's envs and independent of the classification), and one v_dot4_u32_u8 per
    // channel adds n . colours.  (Until round 4: four dependent palette gathers and four masked adds per env and row — 28 of the ~64 vector
    // instructions of a row of phase A, which is bound by exactly those.)  Exact integers either way.
    float4 camx[kDynBatch];                                                  // an env past the batch's end runs with the first env's camera: classified, never read
#pragma unroll
    for (int bi = 0; bi < kDynBatch; ++bi) camx[bi] = bi < nb ? cams[bi] : cams[0];
    unsigned ssr[kDynBatch], ssg[kDynBatch], ssb[kDynBatch];
#pragma unroll
    for (int bi = 0; bi < kDynBatch; ++bi) { ssr[bi] = 0u; ssg[bi] = 0u; ssb[bi] = 0u; }
    const unsigned cnt_a = (unsigned)f.lds_off + (unsigned)(kDynBatch * p.H * 16 + 128 + kDynCntAt * 4);
    const unsigned rowch_a = dyn_rowch_lds(f, p.H);
    {
        // The window rows are taken from the LAST down: each row's pack is pushed into the low byte of the env's 128-bit register cbits (three
        // v_alignbit + one v_lshl_or), so phase C, walking its rows upwards, finds the pack of its next window row in the low byte and shifts it
        // out.  (Until late round 4: a slot counter, four compares and selects per env and row here, three selects and a variable shift in phase C.)
        // Integer sums: the order of the rows does not matter.
        int vl = -1;
        if (f.w1 > rth.vstart) vl = rth.vstart + ((f.w1 - 1 - rth.vstart) / p.rows_per_pass) * p.rows_per_pass;
        const int vstop = max(f.w0, rth.vstart);
        __builtin_amdgcn_s_setprio(2);                                   // phase A is vector-issue bound: the physics wave of this SIMD (the NEXT steps' integration) yields to it and runs in phases B / C
        typedef unsigned lu4 __attribute__((ext_vector_type(4)));
        auto rowch_of = [&](int v) -> lu4 { return *(const __attribute__((address_space(3))) lu4*)(uintptr_t)(rowch_a + ((unsigned)v << 4)); };
        lu4 rc = rowch_of(max(vl, 0));                                   // this row's palette and row-table entry were requested one row earlier
        f2v rt = lrow[max(vl, 0)];
        for (int v = vl; v >= vstop; v -= p.rows_per_pass) {
            const int vn = max(v - p.rows_per_pass, 0);
            const lu4 rcn = rowch_of(vn);
            const f2v rtn = lrow[vn];
            unsigned packs[kDynBatch];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) packs[bi] = 0u;
#if TRS_DYN_ABLATE != 1   /* 1: timing-only, phase A without its classification */
            // ONE branch for the row, all four envs inside it (an env past the batch's end classifies with the first env's camera and is never read):
            // with a test of `bi < nb` per env hipcc gave every env its own exec-masked block, each with its own row-table read and its
            // own counted waits - sixteen dependent LDS round trips per row instead of three
            if (v >= p.uni_rows) {
                classify_pair(rt, camx[0], camx[1], packs[0], packs[1]);
                classify_pair(rt, camx[2], camx[3], packs[2], packs[3]);
            }
#endif
            unsigned cnt[kDynBatch];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) cnt[bi] = *(lds_u32p)(uintptr_t)(cnt_a + (packs[bi] << 2));   // (the four reads together, the pushes underneath)
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
                cbits[bi][3] = __builtin_amdgcn_alignbit(cbits[bi][3], cbits[bi][2], 24);
                cbits[bi][2] = __builtin_amdgcn_alignbit(cbits[bi][2], cbits[bi][1], 24);
                cbits[bi][1] = __builtin_amdgcn_alignbit(cbits[bi][1], cbits[bi][0], 24);
                cbits[bi][0] = (cbits[bi][0] << 8) | packs[bi];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
                ssr[bi] = __builtin_amdgcn_udot4(cnt[bi], rc.x, ssr[bi], false);
                ssg[bi] = __builtin_amdgcn_udot4(cnt[bi], rc.y, ssg[bi], false);
                ssb[bi] = __builtin_amdgcn_udot4(cnt[bi], rc.z, ssb[bi], false);
            }
            rc = rcn; rt = rtn;
        }
    }
#pragma unroll
    for (int bi = 0; bi < kDynBatch; ++bi) {
        if (bi >= nb) continue;
        const unsigned sr = wave_sum_dpp(ssr[bi]), sg = wave_sum_dpp(ssg[bi]), sb = wave_sum_dpp(ssb[bi]);   // (DPP: 6 vector instructions per sum; the ds_bpermute shuffles were 6 LDS round trips)
        if (lane == 63) {
            int* const es = esum + (par * kDynBatch + bi) * 3;
            atomicAdd(&es[0], (int)sr); atomicAdd(&es[1], (int)sg); atomicAdd(&es[2], (int)sb);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    DYN_STAMP(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(dbar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!team_barrier_wait(dbar, nrw * (2 * it + 1), bail)) return false;
    DYN_STAMP(1);
    // (B) delta exactly as ImgPreprocessing computes it (binary64; img_preprocessing.py:88-91), then the palette entries
    {
        const double cnt = (double)(f.w1 - f.w0) * (double)p.W;
        const int per_env = p.H * 4;
        float dl = 0.0f;                                              // lane b of every wave computes env b's delta once (binary64 divisions)
        if (lane < kDynBatch && lane < nb) {
            const int* const es = esum + (par * kDynBatch + lane) * 3;
            double cur = 0.0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) cur = cur + (cnt > 0 ? (double)es[ch] / cnt : 0.0);
            cur = cur + 0.0;
            dl = (float)((f.baseline - cur) / 3);
        }
        float dlt[kDynBatch];
#pragma unroll
        for (int bi = 0; bi < kDynBatch; ++bi) dlt[bi] = __shfl(dl, bi, 64);
        // items: a uniform row (sky, beyond the far plane: its four classes share one colour) is filtered ONCE and written four times; a
        // ground row's four entries are four items.  1,332 items instead of 1,920 entries per batch of four 120-row frames.
        const unsigned* const ltabs = dyn_tabs_lds(lds_base, f, p.H);
        const int items_env = p.uni_rows + 4 * (p.H - p.uni_rows);
        // A thread takes one palette entry for ALL envs of the batch: one read of the raw colour, four independent filter chains (each is ~65 vector
        // instructions behind four dependent table lookups) - until late round 4 an item was (env, entry), found by an integer division, one chain at a time.
        // (An env past the batch's end is filtered with delta 0 into its own, unused, palette.)
        for (int it_e = tid; it_e < items_env; it_e += kRasterThreads) {
            const bool uni = it_e < p.uni_rows;
            const int ent = uni ? 4 * it_e : it_e + 3 * p.uni_rows;          // palette entry (row * 4 + class) of this item
            const uint32_t raw = *(lds_u32p)(uintptr_t)((unsigned)p.off_pal + ((unsigned)ent << 2));
            uint32_t c[kDynBatch];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
#if TRS_DYN_ABLATE == 2   /* timing-only: phase B without the filter arithmetic */
                c[bi] = raw + (unsigned)dlt[bi];
#else
                c[bi] = filter_colour_dev(f, ltabs, raw, dlt[bi]);
#endif
            }
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
                uint32_t* const dst = penv + bi * per_env + ent;
                dst[0] = c[bi];
                if (uni) { dst[1] = c[bi]; dst[2] = c[bi]; dst[3] = c[bi]; }
            }
        }
        if (tid < kDynBatch * 3) esum[(par ^ 1) * kDynBatch * 3 + tid] = 0;   // the next batch's sums start from zero (nobody reads that half now)
    }
    DYN_STAMP(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(dbar, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!team_barrier_wait(dbar, nrw * (2 * it + 2), bail)) return false;
    DYN_STAMP(3);
    // (C) rows outside, envs inside (as phase A): the four envs' palette gathers of a row are sixteen independent LDS reads in flight; one env at a
    // time every row was its own round trip (6.9 of the step's 19 us by the stamps of scripts/dyn_stamps.sh, at 58 % of the vector-issue rate).
    {
        __amdgpu_buffer_rsrc_t rsrc[kDynBatch], drs[kDynBatch];
#pragma unroll
        for (int bi = 0; bi < kDynBatch; ++bi) {
            const int eb = e + min(bi, nb - 1);                               // an env past the batch's end: a descriptor of ZERO bytes - the hardware's bounds check drops its stores
            rsrc[bi] = __builtin_amdgcn_make_buffer_rsrc(img + (size_t)eb * ((size_t)p.gpe * 12), 0, bi < nb ? (int)((size_t)p.gpe * 12) : 0, 0x00020000);
            drs[bi] = rsrc[bi];
            if constexpr (DEPTH) drs[bi] = __builtin_amdgcn_make_buffer_rsrc(dep + (size_t)eb * ((size_t)p.gpe * 4), 0, bi < nb ? (int)((size_t)p.gpe * 16) : 0, 0x00020000);
        }
        const unsigned env_pitch = (unsigned)(p.H * 16);
        for (int v = rth.vstart; v < p.H; v += p.rows_per_pass) {
            const bool in_win = v >= f.w0 && v < f.w1;
            unsigned pack[kDynBatch];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) pack[bi] = 0u;
            if (in_win) {                                                   // phase A's packs of this row (0 for a uniform row), then the next row's move down
#pragma unroll
                for (int bi = 0; bi < kDynBatch; ++bi) {
                    pack[bi] = cbits[bi][0];                                // (only the low byte is read below)
                    cbits[bi][0] = __builtin_amdgcn_alignbit(cbits[bi][1], cbits[bi][0], 8);
                    cbits[bi][1] = __builtin_amdgcn_alignbit(cbits[bi][2], cbits[bi][1], 8);
                    cbits[bi][2] = __builtin_amdgcn_alignbit(cbits[bi][3], cbits[bi][2], 8);
                    cbits[bi][3] >>= 8;
                }
            } else if (v >= p.uni_rows) {
                const f2v rt = lrow[v];
                classify_pair(rt, camx[0], camx[1], pack[0], pack[1]);
                classify_pair(rt, camx[2], camx[3], pack[2], pack[3]);
            }
            const unsigned row_a = (unsigned)f.lds_off + ((unsigned)v << 4);
            unsigned ga[kDynBatch][4];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned cls = __builtin_amdgcn_ubfe(pack[bi], 2 * q, 2);
                    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ga[bi][q]) : "v"(cls), "v"(row_a + (unsigned)bi * env_pitch));
                }
            uint32_t cp[kDynBatch][4];
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi)
#pragma unroll
                for (int q = 0; q < 4; ++q) cp[bi][q] = *(lds_u32p)(uintptr_t)ga[bi][q];
            const unsigned soff = rth.col_off + v * rth.row_bytes;
            unsigned dz = 0; (void)dz;
            if constexpr (DEPTH) dz = __float_as_uint(rth.lrowdepth[v]);
#pragma unroll
            for (int bi = 0; bi < kDynBatch; ++bi) {
                const u3v px3 = {__builtin_amdgcn_perm(cp[bi][1], cp[bi][0], 0x04020100u), __builtin_amdgcn_perm(cp[bi][2], cp[bi][1], 0x05040201u),
                                 __builtin_amdgcn_perm(cp[bi][3], cp[bi][2], 0x06050402u)};
#if TRS_DYN_ABLATE == 3   /* timing-only: phase C without its stores */
                asm volatile("" :: "v"(px3.x), "v"(px3.y), "v"(px3.z)); (void)soff;
#else
                __builtin_amdgcn_raw_buffer_store_b96(px3, rsrc[bi], soff, 0, TRS_STORE_AUX);
                if constexpr (DEPTH) {
                    const u4v d4 = {dz, dz, dz, dz};
                    __builtin_amdgcn_raw_buffer_store_b128(d4, drs[bi], (rth.cg + v * p.gpr) * 16, 0, TRS_STORE_AUX);
                }
#endif
            }
        }
    }
    DYN_STAMP(4);
    return true;
}

// global -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction moves 64 lanes x 16 B = 1 KB, lane-linear, no
// registers and no ds_write pass): `n_waves` waves share the pieces of `bytes` (a multiple of 16), wave `w` of them issues
// its own; the caller waits (s_waitcnt vmcnt(0)) and closes the stage with a barrier.
__device__ __forceinline__ void stage_lds_dma(const unsigned char* blob, int bytes, unsigned lds_byte_addr, int w, int n_waves, int lane)
{
    const u4v* src = reinterpret_cast<const u4v*>(blob);
    const int n16 = bytes >> 4;
    for (int g0 = w * 64; g0 < n16; g0 += n_waves * 64) {
        const int g = g0 + lane;
        if (g < n16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + g),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(lds_byte_addr + g0 * 16), 16, 0, 0);
    }
}

}  // namespace
