// trsim_env.hpp — the handle behind `trs_env*` (include/trsim.h), shared by the translation units of libtrsim.so.
// Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/trsim.h"
#include "trsim_device.hpp"
#include "trsim_tables.hpp"

namespace trsim { struct Resident; struct Comm; }

struct trs_env {
    trs_config cfg{};
    int device = 0, n = 0, H = 0, W = 0, cu_count = 0;
    hipStream_t sP = nullptr;            // the handle's stream: every launch, copy and timing event
    hipEvent_t ev[8] = {};
    // device memory
    unsigned char* slab = nullptr;       // state + controls
    uint8_t* img[2] = {nullptr, nullptr};
    bool uniform_ok[2] = {false, false}; // frame buffer b holds the current palette's uniform rows (sky, beyond the far plane) of every env (launch_step)
    float* depth[2] = {nullptr, nullptr};
    unsigned char* blob_p = nullptr;     // physics LDS image
    unsigned char* blob_r = nullptr;     // raster LDS image
    float* tangent = nullptr;
    float* start_yaw = nullptr;
    float4* cam = nullptr;               // [kRing][n]
    float* cam_pitch = nullptr;          // [kRing][n] the frames' view pitches (tracks with elevation)
    float* dpitch = nullptr;             // [n_points] view pitch per raw track point (device: (float)pitch + dpitch[i])
    bool hilly = false;                       // the loaded track has elevation (include/trsim_spec.h): the HILLS instantiations of the step kernels run
    trsim::HillBlock hill_host{};             // host copy of the block behind the raster image (trs_load_track fills the camera part, upload_palette the frame filter)
    unsigned long long* stats = nullptr;
    double* loc_q = nullptr; int32_t* loc_out = nullptr; int loc_cap = 0;
    uint8_t* pre = nullptr;              // processed frames of the env (trs_preprocess with d_dst == NULL)
    trs_pre_config frame_filter{}; bool has_frame_filter = false, filter_dynamic = false;   // trs_set_frame_filter
    unsigned char* pinned = nullptr; size_t pinned_bytes = 0;   // trs_fetch_outputs staging (hipHostMalloc)
    int32_t* mux_state = nullptr; int mux_tick = 0;   // ControlMultiplexer state per car (trs_control_mux)
    uint8_t *tmp_in = nullptr, *tmp_out = nullptr; float* tmp_f = nullptr; size_t tmp_cap = 0;   // host-frame staging
    int* hsv_tab = nullptr;
    unsigned* dyn_tab = nullptr;        // FParams::tabs of the dynamic-brightness frame filter that is set (hsv reciprocals | in-range byte masks | sel)
    unsigned char* edge_scratch = nullptr; size_t edge_scratch_bytes = 0;   // work arrays of the Canny layer for frames beyond LDS
    trsim::PParams pp{};
    trsim::RParams rp{};
    trsim::TrackTables tab;
    bool track_loaded = false;
    int lds_p = 0, lds_r = 0, pts_bytes = 0;
    uint64_t step_count = 0;
    float *ctl_steer = nullptr, *ctl_thr = nullptr, *ctl_brk = nullptr;
    uint8_t* ctl_reset = nullptr;
    size_t img_bytes = 0;
    int lds_step = 0, lds_off_phys = 0, max_steps_per_launch = 1;
    float* seq_buf = nullptr; size_t seq_cap = 0;   // device copy of host control sequences (trs_step_sequence_host)
    int seq_stride = 0;                  // trs_step_sequence: n_envs while a sequence call is running, else 0
    int max_steps_dyn = 0;               // steps per launch that still fit beside the dynamic-brightness palette (0 = it does not fit at all)
    void* pilot = nullptr;               // trsim_pilot.hip context (cnn_2d_speed_control weights + activations)
    trs_pilot_tuning pilot_tuning{}; bool has_pilot_tuning = false;   // trs_pilot_set_tuning: kernel choices of the next trs_pilot_load
    unsigned long long* fault = nullptr; // pinned host word the kernels set when they refuse to run (dynamic LDS not at offset 0)
    float* glue = nullptr; size_t glue_bytes = 0;   // device scratch of the *_host control glue (trs_driver_assist_host, trs_control_mux_host)
    void* scratch[32] = {}; size_t scratch_bytes[32] = {};   // trs_scratch
    uint64_t d2h_bytes = 0, h2d_bytes = 0;                  // trs_counters: what the library itself copied
    trsim::Comm* comm = nullptr;         // trsim_comm.hip: the RCCL communicator of trs_comm_init, nullptr = none
    hipEvent_t ev_order = nullptr;       // trs_stream_wait_external / trs_stream_signal_external
    trsim::Resident* res = nullptr;      // trsim_resident.hip: the resident worker (trs_set_step_mode), nullptr = never used
};

// ---- trsim_resident.hip (the resident worker: one step per trs_step call without a launch per step) ----
namespace trsim {
bool resident_on(const trs_env* e);                       // resident mode selected for this handle
void resident_retry(trs_env* e);                          // a handle that fell back to launches (GPU shared with another process) tries resident mode again when due
// hand n steps to the worker; controls as in trs_step (device pointers, or host-pinned pointers the device can read);
// stride: elements between consecutive steps' control arrays (0 = held), synth: controls from the spec's generator
// Both return TRS_OK, an error (< 0) or kResidentFellBack (> 0, not an error): the worker's launch was found not co-resident (another process's
// worker on the GPU), the handle is back in TRS_STEP_LAUNCH, the first *n_done steps of the call are on the stream as launches and the caller
// launches the rest itself.
constexpr int kResidentFellBack = 1;
int resident_post(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, int n, size_t stride, int* n_done);
int resident_post_host(trs_env* e, const float* h_st, const float* h_th, const float* h_br, const uint8_t* h_rs, int n_steps, int* n_done);
hipStream_t resident_copy_stream(trs_env* e);             // a stream that is not blocked by the worker (the handle's own when none runs)
int resident_wait(trs_env* e);                            // every posted step complete (the worker stays resident)
int resident_quiesce(trs_env* e);                         // ... and the worker has left the GPU: the stream is free again
void resident_note_launch(trs_env* e);                    // a step was launched on the stream while resident mode is selected (pilot loop)
void resident_destroy(trs_env* e);
int sync_handle(trs_env* e);                               // the handle's stream is idle (a resident worker is asked to leave first)
int quiesce_handle(trs_env* e);                            // a resident worker has left; queued work may still be running
void comm_destroy(trs_env* e);
bool resident_running(const trs_env* e);
bool resident_fits_dynamic_filter(const trs_env* e);      // the worker's LDS need WITH the dynamic-brightness palettes still fits a CU
void resident_clear_fault(trs_env* e);                     // trs_load_track puts every env on a defined state again
int check_fault(trs_env* e);                               // TRS_ERR_DEVICE (sticky) once a kernel has reported a layout fault
}  // namespace trsim
