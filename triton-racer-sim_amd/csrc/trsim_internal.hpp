// trsim_internal.hpp — what trsim_pilot.hip needs from the env handle defined in trsim_hip.hip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/trsim.h"

struct TrsEnvView {
    int device, n, H, W, render;
    hipStream_t stream;
    const uint8_t* latest_frame;      // uint8[n][H][W][3] of the last completed step, or nullptr
    const float* speed;               // 'gym/speed'
    const int32_t* seg_idx; int n_points;   // LocationTracker index and track length ('loc/segment' = idx / n_points * 10)
    float *ctl_steer, *ctl_thr, *ctl_brk;   // the handle's own control staging arrays (device)
    uint64_t step_count;
    unsigned long long* stats;        // TRS_F_STATS (device, uint64[64])
};

bool trs_internal_view(trs_env* e, TrsEnvView* out);
void** trs_internal_pilot_slot(trs_env* e);
const trs_pilot_tuning* trs_internal_pilot_tuning(trs_env* e);   // what trs_pilot_set_tuning stored, or nullptr (defaults)
void trs_internal_set_pilot_tuning(trs_env* e, const trs_pilot_tuning* t);
int trs_internal_fail(int code, const std::string& msg);
int trs_internal_step_launch(trs_env* e, const float* d_st, const float* d_th, const float* d_br);   // one env step by launch, whatever the step mode
// one env step with index `step` by launch, on the handle's stream, WITHOUT moving the step counter: a step that was posted to a resident worker
// (the counter moved at the post) and has to run as a launch after all (trsim_resident.hip, fall_back_to_launches)
int trs_internal_replay_launch(trs_env* e, const float* st, const float* th, const float* br, const uint8_t* rs, int synth, uint64_t step);
void trs_internal_note(const std::string& msg);            // what trs_last_error() returns, without failing the call
void trs_internal_count(trs_env* e, uint64_t d2h_bytes, uint64_t h2d_bytes);   // trs_counters bookkeeping for copies made outside trsim_hip.hip
void trs_pilot_free(void* ctx);       // defined in trsim_pilot.hip, called by trs_destroy
