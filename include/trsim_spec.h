/* trsim_spec.h — the frozen numerical specification of the batched env.
 *
 * The reference (Triton-AI/Triton-Racer-Sim) holds NO vehicle dynamics and NO
 * camera renderer: TritonRacerSim/components/gyminterface.py:47-104 is a TCP
 * client of an external closed Unity binary.  Everything
 * in this header is therefore a build-defined free parameter (SURVEY.md
 * Appendix B).  It is the single source of truth for CONSTANTS ONLY; the HIP
 * library (triton-racer-sim_amd/csrc) and the CPU oracle (oracle/) implement
 * the arithmetic below independently, from this text, and must agree
 * bit-for-bit (integer results) / within 1e-5 (pose, speed).
 *
 * Arithmetic rules (both implementations):
 *   R1  all env state is IEEE-754 binary32; add/sub/mul/div/sqrt correctly
 *       rounded; NO floating-point contraction (-ffp-contract=off); an FMA is
 *       used only where the spec writes fma(a,b,c).
 *   R2  sin/cos come from trs_sincos below (never libm / device intrinsics).
 *   R3  the nearest-point search (reference LocationTracker,
 *       TritonRacerSim/components/track_data_process.py:89-104) runs in
 *       binary64 on the binary64 track points: d_i = (|x-xi| + |y-yi|) + |z-zi|,
 *       best initialised to 100.0, strict '<', lowest index wins.
 *       An implementation may first scan only the points of the 3x3 block of TRS_NEAR_GRID_CELL-sized (x, z) cells around
 *       the query: every point outside the block is at least one cell size away in x or z, so a block result with
 *       d < TRS_NEAR_GRID_CELL is the global result (no outside point can beat or tie it); otherwise it must scan all points.
 *
 * ---- trs_sincos(a), |a| <= pi + 1e-3 -------------------------------------
 *   q  = rint(a * TRS_TWO_OVER_PI)              (round-half-even)
 *   r  = fma(q, -TRS_PIO2_HI, a);  r = fma(q, -TRS_PIO2_LO, r)
 *   z  = r*r
 *   ps = fma(fma(TRS_S0, z, TRS_S1), z, TRS_S2)        sin: s = fma(r*z, ps, r)
 *   pc = fma(fma(TRS_C0, z, TRS_C1), z, TRS_C2)        cos: c = fma(z*z, pc, fma(z, -0.5f, 1.0f))
 *   n  = ((int)q) & 3:  0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
 *
 * ---- one env step (controls steer, thr, brk; state x,y,z,yaw,v) ----------
 *   steer = clamp(steer,-1,1); thr = clamp(thr,-1,1); brk = clamp(brk,0,1)
 *   (sd,cd) = trs_sincos(steer * max_steer);  tan_d = sd / cd
 *   a   = thr*accel_max - drag_lin*v
 *   v1  = v + a*dt
 *   dv  = (roll_res + brk*brake_max) * dt
 *   v2  = v1>0 ? max(v1-dv,0) : (v1<0 ? min(v1+dv,0) : 0);  v2 = clamp(v2,-v_rev_max,v_max)
 *   yaw1 = yaw + ((v2*tan_d)*inv_wheelbase)*dt;  if yaw1> PI: yaw1-=TWO_PI; if yaw1<-PI: yaw1+=TWO_PI
 *   (s,c) = trs_sincos(yaw1)           heading: forward = (s, c) in (x, z); Unity y-up, left-handed
 *   x1 = x + (v2*s)*dt;   z1 = z + (v2*c)*dt
 *   idx = L1 nearest raw track point of ((double)x1,(double)y,(double)z1)        [R3]
 *   y1  = (float)Py[idx]
 *   cte = (x1-(float)Px[idx])*tz[idx] - (z1-(float)Pz[idx])*tx[idx]   (+ = right of the centre line)
 *   lost = (best_d >= 100.0);  done = |cte| > offtrack_cte || lost
 *   d = idx - prev_idx wrapped to [-n/2, n/2);  reward = (float)d - (done ? offtrack_penalty : 0)
 *   ep_return += reward; ep_len += 1
 *   reset (usr/reset truthy, or auto_reset && previous done): state := start pose of the env,
 *   no integration this step, reward 0, last_return := ep_return, ep_return := 0, ep_len := 0.
 *
 * ---- class map (the track surface the camera sees; built on the host in binary64) ----
 *   polyline Q: the raw track points in (x, z), consecutive equal points dropped, a closing duplicate of the first dropped;
 *     it is CLOSED: m points, m segments Q[k] -> Q[(k+1) mod m]; s_k = sum of the lengths of segments 0..k-1.
 *   grid: cell = TRS_MAP_CELL_MIN * 2^j with the smallest j >= 0 for which the packed map fits TRS_MAP_LDS_BUDGET:
 *     x0 = floor((min x of Q - map_margin) / cell) * cell,  GW = ceil((max x of Q + map_margin - x0) / cell);  z0, GH likewise;
 *     MW = ceil(GW / 16) 32-bit words per row (16 cells of 2 bits, cell ix in bits 2*(ix mod 16)..+1 of word ix / 16);
 *     it fits when MW * 4 * GH <= TRS_MAP_LDS_BUDGET.
 *   cell (ix, iz), centre c = (x0 + (ix + 0.5) * cell, z0 + (iz + 0.5) * cell):
 *     per segment k: t = clamp(((c - Q[k]) . (Q[k+1] - Q[k])) / |Q[k+1] - Q[k]|^2, 0, 1), foot = Q[k] + t (Q[k+1] - Q[k])
 *     d   = min over k of |c - foot_k|  (Euclidean), taken at the LOWEST k that attains it
 *     arc = s_k + t_k * |Q[k+1] - Q[k]| at that k  (arc length along the centre line)
 *     class = CENTRE  if d <= centre_half and fmod(arc, dash_period) < dash_on      (dashed centre line)
 *             EDGE    else if |d - road_half| <= edge_half                           (solid edge lines, centred on the road edge)
 *             ROAD    else if d < road_half
 *             GRASS   otherwise
 *     the outermost ring of cells (ix = 0, GW-1 or iz = 0, GH-1) is GRASS whatever the rule says: lookups that fall outside
 *     the map are clamped onto it.  (A builder may skip segments farther from a cell than road_half + edge_half + cell:
 *     they cannot change the class.)
 *   An independent numpy restatement of this paragraph checks both builders: tests/test_independent_spec.py.
 *
 * ---- camera (pinhole over the ground plane y = 0; no roll) ----------------
 *   per image row v (tables built on the host in binary64, stored binary32):
 *     f = (H/2)/tan(fov_v/2);  yn = (H/2-(v+0.5))/f
 *     dy = yn*cos(pitch)-sin(pitch);  dz = yn*sin(pitch)+cos(pitch)
 *     dy >= -1e-6            -> SKY row
 *     t = cam_h/(-dy); t*dz > z_far -> FAR row (fog colour)
 *     else GROUND row: row_lz[v] = t*dz/cell, row_k[v] = (t/f)/cell
 *   per env: camx = ((x1 + cam_fwd*s) - map_x0)*inv_cell, camz likewise with c, z1, map_z0
 *   per pixel (u,v): uf = (float)u + 0.5f - W/2
 *     ax = fma(row_lz[v], s, camx); az = fma(row_lz[v], c, camz)
 *     dx = row_k[v]*c;              dzz = -(row_k[v]*s)
 *     gx = fma(uf, dx, ax); gz = fma(uf, dzz, az)
 *     ix = clamp((int)floor(gx), 0, GW-1); iz = clamp((int)floor(gz), 0, GH-1)
 *     cls = 2-bit class of cell (ix, iz)  (the map border is class 0)
 *     rgb = palette[v][cls]
 *   SKY/FAR rows have row_lz = row_k = 0 and a palette whose 4 entries are equal.
 *   depth channel (optional, binary32 [H][W]): z-depth of the ground plane, constant along an image row:
 *     GROUND row: (float)(t*dz) world units;  SKY and FAR rows: (float)z_far.
 *
 * ---- tracks with elevation (round 5) -----------------------------------------
 *   The raw points carry a height y (reference car_templates/track_data/mountain_track.json: y in 3.19 .. 7.35).  A track is HILLY when
 *   max y - min y > TRS_HILL_MIN_RANGE; flat tracks (generated_track: 0.011) keep everything above, bit for bit.  On a hilly track the car
 *   stands on the road's local slope and looks at a road that tilts against it ahead: the camera is fixed to the car, so what changes in
 *   its image is the angle between its axis and the ground plane AHEAD.  The ground the camera sees is modelled as ONE plane per env and
 *   frame — through the camera's foot point, tilted against the car's own plane by the change of slope between the car's track point
 *   and a point TRS_HILL_AHEAD samples further on — i.e. the flat-ground camera above with a PER-ENV pitch (height cam_h and forward
 *   offset cam_fwd unchanged: the rotation is taken about the camera, a documented approximation).  Row tables and the fogged palette
 *   then depend on the env and the frame; they are evaluated in binary32 with the operation order below (host tables in binary64 as
 *   before):
 *   host, binary64, per raw point i (indices wrap, the track is closed):
 *     h_i     = horizontal length of the step from point i to point i+1:  sqrt((Px[i+1]-Px[i])^2 + (Pz[i+1]-Pz[i])^2)   (0 for duplicates)
 *     d_i     = h_{i-L} + h_{i-L+1} + ... + h_{i+L-1}    (the path from point i-L to point i+L, summed in this order), L = TRS_HILL_SPAN
 *     g_i     = d_i > 1e-9 ? (Py[i+L] - Py[i-L]) / d_i : 0                      (smoothed grade)
 *     theta_i = atan(g_i)
 *     dpitch[i] = (float)clamp(theta_{i+A} - theta_i, -TRS_HILL_MAX_DPITCH, TRS_HILL_MAX_DPITCH),  A = TRS_HILL_AHEAD
 *     sky[v]  = the SKY colour of image row v as above, for EVERY row v (g = min((v+0.5)/(H/2), 1))
 *     far     = the FAR colour as above;  inv_f = (float)(1/f); hh = (float)(H/2); pitch_f = (float)pitch; cam_h_f, z_far_f = (float) of theirs;
 *     inv_zfar_f = (float)(1/z_far); inv_cell_f = (float)(1/cell) (exact); fog_f = (float)TRS_FOG_MAX; base / fog colours as binary32
 *   per env and frame, binary32 (R1: every operation rounded, no contraction), idx = the step's nearest raw track point:
 *     P = pitch_f + dpitch[idx];   (sp, cp) = trs_sincos(P)
 *     per image row v:  yn = (hh - ((float)v + 0.5f)) * inv_f;   dy = yn*cp - sp;   dz = yn*sp + cp
 *       dy >= -1e-6f                  -> SKY row:  row_lz = row_k = 0, depth z_far_f, the four class colours = sky[v]
 *       t = cam_h_f / (-dy);  zd = t*dz;  zd > z_far_f -> FAR row: row_lz = row_k = 0, depth z_far_f, colours = far
 *       else GROUND row: row_lz = zd * inv_cell_f;  row_k = (t * inv_f) * inv_cell_f;  depth = zd
 *            fw = fog_f * (zd * inv_zfar_f);  colour[c][ch] = (int)((base[c][ch] * (1.0f - fw) + fog[ch] * fw) + 0.5f)
 *     pixels exactly as above with these row tables.  The depth frame is constant along a row of one frame, and now differs from env to env
 *     and from frame to frame with the slope ahead.
 */
#ifndef TRSIM_SPEC_H
#define TRSIM_SPEC_H

/* trig */
#define TRS_PI            3.14159274101257324f   /* (float)pi */
#define TRS_TWO_PI        6.28318548202514648f
#define TRS_TWO_OVER_PI   0.636619746685028076f
#define TRS_PIO2_HI       1.5707963705062866211f /* (float)(pi/2) */
#define TRS_PIO2_LO      -4.3711388286737928865e-08f /* (float)(pi/2 - PIO2_HI) */
#define TRS_S0           -1.9515295891e-4f
#define TRS_S1            8.3321608736e-3f
#define TRS_S2           -1.6666654611e-1f
#define TRS_C0            2.443315711809948e-5f
#define TRS_C1           -1.388731625493765e-3f
#define TRS_C2            4.166664568298827e-2f

/* default physics parameters (trs_config overrides) */
#define TRS_DEF_DT              0.05f          /* 20 Hz: car_templates/manage.py:38 */
#define TRS_DEF_MAX_STEER       0.43633231520652770996f /* 25 deg */
#define TRS_DEF_INV_WHEELBASE   0.8333333134651184082f /* 1/1.2 */
#define TRS_DEF_ACCEL_MAX       10.0f
#define TRS_DEF_DRAG_LIN        0.5f           /* terminal speed 20 = pilots' full scale, keras_pilot.py:83 */
#define TRS_DEF_ROLL_RES        0.3f
#define TRS_DEF_BRAKE_MAX       15.0f
#define TRS_DEF_V_MAX           25.0f
#define TRS_DEF_V_REV_MAX       5.0f
#define TRS_DEF_OFFTRACK_CTE    3.0f
#define TRS_DEF_OFFTRACK_PENALTY 10.0f
#define TRS_LOST_L1             100.0          /* track_data_process.py:93 */

/* default track-surface parameters (world units) */
#define TRS_DEF_ROAD_HALF       2.0
#define TRS_DEF_EDGE_HALF       0.10
#define TRS_DEF_CENTRE_HALF     0.075
#define TRS_DEF_DASH_PERIOD     3.0
#define TRS_DEF_DASH_ON         1.5
#define TRS_DEF_MAP_MARGIN      2.5            /* world units of class-0 border around the track bbox (>= road_half + edge_half + one cell;
                                                  lookups outside the map clamp to this grass border) */
#define TRS_NEAR_GRID_CELL      4.0            /* nearest-point accelerator: 3x3 block of cells of this size; exact by the rule below */
#define TRS_MAP_CELL_MIN        0.125          /* cell sizes are 0.125 * 2^k so 1/cell is exact */
#define TRS_MAP_LDS_BUDGET      (96 * 1024)    /* the packed 2-bit map must fit this many bytes */

/* default camera */
#define TRS_DEF_FOV_V_DEG       80.0           /* gyminterface.py:27 (unused there) */
#define TRS_DEF_CAM_H           1.0
#define TRS_DEF_CAM_PITCH_DEG   10.0
#define TRS_DEF_CAM_FWD         0.3f
#define TRS_DEF_Z_FAR           40.0

/* tracks with elevation */
#define TRS_HILL_MIN_RANGE      0.25           /* max y - min y of the raw points above which a track is hilly */
#define TRS_HILL_SPAN           8              /* grade over +- this many samples */
#define TRS_HILL_AHEAD          24             /* the slope this many samples ahead against the slope here */
#define TRS_HILL_MAX_DPITCH     0.2            /* rad */

/* map classes */
#define TRS_CLS_GRASS   0
#define TRS_CLS_ROAD    1
#define TRS_CLS_EDGE    2
#define TRS_CLS_CENTRE  3

/* base colours, RGB */
#define TRS_RGB_GRASS   { 58, 132,  62}
#define TRS_RGB_ROAD    { 92,  92,  98}
#define TRS_RGB_EDGE    {236, 236, 236}
#define TRS_RGB_CENTRE  {232, 200,  40}
#define TRS_RGB_FOG     {176, 196, 208}
#define TRS_RGB_SKY_TOP {104, 156, 228}
#define TRS_RGB_SKY_HOR {192, 216, 240}
#define TRS_FOG_MAX     0.65                   /* fog weight at z_far */

/* synthetic control generator: SplitMix64 keyed by (seed, global env id, step)
 *   z = seed + ((env_gid << 32) | step) * 0x9E3779B97F4A7C15;  z ^= z>>30; z *= 0xBF58476D1CE4E5B9;
 *   z ^= z>>27; z *= 0x94D049BB133111EB; z ^= z>>31
 *   us = (float)(z >> 40) * 2^-24;  ut = (float)((z >> 16) & 0xFFFFFF) * 2^-24
 *   steer_raw = us*2 - 1;  sf = sf + 0.1f*(steer_raw - sf);  steer = sf
 *   thr = 0.2f + 0.6f*ut;  brk = 0
 */
#define TRS_SYNTH_SEED      0x5EEDull
#define TRS_SYNTH_ALPHA     0.1f
#define TRS_SYNTH_THR_LO    0.2f
#define TRS_SYNTH_THR_SPAN  0.6f
#define TRS_START_STRIDE    37                 /* env i starts at track point (37*i) mod n */

#endif /* TRSIM_SPEC_H */
