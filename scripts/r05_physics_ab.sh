#!/bin/bash
# round 5: the physics chain (env_advance) in the in-tree build against scripts/ab_bin/libtrsim_prev.so (the commit before), alternating on one box:
# configs[1] (256 envs, physics only: per launch, 16 steps per launch, posted ticks), the consumer-paced single step, lock step, the env step inside the pilot loop
cd "$(dirname "$0")/.."
for round in 1 2 3; do for v in new prev; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  TRS_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 2000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['also']; p=a['physics_256']
print('$v', 'main', round(d['value']/1e6,2), 'M | physics_256: spl1', p['us_per_step_spl1'], 'us, spl16', p['us_per_step_spl16'], 'us =', round(p['env_steps_per_s_spl16']/1e6,1), 'M, tick launch', p['tick_launch_us'], 'resident', p['tick_resident_us'], 'lock', p['tick_resident_lock_step_us'], '| single_step_call', a['single_step_call']['us_per_call'], '| lock step', a['resident_single_step_call']['lock_step_us_per_call'], '| launch_per_step', a['launch_per_step']['us_per_step'], '| pilot', round(a['pilot_closed_loop']['env_steps_per_s']/1e6,3), round(a['pilot_closed_loop_512x240x320_depth']['env_steps_per_s']/1e6,3), '| hilly', a['hilly_track']['rgb']['us_per_step'], '| shard512', a['shard_512_of_4096']['us_per_step'])"
done; done
