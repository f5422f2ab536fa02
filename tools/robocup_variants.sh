#!/bin/bash
# The RoboCup Full step kernel's build variants side by side (whole-episode mean launch, HBM-side traffic, VALU instructions):
#   default            one environment per wave, one lane per foot (RC_FULL_EPW=1)
#   rpl2               two environments per wave, one lane per robot, common substep in registers (-DRC_FULL_EPW=2)
#   sched              rpl2 for light environments + solo foot-per-lane waves for those with cached arbiters (-DRC_SCHED=1)
# Usage (GPU box; the variant libraries dynenv_amd/libdynenv_hip_{rpl2,sched}.so built beforehand): bash tools/robocup_variants.sh
for v in hip hip_rpl2 hip_sched; do
  export DYNENV_HIP_LIB=$PWD/dynenv_amd/libdynenv_$v.so
  K=rc_step_kernel; [ $v = hip_sched ] && K=rc_step_sched_kernel
  echo "== $v"
  python bench.py --workload robocup --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step (2 episodes, resets included) %.3f   whole-episode mean launch %.3f ms' % (d['ms_per_step'], d['ms_per_step_full_episode']))"
  bash tools/pmc_traffic.sh robocup $K v_$v 240 2>&1 | grep -E "traffic|SQ_INSTS_VALU|SQ_INSTS_VMEM|SQ_WAVE_CYCLES"
done
