#!/bin/bash
# build the product library and the -DDRV_PROFILE variant; fail loudly (so that `&& gpurun ...` never runs a stale .so)
set -e
cd "$(dirname "$0")/.."
python -c "from dynenv_amd import build; build.build(force=True)"
python -c "from dynenv_amd import build as b; b.build(out='dynenv_amd/libdynenv_hip_prof.so', defines=('DRV_PROFILE',))"
