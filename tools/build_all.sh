#!/bin/bash
# build_all.sh          every native piece in-tree: the product library + its test-caps build (hipcc, gfx950) and the CPU oracle (gcc)
# build_all.sh clean    remove what sweeps, profiles and sanitizer runs leave beside them (libdynenv_hip_<variant>.so, liboracle_asan.so,
#                       tools/probe binaries): git ignores them, but `gpurun` ships every built .so in the tree to the GPU box with each
#                       lease.  Keeps dynenv_amd/libdynenv_hip.so, libdynenv_hip_testcaps.so and oracle/liboracle.so.
cd "$(dirname "$0")/.."
if [ "$1" = "clean" ]; then
  for f in dynenv_amd/libdynenv_hip_*.so; do
    case "$f" in dynenv_amd/libdynenv_hip_testcaps.so) ;; *) [ -e "$f" ] && rm -v "$f";; esac
  done
  rm -fv oracle/liboracle_asan.so dynenv_amd/*.o dynenv_amd/*.so.tmp tools/probe/latency_probe tools/probe/exec_probe
  find gpurun_out -mindepth 1 -maxdepth 1 -mtime +0 -exec rm -rf {} + 2>/dev/null   # scratch older than a day
  exit 0
fi
python3 -c "import __graft_entry__ as g; g.build()"
