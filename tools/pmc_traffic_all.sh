#!/bin/bash
# HBM-side traffic per launch of the step kernels of all four bench.py workloads over one whole episode each -> gpurun_out/pmc_traffic.json
# (copy to profiles/pmc_traffic.json: bench.py reports it as roofline.traffic).  rocprofv3 FETCH_SIZE and WRITE_SIZE in separate
# passes, FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-B fabric requests at 64 B).   Usage (GPU box): bash tools/pmc_traffic_all.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
for W in driving:600 robocup:240 driving_partial:600 robocup_partial:240; do
  NAME=${W%%:*}; STEPS=${W##*:}
  for P in f:FETCH_SIZE w:WRITE_SIZE; do
    N=${P%%:*}; C=${P##*:}
    rm -rf gpurun_out/pta_${NAME}_$N
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pta_${NAME}_$N -- python3 bench.py --workload $NAME --steps $STEPS --warmup 0 --no-cpu-baseline --no-extra-legs > gpurun_out/pta_${NAME}_$N.json 2> gpurun_out/pta_${NAME}_$N.err
  done
done
python3 - <<'PY'
import csv, glob, json, collections
out = {"note": "bytes per launch, mean over the launches of whole episodes at 4096 envs; FETCH_SIZE x 2 (gfx950) + WRITE_SIZE; tools/pmc_traffic_all.sh"}
for name in ("driving", "robocup", "driving_partial", "robocup_partial"):
    agg = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
    for f in glob.glob("gpurun_out/pta_%s_*/*/*counter_collection.csv" % name):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in agg and "step" in r["Kernel_Name"]:
                agg[r["Counter_Name"]][r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k in agg["FETCH_SIZE"]:
        if k in agg["WRITE_SIZE"]:
            f = sum(agg["FETCH_SIZE"][k]) / len(agg["FETCH_SIZE"][k]) * 1024 * 2
            w = sum(agg["WRITE_SIZE"][k]) / len(agg["WRITE_SIZE"][k]) * 1024
            out[k + "_bytes_per_launch"] = f + w
            out[k + "_detail"] = {"fetch_bytes_x2": f, "write_bytes": w, "launches": len(agg["FETCH_SIZE"][k]), "workload": name}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
