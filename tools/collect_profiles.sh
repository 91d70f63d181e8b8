#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats, PMC
# passes (each counter in its own run, kernel-trace only), scan stress, PCIe-inclusive rate, the
# device-clock timeline of the pipelined step and the micro-benchmarks (launch boundary, attainable
# read bandwidth, cost of a vector-memory instruction; build them first:
# for t in launch_anatomy hbm_read ta_rate; do hipcc --offload-arch=gfx950 -O3 -o tools/$t tools/$t.hip; done).
# Everything lands in gpurun_out/profile_<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
export OUT
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.log; echo "bench exit=$?"
timeout -k 10 300 python tools/pcie_rate.py > $OUT/pcie_rate.txt 2>&1
timeout -k 10 200 python tools/step_timeline.py > $OUT/step_timeline.txt 2>&1
[ -x tools/launch_anatomy ] && timeout -k 10 120 ./tools/launch_anatomy > $OUT/launch_anatomy.txt 2>&1
[ -x tools/hbm_read ] && timeout -k 10 120 ./tools/hbm_read 1024 > $OUT/hbm_read.txt 2>&1
[ -x tools/ta_rate ] && timeout -k 10 120 ./tools/ta_rate > $OUT/ta_rate.txt 2>&1
timeout -k 10 600 python tools/scan_stress.py --segments 250000 1000000 4000000 16000000 --slots 64 > $OUT/scan_stress.jsonl 2> $OUT/scan_stress.log
timeout -k 10 600 python tools/scan_stress.py --segments 1000000 --slots 128 >> $OUT/scan_stress.jsonl 2>> $OUT/scan_stress.log
cd /tmp && export TMPDIR=/tmp
# (a) one role per launch: the kernels the roofline line names, standalone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 500 --no-cpu-baseline --no-graph --no-pipeline > $OUT/stats.log 2>&1; echo "stats exit=$?"
# (b) the pipelined schedule bench.py times by default (eager instead of hipGraph: rocprofv3 crashes on graph replay here)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 500 --no-cpu-baseline --no-graph > $OUT/stats_pipelined.log 2>&1; echo "stats_pipelined exit=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 600 --no-cpu-baseline --no-graph --no-pipeline > $OUT/pmc_$ctr.log 2>&1; echo "$ctr exit=$?"
done
# keep the merged-back payload small: drop the per-dispatch traces of the PMC runs after summarising
python3 - <<'PY'
import csv, glob, os, collections, json
out = os.environ.get("OUT") or glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "profile_*"))[-1]
summary = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, f"pmc_{ctr}", "*", "*counter_collection.csv")):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        summary[ctr] = {k: {"launches": len(v), "mean_last150_KB": sum(v[-150:]) / len(v[-150:])} for k, v in d.items()}
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls -R $OUT | head -40
