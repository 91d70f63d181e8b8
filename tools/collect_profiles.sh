#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root.  Order matters: the PMC passes come FIRST and record the model state
# they saw (bench.py's roofline.state), so that the bench lines written afterwards carry the current traffic figure --
# bench.py refuses a recorded figure whose state does not match its own.  Then: bench lines, rocprofv3 kernel stats,
# scan stress, cold phase, host-fed rate, sharded rehearsals, device-clock timelines, the stamped diagnostic builds.
# Everything lands in gpurun_out/profile_<tag>/; copy what should be judged into profiles/.
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
export OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# ---- PMC passes: each counter in its own run, kernel-trace only; the bench with the DRIVER's arguments (eager launches of the
# same schedule under the profiler) and the configs[4] leg.  (The sharded launches -- all eight ranks in one process,
# tools/shard_rehearsal.py -- crash under rocprofv3 on this image, eager launches included: no PMC pass of them.)
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed > $OUT/pmc_$ctr.json 2> $OUT/pmc_$ctr.log; echo "$ctr exit=$?"
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_stress_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --stress-only --no-cpu-baseline > $OUT/pmc_stress_$ctr.log 2>&1; echo "$ctr (stress) exit=$?"
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_large_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --large-pool-only --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_large_$ctr.json 2> $OUT/pmc_large_$ctr.log; echo "$ctr (large pool) exit=$?"
done
python3 - <<'PY'
import csv, glob, os, collections, json
out = os.environ["OUT"]
def summarise(prefixes, tag):
    summary = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = collections.defaultdict(list)
        for prefix, suffix in prefixes:
            for f in glob.glob(os.path.join(out, f"{prefix}_{ctr}", "*", "*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == ctr:
                        d[r["Kernel_Name"].split("(")[0] + suffix].append(float(r["Counter_Value"]))
        summary[ctr] = {k: {"launches": len(v), "mean_last150_KB": sum(v[-150:]) / len(v[-150:])} for k, v in d.items()}
    return summary
s = summarise([("pmc", ""), ("pmc_stress", " [configs[4] leg]")], "")
s["large_pool"] = summarise([("pmc_large", "")], "")     # the large_pool leg's own pass (bench.py: recorded_traffic(section="large_pool"))
try:
    line = json.loads(open(os.path.join(out, "pmc_large_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    s["large_pool"]["state"] = line["roofline"]["state"]
    s["large_pool"]["command"] = "bench.py --large-pool-only --steps 20 --warmup 5 --no-cpu-baseline (under rocprofv3: eager launches)"
except Exception as e:
    s["large_pool"]["state_error"] = repr(e)
try:        # the state the PMC pass saw: what bench.py compares its own with (recorded_traffic)
    line = json.loads(open(os.path.join(out, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    s["state"] = line["roofline"]["state"]
    s["command"] = "bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed (under rocprofv3: eager launches)"
except Exception as e:
    s["state_error"] = repr(e)
json.dump(s, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
# (the names bench.py looks for, whatever the tag of this collection)
cp $OUT/pmc_summary.json $GRAFT_REPO_ROOT/profiles/r04_pmc_summary.json
# ---- bench lines (they read the summary just written)
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.log; echo "bench (driver args) exit=$?"
timeout -k 10 400 python bench.py --no-stress > $OUT/bench.json 2> $OUT/bench.log; echo "bench exit=$?"
timeout -k 10 200 python tools/step_timeline.py > $OUT/step_timeline.txt 2>&1
# the large_pool leg's state: timeline of its step, row lengths and what passes the first-chunk test, how the pool grows
TIMELINE_WORKLOAD=large TIMELINE_WARMUP=3500 timeout -k 10 200 python tools/step_timeline.py > $OUT/timeline_large.txt 2>&1
timeout -k 10 200 python tools/row_lengths.py large > $OUT/row_lengths_large.txt 2>&1
timeout -k 10 200 python tools/pool_growth.py 350 12 > $OUT/pool_growth.txt 2>&1
timeout -k 10 300 python tools/step_spans.py 16 > $OUT/step_spans.txt 2>&1
timeout -k 10 200 python tools/soak.py > $OUT/soak.txt 2>&1
# the four-launch schedule, for comparison
BITHTM_LEAN=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-stress --no-large-pool --no-host-fed > $OUT/bench_four_launches.json 2> $OUT/bench_four_launches.log; echo "bench (four launches) exit=$?"
# ... and the three-launch one (the default of rounds 3 and 4 until the two-launch schedule)
BITHTM_LEAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-stress --no-large-pool --no-host-fed > $OUT/bench_three_launches.json 2> $OUT/bench_three_launches.log; echo "bench (three launches) exit=$?"
BITHTM_LEAN=1 timeout -k 10 200 python tools/step_timeline.py > $OUT/step_timeline_three_launches.txt 2>&1
# from scratch: the first 250 steps (every column bursting at first, ~1 300 new segments per step)
timeout -k 10 200 python bench.py --no-cpu-baseline --no-stress --no-large-pool --no-host-fed --pretrain 0 --steps 250 --warmup 0 --reps 1 > $OUT/bench_cold_250.json 2> $OUT/bench_cold_250.log; echo "bench (cold) exit=$?"
timeout -k 10 200 python tools/pcie_rate.py > $OUT/pcie_rate.txt 2>&1
timeout -k 10 200 python tools/hostfed_profile.py >> $OUT/pcie_rate.txt 2>&1
for w in 2 4 8; do timeout -k 10 200 python tools/shard_rehearsal.py --world $w; done > $OUT/shard_rehearsal.jsonl 2>&1
for w in 2 4 8; do timeout -k 10 250 python tools/shard_rehearsal.py --large --world $w; done > $OUT/shard_rehearsal_large.jsonl 2>&1
# bench.py --gpus 2 as two processes on this one GPU, records staged through the host over gloo (the multi-process flow of
# bench_sharded.py; the RCCL path needs one GPU per rank)
BITHTM_DIST_BACKEND=gloo BITHTM_SINGLE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_2ranks_one_gpu_gloo.json 2> $OUT/bench_2ranks_one_gpu_gloo.log; echo "2-rank rehearsal exit=$?"
timeout -k 10 600 python tools/scan_stress.py --segments 250000 1000000 4000000 16000000 --slots 64 > $OUT/scan_stress.jsonl 2> $OUT/scan_stress.log
for sl in 64 128 256; do timeout -k 10 170 python tools/scan_stress.py --segments 1572864 --slots $sl; done > $OUT/scan_stride.jsonl 2>> $OUT/scan_stress.log
timeout -k 10 300 python tools/scan_stress.py --columns 262144 --cells 16 --segments 4000000 16000000 --slots 64 >> $OUT/scan_stress.jsonl 2>> $OUT/scan_stress.log
cd /tmp
# ---- rocprofv3 kernel stats: (a) one role per launch, (b) the pipelined schedule bench.py times (eager instead of hipGraph:
# rocprofv3 crashes on graph replay here), (c) the configs[4] leg alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 100 --reps 1 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed --no-graph --no-pipeline > $OUT/stats.log 2>&1; echo "stats exit=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- python3 $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 100 --reps 1 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed --no-graph > $OUT/stats_pipelined.log 2>&1; echo "stats_pipelined exit=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_large -- python3 $GRAFT_REPO_ROOT/bench.py --large-pool-only --steps 300 --warmup 50 --reps 1 --no-cpu-baseline --no-graph > $OUT/stats_large.log 2>&1; echo "stats_large exit=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_stress -- python3 $GRAFT_REPO_ROOT/bench.py --stress-only --no-cpu-baseline > $OUT/stats_stress.log 2>&1; echo "stats_stress exit=$?"
python3 - <<'PY'
import csv, glob, os, collections, json
out = os.environ["OUT"]
# per-kernel duration of the LAST 300 launches of the pipelined trace (steady state; the --stats file averages the whole run)
def last300(pattern):
    res = {}
    for f in glob.glob(os.path.join(out, pattern, "*", "*kernel_trace.csv")):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        res = {k: {"launches": len(v), "mean_last300_us": sum(v[-300:]) / len(v[-300:]), "mean_us": sum(v) / len(v)} for k, v in d.items()}
    return res
summary = last300("stats_pipelined")
summary["large_pool"] = last300("stats_large")       # (bench.py reads the leg's figure from this section)
json.dump(summary, open(os.path.join(out, "pipelined_kernel_us.json"), "w"), indent=1)
PY
find $OUT -name "*kernel_trace.csv" -delete
# ---- diagnostic builds (device-clock stamps inside the roles), then the normal one again
cd $GRAFT_REPO_ROOT
BITHTM_EXTRA_FLAGS=-DBITHTM_EMIT_STAMPS python -m bithtm_amd.build --force > /dev/null 2>&1 && { timeout -k 10 200 python tools/emit_phases.py > $OUT/emit_phases.txt 2>&1; timeout -k 10 200 python tools/shard_phases.py emit > $OUT/shard_phases.txt 2>&1; }
BITHTM_EXTRA_FLAGS=-DBITHTM_SHARD_STAMPS python -m bithtm_amd.build --force > /dev/null 2>&1 && timeout -k 10 200 python tools/shard_phases.py select >> $OUT/shard_phases.txt 2>&1
BITHTM_EXTRA_FLAGS=-DBITHTM_OVERLAP_STAMPS python -m bithtm_amd.build --force > /dev/null 2>&1 && timeout -k 10 200 python tools/overlap_phases.py > $OUT/overlap_phases.txt 2>&1
BITHTM_EXTRA_FLAGS=-DBITHTM_ROWS_STAMPS python -m bithtm_amd.build --force > /dev/null 2>&1 && timeout -k 10 200 python tools/rows_phases.py > $OUT/rows_phases.txt 2>&1
python -m bithtm_amd.build --force > /dev/null 2>&1
ls -R $OUT | head -80
