# A/B helpers for one gpurun call (source this on the GPU box, from the repo root; results under gpurun_out/ab_<name>.*):
#     source tools/ab.sh && run base && run three BITHTM_LEAN=1 && runlarge large_c128 BITHTM_LEAN2_CLASSIFY=128
# run: the headline at the default arguments, 3 repetitions, no legs; runlarge: the large_pool leg alone.  Each prints the median
# rate and the per-launch times of the timed schedule.  Variants are environment knobs or another build of the
# library (BITHTM_LIBRARY=$PWD/bithtm_amd/<other>.so); join the calls with && so that nothing follows a run that was killed.
set -e
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --steps 2000 --warmup 500 --reps 3 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  echo "$name: $(grep -o 'median [0-9]* timesteps/s' gpurun_out/ab_$name.err | head -1) | $(grep -o 'launches of the timed schedule.*' gpurun_out/ab_$name.err | head -1 | cut -c1-200)"
}
runlarge() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 250 python bench.py --large-pool-only --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  echo "$name: $(grep -o 'median [0-9]* timesteps/s' gpurun_out/ab_$name.err | head -1) | $(grep -o 'launches of the timed schedule.*' gpurun_out/ab_$name.err | head -1 | cut -c1-200)"
}
rundriver() { # name, env...: the driver's arguments (20 timed steps per call), 15 repetitions, no legs
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-large-pool --no-host-fed > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  echo "$name: $(grep -o 'median [0-9]* timesteps/s (min [0-9]*, max [0-9]*)' gpurun_out/ab_$name.err | head -1)"
}
