set -e
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --steps 2000 --warmup 500 --reps 3 --no-cpu-baseline --no-stress --no-large-pool > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  echo "$name: $(grep -o 'median [0-9]* timesteps/s' gpurun_out/ab_$name.err | head -1) | $(grep -o 'launches of the timed schedule.*' gpurun_out/ab_$name.err | head -1 | cut -c1-200)"
}
runlarge() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 250 python bench.py --large-pool-only --steps 500 --warmup 100 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  echo "$name: $(grep -o 'median [0-9]* timesteps/s' gpurun_out/ab_$name.err | head -1) | $(grep -o 'launches of the timed schedule.*' gpurun_out/ab_$name.err | head -1 | cut -c1-200)"
}
