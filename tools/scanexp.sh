for cfg in "0 2048" "1 2048" "1 1024" "1 768" "1 512" "0 1024" "0 768"; do
  set -- $cfg
  echo "LARGE=$1 BLOCKS=$2"
  BITHTM_SCAN_LARGE=$1 BITHTM_SCAN_BLOCKS=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --reps 3 2>&1 >/dev/null | grep -E "median|timed schedule"
done
