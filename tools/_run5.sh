mkdir -p gpurun_out/r04e
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-f32-ref --no-parity"
export PMF_TUNE_CACHE=/tmp/tune.txt
$B > /dev/null 2>&1
for cfg in "0,0,0" "4,4,4" "6,4,4" "4,2,2" "6,6,6" "8,4,4" "4,6,6" "2,4,4" "0,0,0"; do
  PMF_LANE_CUS=$cfg $B 2> gpurun_out/r04e/err_$cfg.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],2), round(d['ms_per_step'],3))" >> gpurun_out/r04e/cus.txt
done
cat gpurun_out/r04e/cus.txt
