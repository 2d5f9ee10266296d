run() { env $1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-ref --no-roofline --no-parity $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 | $2 |', round(d['value'],2), round(d['ms_per_step'],3))"; }
run "A=0" ""
run "PMF_LANE_PRIO=14" ""
run "PMF_LANE_PRIO=12" ""
run "PMF_LANE_PRIO=2" ""
run "GPU_MAX_HW_QUEUES=8" ""
run "GPU_MAX_HW_QUEUES=2" ""
run "A=0" ""
