#!/bin/bash
o=gpurun_out/r05r; mkdir -p $o
run() { env $2 timeout 400 python bench.py --steps $3 --warmup 10 --no-cpu-baseline --no-f32-ref --no-roofline $4 > $o/soak_$1.json 2>$o/soak_$1.err; python -c "
import json
d=json.loads(open('$o/soak_$1.json').read().strip().splitlines()[-1]); p=d['parity']; print('soak $3 $1:', round(d['ms_per_step'],3), p['ok'], round(p['grad_rel_worst_ratio_to_cpu_fp32'],2), {k: (round(v['hip'],6), round(v['cpu_fp32_oracle'],6), v['elements_beyond_1e-3_of_max']) for k,v in p['objective_grad_rel_vs_float64'].items()}, {k.split('.')[1]+'.'+k.split('.')[2]: round(v['hip']/v['cpu_fp32_oracle'],2) for k,v in p['grad_rel_vs_float64'].items()})" || tail -5 $o/soak_$1.err; }
run d20 "A=1" 20
run d1500 "A=1" 1500
run heur1500 "PMF_AUTOTUNE=0" 1500
run wsall1500 "PMF_CONV_WS=1" 1500
run epmf300 "A=1" 300 "--model epmf"
