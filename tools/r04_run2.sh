#!/bin/bash
# round 4, GPU call 2: full GPU suite + the driver's bench command (with the new `also` block and quality pair) + quality recipe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $O/validation.txt 2>&1; echo "pytest rc=$?"; tail -8 $O/validation.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2>$O/bench.err ) 2>&1 | grep real; tail -3 $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench_driver_command.json'))
print(d['value'], d['ms_per_step'], d['one_step_at_a_time'])
print(json.dumps(d['also'])); print(json.dumps(d['cpu_baseline'].get('quality')))
print(json.dumps(d['roofline']))"
python3 tools/quality_recipe.py > $O/train_eval_recipe.txt 2>/dev/null; cat $O/train_eval_recipe.txt
