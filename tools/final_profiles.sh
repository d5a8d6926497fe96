#!/bin/bash
# GPU box: produce the judged artefacts for profiles/ (round tag = $1): bench JSON, rocprofv3 kernel stats of the
# same command, HBM traffic PMC passes, one-step two-queue timeline.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R && python bench.py > $O/${TAG}_bench_cfg2.json 2>$O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_cfg2_kernel_stats.csv
python3 $R/tools/trace_step.py $(find $O/kt -name "*kernel_trace.csv") > $O/${TAG}_step_timeline.txt 2>&1
rm -rf $O/kt
cd $R && bash tools/pmc_bench.sh > $O/${TAG}_pmc_hbm_traffic.txt 2>&1; cp $R/gpurun_out/pmc_bench/pmc_hbm_traffic.json $O/pmc_hbm_traffic.json
for w in cfg3 cfg5; do timeout -k 10 400 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu > $O/${TAG}_bench_${w}.json 2>/dev/null; done
python tools/gemm_sweep.py > $O/${TAG}_gemm_tile_sweep.txt 2>&1
python tools/chain_bench.py 2>/dev/null > $O/${TAG}_chain_bench.txt
python tools/stage_bench.py 2>/dev/null | grep -v Warn > $O/${TAG}_stage_bench.txt
ls -la $O
