#!/bin/bash
# GPU box: produce the judged artefacts for profiles/ (round tag = $1) on ONE build of the PRODUCT library: HBM traffic PMC
# passes per workload and in-graph kernel statistics (both stamped with the library's build id, both read back by bench.py),
# then the bench lines, rocprofv3 kernel stats of the same command, matrix-pipe PMC, the other workloads, the training step.
# Two calls (a GPU call is limited to 20 minutes): `final_profiles.sh r03 a` = the PMC / in-graph passes; copy
# gpurun_out/final/{pmc_hbm_traffic,graph_kernel_stats}.json into profiles/ and run `final_profiles.sh r03 b` = the bench lines.
TAG=${1:-r04}; PART=${2:-ab}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R
unset AVSEP_LIB
if [[ $PART == *a* ]]; then
# ---- 1. HBM-side traffic (bench.py fills roofline.traffic from it only when the build id matches the loaded library)
rm -rf $R/gpurun_out/pmc_bench
bash tools/pmc_bench.sh cfg2 20 5 > $O/${TAG}_pmc_hbm_traffic_cfg2.txt 2>&1
bash tools/pmc_bench.sh cfg3 3 1 > $O/${TAG}_pmc_hbm_traffic_cfg3.txt 2>&1
bash tools/pmc_bench.sh cfg5 2 1 > $O/${TAG}_pmc_hbm_traffic_cfg5.txt 2>&1
cp $R/gpurun_out/pmc_bench/pmc_hbm_traffic.json $O/pmc_hbm_traffic.json; cp $O/pmc_hbm_traffic.json $R/profiles/pmc_hbm_traffic.json
echo "pmc traffic done"
# ---- 2. the timed region alone under rocprofv3 (graph replays only): per-kernel durations INSIDE the step
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu --no-profile --steps 100 --warmup 5 --rounds 1 > /dev/null 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_cfg2_graph_only_kernel_stats.csv; rm -rf $O/kt
# (replays: 105 of the two-in-flight leg + 105 of the one-step-at-a-time leg + one eager forward of the slot check = 211)
python3 $R/tools/graph_stats.py $O/${TAG}_bench_cfg2_graph_only_kernel_stats.csv cfg2 211 $O/graph_kernel_stats.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload cfg3 --no-cpu --no-profile --steps 10 --warmup 2 --rounds 1 > /dev/null 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_cfg3_graph_only_kernel_stats.csv; rm -rf $O/kt
python3 $R/tools/graph_stats.py $O/${TAG}_bench_cfg3_graph_only_kernel_stats.csv cfg3 25 $O/graph_kernel_stats.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload cfg5 --no-cpu --no-profile --steps 6 --warmup 2 --rounds 1 > /dev/null 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_cfg5_graph_only_kernel_stats.csv; rm -rf $O/kt
# (replays: 2 warm-up + 6 of the two-in-flight leg, 2 + 6 of the one-at-a-time leg, one eager forward of the slot check = 17)
python3 $R/tools/graph_stats.py $O/${TAG}_bench_cfg5_graph_only_kernel_stats.csv cfg5 17 $O/graph_kernel_stats.json
cp $O/graph_kernel_stats.json $R/profiles/graph_kernel_stats.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu --no-profile --inflight 1 --steps 50 --rounds 1 > /dev/null 2>&1
python3 $R/tools/trace_step.py $(find $O/kt -name "*kernel_trace.csv") > $O/${TAG}_step_timeline.txt 2>&1; rm -rf $O/kt
echo "graph stats done"
fi
if [[ $PART == *b* ]]; then
# ---- 3. the bench lines (traffic + in-graph columns now match this build)
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_cfg2_driver_command.json 2>$O/bench.err
python3 bench.py --stream > $O/${TAG}_bench_cfg2.json 2>>$O/bench.err
echo "bench cfg2 done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_bench_cfg2_kernel_stats.csv; rm -rf $O/kt
cd $R
bash tools/pmc_mfma.sh > $O/${TAG}_pmc_mfma_cfg2.txt 2>&1
echo "pmc mfma done"
for w in cfg3 cfg5; do timeout -k 10 500 python3 bench.py --workload $w --steps 20 --warmup 3 --rounds 5 --cpu-seconds 6 > $O/${TAG}_bench_${w}.json 2>/dev/null; echo "bench $w done"; done
timeout -k 10 400 python3 bench.py --mode train --steps 10 --warmup 3 --cpu-seconds 4 > $O/${TAG}_bench_train_cfg4.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --mode train --steps 7 --warmup 2 --rounds 1 --no-cpu > /dev/null 2>&1
cp $(find $O/kt -name "*kernel_stats.csv") $O/${TAG}_train_cfg4_kernel_stats.csv; rm -rf $O/kt
echo "train done"
cd $R
AVSEP_SCHEDULE=fork AVSEP_LIB=dev python3 tools/stamps.py cfg2 2>/dev/null | grep " us " > $O/${TAG}_stage_stamps.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -q > $O/${TAG}_validation.txt 2>&1; tail -2 $O/${TAG}_validation.txt
fi
ls -la $O
