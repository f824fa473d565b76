#!/bin/bash
# Round-end evidence on one MI355X box: default bench line, rocprofv3 kernel stats of the same command, the two HBM PMC passes, the SQ counter passes.
#   bash tools/final_profile.sh TAG      -> gpurun_out/TAG/{bench.json, stats/, fetch/, write/, *.log, groups.md, timeline.txt}
set -o pipefail
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra > $out/write.log 2>&1 || exit 1
bash tools/pmc_kernel.sh $out/sq k_conv_ring,k_conv_halo,k_conv_mfma,k_conv_rows,k_dgrad2_patch,k_wgrad -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra > $out/sq.log 2>&1
python3 tools/pmc_groups.py $out/fetch/f_counter_collection.csv $out/write/w_counter_collection.csv 3 > $out/groups.md
python3 tools/step_timeline.py $out/stats/run_kernel_trace.csv > $out/timeline.txt
python3 tools/chain_breakdown.py $out/stats/run_kernel_trace.csv 30 > $out/chains.txt
python3 tools/trace_gaps.py $out/stats/run_kernel_trace.csv > $out/gaps.txt 2>&1
# keep what travels back small: the PMC csvs are summarised above
rm -f $out/fetch/*kernel_trace.csv $out/write/*kernel_trace.csv
ls -la $out $out/stats | head -40
