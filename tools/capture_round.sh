#!/bin/bash
# tools/capture_round.sh <tag>: the evidence of one build, in one gpurun call (run from the repo root on the GPU box):
#   every kernel at its BASELINE shape under HIP events / rocprofv3 --kernel-trace --stats / four --pmc passes
#   (tools/profile_all.sh), the table and the traffic record stamped with the kernel-source hash, the bench command
#   itself under rocprofv3, the default bench line, the two-rank gloo rehearsal of the N > 1 path, per-call latencies.
[ -n "$1" ] || { echo "usage: tools/capture_round.sh <tag>"; exit 2; }
T="$1"; R="$PWD"; O="$R/gpurun_out/$T"
bash tools/profile_all.sh "$T" > "$R/gpurun_out/${T}_capture.log" 2>&1 || { tail -5 "$R/gpurun_out/${T}_capture.log"; exit 1; }
python3 tools/profile_table.py "$O" --md --merge profiles/traffic.json --source "profiles/$T/pmc_summary.csv" > "$O/table.md"
cp profiles/traffic.json "$O/traffic_merged.json"
echo "capture done"
# bench.py may ONLY be profiled with --kernel-trace / --stats: it spawns pool workers and may start hipcc, and with --pmc the
# profiler's preloaded library initialises the GPU first, which turns those spawns into exec-after-GPU-init (refused on the GPU
# boxes).  Counter passes use tools/profile_all.py / s2l_pmc.py / ram_pmc.py, which spawn nothing.
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_trace" -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$O/bench_rocprof_line.json" 2> "$O/bench_rocprof.err" )
find "$O/bench_trace" -name "*kernel_stats.csv" -exec cp {} "$O/bench_rocprof_kernel_stats.csv" \;
rm -rf "$O/bench_trace"
echo "bench under rocprofv3 done"
timeout -k 10 500 python3 bench.py > "$O/bench_line.json" 2> "$O/bench_line.err" || tail -3 "$O/bench_line.err"
echo "bench line done"
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 --c4-windows 16384 > "$O/bench_gpus2_gloo_rehearsal.json" 2> "$O/bench_gpus2.err" || tail -3 "$O/bench_gpus2.err"
echo "rehearsal done"
timeout -k 10 200 python3 tools/call_latency.py > "$O/call_latency.txt" 2>&1
# the screen passes in isolation: time, instruction counts and the radius check (tools/micro/pair_pass_bench.hip)
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../pyperiod_amd/csrc pair_pass_bench.hip -o pair_pass_bench 2> /dev/null ) && rm -f "$O/pair_pass_bench.txt" && bash tools/micro/pair_pass_pmc.sh "gpurun_out/$T/pair_pass_bench.txt" > /dev/null 2>&1
echo "pass test bed done"
cat "$O/table.md"
