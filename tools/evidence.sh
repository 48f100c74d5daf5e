#!/bin/bash
# Developer tool: everything under profiles/<round>/ from the binary in the tree, in ONE gpurun call
# (the boxes of the pool differ; figures that are compared must come from the same box).
#   tools/evidence.sh r03     -> gpurun_out/evidence_r03/ (copy what is to be judged into profiles/r03/)
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/evidence_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, command...
  local name=$1; shift
  rm -rf "$OUT/prof_$name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -- "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$name.csv"
}
# 0. a fresh box's first GPU process pays for cold caches on the HOST side (the first run's ticks read 2-3 % slower
# than the second's, the kernels do not): one short throw-away run first
python3 $ROOT/bench.py --steps 5 --warmup 2 --no-other-configs --no-cpu-baseline > /dev/null 2>&1
# 1. the bench line, un-profiled, then the same command under the kernel trace; then the driver's own command line
python3 $ROOT/bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.stderr"
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_command.json" 2> /dev/null
stats bench_2097152x64 python3 $ROOT/bench.py --no-cpu-baseline --no-other-configs
tail -1 "$OUT/bench_2097152x64.stdout" > "$OUT/bench_under_rocprof.json"
# 2. the other configurations' scoring passes
stats cfg1_65536x64 python3 $ROOT/tools/config_ticks.py 65536 64 200 400
stats cfg2_262144x128_map2000 python3 $ROOT/tools/config_ticks.py 262144 128 2000 200
stats share_262144x64 python3 $ROOT/tools/config_ticks.py 262144 64 200 400
stats regenerate_2097152x64 python3 $ROOT/tools/regen_tick.py 2097152 64 60
# 3. HBM traffic of the headline's scoring pass: separate PMC passes, no trace domains
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/pmc_$C"
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -- python3 $ROOT/tools/traffic.py 2097152 64 200 > "$OUT/pmc_$C.stdout" 2>&1
  find "$OUT/pmc_$C" -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} "$OUT/pmc_${C}_2097152x64.csv"
done
python3 $ROOT/tools/make_traffic_json.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" 2097152 64 200x200 "$OUT/traffic_2097152x64.json" > /dev/null
# 3b. the VALU budget of the headline's lane pass and of the split pass (one PMC pass each, no trace domains)
for SPEC in "lane_2097152x64 2097152 64" "split_32768x64 32768 64" "lane_65536x64 65536 64"; do
  set -- $SPEC
  rm -rf "$OUT/pmc_sq"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq" -- python3 $ROOT/tools/config_ticks.py $2 $3 200 40 > "$OUT/pmc_sq_$1.stdout" 2>&1
  python3 $ROOT/tools/pmc_summary.py "$OUT/pmc_sq" $2 $3 > "$OUT/pmc_SQ_$1.txt" 2>&1
done
rm -rf "$OUT/pmc_sq"
# 3c. who issues the ticks, and the split pass against the wave pass (same call, same box)
python3 $ROOT/tools/caller_ab.py 2000x56 65536x64 262144x64 2097152x64 > "$OUT/ab_caller_python_vs_compiled.txt" 2>&1
python3 $ROOT/tools/tail_ab.py SMPC_NO_SPLIT=1 > "$OUT/ab_split_pass_by_batch.txt" 2>&1
TAILAB_SIZES=12288x56,16384x56,20000x60,32768x56,32768x48 python3 $ROOT/tools/tail_ab.py SMPC_NO_SPLIT=1 > "$OUT/ab_split_pass_short_horizons.txt" 2>&1
# 3d. the N > 1 code path of bench.py with two processes on this one GPU (RCCL refuses that: the torch/gloo driver
# carries the headline, the mailbox exchange beside it), then with the stand-in for the nccl* symbols
( export SMPC_BENCH_SHARE_GPU=1 SMPC_BENCH_BACKEND=gloo
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $ROOT/bench.py --gpus 2 --steps 20 --warmup 5 --total-rollouts 131072 > "$OUT/bench_2ranks_one_gpu_rehearsal.json" 2> "$OUT/bench_2ranks.stderr"
  SMPC_RCCL_LIB=$ROOT/tests/fake_rccl/libfake_rccl.so python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 $ROOT/bench.py --gpus 2 --steps 20 --warmup 5 --total-rollouts 131072 > "$OUT/bench_2ranks_one_gpu_standin_rehearsal.json" 2> "$OUT/bench_2ranks_standin.stderr" )
# 3e. ... and with the real RCCL in a one-rank communicator
SMPC_BENCH_FORCE_DIST=1 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --total-rollouts 262144 --no-other-configs --no-cpu-baseline > "$OUT/bench_1rank_rccl_rehearsal.json" 2> /dev/null
# 4. where a tick's fixed cost goes
: > "$OUT/tick_timeline.txt"
for BT in "2000 56" "65536 64" "262144 64"; do
  rm -rf "$OUT/tl"
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tl" -- python3 $ROOT/tools/tick_timeline.py run $BT 400 > /dev/null 2>&1
  echo "== tick timeline $BT" >> "$OUT/tick_timeline.txt"
  python3 $ROOT/tools/tick_timeline.py report "$OUT/tl" >> "$OUT/tick_timeline.txt"
done
rm -rf "$OUT/tl" "$OUT"/prof_* "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
ls -la "$OUT"
