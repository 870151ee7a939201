#!/bin/bash
# A/B of library builds on ONE box, alternating runs (box-to-box spread is larger than most kernel changes):
#   tools/ab_bench.sh <rounds> <steps> libA.so libB.so ...      (lab builds live in tools/lab/lib/; libffcnn_hip.so is the product in ffcnn_amd/lib/)
# BENCH_ARGS (env or attached as lib.so:BENCH_ARGS=--input=f32) are passed on to bench.py;
# prints frames/s of every run and the per-library median; env for a run can be attached as "lib.so:VAR=1,VAR2=3"
cd "$(dirname "$0")/.." || exit 1
rounds=$1; steps=$2; shift 2
declare -A vals
for ((r = 0; r < rounds; r++)); do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    [[ "$spec" == *:* ]] && envs=${spec#*:}
    bargs=$BENCH_ARGS; evars=""
    for kv in ${envs//,/ }; do
      if [[ "$kv" == BENCH_ARGS=* ]]; then bargs="$bargs ${kv#BENCH_ARGS=}"; else evars="$evars $kv"; fi
    done
    lp=$PWD/tools/lab/lib/$lib; [ -f "$lp" ] || lp=$PWD/ffcnn_amd/lib/$lib
    v=$(env $evars FFCNN_HIP_LIB=$lp python bench.py --steps $steps --warmup 40 --no-cpu-baseline --no-kernel-roofline --no-extras --no-node-line $bargs 2>/dev/null |
        python -c 'import sys, json; d = json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["value"], d["config"]["boxes_match_reference_golden_frame0"])')
    echo "round $r  $spec  $v"
    vals[$spec]+="${v%% *} "
  done
done
for spec in "$@"; do
  python -c 'import sys, statistics as st; v = [float(x) for x in sys.argv[2:]]; print("%-60s median %.0f  min %.0f  max %.0f frames/s (%d runs)" % (sys.argv[1], st.median(v), min(v), max(v), len(v)))' "$spec" ${vals[$spec]}
done
