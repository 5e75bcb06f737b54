#!/bin/bash
# usage (GPU box): bash scripts/ab_step.sh [-w workload] [-n pairs] [-s steps] "ENV_A=.. ENV_B=.." "ENV_A=.. ENV_B=.." [...]
# Alternates the given environments (each argument = one environment, a space-separated list of VAR=value) over `pairs`
# rounds of `python bench.py --workload <workload> --no-cpu-baseline --no-secondary` on ONE box and prints ms/step per run plus the
# mean per environment -- the A/B form every step-level number of DESIGN.md 3.6 was taken in (boxes of the pool differ by
# ~1.5 %, runs on one box by ~0.2 %).  Example:
#   bash scripts/ab_step.sh -n 2 "JDET_FUSED_SGD=1" "JDET_FUSED_SGD=0"
set -u
W=s2anet_train; N=2; S=30
while getopts "w:n:s:" o; do
  case $o in w) W=$OPTARG;; n) N=$OPTARG;; s) S=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { echo "no environments given"; exit 2; }
declare -A SUM CNT
for ((i = 0; i < N; i++)); do
  for e in "$@"; do
    ms=$(env $e timeout 900 python bench.py --workload $W --no-cpu-baseline --no-secondary --steps $S --warmup 8 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    echo "[$e] ${ms:-FAILED}"
    if [ -n "${ms:-}" ]; then
      SUM[$e]=$(python -c "print(${SUM[$e]:-0} + $ms)")
      CNT[$e]=$((${CNT[$e]:-0} + 1))
    fi
  done
done
for e in "$@"; do
  [ "${CNT[$e]:-0}" -gt 0 ] && python -c "print('mean [%s] %.3f ms over %d runs' % ('$e', ${SUM[$e]} / ${CNT[$e]}, ${CNT[$e]}))"
done
