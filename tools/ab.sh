#!/bin/bash
# Same-box A/B of ONE switch, alternating values (box-to-box spread is 3-5 %: only same-box, interleaved runs compare).
#
#   bash tools/ab.sh make <SWITCH> "<v0 v1 v0 v1>" <bench> [kernel-regex]     compile-time switch of csrc/Makefile (K1_ROT, K1_NT,
#                                                                              FIN_SWZ, TW_DEPTH, TW_FOLD_LN2, TW_SPREAD, H16_BF16,
#                                                                              DG_ABLATE, TT_ABL, TT_WPB, LN_ROWS, PHILOX_ROUNDS, DIAG)
#   bash tools/ab.sh env  <VAR>    "<v0 v1 v0 v1>" <bench> [kernel-regex]     environment switch of the library / host package
#                                                                              (DLDKD_LN_DUAL, DLDKD_TN_NST, DLDKD_TN_TARGET,
#                                                                              DLDKD_TOWER_PREPACK, DLDKD_TOWER_LN_SUMS, DLDKD_H0_H16 ...)
#   <bench>: simpool | simpool-pmc | finish | encode | tower | k4b | train-c3 | train-c5 | train-c3-mixed | "<any command>"
#   kernel-regex (train-* only): also print the rocprofv3 --kernel-trace --stats averages of the matching kernels (eager step)
#
# Replaces the 60 one-off tools/r04_*.sh / r05_*.sh scripts of rounds 4-5 (git history has them; tools/README.md maps each table
# under profiles/ to the ab.sh line that reproduces it).  Leaves the shipped build in place.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KIND=$1; SW=$2; VALS=$3; BENCH=$4; PAT=${5:-}
C=$R/dl-dkd_amd/csrc
O=$R/gpurun_out/ab_$SW; mkdir -p $O
objs_of() {   # the objects whose Makefile rule (or DIAGFLAGS: all of them) reads the switch
  if grep -q "DIAGFLAGS.*\$(if \$($1)" $C/Makefile; then ls $C/build/*.o 2>/dev/null; else grep -E "^build/[a-z0-9_]+\.o:.*\\\$\(if \\\$\($1\)" $C/Makefile | cut -d: -f1 | sed "s|^|$C/|"; fi
}
js() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('config',''), {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict) and 'stream_ms_median' in v})"; }
run_bench() {
  case "$BENCH" in
    simpool)  python3 $R/tools/bench_simpool.py --iters 10 2>/dev/null | tail -2 ;;
    simpool-pmc)   # kernel time + HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, one --pmc pass each: no trace domains beside counters)
      python3 $R/tools/bench_simpool.py --iters 10 2>/dev/null | tail -2
      ( cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/bench_simpool.py --iters 3 > $O/f.log 2>&1
        TMPDIR=/tmp rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/bench_simpool.py --iters 3 > $O/w.log 2>&1 )
      python3 $R/tools/pmc_summary.py simpool_eval16p_kernel 17.0 $O/s.json $O/f $O/w > /dev/null 2>&1
      python3 -c "import json; d=json.load(open('$O/s.json'))['derived']; print({k: round(v,3) for k,v in d.items() if 'hbm' in k})"
      rm -rf $O/f $O/w ;;
    finish)   python3 $R/tools/bench_finish.py 2>/dev/null | tail -2 ;;
    encode)   ENC_BATCH=1024 python3 $R/tools/prof_encode.py resident 2>/dev/null | tail -1 ;;
    tower)    python3 $R/tools/bench_tower.py 1024 2>/dev/null | tail -1 ;;
    k4b)      python3 $R/tools/bench_k4b.py 3072 time 2>/dev/null | tail -1 ;;
    train-c3|train-c5|train-c3-mixed)
      cfg=c3; prec=bf16; [ "$BENCH" = train-c5 ] && cfg=c5; [ "$BENCH" = train-c3-mixed ] && prec=mixed
      if [ -n "$PAT" ]; then
        ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/bench_train.py --config $cfg --prec $prec --steps 10 --warmup 2 --modes eager > $O/p.log 2>&1 )
        python3 $R/tools/kstats.py $O/p 70 | grep -E "$PAT" | cut -c1-150; rm -rf $O/p
      fi
      python3 $R/tools/bench_train.py --config $cfg --prec $prec --steps 30 --warmup 8 --modes graph 2>/dev/null | js ;;
    *) bash -c "$BENCH" 2>&1 | tail -3 ;;
  esac
}
for v in $VALS; do
  echo "== $SW=$v"
  if [ "$KIND" = make ]; then
    rm -f $(objs_of $SW)
    make -C $C -j8 $SW=$v > $O/make.log 2>&1 || { echo "build failed"; tail -5 $O/make.log; exit 1; }
  else
    export $SW=$v
  fi
  run_bench
done
if [ "$KIND" = make ]; then rm -f $(objs_of $SW); make -C $C -j8 > $O/make.log 2>&1; fi
