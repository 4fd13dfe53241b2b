# A/B of one environment switch of the library / host package on the training step, same box, alternating:
#   bash tools/r05_ab_env.sh VAR "v0 v1 v0 v1" 'kernel-name regex'
# eager kernel averages under rocprofv3 (C3, bf16 mode) and the replayed C3 / C5 steps without a profiler
R=$GRAFT_REPO_ROOT
VAR=$1; VALS=$2; PAT=$3
O=$R/gpurun_out/r05abenv
mkdir -p $O
for v in $VALS; do
  export $VAR=$v
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$v -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 10 --warmup 2 --modes eager > $O/p_$v.log 2>&1
  cd $R
  echo "== $VAR=$v"; python3 tools/kstats.py gpurun_out/r05abenv/p_$v 70 | grep -E "$PAT" | cut -c1-150
  for c in c3 c5; do python tools/bench_train.py --config $c --prec bf16 --steps 30 --warmup 8 --modes graph 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config'], {k:round(v['stream_ms_median'],3) for k,v in d.items() if isinstance(v,dict)})"; done
  rm -rf $O/p_$v
done
