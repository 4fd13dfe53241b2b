cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
for cfg in c3 c5; do
  timeout 300 python tools/bench_train.py --config $cfg --prec bf16 --modes graph > $O/bt_${cfg}_bf16_fused.json 2>$O/bt_err.log || tail -5 $O/bt_err.log
  timeout 300 python tools/bench_train.py --config $cfg --prec bf16 --modes graph --fused-towers 0 --fused-losses 0 > $O/bt_${cfg}_bf16_unfused.json 2>$O/bt_err.log || tail -5 $O/bt_err.log
done
timeout 300 python tools/bench_train.py --config c3 --prec bf16 --modes graph --fused-towers 0 > $O/bt_c3_bf16_fusedlosses_only.json 2>$O/bt_err.log || tail -5 $O/bt_err.log
grep -H "stream_ms_median" $O/bt_*.json
