#!/bin/bash
# Round-6 profile collection on the MI355X box (run from the repo root through gpurun); summaries land in gpurun_out/r06p/.
# rocprofv3 rules of this pool: --pmc passes carry no trace domains; the profiled program is python3 itself (no env / shell hop).
# usage: bash tools/collect_profiles_r06.sh [bench] [train] [enc] [pmc]   (default: all)
R=$PWD
O=$R/gpurun_out/r06p
mkdir -p $O
WHAT="${*:-bench train enc pmc k5}"
cd /tmp && export TMPDIR=/tmp
copy_stats() { f=$(ls -t $(grep -l "dldkd::" $(find $O/$1 -name "*kernel_stats.csv")) | head -1); [ -n "$f" ] && cp $f $O/$1_kernel_stats.csv; }
for w in $WHAT; do case $w in
bench)   # headline bench: kernel trace + stats of the contract command (extras and CPU baseline off: the timed region only)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 12 > $O/bench.log 2>&1
  copy_stats bench; python3 $R/tools/kstats.py $O/bench 8 ;;
train)   # C3 / C5 training steps: eager kernel stats (one stream) + the replayed multi-graph step's timeline
  for cfg in c3 c5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_${cfg}_bf16 -- python3 $R/tools/bench_train.py --config $cfg --prec bf16 --steps 20 --warmup 3 --modes eager > $O/train_${cfg}_bf16.log 2>&1
    copy_stats train_${cfg}_bf16
  done
  for pr in fp32 mixed; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_c3_$pr -- python3 $R/tools/bench_train.py --config c3 --prec $pr --steps 20 --warmup 3 --modes eager > $O/train_c3_$pr.log 2>&1
    copy_stats train_c3_$pr
  done
  rocprofv3 --kernel-trace --output-format csv -d $O/graph_c3_mixed -- python3 $R/tools/bench_train.py --config c3 --prec mixed --steps 12 --warmup 3 --modes graph > $O/graph_c3_mixed.log 2>&1
  python3 $R/tools/step_timeline.py $O/graph_c3_mixed 20 > $O/step_timeline_c3_mixed_graph.txt 2>&1
  for pr in bf16 mixed; do python3 $R/tools/graph_timeline.py c3 sync $pr > $O/graph_timeline_c3_$pr.json 2>/dev/null; done
  for cfg in c3 c5; do
    rocprofv3 --kernel-trace --output-format csv -d $O/graph_${cfg}_bf16 -- python3 $R/tools/bench_train.py --config $cfg --prec bf16 --steps 12 --warmup 3 --modes graph > $O/graph_${cfg}_bf16.log 2>&1
    python3 $R/tools/step_timeline.py $O/graph_${cfg}_bf16 20 > $O/step_timeline_${cfg}_bf16_graph.txt 2>&1
    head -1 $O/step_timeline_${cfg}_bf16_graph.txt
  done
  python3 $R/tools/kstats.py $O/train_c3_bf16 45 ;;
enc)     # resident gallery encode (K4b over the bf16 table + K5 over all videos)
  ENC_BATCH=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_resident -- python3 $R/tools/prof_encode.py resident > $O/enc_resident.log 2>&1
  copy_stats enc_resident; python3 $R/tools/kstats.py $O/enc_resident 5 ;;
pmc)     # PMC passes (separate runs, no trace domains) for the scorer and for the SHIPPED gallery encode (resident table: K4b + the fp16-h0
         # persistent K5 - what eval_epoch runs; the r04 tower summary was collected on the padded fp32-h0 variant)
  PASS_A="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
  PASS_D="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
  for k in simpool enc; do
    # (round 6: the scorer's counters are collected on bench.py's OWN timed loop - the driver's command minus extras / CPU baseline)
    if [ $k = enc ]; then CMD="$R/tools/prof_encode.py resident"; export ENC_BATCH=1024; else CMD="$R/bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 1"; export DLDKD_BENCH_NO_MFMA_PROBE=1; fi
    rocprofv3 --pmc $PASS_A --output-format csv -d $O/pmc_${k}_a -- python3 $CMD > $O/pmc_${k}_a.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${k}_b -- python3 $CMD > $O/pmc_${k}_b.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${k}_c -- python3 $CMD > $O/pmc_${k}_c.log 2>&1
    rocprofv3 --pmc $PASS_D --output-format csv -d $O/pmc_${k}_d -- python3 $CMD > $O/pmc_${k}_d.log 2>&1
  done
  python3 $R/tools/pmc_summary.py simpool_eval16p_kernel 17.0 $O/pmc_simpool_summary.json $O/pmc_simpool_a $O/pmc_simpool_b $O/pmc_simpool_c $O/pmc_simpool_d > /dev/null && tail -16 $O/pmc_simpool_summary.json
  python3 $R/tools/pmc_summary.py tower_seq_kernel 8.2 $O/pmc_tower_resident_summary.json $O/pmc_enc_a $O/pmc_enc_b $O/pmc_enc_c $O/pmc_enc_d > /dev/null && tail -14 $O/pmc_tower_resident_summary.json
  python3 $R/tools/pmc_summary.py in_proj_rows128b_kernel 6.2 $O/pmc_k4b_resident_summary.json $O/pmc_enc_a $O/pmc_enc_b $O/pmc_enc_c $O/pmc_enc_d > /dev/null && tail -14 $O/pmc_k4b_resident_summary.json ;;
k5)      # K5's phase stamps (diagnostic build of the same code) and the steady-state projection loop in miniature
  python3 $R/tools/tower_timeline.py h16 > $O/k5_phase_stamps_ragged.json 2>/dev/null
  python3 $R/tools/tower_timeline.py h16 full > $O/k5_phase_stamps_full_length.json 2>/dev/null
  ( cd $R/tools/mb && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result stream_issue.hip -o /tmp/stream_issue.bin && /tmp/stream_issue.bin ) > $O/mb_stream_issue.txt 2>&1
  for v in 1 0 1 0; do echo "DLDKD_SKIP_ZERO_ROWS=$v"; DLDKD_SKIP_ZERO_ROWS=$v ENC_BATCH=1024 python3 $R/tools/prof_encode.py resident 2>/dev/null | tail -1; done > $O/ab_k5_skip_zero_rows.txt ;;
esac; done
cd $R
# keep only the small summaries (the merge limit is 64 MiB)
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
