#!/bin/bash
# Round-3 profile collection on the MI355X box (run from the repo root through gpurun); summaries land in gpurun_out/r03/.
# rocprofv3 rules of this pool: --pmc passes carry no trace domains; the profiled program is python3 itself (no env / shell hop).
R=$PWD
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. headline bench: kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 12 > $O/bench_trace.log 2>&1
# 2. throughput-mode gallery encode (K4 with row groups + K5), one 1024-video ragged super-batch per iteration
ENC_BATCH=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_fused -- python3 $R/tools/prof_encode.py fused > $O/enc_fused.log 2>&1
# 2b. the resident gallery encode (K4b over the whole bf16 table + K5 over all videos) and the C3 training step (eager, one stream)
ENC_BATCH=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_resident -- python3 $R/tools/prof_encode.py resident > $O/enc_resident.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_bf16 -- python3 $R/tools/bench_train.py --config c3 --prec bf16 --steps 20 --warmup 3 --modes eager > $O/train_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_fp32 -- python3 $R/tools/bench_train.py --config c3 --prec fp32 --steps 20 --warmup 3 --modes eager > $O/train_fp32.log 2>&1
# 3. PMC passes (separate runs)
PASS_A="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
PASS_D="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
for k in tower simpool; do
  if [ $k = tower ]; then CMD="$R/tools/bench_tower.py 1024"; else CMD="$R/tools/bench_simpool.py --iters 4"; fi
  rocprofv3 --pmc $PASS_A --output-format csv -d $O/pmc_${k}_a -- python3 $CMD > $O/pmc_${k}_a.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${k}_b -- python3 $CMD > $O/pmc_${k}_b.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${k}_c -- python3 $CMD > $O/pmc_${k}_c.log 2>&1
  rocprofv3 --pmc $PASS_D --output-format csv -d $O/pmc_${k}_d -- python3 $CMD > $O/pmc_${k}_d.log 2>&1
done
cd $R
python3 tools/kstats.py gpurun_out/r03/bench_trace 8
python3 tools/kstats.py gpurun_out/r03/enc_fused 6
python3 tools/kstats.py gpurun_out/r03/enc_resident 6
python3 tools/kstats.py gpurun_out/r03/train_bf16 12
bash tools/pmc_k4b.sh > gpurun_out/r03/pmc_k4b.log 2>&1
# (bench.py's child process - the register-operand MFMA micro-benchmark - writes a stats file of its own: take the one with the scorer)
for d in bench_trace enc_fused enc_resident train_bf16 train_fp32; do f=$(grep -l "dldkd::" $(find $O/$d -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp $f gpurun_out/r03/${d}_kernel_stats.csv; done
python3 tools/pmc_summary.py tower_seq_kernel 0.66 gpurun_out/r03/pmc_tower_summary.json $O/pmc_tower_a $O/pmc_tower_b $O/pmc_tower_c $O/pmc_tower_d > /dev/null && cat gpurun_out/r03/pmc_tower_summary.json | tail -22
python3 tools/pmc_summary.py simpool_eval16_kernel 19.0 gpurun_out/r03/pmc_simpool_summary.json $O/pmc_simpool_a $O/pmc_simpool_b $O/pmc_simpool_c $O/pmc_simpool_d > /dev/null && cat gpurun_out/r03/pmc_simpool_summary.json | tail -16
# keep only the small summaries (the merge limit is 64 MiB)
find $O -name "*counter_collection.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -size +4M -delete
