R=$PWD; O=$R/gpurun_out/r03/k4b; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
PASS_A="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
PASS_D="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
rocprofv3 --pmc $PASS_A --output-format csv -d $O/a -- python3 $R/tools/bench_k4b.py 3072 pmc > $O/a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b -- python3 $R/tools/bench_k4b.py 3072 pmc > $O/b.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c -- python3 $R/tools/bench_k4b.py 3072 pmc > $O/c.log 2>&1
rocprofv3 --pmc $PASS_D --output-format csv -d $O/d -- python3 $R/tools/bench_k4b.py 3072 pmc > $O/d.log 2>&1
cd $R
python3 tools/pmc_summary.py in_proj_rows128b_kernel 1.65 gpurun_out/r03/pmc_k4b_summary.json $O/a $O/b $O/c $O/d | tail -40
find $O -name "*counter_collection.csv" -size +2M -delete
