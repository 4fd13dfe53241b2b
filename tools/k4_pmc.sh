cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export K4_KERNELS=rows128
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_k4/p$i -o out --output-format csv -- python3 $R/tools/bench_inproj.py 3072 400000 > $R/gpurun_out/pmc_k4/log$i.txt 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
agg=collections.defaultdict(list)
for f in glob.glob(R+'/gpurun_out/pmc_k4/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'rows128' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    import statistics
    print(f"{k:36s} n={len(v):3d} median={statistics.median(v):.4g}")
PY
