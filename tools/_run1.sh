cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/final/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value']); e=d['extras']; print(e['k4_in_proj_roofline']); print({k:e[k] for k in e if 'videos_per_s' in k}); print(e.get('eval_epoch_gpu_stages_fp32')); print({k:e[k] for k in e if k.endswith('_ms') or k.endswith('ms_bf16') or k.endswith('ms_fp32')})"
