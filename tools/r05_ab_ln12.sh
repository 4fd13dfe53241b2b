# layernorm_kernel<12> (110 VGPRs, 4 waves per SIMD) against <16> (142, 3 waves) for 3072-wide rows: tools/bench_ln_dropout.py on both builds
R=$GRAFT_REPO_ROOT
for v in 12 16 12 16; do
  if [ $v = 16 ]; then sed -i 's/else if (nv <= 12) DLDKD_LAUNCH(layernorm_kernel<12>/else if (nv <= 0) DLDKD_LAUNCH(layernorm_kernel<12>/' $R/dl-dkd_amd/csrc/encoder_f32.hip
  else sed -i 's/else if (nv <= 0) DLDKD_LAUNCH(layernorm_kernel<12>/else if (nv <= 12) DLDKD_LAUNCH(layernorm_kernel<12>/' $R/dl-dkd_amd/csrc/encoder_f32.hip; fi
  make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
  echo "== MAXV=$v"; python3 $R/tools/bench_ln_dropout.py 2>&1 | grep -v amdgpu.ids
done
sed -i 's/else if (nv <= 0) DLDKD_LAUNCH(layernorm_kernel<12>/else if (nv <= 12) DLDKD_LAUNCH(layernorm_kernel<12>/' $R/dl-dkd_amd/csrc/encoder_f32.hip
make -C $R/dl-dkd_amd/csrc > /dev/null 2>&1
