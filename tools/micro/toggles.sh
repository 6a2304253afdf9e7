for t in "SE_DW_BWD_FUSED=0" "SE_LNBWD_FUSED=0" "SE_FF_FUSED=0" "SE_DIFF_GATE_FUSED=0 SE_DIFF_ONE_STREAM=0" "SE_GATE_PROJ_SPLIT=1"; do
  echo "== $t"
  env $t timeout -k 10 500 python -m pytest tests/test_model_gpu.py tests/test_diffuse.py -x -q -k "conformer or tscnet_forward or train_step_vs_reference_loop or diffuse_forward or reverse_sampler" 2>&1 | tail -1
done
