cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "uniform_r:" "fast_occ_exp:-DRN_FAST_OCC_EXP" "uniform_r:" "fast_occ_exp:-DRN_FAST_OCC_EXP"; do
  bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
# parity with the fast exponential in place
timeout 900 python -m pytest tests/test_saturated_golden.py tests/test_hip_parity_gpu.py tests/test_forward_pass_gpu.py tests/test_mrf_backward_gpu.py -q -m gpu 2>&1 | tail -8
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
