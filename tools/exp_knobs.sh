cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "new_default:" "v4_2:-DRN_SWEEP_V4=2" "v4_2_w5:-DRN_SWEEP_V4=2 -DRN_SWEEP_MIN_WAVES=5" "v4_2_w4:-DRN_SWEEP_V4=2 -DRN_SWEEP_MIN_WAVES=4" "new_default:"; do
  bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
