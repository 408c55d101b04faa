cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "base:" "no_sr:-DRN_EXP_NO_SR" "no_msg:-DRN_EXP_NO_MSG" "no_scatter_msg:-DRN_EXP_NO_SCATTER_MSG"; do
  bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
