# Ablations of k_scatter_box (timing only, wrong results on purpose).
cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "base:" "no_global_flush:-DRN_EXP_BOX_NOFLUSH" "no_lds_atomics:-DRN_EXP_BOX_NOLDS" "neither:-DRN_EXP_BOX_NOFLUSH -DRN_EXP_BOX_NOLDS" "neither_no_msg_read:-DRN_EXP_BOX_NOFLUSH -DRN_EXP_BOX_NOLDS -DRN_EXP_NO_SCATTER_MSG"; do
  bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
