cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "cap4096:" "cap3072:-DRN_BOX0_CAP=3072" "cap2560:-DRN_BOX0_CAP=2560" "cap2048:-DRN_BOX0_CAP=2048" "cap5120:-DRN_BOX0_CAP=5120" "cap4096:"; do
  RAYNET_HIP_BOX_PIN=1 bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
