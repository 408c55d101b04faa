# Ablations of k_sweep_map (timing only, results are wrong by construction): what the kernel
# costs with its gathers served by L1 / L2, without the projection arithmetic, without the
# mapping.  usage on the GPU box: bash tools/exp_sweep.sh
cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
for v in "base:" "gathers_from_16KB_window:-DRN_EXP_SWEEP_WINDOW=0x3fff" "gathers_from_2MB_window:-DRN_EXP_SWEEP_WINDOW=0x1fffff" \
         "no_projection:-DRN_EXP_SWEEP_NOPROJ" "no_mapping:-DRN_EXP_SWEEP_NOMAP" \
         "no_projection_16KB:-DRN_EXP_SWEEP_NOPROJ -DRN_EXP_SWEEP_WINDOW=0x3fff" \
         "no_projection_no_mapping_16KB:-DRN_EXP_SWEEP_NOPROJ -DRN_EXP_SWEEP_NOMAP -DRN_EXP_SWEEP_WINDOW=0x3fff"; do
  bash tools/ab_flags.sh "${v%%:*}" "${v#*:}"
done
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
