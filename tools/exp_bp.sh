# what k_bp's time is made of: timing-only builds (wrong results on purpose)
cd $GRAFT_REPO_ROOT
cp raynet_amd/csrc/libraynet_hip.so /tmp/lib_orig.so
bash tools/ab_flags.sh "as is                        " ""
bash tools/ab_flags.sh "no message store             " "-DRN_EXP_BP_NOSTORE"
bash tools/ab_flags.sh "no accumulator gather        " "-DRN_EXP_BP_NOGATHER"
bash tools/ab_flags.sh "no gather, no store          " "-DRN_EXP_BP_NOGATHER -DRN_EXP_BP_NOSTORE"
bash tools/ab_flags.sh "loads + gather + store only  " "-DRN_EXP_BP_NOCOMPUTE"
bash tools/ab_flags.sh "loads + store only           " "-DRN_EXP_BP_NOCOMPUTE -DRN_EXP_BP_NOGATHER"
bash tools/ab_flags.sh "no Sr, no msg read           " "-DRN_EXP_NO_SR -DRN_EXP_NO_MSG"
bash tools/ab_flags.sh "compute only (no rows but vox, no gather, no store)" "-DRN_EXP_NO_SR -DRN_EXP_NO_MSG -DRN_EXP_BP_NOGATHER -DRN_EXP_BP_NOSTORE"
bash tools/ab_flags.sh "as is                        " ""
cp /tmp/lib_orig.so raynet_amd/csrc/libraynet_hip.so
