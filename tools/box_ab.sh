cd $GRAFT_REPO_ROOT
run() { label=$1; flags=$2; shift 2; env "$@" bash tools/ab_build.sh "$label" "$flags"; }
run "r128 s32 cap4096 " ""
run "r128 s32 cap3072 " "-DRN_BOX_CAP=3072"
run "r128 s32 cap2048 " "-DRN_BOX_CAP=2048"
run "r256 s32 cap6144 " "-DRN_BOX_RAYS=256 -DRN_BOX_CAP=6144"
