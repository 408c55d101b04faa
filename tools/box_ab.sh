cd $GRAFT_REPO_ROOT
run() { label=$1; flags=$2; shift 2; env "$@" bash tools/ab_build.sh "$label" "$flags"; }
run "sweep V4=1 (8 lanes/vector) " ""
run "sweep V4=2 (4 lanes/vector) " "-DRN_SWEEP_V4=2"
