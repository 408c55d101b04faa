cd $GRAFT_REPO_ROOT
run() { label=$1; flags=$2; shift 2; env "$@" bash tools/ab_build.sh "$label" "$flags"; }
run "bricks        " ""
run "bricks cap3072" "-DRN_BOX_CAP=3072"
