# gpurun wrapper body: cd to the repo copy, make the scratch directory, run the given command line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a; export TMPDIR=/tmp
eval "$@"
