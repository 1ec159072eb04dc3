cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
RR_AMD_LIBRARY=$PWD/rust_robotics_amd/librust_robotics_amd_timeline.so python tools/plan_timeline.py gpurun_out/r03f/plan_timeline.json 2>&1 | tail -70
