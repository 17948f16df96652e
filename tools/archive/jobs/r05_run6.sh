cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
tools/run_ab_libs.sh "cur fwdnoexpand" 2 --workload config3
tools/run_ab_libs.sh "cur fwdnoexpand" 2
cp gpurun_out/ab.log gpurun_out/run6_ab.log
