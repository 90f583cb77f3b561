set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_med
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_med -- python3 $R/tools/kbench.py --tiles 256 --rounds 3 --what medians > $R/gpurun_out/prof_med.log 2>&1
f=$(find $R/gpurun_out/prof_med -name "*kernel_stats.csv" | head -1)
cut -c1-150 $f | head -12
