#!/bin/bash
# round 3, GPU call 33: kernel trace of cfg4 (cavity, 3 x 5x128) and cfg5 on the final build
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r03zi
mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof4 -o p -- python $R/tools/bench_configs.py cfg4 > $R/$O/cfg4.txt 2> $R/$O/cfg4.err)
python profiles/rocpd_stats.py $O/prof4/p_results.db > $O/cfg4_kernel_stats.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof5 -o p -- python $R/tools/bench_configs.py cfg5 > $R/$O/cfg5.txt 2> $R/$O/cfg5.err)
python profiles/rocpd_stats.py $O/prof5/p_results.db > $O/cfg5_kernel_stats.txt 2>&1
find $O -name "*.db" -delete
cut -c1-200 $O/cfg4_kernel_stats.txt | head -30
