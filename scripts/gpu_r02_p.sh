set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lex512 -o lex512 -- python3 scripts/prof_lex_small.py > gpurun_out/r02p_prof.log 2>&1
echo "prof rc=$?"; tail -3 gpurun_out/r02p_prof.log
find gpurun_out/prof_lex512 -name "*kernel_stats*" | head
