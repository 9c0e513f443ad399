set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 200 python scripts/lex_debug.py > gpurun_out/r02g_lexdebug.log 2>&1
echo "lexdebug rc=$?"; cat gpurun_out/r02g_lexdebug.log | cut -c1-1500
timeout -k 10 400 python scripts/tune_cycles.py > gpurun_out/r02g_cycles.log 2>&1
echo "cycles rc=$?"; cat gpurun_out/r02g_cycles.log
