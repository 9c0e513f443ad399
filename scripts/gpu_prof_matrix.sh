set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_matrix -- python3 $R/scripts/prof_matrix.py > $R/gpurun_out/prof_matrix.log 2>&1
python3 - $R/gpurun_out/prof_matrix <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 14: print('%-90s calls %5s total_ms %9.2f avg_us %9.1f  %s%%' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
