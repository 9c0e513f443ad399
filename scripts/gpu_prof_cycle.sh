# per-kernel times of a few V-cycles at 16384^2 (kernel trace + stats)
set -x
SM=${1:-wjacobi}
TAG=${2:-p}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${SM} -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --smoother $SM > $R/gpurun_out/prof_${TAG}_${SM}.log 2>&1
tail -2 $R/gpurun_out/prof_${TAG}_${SM}.log
python3 - $R/gpurun_out/prof_${TAG}_${SM} <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'][:75], r['Grid_Size_X'], r['VGPR_Count'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print('%-75s grid %8s vgpr %4s  n %3d  avg %8.1f us  min %8.1f' % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v)))
PY
