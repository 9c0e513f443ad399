# per-kernel times of V(nu,nu) cycles at 16384^2
set -x
NU=${1:-1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nu$NU -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --nu $NU > $R/gpurun_out/prof_nu$NU.log 2>&1
python3 - $R/gpurun_out/prof_nu$NU <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'][:75], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print('%-75s grid %8s n %3d  avg %8.1f us  min %8.1f' % (k[0], k[1], len(v), sum(v) / len(v), min(v)))
PY
