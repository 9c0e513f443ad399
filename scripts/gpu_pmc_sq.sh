# SQ counters of the fused kernels in a few V-cycles (one pass, 8 SQ slots)
set -x
SM=${1:-wjacobi}
TAG=${2:-sq}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${PMC_LIST:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES} --output-format csv -d $R/gpurun_out/pmc_${TAG}_${SM} -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --smoother $SM > $R/gpurun_out/pmc_${TAG}_${SM}.log 2>&1
tail -3 $R/gpurun_out/pmc_${TAG}_${SM}.log
python3 - $R/gpurun_out/pmc_${TAG}_${SM} <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
f = glob.glob(root + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in f:
    for row in csv.DictReader(open(path)):
        k = (row['Kernel_Name'][:70], row.get('Grid_Size'))
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[(k, row['Counter_Name'])] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:14]:
    n = cnt[(k, 'SQ_WAVE_CYCLES')] or 1
    print(k, 'launches', n, {a: round(b / n) for a, b in c.items()})
PY
