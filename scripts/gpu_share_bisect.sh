# which leg of bench.py's default line inflates the emulated rank share measured after it (3.9 ms against 2.2 fresh)?
R=$GRAFT_REPO_ROOT
cd $R
for sk in "other,mehrstellen,lex,1d,configs" "mehrstellen,lex,1d,configs" "other,lex,1d,configs" "other,mehrstellen,1d,configs" "other,mehrstellen,lex,configs"; do
  python bench.py --no-cpu-baseline --skip $sk > gpurun_out/r03f.json 2>/dev/null
  python - "$sk" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r03f.json").read().strip().splitlines()[-1])
print("skipped:", sys.argv[1], [round(x["ms_per_rank_share"], 3) for x in d["strong_scaling_base"]["rank_share_of_8"]], flush=True)
PY
done
