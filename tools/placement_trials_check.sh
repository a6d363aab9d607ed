#!/bin/bash
# VERDICT r02 item 2, "done": bench.py --placement-trials 1 against --placement-trials 10 on C3, v1, v2, v4, one box per call:
#   gpurun -- bash tools/placement_trials_check.sh a      -> gpurun_out/r03/placement_trials_a.jsonl
T=${1:-a}
OUT=gpurun_out/r03/placement_trials_$T.jsonl
mkdir -p gpurun_out/r03; : > $OUT
for w in c3 v1 v2 v4; do
  for k in 1 10 1 10; do
    timeout -k 10 300 python bench.py --workload $w --placement-trials $k --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print(json.dumps({'workload': '$w', 'placement_trials': $k, 'us_per_step': round(d['ms_per_step'] * 1e3, 2), 'frac': round(r['frac'], 4),
                  'frac_first_allocation': r.get('frac_first_allocation'), 'frac_untuned_library': r.get('frac_untuned_library'),
                  'launch_hint': d['config']['launch_hint'], 'obs_placement': d['config'].get('obs_placement')}))" >> $OUT || exit 1
  done
done
python - <<EOF
import json, collections
rows = [json.loads(l) for l in open("$OUT")]
by = collections.defaultdict(list)
for r in rows: by[(r["workload"], r["placement_trials"])].append(r["us_per_step"])
for w in ("c3", "v1", "v2", "v4"):
    a, b = min(by[(w, 1)]), min(by[(w, 10)])
    print(w, "trials 1:", by[(w, 1)], "trials 10:", by[(w, 10)], "ratio best/best %.3f" % (a / b))
EOF
