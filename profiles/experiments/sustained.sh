#!/bin/bash
# Sustained-load run with a rocm-smi trace (the script behind profiles/rNN_sustained/; on the GPU box, through gpurun):
#   profiles/experiments/sustained.sh OUTDIR [steps] [extra bench.py flags, e.g. "--gain 2.4"]
# bench.py for `steps` timed steps (default 2000 = ~45 s) while rocm-smi samples clocks / power / temperature once per
# second in the background; writes gpurun_out/OUTDIR/{sustained.json, smi_trace.txt}.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; STEPS=${2:-2000}; XF=${3:-}
mkdir -p $OUT
( while true; do date +%s.%N; rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|Power|junction|GPU use"; sleep 1; done ) > $OUT/smi_trace.txt &
SMI=$!
python $R/bench.py --steps $STEPS --warmup 5 --no-other-configs --no-cpu-baseline $XF > $OUT/sustained.json 2> $OUT/sustained.err
kill $SMI
python - $OUT/sustained.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print({k: j[k] for k in ("value", "ms_per_step", "steps")}, j.get("sclk"))
PY
