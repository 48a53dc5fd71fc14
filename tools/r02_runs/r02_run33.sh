#!/bin/bash
# full GPU suite + default bench after the tiled finite-difference kernels replaced the row-streaming ones
set -u
export TMPDIR=/tmp
O=gpurun_out/r02_run33
mkdir -p $O
( python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_run33/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["sim_steps_per_sec"], d["parity"]["bit_exact"])
print({a: (round(b["us"], 1), round(b.get("frac_of_hbm_peak", 0), 3)) for a, b in d["sim_step_per_operator"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
