#!/bin/bash
# round 4, final GPU call: the measurement set (tools/collect_profiles.sh), then the whole -m gpu suite and smoke() on the same library
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/collect_profiles.sh > gpurun_out/r04_collect.log 2>&1
tail -5 gpurun_out/r04_collect.log
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/gpu_suite.txt 2>&1; tail -4 gpurun_out/r04/gpu_suite.txt
cp gpurun_out/parity_report.txt gpurun_out/r04/parity_report.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04/smoke.txt 2>&1; tail -2 gpurun_out/r04/smoke.txt
python - <<'PY'
import json
for f in ("bench_final", "bench_graph", "bench_cfg1", "bench_cfg4", "bench_cfg5"):
    try:
        d = json.load(open(f"gpurun_out/r04/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d["metric"], "|", d["config"]["workload"][:12], "| roofline", d.get("roofline", {}).get("achieved"), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("traffic"))
    except Exception as e:
        print(f, "failed", e)
PY
