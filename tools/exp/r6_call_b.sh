set -u
export TMPDIR=/tmp
O=gpurun_out/r6
mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  env "$@" FYC_BENCH_SHAPES=$O/shapes_env_$tag.txt timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gpu-reference --no-parity --no-vae > $O/bench_env_$tag.json 2> $O/bench_env_$tag.err
  python - $O/bench_env_$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    fam={k:v["ms_per_ddim_step"] for k,v in d.get("kernel_families",{}).items()}
    print(sys.argv[2], d["value"], "frames/s", round(d["ms_per_step"]/25,2), "ms/ddim-step", fam)
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run default_1 FYC_X=0
run fuse_rows_1 FYC_FUSE_ROWS=1
run panel_all_1 FYC_PANEL_ALL=1
run default_2 FYC_X=0
run fuse_rows_2 FYC_FUSE_ROWS=1
run panel_all_2 FYC_PANEL_ALL=1
run graph FYC_X=0
