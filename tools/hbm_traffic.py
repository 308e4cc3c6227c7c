"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) into per-kernel-family HBM bytes per
launch -> profiles/rNN_hbm_traffic.json, which bench.py reports as roofline.traffic when its `lib_sha256` is the library being benched or its
`source_sha256` is the digest of the sources that library was built from (followyourclick_amd/_build.py::source_digest).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/pmc_fetch -o f -- python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/pmc_write -o w -- python bench.py ... (same)
    python tools/hbm_traffic.py OUT/pmc_fetch OUT/pmc_write profiles/r02_hbm_traffic.json

Units (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so it is doubled; WRITE_SIZE is uncalibrated (taken as is)."""
import csv
import glob
import hashlib
import json
import os
import sys

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "followyourclick_amd", "libfyc_hip.so")


def lib_digest() -> str:
    """sha256 of the library the counters were taken on: bench.py reports `traffic` only when it runs the same binary"""
    with open(LIB, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()

FAMILIES = (("fyc_gemm_kernel", "gemm"), ("fyc_attn_kernel", "attention"), ("temporal_block", "temporal_block"), ("tblock_rr", "temporal_block"), ("ff_block", "ff_block"), ("panel_linear", "panel_linear"), ("tattn", "attn_temporal"), ("gn_stats", "gn_stats"),
            ("chan_stats_reduce", "gn_stats"), ("gn_apply", "gn_apply"), ("layernorm", "row_stats"), ("concat", "concat"),
            ("splitk_finish", "gemm_splitk_finish"))


def family(name: str) -> str:
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def load(d: str, counter: str):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            n, s = out.get(fam, (0, 0.0))
            out[fam] = (n + 1, s + float(r["Counter_Value"]) * 1024.0)
    return out


def main(fetch_dir, write_dir, out_path):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    fams = {}
    for fam in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(fam, (0, 0.0))
        nw, sw = wr.get(fam, (0, 0.0))
        n = max(nf, nw, 1)
        fams[fam] = {"launches": n, "fetch_bytes_per_launch": 2.0 * sf / max(nf, 1), "write_bytes_per_launch": sw / max(nw, 1),
                     "hbm_bytes_per_launch": 2.0 * sf / max(nf, 1) + sw / max(nw, 1)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from followyourclick_amd._build import family_digest, source_digest
    doc = {"lib_sha256": lib_digest(), "source_sha256": source_digest(), "gemm_family_source_sha256": family_digest("gemm"), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the DDIM steps of a `bench.py --ddim-steps 2` run at cfg2 (4 steps: 2 of the timed clip + 2 of the host-timing leg; per-launch averages do not depend on the count; weights packing and the "
                     "one-off context projections included in 'other'/'gemm' launch counts); FETCH_SIZE x2 per MI355X_MICROARCH.md gfx950 "
                     "correction; WRITE_SIZE uncalibrated", "families": fams}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in fams.items():
        print(f"{k:14s} n={v['launches']:5d}  fetch {v['fetch_bytes_per_launch']/1e6:9.2f} MB  write {v['write_bytes_per_launch']/1e6:9.2f} MB per launch")


if __name__ == "__main__":
    main(*sys.argv[1:4])
