"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as text: per-kernel count / total / avg."""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot:.1f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel"]
    for n, c, ms, avg, mn, mx, vg, ag, lds in rows:
        n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
        lines.append(f"{ms:10.1f} {100 * ms / tot:6.2f} {c:7d} {avg:9.1f} {mn:9.1f} {mx:9.1f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}  {n[:140]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
