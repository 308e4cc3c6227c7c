"""Audit of a hipcc -save-temps .s file: per kernel, register / scratch totals and what sits INSIDE the K loop (loop depth >= 2 in
LLVM's block comments): scratch traffic, MFMA / ds_read / DMA / barrier / waitcnt counts.  The GEMM kernels must have no scratch
access at depth 2.   usage: python tools/isa_loop_audit.py file.s [name-filter]"""
import re
import sys


def audit(path, flt=""):
    txt = open(path).read().split("\n")
    kernels, cur = [], None
    for i, l in enumerate(txt):
        m = re.match(r"^(_Z\w+):\s", l)
        if m and "fyc" in m.group(1):
            cur = dict(name=m.group(1), start=i, lines=[])
            kernels.append(cur)
        if cur is not None:
            cur["lines"].append(l)
    for k in kernels:
        if flt and flt not in k["name"]:
            continue
        depth, stats = 0, {}
        vg = sc = None
        for l in k["lines"]:
            m = re.search(r"Depth=(\d)", l)
            if re.match(r"^\.LBB", l):
                depth = 0
            if m:
                depth = int(m.group(1))
            mm = re.search(r"; NumVgprs: (\d+)", l)
            if mm:
                vg = int(mm.group(1))
            mm = re.search(r"; ScratchSize: (\d+)", l)
            if mm:
                sc = int(mm.group(1))
            op = l.strip().split(" ")[0] if l.startswith("\t") else ""
            for key, pat in (("scratch", "scratch_"), ("mfma", "v_mfma"), ("ds_read", "ds_read"), ("dma", "global_load_lds"), ("barrier", "s_barrier"), ("waitcnt", "s_waitcnt")):
                if op.startswith(pat):
                    stats.setdefault(depth, {}).setdefault(key, 0)
                    stats[depth][key] += 1
        print(k["name"][:110])
        print(f"   vgprs {vg} scratch {sc} B")
        for d in sorted(stats):
            print(f"   depth {d}: " + ", ".join(f"{a} {b}" for a, b in sorted(stats[d].items())))


if __name__ == "__main__":
    audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")


def issue_block_check(path, flt=""):
    """the per-K-tile DMA issue block and the MFMA / fragment-read phases of the K loop must not touch scratch: a reload is a
    vmcnt(0) between two DMAs.  Prints, per kernel, the scratch instructions between the first and the last DMA of the issue block
    that sits in front of the first MFMA cluster, and those between the first and last MFMA."""
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w+:\s", l) and "fyc" in l]
    for n, st in enumerate(starts):
        en = starts[n + 1] if n + 1 < len(starts) else len(txt)
        name = txt[st].split(":")[0]
        if flt and flt not in name:
            continue
        k = txt[st:en]
        mf = [i for i, l in enumerate(k) if "v_mfma" in l]
        if not mf:
            continue
        dm = [i for i, l in enumerate(k) if "global_load_lds" in l and i < mf[0]]
        # the cluster closest to the first MFMA
        cl = [dm[-1]]
        for i in reversed(dm[:-1]):
            if cl[0] - i < 120:
                cl.insert(0, i)
            else:
                break
        bad_issue = [i for i in range(cl[0], cl[-1]) if "scratch_" in k[i]]
        bad_mma = [i for i in range(mf[0], mf[-1]) if "scratch_" in k[i]]
        print(f"{name[:100]}\n   issue block: {len(cl)} DMAs, {len(bad_issue)} scratch ops inside; MFMA span: {len(mf)} MFMAs, {len(bad_mma)} scratch ops inside")


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "issue":
    issue_block_check(sys.argv[1], sys.argv[2])
