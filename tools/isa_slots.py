"""Vector-issue slots of a kernel in hipcc's -S output, per barrier-delimited segment (round 4's issue model: a SIMD issues
one vector-type instruction per ~4 cycles whatever the number of waves; a transcendental or an MFMA issue takes two slots).
Loops are weighted by a trip count given on the command line: label=count.
usage: python tools/isa_slots.py file.s kernel_substring [.LBB2_37=8 ...]"""
import re
import sys
from collections import Counter

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def slots(op):
    if op.startswith("v_mfma"):
        return 2.0
    if op.startswith(TRANS):
        return 2.0
    if op.startswith("v_"):
        return 1.0
    return 0.0


def main(path, key, trips):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + key + r"\w*:", l)][0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    b = [l.split(";")[0].strip() for l in lines[start:end]]
    b = [l for l in b if l and (not l.startswith(".") or l.endswith(":"))]
    labels = {l[:-1]: i for i, l in enumerate(b) if l.endswith(":")}
    weight = [1.0] * len(b)
    for i, l in enumerate(b):
        if l.startswith("s_cbranch") or l.startswith("s_branch"):
            t = l.split()[-1]
            if t in labels and labels[t] < i and t in trips:
                for j in range(labels[t], i + 1):
                    weight[j] = max(weight[j], trips[t])
    seg, cur, tot = [], Counter(), Counter()
    for l, w in zip(b, weight):
        if l.endswith(":"):
            continue
        op = l.split()[0]
        cls = ("mfma" if op.startswith("v_mfma") else "trans" if op.startswith(TRANS) else "valu" if op.startswith("v_")
               else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "other")
        cur[cls] += w
        cur["slots"] += w * slots(op)
        if op == "s_barrier":
            seg.append(cur)
            cur = Counter()
    seg.append(cur)
    for i, c in enumerate(seg):
        tot.update(c)
        print(f"seg {i:2d}: slots {c['slots']:7.0f}  valu {c['valu']:6.0f} trans {c['trans']:5.0f} mfma {c['mfma']:5.0f} "
              f"lds {c['lds']:5.0f} vmem {c['vmem']:4.0f}")
    print(f"total : slots {tot['slots']:7.0f}  valu {tot['valu']:6.0f} trans {tot['trans']:5.0f} mfma {tot['mfma']:5.0f} "
          f"lds {tot['lds']:5.0f} vmem {tot['vmem']:4.0f}   (x4 cycles = {4 * tot['slots']:.0f} per wave)")


if __name__ == "__main__":
    tr = {}
    for a in sys.argv[3:]:
        k, v = a.split("=")
        tr[k] = float(v)
    main(sys.argv[1], sys.argv[2], tr)
