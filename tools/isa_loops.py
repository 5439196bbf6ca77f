"""List the loops of a kernel in hipcc's -S output with what matters for a pipelined loop: MFMA count, the s_waitcnt
vmcnt(...) inside it (a vmcnt(0) in a loop that keeps LDS-DMA or register prefetches in flight is a drained pipeline)
and scratch traffic.   usage: python tools/isa_loops.py file.s kernel_name_substring"""
import re
import sys


def main(path, key):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + key + r"\w*:", l)]
    for start in starts:
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
        b = [l.split(";")[0].strip() for l in lines[start:end]]
        b = [l for l in b if l]
        labels = {l[:-1]: i for i, l in enumerate(b) if l.endswith(":")}
        print(lines[start].split(":")[0])
        for i, l in enumerate(b):
            if l.startswith(("s_cbranch", "s_branch")):
                t = l.split()[-1]
                if t in labels and labels[t] < i:
                    body = b[labels[t]:i]
                    nm = sum(1 for x in body if x.startswith("v_mfma"))
                    vm = [x.replace("s_waitcnt ", "") for x in body if x.startswith("s_waitcnt") and "vmcnt" in x]
                    sc = sum(1 for x in body if x.startswith("scratch"))
                    print(f"   {t}: {len(body)} instrs, {nm} mfma, vmcnt waits {vm}, {sc} scratch ops")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
