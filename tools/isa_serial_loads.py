#!/usr/bin/env python3
"""Find memory round trips the compiler put in a row: in the device assembly of a kernel
(hipcc -S --cuda-device-only ...), a load whose value is waited for (s_waitcnt vmcnt(0))
before the next load goes out.  Two or more of those close together are a chain of dependent
round trips to L2 / HBM that the source did not ask for -- conditional loads sunk under the
select that consumes them, loads parked in scratch one at a time, load / store / load / store
copies.  Round 6 found three in the bsts round kernel (10 us of a 103 us round).

usage: isa_serial_loads.py file.s [min_chain=2]
(compile with -gline-tables-only and the chains come with the source lines of their loads)
"""
import bisect
import re
import sys


def main():
    path = sys.argv[1]
    min_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    lines = open(path).read().split("\n")
    kern_at = [i for i, l in enumerate(lines) if l.strip().startswith(".amdhsa_kernel")]
    kern_name = [lines[i].split()[-1] for i in kern_at]
    code = [(i, l.strip()) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith(";")]
    is_load = lambda t: t.startswith(("global_load", "buffer_load", "scratch_load", "flat_load"))
    ev = []
    for n, (i, t) in enumerate(code):
        if not is_load(t):
            continue
        for m in range(n + 1, min(n + 8, len(code))):
            tt = code[m][1]
            if is_load(tt):
                break
            if tt.startswith("s_waitcnt") and "vmcnt(0)" in tt:
                ev.append(i)
                break
    clusters = []
    for e in ev:
        if clusters and e - clusters[-1][-1] < 40:
            clusters[-1].append(e)
        else:
            clusters.append([e])
    depth_re = re.compile(r"Depth=(\d+)")
    # .loc directives: the source line in force at every line of the assembly
    files = {}
    loc_at = []
    cur = None
    file_re = re.compile(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"')
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(".file"):
            m = file_re.search(t)
            if m:
                files[int(m.group(1))] = m.group(2)
        elif t.startswith(".loc"):
            f = t.split()
            cur = (int(f[1]), int(f[2]))
        loc_at.append(cur)
    for c in clusters:
        if len(c) < min_chain:
            continue
        k = bisect.bisect_left(kern_at, c[0])
        name = kern_name[k] if k < len(kern_name) else "?"
        # loop depth: the nearest preceding label's comment
        depth = 0
        for j in range(c[0], max(c[0] - 400, 0), -1):
            if lines[j].startswith(".LBB"):
                m = depth_re.search(lines[j])
                depth = int(m.group(1)) if m else 0
                break
        kinds = sorted(set(lines[e].split()[0] for e in c))
        src = sorted(set("%s:%d" % (files.get(loc_at[e][0], "?"), loc_at[e][1]) for e in c if loc_at[e]))
        print("%-70s lines %d-%d  chain %d  loop depth %d  %s  %s" % (name[-70:], c[0] + 1, c[-1] + 1, len(c), depth, ",".join(kinds), " ".join(src[:6])))


if __name__ == "__main__":
    main()
