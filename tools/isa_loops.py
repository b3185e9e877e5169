#!/usr/bin/env python3
"""Static picture of a kernel's loops from the compiler's assembly (hipcc -S --cuda-device-only):
    tools/isa_loops.py <file.s> <substring of the kernel's mangled name> [min loop size]
Basic blocks and their branch edges -> strongly connected components (a loop with everything nested in it is one component);
per component: blocks, instructions by class, scalar-register spill traffic (v_writelane / v_readlane), barriers, and the same for
the component's blocks that hold no LDS window search (fewer than 8 ds_read_b128: the steady path's blocks)."""
import re
import sys


def kernel_lines(path, name):
    L = open(path).read().split("\n")
    start = next(i for i, l in enumerate(L) if re.match(r"^_Z\S*%s\S*:" % re.escape(name), l))
    end = next(i for i in range(start, len(L)) if ".end_amdhsa_kernel" in L[i] or L[i].startswith(".Lfunc_end"))
    return L[start:end]


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    L = kernel_lines(path, name)
    blocks, cur = [], {"label": "entry", "ins": []}
    for l in L[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    idx = {b["label"]: i for i, b in enumerate(blocks)}
    succ = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        fall = True
        for ins in b["ins"]:
            m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", ins)
            if m:
                succ[i].append(idx[m.group(2)])
                if m.group(1) == "s_branch":
                    fall = False
            if ins.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(blocks):
            succ[i].append(i + 1)
    # Tarjan
    sys.setrecursionlimit(100000)
    index, low, on, st, comps, n = {}, {}, set(), [], [], [0]

    def sc(v):
        index[v] = low[v] = n[0]; n[0] += 1; st.append(v); on.add(v)
        for w in succ[v]:
            if w not in index:
                sc(w); low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            c = []
            while True:
                w = st.pop(); on.discard(w); c.append(w)
                if w == v:
                    break
            comps.append(c)
    for v in range(len(blocks)):
        if v not in index:
            sc(v)

    def stats(bs):
        d = dict(instr=0, valu=0, salu=0, lds=0, vmem=0, wait=0, spill_w=0, spill_r=0, barrier=0, branch=0)
        for i in bs:
            for ins in blocks[i]["ins"]:
                op = ins.split()[0]
                d["instr"] += 1
                if op.startswith("v_writelane"): d["spill_w"] += 1
                elif op.startswith("v_readlane"): d["spill_r"] += 1
                if op.startswith("v_"): d["valu"] += 1
                elif op.startswith("ds_"): d["lds"] += 1
                elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): d["vmem"] += 1
                elif op.startswith("s_waitcnt"): d["wait"] += 1
                elif op.startswith("s_barrier"): d["barrier"] += 1
                elif op.startswith(("s_cbranch", "s_branch")): d["branch"] += 1
                elif op.startswith("s_"): d["salu"] += 1
        return d
    print("kernel: %d blocks, %d instructions" % (len(blocks), sum(len(b["ins"]) for b in blocks)))
    for c in sorted(comps, key=lambda c: min(c)):
        tot = sum(len(blocks[i]["ins"]) for i in c)
        if len(c) < 2 or tot < min_size:
            continue
        calm = [i for i in c if sum(1 for x in blocks[i]["ins"] if x.startswith("ds_read_b128")) < 8]
        print("loop at %s..%s: %d blocks" % (blocks[min(c)]["label"], blocks[max(c)]["label"], len(c)), stats(c))
        print("   blocks without a window search (%d):" % len(calm), stats(calm))


if __name__ == "__main__":
    main()
