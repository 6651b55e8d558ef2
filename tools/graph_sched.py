"""Offline look at the captured step graph (MIMRL_GRAPH_DOT dump): which hardware queue does this HIP runtime give every node?
Hypothesis (tools/hw/graph_order.hip): a depth-first walk from the roots in which a node's FIRST outgoing edge keeps the parent's
queue and every further edge takes the next one (mod 4); a node keeps the queue of whoever reaches it first.
usage: python tools/graph_sched.py step_graph.dot [timeline.txt]"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def parse(path):
    txt = open(path).read()
    names = {}
    for m in re.finditer(r'"graph_0_node_(\d+)"\[style[^\n]*\n(\w+)\n\| \{ID \| \d+ \| ([^\\<\}]+)', txt):
        names[int(m.group(1))] = m.group(3).strip()
    for m in re.finditer(r'"graph_0_node_(\d+)"\[style[^\n]*\n(\w+)\n', txt):
        names.setdefault(int(m.group(1)), m.group(2))
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', txt)]
    return names, edges


def assign(names, edges, nq=4):
    out = {n: [] for n in names}
    indeg = {n: 0 for n in names}
    for a, b in edges:
        out[a].append(b)
        indeg[b] += 1
    q = {}

    def walk(n, s):
        if n not in q:
            q[n] = s
        else:
            return
        for c in out[n]:
            walk(c, s)
            s = (s + 1) % nq
    s = 0
    for r in sorted(n for n in names if indeg[n] == 0):
        walk(r, s)
        s = (s + 1) % nq
    return q, out


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    names, edges = parse(sys.argv[1])
    q, out = assign(names, edges)
    short = {n: re.sub(r"^.*?(\w+_kernel|\w+Kernel\w*|fillBuffer\w*).*$", r"\1", demangle(v)) for n, v in names.items()}
    for n in sorted(names):
        print("%3d q%d  %-32s -> %s" % (n, q.get(n, -1), short[n][:32], " ".join("%d" % c for c in out[n])))
