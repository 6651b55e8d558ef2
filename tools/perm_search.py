"""Coordinate search over MIMRL_GRAPH_PERM (GPU box): the order of every fork node's outgoing edges in the captured step graph decides
which hardware queue each child gets (engine_step.hip: graph_postprocess).  usage: python tools/perm_search.py <children per fork, e.g. 4232...>
[passes] [start]  -> gpurun_out/perm_search.txt"""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIG = "0123456789abcdefghijklmn"


def run(perm):
    env = dict(os.environ, MIMRL_GRAPH_REORDER="4", MIMRL_GRAPH_PERM=perm)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--profile-steps", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        return float("inf")


def main():
    kids = [int(c) for c in sys.argv[1]]
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    perm = list(sys.argv[3]) if len(sys.argv) > 3 else ["0"] * len(kids)
    out = open(os.path.join(ROOT, "gpurun_out", "perm_search.txt"), "a")
    best = min(run("".join(perm)), run("".join(perm)))
    print("start", "".join(perm), best, file=out, flush=True)
    for p in range(passes):
        for i, k in enumerate(kids):
            for d in range(k):
                if DIG[d] == perm[i]:
                    continue
                trial = perm[:i] + [DIG[d]] + perm[i + 1:]
                ms = run("".join(trial))
                if ms < best - 0.004:
                    ms = max(ms, run("".join(trial)))
                print(i, "".join(trial), round(ms, 4), file=out, flush=True)
                if ms < best - 0.004:
                    best, perm = ms, trial
        print("pass", p, "".join(perm), best, file=out, flush=True)


if __name__ == "__main__":
    main()
