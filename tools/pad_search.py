"""Coordinate search over MIMRL_GRAPH_PAD (GPU box): which queue offset (0..2 empty-node pads) should the side children of every fork of the
main chain get under MIMRL_GRAPH_REORDER=3?  usage: python tools/pad_search.py <start string> [passes]  -> gpurun_out/pad_search.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(pad, extra=None):
    env = dict(os.environ, MIMRL_GRAPH_REORDER="3", MIMRL_GRAPH_PAD=pad)
    if extra:
        env.update(extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--profile-steps", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        return float("inf")


def main():
    pad = list(sys.argv[1])
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = open(os.path.join(ROOT, "gpurun_out", "pad_search.txt"), "a")
    best = min(run("".join(pad)), run("".join(pad)))
    print("start", "".join(pad), best, file=out, flush=True)
    for p in range(passes):
        for i in range(len(pad)):
            for d in "012":
                if d == pad[i]:
                    continue
                trial = pad[:i] + [d] + pad[i + 1:]
                ms = run("".join(trial))
                if ms < best - 0.003:           # confirm a gain of more than 3 us with a second run
                    ms = max(ms, run("".join(trial)))
                print(i, "".join(trial), ms, file=out, flush=True)
                if ms < best - 0.003:
                    best, pad = ms, trial
        print("pass", p, "".join(pad), best, file=out, flush=True)


if __name__ == "__main__":
    main()
