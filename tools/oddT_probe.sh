#!/bin/bash
# run-to-run reproducibility of the stage-2 gradients at odd / even sequence lengths (round 4: the odd-T BPTT tail bug), eager and captured
for wl in cfg2@49 cfg2@47 cfg2@1 cfg1@49 cfg2@48; do for g in eager graph; do echo "== $wl $g"; python tools/determinism.py $wl 4 2 $( [ $g = graph ] && echo graph ) 2>&1 | tail -3; done; done
