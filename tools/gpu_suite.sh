#!/bin/bash
# the whole -m gpu tier, failures summarised: tools/gpu_suite.sh [extra pytest args]
timeout 2600 python -m pytest tests/ -q -m gpu --no-header -p no:cacheprovider "$@" 2>&1 | grep -E "^E   .*(Error|assert)|^FAILED|passed|failed" | cut -c1-400 | tail -40
