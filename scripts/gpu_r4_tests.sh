#!/bin/bash
# usage: gpu_r4_tests.sh <pytest args...>   (prints the measurement lines of the round-4 tests and the pass / fail summary)
cd "$(dirname "$0")/.." || exit 1
timeout 1200 python -m pytest "$@" -m gpu -q -s --no-header -p no:cacheprovider 2>&1 | grep -E "heavy tail|benchmarked frame|cfg5 full|decoder, planes|render:|passed|failed|FAILED|Error|error|assert|^E " | tail -150
