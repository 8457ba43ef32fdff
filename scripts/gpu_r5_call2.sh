# round 5, call 2: library default = f16mx (MX_MAX_DEPTH 3): full -m gpu suite + the default bench line
mkdir -p gpurun_out/r5c2; O=gpurun_out/r5c2
timeout 1200 python -m pytest tests -m gpu -q -rP > $O/pytest_full.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_full.log
grep -E "heavy tail|f16mx|f16x3|synthesis golden|^FAILED|^E  " $O/pytest_full.log | cut -c1-250 | head -200 > $O/pytest_lines.txt
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $?"; tail -2 $O/bench.err; cut -c1-300 $O/bench.log
