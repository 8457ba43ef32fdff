# round 5: the driver's round-end sequence on the end state: -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out/r5final; O=gpurun_out/r5final
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_full.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
s=$(date +%s); timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc $? wall $(( $(date +%s) - s )) s"; cut -c1-200 $O/bench.log
