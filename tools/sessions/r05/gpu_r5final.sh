#!/bin/bash
# the driver's round-end sequence on the final commit: GPU suite, smoke(), the bench line with default flags
mkdir -p gpurun_out/r5final
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/r5final/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r5final/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r5final/bench_default_flags.json 2> gpurun_out/r5final/bench.err; cut -c1-300 gpurun_out/r5final/bench_default_flags.json
