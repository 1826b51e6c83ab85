#!/bin/bash
# the whole GPU suite, timed (the driver's limit is 1200 s); failures do not stop the run
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6suite
export TMPDIR=/tmp
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r6suite/suite.log 2>&1
echo "suite rc $? seconds $SECONDS" >> gpurun_out/r6suite/suite.log
grep -E "passed|failed|^FAILED|^ERROR|suite rc" gpurun_out/r6suite/suite.log | tail -30
