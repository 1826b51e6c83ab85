#!/bin/bash
# round 6: shader clock and package power while the headline runs (is the tower's 0.80 of the nominal MFMA peak a clock under the power limit?)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r6clk
export TMPDIR=/tmp
O=gpurun_out/r6clk
(rocm-smi --showclocks --showpower 2>&1 | head -40) > $O/idle.txt
( for i in $(seq 1 120); do echo "== $(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|mclk|fclk|Power|socclk"; sleep 0.25; done ) > $O/samples_headline.txt &
SMI=$!
timeout 300 python bench.py --steps 4000 --warmup 50 --headline-only > $O/headline.json 2> $O/headline.err
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
grep -iE "sclk|Power" $O/idle.txt | head; echo; grep -i sclk $O/samples_headline.txt | sort | uniq -c | sort -rn | head -12; grep -i "power" $O/samples_headline.txt | sort | uniq -c | sort -rn | head -8
