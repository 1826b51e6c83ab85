#!/bin/bash
# small-batch tower comparison: the engine's choice vs the forced tower kernels (AZHIP_TOWER)
cd "$(dirname "$0")/.."
for cfg in "128 128 1" "128 128 2" "128 64 1" "256 64 1" "512 64 1" "512 64 2" "1024 64 1" "512 128 1" "1024 128 1"; do
  set -- $cfg
  for tw in ${TOWERS-"" 3 16 32}; do
    echo -n "slots=$1 F=$2 groups=$3 AZHIP_TOWER='$tw': "
    AZHIP_TOWER=$tw python tools/run_config.py --game connect-four --slots $1 --filters $2 --groups $3 --sims 400 --waves 1200 | sed 's/.*waves=[0-9]*: //'
  done
done
