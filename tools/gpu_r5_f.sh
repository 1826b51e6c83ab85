#!/bin/bash
# round 5, GPU call F (the round's last 3 minutes): bench.py's two-rank path on the one GPU (stub transport), as tests/test_bench_multirank_gpu.py
# runs it -- call E's run of that test file was cut by its 120 s limit after the first test; with AZ_BENCH_TRACE the run says where it stands
O=gpurun_out/r5f; mkdir -p $O; S=$SECONDS
python -c "import torch; print('devices', torch.cuda.device_count())" > $O/import.txt 2>&1; echo "import torch: $((SECONDS-S)) s"
make -s -C tests/rccl_stub > /dev/null 2>&1
export AZHIP_RCCL_LIB=$PWD/tests/rccl_stub/librccl_stub.so HSA_ENABLE_IPC_MODE_LEGACY=0 AZ_BENCH_TRACE=1
for i in 1 2; do
  S=$SECONDS
  timeout 55 python bench.py --gpus 2 --steps 20 --warmup 5 --slots 512 > $O/self_launch_$i.json 2> $O/self_launch_$i.err
  echo "self-launched two-rank run $i: rc $? in $((SECONDS-S)) s"; grep "bench rank" $O/self_launch_$i.err | tail -12
  python -c "import json; d=json.load(open('$O/self_launch_$i.json')); print(d['n_gpus'], d['value'], d['gather'])" 2>&1 | cut -c1-400
done
