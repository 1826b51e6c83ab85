cd /root/repo
WHAT="small_128w_5x128 small_128w_5x128_g2" TAG=r2s bash tools/profile_r2.sh > /dev/null 2>&1
tail -n 1 gpurun_out/r2s/*_stdout.txt
python tools/arena_bench.py 2>&1 | tail -1
python tools/arena_bench.py 2>&1 | tail -1
for cfg in "128 128 1" "128 64 1" "256 128 1" "512 128 1" "32 64 1"; do
  set -- $cfg
  echo -n "slots=$1 F=$2 groups=$3: "
  python tools/run_config.py --game connect-four --slots $1 --filters $2 --groups $3 --sims 600 --waves 2400 | sed 's/.*waves=[0-9]*: //'
done
echo -n "ttt 32 slots: "; python tools/run_config.py --game tictactoe --slots 32 --filters 64 --sims 64 --waves 3200 | sed 's/.*waves=[0-9]*: //'
